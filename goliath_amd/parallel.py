"""View-parallel data parallelism for the render path: one process per GPU, torch.distributed over
RCCL (backend "nccl" on ROCm) / gloo in CPU tests.

The reference has no distributed code at all: every job is one GPU and the B views of
a batch are rendered in a Python loop (ca_code/models/rgca.py:119-138).  Views are independent units
of the hot path, so the multi-GPU scheme is:
  * shard: view b of the global batch goes to rank b % world (`shard_views`);
  * no collective on the data path (shade / project / bin / raster never talk to another GPU);
  * one gradient exchange per step for the trainable parameters, as bucketed reduce-scatter +
    all-gather overlapped with backward (`GradSync`) -- on MI355X the 8 GPUs are fully connected by point-to-point xGMI links
    (7 x ~153 GB/s), so a direct reduce-scatter/all-gather uses all 7 links while a single-ring
    all-reduce is bound by one link (740 MB of RGCA gradients = 8.5 ms on a ring
    vs 1.2 ms direct);
  * scalars that steer control flow (loss for the explosion/rollback test of
    ca_code/utils/train.py:189-204, global grad norm for clip_grad_norm_ :214) are all-reduced so
    every rank takes the same branch (`sync_mean`).
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def shard_views(n_views: int, rank: int = None, world_size: int = None) -> List[int]:
    """Indices of the global batch's views owned by `rank` (round-robin: b % world == rank)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_views, world_size))


def shard_batch(batch: dict, rank: int = None, world_size: int = None) -> dict:
    """Slice every [B, ...] tensor (and length-B list) of a batch dict down to this rank's views."""
    sizes = {v.shape[0] for v in batch.values() if torch.is_tensor(v) and v.dim() > 0}
    B = max(sizes) if sizes else 0
    idx = shard_views(B, rank, world_size)
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B:
            out[k] = v[idx]
        elif isinstance(v, (list, tuple)) and len(v) == B:
            out[k] = [v[i] for i in idx]
        else:
            out[k] = v
    return out


def sync_mean(x: torch.Tensor) -> torch.Tensor:
    """All-reduced mean of a scalar/tensor (identity without a process group)."""
    _, w = world()
    if w == 1:
        return x
    y = x.detach().clone()
    dist.all_reduce(y)
    return y / w


class GradSync:
    """Bucketed gradient averaging, overlapped with backward: reduce-scatter + all-gather per bucket.

    params: the trainable parameters (same order on every rank).  bucket_bytes: target bucket size (default 256 MiB:
    large buckets suit 288 GB of HBM and amortise the per-collective latency; xGMI is point-to-point, so the direct
    reduce-scatter / all-gather pair keeps all 7 links busy where a ring all-reduce is bound by one).

    Every bucket owns ONE persistent flat buffer; the `.grad` of its parameters are views into it, so autograd
    accumulates straight into the communication buffer -- no flatten / copy-back passes over the gradients.  Buckets are
    filled in reverse parameter order (the order backward produces gradients); a post-accumulate hook per parameter
    marks arrivals, and a bucket's collectives are launched asynchronously once its last gradient is in AND every bucket
    before it has been launched, while backward keeps computing the earlier layers.  `finish()` (or `sync()`) after
    `backward()` launches what is left (buckets holding parameters that received no gradient contribute zeros for
    them) and waits.

    Launch order.  Collectives are matched across ranks by ISSUE ORDER, so the order must not depend on the local
    autograd schedule: buckets are launched strictly in index order (0, 1, 2 ...) on every rank, from the hooks as far
    as the ready prefix reaches and the rest from `finish()`.  A rank whose graph leaves a parameter unused (or that
    skipped backward because it had no views) therefore launches the same sequence, only later -- the sizes always
    pair up.  What all ranks must still agree on: the parameter list, the bucket size and one `finish()` per step.

    One backward per step.  A second gradient arrival for a parameter before `finish()` (gradient accumulation, a
    `retain_graph` second loss) would add local gradients to a bucket that is already averaged or still in flight:
    it raises.  Accumulate under `no_sync()` (hooks only keep the `.grad` views; nothing is launched) and run the last
    micro-batch outside it, or call `finish()` between the backwards.

    Use `sync.zero_grad()` instead of `optimizer.zero_grad()`: it zeroes the flat buffers and keeps the views.  If a
    `.grad` is replaced behind its back (zero_grad(set_to_none=True), a fresh tensor from autograd), the hook copies it
    into the bucket and re-points `.grad` -- correct, one extra copy for that parameter.

    `launch_all()` / `wait()` split `finish()` for callers that overlap the exchange with work of their own (bench.py
    overlaps step k's exchange with step k+1's compute).  `single_rank_collectives=True` issues the collectives even in
    a process group of one rank (a test hook: it puts a live RCCL communicator on a 1-GPU box).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 256 << 20, average: bool = True,
                 overlap: bool = True, single_rank_collectives: bool = False):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.overlap = overlap
        self.single_rank_collectives = single_rank_collectives
        self.buckets: List[List[torch.nn.Parameter]] = []
        cur, size = [], 0
        for p in reversed(self.params):  # gradients arrive last-layer first
            nbytes = p.numel() * p.element_size()
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat: List[torch.Tensor] = []     # per bucket: padded flat gradient buffer
        self._shard: List[torch.Tensor] = []    # per bucket: this rank's 1/world slice (reduce-scatter output)
        self._view = {}                          # id(param) -> (bucket index, view into the flat buffer)
        self._seen = set()                       # id(param) of the gradients that arrived this step
        self._arrived: List[int] = []
        self._work: List[list] = []
        self._launched: List[bool] = []
        self._next = 0                           # first bucket not launched yet (launches are in index order)
        self._hooks = []
        self._built_for = None
        self._avg_op = False
        self._rs_ag = True                       # reduce-scatter + all-gather available (else: all-reduce)
        self._accumulating = False               # inside no_sync()

    # ---------------------------------------------------------------------------------------------------------
    def _active(self, w: int) -> bool:
        """Is there anything to exchange?  (no process group, or one rank without the test hook: no)"""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return w > 1 or self.single_rank_collectives

    def _build(self, w: int):
        """Allocate the flat buffers (first use, or the world size changed) and point the .grad views into them."""
        self._flat, self._shard, self._view = [], [], {}
        for bi, ps in enumerate(self.buckets):
            n = sum(p.numel() for p in ps)
            n_pad = n + (-n) % w
            flat = torch.zeros(n_pad, dtype=ps[0].dtype, device=ps[0].device)
            self._flat.append(flat)
            self._shard.append(torch.empty(n_pad // w, dtype=flat.dtype, device=flat.device))
            off = 0
            for p in ps:
                v = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
                self._view[id(p)] = (bi, v)
                off += p.numel()
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.overlap and self._active(w):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._built_for = w
        self._avg_op, self._rs_ag = False, True
        if self._active(w) and self._flat:
            self._rs_ag = self._probe_rs_ag(w)
            self._avg_op = self._rs_ag and self._probe_avg(w)
        self._reset()

    def _probe_rs_ag(self, w: int) -> bool:
        """Does the backend reduce-scatter / all-gather tensors of this device?  (RCCL: yes.  gloo: CPU tensors yes;
        device tensors depend on the build -- then the bucket goes through one all-reduce instead.)  Every rank runs
        the same probe on the same kind of tensor, so all ranks reach the same answer."""
        try:
            src = torch.ones(w, dtype=self._flat[0].dtype, device=self._flat[0].device)
            dst = torch.empty(1, dtype=src.dtype, device=src.device)
            dist.reduce_scatter_tensor(dst, src)
            dist.all_gather_into_tensor(src, dst)
            return bool(abs(float(src[0]) - w) < 1e-6)
        except Exception:  # noqa: BLE001 -- an unsupported op must not take the job down
            return False

    def _probe_avg(self, w: int) -> bool:
        """Can the backend average inside the collective (RCCL: ReduceOp.AVG)?  Probed once with a tiny synchronous
        reduce-scatter; any refusal falls back to dividing the bucket before the sum (gloo always does)."""
        if not self.average or dist.get_backend() != "nccl":
            return False
        try:
            src = torch.ones(w, dtype=self._flat[0].dtype, device=self._flat[0].device)
            dst = torch.empty(1, dtype=src.dtype, device=src.device)
            dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.AVG)
            return bool(abs(float(dst) - 1.0) < 1e-6)
        except Exception:  # noqa: BLE001
            return False

    def _reset(self):
        self._arrived = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._work = [[] for _ in self.buckets]
        self._seen = set()
        self._next = 0

    def _adopt(self, p):
        """Make sure p.grad IS the bucket view (copy a foreign gradient tensor in once)."""
        bi, v = self._view[id(p)]
        if p.grad is None:
            v.zero_()
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
        p.grad = v
        return bi

    def _on_grad(self, p):
        bi, _ = self._view[id(p)]
        if self._accumulating:
            self._adopt(p)       # keep .grad a view of the bucket; nothing is counted or launched
            return
        if self._launched[bi]:
            raise RuntimeError(
                "GradSync: a gradient arrived for a parameter whose bucket is already being averaged (a second "
                "backward() before finish() / wait()).  Accumulate under `with sync.no_sync():` and run the last "
                "backward outside it, or call finish() between the backwards.")
        if id(p) in self._seen:
            self._adopt(p)       # bucket still local: autograd accumulated in place, nothing else to do
            return
        self._seen.add(id(p))
        self._adopt(p)
        self._arrived[bi] += 1
        self._drain()

    def _drain(self):
        """Launch the ready prefix: bucket i only after buckets 0..i-1 (the issue order is what pairs collectives
        across ranks, so it may not follow the local autograd completion order)."""
        while self._next < len(self.buckets) and self._arrived[self._next] == len(self.buckets[self._next]):
            self._launch(self._next)

    def _launch(self, bi: int):
        assert bi == self._next, "buckets are launched in index order"
        w = self._built_for
        flat, shard = self._flat[bi], self._shard[bi]
        avg_op = self._avg_op   # RCCL reduces with AVG natively (probed); otherwise divide first, sum in the collective
        if self.average and not avg_op and w > 1:
            flat.div_(w)
        if self._rs_ag:
            h1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM,
                                            async_op=True)   # every rank reduces 1/w of the bucket
            if dist.get_backend() != "nccl":
                h1.wait()   # RCCL orders the two collectives on its stream; gloo's worker threads do not
            h2 = dist.all_gather_into_tensor(flat, shard, async_op=True)   # ... and shares it
            self._work[bi] = [h1, h2]
        else:
            self._work[bi] = [dist.all_reduce(flat, async_op=True)]
        self._launched[bi] = True
        self._next = bi + 1

    # ---------------------------------------------------------------------------------------------------------
    class _NoSync:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.owner._accumulating = True
            return self.owner

        def __exit__(self, *exc):
            self.owner._accumulating = False
            return False

    def no_sync(self):
        """Context for gradient accumulation: backward() calls inside only accumulate into the bucket views.  The
        backward that ends the step runs outside (its hooks launch the buckets), then finish()."""
        return GradSync._NoSync(self)

    def zero_grad(self):
        """Zero the gradients in place (keeps the bucket views; replaces optimizer.zero_grad())."""
        _, w = world()
        if self._built_for != w:
            self._build(w)
        for flat in self._flat:
            flat.zero_()
        self._reset()

    def launch_all(self):
        """After backward: launch, in index order, every bucket that is not in flight yet (does not wait)."""
        _, w = world()
        if not self._active(w):
            return
        if self._built_for != w:
            self._build(w)     # first use without zero_grad(): adopt the existing .grad tensors
        for bi in range(self._next, len(self.buckets)):
            for p in self.buckets[bi]:
                self._adopt(p)   # (a parameter that received no gradient contributes zeros)
            self._launch(bi)

    def wait(self):
        """Wait for the launched collectives (the current stream waits on RCCL's; gloo blocks the host)."""
        for hs in self._work:
            for h in hs:
                h.wait()
        self._reset()

    def finish(self):
        """After backward: launch the buckets that are not in flight yet, wait for all of them."""
        _, w = world()
        if not self._active(w):
            return
        self.launch_all()
        self.wait()

    def sync(self):
        """Average (or sum) .grad over all ranks, in place.  Call after backward, before clipping."""
        self.finish()

    def close(self):
        """Remove the gradient hooks (the .grad tensors stay views of the buckets until they are replaced)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._built_for = None


def global_grad_norm(params: Iterable[torch.nn.Parameter]) -> torch.Tensor:
    """L2 norm of the (already synchronised) gradients -- identical on every rank."""
    sq = [p.grad.detach().float().pow(2).sum() for p in params if p.grad is not None]
    return torch.stack(sq).sum().sqrt() if sq else torch.zeros(())


class ExplosionGuard:
    """The loss-explosion test of the reference's training loop (ca_code/utils/train.py:170-204) for view-parallel ranks:
    the same 32-step loss history and the same rule -- exploded = loss > 10 x mean(history) or non-finite -- evaluated on the
    ALL-REDUCED mean of the ranks' losses, so every rank takes the same decision (reload the checkpoint and skip the step, or
    go on) and no rank enters a collective the others skip.  One `.item()` per step, like the reference.

        guard = ExplosionGuard()
        ...
        if guard.exploded(loss):      # all ranks together
            load_checkpoint(...); guard.reset(); continue
    """

    def __init__(self, history: int = 32, factor: float = 10.0):
        from collections import deque

        self.history = deque(maxlen=history)
        self.history.append(float("inf"))
        self.factor = factor

    def reset(self):
        self.history.clear()
        self.history.append(float("inf"))

    def exploded(self, loss: torch.Tensor) -> bool:
        value = float(sync_mean(loss.detach().float().reshape(())))
        prev = sum(self.history) / len(self.history)
        bad = value > self.factor * prev or value != value or value in (float("inf"), float("-inf"))
        if not bad:
            self.history.append(value)
        return bad


def finish_scrub_and_clip(sync: "GradSync", params: Iterable[torch.nn.Parameter], max_norm: float = 1.0) -> torch.Tensor:
    """What follows `loss.backward()` in the reference's loop (train.py:206-214), view-parallel: finish the gradient
    exchange, zero non-finite gradient entries, clip to `max_norm` by the GLOBAL norm.  After the exchange every rank holds
    the same averaged gradients, so the scrub and the norm are computed locally and are identical everywhere (a norm taken
    before the exchange would differ per rank).  Returns the norm before clipping."""
    params = list(params)
    if sync is not None:
        sync.finish()
    params = [p for p in params if p.grad is not None]
    for p in params:
        torch.nan_to_num_(p.grad, nan=0.0, posinf=0.0, neginf=0.0)
    norm = global_grad_norm(params)
    scale = (max_norm / (norm + 1e-6)).clamp(max=1.0)     # torch.nn.utils.clip_grad_norm_'s rule
    for p in params:
        p.grad.mul_(scale.to(p.grad.device))
    return norm
