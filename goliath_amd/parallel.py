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
from typing import Iterable, List, Sequence

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
    counts arrivals and launches the bucket's collectives asynchronously as soon as its last gradient is in, while
    backward keeps computing the earlier layers.  `finish()` (or `sync()`) after `backward()` launches what is left
    (buckets holding parameters that received no gradient contribute zeros for them) and waits.

    Use `sync.zero_grad()` instead of `optimizer.zero_grad()`: it zeroes the flat buffers and keeps the views.  If a
    `.grad` is replaced behind its back (zero_grad(set_to_none=True), a fresh tensor from autograd), the hook copies it
    into the bucket and re-points `.grad` -- correct, one extra copy for that parameter.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 256 << 20, average: bool = True,
                 overlap: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.average = average
        self.overlap = overlap
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
        self._arrived: List[int] = []
        self._work: List[list] = []
        self._launched: List[bool] = []
        self._hooks = []
        self._built_for = None
        self._avg_op = False

    # ---------------------------------------------------------------------------------------------------------
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
        if self.overlap and w > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._built_for = w
        self._avg_op = self._probe_avg(w)
        self._reset()

    def _probe_avg(self, w: int) -> bool:
        """Can the backend average inside the collective (RCCL: ReduceOp.AVG)?  Probed once with a tiny synchronous
        reduce-scatter; any refusal falls back to dividing the bucket before the sum (gloo always does)."""
        if not self.average or w == 1 or dist.get_backend() != "nccl" or not self._flat:
            return False
        try:
            src = torch.ones(w, dtype=self._flat[0].dtype, device=self._flat[0].device)
            dst = torch.empty(1, dtype=src.dtype, device=src.device)
            dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.AVG)
            return bool(abs(float(dst) - 1.0) < 1e-6)
        except Exception:  # noqa: BLE001 -- an unsupported op must not take the job down
            return False

    def _reset(self):
        self._arrived = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._work = [[] for _ in self.buckets]

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
        bi = self._adopt(p)
        self._arrived[bi] += 1
        if self._arrived[bi] == len(self.buckets[bi]) and not self._launched[bi]:
            self._launch(bi)

    def _launch(self, bi: int):
        w = self._built_for
        flat, shard = self._flat[bi], self._shard[bi]
        avg_op = self._avg_op   # RCCL reduces with AVG natively (probed); otherwise divide first, sum in the collective
        if self.average and not avg_op:
            flat.div_(w)
        h1 = dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.AVG if avg_op else dist.ReduceOp.SUM,
                                        async_op=True)   # every rank reduces 1/w of the bucket
        if dist.get_backend() != "nccl":
            h1.wait()   # RCCL orders the two collectives on its stream; gloo's worker threads do not
        h2 = dist.all_gather_into_tensor(flat, shard, async_op=True)   # ... and shares it
        self._work[bi] = [h1, h2]
        self._launched[bi] = True

    # ---------------------------------------------------------------------------------------------------------
    def zero_grad(self):
        """Zero the gradients in place (keeps the bucket views; replaces optimizer.zero_grad())."""
        _, w = world()
        if self._built_for != w:
            self._build(w)
        for flat in self._flat:
            flat.zero_()
        self._reset()

    def finish(self):
        """After backward: launch the buckets that are not in flight yet, wait for all of them."""
        _, w = world()
        if w == 1:
            return
        if self._built_for != w:
            self._build(w)     # first use without zero_grad(): adopt the existing .grad tensors
        for bi, ps in enumerate(self.buckets):
            if not self._launched[bi]:
                for p in ps:
                    self._adopt(p)   # (a parameter that received no gradient contributes zeros)
                self._launch(bi)
        for hs in self._work:
            for h in hs:
                h.wait()
        self._reset()

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
