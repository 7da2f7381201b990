"""HIP-graph capture of a whole render step (forward + loss + backward) for launch-bound loops.

A step of the hot path is ~100 kernel launches of 10-700 us each; issued from Python they cost the host about as long
as the GPU needs to execute them.  `CapturedStep` records one step into a HIP graph (torch.cuda.CUDAGraph underneath)
and replays it: inputs and outputs live in fixed buffers (`copy_` new data into the input tensors before `replay()`),
the intersection capacities are frozen at their planned values during capture and checked afterwards.

    step = CapturedStep(lambda: compute_loss_and_backward(inputs))   # runs `warmup` eager steps, then captures
    for batch in data:
        inputs["target"].copy_(batch["target"], non_blocking=True)
        loss = step.replay()          # the tensor(s) the step function returned, refreshed in place
        step.check()                  # raises if a replayed render overflowed its intersection capacity
"""
import torch

from . import splat


class CapturedStep:
    def __init__(self, fn, warmup: int = 3, streams=()):
        """fn: a function without arguments that runs one step on static tensors and returns tensor(s) to keep.
        warmup: eager calls before the capture (they calibrate the capacity planner and warm the allocator).
        streams: side streams `fn` uses, if any (they must be idle when the capture starts)."""
        self.fn = fn
        for _ in range(max(warmup, 1)):
            fn()
        torch.cuda.synchronize()
        splat.PLANNER.frozen = True
        splat.PLANNER.frozen_log.clear()
        self.graph = torch.cuda.CUDAGraph()
        try:
            # thread_local: other threads (an RCCL watchdog, a data loader) may keep issuing HIP calls meanwhile
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = fn()
        finally:
            splat.PLANNER.frozen = False
        self._log = list(splat.PLANNER.frozen_log)
        splat.PLANNER.frozen_log.clear()

    def replay(self):
        self.graph.replay()
        return self.out

    def check(self):
        """Did any replay overflow a captured intersection capacity?  (One small device->host read per captured render.)"""
        for n_isect, capacity in self._log:
            worst = int(n_isect.max()) if n_isect.numel() else 0
            if worst > capacity:
                raise splat._lib.GoliathHipError(
                    f"a captured render overflowed its intersection capacity ({worst} > {capacity}): raise the plan "
                    f"(splat.PLANNER.set) and capture again")
