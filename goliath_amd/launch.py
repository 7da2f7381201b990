"""One-process-per-GPU launcher for the view-parallel path.

The reference has no launcher to mirror: every job is one process on `cuda:0` (ca_code/scripts/run_train.py:32,
scripts/train_bulk/slurm_heads.sh:52-54).  This module is what a `--gpus N` command line needs so that it can never
silently run one rank:

  * `maybe_spawn(n)`     called first thing by a script: when N > 1 ranks are wanted and the process was NOT started by
                         torchrun (no WORLD_SIZE in the environment) it re-executes the script under
                         `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`
                         (the same command the driver uses) and returns the launcher's exit code;
  * `init(n, ...)`       inside a rank: checks the world size against what was asked for, picks the device and the
                         backend and initialises torch.distributed.  RCCL ("nccl") needs one GPU per rank; ranks may share
                         a GPU only when that is asked for explicitly (`share_gpu`, a 1-GPU-box test mode) and then talk
                         over gloo with device tensors -- RCCL refuses two ranks on one device ("Duplicate GPU detected").
                         Too few GPUs without `share_gpu` is an error, never a smaller world.

Everything here is host plumbing; it never touches the kernels.
"""
import os
import socket
import subprocess
import sys
from dataclasses import dataclass

import torch


class LaunchError(RuntimeError):
    pass


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def under_torchrun() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_command(n: int, argv=None, port: int = None):
    """The command line that runs `argv` (default: this process's own script + arguments) as n ranks on this node."""
    argv = list(sys.argv if argv is None else argv)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
            "127.0.0.1", "--master-port", str(port or free_port())] + argv


def maybe_spawn(n: int, argv=None, env=None, timeout=None):
    """N > 1 and not under torchrun: run the script as N ranks and return the launcher's exit code (the caller exits
    with it).  Otherwise return None and the caller carries on as a rank (or as the single process)."""
    if n <= 1 or under_torchrun():
        return None
    e = dict(os.environ if env is None else env)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    e.setdefault("OMP_NUM_THREADS", "4")
    # the host driver only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(spawn_command(n, argv), env=e, timeout=timeout).returncode


@dataclass
class Dist:
    rank: int
    world: int
    local_rank: int
    device: torch.device
    backend: str          # "nccl" (= RCCL), "gloo", or "none" (single process, no process group)
    shared_gpu: bool

    def barrier(self):
        if self.world > 1 or self.backend != "none":
            import torch.distributed as dist

            dist.barrier()
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def max_over_ranks(self, seconds: float) -> float:
        if self.backend == "none":
            return seconds
        import torch.distributed as dist

        t = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t)

    def shutdown(self):
        if self.backend != "none":
            import torch.distributed as dist

            dist.barrier()
            dist.destroy_process_group()


def init(n_requested: int, share_gpu: bool = False, cpu: bool = False, force_group: bool = False) -> Dist:
    """Initialise this process as a rank.  n_requested: the --gpus value (the world size must equal it when it was given
    explicitly, i.e. > 1).  cpu: CPU tensors + gloo (launcher self-tests).  force_group: create the process group even
    for a single rank (puts a live RCCL communicator + watchdog thread in the process: the graph-capture test mode)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n_requested > 1 and world != n_requested:
        raise LaunchError(f"--gpus {n_requested} but WORLD_SIZE={world}: refusing to report a {world}-rank run as "
                          f"{n_requested} GPUs")
    if cpu:
        device, backend, shared = torch.device("cpu"), "gloo", False
    else:
        if not torch.cuda.is_available():
            raise LaunchError("no GPU visible: the product path has no CPU fallback")
        n_dev = torch.cuda.device_count()
        shared = world > n_dev or (share_gpu and world > 1)
        if world > n_dev and not share_gpu:
            raise LaunchError(f"{world} ranks asked for, {n_dev} GPU(s) visible: one GPU per rank is required "
                              f"(--share-gpu lets ranks share a device over gloo -- a functional test mode, not a "
                              f"measurement)")
        dev_index = local % n_dev
        torch.cuda.set_device(dev_index)
        device = torch.device("cuda", dev_index)
        backend = os.environ.get("GOLIATH_DIST_BACKEND") or ("gloo" if shared else "nccl")
    if world == 1 and not force_group:
        return Dist(0, 1, local, device, "none", False)
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29500")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return Dist(rank, world, local, device, backend, shared)
