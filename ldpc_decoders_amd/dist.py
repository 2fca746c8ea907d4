"""Multi-GPU glue: one process per GPU, frames sharded by global frame index, ONE all-reduce of the error
counters per round (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU-only hosts/tests).

There is no data-path collective: frames are independent (src/main.py:37-48 has no cross-frame state).  The
reference has no distributed counterpart; this is the build's own addition.
"""
import os

import numpy as np


class Comm:
    def __init__(self, rank=0, world=1, local_rank=0, backend=None, group=None):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend
        # collectives run whenever a process group exists -- normally world > 1; LDPC_DIST_FORCE_GROUP=1 creates one for a
        # single rank too (exercises RCCL itself on a 1-GPU box)
        self.group = (world > 1) if group is None else bool(group)

    @property
    def is_root(self):
        return self.rank == 0

    def shard(self, frame0, total):
        """Contiguous slice of the global frame range [frame0, frame0+total) owned by this rank -> (start, count)."""
        base, rem = divmod(int(total), self.world)
        cnt = base + (1 if self.rank < rem else 0)
        start = frame0 + self.rank * base + min(self.rank, rem)
        return start, cnt

    def all_reduce_sum(self, counters, async_on_stream=False):
        """Sum an int64 vector over ranks.  Accepts numpy (staged through a tensor) or a torch tensor (in place).

        ``async_on_stream`` (device tensors, RCCL): the collective is only ENQUEUED -- it runs on the communicator's stream behind the
        work already queued on the current stream, and the current stream is made to wait for it; the host does not block."""
        if not self.group:
            return counters
        import torch
        import torch.distributed as td

        if isinstance(counters, np.ndarray):
            dev = "cuda" if self.backend == "nccl" else "cpu"
            t = torch.from_numpy(np.ascontiguousarray(counters, dtype=np.int64)).to(dev)
            td.all_reduce(t, op=td.ReduceOp.SUM)
            return t.cpu().numpy()
        if self.backend == "gloo" and counters.is_cuda:
            t = counters.cpu()
            td.all_reduce(t, op=td.ReduceOp.SUM)
            counters.copy_(t)
            return counters
        if async_on_stream and counters.is_cuda:
            td.all_reduce(counters, op=td.ReduceOp.SUM, async_op=True).wait()  # wait() == stream dependency, not a host sync
            return counters
        td.all_reduce(counters, op=td.ReduceOp.SUM)
        return counters

    def barrier(self):
        if self.group:
            import torch.distributed as td

            td.barrier()

    def max_float(self, x):
        if not self.group:
            return float(x)
        import torch
        import torch.distributed as td

        t = torch.tensor([float(x)], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())


def init_from_env(prefer_gpu=True):
    """Join the process group described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun
    contract); world 1 when they are absent.  Binds this process to GPU LOCAL_RANK."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    backend = None
    import torch

    use_gpu = prefer_gpu and torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    force = os.environ.get("LDPC_DIST_FORCE_GROUP") == "1"
    if world > 1 or force:
        import torch.distributed as td

        # "nccl" is RCCL on ROCm; LDPC_DIST_BACKEND=gloo lets several ranks share one GPU (tests of the N>1 path on a 1-GPU box)
        backend = os.environ.get("LDPC_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not td.is_initialized():
            kw = {}
            if backend == "nccl" and use_gpu:  # bind the communicator to this rank's GPU (also what barrier() then uses)
                kw["device_id"] = torch.device("cuda", local % max(1, torch.cuda.device_count()))
            td.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return Comm(rank, world, local, backend, group=(world > 1 or force))


def finalize():
    try:
        import torch.distributed as td

        if td.is_available() and td.is_initialized():
            td.destroy_process_group()
    except Exception:
        pass
