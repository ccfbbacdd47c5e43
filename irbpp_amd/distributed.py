"""Multi-GPU sharding of bins: one process per GPU, no data-path collective.

Bins are independent, so rank r simply owns the global bins
[r*n, (r+1)*n).  The trajectory of global bin g in episode e is
(traj_start + g + e*global_bins) % n_traj, which makes every result independent of the world
size.  The only exchange is the episode statistics for logging (trainer.py:215-222): one
all-reduce of four doubles over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl", force: bool = False):
    """(rank, world, local_rank) from the torchrun environment; initialises the process group
    when WORLD_SIZE > 1 -- or, with ``force``, also for a single rank, so that every collective below runs through the
    backend's communicator (RCCL with one rank: the multi-rank code path of bench.py on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard(rank: int, world: int, bins_per_rank: int):
    """-> dict(global_offset, global_bins) for GpuPackingEnv / irbpp_config."""
    assert 0 <= rank < world
    return {"global_offset": rank * bins_per_rank, "global_bins": world * bins_per_rank}


def _active() -> bool:
    """a process group exists (one rank only if init_from_env was forced: the collectives then still go through the backend)"""
    return dist.is_available() and dist.is_initialized()


def _all_reduce(t: torch.Tensor, op) -> torch.Tensor:
    """all_reduce that also works for device tensors under the gloo backend (CPU tests, dry runs)."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op)
    return t


def reduce_totals(totals: torch.Tensor) -> torch.Tensor:
    """Sum [episodes, sum ratio, sum counter, sum reward] over ranks (in place)."""
    if _active():
        _all_reduce(totals, dist.ReduceOp.SUM)
    return totals


def max_over_ranks(value: float, device) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if _active():
        _all_reduce(t, dist.ReduceOp.MAX)
    return float(t.item())


def barrier(device=None):
    """Local device work done, every rank arrived, device idle again (the bracket bench.py times between)."""
    on_gpu = device is not None and torch.device(device).type == "cuda"
    if on_gpu:
        torch.cuda.synchronize(device)
    if _active():
        if on_gpu and dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()])
        else:
            dist.barrier()
    if on_gpu:
        torch.cuda.synchronize(device)


def gather_strings(text: str):
    """One line per rank, gathered on every rank (bench.py: which device did each rank really use)."""
    if not _active():
        return [text]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, text)
    return out
