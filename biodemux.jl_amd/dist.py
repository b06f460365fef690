"""One process per GPU: read shards + the single count all-reduce.

The reference's only cross-worker reduction is merge_stats over per-thread DemuxStats
(src/reporting.jl:1-9, called from core.jl:495/:628).  Across GPUs that is ONE all-reduce
(sum, int64, 4 + B1*max(1,B2) words) of the counter vector the kernel accumulates in HBM —
RCCL over xGMI when the backend is "nccl" (which is RCCL on ROCm), gloo on CPU for tests.
Reads are independent, the barcode table is replicated; there is no data-path collective.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np

from .synth import CHUNK


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = None):
    """Initialises torch.distributed when WORLD_SIZE > 1.  backend defaults to "nccl"
    (= RCCL) when a GPU is visible, else "gloo"."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank()
    if world <= 1:
        return rank, local_rank, world
    if backend is None:
        backend = os.environ.get("BDX_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_first_read(rank: int, reads_per_rank: int) -> int:
    """Start of rank's shard in the global synthetic stream (chunk-aligned, weak scaling:
    every rank owns reads_per_rank reads)."""
    per = (reads_per_rank + CHUNK - 1) // CHUNK * CHUNK
    return rank * per


def shard_bounds(n_reads: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous 1/world slice of an existing batch (strong scaling / file input): SURVEY §8(e)."""
    per = (n_reads + world - 1) // world
    lo = min(n_reads, rank * per)
    return lo, min(n_reads, lo + per)


def init_abi_comm(classifier, rank: int = None, world: int = None) -> bool:
    """Gives ``classifier`` (a HipClassifier) the RCCL communicator of the C-ABI (bdx_comm_init_rank): rank 0
    makes the unique id, torch.distributed — whatever its backend — only carries those 128 bytes to the other
    ranks.  Afterwards ``classifier.allreduce_counts()`` is merge_stats across the GPUs without torch in the
    data path.  Returns False when the job has a single rank (nothing to set up)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return False
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    from .hipabi import comm_unique_id

    box = [comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    classifier.comm_init_rank(box[0], rank, world)
    return True


def allreduce_counts(counts):
    """Sum the counter vector over all ranks (merge_stats across GPUs).  ``counts`` is a torch
    int64 tensor (device tensor -> RCCL, CPU tensor -> gloo) or a numpy array (copied).
    Returns the reduced tensor / array; the input is left untouched so per-rank accumulation
    can continue."""
    import torch
    import torch.distributed as dist

    is_np = isinstance(counts, np.ndarray)
    t = torch.from_numpy(counts.copy()) if is_np else counts.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl" and not t.is_cuda:
            t = t.cuda()
        if dist.get_backend() == "gloo" and t.is_cuda:  # CPU rehearsal of the multi-rank path
            dev = t.device
            t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy() if is_np else t
