#!/usr/bin/env python3
"""One-rank RCCL sanity check (the 1-GPU box cannot host more ranks): backend "nccl" initialises and
an int64 all-reduce of a counter vector runs on the same stream discipline bench.py uses."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
from biodemux_jl_amd import dist as bdist
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    c = torch.arange(100, dtype=torch.int64, device="cuda")
    t = bdist.allreduce_counts(c)
torch.cuda.synchronize()
assert torch.equal(t, c) and dist.get_backend() == "nccl"
el = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(el, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl ok:", dist.get_backend(), torch.cuda.get_device_name(0), float(el.item()))
dist.destroy_process_group()
