"""N > 1 path on CPU: two gloo ranks shard the global synthetic stream, classify their shard
(oracle on CPU — the HIP path needs a GPU) and all-reduce the counter vector with
dist.allreduce_counts, exactly as bench.py / a multi-GPU host does over RCCL.  The reduced
vector must equal the single-process counters of the whole stream (merge_stats, reporting.jl:1-9)."""
import os
import subprocess
import sys

import numpy as np

import helpers as H
from biodemux_jl_amd import dist as bdist
from biodemux_jl_amd import synth

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["BDX_ROOT"]); sys.path.insert(0, os.path.join(os.environ["BDX_ROOT"], "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx, bdx_oracle as orc
from biodemux_jl_amd import dist as bdist, synth
rank, local_rank, world = bdist.init_process_group("gloo")
n = synth.CHUNK
bcs = synth.make_barcodes(12, 16, seed=7, min_hamming=5)
cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[16] * 12, ids=[str(i) for i in range(12)], max_error_rate=0.2)
seq, off, _ = synth.make_reads(bcs, n, 60, seed=7, first_read=bdist.shard_first_read(rank, n))
oc = orc.OracleClassifier(cfg, nthreads=2, want_pass=False)
oc.classify(seq, off)
total = bdist.allreduce_counts(torch.from_numpy(oc.counts.copy()))   # torch tensor -> gloo here, RCCL on GPUs
total_np = bdist.allreduce_counts(oc.counts)                          # numpy convenience path
assert np.array_equal(total.numpy(), total_np)
assert oc.counts[0] == n                                              # the input vector is left untouched
if rank == 0:
    np.save(os.environ["BDX_OUT"], total_np)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_two_rank_count_allreduce(tmp_path):
    out = str(tmp_path / "total.npy")
    env = dict(os.environ, BDX_ROOT=H.ROOT, BDX_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                    "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                   check=True, env=env, timeout=300)
    total = np.load(out)
    # single-process reference over both shards
    n = synth.CHUNK
    bcs = synth.make_barcodes(12, 16, seed=7, min_hamming=5)
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[16] * 12, ids=[str(i) for i in range(12)], max_error_rate=0.2)
    seq, off, _ = synth.make_reads(bcs, 2 * n, 60, seed=7)
    oc = H.orc.OracleClassifier(cfg, nthreads=4, want_pass=False)
    oc.classify(seq, off)
    assert np.array_equal(total, oc.counts)
    assert total[0] == 2 * n


def test_shard_helpers():
    assert bdist.shard_first_read(0, 10_000_000) == 0
    assert bdist.shard_first_read(3, 10_000_000) % synth.CHUNK == 0
    assert bdist.shard_first_read(1, 10_000_000) >= 10_000_000
    cover = [bdist.shard_bounds(1001, r, 4) for r in range(4)]
    assert cover[0][0] == 0 and cover[-1][1] == 1001
    assert all(cover[i][1] == cover[i + 1][0] for i in range(3))
