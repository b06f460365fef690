"""Child process of tests/test_comm_fake_gpu.py: runs with BDX_RCCL_LIB = tests/libbdx_fake_rccl.so, i.e. bdx_comm.cpp
bound to the one-process RCCL stand-in, so that FOUR contexts on device 0 go through bdx_comm_init_all +
bdx_allreduce_counts_all (merge_stats across GPUs, reporting.jl:1-58) — the N > 1 branches a 1-GPU box cannot reach
with the real RCCL: the grouped max-all-reduce that agrees on the statistics tables' height, the growth to it, the
grouped sum of the counter vectors and of every table.  Prints one JSON line; any mismatch is an assertion."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import biodemux_jl_amd as bdx  # noqa: E402
from biodemux_jl_amd import hipabi, synth  # noqa: E402


def tables(hc, reduced):
    t = hc.stats_tables(reduced=reduced)
    return {(p, k): v[0] for p in t for k, v in t[p].items()}


def run(kw, label):
    bcs = synth.make_barcodes(24, 20, seed=71, min_hamming=6)
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[20] * 24, ids=[f"b{i}" for i in range(24)], summary=True, **kw)
    maxlens = [90, 130, 170, 260]  # different longest reads per rank: the ranks must agree on the tables' height
    shards = [synth.make_ragged_reads(bcs, 6000 + 500 * r, 30, maxlens[r], seed=100 + r)[:2] for r in range(4)]
    hcs = [bdx.HipClassifier(cfg) for _ in range(4)]
    try:
        hipabi.comm_init_all(hcs)
        assert [h.comm_size for h in hcs] == [4] * 4 and [h.comm_rank for h in hcs] == [0, 1, 2, 3]
        for h, (seq, off) in zip(hcs, shards):
            h.classify(seq, off)
        own = [h.counts.copy() for h in hcs]
        hipabi.allreduce_counts_all(hcs)
        total = sum(own)
        assert total[0] == sum(len(o) - 1 for _, o in shards)
        for h, c in zip(hcs, own):
            assert np.array_equal(h.reduced_counts, total), label
            assert np.array_equal(h.counts, c), label  # the per-rank vectors stay as they are
        # after the agreement every rank's tables have the same shape; the reduced twins are their element-wise sum
        per_rank = [tables(h, False) for h in hcs]
        for key in per_rank[0]:
            shapes = {t[key].shape for t in per_rank}
            assert len(shapes) == 1, (label, key, shapes)
            want = sum(t[key] for t in per_rank)
            for h in hcs:
                got = tables(h, True)[key]
                assert np.array_equal(got, want), (label, key)
        # the same reads through ONE context give the same counters and tables
        with bdx.HipClassifier(cfg) as one:
            for seq, off in shards:
                one.classify(seq, off)
            assert np.array_equal(one.counts, total), label
            t1 = tables(one, False)
            for key in t1:
                a, b = t1[key], sum(t[key] for t in per_rank)
                rows = min(a.shape[0], b.shape[0])
                assert np.array_equal(a[:rows], b[:rows]) and not a[rows:].any() and not b[rows:].any(), (label, key)
        # a later, longer batch on rank 0 grows its tables: the reduced twins keep their old content, new keys read as zero
        bseq, boff, _ = synth.make_ragged_reads(bcs, 3000, 200, 400, seed=200)
        before = tables(hcs[0], True)
        hcs[0].classify(bseq, boff)
        after = tables(hcs[0], True)
        for key in before:
            rows = before[key].shape[0]
            assert after[key].shape[0] >= rows, (label, key)
            assert np.array_equal(after[key][:rows], before[key]) and not after[key][rows:].any(), (label, key, "growth after the all-reduce")
        # and a second all-reduce picks the new batch up on every rank
        hipabi.allreduce_counts_all(hcs)
        total2 = total.copy()
        total2 += hcs[0].counts - own[0]
        for h in hcs:
            assert np.array_equal(h.reduced_counts, total2), label
        grown = {key: (before[key].shape[0], tables(hcs[3], True)[key].shape[0]) for key in before}
        return {"label": label, "reads": int(total2[0]), "matched": int(total2[1]), "table_rows_before_after": {f"{k[0]}:{k[1]}": v for k, v in grown.items()}}
    finally:
        for h in hcs:
            h.close()


if __name__ == "__main__":
    assert os.environ.get("BDX_RCCL_LIB"), "run me through tests/test_comm_fake_gpu.py"
    out = [run(dict(max_error_rate=0.2), "clean class: fixed-height length table"),
           # a binding start range takes the config out of the clean class: the length table grows with the reads
           run(dict(max_error_rate=0.2, barcode_start_range=bdx.parse_dynamic_range("1:60")), "binding start range: growing length table")]
    print(json.dumps({"ok": True, "cases": out}))
