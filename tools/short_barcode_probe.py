#!/usr/bin/env python3
"""Developer probe: typical short barcode sets (8 / 10 / 12 / 16 nt, 96 barcodes) at the reference's default rate."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_configs as bc
from biodemux_jl_amd import synth
import biodemux_jl_amd as bdx

n = int(os.environ.get("N", "1000000"))
for m, hd in ((8, 3), (10, 3), (12, 4), (16, 5), (20, 6)):
    bcs = synth.make_barcodes(96, m, seed=m, min_hamming=hd)
    seq, off, _ = synth.make_reads(bcs, n, 150, seed=m)
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[m] * 96, ids=[str(i) for i in range(96)])
    for name, kw, outs in (("r0.2", dict(max_error_rate=0.2), ("bc1",)),
                           ("r0.2 trim5", dict(max_error_rate=0.2, trim_side=5), ("bc1", "keep_start", "keep_end")),
                           ("r0.2 trim3", dict(max_error_rate=0.2, trim_side=3), ("bc1", "keep_start", "keep_end")),
                           ("r0.2 summary", dict(max_error_rate=0.2, summary=True), ("bc1",)),
                           ("r0.1", dict(max_error_rate=0.1), ("bc1",))):
        bc.run(f"m={m} {name}", bdx.DemuxConfig(**base, **kw), seq, off, check=1500, outs=outs)
