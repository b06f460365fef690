#!/usr/bin/env python3
"""PCIe-inclusive throughput: host buffers through bdx_classify_host (H2D + kernel + D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth
n = int(os.environ.get("N", "10000000"))
bcs = synth.make_barcodes(96, 24)
seq, off, _ = synth.make_reads(bcs, n, 150)
cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], max_error_rate=0.1)
with bdx.HipClassifier(cfg) as hc:
    hc.classify(seq[:150 * 1000], off[:1001])
    ts = []
    for _ in range(4):
        t = time.perf_counter(); out = hc.classify(seq, off); ts.append(time.perf_counter() - t)
print(f"host-entry (pageable numpy, H2D+kernel+D2H, 4 outputs): median {np.median(ts)*1e3:.1f} ms per {n} reads -> {n/np.median(ts)/1e6:.1f} M reads/s; best {n/min(ts)/1e6:.1f}")
