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
# the same through page-locked input buffers (bdx_host_alloc)
from biodemux_jl_amd.hipabi import pinned_empty
pseq = pinned_empty(len(seq), np.uint8); pseq[:] = seq
poff = pinned_empty(len(off), np.int64); poff[:] = off
with bdx.HipClassifier(cfg) as hc:
    hc.classify(pseq[:150 * 1000], poff[:1001])
    ts = []
    for _ in range(4):
        t = time.perf_counter(); out = hc.classify(pseq, poff); ts.append(time.perf_counter() - t)
print(f"host-entry (pinned inputs via bdx_host_alloc, pageable outputs): median {np.median(ts)*1e3:.1f} ms -> {n/np.median(ts)/1e6:.1f} M reads/s; best {n/min(ts)/1e6:.1f}")
# the C call alone with preallocated outputs (what a Julia caller with reusable buffers sees)
import ctypes as C
from biodemux_jl_amd.hipabi import BdxOutputs
for label, alloc in (("pageable", lambda k, dt: np.empty(k, dtype=dt)), ("pinned", pinned_empty)):
    s_in = alloc(len(seq), np.uint8); s_in[:] = seq
    o_in = alloc(len(off), np.int64); o_in[:] = off
    outs = {k: alloc(n, np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")}
    for k in outs: outs[k][:] = 0
    with bdx.HipClassifier(cfg) as hc:
        o = BdxOutputs()
        for k, v in outs.items():
            setattr(o, k, v.ctypes.data)
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = hc.lib.bdx_classify_host(hc.h, s_in.ctypes.data, o_in.ctypes.data, n, C.byref(o))
            ts.append(time.perf_counter() - t)
        assert rc == 0
    print(f"bdx_classify_host with preallocated {label} buffers (4 outputs): median {np.median(ts[1:])*1e3:.1f} ms -> {n/np.median(ts[1:])/1e6:.1f} M reads/s")
    for k in ("bc1",):
        assert np.array_equal(outs[k], out[k])
