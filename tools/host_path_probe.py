#!/usr/bin/env python3
"""Developer probe: bdx_classify_host on pageable vs page-locked buffers (C2, 10 M reads)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth, hipabi

n = int(os.environ.get("N", "10000000"))
bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
seq, off, _ = synth.make_reads(bcs, n, 150, seed=synth.SEED)
cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], max_error_rate=0.1)
with bdx.HipClassifier(cfg) as hc:
    hc.set_read_length_hint(150)
    hc.classify(seq, off)
    for rep in range(2):
        t = time.perf_counter(); a = hc.classify(seq, off); dt = time.perf_counter() - t
    print(f"pageable in, fresh pageable out: {dt * 1e3:7.2f} ms  {n / dt / 1e6:7.1f} M reads/s")
    pseq = hipabi.pinned_empty(seq.size, np.uint8); pseq[:] = seq
    poff = hipabi.pinned_empty(off.size, np.int64); poff[:] = off
    for rep in range(2):
        t = time.perf_counter(); b = hc.classify(pseq, poff); dt = time.perf_counter() - t
    print(f"pinned in, fresh pageable out:   {dt * 1e3:7.2f} ms  {n / dt / 1e6:7.1f} M reads/s")
    outs = {k: hipabi.pinned_empty(n, np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")}
    o = hipabi.BdxOutputs()
    for k, v in outs.items():
        setattr(o, k, v.ctypes.data)
    for rep in range(3):
        t = time.perf_counter()
        rc = hc.lib.bdx_classify_host(hc.h, pseq.ctypes.data, poff.ctypes.data, n, C.byref(o))
        dt = time.perf_counter() - t
    assert rc == 0 and np.array_equal(outs["bc1"], a["bc1"])
    print(f"pinned in, pinned out (reused):  {dt * 1e3:7.2f} ms  {n / dt / 1e6:7.1f} M reads/s")
    o1 = hipabi.BdxOutputs(); o1.bc1 = outs["bc1"].ctypes.data
    for rep in range(3):
        t = time.perf_counter()
        rc = hc.lib.bdx_classify_host(hc.h, pseq.ctypes.data, poff.ctypes.data, n, C.byref(o1))
        dt = time.perf_counter() - t
    print(f"pinned in, only bc1 out:         {dt * 1e3:7.2f} ms  {n / dt / 1e6:7.1f} M reads/s")
