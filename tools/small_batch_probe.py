#!/usr/bin/env python3
"""Developer probe: latency of bdx_classify_host for the reference's chunk size (4000 reads, core.jl:5-10) and larger."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth, hipabi

bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
for name, kw in (("C2", dict(max_error_rate=0.1)), ("rate 0.2 trim5", dict(max_error_rate=0.2, trim_side=5)), ("rate 0.2 summary", dict(max_error_rate=0.2, summary=True))):
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], **kw)
    with bdx.HipClassifier(cfg) as hc:
        for n in (4000, 65536, 1000000):
            seq, off, _ = synth.make_reads(bcs, n, 150, seed=7)
            out = {k: np.empty(n, dtype=np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")}
            for _ in range(3):
                hc.classify(seq, off, out=out)
            reps = 200 if n <= 65536 else 20
            t = time.perf_counter()
            for _ in range(reps):
                hc.classify(seq, off, out=out)
            dt = (time.perf_counter() - t) / reps
            print(f"{name:18s} n={n:8d}: {dt * 1e6:9.1f} us per call  {n / dt / 1e6:8.2f} M reads/s  [{hc.kernel_path}]", flush=True)

# the reference runs nthreads() workers concurrently (core.jl:587-599): one context per OS thread
import threading
cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], max_error_rate=0.1)
n = 4000
seq, off, _ = synth.make_reads(bcs, n, 150, seed=7)
for T in (1, 2, 4, 8, 16):
    ctxs = [bdx.HipClassifier(cfg) for _ in range(T)]
    outs = [{k: np.empty(n, dtype=np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")} for _ in range(T)]
    for hc, o in zip(ctxs, outs):
        hc.classify(seq, off, out=o)
    reps = 300
    bar = threading.Barrier(T + 1)

    def work(k):
        bar.wait()
        for _ in range(reps):
            ctxs[k].classify(seq, off, out=outs[k])
        bar.wait()

    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    print(f"{T:2d} worker threads x 4000-read chunks: {T * reps * n / dt / 1e6:8.1f} M reads/s aggregate ({dt / reps * 1e6:7.1f} us per call and worker)", flush=True)
    for hc in ctxs:
        hc.close()
