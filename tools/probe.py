#!/usr/bin/env python3
"""Developer probe: time the C2 workload under BDX_DEBUG / BDX_BITPAR_R variants in one process
(interleaved rounds, HIP events).  Not part of the product or the test-suite."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth

n = int(os.environ.get("PROBE_READS", "2000000"))
rate = float(os.environ.get("PROBE_RATE", "0.1"))
variants = [v for v in os.environ.get("PROBE_VARIANTS", "dbg=0;dbg=1;dbg=2;dbg=3").split(";") if v]
rounds = int(os.environ.get("PROBE_ROUNDS", "3"))
bcs = synth.make_barcodes(96, 24)
seq, off, _ = synth.make_reads(bcs, n, 150)
cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], max_error_rate=rate)
dev = torch.device("cuda:0")
d_seq = torch.from_numpy(seq).to(dev)
d_off = torch.from_numpy(off).to(dev)
d_bc1 = torch.empty(n, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(dev)
res = {v: [] for v in variants}
for rnd in range(rounds + 1):
    for v in variants:
        kv = dict(x.split("=") for x in v.split(","))
        os.environ["BDX_DEBUG"] = kv.get("dbg", "0")
        if "R" in kv:
            os.environ["BDX_BITPAR_R"] = kv["R"]
        else:
            os.environ.pop("BDX_BITPAR_R", None)
        hc = bdx.HipClassifier(cfg, filter=kv.get("filter", "auto"))
        hc.set_stream(stream.cuda_stream)
        hc.set_read_length_hint(150)
        with torch.cuda.stream(stream):
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr())  # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr())
            e1.record(stream)
        torch.cuda.synchronize()
        if rnd:
            res[v].append(e0.elapsed_time(e1))
        info = hc.launch_info()
        hc.close()
for v in variants:
    t = np.array(res[v])
    print(f"{v:28s} median {np.median(t):9.3f} ms  min {t.min():9.3f} ms  -> {n / np.median(t) / 1e3:8.2f} M reads/s")
