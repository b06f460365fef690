#!/usr/bin/env python3
"""Run ONE survey config a few times (for rocprofv3 --kernel-trace --stats): CFG=c4|trim3|trim5|r02"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth
n = int(os.environ.get("N", "2000000"))
which = os.environ.get("CFG", "c4")
C = bdx.DemuxConfig
if which == "c4":
    b1 = synth.make_barcodes(24, 24, seed=1); b2 = synth.make_barcodes(16, 24, seed=2)
    seq, off, _ = synth.make_reads(b1, n, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg = C(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, trim_side=5, trim_side2=3)
    outs = ("bc1", "bc2", "keep_start", "keep_end")
elif which.startswith("long"):  # CFG=long80: 48 barcodes of 80 nt, 300-base reads, trim_side = 3 (the rolling band's probe shape)
    m = int(which[4:])
    bcs = synth.make_barcodes(48, 24, seed=7, lengths=[m] * 48, min_hamming=10)
    seq, off, _ = synth.make_reads(bcs, n, 300, seed=8)
    cfg = C(bc_seqs=bcs, bc_lengths_no_N=[m] * 48, ids=[str(i) for i in range(48)], max_error_rate=0.1, trim_side=3)
    outs = ("bc1", "keep_start", "keep_end")
else:
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, n, 150)
    kw = {"trim3": dict(max_error_rate=0.1, trim_side=3), "trim5": dict(max_error_rate=0.2, trim_side=5),
          "r02": dict(max_error_rate=0.2)}[which]
    cfg = C(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], **kw)
    outs = ("bc1", "keep_start", "keep_end")
dev = torch.device("cuda:0")
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
d = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in outs}
with bdx.HipClassifier(cfg) as hc:
    for _ in range(4):
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in d.items()})
    torch.cuda.synchronize()
    print(which, hc.kernel_path, hc.launch_info())
