#!/usr/bin/env python3
"""Developer probe: dual barcodes (24 x 16, 24 nt) without trimming (ScoreOnly), 2 M reads, device-resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth
dev = torch.device("cuda:0"); torch.cuda.is_available()
n = int(os.environ.get("N", "2000000"))
b1 = synth.make_barcodes(int(os.environ.get("B1", "24")), 24, seed=1); b2 = synth.make_barcodes(int(os.environ.get("B2", "16")), 24, seed=2)
seq, off, _ = synth.make_reads(b1, n, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
outs = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
for kw in (dict(max_error_rate=0.1), dict(max_error_rate=0.2), dict(max_error_rate=0.1, min_delta=0.05), dict(max_error_rate=0.1, trim_side=5, trim_side2=5)):
    cfg = bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * len(b1), ids=[f"x{i}" for i in range(len(b1))], is_dual=True, bc_seqs2=b2,
                          bc_lengths_no_N2=[24] * len(b2), ids2=[f"y{i}" for i in range(len(b2))], **kw)
    with bdx.HipClassifier(cfg) as hc:
        hc.set_read_length_hint(150)
        ptr = {k: v.data_ptr() for k, v in outs.items()}
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptr); hc.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptr)
        hc.sync()
        dt = (time.perf_counter() - t0) / 3
        exp = orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq[: 3000 * 150], off[:3001])
        ok = all(np.array_equal(outs[k].cpu().numpy()[:3000], exp[k]) for k in outs)
        print(f"dual {kw}  {n / dt / 1e6:9.1f} M reads/s  [{hc.kernel_path}]  oracle-sample {'OK' if ok else 'MISMATCH'}", flush=True)
