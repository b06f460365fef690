#!/usr/bin/env python3
"""Developer survey: throughput with barcodes beyond 32 nt (64- and 128-bit sweep words; > 128 nt: unfiltered exact kernel).
500 k reads of 300 bases x 48 barcodes, rate 0.1, device-resident; the first 2000 reads are checked against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth
dev = torch.device("cuda:0"); torch.cuda.is_available()
n = int(os.environ.get("N", "500000"))
for m, kw in ((24, {}), (24, {}), (48, {}), (64, {}), (80, {}), (128, {}), (48, dict(trim_side=5)), (64, dict(trim_side=3)), (80, dict(trim_side=3)), (128, dict(trim_side=3)), (80, dict(trim_side=5, max_error_rate=0.2)), (160, {})):
    nn = n if m <= 128 else 20000
    bcs = synth.make_barcodes(48, 24, seed=7, lengths=[m] * 48, min_hamming=10)
    seq, off, _ = synth.make_reads(bcs, nn, 300, seed=8)
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[m] * 48, ids=[str(i) for i in range(48)], **{**dict(max_error_rate=0.1), **kw})
    d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
    outs = {k: torch.empty(nn, dtype=torch.int32, device=dev) for k in ("bc1", "keep_start", "keep_end")}
    with bdx.HipClassifier(cfg) as hc:
        ptr = {k: v.data_ptr() for k, v in outs.items()}
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), nn, **ptr); hc.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), nn, **ptr)
        hc.sync()
        dt = (time.perf_counter() - t0) / 3
        exp = orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq[: 2000 * 300], off[:2001])
        ok = all(np.array_equal(outs[k].cpu().numpy()[:2000], exp[k]) for k in outs)
        print(f"m={m:4d} {kw}  {nn / dt / 1e6:9.1f} M reads/s  [{hc.kernel_path}]  oracle-sample {'OK' if ok else 'MISMATCH'}", flush=True)
