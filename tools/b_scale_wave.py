#!/usr/bin/env python3
"""Developer probe: barcode-count scaling (1 M reads of 150 bases, 24-nt barcodes), device-resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth
dev = torch.device("cuda:0"); torch.cuda.is_available()
n = int(os.environ.get("N", "2000000"))
rates = [float(x) for x in os.environ.get("RATES", "0.1,0.2").split(",")]
for B in [int(x) for x in os.environ.get("BS", "96,192,384,768").split(",")]:
    bcs = synth.make_barcodes(B, 24, seed=11, min_hamming=6)
    seq, off, _ = synth.make_reads(bcs, n, 150, seed=12)
    d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    for rate in rates:
        cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * B, ids=[str(i) for i in range(B)], max_error_rate=rate)
        with bdx.HipClassifier(cfg) as hc:
            hc.set_read_length_hint(150)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=out.data_ptr()); hc.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=out.data_ptr())
            hc.sync()
            dt = (time.perf_counter() - t0) / 3
            exp = orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq[: 3000 * 150], off[:3001])
            ok = np.array_equal(out.cpu().numpy()[:3000], exp["bc1"])
            print(f"B={B:5d} rate {rate}  {n / dt / 1e6:9.1f} M reads/s  [{hc.kernel_path}]  oracle-sample {'OK' if ok else 'MISMATCH'}", flush=True)
