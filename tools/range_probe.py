#!/usr/bin/env python3
"""Developer probe: C2 shape with a restricted ref_search_range (barcode planted in the first 40 bases), 2 M reads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth
dev = torch.device("cuda:0"); torch.cuda.is_available()
n = int(os.environ.get("N", "2000000"))
bcs = synth.make_barcodes(96, 24)
seq, off, _ = synth.make_reads(bcs, n, 150, plant_lo=0, plant_hi=16)
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
outs = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "keep_start", "keep_end")}
for kw in (dict(), dict(ref_search_range="1:60"), dict(ref_search_range="1:60", trim_side=5), dict(ref_search_range="1:60", max_error_rate=0.2), dict(barcode_start_range="1:20")):
    kw2 = dict(kw)
    for k in ("ref_search_range", "barcode_start_range"):
        if k in kw2: kw2[k] = bdx.parse_dynamic_range(kw2[k])
    kw2.setdefault("max_error_rate", 0.1)
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)], **kw2)
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
        print(f"{kw}  {n / dt / 1e6:9.1f} M reads/s  [{hc.kernel_path}]  oracle-sample {'OK' if ok else 'MISMATCH'}  matched {(outs['bc1'] > 0).float().mean().item():.2f}", flush=True)
