#!/usr/bin/env python3
"""Developer survey: device-resident throughput of the BASELINE configs and variants (not the
headline bench).  Each case is also checked against the oracle on a small sample."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import torch

import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth

dev = torch.device("cuda:0")


def run(name, cfg, seq, off, check=3000, outs=("bc1",), reps=3):
    check = int(os.environ.get("CHECK", check))  # reads compared with the oracle (all of `outs`)
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        return
    n = len(off) - 1
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in outs}
    stream = torch.cuda.Stream(dev)
    with bdx.HipClassifier(cfg) as hc:
        hc.set_stream(stream.cuda_stream)
        ptrs = {k: v.data_ptr() for k, v in d.items()}
        ts = []
        with torch.cuda.stream(stream):
            for i in range(reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptrs)
                e1.record(stream)
                torch.cuda.synchronize()
                if i:
                    ts.append(e0.elapsed_time(e1))
        path = hc.kernel_path
        info = hc.launch_info()
    ms = float(np.median(ts))
    k = min(check, n)
    exp = orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq[:off[k]], off[:k + 1])
    ok = all(np.array_equal(d[o][:k].cpu().numpy(), exp[o]) for o in outs)
    print(f"{name:34s} {n / ms / 1e3:9.2f} M reads/s  {ms:9.3f} ms  [{path}, R={info['reads_per_block']}, lds={info['lds_bytes_per_block']}]  oracle-sample {'OK' if ok else 'MISMATCH'}  matched {float((exp['bc1'] > 0).mean()):.2f}", flush=True)


def bscale(n):
    """Throughput against the size of the barcode set (the seed tables and LDS plan scale with it)."""
    C = bdx.DemuxConfig
    for B in (24, 96, 384, 768, 1536):
        bcs = synth.make_barcodes(B, 24, seed=B)
        seq, off, _ = synth.make_reads(bcs, n, 150)
        base = dict(bc_seqs=bcs, bc_lengths_no_N=[24] * B, ids=[str(i) for i in range(B)])
        run(f"B={B} rate0.1", C(**base, max_error_rate=0.1), seq, off, check=1500)
        run(f"B={B} rate0.2", C(**base, max_error_rate=0.2), seq, off, check=1500)
        run(f"B={B} rate0.1 trim5", C(**base, max_error_rate=0.1, trim_side=5), seq, off, check=1500,
            outs=("bc1", "keep_start", "keep_end"))


def lscale(n):
    """Throughput against the read length (tile geometry, index width of the diagonal filter)."""
    C = bdx.DemuxConfig
    bcs = synth.make_barcodes(96, 24)
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)])
    for L in (50, 75, 100, 151, 250, 300):
        seq, off, _ = synth.make_reads(bcs, n, L)
        run(f"L={L} rate0.1", C(**base, max_error_rate=0.1), seq, off, check=1500)
        run(f"L={L} rate0.2", C(**base, max_error_rate=0.2), seq, off, check=1500)
        run(f"L={L} rate0.1 trim3", C(**base, max_error_rate=0.1, trim_side=3), seq, off, check=1500,
            outs=("bc1", "keep_start", "keep_end"))


def main():
    n = int(os.environ.get("N", "2000000"))
    if os.environ.get("BSCALE"):
        return bscale(n)
    if os.environ.get("LSCALE"):
        return lscale(n)
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, n, 150)
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[str(i) for i in range(96)])
    C = bdx.DemuxConfig
    run("C2 rate0.1", C(**base, max_error_rate=0.1), seq, off)
    run("C2 rate0.2 (default)", C(**base, max_error_rate=0.2), seq, off)
    run("C2 rate0.1 min_delta0.1", C(**base, max_error_rate=0.1, min_delta=0.1), seq, off)
    run("C2 rate0.1 trim3", C(**base, max_error_rate=0.1, trim_side=3), seq, off, outs=("bc1", "keep_start", "keep_end"))
    run("C2 rate0.1 trim5", C(**base, max_error_rate=0.1, trim_side=5), seq, off, outs=("bc1", "keep_start", "keep_end"))
    run("C2 rate0.2 trim5", C(**base, max_error_rate=0.2, trim_side=5), seq, off, outs=("bc1", "keep_start", "keep_end"))
    run("C2 rate0.2 summary", C(**base, max_error_rate=0.2, summary=True), seq, off)
    run("demo2 costs (mm1 indel2 r.25 d.15)", C(**base, max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15), seq, off)
    run("hamming rate0.1", C(**base, max_error_rate=0.1, matching_algorithm="hamming"), seq, off)
    run("exact", C(**base, matching_algorithm="exact"), seq, off)
    # C4: dual 24 x 16, trim 5/3
    b1 = synth.make_barcodes(24, 24, seed=1)
    b2 = synth.make_barcodes(16, 24, seed=2)
    s4, o4, _ = synth.make_reads(b1, n, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    c4 = C(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
           bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, trim_side=5, trim_side2=3)
    run("C4 dual 24x16 trim5/3 r0.2", c4, s4, o4, outs=("bc1", "bc2", "keep_start", "keep_end"))
    # C5: 10 kbp reads, 24 variable-length barcodes, window 1:200
    lens = np.random.Generator(np.random.PCG64(5)).integers(16, 33, size=24)
    b5 = synth.make_barcodes(24, 24, seed=5, lengths=lens)
    n5 = max(1000, n // 100)
    s5, o5, _ = synth.make_reads(b5, n5, 10000, plant_lo=0, plant_hi=150)
    c5 = C(bc_seqs=b5, bc_lengths_no_N=[len(b) for b in b5], ids=[str(i) for i in range(24)], max_error_rate=0.2,
           ref_search_range=bdx.parse_dynamic_range("1:200"))
    run("C5 10kbp x24 window1:200 r0.2", c5, s5, o5, check=1000)


if __name__ == "__main__":
    main()
