import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
n = 1_000_000
bcs = synth.make_barcodes(B, 24, seed=7, min_hamming=6)
seq, off, _ = synth.make_reads(bcs, n, 150, seed=8)
for rate in (0.1, 0.2):
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * B, ids=[str(i) for i in range(B)], max_error_rate=rate)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
    d_bc1 = torch.empty(n, dtype=torch.int32, device=dev)
    with bdx.HipClassifier(cfg) as hc:
        hc.set_read_length_hint(150)
        for _ in range(2):
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr()); hc.sync()
        t = time.perf_counter()
        for _ in range(3):
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr())
        hc.sync()
        dt = (time.perf_counter() - t) / 3
        print(f"B={B} rate={rate} R_env={os.environ.get('BDX_BITPAR_R')} {n / dt / 1e6:.1f} M reads/s {dt * 1e3:.2f} ms [{hc.kernel_path}] {hc.launch_info()}", flush=True)
