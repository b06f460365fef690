import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import biodemux_jl_amd as bdx
import bdx_oracle as orc
from biodemux_jl_amd import synth
dev = torch.device("cuda:0")
n = 4000000
b1 = synth.make_barcodes(24, 24, seed=1); b2 = synth.make_barcodes(16, 24, seed=2)
seq, off, _ = synth.make_reads(b1, n, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
cfg = bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24]*24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2, bc_lengths_no_N2=[24]*16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2)
d_seq = torch.from_numpy(seq).to(dev); d_off = torch.from_numpy(off).to(dev)
outs = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1","bc2","keep_start","keep_end")}
with bdx.HipClassifier(cfg) as hc:
    ptr = {k: v.data_ptr() for k, v in outs.items()}
    hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptr); hc.sync()
    t0 = time.perf_counter()
    for _ in range(5): hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptr)
    hc.sync(); dt = (time.perf_counter() - t0) / 5
    exp = orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq[:3000*150], off[:3001])
    ok = all(np.array_equal(outs[k].cpu().numpy()[:3000], exp[k]) for k in outs)
    print(f"dual 24x16 ScoreOnly rate 0.2: {n/dt/1e6:.1f} M reads/s [{hc.kernel_path}] oracle-sample {'OK' if ok else 'MISMATCH'} NO_CARRY={os.environ.get('BDX_NO_CARRY')}")
