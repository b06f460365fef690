import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import helpers as H
from biodemux_jl_amd import synth
bcs = synth.make_barcodes(96, 24, seed=41)
seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=42)
cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24]*96, ids=[str(i) for i in range(96)], max_error_rate=0.2)
oc = H.orc.OracleClassifier(cfg, nthreads=16)
exp = oc.classify(seq, off)
with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
    got = hc.classify(seq, off)
    print(hc.kernel_path, "pairs", hc.pair_launches, "wave", hc.wave_launches)
    print("bc1 equal", np.array_equal(got["bc1"], exp["bc1"]), (got["bc1"] != exp["bc1"]).sum())
import torch
print("torch cuda", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device="cuda:0"); print(x.sum().item())
