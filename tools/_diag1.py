import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import torch; torch.cuda.is_available()
import helpers as H
from biodemux_jl_amd import synth
bcs = synth.make_barcodes(80, 24, seed=74)
seq, off, _ = synth.make_ragged_reads(bcs, 20000, 0, 230, seed=75, sub=0.05, ins=0.01, dele=0.01)
cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24]*80, ids=[str(i) for i in range(80)], max_error_rate=0.2, trim_side=5, min_delta=0.05)
exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq, off)
lens = np.diff(off)
for hint in (100, 150):
  for env in ({}, {"BDX_POISON": "1"}):
    os.environ.pop("BDX_POISON", None); os.environ.update(env)
    with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
        hc.set_read_length_hint(hint)
        got = hc.classify(seq, off)
        bad = np.flatnonzero(got["bc1"] != exp["bc1"])
        print(hint, env, hc.kernel_path, "bad", bad.size, "lens of bad: min", lens[bad].min() if bad.size else None, "max", lens[bad].max() if bad.size else None,
              "n too long", (lens > hint).sum(), "bad among too long", (lens[bad] > hint).sum(), "rejected", hc.rejected_windows)
