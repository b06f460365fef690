#!/usr/bin/env python3
"""Long differential campaign (not part of the default test-suite): many random configs and a few
stress shapes, every filter mode against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
assert torch.cuda.is_available()  # (torch's HIP runtime has to come up before the library's first bdx_create: INTEGRATION.md)
import fuzz
import helpers as H
from biodemux_jl_amd import synth

lo, hi = int(os.environ.get("SEED_LO", "1000")), int(os.environ.get("SEED_HI", "1600"))
bad = 0
t0 = time.time()
paths = {}
for seed in range(lo, hi):
    cfg, seq, off = fuzz.random_case(seed, n_reads=int(os.environ.get("READS", "500")))
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt in ("off", "bitpar", "auto"):
        try:
            with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
                got = hc.classify(seq, off)
                paths[hc.kernel_path] = paths.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"seed {seed} filter {flt} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), f"seed {seed} counters"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"fuzz seeds {lo}..{hi - 1}: {bad} mismatches, paths {paths}, {time.time() - t0:.0f} s", flush=True)

# the seeded variants' domain: many barcodes, budgets around the single-seed / two-intact-pieces switch
lo2, hi2 = int(os.environ.get("SEED2_LO", "5000")), int(os.environ.get("SEED2_HI", "5150"))
paths2 = {}
t0 = time.time()
for seed in range(lo2, hi2):
    cfg, seq, off = fuzz.random_case_many_barcodes(seed)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt, want in (("off", True), ("auto", True), ("auto", False)):
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want, filter=flt) as hc:
                got = hc.classify(seq, off)
                paths2[hc.kernel_path] = paths2.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"seed {seed} filter {flt} want_pass {want} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), f"seed {seed} counters"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"many-barcode seeds {lo2}..{hi2 - 1}: paths {paths2}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# the tiered budgets' domain and its borders
lo3, hi3 = int(os.environ.get("SEED3_LO", "9000")), int(os.environ.get("SEED3_HI", "9300"))
paths3 = {}
t0 = time.time()
for seed in range(lo3, hi3):
    cfg, seq, off = fuzz.random_case_tiers(seed)
    for want in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want)
        exp = oc.classify(seq, off)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                got = hc.classify(seq, off)
                paths3[hc.kernel_path] = paths3.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"tier seed {seed} want_pass {want} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), f"tier seed {seed} counters"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"tier seeds {lo3}..{hi3 - 1}: paths {paths3}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# the diagonal-band DP's domain (exact stage; every barcode with 24 or with 32 bases; traceback / weighted costs)
lo4, hi4 = int(os.environ.get("SEED4_LO", "20000")), int(os.environ.get("SEED4_HI", "20300"))
paths4 = {}
band_runs = 0
t0 = time.time()
for seed in range(lo4, hi4):
    cfg, seq, off = fuzz.random_case_band(seed)
    for want in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want)
        exp = oc.classify(seq, off)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                got = hc.classify(seq, off)
                paths4[hc.kernel_path] = paths4.get(hc.kernel_path, 0) + 1
                band_runs += hc.band_launches > 0
                fuzz.assert_same(got, exp, f"band seed {seed} want_pass {want} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), f"band seed {seed} counters"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
    if (seed - lo4) % 50 == 49:
        print(f"  band seeds .. {seed}: mismatches so far {bad}", flush=True)
print(f"band seeds {lo4}..{hi4 - 1}: {band_runs} runs took the band form, paths {paths4}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# one context, three batches (reads, the same reads permuted, reads again): stale per-read buffer entries of the
# previous batch sit at the same indices for OTHER reads
lo5, hi5 = int(os.environ.get("SEED5_LO", "70000")), int(os.environ.get("SEED5_HI", "70200"))
t0 = time.time()
for seed in range(lo5, hi5):
    gen = (fuzz.random_case_band, fuzz.random_case_tiers, fuzz.random_case_many_barcodes)[seed % 3]
    cfg, seq, off = gen(seed)
    pseq, poff, keep = fuzz.permuted_batch(seq, off, seed)
    for want in (True, False):
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want).classify(seq, off)
        pexp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want).classify(pseq, poff)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                fuzz.assert_same(hc.classify(seq, off), exp, f"reuse seed {seed} first batch [{hc.kernel_path}]")
                fuzz.assert_same(hc.classify(pseq, poff), pexp, f"reuse seed {seed} permuted batch [{hc.kernel_path}]")
                fuzz.assert_same(hc.classify(seq, off), exp, f"reuse seed {seed} first batch again [{hc.kernel_path}]")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"context-reuse seeds {lo5}..{hi5 - 1}: {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# round-3 domains: hundreds of barcodes (wave kernel with sized queues, pairs mode in groups of 128), barcodes of 65..128 nt
lo6, hi6 = int(os.environ.get("SEED6_LO", "80000")), int(os.environ.get("SEED6_HI", "80200"))
paths6 = {}
t0 = time.time()
for seed in range(lo6, hi6):
    cfg, seq, off = fuzz.random_case_wide(seed)
    for want in (True, False):
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want).classify(seq, off)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                got = hc.classify(seq, off)
                paths6[hc.kernel_path] = paths6.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"wide seed {seed} want_pass {want} [{hc.kernel_path}]")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"wide seeds {lo6}..{hi6 - 1}: paths {paths6}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# round-4 domains: the known-trim class (unit costs, any mix of trim sides, no per-pass outputs), the same-diagonal pairs
# variants (indels dearer than mismatches) and the window mode (device entry point: reads much longer than their window)
lo7, hi7 = int(os.environ.get("SEED7_LO", "85000")), int(os.environ.get("SEED7_HI", "85300"))
paths7 = {}
t0 = time.time()
for seed in range(lo7, hi7):
    cfg, seq, off = fuzz.random_case_band(seed)
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x7A1))
    kind = seed % 3
    if kind == 0:    # known-trim
        cfg.mismatch, cfg.indel, cfg.summary = 1, 1, False
        if cfg.trim_side is None and (not cfg.is_dual or cfg.trim_side2 is None):
            cfg.trim_side = [3, 5][int(rng.integers(0, 2))]
    elif kind == 1:  # indels dearer than mismatches: budgets 4 .. 8 in mismatches
        cfg.mismatch = int([1, 1, 2][int(rng.integers(0, 3))])
        cfg.indel = cfg.mismatch * int(rng.integers(2, 4))
        cfg.max_error_rate = float([0.2, 0.25, 0.25, 0.3, 0.34][int(rng.integers(0, 5))]) * cfg.mismatch
        cfg.min_delta = float([0.0, 0.1, 0.15][int(rng.integers(0, 3))]) * cfg.mismatch
    for want in (False, True) if kind != 2 else (False,):
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want).classify(seq, off)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                got = hc.classify(seq, off)
                paths7[hc.kernel_path] = paths7.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"round-4 seed {seed} kind {kind} want_pass {want} [{hc.kernel_path}]")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
    if kind == 2:    # window mode: single pass, ScoreOnly, a short window of long reads, through bdx_classify_device
        import torch
        m = len(cfg.bc_seqs[0])
        L = int([400, 1000, 3000][int(rng.integers(0, 3))])
        wl = int(rng.integers(m + 8, 180))
        rs = ["1:%d" % wl, "%d:%d" % (int(rng.integers(2, 60)), int(rng.integers(61, 200))), "end-%d:end" % (wl - 1), "end-%d:end-%d" % (wl + 20, 21)][int(rng.integers(0, 4))]
        c2 = H.bdx.DemuxConfig(bc_seqs=cfg.bc_seqs, bc_lengths_no_N=cfg.bc_lengths_no_N, ids=cfg.ids, max_error_rate=float([0.1, 0.13, 0.2][int(rng.integers(0, 3))]),
                               min_delta=cfg.min_delta, ref_search_range=H.bdx.parse_dynamic_range(rs))
        lo_p = 0 if not rs.startswith("end") else max(0, L - wl - 25)
        seq2, off2, _ = synth.make_ragged_reads(cfg.bc_seqs, 1200, L // 2, L, seed=seed, plant_lo=lo_p, plant_hi=lo_p + max(1, wl - m), sub=0.03, ins=0.01, dele=0.01)
        exp = H.orc.OracleClassifier(c2, nthreads=16, want_pass=False).classify(seq2, off2)
        n2 = len(off2) - 1
        dev = torch.device("cuda:0")
        d_seq, d_off = torch.from_numpy(seq2).to(dev), torch.from_numpy(off2).to(dev)
        out = {k: torch.empty(n2, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
        try:
            with H.bdx.HipClassifier(c2) as hc:
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n2, **{k: v.data_ptr() for k, v in out.items()})
                hc.sync()
                paths7[hc.kernel_path] = paths7.get(hc.kernel_path, 0) + 1
                for k, v in out.items():
                    got = v.cpu().numpy()
                    assert np.array_equal(got, exp[k]), f"round-4 seed {seed} window mode {rs} L {L}: {k} differs at {np.flatnonzero(got != exp[k])[:5].tolist()} [{hc.kernel_path}]"
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"round-4 seeds {lo7}..{hi7 - 1}: paths {paths7}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


# barcodes of 33 .. 128 bases inside the clean class: the rolling diagonal band of the exact kernel (sg_band_roll, round 4)
lo8, hi8 = int(os.environ.get("SEED8_LO", "86000")), int(os.environ.get("SEED8_HI", "86200"))
paths8 = {}
t0 = time.time()
for seed in range(lo8, hi8):
    cfg, seq, off = fuzz.random_case_band_long(seed)
    for want in (True, False):
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want).classify(seq, off)
        try:
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                got = hc.classify(seq, off)
                paths8[hc.kernel_path] = paths8.get(hc.kernel_path, 0) + 1
                fuzz.assert_same(got, exp, f"long-barcode seed {seed} want_pass {want} [{hc.kernel_path}]")
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
print(f"long-barcode seeds {lo8}..{hi8 - 1}: paths {paths8}, {time.time() - t0:.0f} s; mismatches so far {bad}", flush=True)


def stress(name, bcs, seq, off, **kw):
    global bad
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[sum(c != "N" for c in b) for b in bcs],
                            ids=[str(i) for i in range(len(bcs))], **kw)
    exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq, off)
    for flt in ("off", "auto"):
        try:
            with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
                fuzz.assert_same(hc.classify(seq, off), exp, f"{name} filter {flt} [{hc.kernel_path}]")
                path = hc.kernel_path
        except AssertionError as e:
            bad += 1
            print("MISMATCH", e, flush=True)
    print(f"stress {name}: ok [{path}] matched {float((exp['bc1'] > 0).mean()):.2f}", flush=True)


b = synth.make_barcodes(1500, 20, seed=3, min_hamming=5)
s, o, _ = synth.make_reads(b, 6000, 100, seed=3)
stress("B=1500 m=20 rate0.1", b, s, o, max_error_rate=0.1)
stress("B=1500 m=20 rate0.1 trim3 delta", b, s, o, max_error_rate=0.1, trim_side=3, min_delta=0.06)
b = synth.make_barcodes(3000, 16, seed=4, min_hamming=4)
s, o, _ = synth.make_reads(b, 3000, 80, seed=4)
stress("B=3000 m=16", b, s, o, max_error_rate=0.13)
b = synth.make_barcodes(12, 48, seed=5, min_hamming=12)
s, o, _ = synth.make_reads(b, 4000, 200, seed=5)
stress("m=48 (64-bit sweep words)", b, s, o, max_error_rate=0.15)
stress("m=48 trim5 (64-bit sweep words)", b, s, o, max_error_rate=0.15, trim_side=5)
b = synth.make_barcodes(4, 300, seed=6, min_hamming=60)
s, o, _ = synth.make_reads(b, 600, 700, seed=6)
stress("m=300 (LDS-limited geometry)", b, s, o, max_error_rate=0.1, trim_side=3)
b = synth.make_barcodes(1, 24, seed=7)
s, o, _ = synth.make_reads(b, 5, 150, seed=7)
stress("B=1, 5 reads", b, s, o, max_error_rate=0.2)
b = synth.make_barcodes(96, 24)
s, o, _ = synth.make_reads(b, 70001, 150, seed=8)
stress("70001 reads (ragged last tile)", b, s, o, max_error_rate=0.1)
stress("negative match cost (filters off by domain)", b[:12], s[:150 * 3000], o[:3001], max_error_rate=0.2, match=-1, mismatch=2, indel=3)
stress("iupac-ish alphabet (14 symbols)", ["ACGTRYKMSWACGTRYKMSW", "RYKMSWBDHVACGTACGTAC"], s[:150 * 2000], o[:2001], max_error_rate=0.2)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
