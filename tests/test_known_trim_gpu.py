"""GPU parity of the wave kernel's known-trim class (csrc/bdx_wave.hip, KEND).

Configs of the known-score class with trim sides — ``trim_side = 3`` (keep what precedes the barcode: the reference's START,
classification.jl:910-911, :142-153 tie rule, :310-321 origin order), ``trim_side = 5`` (its END, :912-914), either pass
count, any mix — get verdict and keep range from the wave kernel itself: a trim_side = 3 pass is swept right to left with
the reversed barcode.  Every test runs the batch with the class and with ``BDX_NO_KEND`` (filter + exact kernel) and
compares both with the oracle, counters included.
"""
from __future__ import annotations

import numpy as np
import pytest

import fuzz
import helpers as H
from biodemux_jl_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def _cfg(bcs, **kw):
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"bc{i + 1}" for i in range(len(bcs))],
                max_error_rate=0.1)
    base.update(kw)
    return H.bdx.DemuxConfig(**base)


def _dual_cfg(b1, b2, **kw):
    base = dict(bc_seqs=b1, bc_lengths_no_N=[len(b) for b in b1], ids=[f"x{i}" for i in range(len(b1))], is_dual=True, bc_seqs2=b2,
                bc_lengths_no_N2=[len(b) for b in b2], ids2=[f"y{i}" for i in range(len(b2))], max_error_rate=0.1)
    base.update(kw)
    return H.bdx.DemuxConfig(**base)


def _both(cfg, seq, off, monkeypatch, expect=True, hint=None):
    """verdicts + keep range with and without the known-trim class; both equal the oracle."""
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False)
    exp = oc.classify(seq, off)
    for kend in (True, False):
        if kend:
            monkeypatch.delenv("BDX_NO_KEND", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_KEND", "1")
        with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
            monkeypatch.delenv("BDX_NO_KEND", raising=False)
            if hint is not None:
                hc.set_read_length_hint(hint)
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"known-trim {kend} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (kend, hc.kernel_path)
            assert ("wave(end)" in hc.kernel_path) == (kend and expect), hc.kernel_path
            fuzz.assert_same(hc.classify(seq, off), exp, f"known-trim {kend}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("kw", [
    dict(trim_side=3), dict(trim_side=3, min_delta=0.05), dict(trim_side=3, max_error_rate=0.05), dict(trim_side=3, max_error_rate=0.0),
    dict(trim_side=3, max_error_rate=0.2), dict(trim_side=3, max_error_rate=0.2, min_delta=0.1), dict(trim_side=3, max_error_rate=0.15),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_known_start_c2_shape(kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 50000, 150, seed=195, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.15))
    exp = _both(_cfg(bcs, **kw), seq, off, monkeypatch)
    m = exp["bc1"] > 0
    assert m.mean() > 0.3
    assert (exp["keep_end"][m] < 150).mean() > 0.9  # (not vacuous: nearly every match is trimmed)


@pytest.mark.parametrize("t1,t2", [(5, 3), (3, 5), (3, 3), (5, 5), (None, 3), (3, None), (5, None), (None, 5)])
@pytest.mark.parametrize("rate", [0.1, 0.2])
def test_known_trim_dual_c4_shape(t1, t2, rate, monkeypatch):
    """BASELINE config 4's shape (24 x 16 barcodes, the first planted in 1..40, the second in 100..126) with every mix of
    trim sides; rate 0.2 runs tier 1 -> pairs mode -> general kernels."""
    b1 = synth.make_barcodes(24, 24, seed=201)
    b2 = synth.make_barcodes(16, 24, seed=202)
    seq, off, _ = synth.make_reads(b1, 40000, 150, seed=203, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.03, ins=0.008,
                                   dele=0.008, repeat=dict(frac=0.1))
    cfg = _dual_cfg(b1, b2, max_error_rate=rate, trim_side=t1, trim_side2=t2)
    exp = _both(cfg, seq, off, monkeypatch)
    m = exp["bc1"] > 0
    assert m.mean() > 0.3
    if rate == 0.2:
        with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
            hc.classify(seq, off)
            assert hc.kernel_path.startswith("tier1:wave(end) > pairs(end)"), hc.kernel_path


def test_known_trim_dual_with_min_delta_and_tiers(monkeypatch):
    b1 = synth.make_barcodes(24, 24, seed=211)
    b2 = synth.make_barcodes(16, 24, seed=212)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=213, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.04, ins=0.012,
                                   dele=0.012, repeat=dict(frac=0.2))
    for kw in (dict(max_error_rate=0.2, min_delta=0.05), dict(max_error_rate=0.17, min_delta=0.1), dict(max_error_rate=0.13)):
        _both(_dual_cfg(b1, b2, trim_side=5, trim_side2=3, **kw), seq, off, monkeypatch)


def test_known_start_at_the_read_ends_ties_and_ragged_reads(monkeypatch):
    """Alignments that start at the first column (nothing is kept: (1, 0)) or in front of it (a barcode whose head is cut
    off: the reference's start is <= 0), that end at the last column; reads shorter than a barcode, empty reads; ties between
    equally good starts (a run of the barcode's first base in front of it, partial second copies, the same barcode twice)."""
    rng = np.random.Generator(np.random.PCG64(197))
    bcs = synth.make_barcodes(64, 24, seed=197)
    reads = []
    for i in range(20000):
        b = bcs[int(rng.integers(0, 64))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 3))).decode()
        body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(0, 130))))
        kind = i % 9
        if kind == 0:
            reads.append(c + body)                         # starts at the first column
        elif kind == 1:
            reads.append(body + c)                         # ends at the last column
        elif kind == 2:
            reads.append(body[:50] + c[0] * 6 + c + body[50:])  # a run of the barcode's first base in front of it
        elif kind == 3:
            reads.append(body[:20] + c[12:] + c + body[20:])    # a partial copy in front
        elif kind == 4:
            reads.append(c[: int(rng.integers(0, 24))])    # shorter than the barcode (or empty)
        elif kind == 5:
            reads.append(c[int(rng.integers(1, 4)):] + body)    # the barcode's head cut off at the read's start
        elif kind == 6:
            reads.append(body[:40] + c + body[40:80] + c + body[80:])  # the same barcode twice: the last copy's start
        elif kind == 7:
            reads.append(body + c[: 24 - int(rng.integers(1, 4))])  # the tail cut off at the read's end
        else:
            reads.append(body[:60] + c + body[60:])
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=3), dict(trim_side=3, max_error_rate=0.2, min_delta=0.05)):
        exp = _both(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.4
    assert ((exp["keep_start"] == 1) & (exp["keep_end"] == 0)).sum() > 500


def test_known_start_low_complexity_barcodes(monkeypatch):
    """Two-letter barcodes and reads: many optimal alignments per pair, every tie rule in play."""
    rng = np.random.Generator(np.random.PCG64(198))
    bcs = list(dict.fromkeys("".join("AC"[int(x)] for x in rng.integers(0, 2, size=24)) for _ in range(40)))
    reads = []
    for i in range(12000):
        b = bcs[int(rng.integers(0, len(bcs)))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 3))).decode()
        body = "".join("AC"[int(x)] for x in rng.integers(0, 2, size=int(rng.integers(30, 120))))
        k = int(rng.integers(0, len(body)))
        reads.append(body[:k] + c + body[k:])
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=3), dict(trim_side=5), dict(trim_side=3, min_delta=0.05)):
        oc = H.orc.OracleClassifier(_cfg(bcs, **kw), nthreads=16, want_pass=False)
        exp = oc.classify(seq, off)
        with H.bdx.HipClassifier(_cfg(bcs, **kw), want_pass=False) as hc:
            fuzz.assert_same(hc.classify(seq, off), exp, f"{kw} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)


@pytest.mark.parametrize("rng_str", ["1:60", "20:end", "end-90:end", "5:end-3"])
def test_known_trim_with_ref_search_range(rng_str, monkeypatch):
    """Column windows that start inside the read: an alignment out of the reference's initial column has a start <= 0
    whatever the window's first column is (classification.jl:278-283)."""
    bcs = synth.make_barcodes(96, 24, seed=221)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=222, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.1))
    seq = seq.copy()
    rng = np.random.Generator(np.random.PCG64(223))
    first = {"1:60": 1, "20:end": 20, "end-90:end": 60, "5:end-3": 5}[rng_str]
    for i in range(0, 30000, 5):  # copies that straddle the window's first column
        b = np.frombuffer(bcs[int(rng.integers(0, 96))].encode(), dtype=np.uint8)
        at = first - 1 - int(rng.integers(0, 4))
        if at >= 0:
            seq[off[i] + at: off[i] + at + 24] = b
    r = H.bdx.parse_dynamic_range(rng_str)
    for kw in (dict(trim_side=3), dict(trim_side=5), dict(trim_side=3, max_error_rate=0.2)):
        _both(_cfg(bcs, ref_search_range=r, **kw), seq, off, monkeypatch)
    b2 = synth.make_barcodes(16, 24, seed=224)
    _both(_dual_cfg(bcs[:24], b2, trim_side=3, trim_side2=5, ref_search_range=r, ref_search_range2=r), seq, off, monkeypatch)


def test_known_trim_not_with_positions_it_does_not_know(monkeypatch):
    """Per-pass start positions, or end positions of a trim_side = 3 pass, keep the config on filter + exact kernel; so do
    summary statistics and weighted costs."""
    import torch

    bcs = synth.make_barcodes(96, 24, seed=231)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=232)
    for kw in (dict(trim_side=3, summary=True), dict(trim_side=3, mismatch=2, indel=2)):
        _both(_cfg(bcs, **kw), seq, off, monkeypatch, expect=False)
    cfg = _cfg(bcs, trim_side=3)
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True).classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, hc.kernel_path)
        assert "wave(end)" not in hc.kernel_path
    # the device entry point with pass_end requested and pass_start not: trim_side = 5 knows it, trim_side = 3 does not
    n = len(off) - 1
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    for ts, expect in ((5, True), (3, False)):
        cfg = _cfg(bcs, trim_side=ts)
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True).classify(seq, off)
        out_i = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
        out_p = {k: torch.empty((n, 2), dtype=torch.int32, device=dev) for k in ("pass_end", "pass_bc")}
        with H.bdx.HipClassifier(cfg) as hc:
            hc.set_read_length_hint(150)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in {**out_i, **out_p}.items()})
            hc.sync()
            assert ("wave(end)" in hc.kernel_path) == expect, hc.kernel_path
        for k, v in {**out_i, **out_p}.items():
            got = v.cpu().numpy()
            assert np.array_equal(got, exp[k]), (ts, k, np.flatnonzero((got != exp[k]).reshape(n, -1).any(axis=1))[:5])


def test_known_trim_other_barcode_lengths_and_mixed(monkeypatch):
    for m, rate in ((16, 0.13), (20, 0.1), (28, 0.1), (32, 0.1), (32, 0.2)):
        bcs = synth.make_barcodes(64, m, seed=240 + m, min_hamming=max(4, m // 4))
        seq, off, _ = synth.make_ragged_reads(bcs, 20000, 40, 150, seed=241 + m, sub=0.03, ins=0.01, dele=0.01)
        for ts in (3, 5):
            cfg = _cfg(bcs, trim_side=ts, max_error_rate=rate)
            oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False)
            exp = oc.classify(seq, off)
            with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
                fuzz.assert_same(hc.classify(seq, off), exp, f"m {m} rate {rate} trim {ts} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts)
    # mixed lengths 20..32 in one config
    rng = np.random.Generator(np.random.PCG64(251))
    bcs = synth.make_barcodes(48, 32, seed=252, min_hamming=8)
    bcs = [b[: int(rng.integers(20, 33))] for b in bcs]
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=253, sub=0.03, ins=0.01, dele=0.01)
    for ts in (3, 5):
        _both(_cfg(bcs, trim_side=ts), seq, off, monkeypatch)


@pytest.mark.parametrize("seed", range(200, 230))
def test_fuzz_known_trim_vs_oracle(seed):
    """fuzz.random_case_band with unit costs and no summary forced: trim sides 3 / 5 / none per pass, dual, column windows,
    low-complexity barcodes, barcodes hanging over the read's ends, concatemers, ragged reads, tiers."""
    cfg, seq, off = fuzz.random_case_band(seed, n_reads=1500)
    cfg.mismatch = 1
    cfg.indel = 1
    cfg.summary = False
    if cfg.trim_side is None and (not cfg.is_dual or cfg.trim_side2 is None):
        cfg.trim_side = 3
    oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=False)
    exp = oc.classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, f"seed {seed} [{hc.kernel_path}]")
        assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"


@pytest.mark.parametrize("t1,t2", [(5, 3), (3, 5), (5, None), (None, 3), (None, None)])
def test_dual_carried_pass_vs_oracle(t1, t2, monkeypatch):
    """Tier 1 of a dual config lists a read when ONE of its passes is open; the pass it settled travels with the read (two state
    bits on the list entry, the pass's winning survivor beside it) and the pairs mode only looks for the other pass's barcodes.
    C4's shape and a shape whose second barcode is often missing or damaged, every mix of trim sides incl. none: with and without
    the carry (BDX_NO_CARRY) the outputs and counters equal the oracle's; per-pass outputs switch the carry off by themselves."""
    b1 = synth.make_barcodes(24, 24, seed=801)
    b2 = synth.make_barcodes(16, 24, seed=802)
    kw = dict(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
              bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, trim_side=t1, trim_side2=t2)
    cfg = H.bdx.DemuxConfig(**kw)
    for seed, sub in ((803, 0.02), (804, 0.07)):  # (at 7 % substitutions many passes need the full budget: both kinds of carried pass occur)
        seq, off, _ = synth.make_reads(b1, 30000, 150, seed=seed, sub=sub, ins=sub / 4, dele=sub / 4, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
        for want_pass in (False, True):
            oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
            exp = oc.classify(seq, off)
            for carry in (True, False):
                if carry:
                    monkeypatch.delenv("BDX_NO_CARRY", raising=False)
                else:
                    monkeypatch.setenv("BDX_NO_CARRY", "1")
                with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
                    monkeypatch.delenv("BDX_NO_CARRY", raising=False)
                    for rep in range(2):
                        got = hc.classify(seq, off)
                        fuzz.assert_same(got, exp, f"trims {t1}/{t2} sub {sub} want_pass {want_pass} carry {carry} [{hc.kernel_path}]")
                        if rep == 0:
                            assert np.array_equal(hc.counts, oc.counts), (t1, t2, sub, want_pass, carry)
                    assert hc.rejected_windows == 0
                    assert hc.kernel_path.startswith("tier1:wave"), hc.kernel_path
