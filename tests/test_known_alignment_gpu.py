"""GPU parity of the wave kernel's known-ALIGNMENT class (csrc/bdx_wave.hip, KEND = 3).

Unit-cost configs that want what the known-trim class does not know — per-pass start AND end positions (the Python API's
``want_pass=True``), or the DemuxStats histograms of ``summary=True`` (classification.jl:827-865) — get both positions of every
pass's winner from one more (anchored) bit-vector sweep per pass and read, so the reads the wave kernel settles need no exact
kernel either.  Every test runs the batch with the class and with ``BDX_NO_KALN`` (filter + exact kernel), compares both with
the oracle — per-pass vectors, counters and (summary) the statistics tables against the host-side accumulation of the oracle's
per-pass results.
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


FIELDS = [f"{t}_{k}" for t in ("bc1", "bc2") for k in ("pos_counts", "len_counts", "score_counts", "per_bc_pos_counts",
                                                       "per_bc_len_counts", "per_bc_score_counts")]


def _both(cfg, seq, off, monkeypatch, want_pass=True, expect=True):
    from biodemux_jl_amd.classification import DemuxStats

    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True)
    exp = oc.classify(seq, off)
    exp_st = DemuxStats()
    if cfg.summary:  # the histograms accumulated on the host from the oracle's per-pass results (classification.jl:827-865)
        exp_st.add_pass_outputs(exp, float(cfg.min_delta))
    for aln in (True, False):
        if aln:
            monkeypatch.delenv("BDX_NO_KALN", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_KALN", "1")
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            monkeypatch.delenv("BDX_NO_KALN", raising=False)
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"known-alignment {aln} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (aln, hc.kernel_path)
            assert ("(aln)" in hc.kernel_path) == (aln and expect), hc.kernel_path
            if cfg.summary:
                got_st = DemuxStats()
                got_st.add_device_tables(hc.stats_tables(), cfg)
                for f in FIELDS:
                    assert getattr(got_st, f) == getattr(exp_st, f), (f, aln, hc.kernel_path)
            fuzz.assert_same(hc.classify(seq, off), exp, f"known-alignment {aln}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("kw", [
    dict(trim_side=5), dict(trim_side=3), dict(summary=True), dict(summary=True, trim_side=3, min_delta=0.05),
    dict(summary=True, max_error_rate=0.2), dict(trim_side=5, max_error_rate=0.2, min_delta=0.1), dict(trim_side=3, max_error_rate=0.2),
    dict(summary=True, trim_side=5, max_error_rate=0.15),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_known_alignment_c2_shape(kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=401)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=402, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.15))
    exp = _both(_cfg(bcs, **kw), seq, off, monkeypatch)
    m = exp["bc1"] > 0
    assert m.mean() > 0.3
    assert (exp["pass_start"][m, 0] > 0).mean() > 0.9 and (exp["pass_end"][m, 0] > 0).mean() > 0.9  # (not vacuous)


@pytest.mark.parametrize("t1,t2,summary", [(5, 3, False), (3, 5, False), (None, None, True), (5, None, False), (None, 3, True), (3, 3, True)])
@pytest.mark.parametrize("rate", [0.1, 0.2])
def test_known_alignment_dual(t1, t2, summary, rate, monkeypatch):
    b1 = synth.make_barcodes(24, 24, seed=411)
    b2 = synth.make_barcodes(16, 24, seed=412)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=413, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.03, ins=0.008, dele=0.008,
                                   repeat=dict(frac=0.1))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=rate, trim_side=t1, trim_side2=t2,
                            summary=summary)
    _both(cfg, seq, off, monkeypatch)


def test_known_alignment_read_ends_ties_and_windows(monkeypatch):
    """Winners whose start lies at or in front of the first column (handed on), that end at the last column; ties between
    equally good starts / ends (runs of the barcode's first / last base beside it, partial copies, the same barcode twice);
    reads shorter than a barcode; ref_search_range windows that start inside the read."""
    rng = np.random.Generator(np.random.PCG64(421))
    bcs = synth.make_barcodes(64, 24, seed=421)
    reads = []
    for i in range(18000):
        b = bcs[int(rng.integers(0, 64))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 3))).decode()
        body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(0, 130))))
        kind = i % 9
        if kind == 0:
            reads.append(c + body)
        elif kind == 1:
            reads.append(body + c)
        elif kind == 2:
            reads.append(body[:50] + c[0] * 6 + c + c[-1] * 6 + body[50:])
        elif kind == 3:
            reads.append(body[:20] + c[12:] + c + c[:12] + body[20:])
        elif kind == 4:
            reads.append(c[: int(rng.integers(0, 24))])
        elif kind == 5:
            reads.append(c[int(rng.integers(1, 4)):] + body)
        elif kind == 6:
            reads.append(body[:40] + c + body[40:80] + c + body[80:])
        elif kind == 7:
            reads.append(body + c[: 24 - int(rng.integers(1, 4))])
        else:
            reads.append(body[:60] + c + body[60:])
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=3), dict(trim_side=5), dict(summary=True), dict(summary=True, max_error_rate=0.2, min_delta=0.05)):
        _both(_cfg(bcs, **kw), seq, off, monkeypatch)
    for rs in ("20:end", "1:100", "end-90:end"):
        _both(_cfg(bcs, summary=True, trim_side=3, ref_search_range=H.bdx.parse_dynamic_range(rs)), seq, off, monkeypatch)


def test_known_alignment_low_complexity(monkeypatch):
    rng = np.random.Generator(np.random.PCG64(431))
    bcs = list(dict.fromkeys("".join("AC"[int(x)] for x in rng.integers(0, 2, size=24)) for _ in range(40)))
    reads = []
    for i in range(10000):
        b = bcs[int(rng.integers(0, len(bcs)))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 3))).decode()
        body = "".join("AC"[int(x)] for x in rng.integers(0, 2, size=int(rng.integers(30, 120))))
        k = int(rng.integers(0, len(body)))
        reads.append(body[:k] + c + body[k:])
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=3), dict(summary=True), dict(trim_side=5, min_delta=0.05)):
        cfg = _cfg(bcs, **kw)
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True).classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
            fuzz.assert_same(hc.classify(seq, off), exp, f"{kw} [{hc.kernel_path}]")


def test_known_alignment_not_for_weighted_costs(monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=441)
    seq, off, _ = synth.make_reads(bcs, 15000, 150, seed=442)
    _both(_cfg(bcs, summary=True, mismatch=2, indel=2), seq, off, monkeypatch, expect=False)


@pytest.mark.parametrize("seed", range(300, 330))
def test_fuzz_known_alignment_vs_oracle(seed):
    """fuzz.random_case_band with unit costs forced, per-pass outputs wanted: trim sides, summary, dual, windows, low-complexity
    barcodes, barcodes hanging over the read's ends, concatemers, ragged reads, tiers."""
    cfg, seq, off = fuzz.random_case_band(seed, n_reads=1500)
    cfg.mismatch = 1
    cfg.indel = 1
    oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=True)
    exp = oc.classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, f"seed {seed} [{hc.kernel_path}]")
        assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"
