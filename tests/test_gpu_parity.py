"""GPU parity tests proper: everything goes through the C-ABI of libbiodemux_hip.so and is
compared bit-for-bit with the oracle (integer outputs exactly; Float64 scores/deltas by bit
pattern).  Run with `pytest -m gpu` on an MI355X."""
import functools
import os

import numpy as np
import pytest

import fuzz
import helpers as H
from biodemux_jl_amd import synth

pytestmark = pytest.mark.gpu

KAT = H.kat_vectors()
run = H.bdx.execute_demultiplexing  # product default: HIP classifier, no injection


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert os.path.exists(H.bdx.LIB_PATH), "HIP extension missing: run __graft_entry__.build()"


# ---- the reference's unit tests, through the kernel ----
@pytest.mark.parametrize("flt", ["off", "auto"])
@pytest.mark.parametrize("vec", KAT, ids=[f"{v['fn']}@{v['src']}" for v in KAT])
def test_kat_hip(vec, flt):
    """filter="off": the exact kernel alone; "auto" (the product default): the fused filter kernel in front."""
    got, expect = H.run_kat(vec, "hip", filter=flt)
    assert got == expect, (vec["src"], flt)


# ---- the reference's integration tests + golden files, through the kernel ----
def test_demo1_R1_golden(tmp_path):
    assert H.scenario_demo1_R1(run, str(tmp_path)) == 24


def test_demo1_R2_golden(tmp_path):
    assert H.scenario_demo1_R2(run, str(tmp_path)) == 24


def test_demo2_golden(tmp_path):
    assert H.scenario_demo2(run, str(tmp_path)) == 76


@pytest.mark.parametrize("algorithm", ["exact", "hamming"])
def test_demo1_R1_other_modes(tmp_path, algorithm):
    assert H.scenario_demo1_modes(run, str(tmp_path), algorithm) == 24


@pytest.mark.parametrize("scenario", H.SCENARIOS_SMALL, ids=[s.__name__ for s in H.SCENARIOS_SMALL])
def test_reference_integration_scenarios(tmp_path, scenario):
    scenario(run, str(tmp_path))


# ---- randomised differential tests vs the oracle ----
@pytest.mark.parametrize("seed", range(60))
def test_fuzz_vs_oracle(seed):
    cfg, seq, off = fuzz.random_case(seed, n_reads=700)
    oc = H.orc.OracleClassifier(cfg, nthreads=8)
    exp = oc.classify(seq, off)
    for flt in ("off", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"seed {seed} filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"


# ---- the seeded variants' domain (48..160 barcodes of 20..32 nt, rates 0.1..0.25): single-piece seeds,
# two intact pieces, plain sweep; every third seed turns half of the reads into concatemers ----
@pytest.mark.parametrize("seed", range(40))
def test_fuzz_many_barcodes_vs_oracle(seed):
    cfg, seq, off = fuzz.random_case_many_barcodes(seed, n_reads=1500)
    for want_pass in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want_pass)
        exp = oc.classify(seq, off)
        for flt in (("off", "auto") if want_pass else ("auto",)):
            with H.bdx.HipClassifier(cfg, want_pass=want_pass, filter=flt) as hc:
                got = hc.classify(seq, off)
                fuzz.assert_same(got, exp, f"seed {seed} filter {flt} pass outputs {want_pass} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_tiers_vs_oracle(seed):
    """Random configs in and around the tiered budgets' domain (fuzz.random_case_tiers)."""
    cfg, seq, off = fuzz.random_case_tiers(seed, n_reads=1200)
    for want_pass in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want_pass)
        exp = oc.classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"seed {seed} pass outputs {want_pass} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_wide_vs_oracle(seed):
    """Random configs in round 3's domains (fuzz.random_case_wide): 100..520 barcodes (wave kernel with queues sized from
    the chance hits, pairs mode in groups of 128 barcodes) and barcodes of 65..128 nt (128-bit sweep words)."""
    cfg, seq, off = fuzz.random_case_wide(seed, n_reads=1200)
    for want_pass in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want_pass)
        exp = oc.classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"seed {seed} pass outputs {want_pass} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"


_BAND_RUNS = {"cases": 0, "band": 0}


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_band_vs_oracle(seed):
    """Random configs in the diagonal-band DP's domain (fuzz.random_case_band): traceback or weighted costs, every
    barcode of the config with the same 8 / 10 / 12 / 16 / 20 / 24 / 32 bases.  With and without per-pass outputs (trim_side 5 without them runs the end-only
    form), and the summary statistics where the config collects them."""
    cfg, seq, off = fuzz.random_case_band(seed, n_reads=1500)
    for want_pass in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want_pass)
        exp = oc.classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"seed {seed} pass outputs {want_pass} [{hc.kernel_path}] band {hc.band_launches}")
            assert np.array_equal(hc.counts, oc.counts), f"seed {seed}: counters"
            _BAND_RUNS["cases"] += 1
            _BAND_RUNS["band"] += hc.band_launches > 0
    if seed == 39:  # the generator must actually reach the band form in most of its cases
        assert _BAND_RUNS["band"] * 2 > _BAND_RUNS["cases"], _BAND_RUNS


def test_band_dp_equals_the_all_rows_dp(monkeypatch):
    """Same batch through the diagonal-band DP and (BDX_NO_BAND) through the all-rows clean-class DP: every output and
    the device statistics agree, for trim_side 3 / 5 / summary at budgets 2 (9 diagonals) and 4 (17 diagonals)."""
    for m, kw in ((24, dict(max_error_rate=0.1, trim_side=3)), (24, dict(max_error_rate=0.2, trim_side=5)),
                  (24, dict(max_error_rate=0.2, summary=True)), (24, dict(max_error_rate=0.17, trim_side=3, min_delta=0.05)),
                  (24, dict(max_error_rate=0.2, mismatch=1, indel=2)), (10, dict(max_error_rate=0.2, trim_side=3)),
                  (8, dict(max_error_rate=0.2, summary=True)), (16, dict(max_error_rate=0.2, trim_side=5)),
                  (20, dict(max_error_rate=0.2, trim_side=3, min_delta=0.05)), (12, dict(max_error_rate=0.17, trim_side=5)),
                  (32, dict(max_error_rate=0.1, trim_side=3))):
        bcs = synth.make_barcodes(96, m, seed=400 + m, min_hamming=max(3, m // 4))
        seq, off, _ = synth.make_reads(bcs, 40000 if m >= 16 else 12000, 150, seed=400 + m, repeat=dict(frac=0.1))
        cfg = _c2_config(bcs, **kw)
        outs = {}
        for band in (True, False):
            if band:
                monkeypatch.delenv("BDX_NO_BAND", raising=False)
            else:
                monkeypatch.setenv("BDX_NO_BAND", "1")
            with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
                outs[band] = hc.classify(seq, off)
                assert (hc.band_launches > 0) == band, (kw, hc.kernel_path)
                if cfg.summary:
                    outs[band]["tabs"] = hc.stats_tables()
        for k, v in outs[True].items():
            if k == "tabs":
                for p_ in v:
                    for name in v[p_]:
                        assert np.array_equal(v[p_][name][0], outs[False]["tabs"][p_][name][0]), (kw, name)
            else:
                assert np.array_equal(v, outs[False][k], equal_nan=True) if v.dtype.kind == "f" else np.array_equal(v, outs[False][k]), (kw, k)
        nall = len(off) - 1
        exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq[:off[5000]], off[:5001])
        fuzz.assert_same({k: (v[:5000] if v.shape[0] == nall else v[:10000]) for k, v in outs[True].items() if k != "tabs"}, exp, str(kw))


@pytest.mark.parametrize("m,kw", [
    (8, dict(max_error_rate=0.2)),
    (8, dict(max_error_rate=0.25, min_delta=0.1)),
    (10, dict(max_error_rate=0.2)),
    (10, dict(max_error_rate=0.2, min_delta=0.05)),
    (12, dict(max_error_rate=0.2)),
    (12, dict(max_error_rate=0.25, min_delta=0.09)),
    (16, dict(max_error_rate=0.2)),
], ids=lambda x: str(x) if isinstance(x, int) else ",".join(f"{k}={v}" for k, v in x.items()))
def test_short_barcodes_many_genuine_candidates(m, kw):
    """Short barcodes at the reference's default rate: a read holds MANY genuine candidates (a 10-mer within two edits
    of a random 150-base read is common), so the reducer replay takes up to 32 survivors per read and pass instead of
    four.  Every filter mode against the oracle, with and without per-pass outputs, dual as well."""
    bcs = synth.make_barcodes(96, m, seed=200 + m, min_hamming=3)
    seq, off, _ = synth.make_ragged_reads(bcs, 20000, 60, 160, seed=200 + m, sub=0.03, ins=0.005, dele=0.005, repeat=dict(frac=0.1))
    cfg = _c2_config(bcs, **kw)
    exp = _all_filters_agree(cfg, seq, off)
    assert (exp["bc1"] > 0).mean() > 0.3
    with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
        got = hc.classify(seq, off)
        assert np.array_equal(got["bc1"], exp["bc1"])
    b2 = synth.make_barcodes(24, m, seed=300 + m, min_hamming=3)
    s2, o2, _ = synth.make_reads(bcs, 8000, 150, seed=300 + m, plant_lo=0, plant_hi=40, second=(b2, 90, 150 - m))
    cfgd = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[m] * 96, ids=[f"x{i}" for i in range(96)], is_dual=True,
                             bc_seqs2=b2, bc_lengths_no_N2=[m] * 24, ids2=[f"y{i}" for i in range(24)], **kw)
    _all_filters_agree(cfgd, s2, o2)


def _c2_config(bcs, **kw):
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"bc{i + 1}" for i in range(len(bcs))],
                max_error_rate=0.1)
    base.update(kw)
    return H.bdx.DemuxConfig(**base)


@pytest.mark.parametrize("kw", [
    dict(),                                               # C2: ScoreOnly, no_delta, allowed_error 2
    dict(max_error_rate=0.2),                             # reference default rate: allowed_error 4
    dict(max_error_rate=0.2, min_delta=0.1),              # with_delta reducer
    dict(max_error_rate=0.2, trim_side=3),                # traceback, rightmost ties
    dict(max_error_rate=0.2, trim_side=5, summary=True),
    dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15),  # demo2's costs
    dict(matching_algorithm="hamming", max_error_rate=0.15),
    dict(matching_algorithm="exact"),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "C2")
def test_c2_shape_vs_oracle(kw):
    """BASELINE config 2's shape (150 bp x 96 barcodes of 24) at a size the oracle finishes in seconds."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 60000, 150)
    cfg = _c2_config(bcs, **kw)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt in ("off", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"{kw} filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)


def test_c4_dual_trim_vs_oracle():
    """BASELINE config 4: dual 24 x 16 barcodes, bc1 planted early, bc2 late, trim 5 / 3."""
    b1 = synth.make_barcodes(24, 24, seed=1)
    b2 = synth.make_barcodes(16, 24, seed=2)
    seq, off, _ = synth.make_reads(b1, 40000, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True,
                            bc_seqs2=b2, bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)],
                            max_error_rate=0.2, trim_side=5, trim_side2=3)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, "C4")
        assert np.array_equal(hc.counts, oc.counts)
    assert (exp["bc1"] > 0).mean() > 0.5  # the case is not vacuous


def test_c5_long_reads_window_vs_oracle():
    """BASELINE config 5: 10 kbp reads, 24 variable-length barcodes, ref_search_range 1:200."""
    lens = np.random.Generator(np.random.PCG64(5)).integers(16, 33, size=24)
    bcs = synth.make_barcodes(24, 24, seed=5, lengths=lens)
    seq, off, _ = synth.make_reads(bcs, 3000, 10000, plant_lo=0, plant_hi=150)
    cfg = _c2_config(bcs, max_error_rate=0.2, ref_search_range=H.bdx.parse_dynamic_range("1:200"))
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, "C5")
    assert (exp["bc1"] > 0).mean() > 0.5


# ---- edge cases ----
def test_empty_and_tiny_reads():
    cfg = _c2_config(["ACGTACGT", "TTTTCCCC"], max_error_rate=0.25, trim_side=3)
    reads = ["", "A", "ACGTACGT", "ACGTACG", "NNNNNNNN", "acgtacgt", "TTTTCCCC" * 3, "", "ACGTACGTTTTTCCCC"]
    seq, off = H.bdx.pack_reads(reads)
    exp = H.orc.OracleClassifier(cfg).classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, "edge")
        assert hc.classify(np.zeros(0, np.uint8), np.zeros(1, np.int64))["bc1"].shape == (0,)


def test_offsets_need_not_start_at_zero():
    bcs = synth.make_barcodes(8, 12, seed=3, min_hamming=4)
    seq, off, _ = synth.make_ragged_reads(bcs, 5000, 20, 90, seed=3)
    cfg = _c2_config(bcs, max_error_rate=0.2)
    exp = H.orc.OracleClassifier(cfg, nthreads=8).classify(seq, off)
    k = 1234
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        got = hc.classify(seq, off[k:])  # same byte buffer, offsets of a sub-range
    fuzz.assert_same(got, {key: v[k:] for key, v in exp.items()}, "sub-range")


def test_error_paths():
    with pytest.raises(H.bdx.BdxError, match="trim_side must be 3 or 5"):
        H.bdx.HipClassifier(_c2_config(["ACGT"], trim_side=4))
    with pytest.raises(H.bdx.BdxError, match="empty"):
        H.bdx.HipClassifier(_c2_config(["ACGT", ""]))
    with pytest.raises(H.bdx.BdxError, match="indel"):
        H.bdx.HipClassifier(_c2_config(["ACGT"], indel=0))


@pytest.mark.parametrize("n", [3000, 300_000, 1_200_000])
def test_offsets_that_decrease_are_refused(n):
    """bdx_classify_host checks the offsets of EVERY batch size (small: one pass; large: a threaded scan that also finds
    the longest read): a negative length must never reach the kernels' address arithmetic."""
    bcs = synth.make_barcodes(24, 24, seed=5)
    seq, off, _ = synth.make_reads(bcs, n, 100, seed=6)
    for kw in (dict(), dict(max_error_rate=0.2, trim_side=5)):
        with H.bdx.HipClassifier(_c2_config(bcs, **kw)) as hc:
            bad = off.copy()
            bad[n // 2] += 150  # read n/2 - 1 grows over its successor, read n/2 gets a negative length
            with pytest.raises(H.bdx.BdxError, match="not non-decreasing"):
                hc.classify(seq, bad)
            good = hc.classify(seq, off)  # the context stays usable
            assert good["bc1"].shape[0] == n


def test_counts_accumulate_and_reset():
    bcs = synth.make_barcodes(16, 16, seed=9, min_hamming=5)
    seq, off, _ = synth.make_reads(bcs, 20000, 80, seed=9)
    cfg = _c2_config(bcs, max_error_rate=0.2)
    with H.bdx.HipClassifier(cfg) as hc:
        hc.classify(seq, off)
        c1 = hc.counts
        hc.classify(seq, off)
        assert np.array_equal(hc.counts, 2 * c1)
        hc.reset_counts()
        assert hc.counts.sum() == 0
    assert c1[0] == 20000 and c1[1] + c1[2] + c1[3] == c1[0] and c1[4:].sum() == c1[1]


# ---- full BASELINE size: size-independent properties + a sampled oracle check ----
def test_c2_full_size_properties():
    """10 M reads x 96 barcodes (BASELINE config 2) on device-resident buffers:
    (1) counters are consistent with the per-read verdicts (a checksum of checksums),
    (2) classifying a permutation of the reads gives the permuted verdicts,
    (3) a re-run is bit-identical (idempotence), (4) a strided 1-in-200 sample equals the oracle."""
    import torch

    n = 10_000_000
    bcs = synth.make_barcodes(96, 24)
    seq, off, truth = synth.make_reads(bcs, n, 150)
    cfg = _c2_config(bcs)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_bc1 = torch.empty(n, dtype=torch.int32, device=dev)
    d_bc1b = torch.empty(n, dtype=torch.int32, device=dev)
    with H.bdx.HipClassifier(cfg) as hc:
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr())
        hc.sync()
        counts = hc.counts
        bc1 = d_bc1.cpu().numpy()
        assert counts[0] == n
        assert counts[1] == int((bc1 > 0).sum()) and counts[2] == int((bc1 == 0).sum())
        assert np.array_equal(counts[4:], np.bincount(bc1[bc1 > 0] - 1, minlength=96))
        # idempotence
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1b.data_ptr())
        hc.sync()
        assert torch.equal(d_bc1, d_bc1b)
        # permutation (reversal of read order): same bytes rows, reversed
        d_rev = d_seq.view(n, 150).flip(0).contiguous().view(-1)
        torch.cuda.synchronize()  # torch's stream produced d_rev; the classifier runs on its own stream
        hc.classify_device(d_rev.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1b.data_ptr())
        hc.sync()
        assert torch.equal(d_bc1.flip(0), d_bc1b)
    # an independent path over ALL 10 M reads: the plain sweep of every (read, barcode) pair (no seeds, no windows)
    d_bc1c = torch.empty(n, dtype=torch.int32, device=dev)
    with H.bdx.HipClassifier(cfg, filter="bitpar") as hp:
        hp.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1c.data_ptr())
        hp.sync()
        assert hp.kernel_path == "bitpar+verify"
        assert torch.equal(d_bc1, d_bc1c) and np.array_equal(hp.counts, counts)
    # planted barcodes are recovered (sanity of the workload itself)
    planted = truth > 0
    assert (bc1[planted] == truth[planted]).mean() > 0.9
    # sampled oracle check
    idx = np.arange(0, n, 200)
    sseq = seq.reshape(n, 150)[idx].reshape(-1)
    soff = np.arange(len(idx) + 1, dtype=np.int64) * 150
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(sseq, soff)
    assert np.array_equal(bc1[idx], exp["bc1"])


# ---- seeded (q-gram) path: fallbacks must stay lossless ----
def _all_filters_agree(cfg, seq, off, expect_path=None, hint=None):
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    paths = {}
    for flt in ("off", "bitpar", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            if hint is not None:
                hc.set_read_length_hint(hint)
            got = hc.classify(seq, off)
            paths[flt] = hc.kernel_path
            fuzz.assert_same(got, exp, f"filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)
    if expect_path:  # (known-score configs run tier 1 in front: "tier1:qgram+bitpar > <full-budget path>"; split configs that
        # qualify for the wave kernel as their filter report "wave+verify" / (tier 0 of tiered ones) "pairs+verify")
        assert paths["auto"].endswith(expect_path) or paths["auto"].endswith("wave+verify") or paths["auto"].endswith("pairs+verify"), paths
    return exp


def test_seed_path_is_taken_for_c2():
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 30000, 150)
    _all_filters_agree(_c2_config(bcs), seq, off, expect_path="qgram+bitpar+verify")


def test_seed_queue_overflow_low_complexity():
    """Low-complexity barcodes and reads make almost every position a seed hit: the hit and pair
    queues overflow and the affected reads must fall back to sweeping every barcode."""
    rng = np.random.Generator(np.random.PCG64(11))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "AAAACCCCGGGGTTTTAAAACCCC", "ACGT" * 6, "T" * 24, "TTTTTTTTAAAAAAAAGGGGGGGG"]
    bcs += synth.make_barcodes(25, 24, seed=11)
    motifs = ["A", "AC", "ACG", "ACGT", "T", "TTTTAAAA", "AAAACCCCGGGGTTTT"]
    reads = []
    for i in range(6000):
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:150])
        for _ in range(int(rng.integers(0, 4))):  # a few point mutations
            s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(), dict(min_delta=0.05), dict(trim_side=3)):
        exp = _all_filters_agree(_c2_config(bcs, **kw), seq, off, expect_path="qgram+bitpar+verify")
    assert (exp["bc1"] != 0).mean() > 0.3


def test_known_class_handover_many_survivors():
    """A family of near-identical barcodes leaves more than four survivors per read: the fused
    kernel cannot replay the reducer for such reads and hands them (with their candidate masks) to
    the exact kernel's list mode.  Results must not depend on which kernel gave the verdict."""
    base = synth.make_barcodes(1, 24, seed=77)[0]
    fam = [base]
    for i in range(9):  # one substitution each, at different positions
        j = 2 * i + 1
        fam.append(base[:j] + ("A" if base[j] != "A" else "C") + base[j + 1:])
    bcs = fam + synth.make_barcodes(22, 24, seed=78)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=79)
    for kw in (dict(), dict(min_delta=0.05), dict(max_error_rate=0.13)):
        exp = _all_filters_agree(_c2_config(bcs, **kw), seq, off, expect_path="qgram+bitpar+verify")
    assert (exp["bc1"] > 0).mean() > 0.3


@pytest.mark.parametrize("hint", [40, 150, 400])
def test_seed_ragged_reads_and_wrong_hint(hint):
    """A read-length hint that is too small (tiles not staged / tails beyond the planned group
    count) or too large must not change any result."""
    bcs = synth.make_barcodes(48, 24, seed=21)
    seq, off, _ = synth.make_ragged_reads(bcs, 20000, 0, 260, seed=21)
    _all_filters_agree(_c2_config(bcs), seq, off, hint=hint)


def test_seed_with_wildcard_and_short_barcodes():
    """Barcodes with N under NScoring and barcodes too short for a seed are swept unconditionally."""
    bcs = synth.make_barcodes(20, 24, seed=31)
    bcs[3] = bcs[3][:5] + "NN" + bcs[3][7:]
    bcs[7] = bcs[7][:10] + "N" + bcs[7][11:]
    bcs += ["ACGTTGCA", "TTGACCAGT"]  # too short for pieces of >= 5 at k >= 1
    nn = [sum(c != "N" for c in b) for b in bcs]
    seq, off, _ = synth.make_reads([b.replace("N", "G") for b in bcs], 20000, 120, seed=31)
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=nn, ids=[str(i) for i in range(len(bcs))],
                            max_error_rate=0.13, nindel=1)
    _all_filters_agree(cfg, seq, off)


# ---- two-intact-pieces ("diagonal") seeding: budgets too large for single seeds (default rate 0.2) ----
DIAG = "qgram2+bitpar+verify"


@pytest.mark.parametrize("kw", [
    dict(max_error_rate=0.2),                                   # the reference's default: kb = 4, 6 pieces of 4
    dict(max_error_rate=0.2, min_delta=0.1),
    dict(max_error_rate=0.2, trim_side=5),                      # split mode: the sweeps also record the column windows
    dict(max_error_rate=0.2, trim_side=3, summary=True),
    dict(max_error_rate=0.17),                                  # kb = 4 as well (floor(4.08))
    dict(max_error_rate=0.2, mismatch=2, indel=2, min_delta=0.05),  # cmin = 2 -> kb = 2: single seeds take it
    dict(max_error_rate=0.2, matching_algorithm="hamming"),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_diag_c2_shape(kw):
    bcs = synth.make_barcodes(96, 24, seed=41)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=42)
    expect = None if "mismatch" in kw else DIAG
    exp = _all_filters_agree(_c2_config(bcs, **kw), seq, off, expect_path=expect)
    assert (exp["bc1"] > 0).mean() > 0.5


@pytest.mark.parametrize("rate", [0.2, 0.17])
@pytest.mark.parametrize("min_delta", [0.0, 0.05, 0.1])
def test_diag_same_barcode_planted_twice(rate, min_delta):
    """Concatemer / chimeric reads: the same barcode twice (1 vs 2, 0 vs 3, 0 vs 0 edits, >= 2 kb + 3 columns
    apart), two different barcodes in one read, a barcode and an overlapping shifted copy.  The diagonal variant
    sweeps one window per cluster of seed diagonals, so one (read, barcode) pair may deliver several unit
    distances; the reference evaluates every barcode once (classification.jl:676-711) — the reducer replay
    must see one entry per barcode (its minimum), else the second copy becomes sub_min and a clean match
    turns ambiguous."""
    bcs = synth.make_barcodes(96, 24, seed=41)
    seq, off, _ = synth.make_reads(bcs, 12000, 150, seed=43, repeat=dict(frac=0.7))
    cfg = _c2_config(bcs, max_error_rate=rate, min_delta=min_delta)
    exp = _all_filters_agree(cfg, seq, off, expect_path=DIAG)
    assert (exp["bc1"] > 0).mean() > 0.4
    if min_delta > 0:
        # the case is not vacuous: reads with the same barcode twice are matches with delta = Inf in the reference
        assert np.isinf(exp["pass_delta"][:, 0][exp["bc1"] > 0]).mean() > 0.3


def test_diag_repeats_ragged_trim_and_hamming():
    """The same concatemer reads through the split path (trimming: windows of one barcode are united) and
    :hamming, on ragged reads."""
    bcs = synth.make_barcodes(80, 24, seed=51)
    seq, off, _ = synth.make_ragged_reads(bcs, 9000, 60, 152, seed=52, repeat=dict(frac=0.6, other=0.4))
    for kw in (dict(max_error_rate=0.2, trim_side=3, min_delta=0.05), dict(max_error_rate=0.2, trim_side=5),
               dict(max_error_rate=0.2, matching_algorithm="hamming", min_delta=0.05), dict(min_delta=0.1)):
        _all_filters_agree(_c2_config(bcs, **kw), seq, off)


def test_diag_variable_lengths_and_dual():
    """Per-barcode budgets and piece counts differ (24 nt: 6 pieces, 28 / 29 nt: 7, 32 nt: 8; a few of 26 nt
    whose pieces would be too short are swept unconditionally); second pass on the same read."""
    lens = np.random.Generator(np.random.PCG64(43)).choice([24, 28, 29, 32, 24, 28, 32, 26, 24, 29], size=60)
    b1 = synth.make_barcodes(60, 24, seed=43, lengths=lens)
    b2 = synth.make_barcodes(12, 24, seed=44)
    seq, off, _ = synth.make_reads(b1, 15000, 150, seed=45, plant_lo=0, plant_hi=60, second=(b2, 90, 126))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[len(b) for b in b1], ids=[f"x{i}" for i in range(60)],
                            is_dual=True, bc_seqs2=b2, bc_lengths_no_N2=[24] * 12, ids2=[f"y{i}" for i in range(12)],
                            max_error_rate=0.2, trim_side=5, trim_side2=3)
    _all_filters_agree(cfg, seq, off, expect_path=DIAG)
    cfg2 = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[len(b) for b in b1], ids=[f"x{i}" for i in range(60)],
                             is_dual=True, bc_seqs2=b2, bc_lengths_no_N2=[24] * 12, ids2=[f"y{i}" for i in range(12)],
                             max_error_rate=0.2)
    _all_filters_agree(cfg2, seq, off, expect_path=DIAG)


@pytest.mark.parametrize("hint", [60, 150])
def test_diag_ragged_reads(hint):
    """Reads of 0..152 bases (the index holds 152 positions) with a right and a wrong hint; reads the
    index cannot hold fall back to sweeping every barcode."""
    bcs = synth.make_barcodes(72, 24, seed=46)
    seq, off, _ = synth.make_ragged_reads(bcs, 12000, 0, 152, seed=46)
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, hint=hint)


def test_diag_longer_reads():
    """Reads of up to 312 bases use the wide index (10 words per key, 4-read sub-batches); reads longer than
    the planned width fall back to sweeping every barcode; beyond 312 bases the plain sweep is planned."""
    bcs = synth.make_barcodes(72, 24, seed=47)
    seq, off, _ = synth.make_ragged_reads(bcs, 8000, 100, 260, seed=47)
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, hint=150)  # hint says 150, reads are longer
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, expect_path=DIAG)  # planned for 260: wide index
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2, trim_side=3, min_delta=0.05), seq, off, expect_path=DIAG)
    seq, off, _ = synth.make_ragged_reads(bcs, 6000, 0, 312, seed=48)
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, expect_path=DIAG)
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, hint=200)      # some reads beyond the hint
    seq, off, _ = synth.make_ragged_reads(bcs, 4000, 200, 400, seed=49)
    _all_filters_agree(_c2_config(bcs, max_error_rate=0.2), seq, off, expect_path="bitpar+verify")


def test_diag_low_complexity_queue_overflow():
    rng = np.random.Generator(np.random.PCG64(48))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "AAAACCCCGGGGTTTTAAAACCCC", "ACGT" * 6, "T" * 24, "TTTTTTTTAAAAAAAAGGGGGGGG"]
    bcs += synth.make_barcodes(65, 24, seed=48)
    motifs = ["A", "AC", "ACG", "ACGT", "T", "TTTTAAAA", "AAAACCCCGGGGTTTT"]
    reads = []
    for i in range(5000):
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:150])
        for _ in range(int(rng.integers(0, 6))):
            s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(max_error_rate=0.2), dict(max_error_rate=0.2, min_delta=0.08), dict(max_error_rate=0.2, trim_side=3)):
        exp = _all_filters_agree(_c2_config(bcs, **kw), seq, off, expect_path=DIAG)
    assert (exp["bc1"] != 0).mean() > 0.3


def test_diag_with_wildcards_and_short_barcodes():
    bcs = synth.make_barcodes(70, 24, seed=49)
    bcs[3] = bcs[3][:5] + "NN" + bcs[3][7:]  # without nindel an N is an ordinary (fifth) symbol
    bcs += ["ACGTTGCAGTCA", "TTGACCAGTAAC"]  # 12 nt at rate 0.2: kb = 2, 4 pieces of 3 -> swept unconditionally
    nn = [sum(c != "N" for c in b) for b in bcs]
    seq, off, _ = synth.make_reads([b.replace("N", "G") for b in bcs], 15000, 140, seed=49)
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=nn, ids=[str(i) for i in range(len(bcs))], max_error_rate=0.2)
    _all_filters_agree(cfg, seq, off, expect_path=DIAG)
    cfgn = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=nn, ids=[str(i) for i in range(len(bcs))], max_error_rate=0.2,
                             nindel=1)  # N-scoring: the barcode with N is swept unconditionally; split mode
    _all_filters_agree(cfgn, seq, off, expect_path=DIAG)


# ---- window-slot staging (long reads, short column windows) ----
@pytest.mark.parametrize("kw", [
    dict(ref_search_range="1:200"),
    dict(ref_search_range="end-300:end"),
    dict(ref_search_range="500:900", max_error_rate=0.1),
    dict(ref_search_range="end-250:end-20", trim_side=3),
    dict(ref_search_range="1:150", matching_algorithm="hamming", max_error_rate=0.1),
    dict(ref_search_range="100:400", min_delta=0.1),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_long_ragged_reads_with_windows(kw):
    lens = np.random.Generator(np.random.PCG64(7)).integers(16, 33, size=24)
    bcs = synth.make_barcodes(24, 24, seed=7, lengths=lens)
    seq, off, _ = synth.make_ragged_reads(bcs, 1500, 1200, 6000, seed=7, plant_lo=0, plant_hi=None)
    # plant additional copies near both ends so end-anchored windows see matches too
    rng = np.random.Generator(np.random.PCG64(8))
    seq = seq.copy()
    for i in range(0, 1500, 2):
        b = np.frombuffer(bcs[int(rng.integers(0, 24))].encode(), dtype=np.uint8)
        n = int(off[i + 1] - off[i])
        for pos in (int(rng.integers(0, 150)), n - 260 + int(rng.integers(0, 200)), 520 + int(rng.integers(0, 300))):
            if 0 <= pos and pos + len(b) <= n:
                seq[off[i] + pos: off[i] + pos + len(b)] = b
    kw = dict(kw)
    rs = kw.pop("ref_search_range")
    cfg = _c2_config(bcs, **{"max_error_rate": 0.2, **kw}, ref_search_range=H.bdx.parse_dynamic_range(rs))
    exp = _all_filters_agree(cfg, seq, off)
    assert (exp["bc1"] > 0).mean() > 0.1


@pytest.mark.parametrize("kw", [
    dict(ref_search_range="1:200"),
    dict(ref_search_range="end-300:end", trim_side=5),
    dict(ref_search_range="40:260", barcode_start_range="30:end", barcode_end_range="1:300", trim_side=3),
    dict(ref_search_range="1:150", matching_algorithm="hamming", max_error_rate=0.1),
    dict(ref_search_range="end-180:end", matching_algorithm="exact"),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_window_upload_of_the_host_entry_point(kw, monkeypatch):
    """bdx_classify_host copies only each read's column window to the device when that saves most of the bytes.
    Same results as the oracle and as the whole-read upload, including reads shorter than the window, empty reads and
    the device-side statistics (positions are positions of the whole read)."""
    bcs = synth.make_barcodes(24, 20, seed=131, min_hamming=6)
    seq0, off0, _ = synth.make_ragged_reads(bcs, 1200, 1500, 5000, seed=131, plant_lo=0, plant_hi=None)
    rng = np.random.Generator(np.random.PCG64(132))
    reads = []
    for i in range(1200):
        r = seq0[off0[i]:off0[i + 1]].copy()
        n = len(r)
        b = np.frombuffer(bcs[int(rng.integers(0, 24))].encode(), dtype=np.uint8)
        for pos in (int(rng.integers(0, 170)), n - 200 + int(rng.integers(0, 175))):
            r[pos:pos + 20] = b
        reads.append(r.tobytes().decode())
    for i in range(0, 1200, 97):  # short and empty reads in between
        reads[i] = reads[i][:int(rng.integers(0, 260))]
    reads[5] = ""
    seq, off = H.bdx.pack_reads(reads)
    kw = dict(kw)
    rngs = {k: H.bdx.parse_dynamic_range(kw.pop(k)) for k in ("ref_search_range", "barcode_start_range", "barcode_end_range") if k in kw}
    cfg = _c2_config(bcs, **{"max_error_rate": 0.15, **kw}, summary=True, **rngs)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    assert (exp["bc1"] > 0).mean() > 0.2
    for flt in ("auto", "off", "bitpar"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            got = hc.classify(seq, off)
            assert hc.window_uploads == 1, hc.kernel_path
            fuzz.assert_same(got, exp, f"window upload, filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)
            tabs = hc.stats_tables()
            got2 = hc.classify(seq[:off[300]], off[:301])  # a second, smaller batch through the same context
            fuzz.assert_same(got2, {k: v[:300] if v.shape[0] == 1200 else v[:600] for k, v in exp.items()}, "second batch")
    monkeypatch.setenv("BDX_NO_WINDOW_UPLOAD", "1")
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, "whole-read upload")
        assert hc.window_uploads == 0
        ref_tabs = hc.stats_tables()
    for p_ in ref_tabs:
        for k in ref_tabs[p_]:
            assert tabs[p_][k][1] == ref_tabs[p_][k][1] and np.array_equal(tabs[p_][k][0], ref_tabs[p_][k][0]), k
            assert ref_tabs[p_][k][0].sum() > 0


def test_long_reads_dual_windows():
    b1 = synth.make_barcodes(12, 20, seed=41, min_hamming=6)
    b2 = synth.make_barcodes(10, 20, seed=42, min_hamming=6)
    seq, off, _ = synth.make_reads(b1, 1200, 3000, seed=41, plant_lo=0, plant_hi=60, second=(b2, 2900, 2975))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[20] * 12, ids=[f"x{i}" for i in range(12)], is_dual=True,
                            bc_seqs2=b2, bc_lengths_no_N2=[20] * 10, ids2=[f"y{i}" for i in range(10)],
                            max_error_rate=0.15, ref_search_range=H.bdx.parse_dynamic_range("1:100"),
                            ref_search_range2=H.bdx.parse_dynamic_range("end-120:end"))
    exp = _all_filters_agree(cfg, seq, off)
    assert (exp["bc1"] > 0).mean() > 0.4


def test_cli_directory_mode_on_the_gpu(tmp_path):
    """The command line front end ends in the HIP hot path: demo1 (24 files, directory mode) byte-exact."""
    from biodemux_jl_amd import cli
    out = str(tmp_path / "out")
    rc = cli.main([os.path.join(H.REF, "FASTQ_files", "demo1_R1"), os.path.join(H.REF, "reference_files", "demo1.tsv"), out])
    assert rc == 0
    assert H.check_output_files(out, os.path.join(H.REF, "results", "demo1_R1")) > 0


@pytest.mark.parametrize("kw", [
    dict(max_error_rate=0.1, trim_side=5),
    dict(max_error_rate=0.2, trim_side=5, min_delta=0.05),
    dict(max_error_rate=0.25, trim_side=5, mismatch=1, indel=2),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_trim5_without_pass_outputs_uses_the_end_only_dp(kw):
    """With trim_side = 5 and no per-pass outputs requested the exact kernel drops the origin half of the DP
    (only the alignment's end is observable).  Verdicts and trim coordinates must equal the oracle's, for
    every filter mode; also dual with trim 5 / 3 (pass 1 end-only, pass 2 with origins)."""
    bcs = synth.make_barcodes(40, 24, seed=61)
    seq, off, _ = synth.make_reads(bcs, 12000, 150, seed=62)
    cfg = _c2_config(bcs, **kw)
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq, off)
    for flt in ("off", "bitpar", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=False, filter=flt) as hc:
            got = hc.classify(seq, off)
            for k in ("bc1", "bc2", "keep_start", "keep_end"):
                assert np.array_equal(got[k], exp[k]), (flt, k)
    b2 = synth.make_barcodes(10, 24, seed=63)
    s2, o2, _ = synth.make_reads(bcs, 8000, 150, seed=64, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg2 = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 40, ids=[f"x{i}" for i in range(40)], is_dual=True,
                             bc_seqs2=b2, bc_lengths_no_N2=[24] * 10, ids2=[f"y{i}" for i in range(10)],
                             max_error_rate=kw["max_error_rate"], trim_side=5, trim_side2=3)
    exp2 = H.orc.OracleClassifier(cfg2, nthreads=16, want_pass=False).classify(s2, o2)
    with H.bdx.HipClassifier(cfg2, want_pass=False) as hc:
        got2 = hc.classify(s2, o2)
        for k in ("bc1", "bc2", "keep_start", "keep_end"):
            assert np.array_equal(got2[k], exp2[k]), k


@pytest.mark.parametrize("n_reads,max_len", [(4011, 150), (2003, 150), (4005, 300), (1001, 280)])
def test_diag_partial_tiles_and_sub_batches(n_reads, max_len):
    """Last tiles whose read count is not a multiple of the index sub-batch (8 reads, 4 with the wide index)."""
    bcs = synth.make_barcodes(64, 24, seed=71)
    seq, off, _ = synth.make_ragged_reads(bcs, n_reads, max_len // 3, max_len, seed=72 + n_reads)
    for kw in (dict(max_error_rate=0.2), dict(max_error_rate=0.2, trim_side=5, min_delta=0.05)):
        _all_filters_agree(_c2_config(bcs, **kw), seq, off, expect_path=DIAG)


# ---- tiered budgets: tier 1 (capped budgets, single seeds) settles what it can, tier 0 (full budget, list mode) the rest ----
TIER = "tier1:"  # ("tier1:wave > ..." when tier 1 runs as the wave-autonomous kernel, else "tier1:qgram+bitpar > ...")


@pytest.mark.parametrize("kw", [
    dict(max_error_rate=0.2),                                  # no_delta: settled as soon as tier 1 finds a barcode
    dict(max_error_rate=0.2, min_delta=0.05),                  # with_delta: 1/24 < 0.05 <= 2/24 ...
    dict(max_error_rate=0.2, min_delta=0.1),
    dict(max_error_rate=0.17, min_delta=0.12),                 # min_delta just below the score of an unseen barcode (3/24): only perfect matches settle
    dict(max_error_rate=0.25),                                 # kb = 6: the full budget is the plain sweep
    dict(max_error_rate=0.13, min_delta=0.04),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
@pytest.mark.parametrize("want_pass", [True, False])
def test_tiered_budgets_c2_shape(kw, want_pass):
    """Known-score configs whose budget is too large for single seeds.  With per-pass outputs requested the delta
    VALUE must be exact (tier 1 may only settle a with_delta pass when it saw the runner-up); without them a proven
    lower bound on delta is enough.  Concatemers, reads without a barcode and barcodes with 3-4 errors mix the tiers."""
    bcs = synth.make_barcodes(96, 24, seed=101)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=102, sub=0.05, ins=0.015, dele=0.015, repeat=dict(frac=0.15))
    cfg = _c2_config(bcs, **kw)
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
    exp = oc.classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
        got = hc.classify(seq, off)
        assert hc.kernel_path.startswith(TIER), hc.kernel_path
        fuzz.assert_same(got, exp, f"tiered {kw}")
        assert np.array_equal(hc.counts, oc.counts)
        got2 = hc.classify(seq, off)  # a second batch through the same context (scratch words are re-armed)
        fuzz.assert_same(got2, exp, f"tiered {kw} (second batch)")
    assert 0.3 < (exp["bc1"] > 0).mean() < 0.98


def test_tiered_budgets_variable_lengths_and_dual():
    """Barcodes of 16..32 nt: the capped budgets (m / 8 - 1 = 1, 2, 3) and the score of an unseen barcode differ per
    barcode, so "no unseen barcode can tie or beat the winner" has to be decided on the Float64 scores; second pass
    on the same read."""
    rng = np.random.Generator(np.random.PCG64(103))
    lens = rng.choice([16, 17, 20, 23, 24, 25, 28, 31, 32], size=80)
    b1 = synth.make_barcodes(80, 24, seed=103, lengths=lens, min_hamming=6)
    b2 = synth.make_barcodes(48, 24, seed=104)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=105, sub=0.05, ins=0.01, dele=0.01, plant_lo=0, plant_hi=50,
                                   second=(b2, 80, 126))
    for kw in (dict(max_error_rate=0.2), dict(max_error_rate=0.2, min_delta=0.06), dict(max_error_rate=0.15, min_delta=0.03)):
        for want_pass in (True, False):
            cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[len(b) for b in b1], ids=[f"x{i}" for i in range(80)],
                                    is_dual=True, bc_seqs2=b2, bc_lengths_no_N2=[24] * 48, ids2=[f"y{i}" for i in range(48)], **kw)
            oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
            exp = oc.classify(seq, off)
            with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
                got = hc.classify(seq, off)
                assert hc.kernel_path.startswith(TIER), hc.kernel_path
                fuzz.assert_same(got, exp, f"tiered dual {kw} want_pass={want_pass}")
                assert np.array_equal(hc.counts, oc.counts)
    assert (exp["bc1"] > 0).mean() > 0.2


def test_tiered_budgets_ragged_reads_and_hints():
    """Tier 0 stages every listed read into a slot planned from the read-length hint: reads longer than planned,
    empty reads and a wrong hint must not change anything."""
    bcs = synth.make_barcodes(64, 24, seed=106)
    seq, off, _ = synth.make_ragged_reads(bcs, 20000, 0, 200, seed=107, sub=0.06, ins=0.02, dele=0.02)
    for hint in (None, 60, 150, 400):
        _all_filters_agree(_c2_config(bcs, max_error_rate=0.2, min_delta=0.05), seq, off, hint=hint)


@pytest.mark.parametrize("kw", [
    dict(max_error_rate=0.2, trim_side=5),
    dict(max_error_rate=0.2, trim_side=3, min_delta=0.05),
    dict(max_error_rate=0.2, summary=True),
    dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.1),              # demo2's costs: weighted, cmin = 1
    dict(max_error_rate=0.3, mismatch=2, indel=3, trim_side=5),                 # cmin = 2: an unseen barcode costs >= 2 (kb1 + 1)
    dict(max_error_rate=0.2, matching_algorithm="hamming", min_delta=0.05),
    dict(max_error_rate=0.2, nindel=2, trim_side=3),                            # N-scoring (no N in these barcodes)
    dict(max_error_rate=0.2, trim_side=5, barcode_start_range="1:60"),          # a start range that binds for most reads: tier 0
    dict(max_error_rate=0.2, trim_side=3, barcode_end_range="40:end", min_delta=0.05),
    dict(max_error_rate=0.2, trim_side=5, ref_search_range="5:end-3"),          # a column window alone does not bind
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_tiered_budgets_split_mode(kw):
    """Configs outside the known-score class (trimming, summary, weighted costs, Hamming, N-scoring): tier 1 filters at
    the capped budgets, the exact kernel evaluates its candidates and answers only the reads whose verdict cannot
    depend on an unseen barcode (and whose start / end ranges do not bind); the rest goes through the full-budget
    filter and the exact kernel in list mode."""
    kw = dict(kw)
    for k in ("barcode_start_range", "barcode_end_range", "ref_search_range"):
        if k in kw:
            kw[k] = H.bdx.parse_dynamic_range(kw[k])
    bcs = synth.make_barcodes(96, 24, seed=111)
    seq, off, _ = synth.make_ragged_reads(bcs, 25000, 100, 150, seed=112, sub=0.05, ins=0.015, dele=0.015, repeat=dict(frac=0.15))
    cfg = _c2_config(bcs, **kw)
    for want_pass in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
        exp = oc.classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            got = hc.classify(seq, off)
            assert hc.kernel_path.startswith(TIER), hc.kernel_path
            fuzz.assert_same(got, exp, f"tiered split {kw} want_pass={want_pass}")
            assert np.array_equal(hc.counts, oc.counts)
    assert 0.2 < (exp["bc1"] > 0).mean() < 0.99


def test_tiered_budgets_c4_dual_trim():
    b1 = synth.make_barcodes(24, 24, seed=1)
    b2 = synth.make_barcodes(16, 24, seed=2)
    seq, off, _ = synth.make_reads(b1, 40000, 150, sub=0.05, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True,
                            bc_seqs2=b2, bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)],
                            max_error_rate=0.2, trim_side=5, trim_side2=3)
    for want_pass in (True, False):
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass).classify(seq, off)
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            fuzz.assert_same(hc.classify(seq, off), exp, "C4 tiered")
            assert hc.kernel_path.startswith(TIER), hc.kernel_path


# ---- barcodes of 33..64 nt: 64-bit sweep words (no drop to the unfiltered kernel) ----
@pytest.mark.parametrize("lens,kw", [
    ([40], dict(max_error_rate=0.1)),
    ([64], dict(max_error_rate=0.2, min_delta=0.05)),
    ([33, 40, 48, 56, 64], dict(max_error_rate=0.15, trim_side=3)),
    ([20, 24, 32, 33, 47, 64], dict(max_error_rate=0.2, trim_side=5, min_delta=0.03)),   # mixed with short barcodes
    ([48], dict(max_error_rate=0.12, matching_algorithm="hamming")),
    ([36, 60], dict(max_error_rate=0.25, mismatch=1, indel=2, summary=True)),
], ids=lambda v: ",".join(map(str, v)) if isinstance(v, list) else ",".join(f"{k}={x}" for k, x in v.items()))
def test_long_barcodes_use_the_filtered_path(lens, kw):
    rng = np.random.Generator(np.random.PCG64(121))
    ls = rng.choice(lens, size=72)
    bcs = synth.make_barcodes(72, 24, seed=122, lengths=ls, min_hamming=8)
    seq, off, _ = synth.make_ragged_reads(bcs, 15000, 80, 220, seed=123, sub=0.04, ins=0.01, dele=0.01, repeat=dict(frac=0.1))
    cfg = _c2_config(bcs, **kw)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt in ("off", "bitpar", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"long barcodes {lens} {kw} filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)
            if flt != "off":
                assert "bitpar" in hc.kernel_path, hc.kernel_path
    assert (exp["bc1"] > 0).mean() > 0.25


# ---- barcodes of 65..128 nt: 128-bit sweep words ----
@pytest.mark.parametrize("lens,kw", [
    ([80], dict(max_error_rate=0.1)),
    ([128], dict(max_error_rate=0.15, min_delta=0.05)),
    ([65, 66, 80, 100, 127, 128], dict(max_error_rate=0.12, trim_side=3)),
    ([24, 40, 64, 65, 96], dict(max_error_rate=0.15, trim_side=5, min_delta=0.03)),      # mixed with short barcodes
    ([72], dict(max_error_rate=0.1, matching_algorithm="hamming")),
    ([70, 110], dict(max_error_rate=0.2, mismatch=1, indel=2, summary=True)),
], ids=lambda v: ",".join(map(str, v)) if isinstance(v, list) else ",".join(f"{k}={x}" for k, x in v.items()))
def test_barcodes_of_65_to_128_nt_use_the_filtered_path(lens, kw):
    rng = np.random.Generator(np.random.PCG64(126))
    ls = rng.choice(lens, size=40)
    bcs = synth.make_barcodes(40, 24, seed=127, lengths=ls, min_hamming=10)
    seq, off, _ = synth.make_ragged_reads(bcs, 8000, 100, 400, seed=128, sub=0.03, ins=0.008, dele=0.008, repeat=dict(frac=0.1))
    cfg = _c2_config(bcs, **kw)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt in ("off", "bitpar", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"long barcodes {lens} {kw} filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)
            if flt != "off":
                assert "bitpar" in hc.kernel_path, hc.kernel_path
    assert (exp["bc1"] > 0).mean() > 0.25


def test_barcodes_of_65_to_128_nt_dual(monkeypatch):
    b1 = synth.make_barcodes(20, 24, seed=129, lengths=[90] * 20, min_hamming=12)
    b2 = synth.make_barcodes(12, 24, seed=130, lengths=[30] * 12)
    seq, off, _ = synth.make_reads(b1, 6000, 300, seed=131, plant_lo=0, plant_hi=100, second=(b2, 220, 270), sub=0.03, ins=0.005, dele=0.005)
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[90] * 20, ids=[f"x{i}" for i in range(20)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[30] * 12, ids2=[f"y{i}" for i in range(12)], max_error_rate=0.15, trim_side=5, trim_side2=3)
    _all_filters_agree(cfg, seq, off)


def test_barcodes_beyond_128_nt_still_classify_exactly():
    """129 nt and more: outside the sweep's domain, the unfiltered exact kernel answers (slow, but exact)."""
    bcs = synth.make_barcodes(6, 24, seed=124, lengths=[140, 130, 160, 129, 150, 200], min_hamming=10)
    seq, off, _ = synth.make_ragged_reads(bcs, 1200, 200, 460, seed=125, sub=0.04)
    cfg = _c2_config(bcs, max_error_rate=0.15, trim_side=3)
    exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        fuzz.assert_same(hc.classify(seq, off), exp, "barcodes > 128 nt")
        assert hc.kernel_path == "generic"


# ---- full BASELINE sizes: independent kernel paths must agree on EVERY read (the oracle checks a sample) ----
def _device_run(cfg, seq, off, outputs, env=None, monkeypatch=None):
    import torch

    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    n = len(off) - 1
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in outputs}
    with H.bdx.HipClassifier(cfg) as hc:  # the developer switches are read here, once
        if env:
            for k in env:
                monkeypatch.delenv(k)
        hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in d_out.items()})
        hc.sync()
        return {k: v.cpu().numpy() for k, v in d_out.items()}, hc.counts, hc.kernel_path


def test_c2d_full_size_tiers_vs_full_budget(monkeypatch):
    """C2 at the reference's default rate 0.2, 10 M reads: the tiered run (single seeds at capped budgets, then two
    intact pieces in list mode) and the run that filters every read at the full budget are different kernels and
    different algorithms — all 10 M verdicts and the counters must be identical; a strided sample equals the oracle."""
    n = 10_000_000
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, n, 150)
    cfg = _c2_config(bcs, max_error_rate=0.2)
    a, ca, pa = _device_run(cfg, seq, off, ("bc1",))
    b, cb, pb = _device_run(cfg, seq, off, ("bc1",), env={"BDX_NO_TIER": "1"}, monkeypatch=monkeypatch)
    assert pa.startswith(TIER) and not pb.startswith(TIER), (pa, pb)
    assert np.array_equal(a["bc1"], b["bc1"]) and np.array_equal(ca, cb)
    idx = np.arange(0, n, 100)
    sseq = seq.reshape(n, 150)[idx].reshape(-1)
    soff = np.arange(len(idx) + 1, dtype=np.int64) * 150
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(sseq, soff)
    assert np.array_equal(a["bc1"][idx], exp["bc1"])


def test_c4_full_size_paths_agree(monkeypatch):
    """BASELINE config 4 at 10 M reads: tiered + clean-class exact kernel vs. full budget + by-construction register
    DP: bc1, bc2 and both trim coordinates of every read, and the counters; a strided sample equals the oracle."""
    n = 10_000_000
    b1 = synth.make_barcodes(24, 24, seed=synth.SEED + 1)
    b2 = synth.make_barcodes(16, 24, seed=synth.SEED + 2)
    seq, off, _ = synth.make_reads(b1, n, 150, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True,
                            bc_seqs2=b2, bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)],
                            max_error_rate=0.2, trim_side=5, trim_side2=3)
    outs = ("bc1", "bc2", "keep_start", "keep_end")
    a, ca, pa = _device_run(cfg, seq, off, outs)
    b, cb, pb = _device_run(cfg, seq, off, outs, env={"BDX_NO_TIER": "1", "BDX_NO_CLEAN": "1"}, monkeypatch=monkeypatch)
    assert pa.startswith(TIER) and not pb.startswith(TIER), (pa, pb)
    for k in outs:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(ca, cb)
    idx = np.arange(0, n, 250)
    sseq = seq.reshape(n, 150)[idx].reshape(-1)
    soff = np.arange(len(idx) + 1, dtype=np.int64) * 150
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(sseq, soff)
    for k in outs:
        assert np.array_equal(a[k][idx], exp[k]), k


# ---- stress shapes at the borders of the fast paths' domain (formerly only in tools/fuzz_campaign.py) ----
@functools.lru_cache(maxsize=1)
def _stress_cases():
    b1500 = synth.make_barcodes(1500, 20, seed=3, min_hamming=5)
    s1500 = synth.make_reads(b1500, 6000, 100, seed=3)[:2]
    b3000 = synth.make_barcodes(3000, 16, seed=4, min_hamming=4)
    s3000 = synth.make_reads(b3000, 700, 80, seed=4)[:2]
    b5000 = synth.make_barcodes(5000, 16, seed=14, min_hamming=4)
    s5000 = synth.make_reads(b5000, 300, 80, seed=14)[:2]
    b300 = synth.make_barcodes(4, 300, seed=6, min_hamming=60)
    s300 = synth.make_reads(b300, 600, 700, seed=6)[:2]
    b1 = synth.make_barcodes(1, 24, seed=7)
    s1 = synth.make_reads(b1, 5, 150, seed=7)[:2]
    b96 = synth.make_barcodes(96, 24)
    s96 = synth.make_reads(b96, 70001, 150, seed=8)[:2]
    small = (s96[0][:150 * 3000], s96[1][:3001])
    return {
        "B1500_rate0.1": (b1500, s1500, dict(max_error_rate=0.1)),
        "B1500_rate0.2_tiered": (b1500, s1500, dict(max_error_rate=0.2)),
        "B1500_trim3_delta": (b1500, s1500, dict(max_error_rate=0.1, trim_side=3, min_delta=0.06)),
        "B3000": (b3000, s3000, dict(max_error_rate=0.13)),
        "B3000_trim5_delta": (b3000, s3000, dict(max_error_rate=0.13, trim_side=5, min_delta=0.07)),
        "B5000_beyond_the_barcode_limit": (b5000, s5000, dict(max_error_rate=0.13)),
        "m300_lds_limited": (b300, s300, dict(max_error_rate=0.1, trim_side=3)),
        "B1_5reads": (b1, s1, dict(max_error_rate=0.2)),
        "70001_reads_ragged_last_tile": (b96, s96, dict(max_error_rate=0.1)),
        "negative_match_cost": (b96[:12], small, dict(max_error_rate=0.2, match=-1, mismatch=2, indel=3)),
        "fourteen_symbol_alphabet": (["ACGTRYKMSWACGTRYKMSW", "RYKMSWBDHVACGTACGTAC"], small, dict(max_error_rate=0.2)),
        "seventeen_symbol_alphabet": (["ACGTRYKMSWBDHVNEFACG", "ACGTACGTACGTACGTACGT"], small, dict(max_error_rate=0.2, trim_side=3)),
        "min_delta_beyond_an_unseen_barcode": (b96, small, dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15)),
    }


@pytest.mark.parametrize("name", ["B1500_rate0.1", "B1500_rate0.2_tiered", "B1500_trim3_delta", "B3000", "B3000_trim5_delta",
                                  "B5000_beyond_the_barcode_limit",
                                  "m300_lds_limited", "B1_5reads", "70001_reads_ragged_last_tile", "negative_match_cost",
                                  "fourteen_symbol_alphabet", "seventeen_symbol_alphabet", "min_delta_beyond_an_unseen_barcode"])
def test_stress_shapes(name):
    bcs, (seq, off), kw = _stress_cases()[name]
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[sum(c != "N" for c in b) for b in bcs],
                            ids=[str(i) for i in range(len(bcs))], **kw)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for flt in ("off", "auto"):
        with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
            fuzz.assert_same(hc.classify(seq, off), exp, f"{name} filter {flt} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)
            path = hc.kernel_path
    if name in ("B3000", "B3000_trim5_delta"):
        assert "bitpar" in path, path
    if name in ("B5000_beyond_the_barcode_limit", "m300_lds_limited", "negative_match_cost", "seventeen_symbol_alphabet"):
        assert path == "generic"
    if name == "fourteen_symbol_alphabet":  # IUPAC letters are literals for the reference; up to 15 symbols stay filtered
        assert "bitpar" in path
    if name == "B1500_rate0.2_tiered":
        assert path.startswith(TIER)
    if name == "min_delta_beyond_an_unseen_barcode":  # the seed tier could prove nothing: the pairs tier (cost <= 4, slo = 5 / 24 >= min_delta) runs instead
        assert path.startswith("tier1:pairs(diag)"), path


def test_concurrent_contexts_on_os_threads():
    """SURVEY §8(b) threading: worker tasks call the boundary concurrently, each with its own context
    (core.jl:587-599).  Four OS threads (ctypes releases the GIL inside the library), four different configs, several
    batches each, every result against the oracle."""
    import threading
    bcs = synth.make_barcodes(96, 24)
    b16 = synth.make_barcodes(40, 16, seed=77, min_hamming=5)
    jobs = [
        (_c2_config(bcs), synth.make_reads(bcs, 30000, 150, seed=501)[:2]),
        (_c2_config(bcs, max_error_rate=0.2, trim_side=3), synth.make_reads(bcs, 20000, 150, seed=502)[:2]),
        (_c2_config(bcs, max_error_rate=0.2, summary=True, min_delta=0.05), synth.make_reads(bcs, 20000, 150, seed=503)[:2]),
        (_c2_config(b16, max_error_rate=0.2, trim_side=5), synth.make_ragged_reads(b16, 20000, 60, 200, seed=504)[:2]),
    ]
    expected = [H.orc.OracleClassifier(cfg, nthreads=8).classify(seq, off) for cfg, (seq, off) in jobs]
    errors = []
    barrier = threading.Barrier(len(jobs))

    def work(k):
        try:
            cfg, (seq, off) = jobs[k]
            with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
                barrier.wait(timeout=60)
                for rep in range(4):
                    got = hc.classify(seq, off)
                    fuzz.assert_same(got, expected[k], f"thread {k} repetition {rep} [{hc.kernel_path}]")
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)


@pytest.mark.parametrize("kw", [dict(max_error_rate=0.2), dict(max_error_rate=0.2, trim_side=3), dict(max_error_rate=0.2, summary=True, min_delta=0.05),
                                dict(max_error_rate=0.25, trim_side=5, mismatch=1, indel=2)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_dense_tables_change_nothing(kw, monkeypatch):
    """Short barcodes: the dense d / window tables and the class-phased exact stage against the four-entry forms
    (BDX_NO_DENSE) on the same batch — every output identical, and both equal to the oracle."""
    bcs = synth.make_barcodes(96, 10, seed=610, min_hamming=3)
    seq, off, _ = synth.make_ragged_reads(bcs, 15000, 40, 160, seed=611, sub=0.03, ins=0.005, dele=0.005, repeat=dict(frac=0.1))
    cfg = _c2_config(bcs, **kw)
    exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq, off)
    outs = {}
    for dense in (True, False):
        if dense:
            monkeypatch.delenv("BDX_NO_DENSE", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_DENSE", "1")
        with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
            outs[dense] = hc.classify(seq, off)
            fuzz.assert_same(outs[dense], exp, f"dense tables {dense} {kw} [{hc.kernel_path}]")
    for k, v in outs[True].items():
        assert np.array_equal(v, outs[False][k], equal_nan=True) if v.dtype.kind == "f" else np.array_equal(v, outs[False][k]), k


def test_dense_windows_with_reads_beyond_16_bit_columns():
    """The dense window table packs columns into 16 bits; a read longer than that (here hidden behind a wrong length
    hint) must fall back to the whole-window DP for itself — results still exact."""
    bcs = synth.make_barcodes(96, 10, seed=620, min_hamming=3)
    seq0, off0, _ = synth.make_reads(bcs, 3000, 150, seed=621)
    reads = [seq0[off0[i]:off0[i + 1]].tobytes().decode() for i in range(3000)]
    rng = np.random.Generator(np.random.PCG64(622))
    for k, at in enumerate((70, 1500, 2999)):
        big = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=66000 + 1000 * k))
        pos = 64000 + 300 * k  # a barcode far beyond column 60000
        big = big[:pos] + bcs[7 * k] + big[pos + 10:]
        reads[at] = big
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(max_error_rate=0.2, trim_side=3), dict(max_error_rate=0.2)):
        cfg = _c2_config(bcs, **kw)
        exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq, off)
        for hint in (150, None):
            with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
                if hint:
                    hc.set_read_length_hint(hint)
                fuzz.assert_same(hc.classify(seq, off), exp, f"{kw} hint {hint} [{hc.kernel_path}]")


def test_dual_with_one_known_score_pass_hands_over_clean_window_counts():
    """Regression (found by the fuzz campaign as a GPU memory fault): a dual config whose first pass trims
    (split mode) while the second is score-only (known-score class) pushed replay slots into the LDS words split mode
    uses as window counters, so the exact kernel was told about window entries nobody had written and read whatever
    the buffer held before — here: the previous context's per-pass doubles.  Same sequence as the campaign: per-pass
    outputs first, then without them, on fresh contexts."""
    for seed in (53109,):
        cfg, seq, off = fuzz.random_case_band(seed)
        assert cfg.is_dual and cfg.trim_side == 5 and cfg.trim_side2 is None  # the shape of the failing case
        for want in (True, False, True, False):
            oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want)
            exp = oc.classify(seq, off)
            with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
                fuzz.assert_same(hc.classify(seq, off), exp, f"seed {seed} per-pass outputs {want} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts)
    bcs = synth.make_barcodes(19, 20, seed=631, min_hamming=6)
    b2 = synth.make_barcodes(23, 20, seed=632, min_hamming=6)
    seq, off, _ = synth.make_reads(bcs, 6000, 100, seed=633, plant_lo=0, plant_hi=30, second=(b2, 50, 80))
    for t1, t2 in ((5, None), (3, None), (None, 3), (None, 5)):
        cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[20] * 19, ids=[f"x{i}" for i in range(19)], is_dual=True,
                                bc_seqs2=b2, bc_lengths_no_N2=[20] * 23, ids2=[f"y{i}" for i in range(23)],
                                max_error_rate=0.1, trim_side=t1, trim_side2=t2)
        _all_filters_agree(cfg, seq, off)
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=False)
        with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
            fuzz.assert_same(hc.classify(seq, off), oc.classify(seq, off), f"trim {t1}/{t2} without per-pass outputs")


@pytest.mark.parametrize("gen,seed", [("band", s) for s in range(100, 112)] + [("tiers", s) for s in range(100, 108)] +
                         [("many", s) for s in range(100, 106)] + [("band", 53109)])
def test_context_reuse_with_permuted_batches(gen, seed):
    """One context, three batches: the reads, the same reads permuted (a few dropped), the reads again.  Every per-read
    work buffer of the second call still holds the first call's entries at the same indices for OTHER reads, so any
    read of an entry the second call did not write becomes a mismatch instead of going unnoticed."""
    make = {"band": fuzz.random_case_band, "tiers": fuzz.random_case_tiers, "many": fuzz.random_case_many_barcodes}[gen]
    cfg, seq, off = make(seed)
    n = len(off) - 1
    pseq, poff, keep = fuzz.permuted_batch(seq, off, seed)
    for want in (True, False):
        oc = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want)
        exp = oc.classify(seq, off)
        pexp = H.orc.OracleClassifier(cfg, nthreads=8, want_pass=want).classify(pseq, poff)
        with H.bdx.HipClassifier(cfg, want_pass=want) as hc:
            fuzz.assert_same(hc.classify(seq, off), exp, f"{gen} {seed} first batch [{hc.kernel_path}]")
            fuzz.assert_same(hc.classify(pseq, poff), pexp, f"{gen} {seed} permuted batch on the same context [{hc.kernel_path}]")
            fuzz.assert_same(hc.classify(seq, off), exp, f"{gen} {seed} first batch again [{hc.kernel_path}]")


def test_host_entry_point_small_batch_forms():
    """The host entry point picks its transfer path from the batch size (staged input + verdicts written into
    page-locked memory below ~2 MB, a single staged device-to-host copy up to 256 k reads, plain copies beyond) and
    from the outputs asked for.  Sizes on both sides of every threshold, output subsets through the raw C call,
    batches of empty reads, alternating sizes on one context."""
    import ctypes as C
    from biodemux_jl_amd import hipabi
    bcs = synth.make_barcodes(24, 16, seed=640, min_hamming=5)
    seq_all, off_all, _ = synth.make_ragged_reads(bcs, 300000, 20, 120, seed=641)
    for kw in (dict(max_error_rate=0.13), dict(max_error_rate=0.2, trim_side=3)):
        cfg = _c2_config(bcs, **kw)
        exp_all = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq_all, off_all)
        with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
            for n in (1, 3, 4000, 20000, 70000, 5, 262144, 262145, 300000, 2):
                seq, off = seq_all[:off_all[n]], off_all[:n + 1]
                got = hc.classify(seq, off)
                for k in ("bc1", "bc2", "keep_start", "keep_end"):
                    assert np.array_equal(got[k], exp_all[k][:n]), (kw, n, k)
                # output subsets through the raw entry point
                for fields in (("bc1",), ("bc1", "keep_end"), ("bc1", "bc2", "keep_start")):
                    o = hipabi.BdxOutputs()
                    bufs = {f: np.full(n, -77, dtype=np.int32) for f in fields}
                    for f, b in bufs.items():
                        setattr(o, f, b.ctypes.data)
                    assert hc.lib.bdx_classify_host(hc.h, seq.ctypes.data, off.ctypes.data, n, C.byref(o)) == 0
                    for f, b in bufs.items():
                        assert np.array_equal(b, exp_all[f][:n]), (kw, n, fields, f)
            # reads without a single base
            e_seq, e_off = np.zeros(1, np.uint8), np.zeros(6, np.int64)
            got = hc.classify(e_seq[:0], e_off)
            assert np.array_equal(got["bc1"], np.zeros(5, np.int32)) and np.array_equal(got["keep_start"], np.full(5, -1, np.int32))


def test_pipelined_host_upload_equals_the_single_upload(monkeypatch):
    """Large batches of configs with heavier kernels go up in chunks beside the previous chunk's kernels (one classify
    call per chunk on the same context).  Same outputs, counters and statistics as the single upload (BDX_NO_PIPELINE),
    and a sample against the oracle."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 1_000_001, 150, seed=651, repeat=dict(frac=0.05))
    for kw in (dict(max_error_rate=0.2, summary=True, min_delta=0.05), dict(max_error_rate=0.2, trim_side=5)):
        cfg = _c2_config(bcs, **kw)
        res = {}
        for piped in (True, False):
            if piped:
                monkeypatch.delenv("BDX_NO_PIPELINE", raising=False)
            else:
                monkeypatch.setenv("BDX_NO_PIPELINE", "1")
            with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
                out = hc.classify(seq, off)
                res[piped] = (out, hc.counts.copy(), hc.stats_tables() if cfg.summary else None)
                # the chunked path really ran (or really did not): a changed threshold must not turn this test into a
                # comparison of the single upload with itself
                assert (hc.pipelined_calls > 0) == piped, (kw, piped, hc.pipelined_calls)
                assert hc.rejected_windows == 0
        for k, v in res[True][0].items():
            assert np.array_equal(v, res[False][0][k], equal_nan=True) if v.dtype.kind == "f" else np.array_equal(v, res[False][0][k]), (kw, k)
        assert np.array_equal(res[True][1], res[False][1])
        if cfg.summary:
            for p_ in res[True][2]:
                for name in res[True][2][p_]:
                    assert np.array_equal(res[True][2][p_][name][0], res[False][2][p_][name][0]), (kw, name)
        k = 4000
        exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq[:off[k]], off[:k + 1])
        fuzz.assert_same({kk: (v[:k] if v.shape[0] == 1_000_001 else v[:2 * k]) for kk, v in res[True][0].items()}, exp, str(kw))


def test_large_result_downloads_all_forms(monkeypatch):
    """Large batches bring their result vectors back through a page-locked staging buffer (several host threads copy them
    out).  With and without it, output arrays that are fresh / reused / not page-aligned / page-locked, with and without the
    per-pass outputs, single upload and chunked upload: identical results, the staged form really ran (or really did not), a
    sample against the oracle."""
    from biodemux_jl_amd import hipabi
    bcs = synth.make_barcodes(96, 24)
    n = 1_150_003
    seq, off, _ = synth.make_reads(bcs, n, 150, seed=661)
    for kw, want_pass in ((dict(max_error_rate=0.1), False), (dict(max_error_rate=0.2, trim_side=5), True)):
        cfg = _c2_config(bcs, **kw)
        ref = None
        for env in ({}, {"BDX_NO_STAGED_DOWNLOAD": "1"}):
            for k in ("BDX_NO_STAGED_DOWNLOAD",):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
                got = hc.classify(seq, off)  # fresh arrays
                assert (hc.staged_downloads > 0) == ("BDX_NO_STAGED_DOWNLOAD" not in env), (kw, env, hc.staged_downloads)
                if ref is None:
                    ref = got
                    k = 3000
                    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass).classify(seq[:off[k]], off[:k + 1])
                    fuzz.assert_same({kk: (v[:k] if v.shape[0] == n else v[:2 * k]) for kk, v in got.items()}, exp, str(kw))
                for kk, v in got.items():
                    assert np.array_equal(v, ref[kk], equal_nan=True) if v.dtype.kind == "f" else np.array_equal(v, ref[kk]), (kw, env, kk)
                if not want_pass:
                    # reused arrays holding garbage, arrays that start in the middle of a page, page-locked arrays
                    reuse = {f: np.full(n, -77, dtype=np.int32) for f in ("bc1", "bc2", "keep_start", "keep_end")}
                    odd = {f: np.full(n + 3, -77, dtype=np.int32)[3:] for f in reuse}
                    pinned = {f: hipabi.pinned_empty(n, np.int32) for f in reuse}
                    before = hc.staged_downloads
                    for outs in (reuse, odd, pinned):
                        g2 = hc.classify(seq, off, out=outs)
                        for f in reuse:
                            assert np.array_equal(g2[f], ref[f]), (kw, env, f)
                    # page-locked destinations take the direct copies
                    assert hc.staged_downloads - before == (2 if "BDX_NO_STAGED_DOWNLOAD" not in env else 0), (env, hc.staged_downloads - before)
                assert hc.rejected_windows == 0


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_band_roll_vs_oracle(seed, monkeypatch):
    """Barcodes of 33 .. 128 bases inside the clean class (trimming, summary, weighted costs): the exact stage is the rolling
    diagonal band (sg_band_roll) over the filter's end columns — or, filter off, over the whole window in chunks.  Against the
    oracle with and without it (BDX_NO_BAND_ROLL: the by-construction LDS form), filter auto and off; column windows that start
    inside the read, barcodes over the read's ends, concatemers, low-complexity barcodes, mixed lengths, dual."""
    cfg, seq, off = fuzz.random_case_band_long(7100 + seed)
    oc = H.orc.OracleClassifier(cfg, nthreads=16)
    exp = oc.classify(seq, off)
    for roll in (True, False):
        if roll:
            monkeypatch.delenv("BDX_NO_BAND_ROLL", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_BAND_ROLL", "1")
        for flt in ("auto", "off"):
            with H.bdx.HipClassifier(cfg, want_pass=True, filter=flt) as hc:
                got = hc.classify(seq, off)
                fuzz.assert_same(got, exp, f"seed {seed} roll {roll} filter {flt} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), (seed, roll, flt)
                assert hc.rejected_windows == 0
    monkeypatch.delenv("BDX_NO_BAND_ROLL", raising=False)
