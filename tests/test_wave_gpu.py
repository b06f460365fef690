"""GPU parity of the wave-autonomous kernel (csrc/bdx_wave.hip) — the kernel behind the headline configuration.

It answers reads of the known-score class (find_best_matching_bc replayed on unit distances,
classification.jl:632-713) and hands every other read to the general kernel; each test compares the whole chain
with the oracle, checks through the ``wave_launches`` counter that the kernel under test really ran, and most
also run the same batch with ``BDX_NO_WAVE`` (general kernel alone) — every output must be identical.
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
    # (torch's bundled HIP runtime has to come up before the library's first bdx_create in a process that uses both:
    # the other order leaves torch without a device — INTEGRATION.md)
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def _cfg(bcs, **kw):
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"bc{i + 1}" for i in range(len(bcs))],
                max_error_rate=0.1)
    base.update(kw)
    return H.bdx.DemuxConfig(**base)


def _both_kernels(cfg, seq, off, monkeypatch, want_pass=True, expect_wave=True, hint=None, env=None):
    """classify with and without the wave kernel; both equal the oracle (outputs and counters)."""
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
    exp = oc.classify(seq, off)
    for wave in (True, False):
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        if wave:
            monkeypatch.delenv("BDX_NO_WAVE", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_WAVE", "1")
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            monkeypatch.delenv("BDX_NO_WAVE", raising=False)
            for k in (env or {}):
                monkeypatch.delenv(k, raising=False)
            if hint is not None:
                hc.set_read_length_hint(hint)
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"wave {wave} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (wave, hc.kernel_path)
            if wave and expect_wave:
                assert hc.wave_launches > 0 and "wave" in hc.kernel_path, hc.kernel_path
            if not wave:
                assert hc.wave_launches == 0 and "wave" not in hc.kernel_path, hc.kernel_path
            got2 = hc.classify(seq, off)  # the same context again: per-launch scratch words are re-armed
            fuzz.assert_same(got2, exp, f"wave {wave}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("kw", [dict(), dict(min_delta=0.05), dict(max_error_rate=0.13), dict(max_error_rate=0.05),
                                dict(max_error_rate=0.0)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "C2")
def test_wave_c2_shape(kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 60000, 150)
    exp = _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 255, 256, 257, 4000, 4001])
def test_wave_batch_sizes(n, monkeypatch):
    """Partial tiles, partial chunks of the tile queue, fewer tiles than waves."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 4096, 150, seed=5)
    _both_kernels(_cfg(bcs), seq[: off[n]], off[: n + 1].copy(), monkeypatch)


@pytest.mark.parametrize("want_pass", [True, False])
@pytest.mark.parametrize("kw", [dict(max_error_rate=0.2), dict(max_error_rate=0.2, min_delta=0.1), dict(max_error_rate=0.17, min_delta=0.04)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_wave_as_tier_1(kw, want_pass, monkeypatch):
    """The reference's default rate: tier 1 (capped budgets + settle rule) runs as the wave kernel, tier 0 as before."""
    bcs = synth.make_barcodes(96, 24, seed=101)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=102, sub=0.05, ins=0.015, dele=0.015, repeat=dict(frac=0.15))
    cfg = _cfg(bcs, **kw)
    exp = _both_kernels(cfg, seq, off, monkeypatch, want_pass=want_pass)
    with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
        hc.classify(seq, off)
        assert hc.kernel_path.startswith("tier1:wave > "), hc.kernel_path
    assert 0.3 < (exp["bc1"] > 0).mean() < 0.98


def test_wave_bytes_outside_acgt(monkeypatch):
    """Reads with N, lower case, IUPAC letters, control bytes and bytes that alias a letter's 3-bit index ((byte >> 1)
    & 7): none of them equals a barcode base (the reference compares raw bytes, classification.jl:185; reads are never
    upper-cased), inside a planted barcode they are substitutions."""
    bcs = synth.make_barcodes(48, 24, seed=7)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=8, n_rate=0.0)
    rng = np.random.Generator(np.random.PCG64(9))
    seq = seq.copy()
    odd = np.frombuffer(b"NnacgtRYKMSWBDHVU*-.\x00\x01\x7f\xff@BDFPQRSUVEaceg\x21\x23\x27\x34", dtype=np.uint8)
    pos = rng.choice(len(seq), size=len(seq) // 60, replace=False)
    seq[pos] = odd[rng.integers(0, len(odd), size=len(pos))]
    # whole reads in lower case: must never match
    mat = seq.reshape(-1, 150)
    mat[::97] = np.where((mat[::97] >= 65) & (mat[::97] <= 90), mat[::97] + 32, mat[::97])
    exp = _both_kernels(_cfg(bcs), seq, off, monkeypatch)
    assert (exp["bc1"][::97] <= 0).all()
    assert (exp["bc1"] > 0).mean() > 0.3


@pytest.mark.parametrize("hint", [None, 40, 150, 400])
def test_wave_ragged_reads_and_wrong_hints(hint, monkeypatch):
    """Lengths 0 .. 260 (empty reads, reads shorter than a seed or a barcode), hints that are too small (tiles that do
    not fit the images go to the general kernel) or too large."""
    bcs = synth.make_barcodes(48, 24, seed=21)
    seq, off, _ = synth.make_ragged_reads(bcs, 20000, 0, 260, seed=21)
    _both_kernels(_cfg(bcs), seq, off, monkeypatch, hint=hint, expect_wave=hint != 40 or True)


def test_wave_offsets_need_not_start_at_zero_or_be_aligned(monkeypatch):
    bcs = synth.make_barcodes(24, 24, seed=3)
    seq, off, _ = synth.make_reads(bcs, 5000, 151, seed=4)  # odd length: every tile starts at another alignment
    cfg = _cfg(bcs)
    exp = H.orc.OracleClassifier(cfg, nthreads=16).classify(seq, off)
    pad = 13
    seq2 = np.concatenate([np.full(pad, ord("A"), dtype=np.uint8), seq])
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
        got = hc.classify(seq2, off + pad)
        assert hc.wave_launches > 0
        fuzz.assert_same(got, exp, "shifted offsets")


def test_wave_low_complexity_overflows(monkeypatch):
    """Low-complexity barcodes and reads make almost every position a seed hit: the hit queue, the per-read record
    tables and the sweep queue overflow; the affected tiles / reads must come back from the general kernel unchanged."""
    rng = np.random.Generator(np.random.PCG64(11))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "AAAACCCCGGGGTTTTAAAACCCC", "ACGT" * 6, "T" * 24, "TTTTTTTTAAAAAAAAGGGGGGGG"]
    bcs += synth.make_barcodes(25, 24, seed=11)
    motifs = ["A", "AC", "ACG", "ACGT", "T", "TTTTAAAA", "AAAACCCCGGGGTTTT"]
    reads = []
    for i in range(6000):
        if i % 3 == 0:  # ordinary reads between the pathological ones: tiles mix both
            reads.append("".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=150)))
            continue
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:150])
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(), dict(min_delta=0.05)):
        exp = _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] != 0).mean() > 0.3


def test_wave_many_survivors_and_concatemers(monkeypatch):
    """A family of near-identical barcodes leaves more than four survivors per read (list), concatemers seed one
    barcode at two places (one merged, long sweep window: more than one 32-column block)."""
    base = synth.make_barcodes(1, 24, seed=77)[0]
    fam = [base]
    for i in range(9):
        j = 2 * i + 1
        fam.append(base[:j] + ("A" if base[j] != "A" else "C") + base[j + 1:])
    bcs = fam + synth.make_barcodes(22, 24, seed=78)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=79, repeat=dict(frac=0.3))
    for kw in (dict(), dict(min_delta=0.05), dict(max_error_rate=0.13)):
        exp = _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3


@pytest.mark.parametrize("m,rate", [(16, 0.07), (20, 0.1), (24, 0.1), (28, 0.08), (32, 0.1), (21, 0.1), (31, 0.07)])
def test_wave_barcode_lengths(m, rate, monkeypatch):
    """Other barcode lengths / budgets: pieces of 6, 7 and 8 bases, top-aligned patterns of 16 .. 32 rows, the column
    from which a sweep tracks its score (min over the barcodes of m - kb - 1)."""
    bcs = synth.make_barcodes(64, m, seed=m)
    seq, off, _ = synth.make_reads(bcs, 20000, 120, seed=m + 1)
    _both_kernels(_cfg(bcs, max_error_rate=rate), seq, off, monkeypatch, expect_wave=False)


def test_wave_mixed_barcode_lengths(monkeypatch):
    lens = [24, 26, 28, 30, 32, 25, 27, 29] * 6
    bcs = synth.make_barcodes(len(lens), 24, seed=55, lengths=lens)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=56)
    _both_kernels(_cfg(bcs, max_error_rate=0.08), seq, off, monkeypatch, expect_wave=False)


@pytest.mark.parametrize("env", [dict(BDX_WAVE_RW="8"), dict(BDX_WAVE_RW="16"), dict(BDX_WAVE_RW="32", BDX_WAVE_WAVES="4"),
                                 dict(BDX_WAVE_WAVES="16"), dict(BDX_CU_COUNT="32"), dict(BDX_CU_COUNT="7")],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_wave_geometries_and_device_shape(env, monkeypatch):
    """Forced tile sizes / workgroup shapes, and a device that reports fewer compute units (a partitioned part): the
    persistent grids follow the device's shape (hipDeviceAttributeMultiprocessorCount), results do not change."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 50000, 150, seed=17)
    _both_kernels(_cfg(bcs), seq, off, monkeypatch, env=env)


def test_device_shape_override_on_the_other_paths(monkeypatch):
    """BDX_CU_COUNT also shapes the general kernel's persistent grid and the exact kernel's list-mode grid."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 50000, 150, seed=18)
    for kw in (dict(max_error_rate=0.2, trim_side=5), dict(max_error_rate=0.2, min_delta=0.1), dict(max_error_rate=0.25, mismatch=1, indel=2)):
        cfg = _cfg(bcs, **kw)
        oc = H.orc.OracleClassifier(cfg, nthreads=16)
        exp = oc.classify(seq, off)
        monkeypatch.setenv("BDX_CU_COUNT", "24")
        with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
            monkeypatch.delenv("BDX_CU_COUNT")
            fuzz.assert_same(hc.classify(seq, off), exp, f"{kw} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts)


def test_wave_full_size_equals_the_general_kernel(monkeypatch):
    """BASELINE config 2 at its full size: all 10 M verdicts and the counters identical between the two kernels, a
    strided sample equal to the oracle."""
    import torch

    n = 10_000_000
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, n, 150)
    cfg = _cfg(bcs)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    res = {}
    for wave in (True, False):
        if not wave:
            monkeypatch.setenv("BDX_NO_WAVE", "1")
        with H.bdx.HipClassifier(cfg) as hc:
            monkeypatch.delenv("BDX_NO_WAVE", raising=False)
            hc.set_read_length_hint(150)
            out = torch.empty(n, dtype=torch.int32, device=dev)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=out.data_ptr())
            hc.sync()
            res[wave] = (out.cpu().numpy(), hc.counts.copy(), hc.wave_launches)
    assert res[True][2] > 0 and res[False][2] == 0
    assert np.array_equal(res[True][0], res[False][0]) and np.array_equal(res[True][1], res[False][1])
    idx = np.arange(0, n, 100)
    sseq = seq.reshape(n, 150)[idx].reshape(-1)
    soff = np.arange(len(idx) + 1, dtype=np.int64) * 150
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(sseq, soff)
    assert np.array_equal(res[True][0][idx], exp["bc1"])


# ---- split mode: the wave kernel as the filter in front of the exact kernel (trimming, summary, weighted costs, dual) ----
@pytest.mark.parametrize("kw", [
    dict(trim_side=3), dict(trim_side=5), dict(max_error_rate=0.1, summary=True), dict(trim_side=3, min_delta=0.05),
    dict(max_error_rate=0.2, trim_side=5), dict(max_error_rate=0.2, trim_side=3, summary=True),   # tiered: tier 1's filter
    dict(max_error_rate=0.17, mismatch=2, indel=3, trim_side=5),                                  # weighted costs: budget floor(ae / cmin)
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_wave_split_mode(kw, monkeypatch):
    """Candidate masks and column windows written by the wave kernel, verdicts and trim coordinates from the exact
    kernel (diagonal band where it applies): identical to the general filter and to the oracle, statistics included."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=31, repeat=dict(frac=0.1))
    cfg = _cfg(bcs, **kw)
    exp = _both_kernels(cfg, seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3
    if cfg.summary:
        tabs = {}
        for wave in (True, False):
            if not wave:
                monkeypatch.setenv("BDX_NO_WAVE", "1")
            with H.bdx.HipClassifier(cfg) as hc:
                monkeypatch.delenv("BDX_NO_WAVE", raising=False)
                hc.classify(seq, off)
                tabs[wave] = hc.stats_tables()
        for p_ in tabs[True]:
            for name in tabs[True][p_]:
                assert np.array_equal(tabs[True][p_][name][0], tabs[False][p_][name][0]), (kw, name)


@pytest.mark.parametrize("t1,t2,rate", [(5, 3, 0.2), (5, 3, 0.1), (3, 5, 0.1), (5, None, 0.1), (None, 3, 0.2)])
def test_wave_split_mode_dual(t1, t2, rate, monkeypatch):
    """Dual barcodes (C4's shape): both passes' pieces share the wave kernel's hash table, the barcodes of pass 2 are
    numbered behind those of pass 1; masks, windows and counts go out per pass."""
    b1 = synth.make_barcodes(24, 24, seed=1)
    b2 = synth.make_barcodes(16, 24, seed=2)
    seq, off, _ = synth.make_reads(b1, 40000, 150, seed=41, plant_lo=0, plant_hi=40, second=(b2, 100, 126))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=rate, trim_side=t1, trim_side2=t2)
    exp = _both_kernels(cfg, seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3


def test_wave_split_low_complexity_and_ragged(monkeypatch):
    """Overflowing queues in split mode: the affected reads go to the exact kernel with every barcode as a candidate and
    no windows."""
    rng = np.random.Generator(np.random.PCG64(12))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "ACGT" * 6, "T" * 24] + synth.make_barcodes(27, 24, seed=12)
    motifs = ["A", "AC", "ACG", "ACGT", "T", "TTTTAAAA"]
    reads = []
    for i in range(3000):
        if i % 3 == 0:
            reads.append("".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=int(rng.integers(0, 200)))))
            continue
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:int(rng.integers(30, 180))])
        for _ in range(int(rng.integers(0, 4))):
            s[int(rng.integers(0, len(s)))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=3), dict(trim_side=5, summary=True)):
        _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch, expect_wave=False)


@pytest.mark.parametrize("kw", [dict(matching_algorithm="hamming", max_error_rate=0.1), dict(matching_algorithm="hamming", max_error_rate=0.13, trim_side=3),
                                dict(matching_algorithm="exact"), dict(matching_algorithm="exact", trim_side=5),
                                dict(matching_algorithm="hamming", max_error_rate=0.2, min_delta=0.05)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_wave_split_mode_hamming_and_exact(kw, monkeypatch):
    """:hamming / :exact (classification.jl:485-625): the wave kernel's sweeps bound the unit edit distance, which is at
    most the Hamming distance of an occurrence; the window entries carry the first START position and the last end column
    (classification.jl:490-491, :570-571), the scans themselves run in the exact kernel."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=51, repeat=dict(frac=0.1))
    exp = _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch, expect_wave=kw.get("max_error_rate") != 0.2)
    assert (exp["bc1"] > 0).mean() > 0.2


# ---- many barcodes: the hit queue and the sweep list of a tile are sized from the expected chance hits per read ----
@pytest.mark.parametrize("n_bc,kw,expect", [
    (192, dict(), True), (384, dict(), True), (700, dict(), True),          # chance hits per read 1.3 / 2.6 / 4.8
    (384, dict(min_delta=0.05), True), (128, dict(trim_side=3), True),      # replay with with_delta / split mode
    (384, dict(trim_side=3), True), (500, dict(trim_side=5, summary=True), True),   # split mode: 12 / 16 candidate words per read
    (520, dict(trim_side=5), False),                                         # 17 words: the general kernel
    (300, dict(max_error_rate=0.2), True),                                  # as tier 1 (chance 2.1)
    (700, dict(max_error_rate=0.2), False),                                 # tier 1 beyond its limit of 3: the general kernel
    (1000, dict(), False),                                                  # beyond the plain limit of 6
], ids=lambda v: str(v) if not isinstance(v, dict) else ",".join(f"{k}={x}" for k, x in v.items()) or "plain")
def test_wave_many_barcodes(n_bc, kw, expect, monkeypatch):
    bcs = synth.make_barcodes(n_bc, 24, seed=91, min_hamming=6)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=92)
    _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch, want_pass=False, expect_wave=expect)


# ---- known-end class: trim_side = 5 without start positions — the wave kernel answers and trims itself ----
def _kend_both(cfg, seq, off, monkeypatch, expect=True, hint=None):
    """verdicts + keep range with and without the known-end form (BDX_NO_KEND: filter + exact kernel); both equal the oracle."""
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
            fuzz.assert_same(got, exp, f"known-end {kend} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (kend, hc.kernel_path)
            assert ("wave(end)" in hc.kernel_path) == (kend and expect), hc.kernel_path
            fuzz.assert_same(hc.classify(seq, off), exp, f"known-end {kend}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("kw", [
    dict(trim_side=5), dict(trim_side=5, min_delta=0.05), dict(trim_side=5, max_error_rate=0.05), dict(trim_side=5, max_error_rate=0.0),
    dict(trim_side=5, max_error_rate=0.2), dict(trim_side=5, max_error_rate=0.2, min_delta=0.1), dict(trim_side=5, max_error_rate=0.15),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_known_end_c2_shape(kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 50000, 150, seed=95, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.15))
    exp = _kend_both(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3
    m = exp["bc1"] > 0
    assert (exp["keep_start"][m] > 1).mean() > 0.9  # (the case is not vacuous: nearly every match is trimmed)


def test_known_end_not_with_start_positions_or_other_trim_sides(monkeypatch):
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=96)
    # (trim_side = 3 is in the class since round 4: tests/test_known_trim_gpu.py)
    for kw in (dict(trim_side=5, summary=True), dict(trim_side=5, mismatch=2, indel=2)):
        _kend_both(_cfg(bcs, **kw), seq, off, monkeypatch, expect=False)
    cfg = _cfg(bcs, trim_side=5)
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True).classify(seq, off)
    with H.bdx.HipClassifier(cfg, want_pass=True) as hc:  # per-pass outputs include the start positions: filter + exact kernel
        fuzz.assert_same(hc.classify(seq, off), exp, hc.kernel_path)
        assert "wave(end)" not in hc.kernel_path


def test_known_end_barcode_at_the_read_ends_and_ragged_reads(monkeypatch):
    """Alignments that end at the last column (nothing is kept: (1, 0)), at the first columns, reads shorter than a barcode,
    empty reads; ties between equally good ends (homopolymer runs behind the barcode)."""
    rng = np.random.Generator(np.random.PCG64(97))
    bcs = synth.make_barcodes(64, 24, seed=97)
    reads = []
    for i in range(16000):
        b = bcs[int(rng.integers(0, 64))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 3))).decode()
        body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(0, 130))))
        kind = i % 6
        if kind == 0:
            reads.append(body + c)                         # ends at the last column
        elif kind == 1:
            reads.append(c + body)
        elif kind == 2:
            reads.append(c + c[-1] * 6 + body)             # a run of the barcode's last base behind it
        elif kind == 3:
            reads.append(body[:20] + c + c[:12] + body)    # a second, partial copy
        elif kind == 4:
            reads.append(c[: int(rng.integers(0, 24))])    # shorter than the barcode (or empty)
        else:
            reads.append(body[:60] + c + body[60:])
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(trim_side=5), dict(trim_side=5, max_error_rate=0.2, min_delta=0.05)):
        exp = _kend_both(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.4
    assert ((exp["keep_start"] == 1) & (exp["keep_end"] == 0)).sum() > 500


# ---- dual barcodes in the known-score class: both passes replayed by the wave kernel ----
@pytest.mark.parametrize("kw", [
    dict(), dict(min_delta=0.05), dict(max_error_rate=0.05), dict(max_error_rate=0.2), dict(max_error_rate=0.2, min_delta=0.08),
    dict(max_error_rate=0.17),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "dual")
def test_wave_dual_known_score(kw, monkeypatch):
    b1 = synth.make_barcodes(24, 24, seed=101)
    b2 = synth.make_barcodes(16, 24, seed=102)
    seq, off, _ = synth.make_reads(b1, 40000, 150, seed=103, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.03, ins=0.008, dele=0.008,
                                   repeat=dict(frac=0.1))
    base = dict(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.1)
    base.update(kw)
    cfg = H.bdx.DemuxConfig(**base)
    for want_pass in (True, False):
        exp = _both_kernels(cfg, seq, off, monkeypatch, want_pass=want_pass)
    assert (exp["bc1"] > 0).mean() > 0.3 and (exp["bc2"] > 0).mean() > 0.3


def test_wave_dual_many_survivors_and_mixed_lengths(monkeypatch):
    """Pass 1 with a family of near-identical barcodes (more than four survivors: the read is handed on), pass 0 with barcodes
    of 20..28 nt (different budgets and piece counts per barcode)."""
    lens = np.random.Generator(np.random.PCG64(104)).choice([20, 24, 26, 28], size=40)
    b1 = synth.make_barcodes(40, 24, seed=104, lengths=lens)
    base = synth.make_barcodes(1, 24, seed=105)[0]
    fam = [base] + [base[:j] + ("A" if base[j] != "A" else "C") + base[j + 1:] for j in (1, 3, 5, 7, 9, 11)]
    b2 = fam + synth.make_barcodes(9, 24, seed=106)
    seq, off, _ = synth.make_reads(b1, 30000, 160, seed=107, plant_lo=0, plant_hi=40, second=(b2, 100, 134))
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[len(b) for b in b1], ids=[f"x{i}" for i in range(40)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.1, min_delta=0.02)
    _both_kernels(cfg, seq, off, monkeypatch)


# ---- ref_search_range: the kernel resolves every read's column window itself (classification.jl:795-807) ----
@pytest.mark.parametrize("rng_s,kw", [
    ("1:60", dict()), ("1:60", dict(min_delta=0.05)), ("end-59:end", dict()), ("5:end-3", dict()), ("20:90", dict()),
    ("1:60", dict(max_error_rate=0.2)), ("1:40", dict(max_error_rate=0.2, min_delta=0.05)),     # tiers + pairs mode on windows
    ("1:60", dict(trim_side=5)), ("end-70:end", dict(trim_side=5, max_error_rate=0.2)),            # known-end class
    ("1:60", dict(trim_side=3)), ("10:100", dict(trim_side=3, summary=True, max_error_rate=0.2)),  # split mode
    ("1:end-60", dict()), ("end-40:end-10", dict(trim_side=5)), ("1:70", dict(min_delta=0.03, max_error_rate=0.13)),  # windows that depend on the read length; empty for short reads
], ids=lambda v: v if isinstance(v, str) else ",".join(f"{k}={x}" for k, x in v.items()) or "plain")
def test_wave_ref_search_range(rng_s, kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=121)
    # barcodes planted anywhere: many lie outside or across the window's edges
    seq, off, _ = synth.make_ragged_reads(bcs, 40000, 30, 150, seed=122, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.1))
    cfg = _cfg(bcs, ref_search_range=H.bdx.parse_dynamic_range(rng_s), **kw)
    exp = _both_kernels(cfg, seq, off, monkeypatch, want_pass=False)
    assert 0.05 < (exp["bc1"] > 0).mean() < 0.95
    if "summary" not in kw:
        _both_kernels(cfg, seq, off, monkeypatch, want_pass=True)


def test_wave_ref_search_range_dual(monkeypatch):
    b1 = synth.make_barcodes(24, 24, seed=123)
    b2 = synth.make_barcodes(16, 24, seed=124)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=125, plant_lo=0, plant_hi=50, second=(b2, 90, 126), sub=0.03, ins=0.008, dele=0.008)
    for kw in (dict(), dict(max_error_rate=0.2), dict(trim_side=5, trim_side2=3, max_error_rate=0.2)):
        cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                                bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=kw.pop("max_error_rate", 0.1),
                                ref_search_range=H.bdx.parse_dynamic_range("1:60"), ref_search_range2=H.bdx.parse_dynamic_range("end-59:end"), **kw)
        exp = _both_kernels(cfg, seq, off, monkeypatch, want_pass=False)
    assert (exp["bc1"] > 0).mean() > 0.2


def test_wave_not_with_binding_start_or_end_ranges(monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=126)
    seq, off, _ = synth.make_reads(bcs, 10000, 150, seed=127)
    for kw in (dict(barcode_start_range=H.bdx.parse_dynamic_range("1:20")), dict(barcode_end_range=H.bdx.parse_dynamic_range("30:end"))):
        _both_kernels(_cfg(bcs, **kw), seq, off, monkeypatch, want_pass=False, expect_wave=False)


def test_wave_ref_search_range_dual_long_reads(monkeypatch):
    """300-base reads, a window at either end: the scan walks the two windows of every read only."""
    b1 = synth.make_barcodes(24, 24, seed=131)
    b2 = synth.make_barcodes(16, 24, seed=132)
    seq, off, _ = synth.make_reads(b1, 20000, 300, seed=133, plant_lo=0, plant_hi=30, second=(b2, 245, 276), sub=0.03, ins=0.008, dele=0.008)
    matched = []
    for rs1, rs2, kw in (("1:60", "end-59:end", dict()), ("1:80", "end-80:end-5", dict(min_delta=0.04)), ("1:70", "end-69:end", dict(max_error_rate=0.2))):
        cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                                bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=kw.pop("max_error_rate", 0.1),
                                ref_search_range=H.bdx.parse_dynamic_range(rs1), ref_search_range2=H.bdx.parse_dynamic_range(rs2), **kw)
        exp = _both_kernels(cfg, seq, off, monkeypatch, want_pass=False)
        matched.append(float((exp["bc1"] > 0).mean()))
    assert min(matched) > 0.3, matched


def test_known_end_with_per_pass_outputs_but_no_start_positions():
    """The device entry point with pass_end / pass_score / pass_bc / pass_delta requested and pass_start not: still the
    known-end class; the per-pass vectors equal the oracle's (the end is the reference's leftmost best end)."""
    import torch

    bcs = synth.make_barcodes(96, 24, seed=141)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=142, sub=0.03, ins=0.01, dele=0.01, repeat=dict(frac=0.2))
    n = len(off) - 1
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    for kw in (dict(trim_side=5), dict(trim_side=5, max_error_rate=0.2, min_delta=0.05)):
        cfg = _cfg(bcs, **kw)
        exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True).classify(seq, off)
        out_i = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
        out_p = {k: torch.empty((n, 2), dtype=torch.int32, device=dev) for k in ("pass_end", "pass_bc")}
        out_f = {k: torch.empty((n, 2), dtype=torch.float64, device=dev) for k in ("pass_score", "pass_delta")}
        with H.bdx.HipClassifier(cfg) as hc:
            hc.set_read_length_hint(150)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in {**out_i, **out_p, **out_f}.items()})
            hc.sync()
            assert "wave(end)" in hc.kernel_path, hc.kernel_path
        for k, v in {**out_i, **out_p}.items():
            got = v.cpu().numpy()
            assert np.array_equal(got, exp[k]), (kw, k, np.flatnonzero((got != exp[k]).reshape(n, -1).any(axis=1))[:5])
        for k, v in out_f.items():
            got = v.cpu().numpy()
            same = (got == exp[k]) | (np.isnan(got) & np.isnan(exp[k]))
            assert same.all(), (kw, k)


@pytest.mark.parametrize("div", ["1", "4", "64"])
def test_tier0_planned_for_a_fraction_of_the_batch(div, monkeypatch):
    """BDX_TIER0_DIV only changes the tile size tier 0's list launch is planned with (28-nt barcodes at rate 0.2: a budget of 5,
    beyond the pairs mode — the general kernel is tier 0)."""
    bcs = synth.make_barcodes(60, 28, seed=151)
    seq, off, _ = synth.make_ragged_reads(bcs, 30000, 40, 150, seed=152, sub=0.05, ins=0.012, dele=0.012)
    cfg = _cfg(bcs, max_error_rate=0.2)
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq, off)
    monkeypatch.setenv("BDX_TIER0_DIV", div)
    with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
        monkeypatch.delenv("BDX_TIER0_DIV")
        fuzz.assert_same(hc.classify(seq, off), exp, f"div {div} [{hc.kernel_path}]")
        assert hc.kernel_path.startswith("tier1:") and hc.pair_launches == 0, hc.kernel_path


# ---- window mode (round 4): reads much longer than their column window — scattered tiles that hold only the windows ----
def _win_three_ways(cfg, seq, off, monkeypatch, expect_win=True, hint=None):
    """window mode, BDX_NO_WIN (whole tiles / general kernel), BDX_NO_WAVE: all equal the oracle, counters included."""
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False)
    exp = oc.classify(seq, off)
    for env in (None, "BDX_NO_WIN", "BDX_NO_WAVE"):
        if env:
            monkeypatch.setenv(env, "1")
        with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
            if env:
                monkeypatch.delenv(env)
            if hint is not None:
                hc.set_read_length_hint(hint)
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"{env} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (env, hc.kernel_path)
            assert ("wave(win)" in hc.kernel_path) == (env is None and expect_win), (env, hc.kernel_path)
            fuzz.assert_same(hc.classify(seq, off), exp, f"{env}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("rng_str,kw", [
    ("1:60", dict()), ("1:60", dict(max_error_rate=0.2)), ("1:60", dict(min_delta=0.05)), ("20:70", dict()), ("end-59:end", dict()),
    ("end-70:end-10", dict(max_error_rate=0.2, min_delta=0.08)), ("5:40", dict(max_error_rate=0.13)),
], ids=lambda v: str(v))
def test_window_mode_c2_shape(rng_str, kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=301)
    lo, hi = {"1:60": (0, 36), "20:70": (19, 46), "end-59:end": (90, 126), "end-70:end-10": (80, 116), "5:40": (4, 16)}[rng_str]
    seq, off, _ = synth.make_reads(bcs, 50000, 150, seed=302, plant_lo=lo, plant_hi=hi, sub=0.03, ins=0.01, dele=0.01)
    cfg = _cfg(bcs, ref_search_range=H.bdx.parse_dynamic_range(rng_str), **kw)
    exp = _win_three_ways(cfg, seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.2


def test_window_mode_ragged_reads_wrong_hints_and_unaligned_offsets(monkeypatch):
    """Reads of 0..400 bases with a 1:80 window and an end-anchored one: reads shorter than the window, shorter than a barcode,
    empty; the planned length (hint) too small and too large; odd offsets (every slot starts at another address mod 16)."""
    bcs = synth.make_barcodes(64, 24, seed=311)
    seq, off, _ = synth.make_ragged_reads(bcs, 30000, 0, 400, seed=312, plant_hi=50, sub=0.03, ins=0.01, dele=0.01)
    for rng_str in ("1:80", "end-99:end-20"):
        cfg = _cfg(bcs, ref_search_range=H.bdx.parse_dynamic_range(rng_str))
        for hint in (None, 400, 170):
            _win_three_ways(cfg, seq, off, monkeypatch, hint=hint)
        _win_three_ways(cfg, seq, off, monkeypatch, hint=100, expect_win=False)  # (window 80 of 100 planned bases: whole tiles)


def _win_device_three_ways(cfg, seq, off, monkeypatch, hint=None, expect_win=True):
    """The same through the DEVICE entry point (the host entry point uploads only the windows of such batches and classifies
    them as virtual reads on the general kernel: bdx_window_uploads)."""
    import torch

    n = len(off) - 1
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False)
    exp = oc.classify(seq, off)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    for env in (None, "BDX_NO_WIN", "BDX_NO_WAVE"):
        if env:
            monkeypatch.setenv(env, "1")
        out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
        with H.bdx.HipClassifier(cfg) as hc:
            if env:
                monkeypatch.delenv(env)
            if hint is not None:
                hc.set_read_length_hint(hint)
            for _ in range(2):
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in out.items()})
                hc.sync()
                for k, v in out.items():
                    got = v.cpu().numpy()
                    assert np.array_equal(got, exp[k]), (env, k, hc.kernel_path, np.flatnonzero(got != exp[k])[:5])
            assert np.array_equal(hc.counts, 2 * oc.counts), (env, hc.kernel_path)
            assert ("wave(win)" in hc.kernel_path) == (env is None and expect_win), (env, hc.kernel_path)
    return exp


def test_window_mode_c5_shape_long_reads_mixed_barcode_lengths(monkeypatch):
    """BASELINE config 5's shape: 10 kbp reads, 24 barcodes of 16..32 nt, window 1:200, rate 0.2 (tier 1 in window mode,
    tier 0 the general kernel), and rate 0.1."""
    lens = np.random.Generator(np.random.PCG64(5)).integers(16, 33, size=24)
    bcs = synth.make_barcodes(24, 24, seed=5, lengths=lens)
    seq, off, _ = synth.make_reads(bcs, 6000, 10000, seed=321, plant_lo=0, plant_hi=150, sub=0.03, ins=0.01, dele=0.01)
    for rate in (0.2, 0.1):
        cfg = _cfg(bcs, max_error_rate=rate, ref_search_range=H.bdx.parse_dynamic_range("1:200"))
        exp = _win_device_three_ways(cfg, seq, off, monkeypatch, hint=10000)
        assert (exp["bc1"] > 0).mean() > 0.5
    # ragged long reads (2..12 kbp), no hint
    seq, off, _ = synth.make_ragged_reads(bcs, 3000, 2000, 12000, seed=322, plant_hi=150, sub=0.03, ins=0.01, dele=0.01)
    _win_device_three_ways(_cfg(bcs, max_error_rate=0.2, ref_search_range=H.bdx.parse_dynamic_range("1:200")), seq, off, monkeypatch)


def test_wave_sweeps_overflowed_reads_itself(monkeypatch):
    """A read whose record tables overflow (low complexity: more seeded (barcode, diagonal) clusters than its table holds) is
    swept over every barcode inside the wave kernel and replayed like any other read instead of going to the general kernel's
    list (round 4: C2's list is one read in ten million, and a one-read list launch costs 30 us against 5 us for an empty one).
    Both ways (BDX_NO_WAVE_FALLBACK): identical outputs and counters, equal to the oracle; the list really shrinks; reads
    outside the known-score class (empty reads) and reads with more than four survivors stay listed."""
    rng = np.random.Generator(np.random.PCG64(311))
    bcs = ["ACGT" * 6, "AC" * 12, "AAAACCCCGGGGTTTTAAAACCCC", "TTTTTTTTAAAAAAAAGGGGGGGG"] + synth.make_barcodes(44, 24, seed=312)
    reads = []
    for i in range(9000):
        body = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=150))
        if i % 17 == 1:
            reads.append("")  # (n = 0: outside the known-score class)
        elif i % 3 == 0:
            # a 40-base stretch of the first barcode's period inside a random read: its three 8-base pieces (one key) hit at
            # nine positions on thirteen diagonals four apart — thirteen records for a table of eight, nine entries of the
            # tile's hit queue (a fully periodic read would overflow the QUEUE and take its whole tile to the list)
            at = int(rng.integers(0, 110))
            reads.append(body[:at] + "ACGT" * 10 + body[at + 40:])
        else:
            reads.append(body)
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(), dict(min_delta=0.05), dict(trim_side=5), dict(trim_side=3)):
        cfg = _cfg(bcs, **kw)
        oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False)
        exp = oc.classify(seq, off)
        listed = {}
        for fb in (True, False):
            if fb:
                monkeypatch.delenv("BDX_NO_WAVE_FALLBACK", raising=False)
            else:
                monkeypatch.setenv("BDX_NO_WAVE_FALLBACK", "1")
            with H.bdx.HipClassifier(cfg, want_pass=False) as hc:
                monkeypatch.delenv("BDX_NO_WAVE_FALLBACK", raising=False)
                for rep in range(2):  # (twice: the scratch halves alternate between calls)
                    got = hc.classify(seq, off)
                    fuzz.assert_same(got, exp, f"{kw} fallback {fb} [{hc.kernel_path}]")
                    if rep == 0:
                        assert np.array_equal(hc.counts, oc.counts), (kw, fb)
                assert "wave" in hc.kernel_path and hc.wave_launches > 0, hc.kernel_path
                listed[fb] = hc.last_list_reads
                assert hc.rejected_windows == 0
        n_empty = sum(1 for r in reads if r == "")
        assert listed[False] > listed[True] + 1000, (kw, listed)      # the overflowing reads (a third of the batch) are answered in the kernel now
        assert listed[True] >= n_empty, (kw, listed, n_empty)          # what is outside the class is still handed on


def test_wave_c2_list_is_nearly_empty():
    """The headline shape: the wave kernel hands on at most a handful of reads per million."""
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, 1_000_000, 150)
    with H.bdx.HipClassifier(_cfg(bcs), want_pass=False) as hc:
        hc.classify(seq, off)
        assert "wave" in hc.kernel_path
        assert 0 <= hc.last_list_reads <= 8, hc.last_list_reads
