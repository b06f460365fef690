"""GPU parity of the wave kernel's PAIRS mode (csrc/bdx_pairs.hip = bdx_wave.hip with KB > 0).

Tiered configs — budgets too large for selective single seeds, e.g. the reference's default max_error_rate 0.2 on
24-nt barcodes (classification.jl:254: allowed_error = floor(0.2 * 24) = 4) — gather the reads tier 1 cannot settle
and filter them at the FULL budgets by the two-intact-pieces lemma; known-score configs get their verdicts from it,
split configs (trimming, summary, dual, :hamming) its candidate masks and column windows.  Every test compares the
whole chain with the oracle, checks through ``pair_launches`` that the mode really ran, and runs the same batch with
``BDX_NO_PAIRS`` (the general kernel as tier 0) — every output and counter must be identical.
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
    import torch  # (torch's HIP runtime first: see test_wave_gpu.py)

    assert torch.cuda.is_available(), "GPU tests need a HIP device"


def _cfg(bcs, **kw):
    base = dict(bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"bc{i + 1}" for i in range(len(bcs))],
                max_error_rate=0.2)
    base.update(kw)
    return H.bdx.DemuxConfig(**base)


def _with_and_without(cfg, seq, off, monkeypatch, want_pass=True, expect_pairs=True, hint=None):
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=want_pass)
    exp = oc.classify(seq, off)
    for pairs in (True, False):
        if pairs:
            monkeypatch.delenv("BDX_NO_PAIRS", raising=False)
        else:
            monkeypatch.setenv("BDX_NO_PAIRS", "1")
        with H.bdx.HipClassifier(cfg, want_pass=want_pass) as hc:
            monkeypatch.delenv("BDX_NO_PAIRS", raising=False)
            if hint is not None:
                hc.set_read_length_hint(hint)
            got = hc.classify(seq, off)
            fuzz.assert_same(got, exp, f"pairs {pairs} [{hc.kernel_path}]")
            assert np.array_equal(hc.counts, oc.counts), (pairs, hc.kernel_path)
            if pairs and expect_pairs:
                assert hc.pair_launches > 0 and "pairs" in hc.kernel_path, hc.kernel_path
            if not pairs or expect_pairs is False:  # (expect_pairs=None: whatever the planner decides)
                assert hc.pair_launches == 0 and "pairs" not in hc.kernel_path, hc.kernel_path
            got2 = hc.classify(seq, off)  # the same context again: scratch words and list counters are re-armed
            fuzz.assert_same(got2, exp, f"pairs {pairs}, second call [{hc.kernel_path}]")
    return exp


@pytest.mark.parametrize("kw", [
    dict(),                                                    # kb = 4: six pieces
    dict(min_delta=0.1),
    dict(max_error_rate=0.17),                                 # floor(4.08) = 4
    dict(max_error_rate=0.15),                                 # kb = 3: five pieces
    dict(max_error_rate=0.15, min_delta=0.05),
    dict(trim_side=5),                                         # split: masks + windows for the exact kernel
    dict(trim_side=3, summary=True, min_delta=0.05),
    dict(matching_algorithm="hamming"),
    dict(matching_algorithm="hamming", trim_side=3),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "C2d")
def test_pairs_c2d_shape(kw, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=61)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=62, sub=0.05, ins=0.012, dele=0.012)  # plenty of reads beyond tier 1's cap
    exp = _with_and_without(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.4


@pytest.mark.parametrize("n_bc", [3, 24, 32, 33, 40, 64, 65, 96, 97, 128])
def test_pairs_mask_words(n_bc, monkeypatch):
    """One, two, three and four words per barcode mask (table entries of 8 / 16 bytes)."""
    bcs = synth.make_barcodes(n_bc, 24, seed=63)
    seq, off, _ = synth.make_reads(bcs, 12000, 150, seed=64, sub=0.05, ins=0.01, dele=0.01)
    # (a handful of barcodes is swept without any filter: no tiers, no pairs mode)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch, expect_pairs=None if n_bc < 32 else True)
    _with_and_without(_cfg(bcs, trim_side=5), seq, off, monkeypatch, want_pass=False, expect_pairs=None if n_bc < 32 else True)


@pytest.mark.parametrize("n_bc", [129, 200, 256, 257, 384])
def test_pairs_groups_of_128_barcodes(n_bc, monkeypatch):
    """More than 128 barcodes (known-score configs): one set of piece tables per group of 128, the flag queue of a tile is
    drained inside the scan."""
    bcs = synth.make_barcodes(n_bc, 24, seed=65, min_hamming=6)
    seq, off, _ = synth.make_reads(bcs, 12000, 150, seed=66, sub=0.05, ins=0.01, dele=0.01)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch)
    _with_and_without(_cfg(bcs, min_delta=0.05), seq, off, monkeypatch, want_pass=False)


def test_pairs_groups_low_complexity(monkeypatch):
    """Low-complexity reads flag nearly every barcode on nearly every diagonal: the queue runs over between two drains and
    the tile is handed on."""
    rng = np.random.Generator(np.random.PCG64(86))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "ACGT" * 6, "T" * 24] + synth.make_barcodes(195, 24, seed=86, min_hamming=6)
    motifs = ["A", "AC", "ACG", "ACGT", "T"]
    reads = []
    for i in range(3000):
        if i % 2 == 0:
            b = bcs[5 + int(rng.integers(0, 195))]
            body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=126))
            reads.append(body[:50] + synth.mutate_copy(rng, b, int(rng.integers(2, 5))).decode() + body[50:])
            continue
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:150])
        for _ in range(int(rng.integers(3, 6))):
            s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch, want_pass=False)


def test_pairs_not_for_more_than_128_barcodes_in_split_mode(monkeypatch):
    """Split mode keeps four mask words per read: the general kernel stays tier 0."""
    bcs = synth.make_barcodes(130, 24, seed=65)
    seq, off, _ = synth.make_reads(bcs, 6000, 150, seed=66)
    _with_and_without(_cfg(bcs, trim_side=5), seq, off, monkeypatch, expect_pairs=False, want_pass=False)


def test_pairs_dual_c4_shape(monkeypatch):
    """BASELINE config 4: two barcode sets on one read, both trimmed (pass 1 numbers follow pass 0's in the masks)."""
    b1 = synth.make_barcodes(24, 24, seed=67)
    b2 = synth.make_barcodes(16, 24, seed=68)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=69, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.04, ins=0.01, dele=0.01)
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, trim_side=5, trim_side2=3)
    exp = _with_and_without(cfg, seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3
    cfg2 = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                             bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, summary=True)
    _with_and_without(cfg2, seq, off, monkeypatch, want_pass=False)


def test_pairs_mixed_lengths_and_budgets(monkeypatch):
    """20..27-nt barcodes at rate 0.17: budgets 3 and 4 side by side (five / six pieces per barcode in one table)."""
    lens = np.random.Generator(np.random.PCG64(70)).choice([20, 21, 23, 24, 26, 27], size=72)
    bcs = synth.make_barcodes(72, 24, seed=70, lengths=lens)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=71, sub=0.05, ins=0.01, dele=0.01)
    _with_and_without(_cfg(bcs, max_error_rate=0.17), seq, off, monkeypatch)
    _with_and_without(_cfg(bcs, max_error_rate=0.17, trim_side=3, min_delta=0.03), seq, off, monkeypatch, want_pass=False)


def test_pairs_not_when_a_budget_is_too_large(monkeypatch):
    """28 nt at rate 0.2: budget 5 — beyond the instantiated widths: the general kernel stays tier 0."""
    bcs = synth.make_barcodes(60, 28, seed=72)
    seq, off, _ = synth.make_reads(bcs, 6000, 150, seed=73)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch, expect_pairs=False)


@pytest.mark.parametrize("hint", [None, 100, 150, 300])
def test_pairs_ragged_reads_and_wrong_hint(hint, monkeypatch):
    """Reads of 0..230 bases: empty reads, reads shorter than a barcode, reads longer than the planned length (they do
    not fit the slots the scan was planned for and are handed on)."""
    bcs = synth.make_barcodes(80, 24, seed=74)
    seq, off, _ = synth.make_ragged_reads(bcs, 20000, 0, 230, seed=75, sub=0.05, ins=0.01, dele=0.01)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch, hint=hint)
    _with_and_without(_cfg(bcs, trim_side=5, min_delta=0.05), seq, off, monkeypatch, hint=hint, want_pass=False)


def test_pairs_longer_reads(monkeypatch):
    """Slots of up to 368 bytes (six 16-byte vectors per lane); beyond that the general kernel stays tier 0."""
    bcs = synth.make_barcodes(48, 24, seed=76)
    seq, off, _ = synth.make_ragged_reads(bcs, 12000, 180, 360, seed=77, sub=0.05, ins=0.01, dele=0.01)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch)
    seq, off, _ = synth.make_ragged_reads(bcs, 5000, 300, 420, seed=78)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch, expect_pairs=False)


def test_pairs_barcode_at_the_read_ends(monkeypatch):
    """Mutated copies flush with either end of the read (diagonals -kb .. 0 and n - m .. n - m + kb), also cut short."""
    rng = np.random.Generator(np.random.PCG64(79))
    bcs = synth.make_barcodes(64, 24, seed=79)
    reads = []
    for i in range(12000):
        b = bcs[int(rng.integers(0, 64))]
        c = synth.mutate_copy(rng, b, int(rng.integers(0, 6))).decode()
        body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(60, 130))))
        cut = int(rng.integers(0, 4))
        kind = i % 4
        if kind == 0:
            reads.append(c[cut:] + body)            # at the start, the first bases missing
        elif kind == 1:
            reads.append(body + (c[:len(c) - cut] if cut else c))  # at the end, the last bases missing
        elif kind == 2:
            reads.append(c + body)
        else:
            reads.append(body[:40] + c + body[40:])
    seq, off = H.bdx.pack_reads(reads)
    exp = _with_and_without(_cfg(bcs), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.4
    _with_and_without(_cfg(bcs, trim_side=3), seq, off, monkeypatch, want_pass=False)
    _with_and_without(_cfg(bcs, trim_side=5), seq, off, monkeypatch, want_pass=False)


def test_pairs_low_complexity_queue_overflow(monkeypatch):
    """Low-complexity barcodes and reads flag nearly every (barcode, diagonal): the queue of a tile runs over — the
    known-score path hands the tile on, the split path sweeps every barcode over the whole read in the kernel."""
    rng = np.random.Generator(np.random.PCG64(80))
    bcs = ["A" * 24, "AC" * 12, "ACG" * 8, "AAAACCCCGGGGTTTTAAAACCCC", "ACGT" * 6, "T" * 24, "TTTTTTTTAAAAAAAAGGGGGGGG"]
    bcs += synth.make_barcodes(41, 24, seed=80)
    motifs = ["A", "AC", "ACG", "ACGT", "T", "TTTTAAAA", "AAAACCCCGGGGTTTT"]
    reads = []
    for i in range(8000):
        if i % 3 == 0:  # ordinary reads between the pathological ones: tiles mix both
            b = bcs[7 + int(rng.integers(0, 41))]
            body = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=126))
            reads.append(body[:50] + synth.mutate_copy(rng, b, int(rng.integers(2, 5))).decode() + body[50:])
            continue
        mo = motifs[int(rng.integers(0, len(motifs)))]
        s = list((mo * 200)[int(rng.integers(0, 8)):][:150])
        for _ in range(int(rng.integers(2, 6))):  # a few point mutations: beyond tier 1's cap
            s[int(rng.integers(0, 150))] = "ACGT"[int(rng.integers(0, 4))]
        reads.append("".join(s))
    seq, off = H.bdx.pack_reads(reads)
    for kw in (dict(), dict(min_delta=0.05), dict(trim_side=3), dict(trim_side=5, summary=True)):
        exp = _with_and_without(_cfg(bcs, **kw), seq, off, monkeypatch, want_pass=kw == dict())
    assert (exp["bc1"] != 0).mean() > 0.3


def test_pairs_many_survivors(monkeypatch):
    """A family of near-identical barcodes leaves more than four survivors per read: the replay cannot hold them and the
    read goes on to the general kernel's list mode."""
    base = synth.make_barcodes(1, 24, seed=81)[0]
    fam = [base]
    for i in range(9):
        j = 2 * i + 1
        fam.append(base[:j] + ("A" if base[j] != "A" else "C") + base[j + 1:])
    bcs = fam + synth.make_barcodes(38, 24, seed=82)
    seq, off, _ = synth.make_reads(bcs, 20000, 150, seed=83, sub=0.05, ins=0.01, dele=0.01)
    for kw in (dict(), dict(min_delta=0.05)):
        exp = _with_and_without(_cfg(bcs, **kw), seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3


@pytest.mark.parametrize("n", [1, 15, 16, 17, 255, 4097])
def test_pairs_small_batches(n, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=84)
    seq, off, _ = synth.make_reads(bcs, 8192, 150, seed=85, plant_frac=0.5, sub=0.08, ins=0.02, dele=0.02)
    seq, off = seq[: n * 150], off[: n + 1]
    # (a batch none of whose reads reaches tier 0 still launches the mode: its list is empty on the device)
    _with_and_without(_cfg(bcs), seq, off, monkeypatch)


def test_pairs_full_size_c2d_equals_the_general_kernel(monkeypatch):
    """BASELINE config 2 at the reference's default rate, full size: all 10 M verdicts and the counters identical with and
    without the mode, a strided sample equal to the oracle."""
    import torch

    n = 10_000_000
    bcs = synth.make_barcodes(96, 24)
    seq, off, _ = synth.make_reads(bcs, n, 150)
    cfg = _cfg(bcs)
    dev = torch.device("cuda:0")
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    res = {}
    for pairs in (True, False):
        if not pairs:
            monkeypatch.setenv("BDX_NO_PAIRS", "1")
        with H.bdx.HipClassifier(cfg) as hc:
            monkeypatch.delenv("BDX_NO_PAIRS", raising=False)
            hc.set_read_length_hint(150)
            out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
            hc.reset_counts()
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in out.items()})
            hc.sync()
            assert (hc.pair_launches > 0) == pairs, hc.kernel_path
            res[pairs] = ({k: v.cpu().numpy() for k, v in out.items()}, hc.counts.copy())
    for k in res[True][0]:
        assert np.array_equal(res[True][0][k], res[False][0][k]), k
    assert np.array_equal(res[True][1], res[False][1])
    idx = np.arange(0, n, 197)
    sseq = seq.reshape(n, 150)[idx].reshape(-1)
    soff = np.arange(len(idx) + 1, dtype=np.int64) * 150
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(sseq, soff)
    assert np.array_equal(res[True][0]["bc1"][idx], exp["bc1"])


def test_pairs_bytes_outside_acgt(monkeypatch):
    """Reads with N, lower case, IUPAC letters, control bytes and bytes that alias a letter's 3-bit index: none of them
    equals a barcode base (classification.jl:185); the gathered slots carry the raw bytes, padding is 'N'."""
    bcs = synth.make_barcodes(96, 24, seed=87)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=88, n_rate=0.0, sub=0.04, ins=0.01, dele=0.01)
    rng = np.random.Generator(np.random.PCG64(89))
    seq = seq.copy()
    odd = np.frombuffer(b"NnacgtRYKMSWBDHVU*-.\x00\x01\x7f\xff@BDFPQRSUVEaceg\x21\x23\x27\x34", dtype=np.uint8)
    pos = rng.choice(len(seq), size=len(seq) // 60, replace=False)
    seq[pos] = odd[rng.integers(0, len(odd), size=len(pos))]
    mat = seq.reshape(-1, 150)
    mat[::97] = np.where((mat[::97] >= 65) & (mat[::97] <= 90), mat[::97] + 32, mat[::97])  # whole reads in lower case
    for kw in (dict(), dict(trim_side=5, min_delta=0.05)):
        exp = _with_and_without(_cfg(bcs, **kw), seq, off, monkeypatch, want_pass=kw == dict())
    assert (exp["bc1"][::97] <= 0).all()
    assert (exp["bc1"] > 0).mean() > 0.3


def test_pairs_unaligned_device_buffers(monkeypatch):
    """Device buffers that start at odd addresses: the gather funnels every dword out of the aligned dwords it straddles."""
    import torch

    bcs = synth.make_barcodes(96, 24, seed=90)
    seq, off, _ = synth.make_ragged_reads(bcs, 9000, 60, 150, seed=91, sub=0.05, ins=0.01, dele=0.01)
    cfg = _cfg(bcs)
    exp = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=False).classify(seq, off)
    dev = torch.device("cuda:0")
    n = len(off) - 1
    for shift in (1, 2, 3, 5, 13):
        raw = torch.zeros(len(seq) + 64, dtype=torch.uint8, device=dev)
        raw[shift:shift + len(seq)] = torch.from_numpy(seq).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in ("bc1", "bc2", "keep_start", "keep_end")}
        with H.bdx.HipClassifier(cfg) as hc:
            hc.classify_device(raw.data_ptr() + shift, d_off.data_ptr(), n, **{k: v.data_ptr() for k, v in out.items()})
            hc.sync()
            assert hc.pair_launches > 0
            for k, v in out.items():
                assert np.array_equal(v.cpu().numpy(), exp[k]), (shift, k)


def test_pairs_dual_known_score(monkeypatch):
    """Dual barcodes without trimming at the default rate: tier 1 and the pairs mode both replay two passes."""
    b1 = synth.make_barcodes(24, 24, seed=111)
    b2 = synth.make_barcodes(16, 24, seed=112)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=113, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.05, ins=0.012, dele=0.012)
    for kw in (dict(), dict(min_delta=0.08)):
        cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                                bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.2, **kw)
        exp = _with_and_without(cfg, seq, off, monkeypatch)
    assert (exp["bc1"] > 0).mean() > 0.3
    b1 = synth.make_barcodes(100, 24, seed=114, min_hamming=6)   # 100 + 60 barcodes: two groups of 128
    b2 = synth.make_barcodes(60, 24, seed=115, min_hamming=6)
    seq, off, _ = synth.make_reads(b1, 15000, 150, seed=116, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.05, ins=0.012, dele=0.012)
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 100, ids=[f"x{i}" for i in range(100)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 60, ids2=[f"y{i}" for i in range(60)], max_error_rate=0.2)
    _with_and_without(cfg, seq, off, monkeypatch, want_pass=False)


# ---- same-diagonal variants: configs whose indels cost more than their mismatches (round 4) ----
@pytest.mark.parametrize("kw,path", [
    (dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15), "tier1:pairs(diag) > pairs+verify"),   # the reference's demo2 options: budget 6 of 24 — the pairs tier (six 4-base pieces, cost <= 4) in front of eight 3-base pieces
    (dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.22), "pairs(diag)+verify"),   # min_delta beyond the pairs tier's slo = 5 / 24 too: one filter over the whole batch
    (dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.18), "tier1:pairs(diag) > pairs+verify"),   # only perfect matches settle
    (dict(max_error_rate=0.25, mismatch=1, indel=2), None),    # (without min_delta: tier 1 in front, the variant is tier 0's filter)
    (dict(max_error_rate=0.25, mismatch=1, indel=2, trim_side=3), None),
    (dict(max_error_rate=0.25, mismatch=1, indel=3, min_delta=0.1, trim_side=5), None),
    (dict(max_error_rate=0.25, mismatch=1, indel=2, summary=True, min_delta=0.15), "tier1:pairs(diag) > pairs+verify"),
    (dict(max_error_rate=0.25, mismatch=1, indel=2, trim_side=3, min_delta=0.15), "tier1:pairs(diag) > pairs+verify"),
    (dict(max_error_rate=0.25, mismatch=1, indel=3, min_delta=0.2), None),
    (dict(max_error_rate=0.2, mismatch=1, indel=2), None),     # budget 4: six 4-base pieces (as tier 0 behind tier 1)
    (dict(max_error_rate=0.34, mismatch=2, indel=4, min_delta=0.1), None),  # budget 8 = 4 mismatches
    (dict(max_error_rate=0.3, mismatch=1, indel=2), None),     # budget 7: no variant qualifies -> plain sweep
], ids=lambda v: ",".join(f"{k}={x}" for k, x in v.items()) if isinstance(v, dict) else "")
def test_pairs_same_diagonal_variants(kw, path, monkeypatch):
    bcs = synth.make_barcodes(96, 24, seed=71)
    seq, off, _ = synth.make_reads(bcs, 40000, 150, seed=72, sub=0.06, ins=0.015, dele=0.015, repeat=dict(frac=0.1))
    cfg = _cfg(bcs, **kw)
    exp = _with_and_without(cfg, seq, off, monkeypatch, expect_pairs=None if path is None else True)
    assert (exp["bc1"] > 0).mean() > 0.3
    if path is not None:
        with H.bdx.HipClassifier(cfg) as hc:
            hc.classify(seq, off)
            assert hc.kernel_path == path, hc.kernel_path


def test_pairs_same_diagonal_dual_and_other_lengths(monkeypatch):
    b1 = synth.make_barcodes(24, 24, seed=73)
    b2 = synth.make_barcodes(16, 24, seed=74)
    seq, off, _ = synth.make_reads(b1, 30000, 150, seed=75, plant_lo=0, plant_hi=40, second=(b2, 100, 126), sub=0.06, ins=0.015, dele=0.015)
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i}" for i in range(24)], is_dual=True, bc_seqs2=b2,
                            bc_lengths_no_N2=[24] * 16, ids2=[f"y{i}" for i in range(16)], max_error_rate=0.25, mismatch=1, indel=2,
                            min_delta=0.1, trim_side=5, trim_side2=3)
    _with_and_without(cfg, seq, off, monkeypatch, expect_pairs=None)
    for m, rate in ((20, 0.25), (28, 0.25), (32, 0.2), (30, 0.2)):
        bcs = synth.make_barcodes(64, m, seed=76 + m, min_hamming=max(4, m // 4))
        seq, off, _ = synth.make_ragged_reads(bcs, 20000, 30, 150, seed=77 + m, sub=0.06, ins=0.015, dele=0.015)
        _with_and_without(_cfg(bcs, max_error_rate=rate, mismatch=1, indel=2, min_delta=0.1), seq, off, monkeypatch, expect_pairs=None)


def test_pairs_same_diagonal_low_complexity_overflow(monkeypatch):
    """Reads that flag nearly every (barcode, diagonal): the queue runs over, the tile is swept whole."""
    rng = np.random.Generator(np.random.PCG64(78))
    bcs = synth.make_barcodes(96, 24, seed=79)
    seq, off, _ = synth.make_reads(bcs, 12000, 150, seed=80, sub=0.05, ins=0.01, dele=0.01)
    seq = seq.copy()
    for i in range(0, 12000, 3):  # two-letter reads around the planted barcode
        r = seq[off[i]:off[i + 1]]
        r[:] = np.where(rng.random(150) < 0.85, np.frombuffer(b"ACAC" * 38, dtype=np.uint8)[:150], r)
    _with_and_without(_cfg(bcs, max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15), seq, off, monkeypatch)
    _with_and_without(_cfg(bcs, max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.22), seq, off, monkeypatch)


def test_pairs_tier_equals_one_tier(monkeypatch):
    """The pairs tier (tier 1 = same-diagonal pairs mode capped at cost 4 over the whole batch) against BDX_NO_TIER (one
    full-budget filter) on demo2's options: identical outputs incl. per-pass scores and deltas, counters, statistics-free."""
    bcs = synth.make_barcodes(96, 24, seed=81)
    seq, off, _ = synth.make_reads(bcs, 60000, 150, seed=82, sub=0.04, ins=0.01, dele=0.01, repeat=dict(frac=0.15))
    for kw in (dict(), dict(trim_side=5), dict(trim_side=3, summary=True)):
        cfg = _cfg(bcs, max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15, **kw)
        oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True)
        exp = oc.classify(seq, off)
        for env in (None, "BDX_NO_TIER"):
            if env:
                monkeypatch.setenv(env, "1")
            with H.bdx.HipClassifier(cfg, want_pass=True) as hc:
                if env:
                    monkeypatch.delenv(env)
                fuzz.assert_same(hc.classify(seq, off), exp, f"{kw} {env} [{hc.kernel_path}]")
                assert np.array_equal(hc.counts, oc.counts), (env, hc.kernel_path)
                assert hc.kernel_path.startswith("tier1:pairs(diag)") == (env is None), hc.kernel_path
