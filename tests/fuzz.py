"""Randomised differential cases shared by the CPU and GPU test files: seeded configs over the
whole option space of the hot path (algorithm, costs, N-scoring, rates, min_delta, the three
range kinds, trimming, dual mode, summary/traceback, ragged reads incl. empty ones)."""
from __future__ import annotations

import numpy as np

import helpers as H
from biodemux_jl_amd import synth

_RANGES = ["1:end", "1:end", "1:end", "1:30", "5:60", "end-40:end", "end-20:end-3", "1:8", "10:9", "3:end-5"]
_START_RANGES = ["1:end", "1:end", "1:end", "1:10", "1:1", "5:40", "end-30:end"]
_END_RANGES = ["1:end", "1:end", "1:end", "20:end", "end-10:end", "1:40", "30:end"]


def _rand_barcodes(rng, n, lo, hi, with_n):
    out = []
    for _ in range(n):
        m = int(rng.integers(lo, hi + 1))
        s = "".join("ACGT"[int(c)] for c in rng.integers(0, 4, size=m))
        if with_n and rng.random() < 0.5:
            s = list(s)
            for _k in range(int(rng.integers(1, 3))):
                s[int(rng.integers(0, m))] = "N"
            s = "".join(s)
        out.append(s)
    return out


def random_case(seed: int, n_reads: int = 600):
    """Returns (DemuxConfig, seq_bytes, seq_off)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    alg = ["semiglobal", "semiglobal", "semiglobal", "hamming", "exact"][int(rng.integers(0, 5))]
    with_n = rng.random() < 0.3
    B = int(rng.integers(1, 40))
    lo = int(rng.integers(4, 20))
    hi = lo if rng.random() < 0.6 else lo + int(rng.integers(1, 16))
    bcs = _rand_barcodes(rng, B, lo, hi, with_n)
    dual = rng.random() < 0.3
    bcs2 = _rand_barcodes(rng, int(rng.integers(1, 12)), lo, hi, with_n) if dual else []
    kw = dict(
        bc_seqs=bcs, bc_lengths_no_N=[sum(c != "N" for c in b) for b in bcs], ids=[f"a{i}" for i in range(len(bcs))],
        max_error_rate=float([0.0, 0.05, 0.1, 0.2, 0.25, 0.3, 0.5][int(rng.integers(0, 7))]),
        min_delta=float([0.0, 0.0, 0.05, 0.15, 0.3][int(rng.integers(0, 5))]),
        match=0, mismatch=int(rng.integers(1, 4)), indel=int(rng.integers(1, 4)),
        nindel=(int(rng.integers(1, 3)) if (with_n or rng.random() < 0.15) else None),
        ref_search_range=H.bdx.parse_dynamic_range(_RANGES[int(rng.integers(0, len(_RANGES)))]),
        barcode_start_range=H.bdx.parse_dynamic_range(_START_RANGES[int(rng.integers(0, len(_START_RANGES)))]),
        barcode_end_range=H.bdx.parse_dynamic_range(_END_RANGES[int(rng.integers(0, len(_END_RANGES)))]),
        trim_side=[None, None, 3, 5][int(rng.integers(0, 4))],
        summary=bool(rng.random() < 0.2),
        matching_algorithm=alg,
    )
    if dual:
        kw.update(
            is_dual=True, bc_seqs2=bcs2, bc_lengths_no_N2=[sum(c != "N" for c in b) for b in bcs2],
            ids2=[f"b{i}" for i in range(len(bcs2))],
            ref_search_range2=H.bdx.parse_dynamic_range(_RANGES[int(rng.integers(0, len(_RANGES)))]),
            barcode_start_range2=H.bdx.parse_dynamic_range(_START_RANGES[int(rng.integers(0, len(_START_RANGES)))]),
            barcode_end_range2=H.bdx.parse_dynamic_range(_END_RANGES[int(rng.integers(0, len(_END_RANGES)))]),
            trim_side2=[None, 3, 5][int(rng.integers(0, 3))])
    cfg = H.bdx.DemuxConfig(**kw)
    max_len = int(rng.integers(max(hi, 8), 160))
    plain = [b.replace("N", "A") for b in bcs]
    second = ([b.replace("N", "C") for b in bcs2], max_len // 2, None) if dual else None
    seq, off, _ = synth.make_ragged_reads(plain, n_reads, 0, max_len, seed=seed, plant_frac=0.8, sub=0.04,
                                          ins=0.015, dele=0.015, n_rate=0.005,
                                          plant_hi=(max_len // 3 if dual else None), second=second)
    return cfg, seq, off


def random_case_many_barcodes(seed: int, n_reads: int = 1500):
    """Like random_case but in the domain of the seeded variants: 48..160 barcodes of 20..32 nt, reads of up
    to 152 (sometimes longer) bases, rates around the budgets where single seeds / two intact pieces apply."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    alg = ["semiglobal", "semiglobal", "semiglobal", "semiglobal", "hamming"][int(rng.integers(0, 5))]
    B = int(rng.integers(48, 160))
    lo = int([20, 24, 24, 28, 32][int(rng.integers(0, 5))])
    hi = lo if rng.random() < 0.7 else min(32, lo + int(rng.integers(1, 9)))
    bcs = _rand_barcodes(rng, B, lo, hi, False)
    dual = rng.random() < 0.25
    bcs2 = _rand_barcodes(rng, int(rng.integers(8, 40)), lo, hi, False) if dual else []
    unit = rng.random() < 0.7
    kw = dict(
        bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"a{i}" for i in range(len(bcs))],
        max_error_rate=float([0.1, 0.13, 0.17, 0.2, 0.2, 0.22, 0.25][int(rng.integers(0, 7))]),
        min_delta=float([0.0, 0.0, 0.05, 0.1][int(rng.integers(0, 4))]),
        match=0, mismatch=1 if unit else int(rng.integers(1, 3)), indel=1 if unit else int(rng.integers(1, 3)),
        nindel=None,
        ref_search_range=H.bdx.parse_dynamic_range(["1:end", "1:end", "1:end", "5:end-3", "1:120"][int(rng.integers(0, 5))]),
        trim_side=[None, None, 3, 5][int(rng.integers(0, 4))],
        summary=bool(rng.random() < 0.15),
        matching_algorithm=alg,
    )
    if dual:
        kw.update(is_dual=True, bc_seqs2=bcs2, bc_lengths_no_N2=[len(b) for b in bcs2], ids2=[f"b{i}" for i in range(len(bcs2))],
                  trim_side2=[None, 3, 5][int(rng.integers(0, 3))])
    cfg = H.bdx.DemuxConfig(**kw)
    max_len = int([100, 150, 150, 152, 151, 200][int(rng.integers(0, 6))])
    second = (bcs2, max_len // 2, None) if dual else None
    # every third seed: half of the reads are concatemers (the same barcode twice, two barcodes, shifted copies)
    repeat = dict(frac=0.5) if seed % 3 == 2 else None
    seq, off, _ = synth.make_ragged_reads(bcs, n_reads, max_len // 2 if rng.random() < 0.5 else max_len, max_len, seed=seed,
                                          plant_frac=0.85, sub=0.05, ins=0.02, dele=0.02, n_rate=0.003,
                                          plant_hi=(max_len // 3 if dual else None), second=second, repeat=repeat)
    return cfg, seq, off


def random_case_wide(seed: int, n_reads: int = 1500):
    """Round-3 domains: 100..520 barcodes of 24 nt (the wave kernel with queues sized from the chance hits, the pairs mode
    with groups of 128 barcodes), or 16..60 barcodes of 65..128 nt (128-bit sweep words); mostly known-score configs at
    rates 0.1..0.2, some with min_delta / trimming / summary; ragged reads, concatemers."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x31DE))
    long_bc = rng.random() < 0.25
    if long_bc:
        B = int(rng.integers(16, 61))
        lo = int(rng.integers(65, 129))
        hi = lo if rng.random() < 0.5 else min(128, lo + int(rng.integers(1, 30)))
        rate = float([0.05, 0.08, 0.1, 0.12][int(rng.integers(0, 4))])
        max_len = int([200, 260, 320][int(rng.integers(0, 3))])
        err = 0.03
    else:
        B = int(rng.integers(100, 521))
        lo = int([24, 24, 24, 20, 28, 32][int(rng.integers(0, 6))])
        hi = lo if rng.random() < 0.7 else min(32, lo + int(rng.integers(1, 5)))
        rate = float([0.1, 0.1, 0.13, 0.17, 0.2, 0.2][int(rng.integers(0, 6))])
        max_len = int([100, 150, 150, 152, 200][int(rng.integers(0, 5))])
        err = float([0.02, 0.04, 0.06][int(rng.integers(0, 3))])
    bcs = _rand_barcodes(rng, B, lo, hi, False)
    kw = dict(
        bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"a{i}" for i in range(len(bcs))],
        max_error_rate=rate, min_delta=float([0.0, 0.0, 0.0, 0.05, 0.1][int(rng.integers(0, 5))]),
        trim_side=[None, None, None, None, 3, 5][int(rng.integers(0, 6))],
        summary=bool(rng.random() < 0.1),
        matching_algorithm=["semiglobal"] * 6 + ["hamming"],
    )
    kw["matching_algorithm"] = kw["matching_algorithm"][int(rng.integers(0, 7))]
    cfg = H.bdx.DemuxConfig(**kw)
    repeat = dict(frac=0.3) if rng.random() < 0.25 else None
    seq, off, _ = synth.make_ragged_reads(bcs, n_reads, max_len // 2 if rng.random() < 0.4 else max_len, max_len, seed=seed,
                                          plant_frac=0.85, sub=err, ins=err / 3, dele=err / 3, n_rate=0.003, repeat=repeat)
    return cfg, seq, off


def random_case_tiers(seed: int, n_reads: int = 1500):
    """The tiered budgets' domain and its borders: 16..160 barcodes of 12..64 nt (capped budgets 0..7, 32- and 64-bit
    sweep words), rates whose full budget is 1..3 operations beyond the cap, min_delta around the score of an unseen
    barcode, weighted costs (cmin 1..2), N-scoring, dual, trimming / summary, start / end ranges that bind for some
    reads, column windows, ragged reads, concatemers."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x71E5))
    alg = ["semiglobal"] * 5 + ["hamming"]
    alg = alg[int(rng.integers(0, len(alg)))]
    B = int(rng.integers(16, 161))
    lo = int([12, 16, 20, 24, 24, 28, 32, 40, 48, 64][int(rng.integers(0, 10))])
    hi = lo if rng.random() < 0.5 else min(64, lo + int(rng.integers(1, 17)))
    with_n = rng.random() < 0.15  # N in barcodes: a wildcard under N-scoring and :hamming, a literal fifth symbol otherwise
    bcs = _rand_barcodes(rng, B, lo, hi, with_n)
    dual = rng.random() < 0.3
    bcs2 = _rand_barcodes(rng, int(rng.integers(8, 64)), lo, hi, with_n) if dual else []
    unit = rng.random() < 0.6
    rate = float([0.13, 0.15, 0.17, 0.2, 0.2, 0.22, 0.25, 0.3][int(rng.integers(0, 8))])
    md = float([0.0, 0.0, 0.02, 0.04, 0.05, 0.08, 0.1, 0.13, 0.2][int(rng.integers(0, 9))])
    rngs = ["1:end"] * 6 + ["5:end-3", "1:120", "1:60", "20:end", "end-80:end"]
    kw = dict(
        bc_seqs=bcs, bc_lengths_no_N=[sum(c != "N" for c in b) for b in bcs], ids=[f"a{i}" for i in range(len(bcs))],
        max_error_rate=rate, min_delta=md, match=0,
        mismatch=1 if unit else int(rng.integers(1, 4)), indel=1 if unit else int(rng.integers(1, 4)),
        nindel=(int(rng.integers(1, 3)) if rng.random() < (0.6 if with_n else 0.1) else None),
        ref_search_range=H.bdx.parse_dynamic_range(rngs[int(rng.integers(0, len(rngs)))]),
        barcode_start_range=H.bdx.parse_dynamic_range(rngs[int(rng.integers(0, len(rngs)))] if rng.random() < 0.2 else "1:end"),
        barcode_end_range=H.bdx.parse_dynamic_range(rngs[int(rng.integers(0, len(rngs)))] if rng.random() < 0.2 else "1:end"),
        trim_side=[None, None, None, 3, 5][int(rng.integers(0, 5))],
        summary=bool(rng.random() < 0.12),
        matching_algorithm=alg,
    )
    if dual:
        kw.update(is_dual=True, bc_seqs2=bcs2, bc_lengths_no_N2=[sum(c != "N" for c in b) for b in bcs2],
                  ids2=[f"b{i}" for i in range(len(bcs2))], trim_side2=[None, None, 3, 5][int(rng.integers(0, 4))])
    cfg = H.bdx.DemuxConfig(**kw)
    max_len = int([100, 150, 150, 180, 250][int(rng.integers(0, 5))])
    max_len = max(max_len, hi + 20)
    second = (bcs2, max_len // 2, None) if dual else None
    repeat = dict(frac=0.3) if rng.random() < 0.3 else None
    err = float([0.02, 0.04, 0.06, 0.08][int(rng.integers(0, 4))])
    plant = [b.replace("N", "ACGT"[int(rng.integers(0, 4))]) for b in bcs]
    if second is not None:
        second = ([b.replace("N", "C") for b in bcs2], second[1], second[2])
    seq, off, _ = synth.make_ragged_reads(plant, n_reads, max_len // 2 if rng.random() < 0.5 else max_len, max_len, seed=seed,
                                          plant_frac=0.85, sub=err, ins=err / 3, dele=err / 3, n_rate=0.003,
                                          plant_hi=(max_len // 3 if dual else None), second=second, repeat=repeat)
    return cfg, seq, off


def random_case_band(seed: int, n_reads: int = 1500, m_choices=None, read_lens=None, mixed: bool = False):
    """The diagonal-band DP's domain (exact stage, clean class, every barcode of the config with the same 8, 10, 12, 16,
    20, 24 or 32 bases):
    traceback through trimming or summary, or weighted costs; budgets 0..4 and beyond (fallback), tiers, column
    windows that start inside the read, barcodes hanging over either end of the read, concatemers, low-complexity
    barcodes (wide end-column windows: the 17-diagonal body or the all-rows fallback), dual."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0xBA9D))
    mc = m_choices or [24, 24, 24, 32, 32, 8, 10, 12, 16, 20]
    m = int(mc[int(rng.integers(0, len(mc)))])
    m_lo = max(33, m - int(rng.integers(1, 30))) if (mixed and m > 34 and rng.random() < 0.4) else m  # (long barcodes: sometimes mixed lengths)
    B = int(rng.integers(8, 140 if m_choices is None else 60))
    lowc = rng.random() < 0.15
    if lowc:
        bcs = ["".join("AC"[int(x)] for x in rng.integers(0, 2, size=int(rng.integers(m_lo, m + 1)))) for _ in range(B)]
        bcs = list(dict.fromkeys(bcs))
    else:
        bcs = _rand_barcodes(rng, B, m_lo, m, False)
    dual = rng.random() < 0.3
    bcs2 = _rand_barcodes(rng, int(rng.integers(8, 48)), m_lo, m, False) if dual else []
    unit = rng.random() < 0.6
    trim = [None, 3, 3, 5, 5][int(rng.integers(0, 5))]
    summary = bool(rng.random() < (0.6 if trim is None and unit else 0.15))
    rngs = ["1:end"] * 5 + ["5:end-3", "1:120", "20:end", "end-90:end"]
    kw = dict(
        bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"a{i}" for i in range(len(bcs))],
        max_error_rate=float([0.05, 0.1, 0.1, 0.13, 0.17, 0.2, 0.2, 0.25][int(rng.integers(0, 8))]),
        min_delta=float([0.0, 0.0, 0.05, 0.1][int(rng.integers(0, 4))]),
        match=0, mismatch=1 if unit else int(rng.integers(1, 4)), indel=1 if unit else int(rng.integers(1, 4)),
        ref_search_range=H.bdx.parse_dynamic_range(rngs[int(rng.integers(0, len(rngs)))]),
        trim_side=trim, summary=summary,
    )
    if dual:
        kw.update(is_dual=True, bc_seqs2=bcs2, bc_lengths_no_N2=[len(b) for b in bcs2], ids2=[f"b{i}" for i in range(len(bcs2))],
                  trim_side2=[None, 3, 5][int(rng.integers(0, 3))])
    cfg = H.bdx.DemuxConfig(**kw)
    rl = read_lens or [60, 100, 150, 150, 200]
    max_len = int(rl[int(rng.integers(0, len(rl)))])
    second = (bcs2, max_len // 2, None) if dual else None
    repeat = dict(frac=0.3) if rng.random() < 0.3 else None
    err = float([0.0, 0.02, 0.04, 0.07][int(rng.integers(0, 4))])
    seq, off, _ = synth.make_ragged_reads(bcs, n_reads, max_len // 2 if rng.random() < 0.5 else max_len, max_len, seed=seed,
                                          plant_frac=0.85, sub=err, ins=err / 3, dele=err / 3, n_rate=0.003,
                                          plant_hi=(max_len // 3 if dual else None), second=second, repeat=repeat)
    # barcodes hanging over the ends of the read: a prefix cut off at the read's start, a suffix at its end
    seq = seq.copy()
    for i in range(0, n_reads, 7):
        n = int(off[i + 1] - off[i])
        b = np.frombuffer(bcs[int(rng.integers(0, len(bcs)))].encode(), dtype=np.uint8)
        cut = int(rng.integers(1, 5))
        mb = len(b)
        if n >= mb:
            if i % 2:
                seq[off[i]: off[i] + mb - cut] = b[cut:]
            else:
                seq[off[i + 1] - (mb - cut): off[i + 1]] = b[:mb - cut]
    return cfg, seq, off


def random_case_band_long(seed: int, n_reads: int = 600):
    """The rolling diagonal band's domain (exact stage, clean class, barcodes of 33 .. 128 bases, one length or mixed):
    the generator above with long barcodes and longer reads."""
    return random_case_band(seed ^ 0x10C6, n_reads=n_reads, m_choices=[33, 40, 48, 64, 64, 80, 80, 100, 128],
                            read_lens=[120, 200, 300, 300, 400], mixed=True)


def permuted_batch(seq: np.ndarray, off: np.ndarray, seed: int):
    """The same reads in another order, a few dropped: (seq, off, index of each new read in the old batch).  Classifying
    it on a context that has just classified the original batch makes every stale per-read buffer entry a WRONG
    one (same indices, other reads) — reads of unwritten device memory turn into mismatches."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x9E37))
    n = len(off) - 1
    keep = rng.permutation(n)[: max(1, n - int(rng.integers(0, max(2, n // 10))))]
    lens = (off[1:] - off[:-1])[keep]
    noff = np.zeros(len(keep) + 1, dtype=np.int64)
    noff[1:] = np.cumsum(lens)
    nseq = np.empty(int(noff[-1]), dtype=np.uint8)
    for k, i in enumerate(keep):
        nseq[noff[k]:noff[k + 1]] = seq[off[i]:off[i + 1]]
    return nseq, noff, keep


def expected_permuted(exp: dict, keep: np.ndarray, n_old: int) -> dict:
    out = {}
    for k, v in exp.items():
        if v.shape[0] == n_old:
            out[k] = v[keep]
        elif v.shape[0] == 2 * n_old:
            out[k] = v.reshape(n_old, 2)[keep].reshape(-1)
        else:
            out[k] = v
    return out


def assert_same(got: dict, exp: dict, what: str = ""):
    for k in ("bc1", "bc2", "keep_start", "keep_end", "pass_bc", "pass_start", "pass_end"):
        if k in got and k in exp:
            bad = np.flatnonzero((got[k] != exp[k]).reshape(len(got[k]), -1).any(axis=1))
            assert bad.size == 0, f"{what}: {k} differs at reads {bad[:8].tolist()} (got {got[k][bad[:4]].tolist()} exp {exp[k][bad[:4]].tolist()})"
    for k in ("pass_score", "pass_delta"):
        if k in got and k in exp:
            # bit-exact doubles; NaN == NaN (Inf - Inf in the with_delta reducer)
            a, b = got[k].view(np.uint64), exp[k].view(np.uint64)
            nan = np.isnan(got[k]) & np.isnan(exp[k])
            bad = np.flatnonzero(((a != b) & ~nan).any(axis=1))
            assert bad.size == 0, f"{what}: {k} differs at reads {bad[:8].tolist()} (got {got[k][bad[:4]].tolist()} exp {exp[k][bad[:4]].tolist()})"
