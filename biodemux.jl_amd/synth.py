"""Seeded synthetic batches of the shapes SURVEY.md §8(d) / BASELINE.json name.

Barcodes: length m over {A,C,G,T}, i.i.d. uniform, rejection-sampled to a minimum pairwise
Hamming distance.  Reads: i.i.d. uniform background of length n; ``plant_frac`` of the reads
carry one uniformly chosen barcode at a uniform start, mutated per base with substitution /
insertion / deletion rates; ``n_rate`` of all bases become 'N'.  Everything is generated in
independent 2^18-read chunks seeded by (seed, chunk index), so rank r of a multi-GPU run can
produce exactly its shard of the global batch without generating the rest.

The output is the packed-chunk layout of the C-ABI: one uint8 byte vector + int64 offsets.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

SEED = 20260515
CHUNK = 1 << 18
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_barcodes(n: int, m: int = 24, seed: int = SEED, min_hamming: int = 8,
                  lengths: Optional[Sequence[int]] = None) -> List[str]:
    """n barcodes; all of length m unless ``lengths`` (per-barcode) is given (C5)."""
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0xBC])))
    out: List[np.ndarray] = []
    while len(out) < n:
        L = int(lengths[len(out)]) if lengths is not None else m
        c = rng.integers(0, 4, size=L, dtype=np.uint8)
        ok = True
        for o in out:
            k = min(len(o), L)
            if int(np.count_nonzero(o[:k] != c[:k])) + abs(len(o) - L) < min_hamming:
                ok = False
                break
        if ok:
            out.append(c)
    return [_ACGT[c].tobytes().decode() for c in out]


def _chunk(bc_codes: List[np.ndarray], n_reads: int, read_len, rng: np.random.Generator, plant_frac: float,
           sub: float, ins: float, dele: float, n_rate: float, plant_lo: int, plant_hi: Optional[int],
           reads: Optional[np.ndarray] = None):
    n = read_len
    if reads is None:  # background; otherwise plant into the given reads (second barcode set)
        reads = _ACGT[(np.frombuffer(rng.bytes(n_reads * n), dtype=np.uint8) & 3)].reshape(n_reads, n).copy()
    truth = np.zeros(n_reads, dtype=np.int32)
    planted = np.flatnonzero(rng.random(n_reads) < plant_frac)
    if len(planted) and len(bc_codes):
        which = rng.integers(0, len(bc_codes), size=len(planted))
        truth[planted] = which + 1
        maxm = max(len(b) for b in bc_codes)
        bcm = np.zeros((len(bc_codes), maxm), dtype=np.uint8)
        bcl = np.zeros(len(bc_codes), dtype=np.int64)
        for i, b in enumerate(bc_codes):
            bcm[i, :len(b)] = b
            bcl[i] = len(b)
        base = bcm[which]                                   # [P, maxm] codes 0..3
        L0 = bcl[which]
        P = len(planted)
        inlen = np.arange(maxm)[None, :] < L0[:, None]
        u = rng.random((P, maxm))
        is_sub = u < sub
        is_del = (u >= sub) & (u < sub + dele)
        is_ins = rng.random((P, maxm)) < ins
        shift = rng.integers(1, 4, size=(P, maxm), dtype=np.uint8)
        base = np.where(is_sub, (base + shift) & 3, base)
        insb = rng.integers(0, 4, size=(P, maxm), dtype=np.uint8)
        # slot 2i = the barcode base (dropped when deleted), slot 2i+1 = an inserted base
        emit = np.empty((P, 2 * maxm), dtype=np.uint8)
        emit[:, 0::2] = base
        emit[:, 1::2] = insb
        valid = np.empty((P, 2 * maxm), dtype=bool)
        valid[:, 0::2] = inlen & ~is_del
        valid[:, 1::2] = inlen & is_ins
        pos = np.cumsum(valid, axis=1) - 1
        Lm = valid.sum(axis=1)
        hi = (n - Lm) if plant_hi is None else np.minimum(n - Lm, plant_hi)
        lo = np.minimum(plant_lo, np.maximum(hi, 0))
        start = lo + (rng.random(P) * (np.maximum(hi, lo) - lo + 1)).astype(np.int64)
        start = np.clip(start, 0, np.maximum(n - Lm, 0))
        rows = np.repeat(planted, 2 * maxm).reshape(P, 2 * maxm)
        cols = start[:, None] + pos
        ok = valid & (cols < n)
        reads[rows[ok], cols[ok]] = _ACGT[emit[ok]]
    if n_rate > 0:
        k = int(round(n_reads * n * n_rate))
        if k:
            idx = rng.integers(0, n_reads * n, size=k)
            reads.reshape(-1)[idx] = ord("N")
    return reads, truth


def make_reads(barcodes: Sequence[str], n_reads: int, read_len: int = 150, seed: int = SEED,
               first_read: int = 0, plant_frac: float = 0.9, sub: float = 0.02, ins: float = 0.005,
               dele: float = 0.005, n_rate: float = 0.001, plant_lo: int = 0, plant_hi: Optional[int] = None,
               second: Optional[Tuple[Sequence[str], int, Optional[int]]] = None):
    """Returns ``(seq_bytes uint8[n_reads*read_len], seq_off int64[n_reads+1], truth int32[n_reads])``.

    ``first_read`` (a multiple of CHUNK) selects where in the global stream this shard starts.
    ``second = (barcodes2, plant_lo2, plant_hi2)`` plants a second barcode set (dual mode, C4)
    into the same reads afterwards.
    """
    assert first_read % CHUNK == 0, "shards start on a chunk boundary"
    codes = [(np.frombuffer(b.encode(), dtype=np.uint8) >> 1) & 3 for b in barcodes]
    # A=0x41->0, C=0x43->1, G=0x47->3, T=0x54->2 ; remap to the ACGT order used by _ACGT
    remap = np.array([0, 1, 3, 2], dtype=np.uint8)
    codes = [remap[c] for c in codes]
    codes2 = None
    if second is not None:
        codes2 = [remap[(np.frombuffer(b.encode(), dtype=np.uint8) >> 1) & 3] for b in second[0]]
    out = np.empty((n_reads, read_len), dtype=np.uint8)
    truth = np.empty(n_reads, dtype=np.int32)
    done = 0
    chunk_id = first_read // CHUNK
    while done < n_reads:
        k = min(CHUNK, n_reads - done)
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0x5EAD, chunk_id])))
        r, t = _chunk(codes, CHUNK if k == CHUNK else k, read_len, rng, plant_frac, sub, ins, dele, n_rate,
                      plant_lo, plant_hi)
        if codes2 is not None:
            rng2 = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0x5EAE, chunk_id])))
            _chunk(codes2, len(r), read_len, rng2, plant_frac, sub, ins, dele, 0.0, second[1], second[2], reads=r)
        out[done:done + k] = r[:k]
        truth[done:done + k] = t[:k]
        done += k
        chunk_id += 1
    off = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return out.reshape(-1), off, truth


def make_ragged_reads(barcodes: Sequence[str], n_reads: int, min_len: int, max_len: int, seed: int = SEED, **kw):
    """Variable-length reads (edge-case tests): generated at max_len and cut to a per-read length."""
    seq, off, truth = make_reads(barcodes, n_reads, max_len, seed, **kw)
    rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, 0x4A66])))
    lens = rng.integers(min_len, max_len + 1, size=n_reads)
    mat = seq.reshape(n_reads, max_len)
    keep = np.arange(max_len)[None, :] < lens[:, None]
    new_off = np.zeros(n_reads + 1, dtype=np.int64)
    new_off[1:] = np.cumsum(lens)
    return mat[keep].copy(), new_off, truth
