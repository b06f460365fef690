"""Host-side mirror of the functions BioDemuX.jl exports from src/classification.jl, each
evaluated by the gfx950 kernel through the C-ABI (never on the CPU).

  semiglobal_alignment      classification.jl:447
  semiglobal_alignment_N    classification.jl:463
  exact_align               classification.jl:485
  hamming_align             classification.jl:557
  find_best_matching_bc     classification.jl:722
  determine_filename        classification.jl:871
  DemuxStats / merge_stats  classification.jl:736-767 / reporting.jl:1-58 (scalar counters from the
                            device; pos/len/score histograms accumulated on the host)

Argument order and meaning follow the Julia signatures so the parity tests read like the
reference's own unit tests.  ``ref_search_range`` is a ``(first, last)`` tuple or a Python
``range`` standing for Julia's ``first:last`` (1-based, inclusive).  ``ws`` (the
SemiGlobalWorkspace) is accepted and ignored: DP columns live in LDS on the device.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

from .config import DemuxConfig
from .hipabi import HipClassifier, pack_reads
from .ranges import DynamicRange


class SemiGlobalWorkspace:
    """classification.jl:1-6.  Kept for signature compatibility; the device owns the DP."""

    def __init__(self, max_m: int = 0, trim: bool = False):
        self.max_m = max_m
        self.origin = [0] * max_m if trim else None
        self.DP = None


def _rng(r) -> Tuple[int, int]:
    """Julia UnitRange -> (first, last) with the empty-range normalisation last = first-1."""
    if isinstance(r, range):
        a, b = r.start, r.stop - 1
    else:
        a, b = int(r[0]), int(r[1])
    if b < a:
        b = a - 1
    return a, b


_FULL = DynamicRange(1, False, 0, True)


def _one(cfg: DemuxConfig, seq: str, window, device: int = 0, filter: str = "auto"):
    with HipClassifier(cfg, device=device, want_pass=True, windows={0: window}, filter=filter) as hc:
        blob, off = pack_reads([seq])
        return hc.classify(blob, off)


def _align_cfg(query: str, max_error: float, match: int, mismatch: int, indel: int, nindel, non_N_m,
               trim_side, need_traceback: bool, algorithm: str) -> DemuxConfig:
    return DemuxConfig(
        bc_seqs=[query], bc_lengths_no_N=[len(query) if non_N_m is None else non_N_m], ids=["q"],
        max_error_rate=max_error, match=match, mismatch=mismatch, indel=indel, nindel=nindel,
        trim_side=trim_side, summary=need_traceback, matching_algorithm=algorithm)


def semiglobal_alignment(ws, query: str, ref: str, max_error: float, match: int, mismatch: int, indel: int,
                         ref_search_range, max_start_pos: int, min_end_pos: int, trim_side: Optional[int] = None,
                         need_traceback: bool = False, *, device: int = 0, filter: str = "auto"):
    """classification.jl:447-461.  Returns a float (ScoreOnly) or ``(score, start, end)``."""
    cfg = _align_cfg(query, max_error, match, mismatch, indel, None, None, trim_side, need_traceback, "semiglobal")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device, filter)
    score = float(out["pass_score"][0, 0])
    if trim_side is None and not need_traceback:
        return score
    return (score, int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def semiglobal_alignment_N(ws, query: str, ref: str, max_error: float, match: int, mismatch: int, indel: int,
                           nindel: int, ref_search_range, max_start_pos: int, min_end_pos: int, non_N_m: int,
                           trim_side: Optional[int] = None, need_traceback: bool = False, *, device: int = 0,
                           filter: str = "auto"):
    """classification.jl:463-477."""
    cfg = _align_cfg(query, max_error, match, mismatch, indel, nindel, non_N_m, trim_side, need_traceback,
                     "semiglobal")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device, filter)
    score = float(out["pass_score"][0, 0])
    if trim_side is None and not need_traceback:
        return score
    return (score, int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def exact_align(query: str, ref: str, ref_search_range, max_start_pos: int, min_end_pos: int,
                trim_side: Optional[int], *, device: int = 0, filter: str = "auto"):
    """classification.jl:485-548.  ``(0.0, s, e)`` or ``(Inf, -1, -1)``."""
    cfg = _align_cfg(query, 0.0, 0, 1, 1, None, None, trim_side, False, "exact")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device, filter)
    return (float(out["pass_score"][0, 0]), int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def hamming_align(query: str, ref: str, max_error_rate: float, ref_search_range, max_start_pos: int,
                  min_end_pos: int, trim_side: Optional[int], *, device: int = 0, filter: str = "auto"):
    """classification.jl:557-625."""
    cfg = _align_cfg(query, max_error_rate, 0, 1, 1, None, None, trim_side, False, "hamming")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device, filter)
    return (float(out["pass_score"][0, 0]), int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def find_best_matching_bc(seq: str, bc_seqs: List[str], bc_lengths_no_N: List[int], config: DemuxConfig, ws,
                          ref_search_range, max_start_pos: int, min_end_pos: int, trim_side: Optional[int],
                          need_traceback: bool = False, *, device: int = 0, filter: str = "auto"):
    """classification.jl:722-728.  Returns ``(min_score_bc, min_score, delta, best_start, best_end)``."""
    import copy

    cfg = copy.copy(config)
    cfg.bc_seqs, cfg.bc_lengths_no_N = list(bc_seqs), list(bc_lengths_no_N)
    cfg.ids = [str(i) for i in range(len(bc_seqs))]
    cfg.is_dual = False
    cfg.trim_side = trim_side
    # need_tb = trim_side !== nothing || stats !== nothing (:812); here it is given directly
    cfg.summary = bool(need_traceback)
    a, b = _rng(ref_search_range)
    out = _one(cfg, seq, (a, b, max_start_pos, min_end_pos, 1), device, filter)
    return (int(out["pass_bc"][0, 0]), float(out["pass_score"][0, 0]), float(out["pass_delta"][0, 0]),
            int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def filename_for(config: DemuxConfig, bc1: int, bc2: int) -> str:
    """The filename part of determine_filename (classification.jl:877-899)."""
    suffix = ".fastq.gz" if config.gzip_output else ".fastq"
    if bc1 == 0:
        return "unknown" + suffix
    if bc1 < 0:
        return "ambiguous_classification" + suffix
    if config.is_dual:
        return str(config.ids[bc1 - 1]) + "." + str(config.ids2[bc2 - 1]) + suffix
    return str(config.ids[bc1 - 1]) + suffix


def determine_filename(seq: str, config: DemuxConfig, ws=None, *, device: int = 0, filter: str = "auto"):
    """classification.jl:871-938.  Returns ``(filename, keep_start, keep_end)``."""
    with HipClassifier(config, device=device, filter=filter) as hc:
        blob, off = pack_reads([seq])
        out = hc.classify(blob, off)
    bc1, bc2 = int(out["bc1"][0]), int(out["bc2"][0])
    return filename_for(config, bc1, bc2), int(out["keep_start"][0]), int(out["keep_end"][0])


def _round2(x):
    """Julia's round(score, digits=2) (classification.jl:835): round-half-even of fl(x * 100), / 100."""
    import numpy as np

    return np.round(np.asarray(x, dtype=np.float64) * 100.0) / 100.0


@dataclass
class DemuxStats:
    """classification.jl:736-758.  The scalar counters and sample_counts come from the device
    (C-ABI counter vector); the position / length / score histograms (filled at :827-865 for every
    pass whose status is :match) are accumulated on the host from the per-pass outputs."""

    total_reads: int = 0
    matched_reads: int = 0
    unmatched_reads: int = 0
    ambiguous_reads: int = 0
    sample_counts: Dict[Tuple[int, int], int] = field(default_factory=dict)
    bc1_pos_counts: Dict[int, int] = field(default_factory=dict)
    bc1_len_counts: Dict[int, int] = field(default_factory=dict)
    bc1_score_counts: Dict[float, int] = field(default_factory=dict)
    bc1_per_bc_score_counts: Dict[int, Dict[float, int]] = field(default_factory=dict)
    bc1_per_bc_pos_counts: Dict[int, Dict[int, int]] = field(default_factory=dict)
    bc1_per_bc_len_counts: Dict[int, Dict[int, int]] = field(default_factory=dict)
    bc2_pos_counts: Dict[int, int] = field(default_factory=dict)
    bc2_len_counts: Dict[int, int] = field(default_factory=dict)
    bc2_score_counts: Dict[float, int] = field(default_factory=dict)
    bc2_per_bc_score_counts: Dict[int, Dict[float, int]] = field(default_factory=dict)
    bc2_per_bc_pos_counts: Dict[int, Dict[int, int]] = field(default_factory=dict)
    bc2_per_bc_len_counts: Dict[int, Dict[int, int]] = field(default_factory=dict)

    @classmethod
    def from_counts(cls, counts, n_bc1: int, n_bc2: int) -> "DemuxStats":
        """Decode the int64 counter vector of the C-ABI (bdx_get_counts)."""
        stride = max(1, n_bc2)
        s = cls(int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3]))
        for k in range(4, len(counts)):
            c = int(counts[k])
            if c:
                b1, b2 = divmod(k - 4, stride)
                s.sample_counts[(b1 + 1, b2 + 1 if n_bc2 else 0)] = c
        return s

    def add_device_tables(self, tables: dict, config) -> None:
        """Decode the histogram tables of the C-ABI (bdx_get_stats, hipabi.HipClassifier.stats_tables): per pass
        [key][barcode] counts of best_start, best_end - best_start + 1 and the integer score numerator; the score
        keys are round(raw / normalisation, digits = 2) exactly as classification.jl:835 computes them (Float64
        division, then Julia's round-half-even on the scaled value)."""
        import numpy as np

        for p, tag in ((0, "bc1"), (1, "bc2")):
            if p not in tables:
                continue
            seqs = config.bc_seqs if p == 0 else config.bc_seqs2
            nn = config.bc_lengths_no_N if p == 0 else config.bc_lengths_no_N2
            n_scoring = config.nindel is not None and str(config.matching_algorithm).lstrip(":") == "semiglobal"
            for name, field_ in (("pos", "pos_counts"), ("len", "len_counts"), ("raw", "score_counts")):
                tab, key0 = tables[p][name]
                g = getattr(self, f"{tag}_{field_}")
                per = getattr(self, f"{tag}_per_bc_{field_}")
                rows, cols = np.nonzero(tab)
                for r, b in zip(rows.tolist(), cols.tolist()):
                    c = int(tab[r, b])
                    if name == "raw":
                        norm = float(nn[b]) if n_scoring else float(len(seqs[b]))
                        key = float(_round2(np.float64(r) / np.float64(norm)))
                    else:
                        key = r + key0
                    g[key] = g.get(key, 0) + c
                    d = per.setdefault(b + 1, {})
                    d[key] = d.get(key, 0) + c

    def add_pass_outputs(self, out: dict, min_delta: float) -> None:
        """Histogram update of match_barcode_pass (classification.jl:827-865) for one batch: a pass
        contributes iff its status is :match (winner found and delta >= min_delta); pass 2 only ran
        when pass 1 matched (then pass_bc of pass 2 is non-zero or the read went unknown)."""
        import numpy as np

        for p, tag in ((0, "bc1"), (1, "bc2")):
            bc = out["pass_bc"][:, p]
            ok = (bc > 0) & ~(out["pass_delta"][:, p] < min_delta)
            if not ok.any():
                continue
            bc = bc[ok].astype(np.int64)
            start = out["pass_start"][:, p][ok].astype(np.int64)
            length = out["pass_end"][:, p][ok].astype(np.int64) - start + 1
            score = _round2(out["pass_score"][:, p][ok])

            def bump(d, keys, cast):
                u, c = np.unique(keys, return_counts=True)
                for k, n in zip(u, c):
                    d[cast(k)] = d.get(cast(k), 0) + int(n)

            bump(getattr(self, f"{tag}_pos_counts"), start, int)
            bump(getattr(self, f"{tag}_len_counts"), length, int)
            bump(getattr(self, f"{tag}_score_counts"), score, float)
            for b in np.unique(bc):
                m = bc == b
                bump(getattr(self, f"{tag}_per_bc_pos_counts").setdefault(int(b), {}), start[m], int)
                bump(getattr(self, f"{tag}_per_bc_len_counts").setdefault(int(b), {}), length[m], int)
                bump(getattr(self, f"{tag}_per_bc_score_counts").setdefault(int(b), {}), score[m], float)


_HIST_FIELDS = [f"{t}_{k}" for t in ("bc1", "bc2") for k in ("pos_counts", "len_counts", "score_counts")]
_PER_BC_FIELDS = [f"{t}_per_bc_{k}" for t in ("bc1", "bc2") for k in ("score_counts", "pos_counts", "len_counts")]


def merge_stats(stats_list: List[DemuxStats]) -> DemuxStats:
    """reporting.jl:1-58: sum of the per-worker statistics (across GPUs the scalar part is one RCCL
    all-reduce of the counter vector, see dist.py)."""
    m = DemuxStats()
    for s in stats_list:
        m.total_reads += s.total_reads
        m.matched_reads += s.matched_reads
        m.unmatched_reads += s.unmatched_reads
        m.ambiguous_reads += s.ambiguous_reads
        for k, v in s.sample_counts.items():
            m.sample_counts[k] = m.sample_counts.get(k, 0) + v
        for f in _HIST_FIELDS:
            d = getattr(m, f)
            for k, v in getattr(s, f).items():
                d[k] = d.get(k, 0) + v
        for f in _PER_BC_FIELDS:
            dm = getattr(m, f)
            for b, dd in getattr(s, f).items():
                t = dm.setdefault(b, {})
                for k, v in dd.items():
                    t[k] = t.get(k, 0) + v
    return m


def isinf(x: float) -> bool:
    return math.isinf(x)
