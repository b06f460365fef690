"""Host-side mirror of the functions BioDemuX.jl exports from src/classification.jl, each
evaluated by the gfx950 kernel through the C-ABI (never on the CPU).

  semiglobal_alignment      classification.jl:447
  semiglobal_alignment_N    classification.jl:463
  exact_align               classification.jl:485
  hamming_align             classification.jl:557
  find_best_matching_bc     classification.jl:722
  determine_filename        classification.jl:871
  DemuxStats / merge_stats  classification.jl:736-767 / reporting.jl:1-58 (scalar counters +
                            sample_counts only; the histograms are out of scope)

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


def _one(cfg: DemuxConfig, seq: str, window, device: int = 0):
    with HipClassifier(cfg, device=device, want_pass=True, windows={0: window}, filter="off") as hc:
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
                         need_traceback: bool = False, *, device: int = 0):
    """classification.jl:447-461.  Returns a float (ScoreOnly) or ``(score, start, end)``."""
    cfg = _align_cfg(query, max_error, match, mismatch, indel, None, None, trim_side, need_traceback, "semiglobal")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device)
    score = float(out["pass_score"][0, 0])
    if trim_side is None and not need_traceback:
        return score
    return (score, int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def semiglobal_alignment_N(ws, query: str, ref: str, max_error: float, match: int, mismatch: int, indel: int,
                           nindel: int, ref_search_range, max_start_pos: int, min_end_pos: int, non_N_m: int,
                           trim_side: Optional[int] = None, need_traceback: bool = False, *, device: int = 0):
    """classification.jl:463-477."""
    cfg = _align_cfg(query, max_error, match, mismatch, indel, nindel, non_N_m, trim_side, need_traceback,
                     "semiglobal")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device)
    score = float(out["pass_score"][0, 0])
    if trim_side is None and not need_traceback:
        return score
    return (score, int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def exact_align(query: str, ref: str, ref_search_range, max_start_pos: int, min_end_pos: int,
                trim_side: Optional[int], *, device: int = 0):
    """classification.jl:485-548.  ``(0.0, s, e)`` or ``(Inf, -1, -1)``."""
    cfg = _align_cfg(query, 0.0, 0, 1, 1, None, None, trim_side, False, "exact")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device)
    return (float(out["pass_score"][0, 0]), int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def hamming_align(query: str, ref: str, max_error_rate: float, ref_search_range, max_start_pos: int,
                  min_end_pos: int, trim_side: Optional[int], *, device: int = 0):
    """classification.jl:557-625."""
    cfg = _align_cfg(query, max_error_rate, 0, 1, 1, None, None, trim_side, False, "hamming")
    a, b = _rng(ref_search_range)
    out = _one(cfg, ref, (a, b, max_start_pos, min_end_pos, 2), device)
    return (float(out["pass_score"][0, 0]), int(out["pass_start"][0, 0]), int(out["pass_end"][0, 0]))


def find_best_matching_bc(seq: str, bc_seqs: List[str], bc_lengths_no_N: List[int], config: DemuxConfig, ws,
                          ref_search_range, max_start_pos: int, min_end_pos: int, trim_side: Optional[int],
                          need_traceback: bool = False, *, device: int = 0):
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
    out = _one(cfg, seq, (a, b, max_start_pos, min_end_pos, 1), device)
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


def determine_filename(seq: str, config: DemuxConfig, ws=None, *, device: int = 0):
    """classification.jl:871-938.  Returns ``(filename, keep_start, keep_end)``."""
    with HipClassifier(config, device=device) as hc:
        blob, off = pack_reads([seq])
        out = hc.classify(blob, off)
    bc1, bc2 = int(out["bc1"][0]), int(out["bc2"][0])
    return filename_for(config, bc1, bc2), int(out["keep_start"][0]), int(out["keep_end"][0])


@dataclass
class DemuxStats:
    """Scalar part of classification.jl:736-744 (histograms :745-757 are out of scope)."""

    total_reads: int = 0
    matched_reads: int = 0
    unmatched_reads: int = 0
    ambiguous_reads: int = 0
    sample_counts: Dict[Tuple[int, int], int] = field(default_factory=dict)

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


def merge_stats(stats_list: List[DemuxStats]) -> DemuxStats:
    """reporting.jl:1-9: sum of the per-worker counters (across GPUs this is one RCCL
    all-reduce of the counter vector, see dist.py)."""
    m = DemuxStats()
    for s in stats_list:
        m.total_reads += s.total_reads
        m.matched_reads += s.matched_reads
        m.unmatched_reads += s.unmatched_reads
        m.ambiguous_reads += s.ambiguous_reads
        for k, v in s.sample_counts.items():
            m.sample_counts[k] = m.sample_counts.get(k, 0) + v
    return m


def isinf(x: float) -> bool:
    return math.isinf(x)
