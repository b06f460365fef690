"""DemuxConfig and build_config — host-side mirror of the reference's configuration.

Mirrors BioDemuX.jl src/classification.jl:16-58 (DemuxConfig, same field names and defaults)
and src/core.jl:281-358 (build_config, same validation and gzip default).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

from .fileio import preprocess_bc_file
from .ranges import DynamicRange, parse_dynamic_range


def _full() -> DynamicRange:
    return parse_dynamic_range("1:end")


def _sym(s) -> str:
    """Julia Symbols (:semiglobal) are spelled as strings here; a leading ':' is accepted."""
    return str(s).lstrip(":")


@dataclass
class DemuxConfig:
    """classification.jl:16-58."""

    bc_seqs: List[str]
    bc_lengths_no_N: List[int]
    ids: List[str]
    max_error_rate: float = 0.2
    min_delta: float = 0.0
    match: int = 0
    mismatch: int = 1
    indel: int = 1
    nindel: Optional[int] = None
    classify_both: bool = False
    gzip_output: bool = False
    ref_search_range: DynamicRange = field(default_factory=_full)
    barcode_start_range: DynamicRange = field(default_factory=_full)
    barcode_end_range: DynamicRange = field(default_factory=_full)
    is_dual: bool = False
    ref_search_range2: DynamicRange = field(default_factory=_full)
    barcode_start_range2: DynamicRange = field(default_factory=_full)
    barcode_end_range2: DynamicRange = field(default_factory=_full)
    bc_seqs2: List[str] = field(default_factory=list)
    bc_lengths_no_N2: List[int] = field(default_factory=list)
    ids2: List[str] = field(default_factory=list)
    trim_side: Optional[int] = None
    trim_side2: Optional[int] = None
    summary: bool = False
    summary_format: str = "txt"
    matching_algorithm: str = "semiglobal"

    def __post_init__(self):
        self.matching_algorithm = _sym(self.matching_algorithm)
        self.summary_format = _sym(self.summary_format)


def build_config(
    barcode_file: str,
    barcode_file2: Optional[str],
    fastqs: List[str],
    gzip_output: Optional[bool],
    bc_complement: bool,
    bc_rev: bool,
    classify_both: bool,
    max_error_rate: float,
    min_delta: float,
    match: int,
    mismatch: int,
    indel: int,
    nindel: Optional[int],
    ref_search_range: str,
    barcode_start_range: str,
    barcode_end_range: str,
    ref_search_range2: str,
    barcode_start_range2: str,
    barcode_end_range2: str,
    trim_side: Optional[int],
    trim_side2: Optional[int],
    summary: bool,
    summary_format: str,
    matching_algorithm: str,
) -> DemuxConfig:
    """core.jl:281-358."""
    if trim_side is not None and trim_side != 3 and trim_side != 5:  # core.jl:308-310
        raise ValueError(f"trim_side must be 3 or 5, got {trim_side}")
    if trim_side2 is not None and trim_side2 != 3 and trim_side2 != 5:  # core.jl:311-313
        raise ValueError(f"trim_side2 must be 3 or 5, got {trim_side2}")

    # core.jl:316
    final_gzip_output = any(f.endswith(".gz") for f in fastqs) if gzip_output is None else gzip_output

    bc_seqs, bc_lengths_no_N, ids = preprocess_bc_file(barcode_file, bc_complement, bc_rev)  # core.jl:319

    is_dual = barcode_file2 is not None
    bc_seqs2, bc_lengths_no_N2, ids2 = [], [], []
    if is_dual:
        bc_seqs2, bc_lengths_no_N2, ids2 = preprocess_bc_file(barcode_file2, bc_complement, bc_rev)

    return DemuxConfig(
        max_error_rate=max_error_rate,
        min_delta=min_delta,
        match=match,
        mismatch=mismatch,
        indel=indel,
        nindel=nindel,
        classify_both=classify_both,
        gzip_output=final_gzip_output,
        ref_search_range=parse_dynamic_range(ref_search_range),
        barcode_start_range=parse_dynamic_range(barcode_start_range),
        barcode_end_range=parse_dynamic_range(barcode_end_range),
        bc_seqs=bc_seqs,
        bc_lengths_no_N=bc_lengths_no_N,
        ids=ids,
        is_dual=is_dual,
        ref_search_range2=parse_dynamic_range(ref_search_range2),
        barcode_start_range2=parse_dynamic_range(barcode_start_range2),
        barcode_end_range2=parse_dynamic_range(barcode_end_range2),
        bc_seqs2=bc_seqs2,
        bc_lengths_no_N2=bc_lengths_no_N2,
        ids2=ids2,
        trim_side=trim_side,
        trim_side2=trim_side2,
        summary=summary,
        summary_format=summary_format,
        matching_algorithm=matching_algorithm,
    )
