"""biodemux.jl_amd — MI355X (gfx950) drop-in for the classification hot path of BioDemuX.jl.

Only what the path needs lives here:
  csrc/            hand-written HIP kernels + the C-ABI shared library (libbiodemux_hip.so)
  hipabi.py        ctypes binding of include/biodemux_hip.h (counterpart of the Julia ccall shim)
  classification.py / config.py / ranges.py / fileio.py / core.py
                   host-side mirror of the reference's exported API (same names, arguments,
                   defaults, errors) so the parity tests read like the reference's own tests
  dist.py          one-process-per-GPU sharding + the single RCCL all-reduce of the counters
  synth.py         seeded synthetic FASTQ-shaped batches for tests and bench

The directory name contains a dot, so it is imported through the loader ``biodemux_jl_amd.py``
at the repository root: ``import biodemux_jl_amd as bdx``.
"""
from .classification import (DemuxStats, SemiGlobalWorkspace, determine_filename, exact_align, filename_for,
                             find_best_matching_bc, hamming_align, merge_stats, semiglobal_alignment,
                             semiglobal_alignment_N)
from .config import DemuxConfig, build_config
from .core import execute_demultiplexing
from .fileio import preprocess_bc_file, read_fastq, write_fastq
from .hipabi import ABI_SYMBOLS, LIB_PATH, BdxError, HipClassifier, load_library, pack_reads
from .ranges import DynamicRange, parse_dynamic_range, resolve
from .reporting import generate_summary_report

__all__ = [
    "DemuxConfig", "DemuxStats", "DynamicRange", "SemiGlobalWorkspace", "HipClassifier", "BdxError",
    "build_config", "determine_filename", "exact_align", "execute_demultiplexing", "filename_for",
    "find_best_matching_bc", "hamming_align", "load_library", "merge_stats", "pack_reads",
    "generate_summary_report", "parse_dynamic_range", "preprocess_bc_file", "read_fastq", "resolve", "semiglobal_alignment",
    "semiglobal_alignment_N", "write_fastq", "ABI_SYMBOLS", "LIB_PATH",
]
