"""Command line front end — the ArgParse surface of BioDemuX.jl src/cli.jl.

Same positionals, option names, short flags, defaults and modes (file / directory, single / paired) as
`parse_commandline` (cli.jl:3-117) and `julia_main` (cli.jl:119-329); every run ends in the same
execute_demultiplexing call as the reference's, i.e. in the HIP hot path.

    python biodemux_jl_amd.py reads.fastq barcodes.csv out/ -e 0.1 --trim-side 3
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import List, Optional

_FASTQ_ENDINGS = (".fastq", ".fq", ".fastq.gz", ".fq.gz")  # cli.jl:147 (case-sensitive, as there)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="biodemux_jl_amd", description="BioDemuX demultiplexing on MI355X (HIP backend)")
    p.add_argument("fastq1", help="Path to the first FASTQ file (Read 1) OR directory containing FASTQ files")
    p.add_argument("barcode_file", help="Path to the barcode file (CSV/TSV)")
    p.add_argument("output_directory", help="Directory to save demultiplexed files")
    p.add_argument("--fastq2", default=None, help="Path to the second FASTQ file (Read 2) OR directory (paired with fastq1)")
    p.add_argument("--barcode-file2", "-B", default=None, help="Path to the second barcode file for dual indexing")
    p.add_argument("--output-prefix1", "-p", default="", help="Output prefix for Read 1")
    p.add_argument("--output-prefix2", "-P", default="", help="Output prefix for Read 2")
    p.add_argument("--gzip-output", "-z", action="store_true",
                   help="Compress output with GZIP. If not set, inferred from input filenames.")
    p.add_argument("--no-gzip-output", action="store_true", help="Force disable GZIP output")
    p.add_argument("--max-error-rate", "-e", type=float, default=0.2, help="Maximum error rate allowed")
    p.add_argument("--min-delta", "-d", type=float, default=0.0, help="Minimum delta between best and second best match")
    p.add_argument("--match", "-m", type=int, default=0, help="Match score")
    p.add_argument("--mismatch", "-M", type=int, default=1, help="Mismatch penalty")
    p.add_argument("--indel", "-i", type=int, default=1, help="Indel penalty")
    p.add_argument("--nindel", "-I", type=int, default=None, help="N-indel penalty")
    p.add_argument("--classify-both", "-c", action="store_true", help="Classify both reads in paired-end mode")
    p.add_argument("--bc-complement", "-C", action="store_true", help="Use complement of barcode sequences")
    p.add_argument("--bc-rev", "-r", action="store_true", help="Use reverse of barcode sequences")
    p.add_argument("--ref-search-range", default="1:end", help="Range to search in reference (e.g., '1:20')")
    p.add_argument("--barcode-start-range", default="1:end", help="Range for barcode start (e.g., '1:5')")
    p.add_argument("--barcode-end-range", default="1:end", help="Range for barcode end")
    p.add_argument("--ref-search-range2", default="1:end", help="Range to search in reference for barcode 2")
    p.add_argument("--barcode-start-range2", default="1:end", help="Range for barcode 2 start")
    p.add_argument("--barcode-end-range2", default="1:end", help="Range for barcode 2 end")
    p.add_argument("--chunk-size", type=int, default=4000, help="Chunk size for processing")
    p.add_argument("--channel-capacity", type=int, default=64, help="Channel capacity")
    p.add_argument("--trim-side", type=int, default=None, help="Trim side for Read 1 (3 or 5)")
    p.add_argument("--trim-side2", type=int, default=None, help="Trim side for Read 2 (3 or 5)")
    p.add_argument("--summary", action="store_true", help="Generate summary report")
    p.add_argument("--summary-format", default="html", help="Summary format (html, csv, etc.)")
    p.add_argument("--matching-algorithm", default="semiglobal", help="Matching algorithm (semiglobal, hamming, exact)")
    p.add_argument("--log", "-l", action="store_true", help="Enable logging to stderr")
    p.add_argument("--device", type=int, default=0, help="HIP device index (this backend only)")
    return p


def _fastq_files(directory: str) -> List[str]:
    return sorted(os.path.join(directory, f) for f in os.listdir(directory) if f.endswith(_FASTQ_ENDINGS))


def main(argv: Optional[List[str]] = None, _execute=None) -> int:
    """julia_main (cli.jl:119): returns the process exit code.  ``_execute`` is a test seam."""
    from .core import execute_demultiplexing

    run = _execute or execute_demultiplexing
    try:
        a = build_parser().parse_args(argv)
    except SystemExit as e:  # argparse has already printed the message
        return int(e.code or 0)
    try:
        gzip_val = True if a.gzip_output else (False if a.no_gzip_output else None)  # cli.jl:124-129 (3-state)
        common = dict(
            barcode_file2=a.barcode_file2, gzip_output=gzip_val, max_error_rate=a.max_error_rate, min_delta=a.min_delta,
            match=a.match, mismatch=a.mismatch, indel=a.indel, nindel=a.nindel, bc_complement=a.bc_complement,
            bc_rev=a.bc_rev, ref_search_range=a.ref_search_range, barcode_start_range=a.barcode_start_range,
            barcode_end_range=a.barcode_end_range, ref_search_range2=a.ref_search_range2,
            barcode_start_range2=a.barcode_start_range2, barcode_end_range2=a.barcode_end_range2,
            chunk_size=a.chunk_size, channel_capacity=a.channel_capacity, trim_side=a.trim_side, trim_side2=a.trim_side2,
            summary=a.summary, summary_format=a.summary_format, matching_algorithm=a.matching_algorithm, log=a.log)
        if _execute is None:
            common["device"] = a.device

        def paired(f1, f2):
            run(f1, f2, a.barcode_file, a.output_directory, output_prefix1=a.output_prefix1,
                output_prefix2=a.output_prefix2, classify_both=a.classify_both, **common)

        def single(f1):  # prefix1 is the prefix of a single-end run (cli.jl:296)
            run(f1, a.barcode_file, a.output_directory, output_prefix=a.output_prefix1, **common)

        if os.path.isdir(a.fastq1):  # directory mode, cli.jl:145-247
            files1 = _fastq_files(a.fastq1)
            if not files1:
                print(f"Error: No FASTQ files found in directory: {a.fastq1}", file=sys.stderr)
                return 1
            if a.fastq2 is not None:
                if not os.path.isdir(a.fastq2):
                    print("Error: fastq1 is a directory but fastq2 is a file. Both must be directories or both must be files.",
                          file=sys.stderr)
                    return 1
                files2 = _fastq_files(a.fastq2)
                if len(files1) != len(files2):
                    print("Error: File count mismatch between input directories.", file=sys.stderr)
                    print(f"  {a.fastq1}: {len(files1)} files", file=sys.stderr)
                    print(f"  {a.fastq2}: {len(files2)} files", file=sys.stderr)
                    return 1
                for f1, f2 in zip(files1, files2):
                    if a.log:
                        print(f"Processing pair: {os.path.basename(f1)} and {os.path.basename(f2)}", file=sys.stderr)
                    paired(f1, f2)
            else:
                for f1 in files1:
                    if a.log:
                        print(f"Processing file: {os.path.basename(f1)}", file=sys.stderr)
                    single(f1)
        elif a.fastq2 is not None:  # file mode, cli.jl:249-325
            if os.path.isdir(a.fastq2):
                print("Error: fastq1 is a file but fastq2 is a directory. Both must be files.", file=sys.stderr)
                return 1
            paired(a.fastq1, a.fastq2)
        else:
            single(a.fastq1)
        return 0
    except Exception as e:  # cli.jl:326-329: report and exit 1
        import traceback

        traceback.print_exc(file=sys.stderr)
        print(f"{type(e).__name__}: {e}", file=sys.stderr)
        return 1
