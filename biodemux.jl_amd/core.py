"""execute_demultiplexing — the drop-in API surface (BioDemuX.jl src/core.jl:360, :500).

Same positional arguments, keyword names, defaults and file contract as the reference; the
per-read classification (worker_task's loop, core.jl:243-267) is ONE C-ABI call per batch on
the MI355X instead of nthreads() Julia workers.  Reader and writer here are plain host code
that keeps the reference's observable behaviour (record framing, naming, ordering, append
mode, trimming of R1 only); they are the "next" rows of SURVEY §8(f), not the hot path.
"""
from __future__ import annotations

import datetime as _dt
import gzip
import os
import re
import sys
from typing import Callable, Dict, List, Optional

import numpy as np

from .classification import DemuxStats, filename_for
from .config import DemuxConfig, build_config
from .fileio import read_fastq
from . import nativeio
from .hipabi import HipClassifier
from .reporting import canonical_duration, generate_summary_report

_PREFIX_RE = re.compile(r"\.fastq(\.gz)?$")

# Reads handed to the device per C-ABI call.  The reference hands 4000-read chunks to each
# worker (chunk_size); a GPU wants >= 10^5 reads per launch (256 CUs x 8 waves x 64 lanes).
DEFAULT_BATCH_READS = 1 << 19  # reads per C-ABI call of the file pipeline (the kernels are at full speed from ~10^5 reads on; smaller batches fill the pipeline sooner)


def _records(io):
    """FASTQ framing of reader_task (core.jl:96-101): while !eof, four readline() calls; a
    truncated last record is padded with empty lines exactly as readline at EOF returns ""."""
    data = io.read()
    if not data:
        return
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # data ended with '\n': no further line exists
    lines = [ln[:-1] if ln.endswith(b"\r") else ln for ln in lines]  # readline strips "\r\n"
    n = len(lines)
    for i in range(0, n, 4):
        rec = lines[i:i + 4]
        while len(rec) < 4:
            rec.append(b"")
        yield rec


class _Writer:
    """writer_task (core.jl:118-224): lazily opened per-file handles in append mode, gzip when
    config.gzip_output or the name ends with .gz, records written as h\\ns\\np\\nq\\n."""

    def __init__(self, output_dir: str, config: DemuxConfig):
        self.dir = output_dir
        self.config = config
        self.handles: Dict[str, object] = {}

    def get(self, filename: str):
        h = self.handles.get(filename)
        if h is None:
            path = os.path.join(self.dir, filename)
            should_gzip = self.config.gzip_output or path.lower().endswith(".gz")  # core.jl:127
            h = gzip.open(path, "ab") if should_gzip else open(path, "ab")
            self.handles[filename] = h
        return h

    def write_entry(self, filename: str, h: bytes, s: bytes, p: bytes, q: bytes):
        self.get(filename).write(b"".join((h, b"\n", s, b"\n", p, b"\n", q, b"\n")))  # core.jl:134-136

    def close(self):
        for h in self.handles.values():
            h.close()
        self.handles.clear()


def _log(msg: str):
    print(f"[{_dt.datetime.now().strftime('%H:%M:%S')}] {msg}", file=sys.stderr)


def _demux(fastq1: str, fastq2: Optional[str], config: DemuxConfig, output_directory: str, prefix1: str,
           prefix2: str, classifier, batch_reads: int, on_batch=None) -> None:
    writer = _Writer(output_directory, config)
    do_trim = config.trim_side is not None or config.trim_side2 is not None  # core.jl:240

    def flush(r1: List[list], r2: Optional[List[list]]):
        if not r1:
            return
        seqs = [rec[1] for rec in r1]
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        blob = np.frombuffer(b"".join(seqs), dtype=np.uint8) if off[-1] else np.zeros(0, dtype=np.uint8)
        out = classifier.classify(blob, off)  # <- the hot path: one C-ABI call per batch
        if on_batch is not None:
            on_batch(out)
        bc1, bc2, ks, ke = out["bc1"], out["bc2"], out["keep_start"], out["keep_end"]
        for i, (h1, s1, p1, q1) in enumerate(r1):
            filename = filename_for(config, int(bc1[i]), int(bc2[i]))
            if do_trim and ks[i] != -1:  # core.jl:250-254, :162-173
                a = max(int(ks[i]), 1)
                b = min(int(ke[i]), len(s1))
                if a <= b:
                    s1 = s1[a - 1:b]
                    q1 = q1[a - 1:b]
                else:
                    s1 = b""
                    q1 = b""
            if config.classify_both and r2 is not None:  # core.jl:175-185
                writer.write_entry(prefix1 + "." + filename, h1, s1, p1, q1)
                writer.write_entry(prefix2 + "." + filename, *r2[i])
            elif r2 is not None:  # core.jl:186-190
                writer.write_entry(prefix2 + "." + filename, *r2[i])
            else:  # core.jl:191-196
                writer.write_entry(prefix1 + "." + filename, h1, s1, p1, q1)

    try:
        with read_fastq(fastq1) as io1:
            if fastq2 is not None:
                with read_fastq(fastq2) as io2:  # lock-step pairs, core.jl:48-75
                    b1: List[list] = []
                    b2: List[list] = []
                    for rec1, rec2 in zip(_records(io1), _records(io2)):
                        b1.append(rec1)
                        b2.append(rec2)
                        if len(b1) >= batch_reads:
                            flush(b1, b2)
                            b1, b2 = [], []
                    flush(b1, b2)
            else:
                b1 = []
                for rec1 in _records(io1):
                    b1.append(rec1)
                    if len(b1) >= batch_reads:
                        flush(b1, None)
                        b1 = []
                flush(b1, None)
    finally:
        writer.close()


def execute_demultiplexing(*args, _classifier_factory: Optional[Callable[[DemuxConfig], object]] = None,
                           _batch_reads: int = DEFAULT_BATCH_READS, _io: str = "auto", device: int = 0,
                           _timings: Optional[dict] = None, **kw):
    """execute_demultiplexing(FASTQ_file, barcode_file, output_directory; kwargs...)      core.jl:500
    execute_demultiplexing(FASTQ_file1, FASTQ_file2, barcode_file, output_directory; ...)  core.jl:360

    Keyword arguments and defaults are the reference's (core.jl:365-391 / :504-528).
    ``_io`` selects the host-side reader/writer: "native" (csrc/bdx_io.cpp, threads), "python" (the
    plain reference implementation below) or "auto" (native when the library was built).
    ``_classifier_factory`` is a test seam: the parity tests on CPU pass the oracle here to
    check this file contract; the product default is the HIP classifier and nothing else.
    ``_timings`` (a dict) receives the busy seconds of the native pipeline's stages (bench.py's end-to-end figure).
    Returns the DemuxStats scalar counters (the reference returns nothing)."""
    if len(args) == 3:
        fastq1, barcode_file, output_directory = args
        fastq2 = None
        paired = False
    elif len(args) == 4:
        fastq1, fastq2, barcode_file, output_directory = args
        paired = True
    else:
        raise TypeError("execute_demultiplexing takes 3 (single-end) or 4 (paired-end) positional arguments")

    defaults = dict(
        barcode_file2=None, gzip_output=None, max_error_rate=0.2, min_delta=0.0, match=0, mismatch=1, indel=1,
        nindel=None, bc_complement=False, bc_rev=False, ref_search_range="1:end", barcode_start_range="1:end",
        barcode_end_range="1:end", ref_search_range2="1:end", barcode_start_range2="1:end",
        barcode_end_range2="1:end", chunk_size=4000, channel_capacity=64, trim_side=None, trim_side2=None,
        summary=False, summary_format="html", matching_algorithm="semiglobal", log=False)
    if paired:
        defaults.update(output_prefix1="", output_prefix2="", classify_both=False)
    else:
        defaults.update(output_prefix="")
    unknown = set(kw) - set(defaults)
    if unknown:
        raise TypeError(f"unsupported keyword argument(s): {sorted(unknown)}")
    o = {**defaults, **kw}

    start_time = _dt.datetime.now()
    if o["log"]:  # core.jl:394-407 / :531-543
        _log("Info: BioDemuX demultiplexing started (MI355X HIP backend).")
        _log(f"  - Input 1: {os.path.basename(fastq1)}")
        if paired:
            _log(f"  - Input 2: {os.path.basename(fastq2)}")
        _log(f"  - Barcode File: {os.path.basename(barcode_file)}")
        _log(f"  - Output Directory: {output_directory}")
        _log(f"  - Max Error Rate: {o['max_error_rate']}")

    if not os.path.isdir(output_directory):  # core.jl:410-412
        os.mkdir(output_directory)

    if paired:  # core.jl:414-419
        prefix1 = o["output_prefix1"] or _PREFIX_RE.sub("", os.path.basename(fastq1))
        prefix2 = o["output_prefix2"] or _PREFIX_RE.sub("", os.path.basename(fastq2))
        fastqs = [fastq1, fastq2]
        classify_both = o["classify_both"]
    else:  # core.jl:550-552
        prefix1 = o["output_prefix"] or _PREFIX_RE.sub("", os.path.basename(fastq1))
        prefix2 = ""
        fastqs = [fastq1]
        classify_both = False

    config = build_config(
        barcode_file, o["barcode_file2"], fastqs, o["gzip_output"], o["bc_complement"], o["bc_rev"], classify_both,
        float(o["max_error_rate"]), float(o["min_delta"]), o["match"], o["mismatch"], o["indel"], o["nindel"],
        o["ref_search_range"], o["barcode_start_range"], o["barcode_end_range"], o["ref_search_range2"],
        o["barcode_start_range2"], o["barcode_end_range2"], o["trim_side"], o["trim_side2"], o["summary"],
        o["summary_format"], o["matching_algorithm"])

    # summary=true: the HIP classifier collects the histograms of classification.jl:827-865 on the device
    # (bdx_get_stats); a test-injected classifier without such tables hands over per-pass outputs instead
    classifier = _classifier_factory(config) if _classifier_factory else HipClassifier(config, device=device)
    if _timings is not None:  # barcode table + device context: before the first batch can move
        _timings["setup_s"] = (_dt.datetime.now() - start_time).total_seconds()
    device_stats = hasattr(classifier, "stats_tables")
    hist = DemuxStats() if (config.summary and not device_stats) else None

    def on_batch(out):
        if hist is not None and "pass_bc" in out:
            hist.add_pass_outputs(out, float(config.min_delta))

    try:
        if _io not in ("auto", "native", "python"):
            raise ValueError("_io must be 'auto', 'native' or 'python'")
        use_native = _io == "native" or (_io == "auto" and nativeio.available())
        if _timings is not None:
            _timings["pre_s"] = (_dt.datetime.now() - start_time).total_seconds()  # everything before the first batch can be read
        if use_native:
            t_call = _dt.datetime.now()
            nativeio.demux_native(fastq1, fastq2, config, output_directory, prefix1, prefix2, classifier, _batch_reads,
                                  on_batch, _timings)
            if _timings is not None:  # (beyond the pipeline's own wall clock: releasing its batch buffers)
                _timings["native_call_s"] = (_dt.datetime.now() - t_call).total_seconds()
        else:
            _demux(fastq1, fastq2, config, output_directory, prefix1, prefix2, classifier, _batch_reads, on_batch)
        counts = np.asarray(classifier.counts)
        tables = classifier.stats_tables() if (config.summary and device_stats) else None
    finally:
        t_close = _dt.datetime.now()
        classifier.close()
        if _timings is not None:  # releasing the device context (staging buffers, tables)
            _timings["close_s"] = (_dt.datetime.now() - t_close).total_seconds()

    duration = _dt.datetime.now() - start_time  # core.jl:484-485
    if _timings is not None:
        _timings["total_s"] = duration.total_seconds()
    if o["log"]:  # core.jl:487-491
        _log(f"Done: Finished in {canonical_duration(duration)}.")
    stats = DemuxStats.from_counts(counts, len(config.bc_seqs), len(config.bc_seqs2) if config.is_dual else 0)
    if tables is not None:
        stats.add_device_tables(tables, config)
    if hist is not None:
        for f in ("bc1", "bc2"):
            for k in ("pos_counts", "len_counts", "score_counts", "per_bc_score_counts", "per_bc_pos_counts",
                      "per_bc_len_counts"):
                setattr(stats, f"{f}_{k}", getattr(hist, f"{f}_{k}"))
    if config.summary:  # core.jl:493-497 / :626-630 (merge_stats over the workers: one device context here)
        generate_summary_report(stats, config, output_directory, fastq1, barcode_file, fastq2, o["barcode_file2"],
                                bc_complement=o["bc_complement"], bc_rev=o["bc_rev"], trim_side=o["trim_side"],
                                trim_side2=o["trim_side2"], n_threads=1, duration=duration)
    return stats
