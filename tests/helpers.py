"""Shared test helpers: paths, KAT dispatch, golden-file comparator, reference-test scenarios.

The scenarios are the reference's own integration tests (test/integration/*.jl) restated as
data + expectations; each takes ``run`` = execute_demultiplexing bound to a backend, so the
very same scenario checks (a) the oracle + host file contract on CPU and (b) the HIP path on
the GPU.
"""
from __future__ import annotations

import gzip
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(GOLDEN, "reference")

import biodemux_jl_amd as bdx  # noqa: E402
import bdx_oracle as orc  # noqa: E402  (test infrastructure)

INF = float("inf")


def kat_vectors():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def _dec(x):
    if x == "Inf":
        return INF
    if isinstance(x, list):
        return [_dec(v) for v in x]
    return x


def make_config(overrides: dict) -> "bdx.DemuxConfig":
    return bdx.DemuxConfig(**overrides)


def oracle_factory(cfg):
    return orc.OracleClassifier(cfg, nthreads=1, want_pass=bool(cfg.summary))


def run_kat(vec, api: str, filter: str = "auto"):
    """Evaluate one KAT with ``api`` in {"oracle", "hip"}; returns (got, expect).  ``filter`` (hip only)
    selects the kernel path in front of the exact evaluation: every value must give the same answer."""
    fn, args, expect = vec["fn"], [_dec(a) for a in vec["args"]], _dec(vec["expect"])
    hip = api == "hip"
    fk = dict(filter=filter) if hip else {}
    if fn == "parse_dynamic_range":
        dr = bdx.parse_dynamic_range(args[0])
        return [dr.start_offset, dr.start_from_end, dr.end_offset, dr.end_from_end], expect
    if fn == "resolve":
        return list(bdx.resolve(bdx.parse_dynamic_range(args[0]), args[1])), expect
    if fn == "semiglobal_alignment":
        q, r, me, ma, mi, ind, rng, ms, mn = args[:9]
        ts = args[9] if len(args) > 9 else None
        tb = args[10] if len(args) > 10 else False
        if hip:
            got = bdx.semiglobal_alignment(None, q, r, me, ma, mi, ind, tuple(rng), ms, mn, ts, tb, **fk)
        else:
            got = orc.semiglobal_alignment(q, r, me, ma, mi, ind, tuple(rng), ms, mn, ts, tb)
    elif fn == "semiglobal_alignment_N":
        q, r, me, ma, mi, ind, nind, rng, ms, mn, nn = args[:11]
        if hip:
            got = bdx.semiglobal_alignment_N(None, q, r, me, ma, mi, ind, nind, tuple(rng), ms, mn, nn, **fk)
        else:
            got = orc.semiglobal_alignment_N(q, r, me, ma, mi, ind, nind, tuple(rng), ms, mn, nn)
    elif fn == "hamming_align":
        q, r, me, rng, ms, mn, ts = args
        got = (bdx.hamming_align if hip else orc.hamming_align)(q, r, me, tuple(rng), ms, mn, ts, **fk)
    elif fn == "exact_align":
        q, r, rng, ms, mn, ts = args
        got = (bdx.exact_align if hip else orc.exact_align)(q, r, tuple(rng), ms, mn, ts, **fk)
    elif fn == "determine_filename":
        read, over = args
        cfg = make_config(over)
        if hip:
            got = bdx.determine_filename(read, cfg, **fk)
        else:
            v = orc.determine_filename(read, cfg)
            got = (bdx.filename_for(cfg, v.bc1, v.bc2), v.keep_start, v.keep_end)
    else:
        raise KeyError(fn)
    if isinstance(got, tuple):
        got = list(got)
    return got, expect


# ---- golden comparator: reference test/common.jl:4-21 ----
def _read_maybe_gz(path: str) -> bytes:
    if path.lower().endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return f.read()
    with open(path, "rb") as f:
        return f.read()


def check_output_files(output_dir: str, ideal_dir: str):
    names = sorted(os.listdir(ideal_dir))
    assert names, ideal_dir
    for name in names:
        out = os.path.join(output_dir, name)
        assert os.path.isfile(out), f"missing output file {name}"
        assert _read_maybe_gz(out) == _read_maybe_gz(os.path.join(ideal_dir, name)), f"content differs: {name}"
    return len(names)


def count_lines(path: str) -> int:
    with open(path, "rb") as f:
        return sum(1 for _ in f)


# ---- scenarios (reference integration tests) ----
def scenario_demo1_R1(run, tmp):
    """integration/single_barcode.jl:2-11 — 24 plain FASTQ, demo1.tsv, all defaults."""
    src = os.path.join(REF, "FASTQ_files", "demo1_R1")
    for f in sorted(os.listdir(src)):
        run(os.path.join(src, f), os.path.join(REF, "reference_files", "demo1.tsv"), tmp)
    return check_output_files(tmp, os.path.join(REF, "results", "demo1_R1"))


def scenario_demo1_R2(run, tmp):
    """integration/single_barcode.jl:13-23 — paired, defaults: only R2 written."""
    s1 = os.path.join(REF, "FASTQ_files", "demo1_R1")
    s2 = os.path.join(REF, "FASTQ_files", "demo1_R2")
    for f1, f2 in zip(sorted(os.listdir(s1)), sorted(os.listdir(s2))):
        run(os.path.join(s1, f1), os.path.join(s2, f2), os.path.join(REF, "reference_files", "demo1.tsv"), tmp)
    return check_output_files(tmp, os.path.join(REF, "results", "demo1_R2"))


def scenario_demo2(run, tmp):
    """integration/single_barcode.jl:25-45 — paired gz, weighted indel, min_delta, revcomp."""
    import re

    s1 = os.path.join(REF, "FASTQ_files", "demo2_R1")
    s2 = os.path.join(REF, "FASTQ_files", "demo2_R2")
    for f1, f2 in zip(sorted(os.listdir(s1)), sorted(os.listdir(s2))):
        n1 = re.sub(r"\.fastq\.gz$", "", f1)
        n2 = re.sub(r"\.fastq\.gz$", "", f2)
        run(os.path.join(s1, f1), os.path.join(s2, f2), os.path.join(REF, "reference_files", "demo2.csv"), tmp,
            max_error_rate=0.25, min_delta=0.15, mismatch=1, indel=2, classify_both=True, bc_complement=True,
            bc_rev=True, output_prefix1=f"test_prefix1.{n1}", output_prefix2=f"test_prefix2.{n2}",
            gzip_output=False)
    return check_output_files(tmp, os.path.join(REF, "results", "demo2"))


def scenario_demo1_modes(run, tmp, algorithm):
    """every demo1 R1 read carries its own core once, unmutated, so :exact and
    :hamming must reproduce results/demo1_R1 too (BASELINE config 1, C1)."""
    src = os.path.join(REF, "FASTQ_files", "demo1_R1")
    for f in sorted(os.listdir(src)):
        run(os.path.join(src, f), os.path.join(REF, "reference_files", "demo1.tsv"), tmp,
            matching_algorithm=algorithm)
    return check_output_files(tmp, os.path.join(REF, "results", "demo1_R1"))


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)


def scenario_n_and_ranges(run, tmp):
    """integration/single_barcode.jl:47-97 — FASTA barcodes with N, nindel=1, rate 0.6."""
    bc = os.path.join(tmp, "barcodes.fasta")
    _write(bc, ">BC1\nANNC\n>BC2\nTTTT\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1\nATTC\n+\nIIII\n@read2\nTTTT\n+\nIIII\n@read3\nATTG\n+\nIIII\n@read4\nGGGG\n+\nIIII\n")
    out = os.path.join(tmp, "output")
    run(fq, bc, out, max_error_rate=0.6, nindel=1, ref_search_range="1:end", barcode_start_range="1:end",
        barcode_end_range="1:end")
    assert count_lines(os.path.join(out, "reads.BC1.fastq")) == 8
    assert count_lines(os.path.join(out, "reads.BC2.fastq")) == 4
    assert count_lines(os.path.join(out, "reads.unknown.fastq")) == 4


def scenario_range_restrictions(run, tmp):
    """integration/single_barcode.jl:99-173."""
    bc = os.path.join(tmp, "barcodes.fasta")
    _write(bc, ">BC1\nAAAA\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1_start\nAAAATTTT\n+\nIIIIIIII\n@read2_end\nTTTTAAAA\n+\nIIIIIIII\n"
               "@read3_mid\nTTAAAATT\n+\nIIIIIIII\n")
    for k, kw in enumerate([dict(ref_search_range="1:4"), dict(ref_search_range="5:8"),
                            dict(ref_search_range="1:end", barcode_start_range="1:1")]):
        out = os.path.join(tmp, f"output_{k + 1}")
        args = dict(max_error_rate=0.0, nindel=1, ref_search_range="1:end", barcode_start_range="1:end",
                    barcode_end_range="1:end")
        args.update(kw)
        run(fq, bc, out, **args)
        assert count_lines(os.path.join(out, "reads.BC1.fastq")) == 4
        assert count_lines(os.path.join(out, "reads.unknown.fastq")) == 8


def scenario_dual(run, tmp):
    """integration/dual_barcode.jl:5-96."""
    b1 = os.path.join(tmp, "bc1.tsv")
    _write(b1, "Full_seq\tID\tFull_annotation\nAAAA\tID1_A\tBBBB\nCCCC\tID1_C\tBBBB\n")
    b2 = os.path.join(tmp, "bc2.tsv")
    _write(b2, "Full_seq\tID\tFull_annotation\nTTTT\tID2_T\tBBBB\nGGGG\tID2_G\tBBBB\n")
    fq = os.path.join(tmp, "test.fastq")
    q = "IIIIIIIIIIIIIIII"
    _write(fq, f"@read1\nAAAATATATTTTACGT\n+\n{q}\n@read2\nCCCCTATAGGGGACGT\n+\n{q}\n"
               f"@read3\nAAAATATAGGGGACGT\n+\n{q}\n@read4\nAAAATATAAAAAACGT\n+\n{q}\n")
    out = os.path.join(tmp, "output")
    os.mkdir(out)
    run(fq, b1, out, barcode_file2=b2, ref_search_range="1:4", ref_search_range2="9:12", max_error_rate=0.0,
        chunk_size=100)
    for name in ("test.ID1_A.ID2_T.fastq", "test.ID1_C.ID2_G.fastq", "test.ID1_A.ID2_G.fastq", "test.unknown.fastq"):
        assert os.path.isfile(os.path.join(out, name)), name
    assert "@read1" in open(os.path.join(out, "test.ID1_A.ID2_T.fastq")).read()
    assert "@read4" in open(os.path.join(out, "test.unknown.fastq")).read()


def scenario_dual_trim(run, tmp):
    """integration/dual_barcode.jl:98-162."""
    b1 = os.path.join(tmp, "bc1.tsv")
    _write(b1, "Full_seq\tID\tFull_annotation\nAAAA\tID1_A\tBBBB\n")
    b2 = os.path.join(tmp, "bc2.tsv")
    _write(b2, "Full_seq\tID\tFull_annotation\nTTTT\tID2_T\tBBBB\n")
    fq = os.path.join(tmp, "test_trim.fastq")
    _write(fq, "@read1\nAAAATATATTTTACGT\n+\nIIIIIIIIIIIIIIII\n")
    out = os.path.join(tmp, "output_trim")
    os.mkdir(out)
    run(fq, b1, out, barcode_file2=b2, ref_search_range="1:4", ref_search_range2="9:12", max_error_rate=0.0,
        chunk_size=100, trim_side=5, trim_side2=3)
    lines = open(os.path.join(out, "test_trim.ID1_A.ID2_T.fastq")).read().strip().split("\n")
    assert lines[1] == "TATA"
    assert len(lines[3]) == 4


def scenario_hamming(run, tmp):
    """integration/hamming_demux.jl:3-44 and :47-76."""
    bc = os.path.join(tmp, "barcodes.csv")
    _write(bc, "ID,Full_seq,Full_annotation\nBC1,ACGTAC,BBBBBB\nBC2,CCCCCC,BBBBBB\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1\nACGTAC\n+\nIIIIII\n@read2\nCCCCCC\n+\nIIIIII\n@read3\nACATAC\n+\nIIIIII\n"
               "@read4\nACGTAG\n+\nIIIIII\n@read_indel\nACGGTAC\n+\nIIIIIII\n")
    run(fq, bc, tmp, matching_algorithm="hamming", max_error_rate=0.2)
    assert count_lines(os.path.join(tmp, "reads.BC1.fastq")) == 12
    assert count_lines(os.path.join(tmp, "reads.BC2.fastq")) == 4
    assert count_lines(os.path.join(tmp, "reads.unknown.fastq")) == 4
    # paired, classify_both
    bc2 = os.path.join(tmp, "barcodes2.csv")
    _write(bc2, "ID,Full_seq,Full_annotation\nBC1,AAAA,BBBB\n")
    f1, f2 = os.path.join(tmp, "R1.fastq"), os.path.join(tmp, "R2.fastq")
    _write(f1, "@seq1\nAAAA\n+\nIIII\n")
    _write(f2, "@seq1\nGGGG\n+\nIIII\n")
    run(f1, f2, bc2, tmp, matching_algorithm="hamming", max_error_rate=0.0, classify_both=True)
    assert os.path.isfile(os.path.join(tmp, "R1.BC1.fastq"))
    assert os.path.isfile(os.path.join(tmp, "R2.BC1.fastq"))


def scenario_exact(run, tmp):
    """integration/exact_demux.jl:2-43."""
    bc = os.path.join(tmp, "barcodes.csv")
    _write(bc, "ID,Full_seq,Full_annotation\nBC1,ACGTAC,BBBBBB\nBC2,CCCCCC,BBBBBB\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1\nACGTAC\n+\nIIIIII\n@read2\nCCCCCC\n+\nIIIIII\n@read3\nACATAC\n+\nIIIIII\n"
               "@read4\nACGTAG\n+\nIIIIII\n@read_indel\nACGGTAC\n+\nIIIIIII\n")
    run(fq, bc, tmp, matching_algorithm="exact", max_error_rate=0.0)
    assert count_lines(os.path.join(tmp, "reads.BC1.fastq")) == 4
    assert count_lines(os.path.join(tmp, "reads.BC2.fastq")) == 4
    assert count_lines(os.path.join(tmp, "reads.unknown.fastq")) == 12


def scenario_summary_counts(run, tmp):
    """integration/summary_mode.jl:7-53 (counter part only: "Total Reads: 4", "Matched Reads: 2";
    4-bp barcodes at rate 0.2 allow floor(0.8) = 0 errors) and :66-80 (dual, inline)."""
    bc = os.path.join(tmp, "barcodes.fasta")
    _write(bc, ">BC1\nAAAA\n>BC2\nTTTT\n")
    bc2 = os.path.join(tmp, "barcodes2.fasta")
    _write(bc2, ">BC2_1\nCCCC\n>BC2_2\nGGGG\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1\nAAAA\n+\nIIII\n@read2\nTTTT\n+\nIIII\n@read3\nGGGG\n+\nIIII\n@read4\nAAAT\n+\nIIII\n")
    out = os.path.join(tmp, "output")
    os.makedirs(out)
    stats = run(fq, bc, out, max_error_rate=0.2, summary=True, summary_format="txt")
    assert stats.total_reads == 4
    assert stats.matched_reads == 2
    assert stats.unmatched_reads == 2
    assert stats.ambiguous_reads == 0
    assert stats.sample_counts == {(1, 0): 1, (2, 0): 1}
    f1, f2 = os.path.join(tmp, "reads_R1.fastq"), os.path.join(tmp, "reads_R2.fastq")
    _write(f1, "@read1\nAAAACCCC\n+\nIIIIIIII\n@read2\nTTTTGGGG\n+\nIIIIIIII\n")
    _write(f2, "@read1\nNNNN\n+\nIIII\n@read2\nNNNN\n+\nIIII\n")
    # summary_mode.jl:50-57: the text report
    txt = os.path.join(out, "summary.txt")
    content = open(txt).read()
    for needle in ("Total Reads: 4", "Matched Reads: 2", "Run Information:", f"Barcode File: {bc}"):
        assert needle in content, needle
    assert "Barcode File 2:" not in content
    assert "BC1\t1\t25.0%" in content and "Matched Reads: 2 (50.0%)" in content  # reporting.jl:111, :127 (Julia float printing)
    # :59-67: a second run appends after a separator
    run(fq, bc, out, max_error_rate=0.2, summary=True, summary_format="txt")
    content = open(txt).read()
    assert "==================================================" in content
    assert content.split("\n").count("Run Information:") == 2
    # :69-82: dual (inline), JSON
    f1, f2 = os.path.join(tmp, "reads_R1.fastq"), os.path.join(tmp, "reads_R2.fastq")
    _write(f1, "@read1\nAAAACCCC\n+\nIIIIIIII\n@read2\nTTTTGGGG\n+\nIIIIIIII\n")
    _write(f2, "@read1\nNNNN\n+\nIIII\n@read2\nNNNN\n+\nIIII\n")
    stats = run(f1, f2, bc, out, barcode_file2=bc2, max_error_rate=0.2, summary=True, summary_format="json")
    assert stats.total_reads == 2 and stats.matched_reads == 2
    assert stats.sample_counts == {(1, 1): 1, (2, 2): 1}
    js = os.path.join(out, "summary.json")
    content = open(js).read()
    for needle in ('"total_reads": 2', '"matched_reads": 2', f'"barcode_file2": "{bc2}"', '"(1, 1)": 1', '"bc2_pos_counts": {"5": 2}'):
        assert needle in content, needle
    assert len(json.loads(content)) == 1  # the file is valid JSON: a list of run objects
    # :84-95: a second JSON run extends the list
    run(f1, f2, bc, out, barcode_file2=bc2, max_error_rate=0.2, summary=True, summary_format="json")
    content = open(js).read().strip()
    assert content.startswith("[") and content.endswith("]")
    assert sum('"run_info"' in ln for ln in content.split("\n")) == 2
    assert [r["total_reads"] for r in json.loads(content)] == [2, 2]
    # :97-119: HTML, twice
    run(f1, f2, bc, out, barcode_file2=bc2, max_error_rate=0.2, summary=True, summary_format="html")
    content = open(os.path.join(out, "summary.html")).read()
    assert "Total Reads" in content and "Barcode File 2:" in content and bc2 in content
    run(f1, f2, bc, out, barcode_file2=bc2, max_error_rate=0.2, summary=True, summary_format="html")
    content = open(os.path.join(out, "summary.html")).read()
    assert sum("Run Information" in ln for ln in content.split("\n")) == 2 and content.count("</body>") == 1
    # :121-141: stdout
    import contextlib
    import io as _io
    buf = _io.StringIO()
    with contextlib.redirect_stdout(buf):
        run(f1, f2, bc, out, barcode_file2=bc2, max_error_rate=0.2, summary=True, summary_format="stdout")
    output = buf.getvalue()
    assert "BioDemuX Summary Report" in output and f"Barcode File 2: {bc2}" in output and "Total Reads: 2" in output


def scenario_summary_distribution_reads(run, tmp):
    """integration/summary_distributions.jl:7-28 — the five reads (incl. the deletion read
    "AAA") must all classify to BC1 at rate 0.3 with traceback forced on by summary=true."""
    bc = os.path.join(tmp, "barcodes.fasta")
    _write(bc, ">BC1\nAAAA\n")
    fq = os.path.join(tmp, "reads.fastq")
    _write(fq, "@read1\nAAAA\n+\nIIII\n@read2\nNAAAA\n+\nIIIII\n@read3\nNNAAAA\n+\nIIIIII\n"
               "@read4\nAAAT\n+\nIIII\n@read5\nAAA\n+\nIII\n")
    stats = run(fq, bc, os.path.join(tmp, "output_dist"), max_error_rate=0.3, summary=True, summary_format="json")
    assert stats.total_reads == 5 and stats.matched_reads == 5
    # summary_distributions.jl:40-46: start positions 1, 2, 3 and lengths 3, 4 are present
    assert set(stats.bc1_pos_counts) == {1, 2, 3} and sum(stats.bc1_pos_counts.values()) == 5
    assert set(stats.bc1_len_counts) == {3, 4} and sum(stats.bc1_len_counts.values()) == 5
    assert stats.bc1_pos_counts[2] == 1 and stats.bc1_pos_counts[3] == 1
    assert set(stats.bc1_score_counts) == {0.0, 0.25} and stats.bc1_per_bc_pos_counts[1] == stats.bc1_pos_counts
    assert stats.bc2_pos_counts == {}
    # summary_distributions.jl:30-52: the JSON report carries the distributions
    content = open(os.path.join(tmp, "output_dist", "summary.json")).read()
    for needle in ('"bc1_pos_counts": {', '"bc1_len_counts": {', '"bc1_score_counts": {', '"1":', '"2":', '"3":', '"4":',
                   '"bc1_per_bc_pos_counts": {', '"bc1_per_bc_len_counts": {', '"bc1_per_bc_score_counts": {'):
        assert needle in content, needle
    run_obj = json.loads(content)[0]
    assert run_obj["bc1_pos_counts"] == {"1": 3, "2": 1, "3": 1} and run_obj["bc1_len_counts"] == {"3": 2, "4": 3}
    assert run_obj["bc1_score_counts"] == {"0.0": 3, "0.25": 2} and run_obj["bc1_per_bc_len_counts"] == {"1": {"3": 2, "4": 3}}
    # ("AAAT" costs 1 either as a substitution or as AAA + one deleted base; the deletion wins the origin tie
    # (classification.jl:310-321), so its length is 3 — the .jl comment "Len 4" is not what the code records)


SCENARIOS_SMALL = [scenario_n_and_ranges, scenario_range_restrictions, scenario_dual, scenario_dual_trim,
                   scenario_hamming, scenario_exact, scenario_summary_counts,
                   scenario_summary_distribution_reads]
