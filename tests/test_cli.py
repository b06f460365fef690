"""CLI front end (biodemux.jl_amd/cli.py) against the reference's ArgParse surface (src/cli.jl): option
names, defaults, the 3-state gzip flag, file / directory modes and their error paths.  CPU only: the
runs go through execute_demultiplexing with the oracle as classifier (test seam)."""
import os

import pytest

import helpers as H
from biodemux_jl_amd import cli

FQ = os.path.join(H.REF, "FASTQ_files")


def _run_oracle(*a, **kw):
    return H.bdx.execute_demultiplexing(*a, _classifier_factory=H.oracle_factory, **kw)


def test_defaults_and_flags_match_the_reference():
    p = cli.build_parser()
    a = p.parse_args(["r1.fastq", "bc.csv", "out"])
    ref_defaults = dict(fastq2=None, barcode_file2=None, output_prefix1="", output_prefix2="", gzip_output=False,
                        no_gzip_output=False, max_error_rate=0.2, min_delta=0.0, match=0, mismatch=1, indel=1, nindel=None,
                        classify_both=False, bc_complement=False, bc_rev=False, ref_search_range="1:end",
                        barcode_start_range="1:end", barcode_end_range="1:end", ref_search_range2="1:end",
                        barcode_start_range2="1:end", barcode_end_range2="1:end", chunk_size=4000, channel_capacity=64,
                        trim_side=None, trim_side2=None, summary=False, summary_format="html",
                        matching_algorithm="semiglobal", log=False)  # cli.jl:6-112
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    b = p.parse_args("r1 bc out -B b2.csv -p P1 -P P2 -z -e 0.1 -d 0.05 -m 0 -M 2 -i 3 -I 1 -c -C -r -l".split())
    assert (b.barcode_file2, b.output_prefix1, b.output_prefix2, b.gzip_output) == ("b2.csv", "P1", "P2", True)
    assert (b.max_error_rate, b.min_delta, b.match, b.mismatch, b.indel, b.nindel) == (0.1, 0.05, 0, 2, 3, 1)
    assert b.classify_both and b.bc_complement and b.bc_rev and b.log


def test_calls_reach_execute_demultiplexing_like_julia_main(tmp_path):
    calls = []

    def spy(*args, **kw):
        calls.append((args, kw))

    assert cli.main(["a.fastq", "bc.csv", "out", "-p", "X", "--no-gzip-output", "--trim-side", "3"], _execute=spy) == 0
    (args, kw), = calls
    assert args == ("a.fastq", "bc.csv", "out") and kw["output_prefix"] == "X" and kw["gzip_output"] is False
    assert kw["trim_side"] == 3 and "classify_both" not in kw and "output_prefix1" not in kw  # single end: cli.jl:290-325
    calls.clear()
    assert cli.main(["a.fastq", "bc.csv", "out", "--fastq2", "b.fastq", "-c", "-z"], _execute=spy) == 0
    (args, kw), = calls
    assert args == ("a.fastq", "b.fastq", "bc.csv", "out") and kw["classify_both"] is True and kw["gzip_output"] is True
    assert kw["output_prefix1"] == "" and kw["output_prefix2"] == ""
    # directory mode: sorted FASTQ files, pairs matched by position; count mismatch is an error (cli.jl:145-176)
    d1, d2 = tmp_path / "r1", tmp_path / "r2"
    d1.mkdir(), d2.mkdir()
    for n in ("b.fastq", "a.fq.gz", "notes.txt"):
        (d1 / n).write_text("")
    for n in ("y.fastq", "x.fastq"):
        (d2 / n).write_text("")
    calls.clear()
    assert cli.main([str(d1), "bc.csv", "out"], _execute=spy) == 0
    assert [os.path.basename(c[0][0]) for c in calls] == ["a.fq.gz", "b.fastq"]
    calls.clear()
    assert cli.main([str(d1), "bc.csv", "out", "--fastq2", str(d2)], _execute=spy) == 0
    assert [(os.path.basename(c[0][0]), os.path.basename(c[0][1])) for c in calls] == [("a.fq.gz", "x.fastq"), ("b.fastq", "y.fastq")]
    (d2 / "z.fastq").write_text("")
    assert cli.main([str(d1), "bc.csv", "out", "--fastq2", str(d2)], _execute=spy) == 1
    assert cli.main([str(d1), "bc.csv", "out", "--fastq2", "file.fastq"], _execute=spy) == 1
    assert cli.main(["file.fastq", "bc.csv", "out", "--fastq2", str(d2)], _execute=spy) == 1
    empty = tmp_path / "empty"
    empty.mkdir()
    assert cli.main([str(empty), "bc.csv", "out"], _execute=spy) == 1


def test_cli_directory_mode_reproduces_the_demo1_golden(tmp_path):
    """integration/single_barcode.jl:2-11 started from the command line in DIRECTORY mode: the 24 FASTQ
    files of demo1_R1 are processed in sorted order into one output directory (cli.jl:209-247)."""
    out = str(tmp_path / "out")
    rc = cli.main([os.path.join(FQ, "demo1_R1"), os.path.join(H.REF, "reference_files", "demo1.tsv"), out],
                  _execute=_run_oracle)
    assert rc == 0
    assert H.check_output_files(out, os.path.join(H.REF, "results", "demo1_R1")) > 0


def test_cli_reports_errors_with_exit_code_1(tmp_path, capsys):
    rc = cli.main([str(tmp_path / "missing.fastq"), str(tmp_path / "missing.csv"), str(tmp_path / "o")], _execute=_run_oracle)
    assert rc == 1
    assert capsys.readouterr().err  # the exception is reported on stderr (cli.jl:326-329)
