"""Native host I/O (csrc/bdx_io.cpp) must behave exactly like the plain-Python reader/writer of
core.py (itself held to the reference's goldens): same files, same bytes, same order — incl. CRLF
input, a truncated last record, a trailing blank line, paired lock-step, trimming, gzip in/out."""
import functools
import gzip
import os

import numpy as np
import pytest

import helpers as H
from biodemux_jl_amd import nativeio, synth

run_py = functools.partial(H.bdx.execute_demultiplexing, _classifier_factory=H.oracle_factory, _io="python")
run_nat = functools.partial(H.bdx.execute_demultiplexing, _classifier_factory=H.oracle_factory, _io="native")


@pytest.fixture(scope="module", autouse=True)
def _built():
    nativeio.build()


def test_goldens_through_native_io(tmp_path):
    assert H.scenario_demo1_R1(run_nat, str(tmp_path / "a")) == 24
    assert H.scenario_demo1_R2(run_nat, str(tmp_path / "b")) == 24
    assert H.scenario_demo2(run_nat, str(tmp_path / "c")) == 76  # gz in, classify_both, revcomp


@pytest.mark.parametrize("scenario", H.SCENARIOS_SMALL, ids=[s.__name__ for s in H.SCENARIOS_SMALL])
def test_reference_scenarios_through_native_io(tmp_path, scenario):
    scenario(run_nat, str(tmp_path))


def _same_tree(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb
    for f in fa:
        assert H._read_maybe_gz(os.path.join(a, f)) == H._read_maybe_gz(os.path.join(b, f)), f


def _fastq(path, seqs, crlf=False, tail=b"", gz=False):
    nl = b"\r\n" if crlf else b"\n"
    blob = b"".join(b"@r%d some header" % i + nl + s + nl + b"+" + nl + b"I" * len(s) + nl for i, s in enumerate(seqs)) + tail
    (gzip.open if gz else open)(path, "wb").write(blob)


@pytest.mark.parametrize("case", ["plain", "crlf", "truncated", "blankline", "no_final_newline", "gz", "small_batches"])
def test_native_equals_python_io(tmp_path, case):
    bcs = synth.make_barcodes(6, 12, seed=5, min_hamming=4)
    seq, off, _ = synth.make_ragged_reads(bcs, 700, 0, 90, seed=5)
    seqs = [seq[off[i]:off[i + 1]].tobytes() for i in range(700)]
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    tail = {"truncated": b"@last\nACGTAC", "blankline": b"\n", "no_final_newline": b"@x\nACGT\n+\nIIII"}.get(case, b"")
    fq = str(tmp_path / ("reads.fastq.gz" if case == "gz" else "reads.fastq"))
    _fastq(fq, seqs, crlf=(case == "crlf"), tail=tail, gz=(case == "gz"))
    kw = dict(max_error_rate=0.2, trim_side=5)
    if case == "small_batches":
        kw["_batch_reads"] = 37
    run_py(fq, str(bc), str(tmp_path / "py"), **kw)
    run_nat(fq, str(bc), str(tmp_path / "nat"), **kw)
    _same_tree(str(tmp_path / "py"), str(tmp_path / "nat"))


def test_native_paired_lockstep_and_classify_both(tmp_path):
    bcs = synth.make_barcodes(5, 12, seed=6, min_hamming=4)
    seq, off, _ = synth.make_reads(bcs, 500, 60, seed=6)
    s1 = [seq[off[i]:off[i + 1]].tobytes() for i in range(500)]
    s2 = [b"ACGT" * 10 for _ in range(430)]  # R2 is shorter: stop at the shorter file
    bc = tmp_path / "bc.tsv"
    bc.write_text("ID\tFull_seq\tFull_annotation\n" + "".join(f"b{i}\t{b}\t{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    f1, f2 = str(tmp_path / "x_R1.fastq"), str(tmp_path / "x_R2.fastq")
    _fastq(f1, s1)
    _fastq(f2, s2)
    for both in (False, True):
        kw = dict(classify_both=both, trim_side=3, _batch_reads=128)
        run_py(f1, f2, str(bc), str(tmp_path / f"py{both}"), **kw)
        run_nat(f1, f2, str(bc), str(tmp_path / f"nat{both}"), **kw)
        _same_tree(str(tmp_path / f"py{both}"), str(tmp_path / f"nat{both}"))


@pytest.mark.parametrize("case", ["gz_small_batches", "gz_truncated_record", "gz_no_final_newline", "gz_out"])
def test_streamed_gzip_input(tmp_path, case):
    """.gz inputs are inflated by a background thread while batches are already indexed, classified
    and written: batch boundaries must not depend on how far the inflater is,
    released pages must never be read again, the tail rules are those of plain files."""
    bcs = synth.make_barcodes(6, 12, seed=7, min_hamming=4)
    seq, off, _ = synth.make_ragged_reads(bcs, 3000, 0, 120, seed=7)
    seqs = [seq[off[i]:off[i + 1]].tobytes() for i in range(3000)]
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    tail = {"gz_truncated_record": b"@last\nACGTAC", "gz_no_final_newline": b"@x\nACGT\n+\nIIII"}.get(case, b"")
    fq = str(tmp_path / "reads.fastq.gz")
    _fastq(fq, seqs, tail=tail, gz=True)
    kw = dict(max_error_rate=0.2, trim_side=3, _batch_reads=211)
    if case == "gz_out":
        kw["gzip_output"] = True
    run_py(fq, str(bc), str(tmp_path / "py"), **kw)
    run_nat(fq, str(bc), str(tmp_path / "nat"), **kw)
    _same_tree(str(tmp_path / "py"), str(tmp_path / "nat"))


def test_streamed_gzip_pairs_and_errors(tmp_path):
    bcs = synth.make_barcodes(5, 12, seed=8, min_hamming=4)
    seq, off, _ = synth.make_reads(bcs, 900, 60, seed=8)
    s1 = [seq[off[i]:off[i + 1]].tobytes() for i in range(900)]
    s2 = [b"TTGCA" * 9 for _ in range(900)]
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    f1, f2 = str(tmp_path / "y_R1.fastq.gz"), str(tmp_path / "y_R2.fastq.gz")
    _fastq(f1, s1, gz=True)
    _fastq(f2, s2, gz=True)
    kw = dict(classify_both=True, _batch_reads=100)
    run_py(f1, f2, str(bc), str(tmp_path / "py"), **kw)
    run_nat(f1, f2, str(bc), str(tmp_path / "nat"), **kw)
    _same_tree(str(tmp_path / "py"), str(tmp_path / "nat"))
    # a corrupt stream is reported, not silently truncated
    blob = open(f1, "rb").read()
    bad = str(tmp_path / "bad.fastq.gz")
    open(bad, "wb").write(blob[:len(blob) // 2] + b"\x00" * 64 + blob[len(blob) // 2 + 64:])
    with pytest.raises(OSError):
        run_nat(bad, str(bc), str(tmp_path / "bad_out"), _batch_reads=100)


# ---- block-parallel inflate of size-tagged member chains ----
def _bgzf(data: bytes, block: int = 60000) -> bytes:
    """BGZF (bgzip) framing: gzip members of <= 64 KiB whose extra subfield 'BC' holds the member size - 1."""
    import struct
    import zlib

    out = []
    for o in list(range(0, len(data), block)) + [None]:  # + the empty EOF block bgzip appends
        chunk = data[o:o + block] if o is not None else b""
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(chunk) + co.flush()
        bsize = 12 + 6 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
                   + body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))
    return b"".join(out)


def _case(tmp_path, n=30000):
    bcs = synth.make_barcodes(8, 12, seed=17, min_hamming=4)
    seq, off, _ = synth.make_ragged_reads(bcs, n, 20, 90, seed=17)
    seqs = [seq[off[i]:off[i + 1]].tobytes() for i in range(n)]
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    return seqs, str(bc)


@pytest.mark.parametrize("kind", ["bgzf", "own_writer", "tagged_then_plain", "plain"])
def test_parallel_inflate_of_tagged_members(tmp_path, kind):
    """BGZF input and this library's own .gz output (members tagged with their compressed size) are inflated
    block-parallel; an ordinary gzip stream — also one that FOLLOWS tagged members in the same file — is read
    serially from where the chain ends.  The demultiplexed files are the same whatever the framing."""
    seqs, bc = _case(tmp_path)
    plain = str(tmp_path / "reads.fastq")
    _fastq(plain, seqs)
    raw = open(plain, "rb").read()
    fq = str(tmp_path / "framed.fastq.gz")
    if kind == "bgzf":
        open(fq, "wb").write(_bgzf(raw))
    elif kind == "plain":
        open(fq, "wb").write(gzip.compress(raw, 1))
    else:
        # this library's writer: demultiplex with gzip output, then feed the largest of ITS outputs back in
        d0 = str(tmp_path / "first")
        run_nat(plain, bc, d0, max_error_rate=0.2, gzip_output=True, _batch_reads=7000)
        big = max((os.path.join(d0, n) for n in os.listdir(d0)), key=os.path.getsize)
        raw = gzip.open(big, "rb").read()
        blob = open(big, "rb").read()
        assert blob.count(b"DX\x04\x00") >= 4  # several members (one per 7000-read batch), each tagged
        if kind == "tagged_then_plain":
            extra = b"@tail\nACGTACGTAC\n+\nFFFFFFFFFF\n" * 50
            blob += gzip.compress(extra, 6)
            raw += extra
        open(fq, "wb").write(blob)
        plain = str(tmp_path / "again.fastq")
        open(plain, "wb").write(raw)
    assert gzip.open(fq, "rb").read() == raw  # any gzip reader accepts the framing
    for T in (1, 5):
        f = nativeio.FastqFile(fq, T)
        try:
            assert f.size == len(raw)
            assert f.parallel_inflate == (kind != "plain")
        finally:
            f.close()
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    run_nat(fq, bc, a, max_error_rate=0.2, gzip_output=False, _batch_reads=9000, output_prefix="p")
    run_py(plain, bc, b, max_error_rate=0.2, output_prefix="p")
    _same_tree(a, b)


def test_parallel_inflate_rejects_a_corrupt_member(tmp_path):
    from biodemux_jl_amd import nativeio

    blob = bytearray(_bgzf(b"@r\nACGT\n+\nFFFF\n" * 20000))
    blob[len(blob) // 2] ^= 0x55
    fq = str(tmp_path / "bad.fastq.gz")
    open(fq, "wb").write(bytes(blob))
    f = nativeio.FastqFile(fq, 4)
    try:
        assert f.size == -1
    finally:
        f.close()


@pytest.mark.parametrize("where", ["classify", "write"])
def test_pipeline_errors_propagate_instead_of_hanging(tmp_path, where):
    """An error in the classify stage (a BdxError) or in the writer (ENOSPC, a vanished directory) with many batches still
    to come must come out of execute_demultiplexing: no batch buffer may stay behind on an error path (the reader waits
    for buffers) — the pipeline used to hang there, holding the GPU context."""
    import shutil
    import threading

    bcs = synth.make_barcodes(6, 12, seed=7, min_hamming=4)
    seq, off, _ = synth.make_ragged_reads(bcs, 2000, 20, 60, seed=7)
    seqs = [seq[off[i]:off[i + 1]].tobytes() for i in range(2000)]
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    fq = str(tmp_path / "reads.fastq")
    _fastq(fq, seqs)
    out_dir = tmp_path / "out"
    calls = [0]

    class Failing:
        def __init__(self, cfg):
            self.inner = H.oracle_factory(cfg)
            self.counts = self.inner.counts

        def classify(self, s, o):
            calls[0] += 1
            if calls[0] == 2:
                if where == "classify":
                    raise RuntimeError("classifier failed on batch 2")
                shutil.rmtree(out_dir)  # the writer's next open fails
            return self.inner.classify(s, o)

        def close(self):
            pass

    result = {}

    def run():
        try:
            H.bdx.execute_demultiplexing(fq, str(bc), str(out_dir), max_error_rate=0.2, _classifier_factory=Failing, _io="native",
                                         _batch_reads=100)  # 20 batches
            result["ok"] = True
        except BaseException as e:  # noqa: BLE001
            result["err"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(60)
    assert not t.is_alive(), "the pipeline hangs on an error"
    assert "err" in result, result
    if where == "classify":
        assert "batch 2" in str(result["err"])


def test_batch_buffers_are_reused_across_runs(tmp_path):
    """The pipeline's batch buffers (line tables, packed chunk, verdict vectors) are kept in a process-wide pool between runs:
    consecutive runs with other inputs, batch sizes, trim settings and read lengths must not see each other's contents
    (byte-exact against the Python reader / writer every time); `release_buffers()` empties the pool."""
    from biodemux_jl_amd import nativeio
    nativeio.release_buffers()
    bcs = synth.make_barcodes(6, 12, seed=15, min_hamming=4)
    bc = tmp_path / "bc.csv"
    bc.write_text("ID,Full_seq,Full_annotation\n" + "".join(f"b{i},{b},{'B' * len(b)}\n" for i, b in enumerate(bcs)))
    runs = [(900, 20, 90, dict(max_error_rate=0.2, trim_side=5, _batch_reads=128), False),
            (300, 0, 40, dict(max_error_rate=0.1, _batch_reads=64), True),      # fewer, shorter reads, CRLF: stale tails of every buffer
            (1500, 60, 150, dict(max_error_rate=0.2, trim_side=3, _batch_reads=500), False),
            (10, 5, 20, dict(max_error_rate=0.2, _batch_reads=4), False)]
    for k, (n, lo, hi, kw, crlf) in enumerate(runs):
        seq, off, _ = synth.make_ragged_reads(bcs, n, lo, hi, seed=20 + k)
        seqs = [seq[off[i]:off[i + 1]].tobytes() for i in range(n)]
        fq = str(tmp_path / f"reads{k}.fastq")
        _fastq(fq, seqs, crlf=crlf)
        run_py(fq, str(bc), str(tmp_path / f"py{k}"), **kw)
        run_nat(fq, str(bc), str(tmp_path / f"nat{k}"), **kw)
        _same_tree(str(tmp_path / f"py{k}"), str(tmp_path / f"nat{k}"))
        assert 1 <= len(nativeio._BUFFER_POOL) <= nativeio._BUFFER_POOL_MAX
    nativeio.release_buffers()
    assert nativeio._BUFFER_POOL == []
