"""DemuxStats histograms collected ON THE DEVICE (summary = true; classification.jl:827-865) against the same
histograms accumulated on the host from the oracle's per-pass outputs: start positions (incl. origins before
the read), lengths, round(score, digits=2) keys, global and per barcode, pass 1 and pass 2; tables that grow
with the longest read seen; reset; the all-reduced twins (1 rank); gzip output through the HIP path."""
import gzip
import os

import numpy as np
import pytest

import helpers as H
from biodemux_jl_amd import hipabi, synth
from biodemux_jl_amd.classification import DemuxStats

pytestmark = pytest.mark.gpu
FIELDS = [f"{t}_{k}" for t in ("bc1", "bc2") for k in ("pos_counts", "len_counts", "score_counts", "per_bc_pos_counts",
                                                       "per_bc_len_counts", "per_bc_score_counts")]


def _expected(cfg, batches):
    st = DemuxStats()
    oc = H.orc.OracleClassifier(cfg, nthreads=16, want_pass=True)
    for seq, off in batches:
        st.add_pass_outputs(oc.classify(seq, off), float(cfg.min_delta))
    return st, oc.counts


def _same(a: DemuxStats, b: DemuxStats):
    for f in FIELDS:
        assert getattr(a, f) == getattr(b, f), f


@pytest.mark.parametrize("kw", [
    dict(max_error_rate=0.2),
    dict(max_error_rate=0.1, trim_side=3, min_delta=0.05),
    dict(max_error_rate=0.25, mismatch=1, indel=2, min_delta=0.15),
    dict(max_error_rate=0.3, nindel=1),
    dict(max_error_rate=0.15, matching_algorithm="hamming"),
    dict(matching_algorithm="exact"),
], ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_device_histograms_equal_host_accumulation(kw):
    bcs = synth.make_barcodes(96, 24, seed=81)
    seq, off, _ = synth.make_reads(bcs, 30000, 150, seed=82, repeat=dict(frac=0.2))
    cfg = H.bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[f"b{i}" for i in range(96)], summary=True, **kw)
    exp, counts = _expected(cfg, [(seq, off)])
    with H.bdx.HipClassifier(cfg) as hc:
        hc.classify(seq, off)
        got = DemuxStats()
        got.add_device_tables(hc.stats_tables(), cfg)
        assert np.array_equal(hc.counts, counts)
        _same(got, exp)
        assert sum(got.bc1_pos_counts.values()) == int((np.asarray(counts)[1]))  # single pass: every match has a position
        hc.allreduce_counts()  # no communicator: the "sum over one rank"
        red = DemuxStats()
        red.add_device_tables(hc.stats_tables(reduced=True), cfg)
        _same(red, exp)
        hc.reset_counts()
        empty = DemuxStats()
        empty.add_device_tables(hc.stats_tables(), cfg)
        assert not empty.bc1_pos_counts and hc.counts.sum() == 0
    assert len(exp.bc1_score_counts) >= 1 and len(exp.bc1_per_bc_pos_counts) > 50


def test_device_histograms_dual_and_growing_reads():
    """Dual barcodes (pass 2 runs only after pass 1 matched), short barcodes whose origins reach before the read
    (start <= 0), batches of increasing read length (the tables grow by appending rows), a one-rank RCCL
    communicator for the reduced twins."""
    b1 = synth.make_barcodes(20, 16, seed=83, min_hamming=5)
    b2 = synth.make_barcodes(12, 16, seed=84, min_hamming=5)
    cfg = H.bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[16] * 20, ids=[f"x{i}" for i in range(20)], is_dual=True,
                            bc_seqs2=b2, bc_lengths_no_N2=[16] * 12, ids2=[f"y{i}" for i in range(12)],
                            max_error_rate=0.25, trim_side=5, trim_side2=3, summary=True)
    batches = []
    for k, L in enumerate((60, 90, 400, 120)):
        seq, off, _ = synth.make_ragged_reads(b1, 6000, L // 2, L, seed=85 + k, plant_lo=0, plant_hi=L // 4,
                                              second=(b2, L // 2, None))
        seq = seq.copy()
        for i in range(0, 6000, 7):  # truncated barcodes at the very start: alignments that begin "before" the read
            n = int(off[i + 1] - off[i])
            if n >= 12:
                seq[off[i]:off[i] + 12] = np.frombuffer(b1[i % 20][4:].encode(), dtype=np.uint8)
        batches.append((seq, off))
    exp, counts = _expected(cfg, batches)
    with H.bdx.HipClassifier(cfg) as hc:
        hipabi.comm_init_all([hc])
        for seq, off in batches:
            hc.classify(seq, off)
        got = DemuxStats()
        got.add_device_tables(hc.stats_tables(), cfg)
        assert np.array_equal(hc.counts, counts)
        _same(got, exp)
        hipabi.allreduce_counts_all([hc])
        red = DemuxStats()
        red.add_device_tables(hc.stats_tables(reduced=True), cfg)
        _same(red, exp)
        assert np.array_equal(hc.reduced_counts, counts)
    assert min(exp.bc1_pos_counts) <= 0, "the case must contain origins before the read"
    assert exp.bc2_pos_counts and max(exp.bc1_pos_counts) > 60


def test_gzip_in_gzip_out_through_the_hip_path(tmp_path):
    """§8(f) rank 2 on the GPU: .fastq.gz in -> classify (HIP) -> .fastq.gz out (native writer: independent gzip
    members), compared record-for-record with the plain-Python writer driven by the oracle."""
    bcs = synth.make_barcodes(24, 16, seed=91, min_hamming=5)
    seq, off, _ = synth.make_reads(bcs, 40000, 100, seed=92)
    fq = str(tmp_path / "reads.fastq.gz")
    with gzip.open(fq, "wb", compresslevel=1) as g:
        for i in range(40000):
            s = seq[off[i]:off[i + 1]].tobytes()
            g.write(b"@r%d\n%s\n+\n%s\n" % (i, s, b"F" * len(s)))
    bc = str(tmp_path / "bc.fasta")
    with open(bc, "w") as f:
        for i, b in enumerate(bcs):
            f.write(f">s{i}\n{b}\n")
    a, b_ = str(tmp_path / "hip"), str(tmp_path / "ref")
    kw = dict(max_error_rate=0.2, trim_side=5, min_delta=0.05)
    H.bdx.execute_demultiplexing(fq, bc, a, _io="native", _batch_reads=15000, **kw)           # gzip_output defaults to true (.gz input)
    H.bdx.execute_demultiplexing(fq, bc, b_, _io="python", _classifier_factory=H.oracle_factory, **kw)
    names = sorted(os.listdir(b_))
    assert names == sorted(os.listdir(a)) and len(names) > 20 and all(n.endswith(".fastq.gz") for n in names)
    for n in names:
        assert H._read_maybe_gz(os.path.join(a, n)) == H._read_maybe_gz(os.path.join(b_, n)), n
