"""A plain C11 host of the C-ABI (tests/abi_driver.c): the header compiles as C, the struct layouts the Julia
shim of INTEGRATION.md assumes hold (static assertions), config validation answers before any device is
touched; on the GPU it classifies one of the reference's demo1 FASTQ files and runs the one-process
merge_stats sequence over RCCL (bdx_comm_init_all / bdx_allreduce_counts_all)."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from biodemux_jl_amd import hipabi


def _build(tmp_path) -> str:
    exe = str(tmp_path / "abi_driver")
    csrc = os.path.dirname(hipabi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", os.path.join(H.ROOT, "include"),
                           os.path.join(H.ROOT, "tests", "abi_driver.c"), "-o", exe, "-L", csrc, "-lbiodemux_hip",
                           f"-Wl,-rpath,{csrc}"])
    return exe


def test_c_host_compiles_and_layout_holds(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "--layout"], check=True, capture_output=True, text=True).stdout
    assert "layout ok" in out and "trim_side must be 3 or 5" in out
    assert f"config {hipabi.C.sizeof(hipabi.BdxConfig)}" in out  # the ctypes mirror agrees with the C compiler


@pytest.mark.gpu
@pytest.mark.parametrize("trim", [0, 3])
def test_c_host_classifies_demo1(tmp_path, trim):
    exe = _build(tmp_path)
    src = os.path.join(H.REF, "FASTQ_files", "demo1_R1")
    fastq = os.path.join(src, sorted(os.listdir(src))[0])
    bcs, nn, ids = H.bdx.preprocess_bc_file(os.path.join(H.REF, "reference_files", "demo1.tsv"), False, False)
    bcf = tmp_path / "barcodes.txt"
    bcf.write_text("\n".join(bcs) + "\n")
    out = subprocess.run([exe, str(bcf), fastq, "0.2", str(trim)], check=True, capture_output=True, text=True).stdout.splitlines()
    # (RCCL may add lines of its own to stdout: pick ours by their shape)
    cl = [ln for ln in out if ln.startswith("counts:")]
    rl = [ln for ln in out if ln.startswith("reduced (1 rank, rank 0):")]
    assert len(cl) == 1 and len(rl) == 1, out[-5:]
    got = np.array([[int(x) for x in ln.split()] for ln in out[:out.index(cl[0])]], dtype=np.int64)
    counts = np.array(cl[0].split()[1:], dtype=np.int64)
    reduced = np.array(rl[0].split(":")[1].split(), dtype=np.int64)
    # the same reads through the oracle
    with open(fastq, "rb") as f:
        seqs = f.read().split(b"\n")[1::4]
    seq, off = H.bdx.pack_reads(seqs)
    cfg = H.bdx.DemuxConfig(bc_seqs=list(bcs), bc_lengths_no_N=list(nn), ids=list(ids), max_error_rate=0.2,
                            trim_side=trim or None)
    oc = H.orc.OracleClassifier(cfg, nthreads=4, want_pass=False)
    exp = oc.classify(seq, off)
    assert len(got) == len(off) - 1 > 0
    for j, k in enumerate(("bc1", "bc2", "keep_start", "keep_end")):
        assert np.array_equal(got[:, j], exp[k]), k
    assert np.array_equal(counts, oc.counts) and np.array_equal(reduced, counts)
