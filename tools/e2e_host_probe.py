#!/usr/bin/env python3
"""Stage breakdown of the end-to-end file pipeline on the box it runs on.

  N=10000000 REPS=4 python tools/e2e_host_probe.py            # the HIP classifier (GPU box)
  FAKE=1 N=3000000 python tools/e2e_host_probe.py             # verdicts = the planted truth: reader + writer alone, no GPU

Prints per run: wall seconds, busy seconds of index / pack / classify / write, and — BDX_IO_TIMING is set for the child
library — the writer's phases summed over the batches (sizes, gather / iovecs, files).  Environment knobs of libbdx_io.so
(BDX_IO_IOV_LIMIT ...) are passed through, so variants are compared by running the script once per setting."""
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np

    import biodemux_jl_amd as bdx
    from biodemux_jl_amd import synth

    n = int(os.environ.get("N", "10000000"))
    root = os.environ["E2E_ROOT"]
    fq = os.path.join(root, "synthetic.fastq")
    bc = os.path.join(root, "barcodes.csv")
    bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
    seq, off, truth = synth.make_reads(bcs, n, 150, seed=synth.SEED)
    if not os.path.exists(fq) or os.path.getsize(fq) != n * 319:
        rec = np.empty((n, 319), dtype=np.uint8)
        rec[:, 0:5] = np.frombuffer(b"@read", dtype=np.uint8)
        ids = np.arange(n, dtype=np.int64)
        for k in range(9):
            rec[:, 13 - k] = (ids // 10 ** k % 10 + 48).astype(np.uint8)
        rec[:, 14] = 10
        rec[:, 15:165] = seq.reshape(n, 150)
        rec[:, 165:168] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        rec[:, 168:318] = ord("F")
        rec[:, 318] = 10
        rec.tofile(fq)
        del rec
        with open(bc, "w") as f:
            f.write("ID,Full_seq,Full_annotation\n" + "".join(f"bc{i + 1:03d},{b},{'B' * 24}\n" for i, b in enumerate(bcs)))
    del seq, off

    class Fake:
        """test double (FAKE=1): the planted truth as verdicts — times the host stages without a GPU"""

        def __init__(self, config):
            self.pos = 0
            self.counts = np.zeros(4 + 96, dtype=np.int64)

        def classify(self, seq, so, out=None):
            k = len(so) - 1
            t = truth[self.pos:self.pos + k]
            self.pos += k
            return {"bc1": t.astype(np.int32), "bc2": np.zeros(k, np.int32), "keep_start": np.ones(k, np.int32), "keep_end": np.full(k, 150, np.int32)}

        def close(self):
            pass

    kw = {"_classifier_factory": Fake} if os.environ.get("FAKE") else {}
    if os.environ.get("BATCH"):
        kw["_batch_reads"] = int(os.environ["BATCH"])
    for rep in range(int(os.environ.get("REPS", "4"))):
        out = os.path.join(root, "out")
        shutil.rmtree(out, ignore_errors=True)
        tm = {}
        t = time.perf_counter()
        bdx.execute_demultiplexing(fq, bc, out, max_error_rate=0.1, _io="native", _timings=tm, **kw)
        dt = time.perf_counter() - t
        nb = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
        assert nb == n * 319, (nb, n * 319)
        print(f"RUN {rep} {dt:.4f} {tm['index_s']:.4f} {tm['pack_s']:.4f} {tm['classify_s']:.4f} {tm['write_s']:.4f} {tm['wall_s']:.4f} {tm['threads']} {tm.get('pre_s', 0):.4f} {tm.get('close_s', 0):.4f} {tm.get('total_s', 0):.4f} {tm.get('native_call_s', 0):.4f}", flush=True)
        sys.stderr.write(f"[probe] end of run {rep}\n")
        sys.stderr.flush()
    shutil.rmtree(out, ignore_errors=True)


def main():
    n = int(os.environ.get("N", "10000000"))
    root = os.environ.get("E2E_ROOT") or ("/dev/shm/bdx_e2e_probe" if os.path.isdir("/dev/shm") else "/tmp/bdx_e2e_probe")
    os.makedirs(root, exist_ok=True)
    env = dict(os.environ, E2E_ROOT=root, BDX_IO_TIMING="1", BDX_E2E_CHILD="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-3000:])
        sys.exit(p.returncode)
    runs = [l.split() for l in p.stdout.splitlines() if l.startswith("RUN ")]
    phases = []
    cur = [0.0, 0.0, 0.0, 0.0]
    for l in p.stderr.splitlines():
        m = re.search(r"writer: sizes ([\d.]+) ms, buffers ([\d.]+) ms, gather / iovecs ([\d.]+) ms, deflate \+ files ([\d.]+) ms", l)
        if m:
            for k in range(4):
                cur[k] += float(m.group(k + 1)) / 1e3
        elif "[probe] end of run" in l:
            phases.append(cur)
            cur = [0.0, 0.0, 0.0, 0.0]
    label = os.environ.get("LABEL", "")
    for r, ph in zip(runs, phases):
        dt = float(r[2])
        print(f"{label:28s} run {r[1]}: {dt:.3f} s = {n / dt / 1e6:5.1f} M reads/s | index {float(r[3]):.3f} pack {float(r[4]):.3f} classify {float(r[5]):.3f} "
              f"write {float(r[6]):.3f} (sizes {ph[0]:.3f} gather {ph[2]:.3f} files {ph[3]:.3f}) pipeline {float(r[7]):.3f} threads {r[8]} | before {float(r[9]):.3f} close {float(r[10]):.3f} inside-call {float(r[11]):.3f} native-call {float(r[12]):.3f}")
    if not os.environ.get("KEEP"):
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    if os.environ.get("BDX_E2E_CHILD"):
        child()
    else:
        main()
