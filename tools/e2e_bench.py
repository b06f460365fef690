#!/usr/bin/env python3
"""End-to-end survey: FASTQ file on disk -> execute_demultiplexing -> 98 output
files, through the native host I/O and the HIP hot path.  Reports reads/s and the split."""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import biodemux_jl_amd as bdx
from biodemux_jl_amd import synth

n = int(os.environ.get("N", "4000000"))
root = os.environ.get("E2E_DIR") or tempfile.mkdtemp(prefix="bdx_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
bcs = synth.make_barcodes(96, 24)
seq, off, _ = synth.make_reads(bcs, n, 150)
t = time.time()
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
fq = os.path.join(root, "synthetic.fastq")
rec.tofile(fq)
bc = os.path.join(root, "barcodes.csv")
open(bc, "w").write("ID,Full_seq,Full_annotation\n" + "".join(f"bc{i + 1:03d},{b},{'B' * 24}\n" for i, b in enumerate(bcs)))
print(f"wrote {os.path.getsize(fq) / 1e9:.2f} GB FASTQ ({n} reads) in {time.time() - t:.1f} s -> {root}", flush=True)
del rec
gz = bool(os.environ.get("GZ"))  # GZ=1: .fastq.gz in (streamed inflate), gzip out
extra = {}
if gz:
    import subprocess
    t = time.time()
    subprocess.check_call(["gzip", "-1", "-k", fq])
    print(f"gzip -1: {os.path.getsize(fq + '.gz') / 1e9:.2f} GB in {time.time() - t:.1f} s", flush=True)
    t = time.time()
    subprocess.check_call("gzip -dc %s.gz > /dev/null" % fq, shell=True)
    print(f"gzip -dc alone (the serial floor of a single gzip stream): {time.time() - t:.2f} s", flush=True)
    os.remove(fq)
    fq = fq + ".gz"
    extra = dict(gzip_output=True)
for io in ("native",):
    for rep in range(2):
        out = os.path.join(root, f"out_{io}_{rep}")
        t = time.perf_counter()
        st = bdx.execute_demultiplexing(fq, bc, out, max_error_rate=0.1, _io=io, **extra)
        dt = time.perf_counter() - t
        nfiles = len(os.listdir(out))
        print(f"io={io} run {rep}: {dt:.2f} s -> {n / dt / 1e6:.2f} M reads/s end-to-end ({os.path.getsize(fq) / dt / 1e9:.2f} GB/s of FASTQ), "
              f"{nfiles} files, matched {st.matched_reads}/{st.total_reads}", flush=True)
        shutil.rmtree(out)
shutil.rmtree(root, ignore_errors=True)
