#!/usr/bin/env python3
"""bench.py — headline benchmark: reads/s demultiplexed (150 bp, 96 barcodes, :semiglobal).

Default workload (BASELINE.json configs[1], "C2"): 10 M synthetic 150 bp reads x 96 barcodes of 24 bp,
:semiglobal, max_error_rate 0.1 (allowed_error = floor(0.1*24) = 2), min_delta 0, costs 0/1/1,
all ranges "1:end", no trim (ScoreOnly).  Seed 20260515.  ``--config`` selects the other
survey configs (parity-test cases, measured here so their numbers are driver-reproducible too):
  C2d  C2 at the reference's default max_error_rate 0.2 (allowed_error 4)
  C4   dual 24 x 16 barcodes, trim_side 5 / trim_side2 3, rate 0.2 (BASELINE configs[3])
  C5   10 kbp reads x 24 barcodes of 16..32 nt, ref_search_range "1:200", rate 0.2 (configs[4])

A "step" = one pass of the hot path over one batch: the packed batch is ALREADY resident in HBM when the
timed region starts; the step classifies every read of the rank's shard through the C-ABI
(bdx_classify_device) and all-reduces the per-barcode counters (RCCL when N > 1: through the C-ABI's own
communicator, bdx_allreduce_counts).  Weak scaling: every rank owns its own shard of the global stream.

Run:  python bench.py [--gpus N --steps K --warmup W --config C2]
With N > 1 and no torchrun environment the script starts its own N ranks (python -m torch.distributed.run)
before anything touches the GPU, and exits with their code.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
METRIC = "reads/s demultiplexed (150 bp, 96 barcodes, :semiglobal) at 1/2/4/8 MI355X"


def _parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="C2", choices=["C2", "C2d", "C4", "C5", "C2t5", "C2r60", "C2dual"],
                    help="BASELINE configs C2 (headline) / C2d / C4 / C5; C2t5, C2r60, C2dual: common variants of the headline shape (developer legs)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's size)")
    ap.add_argument("--filter", default="auto", help="auto|off|qgram|bitpar (all give identical results)")
    ap.add_argument("--max-error-rate", type=float, default=None, help="override the config's rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short legs for the other BASELINE configs and the end-to-end figure")
    ap.add_argument("--e2e-reads", type=int, default=10_000_000, help="reads of the end-to-end FASTQ leg (0: skip)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--allow-wrong-results", action="store_true",
                    help="tuning only (tools/phase_*.sh: phase-skip timing experiments of a -DBDX_TUNING build give wrong results by design): skip the counter-vector consistency checks of the headline run")
    return ap.parse_args()


def _self_launch(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD job — this process has not
    imported torch nor touched the GPU, and never will — relay its output (inherited stdout: rank 0's JSON
    line) and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def _host_cores() -> int:
    """Threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota (a
    1-GPU share of the host) — oversubscribing a quota only adds scheduling noise."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(p) + 0.5)))
    except Exception:
        pass
    return cores


def build_workload(name: str, n_reads: int, first_read: int, rate_override=None):
    """-> dict(cfg, seq, off, n, read_len, algo_bytes, outputs, desc).  Shapes: SURVEY.md §8(d)."""
    import numpy as np

    import biodemux_jl_amd as bdx
    from biodemux_jl_amd import synth

    if name in ("C2", "C2d"):
        n = n_reads or 10_000_000
        rate = rate_override if rate_override is not None else (0.1 if name == "C2" else 0.2)
        bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
        seq, off, _ = synth.make_reads(bcs, n, 150, seed=synth.SEED, first_read=first_read)
        cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[f"bc{i + 1:03d}" for i in range(96)],
                              max_error_rate=rate, min_delta=0.0, match=0, mismatch=1, indel=1,
                              matching_algorithm="semiglobal")
        ae = int(np.floor(rate * 24))
        return dict(cfg=cfg, seq=seq, off=off, n=n, read_len=150, algo_bytes=150 + 8 + 4, outputs=("bc1",),
                    desc=f"{name}: {n / 1e6:g} M synthetic 150 bp reads x 96 barcodes (24 bp), :semiglobal, "
                         f"max_error_rate={rate} (allowed_error={ae}), min_delta=0, costs 0/1/1, ScoreOnly",
                    barcodes=96, barcode_len=24)
    if name in ("C2t5", "C2r60"):
        # variants of the headline shape the reference's users run all the time (not BASELINE configs: reported under
        # "extra_configs"): trimming the barcode off the 5' end (classification.jl:912-914), a restricted search window
        n = n_reads or 10_000_000
        rate = rate_override if rate_override is not None else 0.1
        bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
        seq, off, _ = synth.make_reads(bcs, n, 150, seed=synth.SEED, first_read=first_read)
        kw = dict(trim_side=5) if name == "C2t5" else dict(ref_search_range=bdx.parse_dynamic_range("1:60"))
        cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[f"bc{i + 1:03d}" for i in range(96)],
                              max_error_rate=rate, min_delta=0.0, match=0, mismatch=1, indel=1, matching_algorithm="semiglobal", **kw)
        what = "trim_side=5 (keep range = behind the alignment's end)" if name == "C2t5" else "ref_search_range=1:60, ScoreOnly"
        return dict(cfg=cfg, seq=seq, off=off, n=n, read_len=150, algo_bytes=150 + 8 + (12 if name == "C2t5" else 4),
                    outputs=("bc1", "keep_start", "keep_end") if name == "C2t5" else ("bc1",),
                    desc=f"{name}: C2 ({n / 1e6:g} M reads, rate {rate}) with {what}", barcodes=96, barcode_len=24)
    if name == "C2dual":
        n = n_reads or 10_000_000
        rate = rate_override if rate_override is not None else 0.1
        b1 = synth.make_barcodes(24, 24, seed=synth.SEED + 1)
        b2 = synth.make_barcodes(16, 24, seed=synth.SEED + 2)
        seq, off, _ = synth.make_reads(b1, n, 150, seed=synth.SEED, first_read=first_read, plant_lo=0, plant_hi=40,
                                       second=(b2, 100, 126))
        cfg = bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i + 1}" for i in range(24)], is_dual=True,
                              bc_seqs2=b2, bc_lengths_no_N2=[24] * 16, ids2=[f"y{i + 1}" for i in range(16)], max_error_rate=rate)
        return dict(cfg=cfg, seq=seq, off=off, n=n, read_len=150, algo_bytes=150 + 8 + 8, outputs=("bc1", "bc2"),
                    desc=f"C2dual: {n / 1e6:g} M synthetic 150 bp reads, dual 24 x 16 barcodes (24 bp), :semiglobal, "
                         f"max_error_rate={rate}, ScoreOnly", barcodes=40, barcode_len=24)
    if name == "C4":
        n = n_reads or 10_000_000
        rate = rate_override if rate_override is not None else 0.2
        b1 = synth.make_barcodes(24, 24, seed=synth.SEED + 1)
        b2 = synth.make_barcodes(16, 24, seed=synth.SEED + 2)
        seq, off, _ = synth.make_reads(b1, n, 150, seed=synth.SEED, first_read=first_read, plant_lo=0, plant_hi=40,
                                       second=(b2, 100, 126))
        cfg = bdx.DemuxConfig(bc_seqs=b1, bc_lengths_no_N=[24] * 24, ids=[f"x{i + 1}" for i in range(24)], is_dual=True,
                              bc_seqs2=b2, bc_lengths_no_N2=[24] * 16, ids2=[f"y{i + 1}" for i in range(16)],
                              max_error_rate=rate, trim_side=5, trim_side2=3)
        return dict(cfg=cfg, seq=seq, off=off, n=n, read_len=150, algo_bytes=150 + 8 + 4 + 8 + 4,
                    outputs=("bc1", "bc2", "keep_start", "keep_end"),
                    desc=f"C4: {n / 1e6:g} M synthetic 150 bp R1 reads, dual 24 x 16 barcodes (384 pairs, 24 bp), "
                         f":semiglobal, max_error_rate={rate}, trim_side=5, trim_side2=3 (traceback)",
                    barcodes=40, barcode_len=24)
    if name == "C5":
        n = n_reads or 400_000
        rate = rate_override if rate_override is not None else 0.2
        lens = np.random.Generator(np.random.PCG64(5)).integers(16, 33, size=24)
        bcs = synth.make_barcodes(24, 24, seed=synth.SEED + 5, lengths=lens)
        seq, off, _ = synth.make_reads(bcs, n, 10000, seed=synth.SEED, first_read=first_read, plant_lo=0, plant_hi=150)
        cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[len(b) for b in bcs], ids=[f"bc{i + 1}" for i in range(24)],
                              max_error_rate=rate, ref_search_range=bdx.parse_dynamic_range("1:200"))
        # algorithmic bytes: the path (like the reference, classification.jl:795-809) only ever looks at the 200
        # columns of ref_search_range, so a read costs 200 + 8 + 4 B.  SURVEY §8(d) lists the whole 10 kbp read
        # (10 012 B); pricing the launch with that figure would put "achieved" above the HBM peak.
        return dict(cfg=cfg, seq=seq, off=off, n=n, read_len=10000, algo_bytes=200 + 8 + 4, survey_bytes=10000 + 8 + 4, outputs=("bc1",),
                    desc=f"C5: {n / 1e3:g} k synthetic 10 kbp reads x 24 barcodes (16..32 nt), :semiglobal, "
                         f"max_error_rate={rate}, ref_search_range=1:200, ScoreOnly",
                    barcodes=24, barcode_len="16..32")
    raise KeyError(name)


def kernel_source_sha16() -> str:
    """sha256 over the sources the HIP library is built from (csrc/*.hip, *.h, *.cpp except the host-only I/O and the
    generator, include/*.h), first 16 hex digits — the identity of "these kernels" for the committed counter profiles."""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "biodemux.jl_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc)
                   if f.endswith((".hip", ".h", ".cpp", ".inc")) and f not in ("bdx_io.cpp",))
    files += sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _profile_counters(config: str, n: int):
    """(traffic bytes per step, source, per-kernel VALU figures) from the latest committed rocprofv3 --pmc summary of
    this config (profiles/*_<config>_*pmc.json, tools/summarize_prof.py), or (None, None, None)."""
    traffic, src_s, valu = None, None, None
    try:
        tagged = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc.json") and f"_{config}_" in f)
        if not tagged:
            return None, None, None
        pj = json.load(open(os.path.join(ROOT, "profiles", tagged[-1])))
        if pj.get("reads_per_launch") != n:
            return None, None, None
        # the counters belong to the kernels they were collected on: a profile taken on other kernel sources than the
        # ones this library was built from is not paired with this run's time (tools/summarize_prof.py records the hash)
        if pj.get("kernel_source_sha16") != kernel_source_sha16():
            return None, (f"profiles/{tagged[-1]} is STALE: taken on kernel sources {pj.get('kernel_source_sha16')}, this tree has "
                          f"{kernel_source_sha16()} — re-profile (tools/profile_round.sh)"), None
        tot, src = 0.0, []
        for kname, e in pj["kernels"].items():
            if "hbm_read_bytes_corrected" in e:
                tot += e["hbm_read_bytes_corrected"] + e.get("hbm_write_bytes", 0.0)
                src.append(kname)
            c = e.get("counters_per_step", {})
            if c.get("SQ_INSTS_VALU", 0.0) / n >= 1.0 and c.get("SQ_WAVE_CYCLES"):
                # VALU wave-instructions per read of the batch (all of the kernel's launches in one step) and the fraction
                # of its waves' lifetime spent issuing.  SIMD cycles per issued VALU wave-instruction: SQ_BUSY_CYCLES has
                # 32 counter instances for 1024 SIMDs; the pipe's limit depends on the instruction mix (tools/ubench_ops.hip:
                # v_and/or/xor/add/sub/lshrrev ~2.4, nearly everything else — v_bfe, v_alignbit, v_perm, v_lshl_or,
                # carries, v_min — ~4.3 at the clock the chip holds under load)
                valu = (valu or []) + [{
                    "kernel": kname, "launches_per_step": e.get("meta", {}).get("dispatches_per_step"),
                    "valu_wave_instr_per_read": round(c.get("SQ_INSTS_VALU", 0.0) / n, 1),
                    "simd_cycles_per_valu_instr": (round(c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0 / c["SQ_INSTS_VALU"], 2)
                                                   if c.get("SQ_BUSY_CYCLES") else None),
                    "valu_ceiling_cycles_per_instr_by_class": {"v_and/or/xor/add/sub/lshrrev": 2.4, "everything else": 4.3, "source": "profiles/r03_ubench_ops.txt"},
                    "wave_issue_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}]
        if src:
            traffic, src_s = tot, f"profiles/{tagged[-1]} ({' + '.join(src)})"
    except Exception:
        pass
    return traffic, src_s, valu


def run_leg(name: str, dev, dev_index: int, steps: int = 5, warmup: int = 2, sample: int = 50_000):
    """One of the other BASELINE configs, short: device-resident throughput over `steps` classify calls, the kernels'
    HIP-event time, the HBM fraction, the traffic ratio of the committed profile, and an oracle check on a sample."""
    import numpy as np
    import torch

    import biodemux_jl_amd as bdx

    t0 = time.time()
    wl = build_workload(name, 0, 0, None)
    gen_s = time.time() - t0
    cfg, seq, off, n = wl["cfg"], wl["seq"], wl["off"], wl["n"]
    hc = bdx.HipClassifier(cfg, device=dev_index)
    try:
        stream = torch.cuda.Stream(dev)
        hc.set_stream(stream.cuda_stream)
        hc.set_read_length_hint(wl["read_len"])
        d_seq = torch.from_numpy(seq).to(dev)
        d_off = torch.from_numpy(off).to(dev)
        d_out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in wl["outputs"]}
        ptrs = {k: v.data_ptr() for k, v in d_out.items()}
        torch.cuda.synchronize(dev)
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptrs)
            torch.cuda.synchronize(dev)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            t1 = time.perf_counter()
            for a, b in ev:
                a.record(stream)
                hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **ptrs)
                b.record(stream)
            torch.cuda.synchronize(dev)
            elapsed = time.perf_counter() - t1
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        path = hc.kernel_path
        # oracle check on the first `sample` reads (the checker, never the product path)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bdx_oracle as orc

        L = wl["read_len"]
        k = min(n, sample if L <= 1000 else max(2000, sample * 150 // L))
        exp = orc.OracleClassifier(cfg, nthreads=_host_cores(), want_pass=False).classify(seq[:k * L], off[:k + 1])
        for key in wl["outputs"]:
            got = d_out[key][:k].cpu().numpy()
            assert np.array_equal(got, exp[key]), f"bench leg {name}: HIP {key} differs from the oracle on the first {k} reads"
    finally:
        hc.close()
    algo = wl["algo_bytes"]
    achieved = algo * n / (kern_ms * 1e-3) / 1e9
    traffic, tsrc, _ = _profile_counters(name, n)
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_read": algo, "survey_bytes_per_read": wl.get("survey_bytes", algo)}
    if traffic is not None or tsrc is not None:  # (a config without a committed counter profile carries no traffic fields at all)
        roof.update({"traffic": traffic, "traffic_ratio": (traffic / (algo * n)) if traffic else None, "traffic_source": tsrc})
    return {"name": name, "workload": wl["desc"], "value": n * steps / elapsed, "unit": "reads/s", "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "kernel_ms_avg": kern_ms, "kernel_path": path, "roofline": roof,
            "oracle_checked_reads": int(k), "gen_seconds": round(gen_s, 1)}


def run_e2e(n: int, dev_index: int, expect_matched=None):
    """End to end: a FASTQ file of the C2 shape on tmpfs -> execute_demultiplexing (native reader /
    packer, ONE C-ABI classify call per 2^19-read batch, native in-order writer) -> 97 files.  reads/s of the whole
    call, the busy seconds of the three overlapped stages and which of them bounds it."""
    import shutil
    import tempfile

    import numpy as np

    import biodemux_jl_amd as bdx
    from biodemux_jl_amd import synth

    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    need = n * 319 * 2.2
    if base and shutil.disk_usage(base).free < need:
        base = None
    root = tempfile.mkdtemp(prefix="bdx_e2e_", dir=base)
    try:
        bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
        seq, off, _ = synth.make_reads(bcs, n, 150, seed=synth.SEED)
        t0 = time.time()
        rec = np.empty((n, 319), dtype=np.uint8)  # "@read000000000\n" + 150 bases + "\n+\n" + 150 x 'F' + "\n"
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
        del rec, seq, off
        bc = os.path.join(root, "barcodes.csv")
        with open(bc, "w") as f:
            f.write("ID,Full_seq,Full_annotation\n" + "".join(f"bc{i + 1:03d},{b},{'B' * 24}\n" for i, b in enumerate(bcs)))
        write_s = time.time() - t0
        best = None
        all_runs = []
        for rep in range(3):  # (first run: page cache / allocator warm-up, the batch buffers of the pipeline are faulted in; later runs reuse them — nativeio._BUFFER_POOL — as a long-running host would)
            out = os.path.join(root, f"out{rep}")
            tm = {}
            t1 = time.perf_counter()
            st = bdx.execute_demultiplexing(fq, bc, out, max_error_rate=0.1, _io="native", device=dev_index, _timings=tm)
            dt = time.perf_counter() - t1
            nfiles = len(os.listdir(out))
            out_bytes = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
            assert st.total_reads == n and out_bytes == n * 319, (st.total_reads, out_bytes)
            if expect_matched is not None:
                assert st.matched_reads == expect_matched, (st.matched_reads, expect_matched)
            shutil.rmtree(out)
            all_runs.append(round(n / dt))
            if best is None or dt < best["seconds"]:
                stages = {"reader (index + pack)": tm.get("index_s", 0.0) + tm.get("pack_s", 0.0), "classify (bdx_classify_host)": tm.get("classify_s", 0.0),
                          "writer": tm.get("write_s", 0.0)}
                best = {"value": n / dt, "unit": "reads/s", "seconds": dt, "reads": n, "fastq_gb": n * 319 / 1e9, "fastq_gb_per_s": n * 319 / dt / 1e9,
                        "output_files": nfiles, "matched": int(st.matched_reads), "host_threads": tm.get("threads"), "batches": tm.get("batches"),
                        "stage_busy_seconds": {k: round(v, 3) for k, v in stages.items()}, "bound_by": max(stages, key=stages.get),
                        "setup_seconds": round(tm.get("setup_s", 0.0), 3), "pipeline_seconds": round(tm.get("wall_s", 0.0), 3),
                        "close_seconds": round(tm.get("close_s", 0.0), 3),
                        "where": "tmpfs" if base else "tmp dir (no room on /dev/shm)", "generate_fastq_seconds": round(write_s, 1),
                        "note": "execute_demultiplexing(fastq, barcodes.csv, out_dir, max_error_rate=0.1) on the C2 shape: overlapped stages "
                                "(reader thread | classify on the calling thread | writer thread, each with a pool of host threads for its batch); setup = barcode table + device context; "
                                "output bytes = input bytes, file count and matched reads checked; best of three runs (the first faults the pipeline's batch buffers in, the others reuse them)"}
        best["runs_reads_per_s"] = all_runs  # (every run, in order: the figure above is the best of them)
        return best
    finally:
        shutil.rmtree(root, ignore_errors=True)



def main():
    args = _parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))

    import numpy as np
    import torch

    import biodemux_jl_amd as bdx
    from biodemux_jl_amd import dist as bdist
    from biodemux_jl_amd import synth

    rank, local_rank, world = bdist.init_process_group()
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    dev_index = local_rank if world > 1 else 0
    if os.environ.get("BDX_FORCE_DEVICE"):  # rehearsal of N > 1 on a 1-GPU box (with BDX_DIST_BACKEND=gloo)
        dev_index = int(os.environ["BDX_FORCE_DEVICE"])
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    t0 = time.time()
    per_rank = args.reads or {"C5": 400_000}.get(args.config, 10_000_000)
    wl = build_workload(args.config, per_rank, bdist.shard_first_read(rank, per_rank), args.max_error_rate)
    gen_s = time.time() - t0
    cfg, seq, off, n = wl["cfg"], wl["seq"], wl["off"], wl["n"]

    hc = bdx.HipClassifier(cfg, device=dev_index, filter=args.filter)
    # a dedicated (non-default) HIP stream: the kernels, the timing events, the counter memset and the
    # all-reduce all live on it (handle 0 would mean "library's own stream" to bdx_set_stream)
    stream = torch.cuda.Stream(dev)
    hc.set_stream(stream.cuda_stream)
    # the batch's read length is known (synthetic, fixed length): skips the per-batch max-length measurement
    # (one tiny kernel + a 4-byte copy that synchronises the stream, ~0.1 ms)
    hc.set_read_length_hint(wl["read_len"])
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = {k: torch.empty(n, dtype=torch.int32, device=dev) for k in wl["outputs"]}
    d_counts = torch.zeros(hc.counts_len, dtype=torch.int64, device=dev)
    hc.set_counts_buffer(d_counts.data_ptr())
    torch.cuda.synchronize(dev)

    # merge_stats across GPUs: the C-ABI's own RCCL communicator (what a Julia / C host would use); if that
    # cannot be set up on this node the counters fall back to torch.distributed's all_reduce — either way one
    # all-reduce (sum, int64) per step, and the line says which one ran.
    allreduce_via = "none (1 rank)"
    if world > 1:
        try:
            if os.environ.get("BDX_DIST_BACKEND") == "gloo":
                raise RuntimeError("CPU rehearsal: no RCCL communicator")
            bdist.init_abi_comm(hc, rank, world)
            allreduce_via = "C-ABI bdx_allreduce_counts (RCCL)"
        except Exception as e:  # noqa: BLE001
            allreduce_via = f"torch.distributed all_reduce ({type(e).__name__}: {str(e)[:80]})"
        flags = [None] * world
        torch.distributed.all_gather_object(flags, allreduce_via.startswith("C-ABI"))
        if not all(flags) and allreduce_via.startswith("C-ABI"):
            hc.comm_destroy()
            allreduce_via = "torch.distributed all_reduce (another rank could not open the C-ABI communicator)"

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    total = None
    out_ptrs = {k: v.data_ptr() for k, v in d_out.items()}

    def step(ev=None):
        nonlocal total
        with torch.cuda.stream(stream):
            d_counts.zero_()
            if ev:
                ev[0].record(stream)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, **out_ptrs)
            if ev:
                ev[1].record(stream)
            if allreduce_via.startswith("C-ABI"):
                hc.allreduce_counts()   # enqueued on the same stream; the sum lands in the reduced vector
            elif world > 1:
                total = bdist.allreduce_counts(d_counts)
            else:
                total = d_counts        # one rank: merge_stats over one worker is the identity (reporting.jl:1-9) — no collective, no copy

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    # max over ranks
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        if torch.distributed.get_backend() == "gloo":
            el = el.cpu()
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed_max = float(el.item())
    kern_ms = [a.elapsed_time(b) for a, b in events]          # device time of the classify launch(es)
    kern_ms_avg = float(np.mean(kern_ms)) if kern_ms else float("nan")

    counts = hc.reduced_counts if allreduce_via.startswith("C-ABI") else total.cpu().numpy()
    if not args.allow_wrong_results:  # (only tools/phase_*.sh pass the flag; a stray environment variable disables nothing)
        assert counts[0] == n * world, (counts[0], n * world)
        assert counts[1] + counts[2] + counts[3] == counts[0] and counts[4:].sum() == counts[1], "counter vector inconsistent"

    info = hc.launch_info()
    path = hc.kernel_path

    # ---- host-buffer path (SURVEY §8d "two throughputs"): numpy buffers -> H2D -> kernels -> D2H through
    # bdx_classify_host.  PCIe-inclusive; reported beside the headline, never as `value`. ----
    host_path = None
    if rank == 0 and world == 1 and not args.no_host_path:
        hc.set_counts_buffer(0)
        hc.set_stream(0)
        hc.classify(seq, off)  # sizes the staging buffers
        wu0 = hc.window_uploads
        t1 = time.perf_counter()
        got_host = hc.classify(seq, off)
        hs = time.perf_counter() - t1
        moved = seq.nbytes + off.nbytes + 4 * 4 * n
        for k in wl["outputs"]:
            assert np.array_equal(got_host[k], d_out[k].cpu().numpy()), f"host path and device path disagree on {k}"
        windowed = hc.window_uploads > wu0
        # the same call the way a long-running host makes it: page-locked input buffers, result vectors kept across
        # chunks (fresh pageable result arrays are page-faulted in by the device-to-host copies)
        hs_pinned = None
        try:
            from biodemux_jl_amd import hipabi as _abi
            pseq = _abi.pinned_empty(seq.size, np.uint8)
            pseq[:] = seq
            poff = _abi.pinned_empty(off.size, np.int64)
            poff[:] = off
            pout = {k: _abi.pinned_empty(n, np.int32) for k in ("bc1", "bc2", "keep_start", "keep_end")}
            hc.classify(pseq, poff, out=pout)
            t1 = time.perf_counter()
            hc.classify(pseq, poff, out=pout)
            hs_pinned = time.perf_counter() - t1
            for k in wl["outputs"]:
                assert np.array_equal(pout[k], got_host[k]), f"page-locked host path disagrees on {k}"
            del pseq, poff
        except MemoryError:  # (page-locked memory is a limited resource: the figure is optional)
            hs_pinned = None
        host_path = {"value": n / hs, "unit": "reads/s", "ms": hs * 1e3,
                     "pcie_gb_per_s": None if windowed else moved / hs / 1e9, "window_upload": windowed,
                     "host_gb_per_s": moved / hs / 1e9,
                     "page_locked_reused": (None if hs_pinned is None else
                                            {"value": n / hs_pinned, "unit": "reads/s", "ms": hs_pinned * 1e3,
                                             "note": "same call on page-locked input buffers with result vectors kept across calls"}),
                     "note": "bdx_classify_host on pageable numpy buffers: H2D + kernels + D2H of bc1, bc2, keep_start, "
                             "keep_end (+ the Python wrapper's output allocation); verdicts equal the device-resident run"
                             + ("; window upload: only each read's column window crossed PCIe (host_gb_per_s = whole "
                                "host buffers / time)" if windowed else "")}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bdx_oracle as orc  # the checker / reported CPU baseline — never the product path

        cores = _host_cores()
        L = wl["read_len"]
        oc = orc.OracleClassifier(cfg, nthreads=cores, want_pass=False)
        probe = min(n, max(64, (2000 * 150 // L)) * cores)
        t1 = time.perf_counter()
        exp = oc.classify(seq[:probe * L], off[:probe + 1])
        rate = probe / max(time.perf_counter() - t1, 1e-9)
        sample = int(min(n, max(probe, rate * args.cpu_seconds)))
        oc2 = orc.OracleClassifier(cfg, nthreads=cores, want_pass=False)
        t1 = time.perf_counter()
        exp = oc2.classify(seq[:sample * L], off[:sample + 1])
        cpu_s = time.perf_counter() - t1
        for k in wl["outputs"]:
            got = d_out[k][:sample].cpu().numpy()
            assert np.array_equal(got, exp[k]), f"bench: HIP {k} differs from the oracle on the CPU-baseline sample"
        cpu_model = "unknown"
        try:
            for ln in open("/proc/cpuinfo"):
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": sample / cpu_s, "unit": "reads/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
               "nproc": os.cpu_count(),
               "sample": f"first {sample} reads of the same {args.config} batch, oracle (C restatement of the reference "
                         f"algorithm) on {cores} host threads, {cpu_s:.1f} s; {', '.join(wl['outputs'])} equal the HIP output"}

    # HBM traffic per launch: PMC counters cannot be read from inside the process; use the committed
    # rocprofv3 --pmc summary of this same command (separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    # x2 correction on FETCH_SIZE — MI355X_MICROARCH.md §HBM), produced by tools/summarize_prof.py.
    traffic, traffic_src, valu = (None, None, None)
    if args.max_error_rate is None and args.filter == "auto":
        traffic, traffic_src, valu = _profile_counters(args.config, n)

    # the other BASELINE configs (short legs) and the end-to-end figure ride on the default single-GPU run, so that one
    # driver invocation carries every config; the headline above is unaffected (it has been timed already)
    other, e2e, extra = None, None, None
    matched_headline = int(counts[1]) if world == 1 else None  # (the counter vector is zeroed at the top of every step)
    if rank == 0 and world == 1 and args.config == "C2" and not args.no_other_configs and not args.reads and args.max_error_rate is None:
        del d_seq, d_off
        torch.cuda.empty_cache()
        other, extra = [], []
        for name in ("C2d", "C4", "C5"):
            try:
                other.append(run_leg(name, dev, dev_index))
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001 — a leg that cannot run must not take the headline down
                other.append({"name": name, "error": f"{type(e).__name__}: {str(e)[:200]}"})
        for name in ("C2t5", "C2r60", "C2dual"):  # (not BASELINE configs: common variants of the headline shape)
            try:
                extra.append(run_leg(name, dev, dev_index))
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                extra.append({"name": name, "error": f"{type(e).__name__}: {str(e)[:200]}"})
        if args.e2e_reads > 0:
            try:
                e2e = run_e2e(args.e2e_reads, dev_index, matched_headline if args.e2e_reads == n else None)
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                e2e = {"error": f"{type(e).__name__}: {str(e)[:200]}"}

    # what the communicator itself says about the job (world > 1): its size as RCCL sees it and the library version —
    # the first real multi-GPU run proves its N ranks by this line alone
    comm_info = {}
    if world > 1:
        comm_info["comm_size"] = int(hc.lib.bdx_comm_size(hc.h)) if allreduce_via.startswith("C-ABI") else int(torch.distributed.get_world_size())
        comm_info["comm_size_source"] = "bdx_comm_size (ncclCommCount of the C-ABI communicator)" if allreduce_via.startswith("C-ABI") else "torch.distributed.get_world_size"
        try:
            comm_info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            comm_info["rccl_version"] = f"unavailable ({type(e).__name__})"
        comm_info["counts_total_reads"] = int(counts[0])
    if rank == 0:
        reads_total = n * world * args.steps
        value = reads_total / elapsed_max
        algo = wl["algo_bytes"]
        achieved = (algo * n) / (kern_ms_avg * 1e-3) / 1e9
        line = {
            "metric": METRIC, "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": wl["desc"], "name": args.config,
                       "reads_per_gpu": n, "read_len": wl["read_len"], "barcodes": wl["barcodes"],
                       "barcode_len": wl["barcode_len"], "seed": synth.SEED,
                       "read_length_hint": wl["read_len"],
                       "kernel_path": path, "threads_per_block": info["threads_per_block"],
                       "lds_bytes_per_block": info["lds_bytes_per_block"], "parallelism": f"reads sharded x{world}",
                       "allreduce": allreduce_via, **comm_info,
                       "matched_fraction": float(counts[1]) / max(float(counts[0]), 1.0), "gen_seconds": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per step (all kernels of one classify call)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": algo * n,
                         "kernel_ms_avg": kern_ms_avg, "algorithmic_bytes_per_read": algo,
                         "survey_bytes_per_read": wl.get("survey_bytes", algo),
                         "valu": valu,
                         "note": "integer-VALU / latency bound path (SURVEY F6); HBM fraction reported as asked; "
                                 "kernel_ms_avg = HIP events around all launches of one classify call, on the launch stream"},
            "host_buffer_path": host_path,
            "cpu_baseline": cpu,
            "other_configs": other,
            "extra_configs": extra,
            "e2e": e2e,
        }
        print(json.dumps(line), flush=True)

    hc.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
