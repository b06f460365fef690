#!/usr/bin/env python3
"""bench.py — headline benchmark: reads/s demultiplexed (150 bp, 96 barcodes, :semiglobal).

Workload (BASELINE.json configs[1], "C2"): 10 M synthetic 150 bp reads x 96 barcodes of 24 bp,
:semiglobal, max_error_rate 0.1 (allowed_error = floor(0.1*24) = 2), min_delta 0, costs 0/1/1,
all ranges "1:end", no trim (ScoreOnly).  Seed 20260515.

A "step" = one pass of the hot path over one batch: the packed batch is ALREADY resident in
HBM when the timed region starts; the step classifies every read of the rank's shard through
the C-ABI (bdx_classify_device) and all-reduces the per-barcode counters (RCCL when N > 1).
Weak scaling: every rank owns its own 10 M-read shard of the global synthetic stream.

Run:  python bench.py [--gpus N --steps K --warmup W]      (N > 1: under torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALGO_BYTES_PER_READ = 150 + 8 + 4   # n bases + one int64 offset + one int32 verdict
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
METRIC = "reads/s demultiplexed (150 bp, 96 barcodes, :semiglobal) at 1/2/4/8 MI355X"


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (C2 = 10 M)")
    ap.add_argument("--filter", default="auto", help="auto|off|qgram|bitpar (all give identical results)")
    ap.add_argument("--max-error-rate", type=float, default=0.1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    args = ap.parse_args()

    import torch

    import biodemux_jl_amd as bdx
    from biodemux_jl_amd import dist as bdist
    from biodemux_jl_amd import synth

    rank, local_rank, world = bdist.init_process_group()
    assert world == args.gpus or world == 1 and args.gpus == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    dev_index = local_rank if world > 1 else 0
    if os.environ.get("BDX_FORCE_DEVICE"):  # rehearsal of N > 1 on a 1-GPU box (with BDX_DIST_BACKEND=gloo)
        dev_index = int(os.environ["BDX_FORCE_DEVICE"])
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    n = args.reads
    bcs = synth.make_barcodes(96, 24, seed=synth.SEED)
    t0 = time.time()
    seq, off, _truth = synth.make_reads(bcs, n, 150, seed=synth.SEED, first_read=bdist.shard_first_read(rank, n))
    gen_s = time.time() - t0
    cfg = bdx.DemuxConfig(bc_seqs=bcs, bc_lengths_no_N=[24] * 96, ids=[f"bc{i + 1:03d}" for i in range(96)],
                          max_error_rate=args.max_error_rate, min_delta=0.0, match=0, mismatch=1, indel=1,
                          matching_algorithm="semiglobal")

    hc = bdx.HipClassifier(cfg, device=dev_index, filter=args.filter)
    # a dedicated (non-default) HIP stream: the kernels, the timing events and the counter
    # memset all live on it (handle 0 would mean "library's own stream" to bdx_set_stream)
    stream = torch.cuda.Stream(dev)
    hc.set_stream(stream.cuda_stream)
    hc.set_read_length_hint(150)  # C2 reads are 150 bp: skips the per-batch max-length measurement (a sync)
    d_seq = torch.from_numpy(seq).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_bc1 = torch.empty(n, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(hc.counts_len, dtype=torch.int64, device=dev)
    hc.set_counts_buffer(d_counts.data_ptr())
    torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    total = None

    def step(ev=None):
        nonlocal total
        with torch.cuda.stream(stream):
            d_counts.zero_()
            if ev:
                ev[0].record(stream)
            hc.classify_device(d_seq.data_ptr(), d_off.data_ptr(), n, bc1=d_bc1.data_ptr())
            if ev:
                ev[1].record(stream)
            total = bdist.allreduce_counts(d_counts)  # merge_stats across GPUs: one RCCL all-reduce

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

    counts = total.cpu().numpy()
    # (BDX_DEBUG: the kernel's phase-skip flags of tools/phase_counters.sh — timing experiments, no results)
    assert counts[0] == n * world or os.environ.get("BDX_DEBUG"), (counts[0], n * world)

    # correctness spot check inside the bench: a strided sample vs the oracle (not timed)
    info = hc.launch_info()
    path = hc.kernel_path

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import bdx_oracle as orc  # the checker / reported CPU baseline — never the product path

        cores = _host_cores()
        oc = orc.OracleClassifier(cfg, nthreads=cores, want_pass=False)
        probe = min(n, 2000 * cores)
        t1 = time.perf_counter()
        exp = oc.classify(seq[:probe * 150], off[:probe + 1])
        rate = probe / max(time.perf_counter() - t1, 1e-9)
        sample = int(min(n, max(probe, rate * args.cpu_seconds)))
        oc2 = orc.OracleClassifier(cfg, nthreads=cores, want_pass=False)
        t1 = time.perf_counter()
        exp = oc2.classify(seq[:sample * 150], off[:sample + 1])
        cpu_s = time.perf_counter() - t1
        got = d_bc1[:sample].cpu().numpy()
        assert np.array_equal(got, exp["bc1"]), "bench: HIP verdicts differ from the oracle on the CPU-baseline sample"
        cpu = {"value": sample / cpu_s, "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {sample} reads of the same C2 batch, oracle (C restatement of the reference "
                         f"algorithm) on {cores} host threads, {cpu_s:.1f} s; verdicts equal the HIP output"}

    # HBM traffic per launch: PMC counters cannot be read from inside the process; use the committed
    # rocprofv3 --pmc summary of this same command (separate FETCH_SIZE / WRITE_SIZE passes, gfx950
    # x2 correction on FETCH_SIZE — MI355X_MICROARCH.md §HBM), produced by tools/summarize_prof.py.
    traffic, traffic_src = None, None
    valu = None  # from the same committed profile: what actually bounds the kernel
    try:
        prof = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_final_pmc.json"))
        if prof and n == 10_000_000 and path.startswith("qgram"):
            pj = json.load(open(os.path.join(ROOT, "profiles", prof[-1])))
            for kname, e in pj["kernels"].items():
                if "bitpar" in kname and "hbm_read_bytes_corrected" in e:
                    traffic = e["hbm_read_bytes_corrected"] + e.get("hbm_write_bytes", 0.0)
                    traffic_src = f"profiles/{prof[-1]} ({kname})"
                    c = e.get("counters_per_dispatch", {})
                    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
                        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1024 SIMDs, SQ_BUSY_CYCLES cycles
                        # summed over the 32 shader engines
                        valu = {"valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["SQ_BUSY_CYCLES"] / 32), 3),
                                "valu_wave_instr_per_read": round(c.get("SQ_INSTS_VALU", 0.0) / n, 1),
                                "wave_issue_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}
    except Exception:
        pass

    if rank == 0:
        reads_total = n * world * args.steps
        value = reads_total / elapsed_max
        achieved = (ALGO_BYTES_PER_READ * n) / (kern_ms_avg * 1e-3) / 1e9
        line = {
            "metric": METRIC, "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "C2: 10 M synthetic 150 bp reads x 96 barcodes (24 bp), :semiglobal, "
                                   f"max_error_rate={args.max_error_rate} (allowed_error=2), min_delta=0, costs 0/1/1, ScoreOnly",
                       "reads_per_gpu": n, "read_len": 150, "barcodes": 96, "barcode_len": 24, "seed": synth.SEED,
                       "kernel_path": path, "threads_per_block": info["threads_per_block"],
                       "lds_bytes_per_block": info["lds_bytes_per_block"], "parallelism": f"reads sharded x{world}",
                       "matched_fraction": float(counts[1]) / max(float(counts[0]), 1.0), "gen_seconds": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_READ * n,
                         "kernel_ms_avg": kern_ms_avg, "algorithmic_bytes_per_read": ALGO_BYTES_PER_READ,
                         "valu": valu,
                         "note": "integer-VALU / latency bound path (SURVEY F6); HBM fraction reported as asked"},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)

    hc.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
