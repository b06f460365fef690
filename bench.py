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
    ap.add_argument("--config", default="C2", choices=["C2", "C2d", "C4", "C5"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the config's size)")
    ap.add_argument("--filter", default="auto", help="auto|off|qgram|bitpar (all give identical results)")
    ap.add_argument("--max-error-rate", type=float, default=None, help="override the config's rate")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-buffer (PCIe-inclusive) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
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
            else:
                total = bdist.allreduce_counts(d_counts)

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
    if not os.environ.get("BDX_DEBUG"):  # (phase-skip timing experiments of a -DBDX_TUNING build give wrong results by design)
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
    traffic, traffic_src = None, None
    valu = None  # from the same committed profile, per kernel: what actually bounds the path
    try:
        tagged = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles"))
                        if f.endswith("_pmc.json") and f"_{args.config}_" in f)
        if tagged and args.max_error_rate is None and args.filter == "auto":
            pj = json.load(open(os.path.join(ROOT, "profiles", tagged[-1])))
            if pj.get("reads_per_launch") == n:
                tot, src = 0.0, []
                for kname, e in pj["kernels"].items():
                    if "hbm_read_bytes_corrected" in e:
                        tot += e["hbm_read_bytes_corrected"] + e.get("hbm_write_bytes", 0.0)
                        src.append(kname)
                    c = e.get("counters_per_step", {})
                    if c.get("SQ_INSTS_VALU", 0.0) / n >= 1.0 and c.get("SQ_WAVE_CYCLES"):
                        # VALU wave-instructions per read of the batch (all of the kernel's launches in one step) and the
                        # fraction of its waves' lifetime spent issuing; peak issue = 1024 SIMDs x 1 wave-instruction / 2 cycles
                        valu = (valu or []) + [{
                            "kernel": kname, "launches_per_step": e.get("meta", {}).get("dispatches_per_step"),
                            "valu_wave_instr_per_read": round(c.get("SQ_INSTS_VALU", 0.0) / n, 1),
                            # SIMD cycles per issued VALU wave-instruction (SQ_BUSY_CYCLES: 32 counter instances, 1024
                            # SIMDs); the pipe's limit for this instruction mix is ~3.2 (tools/ubench_valu.hip: 2.6-2.8
                            # for v_and/or/xor/add and 3-source v_bitop3, 3.9-4.4 for shifts, v_add3, carries)
                            "simd_cycles_per_valu_instr": (round(c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0 / c["SQ_INSTS_VALU"], 2)
                                                           if c.get("SQ_BUSY_CYCLES") else None),
                            "valu_ceiling_cycles_per_instr": 3.2,
                            "wave_issue_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3)}]
                if src:
                    traffic, traffic_src = tot, f"profiles/{tagged[-1]} ({' + '.join(src)})"
    except Exception:
        pass

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
                       "allreduce": allreduce_via,
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
        }
        print(json.dumps(line), flush=True)

    hc.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
