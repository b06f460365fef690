#!/usr/bin/env python3
"""Collect rocprofv3 outputs from gpurun_out/ into profiles/<tag>_*.{csv,json} (tracked).
usage: summarize_prof.py <tag> <reads per launch> <stats_dir> [<pmc_dir> ...]
<tag> should carry the bench config between underscores (r02_C2_final, r02_C4_...): bench.py looks for
profiles/*_<config>_*_pmc.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, reads, stats_dir, pmc_dirs = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4:]
os.makedirs("profiles", exist_ok=True)
ks = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha16: the identity of the kernels these counters were collected on — run this
              # script on the tree the profile was taken from, before touching csrc/ again)
summary = {"tag": tag, "reads_per_launch": reads, "kernel_source_sha16": bench.kernel_source_sha16(), "kernels": {}}
for d in pmc_dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        meta = {}
        for r in csv.DictReader(open(f)):
            if "bdx_" not in r["Kernel_Name"] or "maxlen" in r["Kernel_Name"]:
                continue
            import re
            k = re.search(r"bdx_\w+(<[^>]*>)?", r["Kernel_Name"]).group(0)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            meta[k] = {"last_dispatch_ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6,
                       "vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "grid": int(r["Grid_Size"]),
                       "workgroup": int(r["Workgroup_Size"])}
        if not agg:
            continue
        # one classify call ("step") may launch a kernel more than once (tiers, list mode): counters are summed per
        # STEP = all of a kernel's dispatches divided by the number of steps (the kernel launched least often runs once a step)
        steps = min(len(v) for v in disp.values())
        for k in agg:
            e = summary["kernels"].setdefault(k, {"counters_per_step": {}, "meta": meta[k]})
            e["counters_per_step"].update({c: v / steps for c, v in agg[k].items()})
            e["meta"]["dispatches_per_step"] = len(disp[k]) / steps
for k, e in summary["kernels"].items():
    c = e["counters_per_step"]
    if "FETCH_SIZE" in c:  # KB; gfx950: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream (MI355X_MICROARCH §HBM)
        e["hbm_read_bytes_corrected"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
json.dump(summary, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(summary, indent=1)[:1500])
