#!/usr/bin/env python3
"""Tabulate gpurun_out/ph*/ph_counter_collection.csv written by tools/phase_counters.sh / tools/phase_wave.sh.
usage: phase_table.py [kernel-name substring] [debug values ...]"""
import collections
import csv
import os
import sys

kern = sys.argv[1] if len(sys.argv) > 1 else "bdx_bitpar"
ds = [int(x) for x in sys.argv[2:]] or [0, 1, 3, 7, 15, 31, 63, 127]
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
         "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES"]
print("dbg   ms   " + " ".join(f"{n[3:]:>16s}" for n in names))
for d in ds:
    agg = collections.defaultdict(float)
    ms = 0.0
    for base in (f"gpurun_out/ph{d}", f"gpurun_out/phl{d}"):
        f = f"{base}/ph_counter_collection.csv"
        if not os.path.exists(f):
            continue
        rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if base.endswith(f"ph{d}"):
                    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{d:3d} {ms:6.3f} " + " ".join(f"{agg[n] / 1e6:16.1f}" for n in names))
