#!/usr/bin/env python3
"""Tabulate gpurun_out/ph*/ph_counter_collection.csv written by tools/phase_counters.sh."""
import collections
import csv

names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY",
         "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT"]
print("dbg   ms   " + " ".join(f"{n[3:]:>16s}" for n in names))
for d in (0, 1, 3, 7, 15, 31, 63, 127):
    rows = [r for r in csv.DictReader(open(f"gpurun_out/ph{d}/ph_counter_collection.csv")) if "bdx_bitpar" in r["Kernel_Name"]]
    last = max(int(r["Dispatch_Id"]) for r in rows)
    agg = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(f"{d:3d} {ms:6.3f} " + " ".join(f"{agg[n] / 1e6:16.1f}" for n in names))
