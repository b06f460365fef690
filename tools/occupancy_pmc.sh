#!/bin/bash
# usage (GPU box): tools/occupancy_pmc.sh — mean resident waves of the wave kernel under the geometries of tools/occupancy_probe.sh
# (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES per launch, relative to the product geometry)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
A="--no-cpu-baseline --no-host-path --no-other-configs --e2e-reads 0 --steps 2 --warmup 1"
L6=$PWD/tests/redgreen/libbiodemux_hip_occ6.so
one() {
  tag=$1; shift
  mkdir -p gpurun_out/occ_$tag
  "$@" rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d gpurun_out/occ_$tag -o p --output-format csv -- python bench.py $A > gpurun_out/occ_$tag/run.log 2>&1 || { echo "$tag failed"; tail -3 gpurun_out/occ_$tag/run.log; return 1; }
  python - "$tag" <<'P'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(float); nd = set()
for f in glob.glob(f"gpurun_out/occ_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bdx_wave_kernel" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); nd.add(r["Dispatch_Id"])
n = max(len(nd), 1)
print("%-28s launches %d  waves %.0f  wave_cycles/busy_cycles %.2f  wave_cycles/gui_active %.1f  valu/launch %.3g" % (
    tag, n, agg["SQ_WAVES"] / n, agg["SQ_WAVE_CYCLES"] / max(agg["SQ_BUSY_CYCLES"], 1), agg["SQ_WAVE_CYCLES"] / max(agg["GRBM_GUI_ACTIVE"], 1), agg["SQ_INSTS_VALU"] / n))
P
}
one product_1x16 env || exit 1
one product_rw32_2x8 env BDX_WAVE_RW=32 BDX_WAVE_WAVES=8 || exit 1
one occ6_rw16_3x8 env BDX_LIB_PATH=$L6 BDX_WAVE_RW=16 BDX_WAVE_WAVES=8 BDX_WAVE_MAXRES=24 || exit 1
