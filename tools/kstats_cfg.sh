#!/bin/bash
# usage (GPU box): tools/kstats_cfg.sh <tag> <ONLY pattern> [N] — kernel-trace summary of one case of tools/bench_configs.py
tag=${1:?tag}; only=${2:?pattern}; n=${3:-2000000}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
export ONLY="$only" N=$n
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o p --output-format csv -- python tools/bench_configs.py > gpurun_out/$tag/run.log 2>&1 || exit 1
tail -2 gpurun_out/$tag/run.log
python - "$tag" <<'P'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.3: continue
    name = r["Name"].replace("void (anonymous namespace)::", "")[:70]
    print("%-70s calls %4s avg %9.1f us  %s%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
