#!/bin/bash
# usage (GPU box): tools/kstats.sh <tag> [bench.py args] — kernel-trace summary of a short bench run, printed
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o p --output-format csv -- python bench.py --no-cpu-baseline --no-host-path --no-other-configs --steps 10 --warmup 2 "$@" > gpurun_out/$tag/run.log 2>&1 || exit 1
python - "$tag" <<'P'
import csv, glob, sys, re
f = glob.glob(f"gpurun_out/{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) < 0.3: continue
    name = r["Name"].replace("void (anonymous namespace)::", "")[:70]
    print("%-70s calls %4s avg %9.1f us  %s%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
