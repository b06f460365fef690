#!/bin/bash
# usage (GPU box): tools/pmc_bench.sh <tag> [bench.py args] — SQ counters per kernel of a short bench.py run (per read of the batch)
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$tag
A="--no-cpu-baseline --no-host-path --no-other-configs --e2e-reads 0 --steps 2 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d gpurun_out/$tag/a -o p --output-format csv -- python bench.py $A "$@" > gpurun_out/$tag/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY -d gpurun_out/$tag/b -o p --output-format csv -- python bench.py $A "$@" > gpurun_out/$tag/b.log 2>&1 || exit 1
python - "$tag" <<'P'
import csv, glob, sys, collections, re, json
tag = sys.argv[1]
n = 10_000_000
try:
    n = json.loads(open(f"gpurun_out/{tag}/a.log").read().strip().splitlines()[-1])["config"]["reads_per_gpu"]
except Exception:
    pass
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(float)
for f in glob.glob(f"gpurun_out/{tag}/*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "bdx_" not in r["Kernel_Name"]: continue
        k = re.search(r"bdx_\w+(<[^>]*>)?", r["Kernel_Name"]).group(0)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, f)].add(r["Dispatch_Id"])
for k, c in agg.items():
    nd = max(len(v) for (kk, f), v in disp.items() if kk == k)
    v = c.get("SQ_INSTS_VALU", 0) / nd
    if v / n < 0.5: continue
    busy = c.get("SQ_BUSY_CYCLES", 0) / nd / 2
    print("%-62s x%d valu/read %7.1f cyc/valu %5.2f issue %.2f lds/read %5.1f salu/read %5.1f vmem/read %5.2f lds-conflict %.2f" % (
        k[:62], nd, v / n, busy / 32 * 1024 / max(v, 1), c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
        c.get("SQ_INSTS_LDS", 0) / nd / n, c.get("SQ_INSTS_SALU", 0) / nd / n, c.get("SQ_INSTS_VMEM", 0) / nd / n,
        c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
P
