#!/bin/bash
# needs a library built with -DBDX_TUNING (HIPCC_EXTRA=-DBDX_TUNING python -c "import __graft_entry__ as g; g.build_hip(force=True)"): the product build has no phase-skip switches
# per-phase instruction attribution: cumulative phase-skip flags + SQ counters (developer tool)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 3 7 15 31 63 127; do
  BDX_DEBUG=$d rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/ph$d -o ph --output-format csv -- python bench.py --allow-wrong-results --steps 1 --warmup 1 --no-cpu-baseline --reads ${READS:-4000000} $BENCH_ARGS > gpurun_out/ph$d.log 2>&1 || exit 1
  echo "done $d"
done
