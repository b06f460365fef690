#!/bin/bash
# needs a library built with -DBDX_TUNING; per-phase attribution of the wave kernel's PAIRS mode (phase-skip bits << 8)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 3 11 43 107; do
  BDX_DEBUG=$((d * 256)) rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d gpurun_out/ph$d -o ph --output-format csv -- python bench.py --allow-wrong-results --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --reads ${READS:-4000000} $BENCH_ARGS > gpurun_out/ph$d.log 2>&1 || exit 1
  BDX_DEBUG=$((d * 256)) rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/phl$d -o ph --output-format csv -- python bench.py --allow-wrong-results --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --reads ${READS:-4000000} $BENCH_ARGS > gpurun_out/phl$d.log 2>&1 || exit 1
  echo "done $d"
done
python tools/phase_table.py "${KERN:-false, 4, 3, false>}" 0 1 3 11 43 107
