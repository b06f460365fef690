#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag> [bench.py args, e.g. --config C4]
# The official bench line, the rocprofv3 kernel-trace summary of the same command and the HBM
# traffic counters (separate --pmc passes), all under gpurun_out/<tag>_*; afterwards, here:
#   python tools/summarize_prof.py <tag> <reads per launch> gpurun_out/<tag>_stats gpurun_out/<tag>_fetch gpurun_out/<tag>_write gpurun_out/<tag>_sq
tag=${1:?tag}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
# (the profiled runs carry ONE config: the short legs of the default run launch the same kernel instantiations on other batches)
P="--no-cpu-baseline --no-host-path --no-other-configs --e2e-reads 0"
python bench.py "$@" > gpurun_out/${tag}_bench.json.log 2> gpurun_out/${tag}_bench.err || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats -o p --output-format csv -- python bench.py $P "$@" > gpurun_out/${tag}_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${tag}_fetch -o p --output-format csv -- python bench.py $P --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${tag}_write -o p --output-format csv -- python bench.py $P --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/${tag}_sq -o p --output-format csv -- python bench.py $P --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_sq.log 2>&1 || exit 1
cat gpurun_out/${tag}_bench.json.log
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM -d gpurun_out/${tag}_sq2 -o p --output-format csv -- python bench.py $P --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_sq2.log 2>&1 || exit 1
