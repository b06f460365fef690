#!/bin/bash
# Evidence for the regression test of fuzz seed 53109 (commit 5cb01f9: a dual config with one split pass and one
# score-only pass handed the exact kernel window counts for entries nobody had written).
#   here (no GPU needed):   tools/red_green_53109.sh build     -> tests/redgreen/libbiodemux_hip_red.so
#   on the GPU box:         tools/red_green_53109.sh run       -> gpurun_out/red_green_53109.txt
# The red library is HEAD with exactly that fix taken out again (-DBDX_REVERT_5CB01F9).  It is NOT the tree at
# 5cb01f9^: that tree has no window validation, so with poisoned hand-over buffers its exact kernel would follow a
# wild column into a GPU memory fault — on a shared host that takes other users' work down.  With HEAD's validation
# the defect shows up deterministically as refused windows (bdx_rejected_windows > 0), which the test-suite's
# per-test check (tests/conftest.py) turns into a failure.
set -u
cd "$(dirname "$0")/.."
T=test_dual_with_one_known_score_pass_hands_over_clean_window_counts
if [ "${1:-}" = "build" ]; then
  mkdir -p tests/redgreen
  hipcc=/opt/rocm/bin/hipcc
  flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -pthread"
  cd biodemux.jl_amd/csrc
  $hipcc $flags -DBDX_REVERT_5CB01F9 -c bdx_bitpar.hip -o build/bdx_bitpar.red.o 2>/dev/null || exit 1
  $hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../tests/redgreen/libbiodemux_hip_red.so build/bdx_abi.cpp.o build/bdx_comm.cpp.o \
      build/bdx_device.hip.o build/bdx_wave.hip.o build/bdx_bitpar.red.o || exit 1
  echo "built tests/redgreen/libbiodemux_hip_red.so"
else
  out=gpurun_out/red_green_53109.txt
  mkdir -p gpurun_out
  {
    echo "== RED: HEAD minus the fix of 5cb01f9 (tests/redgreen/libbiodemux_hip_red.so), BDX_POISON on =="
    BDX_LIB_PATH=$PWD/tests/redgreen/libbiodemux_hip_red.so python -m pytest tests/test_gpu_parity.py -q -k $T 2>&1 | tail -15
    echo
    echo "== GREEN: HEAD (biodemux.jl_amd/csrc/libbiodemux_hip.so), BDX_POISON on =="
    python -m pytest tests/test_gpu_parity.py -q -k $T 2>&1 | tail -5
    echo
    echo "== the same test on HEAD without the poison switch (what a lucky allocator gives) =="
    BDX_TEST_NO_POISON=1 python -m pytest tests/test_gpu_parity.py -q -k $T 2>&1 | tail -3
    echo
    echo "== RED library without the poison switch (the round-2 record: passes unless the allocator hands back dirty memory) =="
    BDX_TEST_NO_POISON=1 BDX_LIB_PATH=$PWD/tests/redgreen/libbiodemux_hip_red.so python -m pytest tests/test_gpu_parity.py -q -k $T 2>&1 | tail -3
  } > $out
  cat $out
fi
