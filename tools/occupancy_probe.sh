#!/bin/bash
# usage (GPU box): tools/occupancy_probe.sh — the occupancy experiment of DESIGN §4 on the headline config: the wave kernel
# compiled for 6 / 5 waves per SIMD (side libraries under tests/redgreen/: bdx_wave.hip compiled with
# -DBDX_WAVE_BOUNDS=__launch_bounds__(512,6) / (640,5), linked with the product objects) against the product build
# (4 per SIMD, 114 VGPRs), each with the geometry that fills its residency.
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-host-path --no-other-configs --e2e-reads 0 --steps 20 --warmup 5"
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-62s %7.1f M reads/s  %.4f ms  block %s threads, %s B LDS, path %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], c.get('threads_per_block'), c.get('lds_bytes_per_block'), c.get('kernel_path')))" "$1"; }
L6=$PWD/tests/redgreen/libbiodemux_hip_occ6.so
L5=$PWD/tests/redgreen/libbiodemux_hip_occ5.so
$B | show "product (4 waves/SIMD, 114 VGPRs): RW 32, 1 x 16 waves" || exit 1
BDX_WAVE_RW=32 BDX_WAVE_WAVES=8 $B | show "product: RW 32, 1 x 8 waves (two do not fit: 8 resident)" || exit 1
BDX_WAVE_RW=16 BDX_WAVE_WAVES=8 $B | show "product: RW 16, 2 x 8 waves (16 resident)" || exit 1
BDX_LIB_PATH=$L6 BDX_WAVE_RW=32 BDX_WAVE_WAVES=8 $B | show "6/SIMD build (80 VGPRs, 116 B scratch): RW 32, 1 x 8 waves" || exit 1
BDX_LIB_PATH=$L6 BDX_WAVE_RW=16 BDX_WAVE_WAVES=8 BDX_WAVE_MAXRES=24 $B | show "6/SIMD build: RW 16, 3 x 8 waves (24 resident)" || exit 1
BDX_LIB_PATH=$L5 BDX_WAVE_RW=32 BDX_WAVE_WAVES=8 $B | show "5/SIMD build (96 VGPRs, 48 B scratch): RW 32, 1 x 8 waves" || exit 1
BDX_LIB_PATH=$L5 BDX_WAVE_RW=16 BDX_WAVE_WAVES=10 BDX_WAVE_MAXRES=20 $B | show "5/SIMD build: RW 16, 2 x 10 waves (20 resident)" || exit 1
