#!/bin/bash
# usage: tools/kernel_regs.sh [object ...] — VGPRs, SGPRs, scratch (private segment) and LDS of every kernel in the gfx950 code
# objects of the build (default: all of csrc/build/*.o), read from the code-object metadata notes.
cd "$(dirname "$0")/.." || exit 1
objs=("$@"); [ ${#objs[@]} -eq 0 ] && objs=(biodemux.jl_amd/csrc/build/*.o)
for o in "${objs[@]}"; do
  tmp=$(mktemp -d)
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin="$tmp/fat.bin" "$o" 2>/dev/null || { rm -rf "$tmp"; continue; }
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --input="$tmp/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$tmp/dev.co" 2>/dev/null || { rm -rf "$tmp"; continue; }
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" 2>/dev/null | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in re.split(r"\n\s+- \.agpr_count", txt)[1:]:
    blk = blk.split("amdhsa.target")[0]
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("void (anonymous namespace)::", "").split("(")[0]
    print("%-86s vgpr %3s sgpr %3s scratch %4s lds %6s" % (dem[:86], g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
  rm -rf "$tmp"
done
