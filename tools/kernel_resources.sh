#!/bin/bash
# tools/kernel_resources.sh [object ...] -- VGPRs, spilled VGPRs / SGPRs, scratch bytes and LDS of every kernel in the gfx950 code objects
# of the library's translation units (default: henjou-renderer_amd/build/hjr_launch_*.o).  No GPU needed.
R=$(cd "$(dirname "$0")/.." && pwd)
LLVM=/opt/rocm/lib/llvm/bin
OBJS=${@:-$R/henjou-renderer_amd/build/hjr_launch_*.o}
T=$(mktemp -d)
for o in $OBJS; do
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $o 2>/dev/null || continue
  $LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co || continue
  echo "== $(basename $o)"
  $LLVM/llvm-readelf --notes $T/k.co | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print("  %-86s vgpr %3s spill %3s sgpr_spill %3s scratch %4s lds %6s" % (g("name")[:86], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
done
rm -rf $T
