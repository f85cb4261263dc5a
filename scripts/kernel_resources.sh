#!/bin/bash
# kernel_resources.sh [file.hip ...]: registers, scratch and LDS of every kernel of the trace / LBVH translation units as hipcc compiles
# them for gfx950 with the product's flags (no GPU needed) -- the numbers DESIGN.md quotes (waves per SIMD = min(8, 512 / ceil8(VGPRs))).
cd "$(dirname "$0")/../ntrace_amd/csrc" || exit 1
FILES=${@:-trace_kernels.hip lbvh_kernels.hip sched_kernels.hip}
TMP=$(mktemp -d)
# the Makefile's own CXXFLAGS / KERNELFLAGS (target `resources`), so the numbers cannot drift from the build
make -s resources RES_DIR="$TMP" RES_FILES="$FILES" || exit 1
for F in $FILES; do
  cp "$TMP/kr_$F.s" /tmp/kr_$$.s
  echo "== $F"
  grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name|vgpr_spill_count):" /tmp/kr_$$.s | paste - - - - - - |
    sed 's/ \+/ /g;s/_segment_fixed_size//g;s/_count//g' | while read -r L; do
      N=$(echo "$L" | sed -n 's/.*\.name: \([^ \t]*\).*/\1/p' | c++filt | sed 's/(ntr::TraceParams)//;s/ntr:://')
      V=$(echo "$L" | sed -n 's/.*\.vgpr: \([0-9]*\).*/\1/p'); S=$(echo "$L" | sed -n 's/.*\.sgpr: \([0-9]*\).*/\1/p')
      P=$(echo "$L" | sed -n 's/.*\.private: \([0-9]*\).*/\1/p'); G=$(echo "$L" | sed -n 's/.*\.group: \([0-9]*\).*/\1/p')
      SP=$(echo "$L" | sed -n 's/.*\.vgpr_spill: \([0-9]*\).*/\1/p')
      A=$(( (V + 7) / 8 * 8 )); W=$(( 512 / A )); [ $W -gt 8 ] && W=8
      printf "%-62s vgpr %3s (waves/SIMD %d) sgpr %3s scratch %4s B lds %5s B spills %s\n" "$N" "$V" "$W" "$S" "$P" "$G" "$SP"
    done
  rm -f /tmp/kr_$$.s
done
rm -rf "$TMP"
