#!/bin/bash
# CPU-side AddressSanitizer + UBSan run of the host code (the C-ABI library's host half, the C++ mirror, the threaded SAH builder, the
# host tracer): the library is rebuilt with -fsanitize=address,undefined for the HOST compilation only (-fno-gpu-sanitize; GPU ASAN is
# not available on this pool; `asan_host.sh <dir> thread` builds with ThreadSanitizer instead), then tests/host/host_test.cpp (cpu mode) and tests/host/sanitize_driver.cpp run against it.  No GPU needed.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
KIND=${2:-address}                     # "address" (ASan + UBSan, leak check) or "thread" (TSan: the builder's and the converter's threads)
OUT=${1:-/tmp/ntr_san_$KIND}
mkdir -p "$OUT"
R=$ROOT/ntrace_amd
INC="-I$ROOT/include -I$R/csrc -I$R/host"
if [ "$KIND" = thread ]; then SAN="-fsanitize=thread -fno-omit-frame-pointer"; else SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"; fi
OBJS=""
for f in $R/csrc/*.hip $R/csrc/*.cpp $R/host/*.cpp $R/host/bvh/*.cpp; do
  o=$OUT/$(echo "$f" | tr '/' '_').o
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $SAN -fno-gpu-sanitize $INC -x hip -c "$f" -o "$o"
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $SAN -fno-gpu-sanitize -shared -fPIC -o "$OUT/libntrace_amd.so" $OBJS
CXX=/opt/rocm/lib/llvm/bin/clang++
$CXX -O1 -g -std=c++17 -ffp-contract=off $SAN $INC "$ROOT/tests/host/host_test.cpp" -o "$OUT/host_test_asan" -L"$OUT" -lntrace_amd -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib
$CXX -O1 -g -std=c++17 $SAN -I"$ROOT/include" "$ROOT/tests/host/sanitize_driver.cpp" -o "$OUT/sanitize_driver" -L"$OUT" -lntrace_amd -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib
export ASAN_OPTIONS=detect_leaks=1
"$OUT/host_test_asan" cpu
"$OUT/sanitize_driver" 300000
"$OUT/sanitize_driver" 700000      # large enough for the threaded sweeps of the nodes near the root
echo "asan_host ($KIND): clean"
