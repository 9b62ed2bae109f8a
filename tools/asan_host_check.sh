#!/bin/bash
# SURVEY 5 "debug build with sanitizers": the library's HOST code under AddressSanitizer (make ASAN=1 -> libirsde_hip_asan.so; device code is not
# instrumented — GPU ASan needs xnack+ code objects, refused on this pool) driven by the host-only C-ABI program tests/c/cabi_host.c (engine creation of the
# three network kinds, weight inventories, error paths) and by the Python host-logic tests that load the library.  CPU only.
#   usage: bash tools/asan_host_check.sh [out.txt]
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/dev/stdout}
CLANG=/opt/rocm/lib/llvm/bin/clang
make -C "$REPO/image_restoration_sde_amd/csrc" ASAN=1 -j4 > /tmp/asan_build.log 2>&1 || { tail -20 /tmp/asan_build.log; exit 1; }
RT=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so)
T=$(mktemp -d)
$CLANG -std=c99 -g -fsanitize=address -shared-libsan -I"$REPO/include" "$REPO/tests/c/cabi_host.c" -o "$T/cabi_host" \
    "$REPO/image_restoration_sde_amd/libirsde_hip_asan.so" -Wl,-rpath,"$REPO/image_restoration_sde_amd" -Wl,-rpath,"$(dirname "$RT")" || exit 1
{
  echo "# make ASAN=1 + tests/c/cabi_host.c under AddressSanitizer ($(date -u +%F)); detect_leaks=1"
  ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 "$T/cabi_host"; echo "cabi_host exit code $?"
  echo "# tests/test_cabi.py + tests/test_host_logic.py with the ASan library loaded (LD_PRELOAD of the runtime; python itself is not instrumented: leaks off)"
  IRSDE_LIB_PATH="$REPO/image_restoration_sde_amd/libirsde_hip_asan.so" LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 \
      python -m pytest "$REPO/tests/test_cabi.py" "$REPO/tests/test_host_logic.py" -q -x -m "not gpu" -k "not graft_entry and not plain_c_host" 2>&1 | tail -5
} > "$OUT" 2>&1
rm -rf "$T"
