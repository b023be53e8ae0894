#!/usr/bin/env bash
# CPU sanitizer job (the reference has none; the oracle's index arithmetic DEFINES correctness here, so it gets one).
#   1. oracle/isx_oracle.c built with AddressSanitizer + UndefinedBehaviorSanitizer (gcc) and every CPU test that drives it run against
#      that build (golden vectors, sharded AP, drop-in surface);
#   2. the HOST halves of libisx's launchers (argument validation, workspace arithmetic, error strings) built with the same sanitizers
#      on the host side only (hipcc -fsanitize=address,undefined -fno-gpu-sanitize: device code is the ordinary gfx950 code and is
#      never launched here) and the ABI tests run against that library.
# CPU ONLY -- never run on the GPU box (GPU ASan / xnack+ builds are refused there).  Exits non-zero on any sanitizer report.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
OUT="$ROOT/build/sanitize"
mkdir -p "$OUT"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"          # CPython itself leaks by design; everything else is fatal
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
export PYTHONMALLOC=malloc

echo "== 1. oracle under ASan + UBSan (gcc)"
gcc -O1 -g -fPIC -std=c99 -ffp-contract=off -fno-fast-math -Wall -Wextra -fsanitize=address,undefined -fno-sanitize-recover=all \
    -fno-omit-frame-pointer -shared -fvisibility=hidden -o "$OUT/libisx_oracle_san.so" "$ROOT/oracle/isx_oracle.c" -lm
( cd "$ROOT" && LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ISX_ORACLE_LIB="$OUT/libisx_oracle_san.so" \
    python -m pytest -q -x -m "not gpu" -p no:cacheprovider tests/test_oracle_golden.py tests/test_sharded_ap.py tests/test_dropin_cpu.py )

echo "== 2. libisx host halves under ASan + UBSan (clang, host side only)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CSRC="$ROOT/instance-search_amd/csrc"
OBJS=()
for f in "$CSRC"/*.hip "$CSRC"/api.cpp "$CSRC"/comm.cpp; do
    o="$OUT/$(basename "${f%.*}").o"
    "$HIPCC" --offload-arch=gfx950 -x hip -O1 -g -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-function \
        -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=all -fno-omit-frame-pointer -c "$f" -o "$o" &
    OBJS+=("$o")
    while [ "$(jobs -r | wc -l)" -ge 6 ]; do wait -n; done
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$OUT/libisx_host_san.so" "${OBJS[@]}" -ldl
RT="$(dirname "$("$HIPCC" -print-file-name=libclang_rt.asan-x86_64.so)")"
( cd "$ROOT" && LD_PRELOAD="$RT/libclang_rt.asan-x86_64.so" ISX_LIB="$OUT/libisx_host_san.so" \
    python -m pytest -q -x -m "not gpu" -p no:cacheprovider tests/test_abi.py )
echo "sanitize_cpu: clean"
