#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of libhcspmm (preprocess, plan build, all LOI variants, permutation) on
# random graphs -- CPU build only (GPU sanitizers are not available on the pool).   bash tools/host_sanitizer_run.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d)
g++ -O1 -g -fPIC -std=c++17 -ffp-contract=off -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -I"$ROOT/include" -I"$ROOT/hc-spmm_amd/csrc" "$ROOT"/hc-spmm_amd/csrc/{preprocess_host,plan_host,loi_host}.cpp -o "$OUT/libhost_asan.so"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 "$ROOT/tools/host_sanitizer_cases.py" "$OUT/libhost_asan.so"
rm -rf "$OUT"
