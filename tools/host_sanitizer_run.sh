#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of libhcspmm (preprocess, plan build, all LOI variants incl. the relaxed parallel one, permutation) on
# random graphs -- CPU build only (GPU sanitizers are not available on the pool).   bash tools/host_sanitizer_run.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$(mktemp -d)
g++ -O1 -g -fPIC -std=c++17 -ffp-contract=off -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared \
    -I"$ROOT/include" -I"$ROOT/hc-spmm_amd/csrc" "$ROOT"/hc-spmm_amd/csrc/{preprocess_host,plan_host,loi_host,loi_fast_host}.cpp -o "$OUT/libhost_asan.so"
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python3 "$ROOT/tools/host_sanitizer_cases.py" "$OUT/libhost_asan.so"
# ThreadSanitizer over the threaded passes (relaxed LOI: reservation rounds, placement bitmap, list cursors; parallel permutation;
# window pass and the plan passes on their call-local pool, on the graphs large enough to use threads)
g++ -O1 -g -fPIC -std=c++17 -ffp-contract=off -pthread -fsanitize=thread -fno-omit-frame-pointer -shared \
    -I"$ROOT/include" -I"$ROOT/hc-spmm_amd/csrc" "$ROOT"/hc-spmm_amd/csrc/{preprocess_host,plan_host,loi_host,loi_fast_host}.cpp -o "$OUT/libhost_tsan.so"
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0" \
    python3 "$ROOT/tools/host_sanitizer_cases.py" "$OUT/libhost_tsan.so" tsan
rm -rf "$OUT"
