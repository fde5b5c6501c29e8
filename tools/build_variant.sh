#!/bin/bash
# Builds a variant of libhcspmm.so for an A/B run on one box: tools/build_variant.sh NAME "-DMACRO=.. -DMACRO2=.."
# -> _ab_libs/NAME.so (git-ignored; selected at run time with HCSPMM_LIB=_ab_libs/NAME.so)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
TMP=$(mktemp -d)
cp -r "$ROOT/hc-spmm_amd/csrc" "$TMP/csrc"
mkdir -p "$TMP/include" && cp "$ROOT"/include/*.h "$TMP/include/"
rm -f "$TMP"/csrc/*.o "$TMP"/csrc/*.so
make -C "$TMP/csrc" -j"${JOBS:-8}" ROOT="$TMP" EXTRA="$*" libhcspmm.so > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/_ab_libs" && cp "$TMP/csrc/libhcspmm.so" "$ROOT/_ab_libs/$NAME.so"
rm -rf "$TMP"
echo "built _ab_libs/$NAME.so ($*)"
