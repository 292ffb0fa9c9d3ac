#!/bin/bash
# Builds the host-only sources with AddressSanitizer + UBSan and runs tools/sanitize/host_sanitize.cpp on the fixtures.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SRC=$ROOT/glaze_amd/csrc
OUT=${TMPDIR:-/tmp}/glaze_sanitize
mkdir -p $OUT
g++ -O1 -g -std=c++17 -fsanitize=address,undefined,bounds-strict -fno-omit-frame-pointer -ffp-contract=off -I$SRC -I$ROOT/include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
  $ROOT/tools/sanitize/host_sanitize.cpp $SRC/parser.cpp $SRC/serializer.cpp $SRC/converter.cpp $SRC/bvh_sah.cpp \
  $SRC/codec/xz_dec.cpp $SRC/codec/xz_enc.cpp $SRC/codec/png_dec.cpp $SRC/codec/png_enc.cpp $SRC/codec/jpeg.cpp -lz -lpthread -o $OUT/host_sanitize
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $OUT/host_sanitize $ROOT/tests/golden/mattest.glaze $ROOT/tests/golden/cube.obj $OUT $ROOT/tests/golden/checker.jpg
