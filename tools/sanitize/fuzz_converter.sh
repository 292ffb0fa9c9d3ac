#!/bin/bash
# Damaged Wavefront files (tools/sanitize/mutate_obj.py) through the converter built with AddressSanitizer + UBSan:
#   tools/sanitize/fuzz_converter.sh [files per seed] [seeds]
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SRC=$ROOT/glaze_amd/csrc
OUT=${TMPDIR:-/tmp}/glaze_sanitize
mkdir -p $OUT/obj
g++ -O1 -g -std=c++17 -fsanitize=address,undefined,bounds-strict -fno-omit-frame-pointer -ffp-contract=off -I$SRC -I$ROOT/include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
  $ROOT/tools/sanitize/convert_many.cpp $SRC/converter.cpp $SRC/parser.cpp $SRC/serializer.cpp $SRC/codec/xz_dec.cpp $SRC/codec/xz_enc.cpp $SRC/codec/png_dec.cpp \
  $SRC/codec/png_enc.cpp $SRC/codec/jpeg.cpp -lz -lpthread -o $OUT/convert_many
cp $ROOT/tests/golden/checker.jpg $OUT/obj/ 2>/dev/null || true
N=${1:-1000}
for seed in $(seq 1 ${2:-3}); do
  rm -f $OUT/obj/m*.obj $OUT/obj/m*.mtl
  python3 $ROOT/tools/sanitize/mutate_obj.py $seed $N $OUT/obj
  ls $OUT/obj/m*.obj | ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 xargs $OUT/convert_many $OUT/obj/out.glaze
done
