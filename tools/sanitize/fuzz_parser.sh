#!/bin/bash
# Mutated `.glaze` files (bytes inside a chunk's body changed, the chunk's XXH64 prefix made right again, so the xz / PNG decoders see
# them) through the parser built with AddressSanitizer + UBSan:  tools/sanitize/fuzz_parser.sh [files per seed] [seeds]
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SRC=$ROOT/glaze_amd/csrc
OUT=${TMPDIR:-/tmp}/glaze_sanitize
mkdir -p $OUT/mut
g++ -O1 -g -std=c++17 -fsanitize=address,undefined,bounds-strict -fno-omit-frame-pointer -ffp-contract=off -I$SRC -I$ROOT/include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
  $ROOT/tools/sanitize/parse_many.cpp $SRC/parser.cpp $SRC/serializer.cpp $SRC/codec/xz_dec.cpp $SRC/codec/xz_enc.cpp $SRC/codec/png_dec.cpp $SRC/codec/png_enc.cpp -lz -lpthread -o $OUT/parse_many
N=${1:-1500}
for seed in $(seq 1 ${2:-3}); do
  rm -f $OUT/mut/*.glaze
  python3 $ROOT/tools/sanitize/mutate_glaze.py $seed $N $OUT/mut
  ls $OUT/mut/*.glaze | ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 xargs $OUT/parse_many
done
