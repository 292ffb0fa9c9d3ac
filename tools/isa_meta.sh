#!/bin/bash
# Register / scratch / LDS figures of every kernel inside a hipcc-built object file: tools/isa_meta.sh glaze_amd/csrc/build/kernels_render.o
set -e
OBJ=$(readlink -f "$1")
TMP=$(mktemp -d)
cp "$OBJ" "$TMP/in.o"
(cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o > /dev/null)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP"/in.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | grep -E "\.name:|vgpr_count|vgpr_spill|sgpr_spill|private_segment_fixed|group_segment_fixed|agpr_count" | sed 's/^ *//' | paste - - - - - - - | sed 's/\.name: *//' | awk '{print}' | cut -c1-260
rm -rf "$TMP"
