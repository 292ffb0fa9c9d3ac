#!/bin/bash
# Disassembles the gfx950 code object inside a hipcc-built object file: tools/isa_dump.sh glaze_amd/csrc/build/kernels_path.o out.s
# (the .o is a fat binary: llvm-objdump --offloading extracts the device ELF next to it first)
set -e
OBJ=$(readlink -f "$1"); OUT=$(readlink -f "$2")
TMP=$(mktemp -d)
cp "$OBJ" "$TMP/in.o"
(cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.o > /dev/null)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 "$TMP"/in.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 > "$OUT"
rm -rf "$TMP"
