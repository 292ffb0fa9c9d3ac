#!/bin/bash
# Builds a tuning variant of the library: tools/build_variant.sh NAME "-DGLZ_REFILL=8 ..."   -> gpurun_out/variants/libglaze_hip_NAME.so
set -e
cd "$(dirname "$0")/../glaze_amd/csrc"
mkdir -p ../../variants build_var
NAME=$1; shift
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I. -I../../include --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
# PATH_FLAGS / RENDER_FLAGS: the per-file flags of the two render translation units (default: what the Makefile gives them; RENDER_FLAGS= builds kernels_render.hip without)
/opt/rocm/bin/hipcc $FLAGS "$@" ${RENDER_FLAGS--mllvm -disable-machine-licm} -c kernels_render.hip -o build_var/kr_$NAME.o &
/opt/rocm/bin/hipcc $FLAGS "$@" ${PATH_FLAGS--mllvm -disable-machine-licm} -c kernels_path.hip -o build_var/kp_$NAME.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC -o ../../variants/libglaze_hip_$NAME.so build/abi.o build/parser.o build/serializer.o build/converter.o build/scene.o build/renderer.o build/bvh_sah.o build/xz_dec.o build/xz_enc.o build/png_dec.o build/png_enc.o build/jpeg.o build/kernels_build.o build_var/kr_$NAME.o build_var/kp_$NAME.o -lz -lpthread -ldl
rm -f build_var/kr_$NAME.o build_var/kp_$NAME.o   # (the snapshot gpurun sends would carry them)
echo built variants/libglaze_hip_$NAME.so
