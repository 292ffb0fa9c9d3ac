#!/bin/bash
# Builds a tuning variant whose flag reaches the host side and the scene-build kernels too (tools/build_variant.sh only recompiles the two
# render translation units): tools/build_variant_full.sh NAME "-DGLZ_NODE48 ..."   -> variants/libglaze_hip_NAME.so
set -e
cd "$(dirname "$0")/../glaze_amd/csrc"
mkdir -p ../../variants build_var
NAME=$1; shift
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I. -I../../include"
HIPFLAGS="$COMMON --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero"
/opt/rocm/bin/hipcc $HIPFLAGS "$@" -mllvm -disable-machine-licm -c kernels_render.hip -o build_var/kr_$NAME.o &
/opt/rocm/bin/hipcc $HIPFLAGS "$@" -mllvm -disable-machine-licm -c kernels_path.hip -o build_var/kp_$NAME.o &
/opt/rocm/bin/hipcc $HIPFLAGS "$@" -c kernels_build.hip -o build_var/kb_$NAME.o &
g++ $COMMON -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ "$@" -c scene.cpp -o build_var/sc_$NAME.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC -o ../../variants/libglaze_hip_$NAME.so build/abi.o build/parser.o build/serializer.o build/converter.o build_var/sc_$NAME.o build/renderer.o build/bvh_sah.o build/xz_dec.o build/xz_enc.o build/png_dec.o build/png_enc.o build/jpeg.o build_var/kb_$NAME.o build_var/kr_$NAME.o build_var/kp_$NAME.o -lz -lpthread -ldl
rm -f build_var/kr_$NAME.o build_var/kp_$NAME.o build_var/kb_$NAME.o build_var/sc_$NAME.o
echo built variants/libglaze_hip_$NAME.so
