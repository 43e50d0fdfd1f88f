#!/bin/bash
# Build a DIAGNOSTIC copy of the library with extra hipcc flags into gpurun_out/diag/<tag>/libos2d_hip.so (run on the GPU
# box; the product library in os2d_amd/lib is untouched).  Usage: tools/diag_build.sh <tag> <flags...>; then run with
# OS2D_HIP_LIB=gpurun_out/diag/<tag>/libos2d_hip.so
TAG=$1; shift
OUT=gpurun_out/diag/$TAG
mkdir -p $OUT/obj
for f in abi prep corr_mfma conv_mfma conv_f16x3 conv3_f16x3 corr_f16x3 sample_decode nms detect detect_pyramid spectral spectral_f16 fft; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c os2d_amd/csrc/$f.hip -o $OUT/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libos2d_hip.so $OUT/obj/*.o && rm -rf $OUT/obj && echo built $OUT/libos2d_hip.so
