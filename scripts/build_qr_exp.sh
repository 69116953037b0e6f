#!/bin/bash
# Dev only: copies of the library whose four-lane recording rollout at (20,3) leaves out one part (HK_QR_EXP: 1 = no
# stages, the observation stores only; 2 = no observation stores; 4 = ordinary instead of non-temporal stores) as
# build_probe/libqr_exp<N>.so -- where a recording episode's time goes.  The product library never defines HK_QR_EXP.
set -e
cd "$(dirname "$0")/.."
C=hironaka_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=8"
mkdir -p build_probe
OTHERS=$(ls $C/build/*.o | grep -v quadroll_20_3)
for e in ${1:-1 2}; do
  /opt/rocm/bin/hipcc $FLAGS -DHK_QR_EXP=$e -DHK_SPEC_M=20 -DHK_SPEC_D=3 -c $C/hk_quadroll_spec.hip -o build_probe/quadroll_20_3_exp$e.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS build_probe/quadroll_20_3_exp$e.o -o build_probe/libqr_exp$e.so
done
ls -la build_probe/libqr_exp*.so
