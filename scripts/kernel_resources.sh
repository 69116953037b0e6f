#!/bin/bash
# VGPR / scratch / occupancy of one spec TU:  scripts/kernel_resources.sh hk_quadroll_spec.hip 50 4
here=$(dirname "$0")/../hironaka_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
  -mllvm -amdgpu-kernarg-preload-count=8 -Rpass-analysis=kernel-resource-usage \
  -DHK_SPEC_M=$2 -DHK_SPEC_D=$3 -c $here/$1 -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/\[-Rpass.*//' | paste - - - - |
  sed -E 's/Function Name: //; s/ +/ /g' | c++filt | sed -E 's/\(float const\*.*\)//; s/\[[^]]*\]//'
