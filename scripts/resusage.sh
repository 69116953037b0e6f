#!/bin/bash
# dev helper: per-kernel register / spill / LDS usage of the HIP library
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero \
  \
  -Rpass-analysis=kernel-resource-usage hironaka_amd/csrc/hironaka_hip.hip -o /dev/null 2> /tmp/hk_remarks.txt
awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)} / VGPRs:/ {v=$(NF-1)} / AGPRs:/ {a=$(NF-1)} /SGPRs Spill:/ {ss=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {print name, "vgpr="v, "agpr="a, "sspill="ss, "scratch="sc, "occ="o, "lds="$(NF-1)}' /tmp/hk_remarks.txt | sed 's/_ZN2hk//'
