#!/bin/bash
# dev helper: per-kernel register / scratch / occupancy table of every HIP kernel (all specialisations)
cd "$(dirname "$0")/.."
make -C hironaka_amd/csrc -s resources 2> /tmp/hk_remarks.txt  # serial: ~10 minutes
awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)} / VGPRs:/ {v=$(NF-1)} / AGPRs:/ {a=$(NF-1)} /SGPRs Spill:/ {ss=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {o=$(NF-1)} /LDS Size/ {print name, "vgpr="v, "agpr="a, "sspill="ss, "scratch="sc, "occ="o, "lds="$(NF-1)}' /tmp/hk_remarks.txt | sed 's/_ZN2hk//' 
