#!/bin/bash
# Dev only: a copy of the library with the quad kernel's probe hooks (-DHK_QUAD_PROBE: HK_QUAD_CUT / HK_QUAD_WPB read
# from the environment per launch) as build_probe/libhk_probe.so.  The product library never contains them.
set -e
cd "$(dirname "$0")/.."
C=hironaka_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=8 -DHK_QUAD_PROBE"
mkdir -p build_probe
SPECS=${1:-"20_3"}
OBJS=""
for s in 10_3 20_3 20_4 50_4; do
  if [[ " $SPECS " == *" $s "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS -DHK_SPEC_M=${s%_*} -DHK_SPEC_D=${s#*_} -c $C/hk_quad_spec.hip -o build_probe/quad_$s.o &
    OBJS="$OBJS build_probe/quad_$s.o"
  else
    OBJS="$OBJS $C/build/quad_$s.o"
  fi
done
wait
# the two-lane kernel's time-line probe (HK_DUO_PROBE: per-wave time stamps over game_length_out) at (20,3)
/opt/rocm/bin/hipcc ${FLAGS/-DHK_QUAD_PROBE/-DHK_DUO_PROBE} -DHK_SPEC_M=20 -DHK_SPEC_D=3 -c $C/hk_duo_spec.hip -o build_probe/duo_20_3.o
OTHERS=$(ls $C/build/*.o | grep -v "/quad_" | grep -v duo_20_3)
OBJS="$OBJS build_probe/duo_20_3.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS $OBJS -o build_probe/libhk_probe.so
ls -la build_probe/libhk_probe.so
