#!/bin/bash
# Run ON THE GPU BOX (through gpurun): same-box A/B of the headline between the product library and alternative builds.
# Boxes differ by a few per cent, runs on one box by ~0.5 %: a variant is only ever compared with the library built from
# HEAD in alternation on the SAME box.
#   build:  compile the variant's objects into hironaka_amd/csrc/alt_<name>.so (not tracked; it travels with gpurun)
#   usage:  scripts/ab_headline.sh <rounds> <name> [<name> ...]      ("base" is always the first leg of a round)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=$ROOT/hironaka_amd/csrc
ROUNDS=$1; shift
B="python $ROOT/bench.py --steps 20 --warmup 5 --no-search --no-cpu-baseline --no-single-step"
cp $C/libhironaka_hip.so /tmp/hk_base.so
for i in $(seq 1 $ROUNDS); do
  for v in base "$@"; do
    if [ "$v" = base ]; then cp /tmp/hk_base.so $C/libhironaka_hip.so; else cp $C/alt_$v.so $C/libhironaka_hip.so; fi
    $B > $ROOT/gpurun_out/ab_${v}_$i.json 2> /dev/null
    echo "$v $(python $ROOT/scripts/bench_summary.py $ROOT/gpurun_out/ab_${v}_$i.json | head -1)"
  done
done
cp /tmp/hk_base.so $C/libhironaka_hip.so
