#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py,
# raw CSVs under gpurun_out/, then scripts/summarise_profile.py condenses them into profiles/.
# usage: scripts/profile_round.sh r01
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace exit $?"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- python3 "$ROOT/bench.py" --steps 2000 --warmup 100 --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$tag.err"
  echo "pmc $tag exit $?"
done
