#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py and a kernel trace of
# the search workload; raw CSVs under gpurun_out/, then scripts/summarise_profile.py condenses them into profiles/.
# usage: scripts/profile_round.sh r04
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the driver's own command line: --steps 20 --warmup 5)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-search --no-overlap --no-binned > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
echo "trace exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/search" -- python3 "$ROOT/scripts/profile_search.py" > "$OUT/search.log" 2>&1
echo "search trace exit $?"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d" " -f1)
  HK_BENCH_TIME_SCALE=0.05 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$tag" -- python3 "$ROOT/bench.py" --steps 400 --warmup 40 --no-cpu-baseline --no-search --no-overlap --no-binned > /dev/null 2> "$OUT/pmc_$tag.err"
  echo "pmc $tag exit $?"
done
# condense on the box (the raw counter CSVs can exceed what gpurun merges back), keep the small files
python3 "$ROOT/scripts/summarise_profile.py" "$TAG" "$ROOT/gpurun_out/profiles_$TAG" > "$OUT/summary.log" 2>&1
echo "summarise exit $?"
find "$OUT" -name "*counter_collection.csv" -size +4M -delete
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
tail -30 "$OUT/summary.log"
