#!/bin/bash
# Run ON THE GPU BOX: PMC passes (separate runs, --kernel-trace only) of a small python command.
# usage: scripts/pmc_run.sh <outdir-under-gpurun_out> <python script + args...>
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i exit $?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70] + " grid=" + r["Grid_Size"]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/summary.txt", "w") as fo:
    for k, cs in agg.items():
        n = len(cs["dur_ns"])
        if n < 10: continue
        fo.write(f"{k}  dispatch-samples={n}\n")
        waves = sum(cs.get("SQ_WAVES", [1])) / max(1, len(cs.get("SQ_WAVES", [1])))
        for c, v in sorted(cs.items()):
            mean = sum(v) / len(v)
            fo.write(f"    {c:24s} {mean:14.1f}   per wave {mean / waves:10.1f}\n")
print(open(out + "/summary.txt").read())
PY
