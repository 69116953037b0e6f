#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into profiles/<tag>_*:
kernel stats CSV, a PMC table per kernel variant, and the measured HBM traffic per launch that
bench.py reports as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench_under_rocprof.json"))

# per-dispatch durations of the rollout kernel from the kernel trace (T=20 episodes vs T=1 steps)
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
lines = []
if trace:
    rows = list(csv.DictReader(open(trace[0])))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines.append("kernel,calls,mean_us,p50_us,p95_us")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = np.array(v)
        if "fast_kernel" in k and "1>" in k:  # rollout mode: split fused episodes from single steps
            for name, sel in (("T=20 episodes", v > 0.6 * v.max()), ("T=1 steps", v <= 0.6 * v.max())):
                if sel.any():
                    w = v[sel]
                    lines.append(f"\"{k} [{name}]\",{len(w)},{w.mean()/1e3:.3f},{np.median(w)/1e3:.3f},{np.percentile(w,95)/1e3:.3f}")
        else:
            lines.append(f"\"{k}\",{len(v)},{v.mean()/1e3:.3f},{np.median(v)/1e3:.3f},{np.percentile(v,95)/1e3:.3f}")
    open(os.path.join(dst, f"{tag}_kernel_durations.csv"), "w").write("\n".join(lines) + "\n")

pmc_rows = ["# rocprofv3 --pmc passes (own runs, --kernel-trace only) of bench.py --steps 2000; per dispatch of the",
            "# rollout kernel hk::fast_kernel<20,3,rollout> (1024 waves x 64 games).  FETCH_SIZE / WRITE_SIZE in KiB;",
            "# on gfx950 FETCH_SIZE tallies 64 B per 128-B request: read bytes = 2 x FETCH_SIZE KiB (MI355X_MICROARCH.md, HBM).",
            "counter,T20_per_dispatch,T20_per_wave,T1_per_dispatch,T1_per_wave"]
traffic = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if "fast_kernel<20, 3, 1>" not in r["Kernel_Name"]:
            continue
        e = per[r["Dispatch_Id"]]
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["duration_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not per:
        continue
    dur = np.array([e["duration_ns"] for e in per.values()])
    cut = 0.6 * dur.max()
    L = [e for e in per.values() if e["duration_ns"] > cut]
    S = [e for e in per.values() if e["duration_ns"] <= cut]
    for k in sorted(L[0]):
        a = np.mean([e[k] for e in L])
        b = np.mean([e[k] for e in S]) if S else float("nan")
        pmc_rows.append(f"{k},{a:.6g},{a/1024:.6g},{b:.6g},{b/1024:.6g}")
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic[k] = {"T20_KiB": a, "T1_KiB": b}
open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w").write("\n".join(pmc_rows) + "\n")
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/{tag}_pmc_summary.csv",
           "correction": "gfx950: read bytes = 2 x FETCH_SIZE (64 B tallied per 128-B request); WRITE_SIZE exact",
           "rollout_T20_bytes_per_launch": int((2 * traffic["FETCH_SIZE"]["T20_KiB"] + traffic["WRITE_SIZE"]["T20_KiB"]) * 1024),
           "single_step_bytes_per_launch": int((2 * traffic["FETCH_SIZE"]["T1_KiB"] + traffic["WRITE_SIZE"]["T1_KiB"]) * 1024),
           "batch": 65536, "max_points": 20, "dim": 3}
    json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    print(out)
print("\n".join(lines[:8]))
