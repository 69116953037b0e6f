#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into profiles/<tag>_*:
kernel stats CSV, per-kernel durations split by launch shape, a PMC table per kernel variant, and the
measured HBM traffic per launch that bench.py reports as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
HEADLINE_GRID = "131072"  # 2048 waves x 64 lanes = 65 536 games, two lanes each (BASELINE configs[1])
HEADLINE_KERNEL = re.compile(r"duo_kernel<20, 3, 1, 1>")  # hk::duo_kernel<20,3,rollout,kHotJax>
STEP_KERNEL = re.compile(r"duo_kernel<20, 3, 0, 1>")      # hk::duo_kernel<20,3,step,jax> (hk_step as bench.py calls it)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_kernel_stats.csv"))
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench_under_rocprof.json"))


def split_rollout(durs):
    """dispatches of the rollout kernel at one grid size: fused 20-step episodes vs single steps"""
    v = np.asarray(durs)
    cut = 0.6 * v.max()
    return v > cut, v <= cut


# per-dispatch durations from the kernel trace, per (kernel, grid size)
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
lines = []
if trace:
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        by[(r["Kernel_Name"], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines.append("kernel,grid_threads,calls,mean_us,p50_us,p95_us")
    for (k, grid), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = np.array(v)
        if ("fast_kernel<" in k or "duo_kernel<" in k) and k.split("<")[1].split(",")[2].strip(" >") == "1":  # rollout mode
            big, small = split_rollout(v)
            parts = (("T=20 episodes", big), ("T=1 steps", small))
        else:
            parts = (("", np.ones(len(v), bool)),)
        for name, sel in parts:
            if sel.any():
                w = v[sel]
                label = f"{k} [{name}]" if name else k
                lines.append(f"\"{label}\",{grid},{len(w)},{w.mean()/1e3:.3f},{np.median(w)/1e3:.3f},"
                             f"{np.percentile(w, 95)/1e3:.3f}")
    open(os.path.join(dst, f"{tag}_kernel_durations.csv"), "w").write("\n".join(lines) + "\n")

pmc_rows = ["# rocprofv3 --pmc passes (own runs, --kernel-trace only) of bench.py --steps 2000; means per dispatch at",
            "# the headline launch shape (2048 waves x 32 games): rollout = hk::duo_kernel<20,3,rollout,jax> with 20 steps",
            "# (T20) or one step (T1) per launch, step = hk::duo_kernel<20,3,step> (hk_step).  FETCH_SIZE / WRITE_SIZE in",
            "# KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request: read bytes = 2 x FETCH_SIZE KiB",
            "# (MI355X_MICROARCH.md, HBM).",
            "counter,rollout_T20,rollout_T20_per_wave,rollout_T1,rollout_T1_per_wave,step,step_per_wave"]
traffic = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    roll = collections.defaultdict(dict)
    step = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if r["Grid_Size"] != HEADLINE_GRID:
            continue
        if HEADLINE_KERNEL.search(r["Kernel_Name"]):  # the JAX-configuration rollout kernel only
            e = roll[r["Dispatch_Id"]]
        elif STEP_KERNEL.search(r["Kernel_Name"]):
            e = step[r["Dispatch_Id"]]
        else:
            continue
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["duration_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not roll:
        continue
    vals = list(roll.values())
    big, small = split_rollout([e["duration_ns"] for e in vals])
    L = [e for e, s in zip(vals, big) if s]
    S = [e for e, s in zip(vals, small) if s]
    P = list(step.values())
    mean = lambda rows, k: float(np.mean([e[k] for e in rows])) if rows else float("nan")
    for k in sorted(L[0]):
        a, b, c = mean(L, k), mean(S, k), mean(P, k)
        pmc_rows.append(f"{k},{a:.6g},{a/2048:.6g},{b:.6g},{b/2048:.6g},{c:.6g},{c/2048:.6g}")
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic[k] = {"T20_KiB": a, "T1_KiB": b, "step_KiB": c}
open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w").write("\n".join(pmc_rows) + "\n")
if "FETCH_SIZE" in traffic and "WRITE_SIZE" in traffic:
    total = lambda key: int((2 * traffic["FETCH_SIZE"][key] + traffic["WRITE_SIZE"][key]) * 1024)
    out = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/{tag}_pmc_summary.csv",
           "correction": "gfx950: read bytes = 2 x FETCH_SIZE (64 B tallied per 128-B request); WRITE_SIZE exact",
           "rollout_T20_bytes_per_launch": total("T20_KiB"),
           "single_step_bytes_per_launch": total("T1_KiB"),
           "batch": 65536, "max_points": 20, "dim": 3}
    if not np.isnan(traffic["FETCH_SIZE"]["step_KiB"]):
        out["boundary_step_bytes_per_launch"] = total("step_KiB")
    json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
    print(out)
print("\n".join(lines[:14]))
