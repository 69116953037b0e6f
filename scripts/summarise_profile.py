#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into <outdir>/<tag>_* (default outdir: profiles/):
rocprofv3's kernel stats, per-kernel durations split by launch shape, a PMC table per (kernel, launch shape), and
the per-launch constants bench.py reports (HBM bytes, VALU instructions, share of HIP kernels in the search).

    python scripts/summarise_profile.py r02 [outdir]
"""
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
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# the launches bench.py's headline numbers come from (kernel name pattern, grid size in threads)
ROLLOUT = (re.compile(r"duo_kernel<20, 3, 1, 1(, false)*>"), "131072")   # fused rollout, 65 536 games, two lanes per game
STEP = (re.compile(r"quad_kernel<20, 3, 1, \d, 1(, false)*>"), "262144")  # hk_step (JAX-trainer configuration, f32 mask + i32 axis)
STEP3 = (re.compile(r"quad_kernel<50, 4, 1, \d, 1(, false)*>"), "1048576")  # hk_step at (50,4) x 262 144
ROLL3 = (re.compile(r"quadroll_kernel<50, 4, 1, 1(, false)*>"), "1048576")  # fused rollout at (50,4) x 262 144
# one in-kernel-policy step of 65 536 games (rollouts of <= 6 steps run on the four-lane kernel since round 3)
SINGLE = (re.compile(r"quadroll_kernel<20, 3, 1, 4(, false)*>"), "262144")
# round 4: the generator on four lanes, the compute_rho loop as one launch (initial states drawn in the kernel, counts
# only: GEN), episodes back to back from resident states (EPI)
GENERATE = (re.compile(r"quadgen_kernel<20, 3, 4>"), "262144")
GENERATE3 = (re.compile(r"quadgen_kernel<50, 4, 4>"), "1048576")
RHO = (re.compile(r"quadroll_kernel<20, 3, 1, 4, false, false, true, true>"), "262144")
RHO3 = (re.compile(r"quadroll_kernel<50, 4, 1, 1, false, false, true, true>"), "1048576")
EPISODES = (re.compile(r"quadroll_kernel<20, 3, 1, 4, false, false, false, true>"), "262144")


def short(name):
    return name.split("(")[0].replace("void ", "")[:80]


def split_rollout(durs):
    """dispatches of the rollout kernel at one grid size: fused 20-step episodes vs single steps.  Two clusters are
    told apart at the geometric mean of the 10th and 90th percentile (an outlier dispatch must not move the cut);
    one cluster (p90 < 1.6 x p10) is all episodes or all steps -- episodes if at least 15 ns per game."""
    v = np.asarray(durs, dtype=float)
    lo, hi = np.percentile(v, 10), np.percentile(v, 90)
    if hi < 1.6 * lo:
        whole = np.ones(len(v), bool)
        return (whole, ~whole) if np.median(v) > 15e3 else (~whole, whole)
    cut = float(np.sqrt(lo * hi))
    return v > cut, v <= cut


def first(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


stats = first("trace/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bj = os.path.join(src, "bench_trace.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(dst, f"{tag}_bench_under_rocprof.json"))

# ---- per-dispatch durations from the kernel trace, per (kernel, grid size) --------------------------------------
lines = []
trace = first("trace/**/*kernel_trace.csv")
if trace:
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        by[(r["Kernel_Name"], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines.append("kernel,grid_threads,calls,mean_us,p50_us,p95_us")
    for (k, grid), v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        v = np.array(v)
        is_rollout = re.search(r"(fast|duo)_kernel<\d+, \d+, 1, \d(, false)*>", k) is not None
        parts = (("T=20 episodes", None), ("T=1 steps", None)) if is_rollout else (("", None),)
        if is_rollout:
            big, small = split_rollout(v)
            parts = (("T=20 episodes", big), ("T=1 steps", small))
        else:
            parts = (("", np.ones(len(v), bool)),)
        for name, sel in parts:
            if sel.any():
                w = v[sel]
                label = f"{short(k)} [{name}]" if name else short(k)
                lines.append(f"\"{label}\",{grid},{len(w)},{w.mean()/1e3:.3f},{np.median(w)/1e3:.3f},"
                             f"{np.percentile(w, 95)/1e3:.3f}")
    open(os.path.join(dst, f"{tag}_kernel_durations.csv"), "w").write("\n".join(lines) + "\n")

# ---- PMC passes: means per dispatch and per wave for every launch shape with enough samples ------------------------
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    per_dispatch = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        e = per_dispatch[r["Dispatch_Id"]]
        e["key"] = (r["Kernel_Name"], r["Grid_Size"])
        e[r["Counter_Name"]] = float(r["Counter_Value"])
        e["duration_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    groups = collections.defaultdict(list)
    for e in per_dispatch.values():
        groups[e["key"]].append(e)
    for (k, grid), rows in groups.items():
        if ROLLOUT[0].search(k):
            big, small = split_rollout([e["duration_ns"] for e in rows])
            sets = ((f"{short(k)} [T=20]", [e for e, s in zip(rows, big) if s]),
                    (f"{short(k)} [T=1]", [e for e, s in zip(rows, small) if s]))
        else:
            sets = ((short(k), rows),)
        for label, sub in sets:
            for e in sub:
                for c, v in e.items():
                    if c != "key":
                        agg[(label, grid)][c].append(v)

pmc_rows = ["# rocprofv3 --pmc passes (own runs, --kernel-trace only) of bench.py; means per dispatch and per wave for every",
            "# launch shape with >= 20 dispatches.  FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies 64 B per",
            "# 128-B request: read bytes = 2 x FETCH_SIZE KiB (MI355X_MICROARCH.md, HBM).  SQ_* cycle counters are quad-cycles.",
            "kernel,grid_threads,dispatches,counter,per_dispatch,per_wave"]
summary = {}
for (label, grid), cs in sorted(agg.items()):
    n = len(cs["duration_ns"])
    if n < 20:
        continue
    waves = int(grid) // 64
    for c in sorted(cs):
        mean = float(np.mean(cs[c]))
        pmc_rows.append(f"\"{label}\",{grid},{len(cs[c])},{c},{mean:.6g},{mean / waves:.6g}")
        summary[(label, grid, c)] = mean
open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w").write("\n".join(pmc_rows) + "\n")


def find(pat_grid, suffix, counter):
    pat, grid = pat_grid
    for (label, g, c), v in summary.items():
        if g == grid and c == counter and pat.search(label) and label.endswith(suffix):
            return v
    return None


def traffic(pat_grid, suffix=""):
    f, w = find(pat_grid, suffix, "FETCH_SIZE"), find(pat_grid, suffix, "WRITE_SIZE")
    return None if f is None or w is None else int((2 * f + w) * 1024)


out = {"source": f"rocprofv3 --pmc passes (separate runs), profiles/{tag}_pmc_summary.csv",
       "correction": "gfx950: read bytes = 2 x FETCH_SIZE (64 B tallied per 128-B request); WRITE_SIZE exact",
       "batch": 65536, "max_points": 20, "dim": 3,
       "rollout_T20_bytes_per_launch": traffic(ROLLOUT, "[T=20]"),
       "rollout_T20_valu_insts_per_launch": find(ROLLOUT, "[T=20]", "SQ_INSTS_VALU"),
       "single_step_bytes_per_launch": traffic(ROLLOUT, "[T=1]") or traffic(SINGLE),
       "boundary_step_bytes_per_launch": traffic(STEP),
       "boundary_step_valu_insts_per_launch": find(STEP, "", "SQ_INSTS_VALU"),
       "boundary_step_wait_any_frac": (find(STEP, "", "SQ_WAIT_ANY") or 0) / (find(STEP, "", "SQ_WAVE_CYCLES") or 1),
       "config3_step_bytes_per_launch": traffic(STEP3),
       "config3_rollout_bytes_per_launch": traffic(ROLL3),
       "config3_rollout_valu_insts_per_launch": find(ROLL3, "", "SQ_INSTS_VALU"),
       "generate_bytes_per_launch": traffic(GENERATE),
       "generate_valu_insts_per_launch": find(GENERATE, "", "SQ_INSTS_VALU"),
       "config3_generate_bytes_per_launch": traffic(GENERATE3),
       "rho_loop_bytes_per_launch": traffic(RHO),
       "rho_loop_valu_insts_per_launch": find(RHO, "", "SQ_INSTS_VALU"),
       "config3_rho_loop_bytes_per_launch": traffic(RHO3),
       "persistent_episodes_bytes_per_launch": traffic(EPISODES)}

# ---- share of the GPU time of the search workload spent in this package's HIP kernels ----------------------------
sstats = first("search/**/*kernel_stats.csv")
if sstats:
    shutil.copy(sstats, os.path.join(dst, f"{tag}_search_kernel_stats.csv"))
    tot = ours = 0.0
    for r in csv.DictReader(open(sstats)):
        t = float(r["TotalDurationNs"])
        tot += t
        if "hk::" in r["Name"]:
            ours += t
    if tot:
        out["search_hip_kernel_share"] = ours / tot
out = {k: v for k, v in out.items() if v is not None}
json.dump(out, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
print("\n".join(lines[:16]))
