"""Dev probe: per-wave time line of the pool rollout kernel (probe build with HK_POOL_PROBE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libhk_probe.so")
import numpy as np
import torch
from hironaka_amd import ops, _abi as A

b, m, d, T = int(os.environ.get("B", 65536)), 20, 3, 20
P = ops.generate_points(b, m, d, 20, seed=42)
Q = torch.empty_like(P)
for rep in range(3):
    res = ops.rollout(Q, T, 1 + rep, initial=P, record=("game_length",), flags=A.HK_FLAG_FORCE_POOL)
torch.cuda.synchronize()
gl = res["game_length"].cpu().numpy().reshape(-1, 32).astype(np.int64)  # one row per wave
base = gl[:, 0].min()
us = lambda x: ((x - base) & 0xFFFFFFFF) / 100.0
nw = gl.shape[0]
ends = []
# aggregate: for each tag kind, time since the wave's previous stamp
from collections import defaultdict
agg = defaultdict(list)
for w in range(nw):
    prev = None
    for i in range(0, 32, 2):
        if gl[w, i + 1] < 0 and gl[w, i] < 0:
            break
        t, tag = us(gl[w, i]), int(gl[w, i + 1])
        kind = tag & 0xF00
        rnd = tag & 0xF
        key = (kind >> 8, rnd)
        agg[key].append((t, (t - prev) if prev is not None else 0.0, tag))
        prev = t
names = {0: "start", 1: "scanned+barrier", 2: "round end (scatter done)", 3: "sorted+dealt", 4: "loop exit", 5: "stored", 1 + 0: "scanned"}
names = {0: "start", 0x1: "stair done", 0x2: "scattered", 0x3: "sorted+dealt", 0x4: "loop exit", 0x5: "stored"}
print(f"waves {nw}")
for key in sorted(agg, key=lambda k: np.median([x[0] for x in agg[k]])):
    v = agg[key]
    at = np.array([x[0] for x in v]); dt = np.array([x[1] for x in v])
    extra = ""
    if key[0] == 3:
        extra = f" live~{np.median([x[2] >> 12 for x in v]):.0f}"
    if key[0] == 1:
        extra = f" smax~{np.mean([(x[2] >> 4) & 0xF for x in v]):.2f}"
    nm = "start" if key == (0, 0) and np.median(at) < 1 else names.get(key[0], str(key[0]))
    if key == (0, 1):
        nm = "scanned+barrier"
    print(f"{nm:18s} round {key[1]:2d}  n {len(v):5d}  at median {np.median(at):7.2f} max {at.max():7.2f}   since previous: median {np.median(dt):6.2f} p90 {np.percentile(dt,90):6.2f} max {dt.max():6.2f}{extra}")

# by wave index within the workgroup: the time from the last sort to the end of the last staircase
print("by wave of the workgroup: staircase of the last round, since the deal [us] (median / p90 / max), initial tag")
for w8 in range(8):
    durs = []
    for w in range(w8, nw, 8):
        last = None
        for i in range(0, 32, 2):
            if gl[w, i] < 0 and gl[w, i + 1] < 0:
                break
            if (int(gl[w, i + 1]) & 0xF00) == 0x100:
                last = (us(gl[w, i]) - us(gl[w, i - 2]))
        if last is not None:
            durs.append(last)
    if durs:
        print(f"  wave {w8}: n {len(durs):4d}  median {np.median(durs):6.2f}  p90 {np.percentile(durs, 90):6.2f}  max {max(durs):6.2f}")
