import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = 65536, 20, 3
P = ops.generate_points(b, m, d, 20, seed=42)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
for T in (1, 2, 3, 4, 6, 8, 12):
    out = []
    for name, fl in (("default", 0), ("four", A.HK_FLAG_FORCE_FOUR_LANES), ("two", A.HK_FLAG_FORCE_TWO_LANES), ("one", A.HK_FLAG_FORCE_ONE_LANE)):
        def ep():
            for _ in range(5):
                ops.rollout(Q, T, 1, initial=P, defer_counts=True, workspace=ws, flags=fl)
        out.append(f"{name} {timed(ep) / 5 * 1e6:6.2f}")
    print(f"T={T:2d}: " + "  ".join(out), flush=True)
