"""Dev probe: a handful of fused 20-step rollouts for rocprofv3 --pmc passes.
usage: pmc_rollout.py m d batch [flagname]   (flagname: four | two | one | team | none)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops

m, d, b = (int(x) for x in sys.argv[1:4])
force = {"four": A.HK_FLAG_FORCE_FOUR_LANES, "two": A.HK_FLAG_FORCE_TWO_LANES, "one": A.HK_FLAG_FORCE_ONE_LANE,
         "team": A.HK_FLAG_FORCE_TEAM, "none": 0}[sys.argv[4] if len(sys.argv) > 4 else "none"]
P = ops.generate_points(b, m, d, 20, seed=42)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
for rep in range(30):
    ops.rollout(Q, 20, 7, initial=P, flags=force, defer_counts=True, workspace=ws)
torch.cuda.synchronize()
