"""(50,4) x 262 144: the fused 20-step rollout (generated order, binned order) under rocprofv3 (scripts/pmc_run.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
b, m, d = 262144, 50, 4
P = ops.generate_points(b, m, d, 20, seed=42)
B, ids = ops.bin_by_live_rows(P)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
for _ in range(12):
    ops.rollout(Q, 20, 1, initial=P, defer_counts=True, workspace=ws)
    ops.rollout(Q, 20, 1, initial=B, game_ids=ids, defer_counts=True, workspace=ws)
torch.cuda.synchronize()
