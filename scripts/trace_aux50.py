"""(50,4) x 262 144: the observation-feature sorts, Zeillinger's class and -- for comparison, same states -- one hk_step
(sparse states) and a device copy; under rocprofv3 (kernel trace / PMC passes: scripts/pmc_run.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
b, m, d = 262144, 50, 4
P = ops.generate_points(b, m, d, 20, seed=42)
out = torch.empty((b, m * d), device="cuda")
nxt = torch.empty_like(P)
cls = torch.randint(0, 2 ** d - d - 1, (b,), device="cuda", dtype=torch.int32)
ax = torch.randint(0, d, (b,), device="cuda", dtype=torch.int32)
for _ in range(12):
    ops.get_features(P, out=out)
    ops.get_features_torch(P)
    ops.zeillinger(P)
    ops.step(P, cls, ax, stages=7, out=nxt)
    nxt.copy_(P)
torch.cuda.synchronize()
