import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
b, m, d = 262144, 50, 4
P = ops.generate_points(b, m, d, 20, seed=42)
out = torch.empty((b, m * d), device="cuda")
for _ in range(5):
    ops.get_features(P, out=out)
    ops.get_features_torch(P)
    ops.zeillinger(P)
torch.cuda.synchronize()
