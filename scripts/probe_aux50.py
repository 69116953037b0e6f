"""Dev probe: the other operators at (50,4) x 262144 -- observation features, the torch container's features, Zeillinger's
class -- per kernel family."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = 262144, 50, 4
P = ops.generate_points(b, m, d, 20, seed=42)
dense = ops.generate_points(b, m, d, 20, seed=43, newton=False, reposition=False)
out = torch.empty((b, m * d), device="cuda")
for name, fl in (("default", 0), ("team", A.HK_FLAG_FORCE_TEAM)):
    with ops.forced(fl):
        f = timed(lambda: [ops.get_features(P, out=out) for _ in range(2)]) / 2 * 1e6
        fd = timed(lambda: [ops.get_features(dense, out=out) for _ in range(2)]) / 2 * 1e6
        ft = timed(lambda: [ops.get_features_torch(P) for _ in range(2)]) / 2 * 1e6
        z = timed(lambda: [ops.zeillinger(P) for _ in range(2)]) / 2 * 1e6
    print(f"{name:8s}: get_features {f:6.1f} us (dense states {fd:6.1f})  get_features_torch {ft:6.1f}  zeillinger {z:6.1f}", flush=True)
