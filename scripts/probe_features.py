"""Dev probe: time hk_get_features (rescale + row sort, the observation transform in front of the network)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
from probe_stages import timeit

for b, m, d in ((65536, 20, 3), (8192, 20, 3), (262144, 50, 4)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    cls = torch.randint(0, 2 ** d - d - 1, (b,), device="cuda", dtype=torch.int32)
    ax = torch.randint(0, d, (b,), device="cuda", dtype=torch.int32)
    Q = ops.step(P, cls, ax, stages=7)["points"]
    for sc in (True, False):
        t = timeit(lambda: ops.get_features(Q, scale_observation=sc), iters=20, reps=5)
        print(f"get_features b={b} ({m},{d}) scale={sc}: {t:.1f} us")
