"""Dev probe: fused rollouts that also write the per-step records a trainer consumes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_stages import timeit

for b, m, d in ((65536, 20, 3), (8192, 20, 3), (262144, 50, 4)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    Q = torch.empty_like(P)
    for rec in ((), ("host_class", "axis", "done", "reward"), ("obs", "host_class", "axis", "done", "reward", "game_length")):
        t = timeit(lambda: ops.rollout(Q, 20, 1, initial=P, record=rec), iters=5, reps=3)
        nbytes = b * 20 * (m * d * 4 if "obs" in rec else 0)
        print(f"rollout T=20 b={b} ({m},{d}) record={len(rec)} fields: {t:.1f} us" + (f"  (obs writes {nbytes/1e6:.0f} MB -> {nbytes/t/1e6:.2f} TB/s)" if nbytes else ""))
