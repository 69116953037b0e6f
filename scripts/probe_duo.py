"""Dev probe: two lanes per game (hk::duo_kernel) against one (hk::fast_kernel) for fused rollouts of T steps.
The kernel family is forced per launch with HK_FLAG_FORCE_TWO_LANES / HK_FLAG_FORCE_ONE_LANE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops
from probe_stages import timeit

for m, d in ((20, 3), (10, 3), (20, 4)):
    for b in (8192, 32768, 65536, 131072, 262144):
        P = ops.generate_points(b, m, d, 20, seed=42)
        S = torch.empty_like(P)
        ws = ops.rollout_workspace(b, 20, (m, d))
        for name, force in (("two lanes", A.HK_FLAG_FORCE_TWO_LANES), ("one lane ", A.HK_FLAG_FORCE_ONE_LANE)):
            line = []
            for T in (1, 2, 4, 8, 20):
                t = timeit(lambda: ops.rollout(S, T, 7, initial=P, defer_counts=True, workspace=ws, flags=force),
                           iters=10, reps=5)
                line.append(f"T={T}: {t:7.2f}")
            print(f"{name} ({m},{d}) b={b:7d}  " + "  ".join(line), flush=True)
