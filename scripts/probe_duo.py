"""Dev probe: two lanes per game (hk::duo_kernel) against one (hk::fast_kernel) for fused rollouts of T steps.
Run once with HK_DUO=1 and once with HK_DUO=0 (the hook is read once per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
from probe_stages import timeit

for m, d in ((20, 3), (10, 3), (20, 4)):
    for b in (8192, 32768, 65536, 131072, 262144):
        P = ops.generate_points(b, m, d, 20, seed=42)
        S = torch.empty_like(P)
        ws = ops.rollout_workspace(b, 20, (m, d))
        line = []
        for T in (1, 2, 4, 8, 20):
            t = timeit(lambda: ops.rollout(S, T, 7, initial=P, defer_counts=True, workspace=ws), iters=10, reps=5)
            line.append(f"T={T}: {t:7.2f}")
        print(f"HK_DUO={os.environ.get('HK_DUO')} ({m},{d}) b={b:7d}  " + "  ".join(line), flush=True)
