"""Dev probe: cost of the Zeillinger host (hk_zeillinger, and as the host policy of fused rollouts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_stages import timeit

for b, m, d in ((65536, 20, 3), (32, 10, 3), (262144, 50, 4)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    t = timeit(lambda: ops.zeillinger(P), iters=10, reps=3)
    print(f"zeillinger b={b} ({m},{d}): {t:.1f} us")
    Q = torch.empty_like(P)
    dc = torch.zeros(21, dtype=torch.int64, device="cuda")
    for hp, name in ((A.HK_HOST_RANDOM, "random"), (A.HK_HOST_ZEILLINGER, "zeillinger")):
        t = timeit(lambda: ops.rollout(Q, 20, 1, initial=P, done_count=dc, host_policy=hp,
                                       agent_policy=A.HK_AGENT_RANDOM_LEGAL), iters=5, reps=3)
        print(f"rollout T=20 host={name} b={b} ({m},{d}): {t:.1f} us")
