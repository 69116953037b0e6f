"""Dev probe: fused rollouts against Zeillinger's host (jax/players.py:55-109) per kernel family, and the standalone
hk_zeillinger operator."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (32768, 20, 3), (8192, 20, 3), (262144, 50, 4)):
        P = ops.generate_points(b, m, d, 20, seed=42)
        Q = torch.empty_like(P)
        ws = ops.rollout_workspace(b, 20, (m, d))
        out = []
        for name, fl in (("default", 0), ("four", A.HK_FLAG_FORCE_FOUR_LANES), ("two", A.HK_FLAG_FORCE_TWO_LANES),
                         ("one", A.HK_FLAG_FORCE_ONE_LANE), ("team", A.HK_FLAG_FORCE_TEAM)):
            def ep():
                for _ in range(3):
                    ops.rollout(Q, 20, 1, initial=P, defer_counts=True, workspace=ws, flags=fl,
                                host_policy=A.HK_HOST_ZEILLINGER, agent_policy=A.HK_AGENT_RANDOM_LEGAL)
            try:
                out.append(f"{name} {timed(ep) / 3 * 1e6:8.1f} us")
            except Exception as e:
                out.append(f"{name} n/a")
        def plain():
            for _ in range(3):
                ops.rollout(Q, 20, 1, initial=P, defer_counts=True, workspace=ws)
        print(f"({m},{d}) b={b}: Zeillinger host " + "  ".join(out) + f"   random host {timed(plain) / 3 * 1e6:8.1f} us", flush=True)
