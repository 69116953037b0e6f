import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = 65536, 20, 3
P = ops.generate_points(b, m, d, 20, seed=42)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
for T in (1, 2, 3, 4, 6, 8, 12, 20):
    out = []
    for name, fl in (("two", A.HK_FLAG_FORCE_TWO_LANES), ("one", A.HK_FLAG_FORCE_ONE_LANE)):
        def ep():
            for _ in range(3):
                ops.rollout(Q, T, 1, initial=P, defer_counts=True, workspace=ws, flags=fl, host_policy=A.HK_HOST_ZEILLINGER, agent_policy=A.HK_AGENT_RANDOM_LEGAL)
        out.append(f"{name} {timed(ep) / 3 * 1e6:7.2f}")
    r = ops.rollout(Q, T, 1, initial=P, host_policy=A.HK_HOST_ZEILLINGER, agent_policy=A.HK_AGENT_RANDOM_LEGAL, record=("game_length",))
    gl = r["game_length"]
    print(f"T={T:2d}: " + "  ".join(out) + f"   finished {(gl >= 0).float().mean().item():.3f}  live rows now {ops.get_num_points(Q).float().mean().item():.2f} max {ops.get_num_points(Q).max().item()}", flush=True)
