"""Dev probe: fused 20-step rollouts at (50,4) x 262144 -- generated order, binned, Zeillinger's host, torch semantics,
the run-time configured kernel, the recording kernels, generated initial states."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = 262144, 50, 4
P = ops.generate_points(b, m, d, 20, seed=42)
B, ids = ops.bin_by_live_rows(P)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
def run(init, reps=3, **kw):
    def ep():
        for _ in range(reps):
            ops.rollout(Q, 20, 1, initial=init, defer_counts=True, workspace=ws, **kw)
    return timed(ep) / reps * 1e6
tf = ops.make_flags("torch", noop_if_invalid=True, ignore_ended=True)
print(f"plain jax {run(P):7.1f} us   binned + ids {run(B, game_ids=ids):7.1f}   torch/legal {run(P, flags=tf, agent_policy=A.HK_AGENT_RANDOM_LEGAL):7.1f}   "
      f"run-time configured (all_coord/choose_first) {run(P, host_policy=A.HK_HOST_ALL_COORD, agent_policy=A.HK_AGENT_CHOOSE_FIRST):7.1f}   "
      f"zeillinger {run(P, reps=1, host_policy=A.HK_HOST_ZEILLINGER):7.1f}", flush=True)
def gen(E):
    return timed(lambda: ops.rollout_generated(b, (m, d), 20, 7, max_value=20, episodes=E, defer_counts=True, workspace=ws)) / E * 1e6
print(f"generated initial states, counts only: 1 episode {gen(1):7.1f} us   4 episodes {gen(4):7.1f} us per episode", flush=True)
