"""Dev probe: E episodes back to back inside ONE launch (hk_rollout_desc.episodes, states resident in memory: every episode
reads its slab again) against one launch per episode of the headline kernel."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from hironaka_amd._lib import check, lib
from probe_records import timed

def episodes_launch(P0, P, ws, E, seed=7, flags=0):
    b, m, d = P0.shape
    r = A.hk_rollout_desc()
    r.points, r.points_in = P.data_ptr(), P0.data_ptr()
    r.seed, r.padding_value, r.reward_sign, r.episodes = seed, -1.0, 1.0, E
    r.batch, r.max_points, r.dim, r.dtype, r.steps = b, m, d, A.HK_F32, 20
    r.stages, r.flags = 7, flags | A.HK_FLAG_DEFER_COUNTS
    r.workspace, r.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib().hk_rollout(C.byref(r), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "hk_rollout")

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (32768, 20, 3), (131072, 20, 3), (262144, 50, 4)):
        P0 = ops.generate_points(b, m, d, 20, seed=42)
        P = torch.empty_like(P0)
        ws = ops.rollout_workspace(b, 20, (m, d))
        one = timed(lambda: [ops.rollout(P, 20, 7 + e, initial=P0, defer_counts=True, workspace=ws) for e in range(10)]) / 10 * 1e6
        line = f"({m},{d}) x {b}: one launch per episode {one:6.1f} us;  episodes per launch:"
        for E in (1, 2, 4, 10, 20):
            t = timed(lambda: episodes_launch(P0, P, ws, E)) / E * 1e6
            line += f"  E={E}: {t:6.1f}"
        print(line + " us per episode", flush=True)
    # the same through ops.rollout (bench.py's call)
    b, m, d = 65536, 20, 3
    P0 = ops.generate_points(b, m, d, 20, seed=42)
    P = torch.empty_like(P0)
    ws = ops.rollout_workspace(b, 20, (m, d))
    kw = dict(game_offset=0, stages=7, host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM)
    for reps in (1, 2):
        t = timed(lambda: [ops.rollout(P, 20, 7, initial=P0, episodes=10, defer_counts=True, workspace=ws, **kw) for _ in range(reps)]) / reps / 10 * 1e6
        print(f"ops.rollout(episodes=10), {reps} launch(es) per graph: {t:6.1f} us per episode")
    t = timed(lambda: episodes_launch(P0, P, ws, 10)) / 10 * 1e6
    print(f"raw descriptor again: {t:6.1f}")
