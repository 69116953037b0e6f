import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
for b, m, d in ((262144, 50, 4), (8192, 20, 3)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    B, ids = ops.bin_by_live_rows(P)
    Q = torch.empty_like(P)
    ws = ops.rollout_workspace(b, 20, (m, d))
    def run(init, **kw):
        def ep():
            for _ in range(3):
                ops.rollout(Q, 20, 1, initial=init, defer_counts=True, workspace=ws, **kw)
        return timed(ep) / 3 * 1e6
    print(f"({m},{d}) b={b}: generated {run(P):7.1f} us  binned + ids {run(B, game_ids=ids):7.1f} us  binned {run(B):7.1f} us", flush=True)
