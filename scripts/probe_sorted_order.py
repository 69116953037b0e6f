"""Dev probe: what binning the games by live rows would buy the fused rollout -- the SAME kernel on a batch whose games
are physically sorted by their number of live rows (no kernel change, no indirection): waves of one bucket each, and
different placements of the heavy and the light waves over the grid (which workgroups share a SIMD)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
from probe_records import timed

if __name__ == "__main__":
    b, m, d = 65536, 20, 3
    P = ops.generate_points(b, m, d, 20, seed=42)
    cnt = (P[:, :, 0] >= 0).sum(1)
    perm = torch.argsort(cnt, descending=True, stable=True)
    nw = b // 32
    chunks = perm.view(nw, 32)
    half = nw // 2
    layouts = {
        "as generated": torch.arange(b, device=P.device),
        "sorted, heavy first": perm,
        "sorted, light first": perm.flip(0),
        "halves: heavy->light, then light->heavy": torch.cat([chunks[:half], chunks[half:].flip(0)]).reshape(-1),
        "halves: heavy->light | heaviest last": torch.cat([chunks[half:], chunks[:half]]).reshape(-1),
        "alternating heavy / light": torch.stack([chunks[:half], chunks[half:].flip(0)], 1).reshape(-1),
        "quarters interleaved": chunks.view(4, nw // 4, 32).transpose(0, 1).reshape(-1),
    }
    ws = ops.rollout_workspace(b, 20, (m, d))
    for name, order in layouts.items():
        X = P[order].contiguous()
        Q = torch.empty_like(X)
        def ep():
            for _ in range(5):
                ops.rollout(Q, 20, 1, initial=X, defer_counts=True, workspace=ws)
        print(f"{name:>42}: {timed(ep) / 5 * 1e6:6.2f} us per episode", flush=True)
