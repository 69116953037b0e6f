"""Dev probe: on-device binning (hk_generate_points_binned / hk_bin_by_live_rows, one launch each) against the plain
generator and the tensor-library sort of round 3, and the 20-step episode on the three orders."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed

def torch_bin(P):
    order = torch.argsort(ops.get_num_points(P), descending=True, stable=True)
    return P.index_select(0, order).contiguous(), order.to(torch.int32)

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (8192, 20, 3), (65536, 20, 4), (262144, 50, 4)):
        out = torch.empty((b, m, d), device="cuda")
        t_gen = timed(lambda: [ops.generate_points(b, m, d, 20, seed=42 + i, out=out) for i in range(3)]) / 3 * 1e6
        t_genb = timed(lambda: [ops.generate_points_binned(b, m, d, 20, 42 + i, out=out) for i in range(3)]) / 3 * 1e6
        P = ops.generate_points(b, m, d, 20, seed=42)
        t_bin = timed(lambda: [ops.bin_by_live_rows(P, out=out) for i in range(3)]) / 3 * 1e6
        t_tb = timed(lambda: torch_bin(P)) * 1e6
        Bn, ids = ops.bin_by_live_rows(P)
        Bg, idg = torch_bin(P)
        Q = torch.empty_like(P)
        ws = ops.rollout_workspace(b, 20, (m, d))
        def run(init, **kw):
            return timed(lambda: [ops.rollout(Q, 20, 1, initial=init, defer_counts=True, workspace=ws, **kw) for _ in range(3)]) / 3 * 1e6
        print(f"({m},{d}) x {b}: generate {t_gen:6.1f} us, generate binned {t_genb:6.1f}, bin an existing batch {t_bin:6.1f} "
              f"(tensor library: {t_tb:6.1f});  episode: generated order {run(P):6.1f} us, binned in groups + ids {run(Bn, game_ids=ids):6.1f}, "
              f"globally sorted + ids {run(Bg, game_ids=idg):6.1f}", flush=True)
