"""Dev probe: which property of a sorted batch makes the (50,4) episode faster -- homogeneous waves, or heavy waves first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (262144, 50, 4)))
gpw = 16 if m > 32 else 32
P = ops.generate_points(b, m, d, 20, seed=42)
npts = ops.get_num_points(P)
Q = torch.empty_like(P)
ws = ops.rollout_workspace(b, 20, (m, d))
def run(order):
    init = P.index_select(0, order).contiguous()
    ids = order.to(torch.int32)
    return timed(lambda: [ops.rollout(Q, 20, 1, initial=init, defer_counts=True, workspace=ws, game_ids=ids) for _ in range(3)]) / 3 * 1e6
ident = torch.arange(b, device="cuda")
glob = torch.argsort(npts, descending=True, stable=True)
g = torch.Generator(device="cuda").manual_seed(0)
waves = glob.reshape(-1, gpw)
shuf = waves[torch.randperm(waves.shape[0], device="cuda", generator=g)].reshape(-1)
rev = waves.flip(0).reshape(-1)
print(f"({m},{d}) x {b}: generated {run(ident):6.1f}  global sort {run(glob):6.1f}  global, waves shuffled {run(shuf):6.1f}  "
      f"global, light waves first {run(rev):6.1f}", flush=True)
for grp in (64, 128, 256, 1024):
    loc = torch.argsort(npts.reshape(-1, grp), dim=1, descending=True, stable=True) + torch.arange(0, b, grp, device="cuda")[:, None]
    plain = loc.reshape(-1)
    # heavy-first across groups: wave k of every group, then wave k + 1 of every group, ...
    hf = loc.reshape(-1, grp // gpw, gpw).transpose(0, 1).reshape(-1)
    print(f"   groups of {grp:5d}: sorted in place {run(plain):6.1f}  the groups' k-th waves together, heavy first {run(hf):6.1f}", flush=True)
Bn, ids = ops.bin_by_live_rows(P)
print(f"   hk_bin_by_live_rows: {timed(lambda: [ops.rollout(Q, 20, 1, initial=Bn, defer_counts=True, workspace=ws, game_ids=ids) for _ in range(3)]) / 3 * 1e6:6.1f}")
