"""Dev probe: the DQN trainers' roll-out step (FusedGame.step over HipPoints) and its pieces, against the same
step written the reference's way (features = argsort + gather, move = three tensor programs' worth of launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops
from hironaka_amd.core import HipPoints
from hironaka_amd.fused_game import FusedGame
from probe_stages import timeit


def eager(fn, iters=20):
    """us per call without graph capture (FusedGame.step selects rows with a boolean mask: a sync per step)"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def torch_features(p):
    order = torch.argsort(p[:, :, 0], dim=1, descending=True)
    return p.gather(1, order.unsqueeze(-1).repeat(1, 1, p.shape[2])).clone()


for b, m, d in ((65536, 20, 3), (262144, 50, 4)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    t_hip = timeit(lambda: ops.get_features_torch(P), iters=20, reps=3)
    t_torch = timeit(lambda: torch_features(P), iters=20, reps=3)
    print(f"get_features b={b} ({m},{d}): hk_get_features_torch {t_hip:.1f} us, argsort+gather {t_torch:.1f} us")
    flat = torch.nn.Flatten()

    class AgentNet(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(m * d + d, 256), torch.nn.ReLU(), torch.nn.Linear(256, d))

        def forward(self, x):
            return self.body(torch.cat([flat(x["points"]), x["coords"]], dim=1))

    host_net = torch.nn.Sequential(flat, torch.nn.Linear(m * d, 256), torch.nn.ReLU(), torch.nn.Linear(256, 2 ** d - d - 1))
    game = FusedGame(host_net, AgentNet(), log_time=False)
    pts = HipPoints(P.clone())

    def one(role):
        pts.points.copy_(P)
        return game.step(pts, role, scale_observation=True, exploration_rate=0.2)

    for role in ("host", "agent"):
        t = eager(lambda: one(role))
        print(f"FusedGame.step({role}) b={b} ({m},{d}): {t:.1f} us per step = {b / t:.1f} M env-steps/s")
    mask = torch.ones((b, d), device="cuda")
    ax = torch.zeros(b, device="cuda", dtype=torch.int64)
    t3 = timeit(lambda: (pts.shift(mask, ax), pts.get_newton_polytope(), pts.rescale()), iters=10, reps=3)
    t1 = timeit(lambda: pts.step(mask, ax, rescale=True), iters=10, reps=3)
    print(f"move b={b} ({m},{d}): three launches {t3:.1f} us, fused {t1:.1f} us")
