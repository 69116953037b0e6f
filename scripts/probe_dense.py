"""Dev probe: hk_step from states with every row live (generate_points(newton=False)): the worst case of the domination
test, per shape and kernel family."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed

for b, m, d in ((262144, 50, 4), (65536, 20, 3), (65536, 20, 4)):
    for newton in (False, True):
        P = ops.generate_points(b, m, d, 20, seed=43, newton=newton, reposition=newton)
        cls = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
        mask = ops.decode_host_class(cls, d, torch.float32)
        ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
        out = torch.empty_like(P)
        res = []
        for name, fl in (("default", 0), ("team", A.HK_FLAG_FORCE_TEAM)):
            def f():
                for _ in range(4):
                    ops.step(P, mask, ax, stages=7, flags=fl, out=out, want=("done", "reward"))
            res.append(f"{name} {timed(f) / 4 * 1e6:8.2f} us")
        print(f"({m},{d}) b={b} {'generated' if newton else 'dense    '}: " + "  ".join(res), flush=True)
