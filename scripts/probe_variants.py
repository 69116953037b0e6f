"""Dev probe (not part of the product): launch named variants of the fused rollout back to back so that a
rocprofv3 --pmc pass can be split per variant by dispatch order (prints the order it used)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A

b, reps = 65536, 30
fresh = ops.generate_points(b, 20, 3, 20, seed=42)
state = torch.empty_like(fresh)
S, R, N = A.HK_STAGE_SHIFT, A.HK_STAGE_REPOSITION, A.HK_STAGE_NEWTON
variants = [
    ("T20 random", dict(steps=20, stages=S | R | N)),
    ("T40 random", dict(steps=40, stages=S | R | N)),
    ("T20 fixed policies", dict(steps=20, stages=S | R | N, host_policy=A.HK_HOST_ALL_COORD,
                                agent_policy=A.HK_AGENT_CHOOSE_FIRST)),
    ("T20 random no reposition", dict(steps=20, stages=S | N)),
    ("T20 random shift only", dict(steps=20, stages=S)),
]
for name, kw in variants:
    kw = dict(kw)
    steps = kw.pop("steps")
    nocount = kw.pop("nocount", False)
    dc = None if nocount else torch.zeros(steps + 1, dtype=torch.int64, device="cuda")
    for _ in range(reps):
        ops.rollout(state, steps, 1, done_count=dc, initial=fresh, **kw)
    torch.cuda.synchronize()
    print(name, reps)
