"""Dev probe: list-semantics steps (sorted + compacted output, what the gym environments run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_stages import timeit

for b, m, d in ((65536, 20, 3), (1024, 20, 3), (262144, 50, 4)):
    P = ops.generate_points(b, m, d, 20, seed=42)
    Q = torch.empty_like(P)
    cls = torch.randint(0, 2 ** d - d - 1, (b,), device="cuda", dtype=torch.int32)
    ax = torch.randint(0, d, (b,), device="cuda", dtype=torch.int32)
    for name, fl in (("list", ops.make_flags("list", True)), ("list generic", ops.make_flags("list", True, force_generic=True))):
        t = timeit(lambda: ops.step(P, cls, ax, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON, flags=fl, out=Q), iters=10, reps=3)
        print(f"step {name} b={b} ({m},{d}): {t:.1f} us")
