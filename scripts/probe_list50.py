"""list-semantics plain rollouts at (50,4) x 262144 and (20,3) x 65536: four-lane kernel vs team / two-lane kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (m, d, b) in ((50, 4, 262144), (20, 3, 65536), (20, 4, 65536)):
    P = ops.generate_points(b, m, d, 20, seed=1)
    Q = torch.empty_like(P)
    for sem, st in (("list", 7), ("list", 15), ("jax", 7)):
        fl = ops.make_flags(sem, sem != "jax", False)
        for name, fam in (("default", 0), ("four", A.HK_FLAG_FORCE_FOUR_LANES), ("team", A.HK_FLAG_FORCE_TEAM),
                          ("two", A.HK_FLAG_FORCE_TWO_LANES)):
            if fam == A.HK_FLAG_FORCE_TWO_LANES and m > 32: continue
            def run():
                Q.copy_(P)
                ops.rollout(Q, 20, 7, stages=st, flags=fl | fam, agent_policy=A.HK_AGENT_RANDOM_LEGAL if sem != "jax" else A.HK_AGENT_RANDOM)
            def base():
                Q.copy_(P)
            print(f"({m},{d})x{b} {sem} stages={st} {name}: {timeit(run) - timeit(base):.1f} us", flush=True)
