import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops
from probe_stages import timeit
for m, d in ((20, 3), (10, 3), (20, 4)):
    for b in (32768, 65536, 98304, 131072, 196608, 262144, 524288):
        P = ops.generate_points(b, m, d, 20, seed=42)
        S = torch.empty_like(P)
        ws = ops.rollout_workspace(b, 20, (m, d))
        line = []
        for name, force in (("two", A.HK_FLAG_FORCE_TWO_LANES), ("one", A.HK_FLAG_FORCE_ONE_LANE)):
            t = timeit(lambda: ops.rollout(S, 20, 7, initial=P, defer_counts=True, workspace=ws, flags=force), iters=10, reps=5)
            line.append(f"{name}-lane {t:7.2f} us")
        # hk_step: four-lane vs one-lane
        cls = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
        mask = ops.decode_host_class(cls, d, torch.float32)
        ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
        for name, force in (("step4", A.HK_FLAG_FORCE_FOUR_LANES), ("step1", A.HK_FLAG_FORCE_ONE_LANE)):
            t = timeit(lambda: ops.step(P, mask, ax, stages=7, flags=force, out=S, want=("done", "reward")), iters=10, reps=5)
            line.append(f"{name} {t:7.2f} us")
        print(f"({m},{d}) b={b:7d}  " + "  ".join(line), flush=True)
