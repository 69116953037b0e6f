"""Dev probe: a handful of hk_step launches (episode of 20 dependent steps) for rocprofv3 --pmc passes.
usage: pmc_step.py m d batch [flagname]   (flagname: four | two | one | team | none)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops

m, d, b = (int(x) for x in sys.argv[1:4])
force = {"four": A.HK_FLAG_FORCE_FOUR_LANES, "two": A.HK_FLAG_FORCE_TWO_LANES, "one": A.HK_FLAG_FORCE_ONE_LANE,
         "team": A.HK_FLAG_FORCE_TEAM, "none": 0}[sys.argv[4] if len(sys.argv) > 4 else "none"]
sem = sys.argv[5] if len(sys.argv) > 5 else "jax"
fl = ops.make_flags(sem, sem != "jax", sem == "torch") | force
P = ops.generate_points(b, m, d, 20, seed=42)
cls = torch.randint(0, 2 ** d - d - 1, (20, b), dtype=torch.int32, device="cuda")
masks = ops.decode_host_class(cls.reshape(-1), d, torch.float32).reshape(20, b, d).contiguous()
axes = torch.randint(0, d, (20, b), dtype=torch.int32, device="cuda")
bufs = [torch.empty_like(P), torch.empty_like(P)]
for rep in range(3):
    src = P
    for t in range(20):
        ops.step(src, masks[t] if sem != "list" else cls[t], axes[t], stages=7, flags=fl, out=bufs[t & 1], want=("done", "reward"))
        src = bufs[t & 1]
torch.cuda.synchronize()
