"""Dev probe: a handful of hk_generate_points launches for rocprofv3 --pmc passes.
usage: pmc_generate.py m d batch [max_value] [flagname] [stages]   (flagname: four | one | team | none; stages: e.g. nr, n, raw)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops

m, d, b = (int(x) for x in sys.argv[1:4])
mv = int(sys.argv[4]) if len(sys.argv) > 4 else 20
force = {"four": A.HK_FLAG_FORCE_FOUR_LANES, "one": A.HK_FLAG_FORCE_ONE_LANE, "team": A.HK_FLAG_FORCE_TEAM,
         "none": 0}[sys.argv[5] if len(sys.argv) > 5 else "none"]
st = sys.argv[6] if len(sys.argv) > 6 else "nr"
out = torch.empty((b, m, d), device="cuda")
for rep in range(30):
    ops.generate_points(b, m, d, mv, seed=42 + rep, flags=force, out=out, newton="n" in st, reposition="r" in st)
torch.cuda.synchronize()
