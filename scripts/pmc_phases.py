"""Dev probe for rocprofv3 --pmc passes: the four-lane hk_step stopped after each phase (probe build, HK_QUAD_CUT), on
the 20 states of one episode -- instruction counts per phase.  The cut is told apart by the grid size: batch minus
64 x cut games.  Needs build_probe/libhk_probe.so (scripts/build_probe.sh).
usage: pmc_phases.py [m d batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libhk_probe.so")
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops

m, d, b = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (20, 3, 65536)))
os.environ["HK_QUAD_WPB"] = "4"
os.environ["HK_QUAD_DLDS"] = "0"
os.environ["HK_QUAD_CUT"] = "0"
P = ops.generate_points(b, m, d, 20, seed=42)
cls = torch.randint(0, 2 ** d - d - 1, (20, b), dtype=torch.int32, device="cuda")
masks = ops.decode_host_class(cls.reshape(-1), d, torch.float32).reshape(20, b, d).contiguous()
axes = torch.randint(0, d, (20, b), dtype=torch.int32, device="cuda")
states = [P]
for t in range(20):
    states.append(ops.step(states[-1], masks[t], axes[t], stages=7, flags=A.HK_FLAG_FORCE_FOUR_LANES, want=())["points"])
torch.cuda.synchronize()
for cut in (1, 2, 3, 4, 0):
    os.environ["HK_QUAD_CUT"] = str(cut)
    n = b - 64 * (cut if cut else 5)
    out = torch.empty_like(P[:n])
    for t in range(20):
        ops.step(states[t][:n], masks[t][:n], axes[t][:n], stages=7, flags=A.HK_FLAG_FORCE_FOUR_LANES, out=out,
                 want=("done", "reward", "num_points"))
torch.cuda.synchronize()
print("grids:", {cut: (b - 64 * (cut if cut else 5)) * 4 for cut in (1, 2, 3, 4, 0)})
