"""Dev probe: where does a four-lane hk_step launch spend its time?  Needs build_probe/libhk_probe.so
(scripts/build_probe.sh); HK_QUAD_CUT stops the kernel after a phase, HK_QUAD_WPB picks waves per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libhk_probe.so")
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops
from probe_stages import timeit

m, d, b = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (20, 3, 65536)))
P = ops.generate_points(b, m, d, 20, seed=42)
cls = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
mask = ops.decode_host_class(cls, d, torch.float32)
ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
out = torch.empty_like(P)
def episode_us(b_):
    """20 dependent steps (the bench's boundary_step protocol)"""
    P_ = ops.generate_points(b_, m, d, 20, seed=42)
    cl = torch.randint(0, 2 ** d - d - 1, (20, b_), dtype=torch.int32, device="cuda")
    mk = ops.decode_host_class(cl.reshape(-1), d, torch.float32).reshape(20, b_, d).contiguous()
    axs = torch.randint(0, d, (20, b_), dtype=torch.int32, device="cuda")
    bufs = [torch.empty_like(P_), torch.empty_like(P_)]

    def episode():
        src = P_
        for t in range(20):
            ops.step(src, mk[t], axs[t], stages=7, flags=A.HK_FLAG_FORCE_FOUR_LANES, out=bufs[t & 1], want=("done", "reward"))
            src = bufs[t & 1]
    return timeit(episode, iters=1, reps=50) / 20


os.environ["HK_QUAD_CUT"] = "0"
for wpb in (4, 1, 2):
    os.environ["HK_QUAD_WPB"] = str(wpb)
    for dl in (0, 12000 // (4 // wpb) if wpb < 4 else 12000, 27000 // (4 // wpb) if wpb < 4 else 27000):
        os.environ["HK_QUAD_DLDS"] = str(dl)
        print(f"episode protocol: waves/wg {wpb} dynamic LDS {dl:6d}: " + "  ".join(f"b={bb}: {episode_us(bb):6.2f} us" for bb in (32768, 65536, 131072)), flush=True)
os.environ["HK_QUAD_DLDS"] = "0"
names = {0: "full", 1: "slab in/out only", 2: "+ scan", 3: "+ compaction", 4: "+ stages (no write-back)"}
for wpb in (4, 1, 2, 8):
    os.environ["HK_QUAD_WPB"] = str(wpb)
    for cut in (1, 2, 3, 4, 0):
        os.environ["HK_QUAD_CUT"] = str(cut)
        t = timeit(lambda: ops.step(P, mask, ax, stages=7, flags=A.HK_FLAG_FORCE_FOUR_LANES, out=out,
                                    want=("done", "reward", "num_points")), iters=20, reps=50)
        print(f"({m},{d}) b={b} waves/wg {wpb}  {names[cut]:28s} {t:6.2f} us", flush=True)
