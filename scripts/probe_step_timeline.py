"""Dev probe: per-wave time line of ONE hk_step launch on the four-lane kernel (probe build, HK_QUAD_CUT=9): when each
wave starts, has its slab, is done with the stages, has its stores out -- and which waves end the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_probe", "libhk_probe.so")
os.environ["HK_QUAD_CUT"] = "9"
import numpy as np
import torch
from hironaka_amd import ops, _abi as A

m, d = 20, 3
for b in (65536, 524288):
    P = ops.generate_points(b, m, d, 20, seed=42)
    cls = torch.randint(0, 4, (b,), dtype=torch.int32, device="cuda")
    mask = ops.decode_host_class(cls, d, torch.float32)
    ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
    out = torch.empty_like(P)
    for rep in range(5):
        res = ops.step(P, mask, ax, stages=7, out=out, want=("done", "reward", "num_points"))
    torch.cuda.synchronize()
    w = res["num_points"].cpu().numpy().reshape(-1, 16).astype(np.int64)
    t0, t1, t2, t3, smax, hwid, blk, wv = (w[:, i] for i in range(8))
    base = t0.min()
    us = lambda x: ((x - base) & 0xFFFFFFFF) / 100.0
    print(f"(20,3) x {b}: waves {len(t0)}; launch span (first start .. last stores out) {us(t3).max():.2f} us")
    for name, v in (("start", us(t0)), ("slab landed", us(t1)), ("stages done", us(t2)), ("stores out", us(t3))):
        print(f"  {name:12s} min {v.min():6.2f}  p10 {np.percentile(v,10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}")
    for name, v in (("wait for slab", us(t1) - us(t0)), ("scan..stages", us(t2) - us(t1)), ("store", us(t3) - us(t2)), ("lifetime", us(t3) - us(t0))):
        print(f"  {name:12s} mean {v.mean():6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v,90):6.2f}  max {v.max():6.2f}")
    late = np.argsort(us(t3))[-int(0.05 * len(t0)):]
    print(f"  the last 5% of the waves to finish: started at {us(t0)[late].mean():.2f} (all: {us(t0).mean():.2f}), waited {(us(t1)-us(t0))[late].mean():.2f} "
          f"for the slab (all: {(us(t1)-us(t0)).mean():.2f}), computed {(us(t2)-us(t1))[late].mean():.2f} (all: {(us(t2)-us(t1)).mean():.2f}), "
          f"stored in {(us(t3)-us(t2))[late].mean():.2f} (all: {(us(t3)-us(t2)).mean():.2f}); slots per lane {smax[late].mean():.2f} (all: {smax.mean():.2f})")
    xcc = (hwid >> 20) & 0xF if False else None
