"""(20,3) / (20,4) / (10,3) x 65 536: observation features (both orders), Zeillinger's class, per-call timings"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (m, d) in ((20, 3), (20, 4), (10, 3)):
    b = 65536
    P = ops.generate_points(b, m, d, 20, seed=3)
    D = ops.generate_points(b, m, d, 20, seed=4, newton=False, reposition=False)
    out = torch.empty((b, m * d), device="cuda")
    outp = torch.empty_like(P)
    cp = timeit(lambda: outp.copy_(P))
    print(f"({m},{d}) x {b}: copy {cp:.1f} us | get_features {timeit(lambda: ops.get_features(P, out=out)):.1f} (dense {timeit(lambda: ops.get_features(D, out=out)):.1f})"
          f" | get_features_torch {timeit(lambda: ops.get_features_torch(P)):.1f} (dense {timeit(lambda: ops.get_features_torch(D)):.1f})"
          f" | zeillinger {timeit(lambda: ops.zeillinger(P)):.1f} (dense {timeit(lambda: ops.zeillinger(D)):.1f})", flush=True)
