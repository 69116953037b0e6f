"""Dev probe: hk_get_features / hk_get_features_torch timings (default dispatch vs the one-lane kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops
from probe_stages import timeit

for m, d in ((20, 3), (10, 3), (20, 4)):
    for b in (8192, 65536):
        P = ops.generate_points(b, m, d, 20, seed=42)
        for name, fn in (("get_features(scale)", lambda: ops.get_features(P, True)), ("get_features(raw)", lambda: ops.get_features(P, False)),
                         ("get_features_torch", lambda: ops.get_features_torch(P))):
            t = timeit(fn, iters=20, reps=20)
            with ops.forced(A.HK_FLAG_FORCE_ONE_LANE):
                t1 = timeit(fn, iters=20, reps=20)
            print(f"({m},{d}) b={b:6d} {name:20s} default {t:7.2f} us   one-lane {t1:7.2f} us", flush=True)
