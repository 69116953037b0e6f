"""Dev probe: hk_generate_points (randint -> newton -> reposition, jax/util.py:385-392) per shape, next to the same
result from the raw generator + the stage operators on the step kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (8192, 20, 3), (262144, 50, 4), (65536, 20, 4)):
        def gen():
            for i in range(3):
                ops.generate_points(b, m, d, 20, seed=42 + i)
        t_gen = timed(gen) / 3 * 1e6
        raw = ops.generate_points(b, m, d, 20, seed=42, newton=False, reposition=False)
        out = torch.empty_like(raw)
        def two():
            for i in range(3):
                r = ops.generate_points(b, m, d, 20, seed=42 + i, newton=False, reposition=False)
                ops.step(r, stages=A.HK_STAGE_NEWTON | A.HK_STAGE_REPOSITION, out=out)
        try:
            t_two = timed(two) / 3 * 1e6
            same = torch.equal(out, ops.generate_points(b, m, d, 20, seed=44))
        except Exception as e:
            t_two, same = float("nan"), str(e)[:60]
        print(f"({m},{d}) b={b}: generate_points {t_gen:8.1f} us   raw + newton + reposition {t_two:8.1f} us  (equal: {same})", flush=True)
