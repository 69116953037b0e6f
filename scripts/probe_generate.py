"""Dev probe: hk_generate_points (randint -> newton -> reposition, jax/util.py:385-392) per shape and kernel family
(default = hk::quadgen_kernel where it exists; one lane per game / team forced by flag), the raw draws alone, and the
fraction of the HBM peak the writes reach."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (8192, 20, 3), (524288, 20, 3), (65536, 10, 3), (65536, 20, 4), (262144, 50, 4)):
        out = torch.empty((b, m, d), device="cuda")
        line = f"({m},{d}) b={b}:"
        ref = None
        for name, fl in (("default", 0), ("one", A.HK_FLAG_FORCE_ONE_LANE), ("team", A.HK_FLAG_FORCE_TEAM)):
            def gen():
                for i in range(3):
                    ops.generate_points(b, m, d, 20, seed=42 + i, flags=fl, out=out)
            t = timed(gen) / 3 * 1e6
            got = ops.generate_points(b, m, d, 20, seed=44, flags=fl)
            ref = got if ref is None else ref
            line += f"  {name} {t:7.1f} us ({b * m * d * 4 / t / 8e6:.2f} of HBM peak{'' if torch.equal(ref, got) else ' DIFFERENT'})"
        def raw():
            for i in range(3):
                ops.generate_points(b, m, d, 20, seed=42 + i, newton=False, reposition=False, out=out)
        line += f"  raw draws {timed(raw) / 3 * 1e6:7.1f} us"
        print(line, flush=True)
