import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from hironaka_amd import ops, _abi as A
from probe_records import timed
b, m, d = 262144, 50, 4
out = torch.empty((b, m, d), device="cuda")
for mv in (20, 200):
    def gen():
        for i in range(3):
            ops.generate_points(b, m, d, mv, seed=42 + i, out=out)
    print(f"(50,4) max_value {mv}: {timed(gen) / 3 * 1e6:7.1f} us", flush=True)
