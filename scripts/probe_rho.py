import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd.rollout import compute_rho
for loops in (10, 50):
    compute_rho("random", "random", spec=(20, 3), batch_size=65536, max_value=20, max_length=21, num_of_loops=2, reposition=True, key=1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rho, det = compute_rho("random", "random", spec=(20, 3), batch_size=65536, max_value=20, max_length=21, num_of_loops=loops, reposition=True, key=7)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"compute_rho 65536 games x {loops} loops: {dt*1e3:.2f} ms = {dt/loops*1e6:.1f} us per loop, rho {rho:.4f}, details[:4] {det[:4]}")
