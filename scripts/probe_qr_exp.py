"""Dev probe: one recording episode (observations + small records, counts deferred: the rollout kernel alone) with a
library given on the command line (scripts/build_qr_exp.sh: variants that leave out one part of the kernel)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from hironaka_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "product":
    _lib.LIB_PATH = os.path.join(root, "build_probe", sys.argv[1])
import torch
from hironaka_amd import ops
from probe_records import timed

if __name__ == "__main__":
    b, m, d = 65536, 20, 3
    P = ops.generate_points(b, m, d, 20, seed=42)
    Q = torch.empty_like(P)
    ws = ops.rollout_workspace(b, 20, (m, d))
    out = []
    for rec in ((), ("host_class", "axis", "done", "reward"), ("obs",), ("obs", "host_class", "axis", "done", "reward", "game_length")):
        def ep():
            for _ in range(5):
                ops.rollout(Q, 20, 1, initial=P, record=rec, defer_counts=True, workspace=ws)
        out.append(f"{len(rec)} fields {timed(ep) / 5 * 1e6:7.1f} us")
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'product':>16}: " + "  ".join(out), flush=True)
