"""Fused 20-step rollouts: the rollout kernel families (hk::duo_kernel / hk::fast_kernel) over batch
sizes; hipGraph of 10 episodes, median of 8 event segments."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import _abi as A, ops

def timed(fn, min_s=0.04, seg=8):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(seg + 1)]
    g.replay(); torch.cuda.synchronize()
    e[0].record(); [g.replay() for _ in range(3)]; e[1].record(); torch.cuda.synchronize()
    est = max(e[0].elapsed_time(e[1]) / 3e3, 1e-7)
    n = max(1, math.ceil(min_s / seg / est))
    for _ in range(seg * n): g.replay()
    torch.cuda.synchronize()
    e[0].record()
    for k in range(seg):
        for _ in range(n): g.replay()
        e[k + 1].record()
    torch.cuda.synchronize()
    t = sorted(e[k].elapsed_time(e[k + 1]) / 1e3 / n for k in range(seg))
    return 0.5 * (t[seg // 2 - 1] + t[seg // 2])

shapes = [(20, 3)] if len(sys.argv) < 2 else [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for m, d in shapes:
    for b in ((4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288) if m * d <= 64 else (16384, 65536, 262144)):
        fresh = ops.generate_points(b, m, d, 20, seed=42)
        state = torch.empty_like(fresh)
        ws = ops.rollout_workspace(b, 20, (m, d))
        out = {}
        for name, fl in (("default", 0), ("two", A.HK_FLAG_FORCE_TWO_LANES), ("quad", A.HK_FLAG_FORCE_FOUR_LANES),
                         ("one", A.HK_FLAG_FORCE_ONE_LANE)):
            def ep():
                for _ in range(10):
                    ops.rollout(state, 20, 7, initial=fresh, flags=fl, defer_counts=True, workspace=ws)
            out[name] = timed(ep) / 10 * 1e6
        print(f"({m},{d}) b={b:7d}  " + "  ".join(f"{k} {v:8.2f} us" for k, v in out.items()), flush=True)
