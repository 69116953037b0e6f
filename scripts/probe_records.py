"""Dev probe: fused rollouts that also write the per-step records a trainer consumes, per kernel family (hipGraph of
10 episodes, median of 8 event segments: no host time in the figures)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A

def timed(fn, min_s=0.03, seg=8):
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

if __name__ == "__main__":
  shapes = ((65536, 20, 3), (8192, 20, 3), (262144, 20, 3), (65536, 50, 4))
  for b, m, d in shapes:
      P = ops.generate_points(b, m, d, 20, seed=42)
      Q = torch.empty_like(P)
      for rec in ((), ("host_class", "axis", "done", "reward"), ("obs", "host_class", "axis", "done", "reward", "game_length")):
          out = []
          for name, fl in (("default", 0), ("four", A.HK_FLAG_FORCE_FOUR_LANES), ("two", A.HK_FLAG_FORCE_TWO_LANES),
                           ("one", A.HK_FLAG_FORCE_ONE_LANE)):
              reps = 5
              def ep():
                  for _ in range(reps):
                      ops.rollout(Q, 20, 1, initial=P, record=rec, flags=fl)
              out.append(f"{name} {timed(ep) / reps * 1e6:8.1f} us")
          nbytes = b * 20 * (m * d * 4 if "obs" in rec else 0)
          print(f"b={b} ({m},{d}) record={len(rec)} fields: " + "  ".join(out) + (f"   (obs {nbytes/1e6:.0f} MB)" if nbytes else ""), flush=True)
