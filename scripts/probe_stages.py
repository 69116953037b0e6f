"""Dev probe (not part of the product): time one-step launches with different stage masks /
kernel families via hipGraph replay + HIP events, to see where a launch's time goes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A

def timeit(fn, iters=50, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(iters):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)  # us per call

def main():
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    m, d = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 3)
    P = ops.generate_points(b, m, d, 20, seed=42)
    Q = torch.empty_like(P)
    ncls = 2 ** d - d - 1
    cls = torch.randint(0, ncls, (b,), device="cuda", dtype=torch.int32)
    ax = torch.randint(0, d, (b,), device="cuda", dtype=torch.int32)
    dc = torch.zeros(2, dtype=torch.int64, device="cuda")
    print(f"batch {b} spec ({m},{d}) state {P.numel()*4/1e6:.1f} MB")
    print("d2d copy            %.2f us" % timeit(lambda: Q.copy_(P)))
    for name, st in (("none", 0), ("shift", 1), ("reposition", 2), ("newton", 4), ("shift+repos+newton", 7), ("all4", 15)):
        for kind, fl in (("default", 0), ("team", A.HK_FLAG_FORCE_TEAM), ("generic", A.HK_FLAG_FORCE_GENERIC)):
            t = timeit(lambda: ops.step(P, cls, ax, stages=st, flags=fl, out=Q))
            print(f"step stages={name:20s} {kind} {t:8.2f} us")
    S = P.clone()
    dcs = {T: torch.zeros(T + 1, dtype=torch.int64, device="cuda") for T in (1, 2, 5, 20)}
    for kind, fl in (("default", 0), ("team", A.HK_FLAG_FORCE_TEAM)):
        for T in (1, 2, 5, 20):
            def f():
                ops.rollout(S, T, 7, done_count=dcs[T], initial=P, flags=fl)
            t = timeit(f, iters=10)
            print(f"rollout T={T:2d} {kind} {t:8.2f} us  ({t/T:.2f} us/step)")

if __name__ == "__main__":
    main()
