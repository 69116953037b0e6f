"""Dev probe: JAXTrainer.compute_rho by name -- ONE launch with the batches drawn inside the kernel (round 4) against
the round-3 form of the loop (generate_points + rollout per loop, alternating between two streams)."""
import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd import ops, _abi as A
from hironaka_amd.rollout import compute_rho, details_from_done_counts, rho_from_details
from probe_records import timed

def old_loop(b, m, d, loops, key, host=A.HK_HOST_RANDOM):
    ws = ops.rollout_workspace(b, 20, (m, d))
    totals = torch.zeros(21, dtype=torch.int64, device="cuda")
    main = torch.cuda.current_stream()
    lanes = [main, torch.cuda.Stream()]
    lanes[-1].wait_stream(main)
    pts = [torch.empty((b, m, d), device="cuda") for _ in lanes]
    for loop in range(loops):
        with torch.cuda.stream(lanes[loop % 2]):
            ops.generate_points(b, m, d, 20, seed=key + loop, out=pts[loop % 2])
            ops.rollout(pts[loop % 2], 20, key + loop, host_policy=host, defer_counts=True, workspace=ws)
    main.wait_stream(lanes[-1])
    ops.reduce_counts(ws, totals, b, 20, (m, d))
    return totals

if __name__ == "__main__":
    for b, m, d in ((65536, 20, 3), (8192, 20, 3), (524288, 20, 3), (65536, 20, 4), (262144, 50, 4)):
        for loops in (10,):
            kw = dict(spec=(m, d), batch_size=b, max_value=20, max_length=21, num_of_loops=loops, reposition=True)
            compute_rho("random", "random", key=1, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rho, det = compute_rho("random", "random", key=7, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            want = details_from_done_counts(old_loop(b, m, d, loops, 7), b * loops)
            t_new = timed(lambda: ops.rollout_generated(b, (m, d), 20, 7, max_value=20, episodes=loops, defer_counts=True,
                                                        workspace=ops.rollout_workspace(b, 20, (m, d)))) / loops * 1e6
            t_one = timed(lambda: [ops.rollout_generated(b, (m, d), 20, 7 + e, max_value=20, episodes=1, defer_counts=True,
                                                         workspace=ops.rollout_workspace(b, 20, (m, d))) for e in range(loops)]) / loops * 1e6
            torch.cuda.synchronize(); t0 = time.perf_counter(); old_loop(b, m, d, loops, 7); torch.cuda.synchronize()
            t_old = (time.perf_counter() - t0) / loops * 1e6
            print(f"({m},{d}) x {b}, {loops} loops: compute_rho wall {dt / loops * 1e6:7.1f} us/loop (rho {rho:.4f}, equal to the "
                  f"round-3 loop: {det == want}); one launch of {loops} episodes {t_new:7.1f} us/loop, one launch per loop "
                  f"{t_one:7.1f}, round-3 loop (wall) {t_old:7.1f}", flush=True)
