"""Dev probe: hk_step on four lanes per game (hk::quad_kernel, HK_FLAG_FORCE_FOUR_LANES) against the default kernels:
bit-exact agreement on random and dense states, then time per launch (episodes of 20 dependent steps in a hipGraph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hironaka_amd import _abi as A
from hironaka_amd import ops
from probe_stages import timeit

STAGES = 7
Q = A.HK_FLAG_FORCE_FOUR_LANES


def check(m, d, b, sem="jax", stages=STAGES, dense=False, act=0):
    P = ops.generate_points(b, m, d, 20, seed=5, newton=not dense, reposition=not dense)
    cls = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
    ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
    if act:  # the compiled action layouts: f32 mask + i32 / i64 / f32 axis (some out of range / non-integral)
        cls = ops.decode_host_class(cls, d, torch.float32)
        ax = ax.to([torch.int32, torch.int32, torch.int64, torch.float32][act])
        if b > 8:
            ax[3] = d
            ax[5] = -1
            if act == 3:
                ax[7] = 0.5
    fl = ops.make_flags(sem, sem != "jax", sem == "torch")
    want = ("done", "prev_done", "reward", "num_points")
    ref = ops.step(P, cls, ax, stages=stages, flags=fl | A.HK_FLAG_FORCE_TWO_LANES, want=want)
    got = ops.step(P, cls, ax, stages=stages, flags=fl | Q, want=want)
    bad = [k for k in ref if not torch.equal(ref[k], got[k])]
    print(f"parity ({m},{d}) b={b} sem={sem} stages={stages} dense={dense} act={act}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
    if bad and "points" in bad:
        g = int((ref["points"] != got["points"]).any(dim=2).any(dim=1).nonzero()[0])
        print("game", g, "\nin\n", P[g].cpu().numpy(), "\nref\n", ref["points"][g].cpu().numpy(), "\ngot\n", got["points"][g].cpu().numpy(),
              "cls", cls[g].cpu().numpy(), "axis", ax[g].cpu().numpy())
    return not bad


def episode_time(m, d, b, force, dense=False, flags=0):
    P = ops.generate_points(b, m, d, 20, seed=42, newton=not dense, reposition=not dense)
    cls = torch.randint(0, 2 ** d - d - 1, (20, b), dtype=torch.int32, device="cuda")
    masks = ops.decode_host_class(cls.reshape(-1), d, torch.float32).reshape(20, b, d).contiguous()
    axes = torch.randint(0, d, (20, b), dtype=torch.int32, device="cuda")
    bufs = [torch.empty_like(P), torch.empty_like(P)]

    def episode():
        src = P
        for t in range(20):
            ops.step(src, masks[t], axes[t], stages=STAGES, flags=flags | force, out=bufs[t & 1], want=("done", "reward"))
            src = P if dense else bufs[t & 1]

    return timeit(episode, iters=1, reps=50) / 20


def fresh_step_time(m, d, b, force):
    """bench.py's config3 protocol: independent launches from generate_pts states"""
    P = ops.generate_points(b, m, d, 20, seed=42)
    cls = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
    mask = ops.decode_host_class(cls, d, torch.float32)
    ax = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
    out = torch.empty_like(P)
    return timeit(lambda: ops.step(P, mask, ax, stages=STAGES, flags=force, out=out, want=("done", "reward")), iters=5, reps=20)


ok = True
for m, d in ((20, 3), (10, 3), (20, 4), (50, 4)):
    for b in (1, 17, 1000, 4099):
        for sem in ("jax", "torch"):
            ok &= check(m, d, b, sem)
            ok &= check(m, d, b, sem, stages=15)
        ok &= check(m, d, b, dense=True)
        for act in (1, 2, 3):
            ok &= check(m, d, b, act=act)
            ok &= check(m, d, b, "torch", act=act)
print("ALL PARITY OK" if ok else "PARITY FAILURES", flush=True)
for m, d, b in ((20, 3, 65536), (20, 3, 32768), (20, 3, 131072), (20, 3, 524288), (10, 3, 65536), (20, 4, 65536), (50, 4, 262144)):
    for dense in (False, True):
        t2 = episode_time(m, d, b, A.HK_FLAG_FORCE_TWO_LANES if m * d <= 80 else 0, dense)
        t4 = episode_time(m, d, b, Q, dense)
        print(f"({m},{d}) b={b:7d} dense={dense}: default/two-lane {t2:7.2f} us   four-lane {t4:7.2f} us", flush=True)
print(f"(50,4) b=262144 from generate_pts states: team {fresh_step_time(50, 4, 262144, 0):7.2f} us   four-lane {fresh_step_time(50, 4, 262144, Q):7.2f} us", flush=True)
print(f"(20,3) b=65536 from generate_pts states: two-lane {fresh_step_time(20, 3, 65536, A.HK_FLAG_FORCE_TWO_LANES):7.2f} us   four-lane {fresh_step_time(20, 3, 65536, Q):7.2f} us", flush=True)
