"""Randomised parity: random shapes, semantics, stage masks, padding values, policies and batch sizes through
the HIP kernels against the C oracle.

    python tests/fuzz_parity.py --minutes 3 [--seed 0]      # by hand, for a given number of minutes on the GPU

tests/test_gpu_fuzz.py runs a fixed, seeded slice of it (`run_cases`) as a `-m gpu` test.  A mismatch raises
`Mismatch` with the failing configuration (and dumps the arrays of a recording-rollout mismatch under
gpurun_out/)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hironaka_amd import _abi as A
from hironaka_amd import ops
from oracle import c_oracle as CO

FAST = [(4, 3), (5, 3), (10, 3), (16, 3), (20, 3), (8, 4), (20, 4), (50, 4)]


class Mismatch(AssertionError):
    pass


def operator_case(rng):
    """the two state-reading operators: Zeillinger's class (both variants, every kernel family) and the sorted
    observation features"""
    if rng.integers(0, 2) == 0:
        m, d = FAST[rng.integers(0, len(FAST))]
    else:
        d = int(rng.integers(2, 8))
        m = int(rng.integers(2, 65))
    b = int(rng.choice([1, 63, 65, 200, 513]))
    kind = int(rng.integers(4))
    cfg = dict(op=1, m=m, d=d, b=b, kind=kind)
    p = rng.integers(0, int(rng.choice([2, 3, 6, 20])), (b, m, d)).astype(np.float32)
    p[rng.random((b, m)) < float(rng.choice([0.0, 0.3, 0.8, 0.97]))] = -1.0
    if kind == 1:  # dyadic fractions: ties in (L, S) that survive a rescale
        p = np.where(p >= 0, p / np.float32(4.0), p).astype(np.float32)
    elif kind == 2:  # thirds: the isclose filter and rounding in the differences
        p = np.where(p >= 0, p / np.float32(3.0), p).astype(np.float32)
    elif kind == 3 and b > 4 and m > 2:  # irregular rows: the exact path
        p[3, 1] = -3.0
        p[2, 0, 0] = -0.5
    P = torch.as_tensor(p).cuda()
    force = [{}, {"force_generic": True}]
    if m <= 64 and d <= 6:
        force.append({"force_team": True})
    for sem in ("jax", "list"):
        want = CO.zeillinger(p, sem)
        for kw in force:
            if not np.array_equal(ops.zeillinger(P, sem, **kw).cpu().numpy(), want):
                raise Mismatch(f"ZEILLINGER MISMATCH {dict(cfg, sem=sem, **kw)}")
    if kind != 3:
        for scale in (True, False):
            want = CO.get_features(p, scale)
            got = ops.get_features(P, scale_observation=scale).cpu().numpy()
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                raise Mismatch(f"FEATURES MISMATCH {dict(cfg, scale=scale)}")


def one_case(rng):
    """one random configuration: a step, a recording rollout and the same rollout without records (or, one time in
    eight, the state-reading operators) against the C oracle"""
    if rng.integers(0, 8) == 0:
        operator_case(rng)
        return
    kind = rng.integers(0, 3)
    if kind == 0:
        m, d = FAST[rng.integers(0, len(FAST))]
    else:
        d = int(rng.integers(2, 7))
        m = int(rng.integers(2, 65))
    b = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 200, 1000, 3000]))
    sem = ["jax", "torch", "list"][rng.integers(0, 3)]
    pad = float(rng.choice([-1.0, -1.0, -1.0, -1e-8, -2.5]))
    force = int(rng.choice([0, 0, A.HK_FLAG_FORCE_ONE_LANE, A.HK_FLAG_FORCE_TWO_LANES, A.HK_FLAG_FORCE_FOUR_LANES,
                           A.HK_FLAG_FORCE_TEAM, A.HK_FLAG_FORCE_GENERIC]))
    noop, ign = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if rng.integers(0, 4) == 0:  # the flag sets of the compiled rollout configurations
        noop = ign = (sem == "torch")
        pad = -1.0 if sem == "jax" else pad
    stages = int(rng.choice([1, 2, 4, 5, 7, 7, 7, 8, 15]))
    maxv = int(rng.choice([2, 3, 6, 20, 1000]))
    holes = float(rng.choice([0.0, 0.3, 0.8, 0.97]))
    p = rng.integers(0, maxv, (b, m, d)).astype(np.float32)
    p[rng.random((b, m)) < holes] = pad
    if rng.random() < 0.3:  # fractional (rescaled) states
        p = np.where(p >= 0, p / np.float32(maxv), p).astype(np.float32)
    if rng.random() < 0.2 and b > 4 and m > 2:  # irregular rows: the exact slow path
        p[3, 1] = -3.0
        p[2, 0, 0] = -0.5
    cfg = dict(m=m, d=d, b=b, sem=sem, pad=pad, force=force, noop=noop, ign=ign, stages=stages, maxv=maxv,
               holes=holes)
    compact = bool(rng.integers(0, 4) == 0)
    cfg["compact"] = compact
    flags_o = CO.flags_of(sem=sem, noop_if_invalid=noop, ignore_ended=ign, compact_sorted=compact)
    flags_p = ops.make_flags(sem, noop, ign, compact_sorted=compact) | force
    cls = rng.integers(0, 2 ** d - d - 1, b).astype(np.int64)
    ax = rng.integers(0, d, b).astype(np.int32)
    P = torch.as_tensor(p).cuda()
    want = CO.step(p, cls, ax, stages=stages, flags=flags_o, padding_value=pad)
    got = ops.step(P, torch.as_tensor(cls).cuda(), torch.as_tensor(ax).cuda(), stages=stages, flags=flags_p,
                   padding_value=pad, want=("done", "prev_done", "reward", "num_points"))
    for k in ("points", "done", "prev_done", "reward", "num_points"):
        if not np.array_equal(got[k].cpu().numpy(), want[k]):
            raise Mismatch(f"STEP MISMATCH {k} {cfg}")
    # hk_step_features / HK_AXIS_MASKED_LOGITS (the search's expansion) where the four-lane step kernel serves them
    if (m, d) in ((20, 3), (10, 3), (20, 4)) and sem != "list" and not compact and (stages & 1) and \
            force in (0, A.HK_FLAG_FORCE_FOUR_LANES):
        scale = bool(rng.integers(0, 2))
        lg = rng.standard_normal((b, d)).astype(np.float32)
        lg[rng.random((b, d)) < 0.1] = np.nan
        lg[rng.random((b, d)) < 0.2] = 0.5
        cls32 = cls.astype(np.int32)
        for axis, is_logits in ((ax, False), (lg, True)):
            wantf = CO.step(p, cls32, axis, stages=stages, flags=flags_o, padding_value=pad, axis_logits=is_logits,
                            features=scale)
            feat = torch.empty((b, m * d), dtype=torch.float32, device="cuda")
            gotf = ops.step(P, torch.as_tensor(cls32).cuda(), torch.as_tensor(axis).cuda(), stages=stages,
                            flags=flags_p, padding_value=pad, want=("done", "reward"), features_out=feat,
                            scale_observation=scale)
            if not (np.array_equal(gotf["points"].cpu().numpy(), wantf["points"], equal_nan=True)
                    and np.array_equal(feat.cpu().numpy(), wantf["features"], equal_nan=True)
                    and np.array_equal(gotf["reward"].cpu().numpy(), wantf["reward"])):
                raise Mismatch(f"STEP FEATURES MISMATCH {dict(cfg, logits=is_logits, scale=scale)}")
    # fused rollout with records (JAX semantics flags only make sense with fixed policies too)
    T = int(rng.integers(1, 25))
    hp = int(rng.choice([A.HK_HOST_RANDOM, A.HK_HOST_RANDOM, A.HK_HOST_ALL_COORD, A.HK_HOST_ZEILLINGER]))
    apol = int(rng.choice([A.HK_AGENT_RANDOM, A.HK_AGENT_RANDOM_LEGAL, A.HK_AGENT_CHOOSE_FIRST,
                           A.HK_AGENT_CHOOSE_LAST]))
    rstages = int(rng.choice([7, 7, 5, 15]))
    seed = int(rng.integers(0, 1 << 40))
    off = int(rng.integers(0, 1 << 33))
    wp, wrec = CO.rollout(p, T, seed, game_offset=off, host_policy=hp, agent_policy=apol, stages=rstages,
                          flags=flags_o, padding_value=pad, record=True)
    Q = P.clone()
    rec = ops.rollout(Q, T, seed, game_offset=off, host_policy=hp, agent_policy=apol, stages=rstages,
                      flags=flags_p, padding_value=pad,
                      record=("obs", "host_class", "axis", "done", "reward", "game_length"))
    cfg.update(T=T, hp=hp, ap=apol, rstages=rstages, seed=seed, off=off)
    if not np.array_equal(Q.cpu().numpy(), wp):
        raise Mismatch(f"ROLLOUT MISMATCH points {cfg}")
    for k in ("obs", "host_class", "axis", "done", "reward", "game_length"):
        if not np.array_equal(rec[k].cpu().numpy(), wrec[k]):
            dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            os.makedirs(dump, exist_ok=True)
            np.savez(os.path.join(dump, "fuzz_mismatch.npz"), p=p, got=rec[k].cpu().numpy(), want=wrec[k],
                     got_final=Q.cpu().numpy(), want_final=wp, cfg=np.array(repr(cfg)))
            raise Mismatch(f"ROLLOUT MISMATCH {k} {cfg}")
    if not np.array_equal(rec["done_count"].cpu().numpy().astype(np.uint64), wrec["done_count"]):
        raise Mismatch(f"ROLLOUT MISMATCH done_count {cfg}")
    # the same rollout without records: the plain rollout kernels (incl. the two compiled configurations)
    Q2 = P.clone()
    plain = ops.rollout(Q2, T, seed, game_offset=off, host_policy=hp, agent_policy=apol, stages=rstages,
                        flags=flags_p, padding_value=pad)
    if not (np.array_equal(Q2.cpu().numpy(), wp)
            and np.array_equal(plain["done_count"].cpu().numpy().astype(np.uint64), wrec["done_count"])):
        raise Mismatch(f"PLAIN ROLLOUT MISMATCH {cfg}")
    # ... and with the small records alone (no observations): written per action window by the plain rollout kernels
    small = ("host_class", "axis", "done", "reward", "game_length")
    Q3 = P.clone()
    rec3 = ops.rollout(Q3, T, seed, game_offset=off, host_policy=hp, agent_policy=apol, stages=rstages,
                       flags=flags_p, padding_value=pad, record=small)
    if not (np.array_equal(Q3.cpu().numpy(), wp) and all(np.array_equal(rec3[k].cpu().numpy(), wrec[k]) for k in small)):
        raise Mismatch(f"SMALL RECORDS ROLLOUT MISMATCH {cfg}")


def run_cases(count: int, seed: int) -> int:
    """`count` seeded configurations (the slice the -m gpu test runs)"""
    rng = np.random.default_rng(seed)
    for _ in range(count):
        one_case(rng)
    return count


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + 60 * args.minutes
    n = 0
    while time.time() < t_end:
        try:
            one_case(rng)
        except Mismatch as e:
            print(e, flush=True)
            sys.exit(1)
        n += 1
        if n % 200 == 0:
            print(f"{n} configurations ok", flush=True)
    print(f"fuzz ok: {n} configurations", flush=True)


if __name__ == "__main__":
    main()
