#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Hironaka-game step on MI355X.

Workload (BASELINE.json configs[1]): dim=3, max_points=20, batch=65 536 games per GPU, float32,
max_value=20, reposition on / rescale off (hironaka/jax/jax_config.yml), uniformly random host
class and agent axis sampled INSIDE the kernel (Philox), episodes of T=20 steps
(max_length_game) from freshly generated states (generate_pts: randint -> newton -> reposition).

One "step" = one pass of the hot path over the whole per-GPU batch: host subset -> agent axis ->
shift -> reposition -> Newton polytope -> done/reward for every one of the 65 536 games, finished
games included (the reference steps them too).  The timed region runs EXACTLY K such steps as
K//20 episodes of 20 steps (+ one shorter episode of K%20 steps); an episode is ONE launch of the
fused rollout kernel hk::duo_kernel<20,3,rollout> (two lanes per game; SURVEY.md section 7 stage 5): the state is
read from HBM once, stays in registers for the 20 steps and is written back once.  Every episode
restarts from the resident fresh states (the kernel reads them and writes the working state: no
copy) and is followed by the reduction of the per-step finished-game counts (second tiny kernel).
Launches are captured once into hipGraphs (10 episodes per graph) and replayed, so python is not
in the timed loop.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL), games sharded by rank
(game_offset = rank*batch), no collective in the data path; one all-gather of the final states at
the end of the timed region (the trainer boundary).  value = steps of ALL ranks / max time.

Also on the JSON line:
  roofline      the rollout kernel: algorithmic bytes per launch (501 B per env-step, SURVEY.md
                8(d), x 65 536 games x 20 steps) / mean launch time from HIP events around the
                timed region.  Because the state never leaves the registers between steps the
                kernel's real HBM traffic is ~1/20 of the algorithmic figure, so `frac` can exceed
                1; `traffic` carries the measured bytes when a PMC profile is available.
  single_step   the same workload as one launch per env step (state through HBM every step: the
                take_actions-shaped drop-in), with its own algorithmic-bytes roofline figure.
  cpu_baseline  the scalar C/OpenMP oracle (oracle/hironaka_oracle.c) on this box's host cores,
                same episodes, bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

DIM, MAX_POINTS, BATCH, MAX_VALUE, EPISODE = 3, 20, 65536, 20, 20
BLOCK = 10  # episodes captured per hipGraph replay
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SEED = 7


def algorithmic_bytes_per_step(m: int, d: int) -> int:
    """read state + write state + action pair + reward + done."""
    return 2 * m * d * 4 + 4 * d + 4 + 4 + 1


def measured_traffic(key: str):
    """HBM bytes per launch from the committed rocprofv3 PMC profile (profiles/r*_hbm_traffic.json,
    written by scripts/summarise_profile.py; FETCH_SIZE x2 on gfx950 + WRITE_SIZE), newest round first.
    PMC counters cannot be collected from inside the benchmark process, hence the committed file."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                data = json.load(f)
            if data.get("batch") == BATCH and key in data:
                return data[key], os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def cpu_array_baseline(games: int = 8192):
    """The reference's own CPU formulation of the step -- broadcast [B,m,m,d] difference tensors and masks, as
    in _jax_ops.py / _torch_ops.py -- restated in numpy (oracle/np_oracle.py), one 20-step episode of a bounded
    sample.  (SURVEY 6 measured the reference's torch ops themselves at 0.09-0.11 M env-steps/s on 8 cores.)"""
    from oracle import np_oracle as NO
    fresh = NO.generate_points(games, MAX_POINTS, DIM, MAX_VALUE, 42)
    t0 = time.perf_counter()
    NO.rollout(fresh, EPISODE, SEED)
    dt = time.perf_counter() - t0
    return {"value": games * EPISODE / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"1 episode of {games} games x {EPISODE} steps ({dt:.1f} s) with oracle/np_oracle.py (numpy, the "
                      f"reference's broadcast-tensor formulation)"}


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def cpu_baseline(seconds: float = 12.0):
    """C/OpenMP oracle on the host cores: the same episodes (64k games x 20 steps), repeated
    until `seconds` of CPU work were timed."""
    from oracle import c_oracle as CO
    from hironaka_amd import _abi as A
    threads = CO.set_threads(0)
    fresh = CO.generate_points(BATCH, MAX_POINTS, DIM, MAX_VALUE, 42)
    stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    CO.rollout(fresh[:4096], EPISODE, SEED, record=False, stages=stages)  # warm-up
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        CO.rollout(fresh, EPISODE, SEED, record=False, stages=stages)
        done += BATCH * EPISODE
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{done // (BATCH * EPISODE)} episodes of {BATCH} games x {EPISODE} steps "
                      f"({dt:.1f} s) with oracle/hironaka_oracle.c (scalar C + OpenMP, {threads} threads "
                      f"on {os.cpu_count()} logical cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=BATCH, help="games per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # HK_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks then
    # share devices and the final gather is staged through host memory); the real path is nccl = RCCL.
    backend = os.environ.get("HK_BENCH_BACKEND", "nccl")
    device_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device_index)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from hironaka_amd import _abi as A
    from hironaka_amd import ops

    b, m, d = args.batch, MAX_POINTS, DIM
    stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    game_offset = rank * b
    fresh = ops.generate_points(b, m, d, MAX_VALUE, seed=42, game_offset=game_offset)
    state = torch.empty_like(fresh)
    done_count = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")
    step_counts = torch.zeros((EPISODE, 2), dtype=torch.int64, device="cuda")
    from hironaka_amd import distributed as hkdist

    def gather_final_states():
        """the trainer boundary: every rank ends up with all B*world final states"""
        if not distributed:
            return state
        if backend == "nccl":
            return hkdist.all_gather_games(state)
        return hkdist.all_gather_games(state.cpu())

    def max_over_ranks(x: float) -> float:
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    K, W = args.steps, args.warmup
    n_full, rem = divmod(K, EPISODE)
    kw = dict(game_offset=game_offset, stages=stages, host_policy=A.HK_HOST_RANDOM,
              agent_policy=A.HK_AGENT_RANDOM)

    count_ws = ops.rollout_workspace(b, EPISODE, (m, d))

    def episode(n_steps):
        # the episode restarts from the resident fresh states: the kernel reads `fresh`, writes `state`
        ops.rollout(state, n_steps, SEED, done_count=done_count[: n_steps + 1], initial=fresh, **kw)

    def episodes_deferred(n_episodes):
        # full episodes back to back; the per-workgroup finished-game counts accumulate in `count_ws` and are
        # summed into done_count ONCE (the reference sums its per-loop histograms the same way,
        # jax_trainer.py:513,533-534)
        for _ in range(n_episodes):
            ops.rollout(state, EPISODE, SEED, initial=fresh, defer_counts=True, workspace=count_ws, **kw)
        ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))

    def episode_stepwise(n_steps):
        for t in range(n_steps):
            ops.rollout(state, 1, SEED, step_offset=t, done_count=step_counts[t],
                        initial=fresh if t == 0 else None, **kw)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        episode(EPISODE)  # allocates the rollout workspace outside of any capture
        episode_stepwise(1)
        torch.cuda.synchronize()
        g_episode = capture(lambda: episode(EPISODE))
        episodes_deferred(1)
        torch.cuda.synchronize()
        g_block = capture(lambda: episodes_deferred(BLOCK))  # BLOCK episodes + one counter reduce per replay
        g_rem = capture(lambda: episode(rem)) if rem else None
        g_stepwise = None if args.no_single_step else capture(lambda: episode_stepwise(EPISODE))
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed) -------------------------------------------------------------------
    for _ in range(max(1, W // EPISODE)):
        g_episode.replay()
    gather_final_states()
    barrier()

    # ---- exactly K timed steps ---------------------------------------------------------------
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    done_count.zero_()
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(n_full // BLOCK):
        g_block.replay()
    for _ in range(n_full % BLOCK):
        g_episode.replay()
    if g_rem is not None:
        g_rem.replay()
    ev1.record()
    final_states = gather_final_states()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    region_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream, around the K steps
    assert final_states.shape[0] == world * b
    finished = int(done_count[EPISODE].item()) // max(1, n_full) if n_full else 0

    # ---- the one-launch-per-step variant, timed separately (rank 0's shard) --------------------
    single = None
    if g_stepwise is not None:
        n_ep = max(1, min(n_full, 50))
        g_stepwise.replay()
        barrier()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(n_ep):
            g_stepwise.replay()
        s1.record()
        torch.cuda.synchronize()
        step_s = s0.elapsed_time(s1) / 1e3 / (n_ep * EPISODE)
        single = {"value": b / step_s, "unit": "env-steps/s per GPU", "us_per_step": step_s * 1e6,
                  "steps": n_ep * EPISODE,
                  "algorithmic_GBps": b * algorithmic_bytes_per_step(m, d) / step_s / 1e9,
                  "frac_of_hbm_peak": b * algorithmic_bytes_per_step(m, d) / step_s / 1e9 / HBM_PEAK_GBS,
                  "note": "one launch of hk::duo_kernel<20,3,rollout> (T=1) + counter reduce per env step; "
                          "state read from and written to HBM every step"}

    def time_boundary_steps(start, n_ep):
        """seconds per hk_step launch, over episodes of EPISODE steps from `start` with pre-drawn actions"""
        nb = start.shape[0]
        cls = torch.randint(0, 2 ** d - d - 1, (EPISODE, nb), dtype=torch.int32, device="cuda")
        masks = ops.decode_host_class(cls.reshape(-1), d, torch.float32).reshape(EPISODE, nb, d).contiguous()
        axes = torch.randint(0, d, (EPISODE, nb), dtype=torch.int32, device="cuda")
        bufs = [torch.empty_like(start), torch.empty_like(start)]

        def episode_api():
            src = start
            for t in range(EPISODE):
                ops.step(src, masks[t], axes[t], stages=stages, out=bufs[t & 1], want=("done", "reward"))
                src = bufs[t & 1]

        with torch.cuda.stream(side):
            episode_api()
            torch.cuda.synchronize()
            g_api = capture(episode_api)
        torch.cuda.synchronize()
        g_api.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n_ep):
            g_api.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 1e3 / (n_ep * EPISODE)

    # ---- the drop-in boundary itself: hk_step with the trainer's arrays (state in, [B,d] subset mask and
    # [B] axis in, state + done + reward out) -- exactly SURVEY 8(d)'s 501 algorithmic bytes per env step ----
    api = None
    if g_stepwise is not None:
        api_s = time_boundary_steps(fresh, max(1, min(n_full, 50)))
        # what a plain device-to-device copy reaches on this box (read + write bytes per second)
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(big)
        dst.copy_(big)
        torch.cuda.synchronize()
        s0.record()
        for _ in range(10):
            dst.copy_(big)
        s1.record()
        torch.cuda.synchronize()
        copy_gbps = 10 * 2 * big.numel() * 4 / (s0.elapsed_time(s1) / 1e3) / 1e9
        del big, dst
        # ... and what a copy of exactly one state batch (15.7 MB in, 15.7 MB out: the traffic of one hk_step)
        # takes as a kernel of its own, replayed from a hipGraph like the steps above
        same = torch.empty_like(fresh)
        with torch.cuda.stream(side):
            same.copy_(fresh)
            torch.cuda.synchronize()
            g_copy = capture(lambda: [same.copy_(fresh) for _ in range(20)])
        torch.cuda.synchronize()
        g_copy.replay()
        torch.cuda.synchronize()
        s0.record()
        for _ in range(10):
            g_copy.replay()
        s1.record()
        torch.cuda.synchronize()
        same_us = s0.elapsed_time(s1) * 1e3 / 200
        del same
        gbps = b * algorithmic_bytes_per_step(m, d) / api_s / 1e9
        api = {"value": b / api_s, "unit": "env-steps/s per GPU", "us_per_step": api_s * 1e6,
               "steps": max(1, min(n_full, 50)) * EPISODE, "algorithmic_GBps": gbps,
               "frac_of_hbm_peak": gbps / HBM_PEAK_GBS, "device_copy_GBps": copy_gbps, "frac_of_device_copy": gbps / copy_gbps,
               "state_copy_us": same_us,
               "note": "one hk_step launch per env step (hk::duo_kernel<20,3,step,jax>): f32 state + f32 [B,d] mask "
                       "+ i32 axis read from HBM, state + done + reward written back; device_copy_GBps = a 1 GiB "
                       "device-to-device copy, state_copy_us = a copy kernel over one state batch (the same bytes as "
                       "one hk_step without the actions and outcomes)"}

    # ---- SURVEY 8(d), config 2's second protocol: the agent draws its axis among the host's coordinates
    # only, under the torch and the list sibling's semantics (illegal / finished games not shifted; list:
    # survivors sorted + compacted after every step) -- fused 20-step rollouts and single hk_step launches ----
    legal = None
    if world == 1 and b == BATCH and not args.no_single_step:
        legal = {"agent": "uniform over the host's subset (HK_AGENT_RANDOM_LEGAL)"}
        cls_l = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
        axis_l = torch.zeros(b, dtype=torch.int32, device="cuda")  # coordinate 0 is in 3 of the 4 subsets
        out_l = torch.empty_like(fresh)
        for sem in ("torch", "list"):
            fl = ops.make_flags(sem, noop_if_invalid=True, ignore_ended=True)

            def roll_sem():
                ops.rollout(state, EPISODE, SEED, done_count=done_count, initial=fresh, stages=stages, flags=fl,
                            host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM_LEGAL)

            def step_sem():
                ops.step(fresh, cls_l, axis_l, stages=stages, flags=fl, out=out_l, want=("done", "reward"))

            ts = []
            for fn in (roll_sem, step_sem):
                with torch.cuda.stream(side):
                    fn()
                    torch.cuda.synchronize()
                    gl = capture(lambda: [fn() for _ in range(10)])
                torch.cuda.synchronize()
                gl.replay()
                torch.cuda.synchronize()
                l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                l0.record()
                for _ in range(10):
                    gl.replay()
                l1.record()
                torch.cuda.synchronize()
                ts.append(l0.elapsed_time(l1) / 1e3 / 100)
            legal[sem] = {"fused_rollout_us_per_episode": ts[0] * 1e6, "fused_env_steps_per_s": b * EPISODE / ts[0],
                          "hk_step_us": ts[1] * 1e6, "hk_step_env_steps_per_s": b / ts[1]}
        done_count.zero_()

    # ---- same kernels at the batch that saturates one GPU (BASELINE configs[3]'s 524 288 games on ONE
    # device): one lane per game means 65 536 games are only 1024 instruction streams for 1024 SIMDs ----
    large = None
    if world == 1 and b == BATCH and not args.no_single_step:
        bl = 8 * BATCH
        fresh_l = ops.generate_points(bl, m, d, MAX_VALUE, seed=42)
        state_l = torch.empty_like(fresh_l)
        dc_l = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")
        sc_l = torch.zeros((EPISODE, 2), dtype=torch.int64, device="cuda")
        kw_l = dict(stages=stages, host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM)

        def ep_fused():
            ops.rollout(state_l, EPISODE, SEED, done_count=dc_l, initial=fresh_l, **kw_l)

        def ep_steps():
            for t in range(EPISODE):
                ops.rollout(state_l, 1, SEED, step_offset=t, done_count=sc_l[t],
                            initial=fresh_l if t == 0 else None, **kw_l)

        with torch.cuda.stream(side):
            ep_fused()
            ep_steps()
            torch.cuda.synchronize()
            gf, gs = capture(ep_fused), capture(ep_steps)
        torch.cuda.synchronize()
        times = []
        for gr in (gf, gs):
            gr.replay()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(20):
                gr.replay()
            a1.record()
            torch.cuda.synchronize()
            times.append(a0.elapsed_time(a1) / 1e3 / (20 * EPISODE))
        bs = algorithmic_bytes_per_step(m, d)
        api_l = time_boundary_steps(fresh_l, 20)
        large = {"batch": bl, "boundary_step_us": api_l * 1e6,
                 "boundary_step_frac_of_hbm_peak": bl * bs / api_l / 1e9 / HBM_PEAK_GBS, "fused_env_steps_per_s": bl / times[0], "single_step_env_steps_per_s": bl / times[1],
                 "single_step_us": times[1] * 1e6, "single_step_algorithmic_GBps": bl * bs / times[1] / 1e9,
                 "single_step_frac_of_hbm_peak": bl * bs / times[1] / 1e9 / HBM_PEAK_GBS,
                 "note": "not the headline config: shows where the kernels saturate one MI355X"}
        del fresh_l, state_l

    # ---- BASELINE configs[2]: dim 4, 50 points, 262 144 games on one GPU (the team kernel: four lanes per
    # game); a parity-test configuration, measured here so that its numbers come from the same run --------
    config3 = None
    if world == 1 and b == BATCH and not args.no_single_step:
        m3, d3, b3 = 50, 4, 262144
        fresh3 = ops.generate_points(b3, m3, d3, MAX_VALUE, seed=42)
        cls3 = torch.randint(0, 2 ** d3 - d3 - 1, (b3,), dtype=torch.int32, device="cuda")
        mask3 = ops.decode_host_class(cls3, d3, torch.float32)
        axis3 = torch.randint(0, d3, (b3,), dtype=torch.int32, device="cuda")
        out3 = torch.empty_like(fresh3)
        state3 = torch.empty_like(fresh3)
        dc3 = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")

        def step3():
            ops.step(fresh3, mask3, axis3, stages=stages, out=out3, want=("done", "reward"))

        def roll3():
            ops.rollout(state3, EPISODE, SEED, done_count=dc3, initial=fresh3, stages=stages,
                        host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM)

        times3 = []
        for fn in (step3, roll3):
            with torch.cuda.stream(side):
                fn()
                torch.cuda.synchronize()
                g3 = capture(lambda: [fn() for _ in range(5)])
            torch.cuda.synchronize()
            g3.replay()
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(4):
                g3.replay()
            c1.record()
            torch.cuda.synchronize()
            times3.append(c0.elapsed_time(c1) / 1e3 / 20)
        bs3 = algorithmic_bytes_per_step(m3, d3)
        config3 = {"workload": f"dim={d3}, max_points={m3}, batch={b3} (BASELINE configs[2]), hk::team_kernel<4>",
                   "hk_step_us": times3[0] * 1e6, "hk_step_env_steps_per_s": b3 / times3[0],
                   "hk_step_algorithmic_GBps": b3 * bs3 / times3[0] / 1e9,
                   "hk_step_frac_of_hbm_peak": b3 * bs3 / times3[0] / 1e9 / HBM_PEAK_GBS,
                   "fused_rollout_us_per_episode": times3[1] * 1e6,
                   "fused_env_steps_per_s": b3 * EPISODE / times3[1],
                   "algorithmic_bytes_per_env_step": bs3}
        del fresh3, out3, state3

    if rank == 0:
        bytes_step = algorithmic_bytes_per_step(m, d)
        launches = n_full + (1 if rem else 0)
        launch_s = (region_ms / 1e3) / max(1, launches)
        steps_per_launch = K / max(1, launches)
        achieved = b * bytes_step * steps_per_launch / launch_s / 1e9
        traffic, traffic_src = (None, None)
        if b == BATCH and rem == 0:
            traffic, traffic_src = measured_traffic("rollout_T20_bytes_per_launch")
        if single is not None and b == BATCH:
            single["traffic"], _ = measured_traffic("single_step_bytes_per_launch")
        if api is not None and b == BATCH:
            api["traffic"], _ = measured_traffic("boundary_step_bytes_per_launch")
        out = {
            "metric": "env-steps/sec at dim=3, max_pts=20, batch=65536; 1/2/4/8 GPUs",
            "value": world * b * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"dim={d}, max_points={m}, batch={b} games per GPU (BASELINE configs[1]), random "
                            f"host+agent policies sampled in-kernel, reposition=True, rescale=False, episodes of "
                            f"{EPISODE} steps from generate_pts states (max_value={MAX_VALUE}); fused rollout: "
                            f"{EPISODE} env steps per launch",
                "parallelism": f"{world} x independent game shards, all-gather of final states",
                "launch": f"hipGraph replay ({BLOCK} episodes per graph): one rollout kernel per episode, one "
                          f"counter reduce per graph",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"hk::duo_kernel<{m},{d},rollout,jax> ({b} games x {EPISODE} steps per launch, two lanes per game)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(b * bytes_step * steps_per_launch),
                "mean_launch_us": launch_s * 1e6,
                "note": "algorithmic = 501 B/env-step (SURVEY 8d) x games x steps; the fused kernel keeps the "
                        "state in registers between steps, so real HBM traffic per launch is ~2*240 B/game",
            },
            "games_finished_per_episode": finished,
        }
        if single is not None:
            out["single_step"] = single
        if api is not None:
            out["boundary_step"] = api
        if large is not None:
            out["large_batch"] = large
        if legal is not None:
            out["legal_axis_torch_list_semantics"] = legal
        if config3 is not None:
            out["config3_dim4_50points"] = config3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_array_formulation"] = cpu_array_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
