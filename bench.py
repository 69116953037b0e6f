#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Hironaka-game step on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): dim=3, max_points=20, batch=65 536 games per GPU, float32,
max_value=20, reposition on / rescale off (hironaka/jax/jax_config.yml), uniformly random host
class and agent axis sampled INSIDE the kernel (Philox), episodes of T=20 steps
(max_length_game) from freshly generated states (generate_pts: randint -> newton -> reposition).

One "step" = one pass of the hot path over the whole per-GPU batch: host subset -> agent axis ->
shift -> reposition -> Newton polytope -> done/reward for every one of the 65 536 games, finished
games included (the reference steps them too).  K steps are K//20 episodes of 20 steps (+ one shorter
episode of K%20 steps); an episode is ONE launch of the fused rollout kernel: the state is read from
HBM once, stays in registers for the 20 steps and is written back once.  Every episode restarts
from the resident fresh states.  The per-step finished-game counts are reduced once per 10 episodes
(second tiny kernel).  Launches are captured once into hipGraphs and replayed.

Timing.  After W untimed warm-up steps and an untimed clock warm-up as long as the measurement, the
K-step region is REPEATED back to back until >= 0.2 s of GPU time have been timed (`repeats` on the JSON
line; `steps` stays K).  The whole repeated region is bracketed by barrier + torch.cuda.synchronize()
on both sides (wall clock, max over ranks -> `value`, `ms_per_step`); HIP events on the launch stream
split it into 8 segments whose MEDIAN gives `ms_per_step_median` and `roofline.mean_launch_us`.

N>1: `python bench.py --gpus N` starts its own ranks (python -m torch.distributed.run, one process
per GPU, backend nccl = RCCL) before touching the GPU and relays rank 0's line; launched by
torch.distributed.run itself it reads RANK / LOCAL_RANK / WORLD_SIZE.  Games are sharded by rank
(game_offset = rank*batch), no collective in the data path; one all-gather of the final states at
the end of the timed region (the trainer boundary).  value = steps of ALL ranks / max time.

Also on the JSON line (N=1): `roofline` (binding resource of the fused kernel + measured HBM fraction
+ SURVEY 8(d)'s algorithmic figure), `boundary_step` (hk_step, the drop-in boundary, HBM roofline),
`single_step`, `single_step_dense` (20 live rows: worst case of the domination test), the list / torch
sibling protocols, the 524 288-game and (50,4) configurations, `config5_mcts_simulate` (BASELINE
configs[4]) and the CPU baselines (C/OpenMP oracle; torch-CPU array formulation on all cores).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM, MAX_POINTS, BATCH, MAX_VALUE, EPISODE = 3, 20, 65536, 20, 20
BLOCK = 10  # episodes per counter reduction
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# VALU issue ceiling: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles per SIMD at 2.4 GHz
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 2
SEED = 7
# HK_BENCH_TIME_SCALE < 1 shortens every timed region (the rocprofv3 --pmc passes of scripts/profile_round.sh, where
# each dispatch is serialised and costs ~100 us); the driver's runs use the defaults
_SCALE = float(os.environ.get("HK_BENCH_TIME_SCALE", "1"))
MIN_TIMED_S = 0.2 * _SCALE  # GPU time of the headline measurement
MIN_SECTION_S = 0.04 * _SCALE  # ... of every secondary section
SEGMENTS = 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=BATCH, help="games per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-step", action="store_true", help="headline only (skip every secondary section)")
    ap.add_argument("--no-search", action="store_true", help="skip config5 (MCTS simulate)")
    ap.add_argument("--binning", action="store_true",
                    help="roll the headline episodes out on the batch ordered by live rows (game ids keep the streams)")
    ap.add_argument("--no-binned", action="store_true",
                    help="skip the binned_by_live_rows section (profile passes: its launches share the headline kernel's name)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="skip the two-episodes-in-flight section (profiling runs: its launches overlap in the trace)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as CHILD processes (never exec: this process may not
    replace itself once anything touched the GPU, and nothing has) and return their exit status.  Rank 0 of the
    children prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def algorithmic_bytes_per_step(m: int, d: int) -> int:
    """read state + write state + action pair + reward + done."""
    return 2 * m * d * 4 + 4 * d + 4 + 4 + 1


def profile_constants():
    """Per-launch PMC figures of the committed rocprofv3 passes (profiles/r*_hbm_traffic.json, written by
    scripts/summarise_profile.py: HBM bytes = FETCH_SIZE x2 on gfx950 + WRITE_SIZE; SQ_INSTS_VALU), newest round
    first.  PMC counters cannot be collected from inside the benchmark process, hence the committed file."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                data = json.load(f)
            if data.get("batch") == BATCH:
                return data, os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return {}, None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # HK_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs than ranks (ranks then
    # share devices and the final gather is staged through host memory); the real path is nccl = RCCL.
    backend = os.environ.get("HK_BENCH_BACKEND", "nccl")
    n_dev = max(1, torch.cuda.device_count())
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"--gpus {world} needs {world} devices, this box has {n_dev} "
                         f"(HK_BENCH_BACKEND=gloo rehearses the flow with ranks sharing a device)")
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    from hironaka_amd import _abi as A
    from hironaka_amd import distributed as hkdist
    from hironaka_amd import ops

    b, m, d = args.batch, MAX_POINTS, DIM
    stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    game_offset = rank * b
    fresh = ops.generate_points(b, m, d, MAX_VALUE, seed=42, game_offset=game_offset)
    # The headline rolls out the GENERATED order.  --binning: the games ordered by live rows at generate time (widest
    # first, so that a wave holds games of one size), the permutation handed to hk_rollout as game ids -- every game
    # keeps its policy stream, an episode is game by game the episode of the generated order
    # (tests/test_gpu_parity.py::test_reordered_batch_with_game_ids_rolls_out_like_the_original).  Not the default:
    # binning a batch costs more than one episode gains (`binned_by_live_rows` below), it pays for batches that are
    # rolled out many times or binned off the clock.
    binned, game_ids = ops.bin_by_live_rows(fresh)
    head = dict(initial=binned, game_ids=game_ids) if args.binning else dict(initial=fresh)
    state = torch.empty_like(fresh)
    done_count = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")
    step_counts = torch.zeros((EPISODE, 2), dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()

    def capture(fn):
        with torch.cuda.stream(side):
            fn()  # eager first: allocations and lazy initialisation happen outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):  # the stream of the eager call: per-stream workspaces exist
                fn()
        torch.cuda.synchronize()
        return g

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    def timed_replays(replay, min_seconds):
        """`replay()` enqueues one unit of work (graph replays) on the current stream.  Estimate its duration,
        run an untimed clock warm-up as long as the measurement, then time SEGMENTS segments of n units each
        with HIP events on the launch stream.  Returns (median, mean) seconds per unit and the unit count."""
        e = [torch.cuda.Event(enable_timing=True) for _ in range(SEGMENTS + 1)]
        replay()
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(3):
            replay()
        e[1].record()
        torch.cuda.synchronize()
        est = max(e[0].elapsed_time(e[1]) / 3e3, 1e-7)
        n = max(1, math.ceil(min_seconds / SEGMENTS / est))
        for _ in range(SEGMENTS * n):  # clock warm-up
            replay()
        torch.cuda.synchronize()
        e[0].record()
        for s in range(SEGMENTS):
            for _ in range(n):
                replay()
            e[s + 1].record()
        torch.cuda.synchronize()
        seg = sorted(e[s].elapsed_time(e[s + 1]) / 1e3 / n for s in range(SEGMENTS))
        return 0.5 * (seg[SEGMENTS // 2 - 1] + seg[SEGMENTS // 2]), sum(seg) / SEGMENTS, SEGMENTS * n

    K, W = args.steps, args.warmup
    n_full, rem = divmod(K, EPISODE)
    kw = dict(game_offset=game_offset, stages=stages, host_policy=A.HK_HOST_RANDOM,
              agent_policy=A.HK_AGENT_RANDOM)
    count_ws = ops.rollout_workspace(b, EPISODE, (m, d))

    def episodes_deferred(n_episodes, n_steps=EPISODE):
        # episodes back to back, each restarting from the resident fresh states (the kernel reads `fresh`, writes
        # `state`: no copy); the per-workgroup finished-game counts accumulate in `count_ws`
        for _ in range(n_episodes):
            ops.rollout(state, n_steps, SEED, defer_counts=True, workspace=count_ws, **head, **kw)

    def reduce_counts():
        # ... and are summed into done_count once per BLOCK episodes (the reference sums its per-loop histograms
        # the same way, jax_trainer.py:513,533-534)
        ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))

    def one_region():
        """exactly K env steps: n_full episodes of 20 steps (+ one shorter episode of K % 20 steps)"""
        if n_full:
            episodes_deferred(n_full)
        if rem:
            episodes_deferred(1, rem)

    launches_per_region = n_full + (1 if rem else 0)
    reduce_every = max(1, BLOCK // max(1, launches_per_region))  # regions between two counter reductions
    g_episode = capture(lambda: episodes_deferred(1))
    gather_events = []
    if distributed:
        # One rank-local graph per region, and after EVERY region the trainer-boundary all-gather of its final states
        # (an episode, then the gather).  Episodes are independent, so the gather of region i does
        # not sit on the launch stream: the regions alternate between two state buffers and hkdist.GatherPipeline runs
        # the collective on a side stream behind an event while region i + 1 computes (`value`); the same regions
        # without any gather (`value_without_gather`) and with the gather serialised on the launch stream and timed with
        # its own events (`gather_us`) are measured next to it.
        group = 1
        states = [state, torch.empty_like(state)]

        def region_into(buf):
            def run():
                for n_ep, n_st in ((n_full, EPISODE), (1 if rem else 0, rem)):
                    for _ in range(n_ep):
                        ops.rollout(buf, n_st, SEED, defer_counts=True, workspace=count_ws, **head, **kw)
            return run
        g_region = [capture(region_into(buf)) for buf in states]
        g_region_reduce = [capture(lambda buf=buf: (region_into(buf)(), reduce_counts())) for buf in states]
        direct = os.environ.get("HK_BENCH_GATHER", "rccl") == "direct"
        shard = hkdist.Shard(rank * b, b, world * b)
        pipe = hkdist.GatherPipeline(shard, depth=2, direct=direct)

        def boundary(buf):  # (gloo rehearsal: the states are staged through host memory)
            return buf if backend == "nccl" else buf.cpu()
    else:
        # ONE graph holds `group` regions + the counter reduction, so a short region (the driver's --steps 20 is a
        # single 12-23 us launch) does not pay a graph replay of its own: the launches of a group run back to back
        group = reduce_every
        g_group = capture(lambda: ([one_region() for _ in range(group)], reduce_counts()))

    def run_groups(count, start=0, gather="pipelined"):
        """count x group regions (every region = exactly K env steps); N > 1: the gather of every region's final states
        `pipelined` behind the next region, `serial` on the launch stream (timed with its own events), or `none`"""
        out = None
        for r in range(start, start + count):
            if not distributed:
                g_group.replay()
                continue
            p = r & 1
            pipe.wait(tickets[p])  # the last gather that reads this buffer is done before the region overwrites it
            (g_region_reduce[p] if (r + 1) % reduce_every == 0 else g_region[p]).replay()
            if gather == "pipelined":
                tickets[p] = pipe.submit(boundary(states[p]))
            elif gather == "serial":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = hkdist.all_gather_games(boundary(states[p]), shard, direct=direct, validate=False)
                e1.record()
                gather_events.append((e0, e1))
        if distributed and gather == "pipelined" and count:
            out = pipe.result()
        return out
    tickets = [None, None]

    # ---- warm-up (untimed): W steps, then a clock warm-up as long as the measurement ----------------------------
    for _ in range(max(1, W // EPISODE)):
        g_episode.replay()
    g_reduce = capture(reduce_counts)
    g_reduce.replay()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(SEGMENTS + 1)]

    def timed_pass(gather):
        """an untimed estimate + clock warm-up, then SEGMENTS x per_seg groups inside one barrier + synchronize bracket;
        returns (wall seconds, median seconds per region, regions, last gathered tensor)"""
        run_groups(2, gather=gather)
        torch.cuda.synchronize()
        ev[0].record()
        run_groups(4, gather=gather)
        ev[1].record()
        torch.cuda.synchronize()
        est = max(ev[0].elapsed_time(ev[1]) / 1e3 / 4, 1e-7)  # seconds per group
        per_seg = max(1, math.ceil(MIN_TIMED_S / SEGMENTS / est))
        if distributed:
            per_seg = math.ceil(per_seg / (2 * reduce_every)) * 2 * reduce_every  # (whole reduce periods, both buffers)
        per_seg = int(max_over_ranks([per_seg])[0])  # the same on every rank
        run_groups(SEGMENTS * per_seg, gather=gather)
        barrier()
        done_count.zero_()
        barrier()
        t0 = time.perf_counter()
        ev[0].record()
        last = None
        for sgm in range(SEGMENTS):
            got = run_groups(per_seg, sgm * per_seg, gather=gather)
            last = got if got is not None else last
            ev[sgm + 1].record()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        wall = max_over_ranks([time.perf_counter() - t0])[0]
        seg = sorted(max_over_ranks([ev[i].elapsed_time(ev[i + 1]) / 1e3 for i in range(SEGMENTS)]))
        median_region = 0.5 * (seg[SEGMENTS // 2 - 1] + seg[SEGMENTS // 2]) / (per_seg * group)
        return wall, median_region, SEGMENTS * per_seg * group, last

    # ---- the timed region: `repeats` x exactly K steps ----------------------------------------------------------
    elapsed, median_region_s, repeats, final_states = timed_pass("pipelined")
    final_states = state if final_states is None else final_states
    assert final_states.shape[0] == world * b
    timed_episodes = repeats * n_full
    finished = int(done_count[EPISODE].item()) // timed_episodes if timed_episodes else 0
    gather_us = value_without_gather = overlap_hidden = None
    if distributed:
        # the pipelined result equals the serial gather of the same buffer
        last_buf = states[(SEGMENTS * (repeats // SEGMENTS) - 1) & 1]
        serial = hkdist.all_gather_games(boundary(last_buf), shard, direct=direct, validate=False)
        torch.cuda.synchronize()
        assert torch.equal(serial, final_states), "the pipelined gather differs from the serial one"
        t_nog, _, reps_nog, _ = timed_pass("none")
        value_without_gather = world * b * K * reps_nog / t_nog
        run_groups(2 * reduce_every, gather="serial")
        gather_events.clear()
        run_groups(16 * reduce_every, gather="serial")
        torch.cuda.synchronize()
        gs = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in gather_events)
        gather_us = max_over_ranks([gs[len(gs) // 2]])[0]
        # what a region costs with the gather on the launch stream (compute + gather) against what it cost pipelined
        region_alone_us = t_nog / reps_nog * 1e6
        pipelined_us = elapsed / repeats * 1e6
        hideable = min(region_alone_us, gather_us)
        overlap_hidden = max(0.0, min(1.0, (region_alone_us + gather_us - pipelined_us) / hideable)) if hideable > 0 else None

    extras = world == 1 and b == BATCH and not args.no_single_step
    bytes_step = algorithmic_bytes_per_step(m, d)
    prof, prof_src = profile_constants()

    def hbm_roofline(seconds, nbytes, traffic=None, kernel=None):
        gbps = nbytes / seconds / 1e9
        r = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBS,
             "traffic": traffic, "algorithmic_bytes_per_launch": int(nbytes), "mean_launch_us": seconds * 1e6}
        if kernel:
            r["kernel"] = kernel
        if traffic:
            r["traffic_source"] = prof_src
        return r

    # ---- the one-launch-per-step variants (rank 0's shard, N=1 only) ----------------------------------------------
    single = api = dense = legal = large = config3 = config5 = overlapped = unbinned = generate = compute_rho = persistent = None
    if extras:
        def episode_stepwise():
            for t in range(EPISODE):
                ops.rollout(state, 1, SEED, step_offset=t, done_count=step_counts[t],
                            initial=fresh if t == 0 else None, **kw)

        g_stepwise = capture(episode_stepwise)
        med, mean, n = timed_replays(g_stepwise.replay, MIN_SECTION_S)
        step_s = med / EPISODE
        single = {"value": b / step_s, "unit": "env-steps/s per GPU", "us_per_step": step_s * 1e6,
                  "steps": n * EPISODE,
                  "roofline": hbm_roofline(step_s, b * bytes_step, prof.get("single_step_bytes_per_launch")),
                  "note": "one launch of the rollout kernel with T=1 (policies in-kernel) + counter reduce per env "
                          "step; state read from and written to HBM every step"}

    def boundary_graph(start, flags=0, coords_as_class=False):
        """one episode of hk_step launches (pre-drawn actions, ping-pong state buffers) as a hipGraph"""
        nb = start.shape[0]
        dd = start.shape[2]
        cls = torch.randint(0, 2 ** dd - dd - 1, (EPISODE, nb), dtype=torch.int32, device="cuda")
        masks = cls if coords_as_class else \
            ops.decode_host_class(cls.reshape(-1), dd, torch.float32).reshape(EPISODE, nb, dd).contiguous()
        axes = torch.randint(0, dd, (EPISODE, nb), dtype=torch.int32, device="cuda")
        bufs = [torch.empty_like(start), torch.empty_like(start)]

        def episode_api():
            src = start
            for t in range(EPISODE):
                ops.step(src, masks[t], axes[t], stages=stages, flags=flags, out=bufs[t & 1], want=("done", "reward"))
                src = bufs[t & 1]

        return capture(episode_api)

    if extras:
        # ---- the drop-in boundary itself: hk_step with the trainer's arrays (state in, [B,d] subset mask and
        # [B] axis in, state + done + reward out) -- exactly SURVEY 8(d)'s 501 algorithmic bytes per env step ----
        g_api = boundary_graph(fresh)
        med, mean, n = timed_replays(g_api.replay, MIN_SECTION_S)
        api_s = med / EPISODE
        # what a plain device-to-device copy reaches on this box (read + write bytes per second)
        big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        dst = torch.empty_like(big)
        cmed, _, _ = timed_replays(lambda: dst.copy_(big), MIN_SECTION_S)
        copy_gbps = 2 * big.numel() * 4 / cmed / 1e9
        del big, dst
        # ... and what a copy of exactly one state batch (15.7 MB in, 15.7 MB out: the traffic of one hk_step)
        # takes as a kernel of its own, replayed from a hipGraph like the steps above
        same = torch.empty_like(fresh)
        g_copy = capture(lambda: [same.copy_(fresh) for _ in range(20)])
        smed, _, _ = timed_replays(g_copy.replay, MIN_SECTION_S)
        del same
        api = {"value": b / api_s, "unit": "env-steps/s per GPU", "us_per_step": api_s * 1e6, "steps": n * EPISODE,
               "roofline": hbm_roofline(api_s, b * bytes_step, prof.get("boundary_step_bytes_per_launch"),
                                        f"hk_step at {b} games (JAX-trainer configuration)"),
               "device_copy_GBps": copy_gbps, "frac_of_device_copy": b * bytes_step / api_s / 1e9 / copy_gbps,
               "state_copy_us": smed / 20 * 1e6,
               "note": "one hk_step launch per env step over an episode from generate_pts states: f32 state + f32 "
                       "[B,d] mask + i32 axis read from HBM, state + done + reward written back; device_copy_GBps = "
                       "a 1 GiB device-to-device copy, state_copy_us = a copy kernel over one state batch (the same "
                       "bytes as one hk_step without the actions and outcomes)"}

        # ---- "single-step from fresh states (all 20 points live -- worst case for the domination
        # test)": hk_step from randint states BEFORE any Newton pass, 20 launches per graph, all from the same input
        fresh_dense = ops.generate_points(b, m, d, MAX_VALUE, seed=43, newton=False, reposition=False)
        live = float(ops.get_num_points(fresh_dense).float().mean().item())
        cls_d = torch.randint(0, 2 ** d - d - 1, (b,), dtype=torch.int32, device="cuda")
        mask_d = ops.decode_host_class(cls_d, d, torch.float32)
        axis_d = torch.randint(0, d, (b,), dtype=torch.int32, device="cuda")
        out_d = torch.empty_like(fresh_dense)
        g_dense = capture(lambda: [ops.step(fresh_dense, mask_d, axis_d, stages=stages, out=out_d,
                                            want=("done", "reward")) for _ in range(EPISODE)])
        med, _, n = timed_replays(g_dense.replay, MIN_SECTION_S)
        dense_s = med / EPISODE
        dense = {"value": b / dense_s, "unit": "env-steps/s per GPU", "us_per_step": dense_s * 1e6,
                 "steps": n * EPISODE, "mean_live_points": live,
                 "roofline": hbm_roofline(dense_s, b * bytes_step),
                 "pair_tests_per_game": m * (m - 1) // 2,
                 "note": "hk_step from generate_points(newton=False) states: every game holds 20 live rows, the "
                         "domination test runs all 190 pairs (rollout states hold ~5)"}
        del fresh_dense, out_d

        # ---- SURVEY 8(d), config 2's second protocol: the agent draws its axis among the host's coordinates
        # only, under the torch and the list sibling's semantics (illegal / finished games not shifted; list:
        # survivors sorted + compacted after every step) -- fused 20-step rollouts and hk_step episodes ----
        legal = {"agent": "uniform over the host's subset (HK_AGENT_RANDOM_LEGAL)"}
        for sem in ("torch", "list"):
            fl = ops.make_flags(sem, noop_if_invalid=True, ignore_ended=True)

            def roll_sem():
                for _ in range(BLOCK):
                    ops.rollout(state, EPISODE, SEED, initial=fresh, stages=stages, flags=fl, defer_counts=True,
                                workspace=count_ws, host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM_LEGAL)
                ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))

            rmed, _, _ = timed_replays(capture(roll_sem).replay, MIN_SECTION_S)
            smed, _, _ = timed_replays(boundary_graph(fresh, flags=fl, coords_as_class=True).replay, MIN_SECTION_S)
            legal[sem] = {"fused_rollout_us_per_episode": rmed / BLOCK * 1e6,
                          "fused_env_steps_per_s": b * EPISODE * BLOCK / rmed,
                          "hk_step_us": smed / EPISODE * 1e6, "hk_step_env_steps_per_s": b * EPISODE / smed,
                          "hk_step_frac_of_hbm_peak": b * bytes_step * EPISODE / smed / 1e9 / HBM_PEAK_GBS}
        done_count.zero_()

        # ---- the reference's benchmark opponent: Zeillinger's host (jax/players.py:55-109; SURVEY f-4) against the
        # random agent on the host's subset -- its choice reads the state, so the policies run inside the step loop
        def roll_zeillinger():
            for _ in range(BLOCK):
                ops.rollout(state, EPISODE, SEED, initial=fresh, stages=stages, defer_counts=True, workspace=count_ws,
                            host_policy=A.HK_HOST_ZEILLINGER, agent_policy=A.HK_AGENT_RANDOM_LEGAL)
            ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))

        zmed, _, _ = timed_replays(capture(roll_zeillinger).replay, MIN_SECTION_S)
        legal["zeillinger_host"] = {"fused_rollout_us_per_episode": zmed / BLOCK * 1e6,
                                    "fused_env_steps_per_s": b * EPISODE * BLOCK / zmed,
                                    "note": "two lanes per game, the pair test split over them (duo_kernel<..., ZEIL>)"}
        done_count.zero_()

        # ---- state generation (generate_pts; 8(d) keeps it out of `value`), the batch binned by live
        # rows on the device (round 4: one launch, local to groups of games; hk_rollout_desc.game_ids keeps every game's
        # policy stream), and the episode on the binned batch ------------------------------------------------------------
        gen_buf = torch.empty_like(fresh)

        def us_per(fn, reps=BLOCK):
            med, _, _ = timed_replays(capture(lambda: [fn(i) for i in range(reps)]).replay, MIN_SECTION_S)
            return med / reps * 1e6
        gen_us = us_per(lambda i: ops.generate_points(b, m, d, MAX_VALUE, seed=42 + i, out=gen_buf))
        raw_us = us_per(lambda i: ops.generate_points(b, m, d, MAX_VALUE, seed=42 + i, out=gen_buf, newton=False,
                                                      reposition=False))
        generate = {"us_per_batch": gen_us, "raw_draws_us_per_batch": raw_us, "games_per_s": b / gen_us * 1e6,
                    "roofline": hbm_roofline(gen_us / 1e6, b * m * d * 4, prof.get("generate_bytes_per_launch"),
                                             f"hk::quadgen_kernel<{m}, {d}> at {b} games"),
                    "note": "hk_generate_points: randint[0, max_value) -> Newton polytope -> reposition (jax/util.py:385-392), "
                            "four lanes per game; algorithmic bytes = the state written once (nothing is read); "
                            "raw_draws = the same launch without the stages (Philox + the store)"}
        genb_us = us_per(lambda i: ops.generate_points_binned(b, m, d, MAX_VALUE, 42 + i, out=gen_buf))
        bin_us = us_per(lambda i: ops.bin_by_live_rows(fresh, out=gen_buf))

        def binned_episodes(**ids):
            if args.no_binned:  # (profile passes: these launches share the headline kernel's name and grid)
                return None

            def run():
                for _ in range(BLOCK):
                    ops.rollout(state, EPISODE, SEED, initial=binned, defer_counts=True, workspace=count_ws, **ids, **kw)
                ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))
            med, _, _ = timed_replays(capture(run).replay, MIN_SECTION_S)
            return med / BLOCK * 1e6
        us_ids, us_pos = binned_episodes(game_ids=game_ids), binned_episodes()
        group, unit = ops.bin_group(m, d)
        unbinned = {"generate_us_per_batch": gen_us, "generate_binned_us_per_batch": genb_us,
                    "binning_us_per_batch": bin_us, "us_per_episode_with_game_ids": us_ids,
                    "us_per_episode_positions_as_ids": us_pos, "group_games": group, "unit_games": unit,
                    "note": "secondary, NOT the headline: hk_generate_points_binned (the generator bins before it "
                            "stores: one launch) / hk_bin_by_live_rows (an existing batch: one launch; round 3 used "
                            "hk_get_num_points + a stable sort + a gather of the tensor library, 85 us host-timed) "
                            "order the games by live rows inside groups of `group_games`, the k-th `unit_games` of "
                            "all groups together; hk_rollout_desc.game_ids keeps every game's policy stream "
                            "(results identical game by game).  At (20,3) the order gains nothing; it pays at "
                            "(50,4): see config3_dim4_50points"}
        done_count.zero_()

        # ---- JAXTrainer.compute_rho's loop (jax_trainer.py:502-555: draw a batch, roll it out, keep the histogram) as ONE
        # launch: the batch is drawn inside the kernel, nothing but the per-step counts leaves it, E loops per launch --------
        rho_ws = ops.rollout_workspace(b, EPISODE, (m, d))
        E = 10
        rho_one = us_per(lambda i: ops.rollout_generated(b, (m, d), EPISODE, SEED, max_value=MAX_VALUE, episodes=E,
                                                         defer_counts=True, workspace=rho_ws, **kw), reps=2) / E
        rho_each = us_per(lambda i: ops.rollout_generated(b, (m, d), EPISODE, SEED + i, max_value=MAX_VALUE, episodes=1,
                                                          defer_counts=True, workspace=rho_ws, **kw))

        def rho_round3(i):
            ops.generate_points(b, m, d, MAX_VALUE, seed=SEED + i, out=gen_buf)
            ops.rollout(gen_buf, EPISODE, SEED + i, defer_counts=True, workspace=rho_ws, **kw)
        rho_two = us_per(rho_round3)
        ops.reduce_counts(rho_ws, done_count, b, EPISODE, (m, d))
        done_count.zero_()
        compute_rho = {"us_per_loop": rho_one, "loops_per_launch": E, "us_per_loop_one_launch_each": rho_each,
                       "us_per_loop_generate_then_rollout": rho_two, "env_steps_per_s": b * EPISODE / rho_one * 1e6,
                       "traffic_per_loop": prof.get("rho_loop_bytes_per_launch"),
                       "note": "rollout.compute_rho by name = ONE hk_rollout launch (hk_rollout_desc.gen_max_value, "
                               "episodes, points = NULL): `us_per_loop` per batch of 65 536 games INCLUDING its generation "
                               "(Philox + the dense first Newton pass), E loops back to back inside the launch -- a wave "
                               "starts its next episode when its own games are finished (the in-kernel form of "
                               "`overlapped_episodes`); us_per_loop_generate_then_rollout = the round-3 form of the loop "
                               "(hk_generate_points + hk_rollout: ~47 MB of state traffic per loop)"}
        del gen_buf

        # ---- E episodes back to back inside ONE launch, from the states resident in memory (hk_rollout_desc.episodes): a
        # wave starts its next episode when its own 16 games are finished -- no launch boundary, nothing waits for the
        # launch's slowest waves (the in-kernel form of `overlapped_episodes`; `value` stays one episode at a time) ------
        pers_us = us_per(lambda i: ops.rollout(state, EPISODE, SEED, initial=fresh, episodes=E, defer_counts=True,
                                               workspace=rho_ws, **kw), reps=2) / E
        ops.reduce_counts(rho_ws, done_count, b, EPISODE, (m, d))
        done_count.zero_()
        persistent = {"episodes_per_launch": E, "us_per_episode": pers_us, "env_steps_per_s": b * EPISODE / pers_us * 1e6,
                      "note": "secondary: hk_rollout with episodes = E and the initial states in memory (episode e: seed "
                              "+ e, every episode reads its slab again; four lanes per game); oracle-exact per episode "
                              "(tests/test_gpu_parity.py::test_episodes_from_resident_states_match_oracle)"}

        # ---- two independent episodes in flight (NOT the headline number): the launch of 65 536 games ends with its
        # slowest waves (mean wave lifetime 15 us inside a 21 us kernel, scripts/probe_timeline.py) and nothing
        # backfills the SIMDs that are done; episodes are independent (the reference's compute_rho loops are), so a
        # second stream with its own state buffer fills the tail with the next episode's waves ----------------------
        if not args.no_overlap:
            state_b = torch.empty_like(fresh)
            count_ws_b = ops.rollout_workspace(b, EPISODE, (m, d))
            s2 = torch.cuda.Stream()

            def two_streams():
                cur = torch.cuda.current_stream()
                s2.wait_stream(cur)
                with torch.cuda.stream(s2):
                    for _ in range(BLOCK):
                        ops.rollout(state_b, EPISODE, SEED + 1, defer_counts=True, workspace=count_ws_b, **head, **kw)
                for _ in range(BLOCK):
                    ops.rollout(state, EPISODE, SEED, defer_counts=True, workspace=count_ws, **head, **kw)
                cur.wait_stream(s2)

            omed, _, _ = timed_replays(capture(two_streams).replay, MIN_SECTION_S)
            ops.reduce_counts(count_ws, done_count, b, EPISODE, (m, d))
            ops.reduce_counts(count_ws_b, done_count, b, EPISODE, (m, d))
            done_count.zero_()
            overlapped = {"episodes_in_flight": 2, "us_per_episode": omed / (2 * BLOCK) * 1e6,
                          "env_steps_per_s": b * EPISODE * 2 * BLOCK / omed,
                          "note": "secondary: two hipGraph branches (two streams, separate state buffers and count "
                                  "workspaces), each a chain of 65 536-game episodes; `value` above is ONE episode at a time"}
            del state_b

        # ---- same kernels at the batch that saturates one GPU (BASELINE configs[3]'s 524 288 games on ONE
        # device): one lane per game means 65 536 games are only 1024 instruction streams for 1024 SIMDs ----
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

        fmed, _, _ = timed_replays(capture(ep_fused).replay, MIN_SECTION_S)
        smed, _, _ = timed_replays(capture(ep_steps).replay, MIN_SECTION_S)
        amed, _, _ = timed_replays(boundary_graph(fresh_l).replay, MIN_SECTION_S)
        large = {"batch": bl, "boundary_step_us": amed / EPISODE * 1e6,
                 "boundary_step_frac_of_hbm_peak": bl * bytes_step * EPISODE / amed / 1e9 / HBM_PEAK_GBS,
                 "fused_env_steps_per_s": bl * EPISODE / fmed, "fused_us_per_episode": fmed * 1e6,
                 "single_step_env_steps_per_s": bl * EPISODE / smed, "single_step_us": smed / EPISODE * 1e6,
                 "single_step_frac_of_hbm_peak": bl * bytes_step * EPISODE / smed / 1e9 / HBM_PEAK_GBS,
                 "note": "not the headline config: shows where the kernels saturate one MI355X"}
        del fresh_l, state_l

        # ---- BASELINE configs[2]: dim 4, 50 points, 262 144 games on one GPU; a parity-test configuration,
        # measured here so that its numbers come from the same run --------
        m3, d3, b3 = 50, 4, 262144
        fresh3 = ops.generate_points(b3, m3, d3, MAX_VALUE, seed=42)
        state3 = torch.empty_like(fresh3)
        dc3 = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")
        cls3 = torch.randint(0, 2 ** d3 - d3 - 1, (b3,), dtype=torch.int32, device="cuda")
        mask3 = ops.decode_host_class(cls3, d3, torch.float32)
        axis3 = torch.randint(0, d3, (b3,), dtype=torch.int32, device="cuda")
        out3 = torch.empty_like(fresh3)
        dense3 = ops.generate_points(b3, m3, d3, MAX_VALUE, seed=43, newton=False, reposition=False)

        def roll3():
            ops.rollout(state3, EPISODE, SEED, done_count=dc3, initial=fresh3, stages=stages,
                        host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM)

        bs3 = algorithmic_bytes_per_step(m3, d3)
        s3, _, _ = timed_replays(capture(lambda: [ops.step(fresh3, mask3, axis3, stages=stages, out=out3,
                                                           want=("done", "reward")) for _ in range(5)]).replay,
                                 MIN_SECTION_S)
        s3d, _, _ = timed_replays(capture(lambda: [ops.step(dense3, mask3, axis3, stages=stages, out=out3,
                                                            want=("done", "reward")) for _ in range(2)]).replay,
                                  MIN_SECTION_S)
        r3, _, _ = timed_replays(capture(roll3).replay, MIN_SECTION_S)
        c3, _, _ = timed_replays(capture(lambda: out3.copy_(fresh3)).replay, MIN_SECTION_S)
        def us3(fn, reps=2):
            med, _, _ = timed_replays(capture(lambda: [fn(i) for i in range(reps)]).replay, MIN_SECTION_S)
            return med / reps * 1e6
        gen3 = us3(lambda i: ops.generate_points(b3, m3, d3, MAX_VALUE, seed=42 + i, out=out3))
        genb3 = us3(lambda i: ops.generate_points_binned(b3, m3, d3, MAX_VALUE, 42 + i, out=out3))
        bin3 = us3(lambda i: ops.bin_by_live_rows(fresh3, out=out3))
        binned3 = None
        if not args.no_binned:  # the same episode on the batch binned by live rows on the device, game ids keep the streams
            b3_pts, b3_ids = ops.bin_by_live_rows(fresh3)

            def roll3_binned():
                ops.rollout(state3, EPISODE, SEED, done_count=dc3, initial=b3_pts, game_ids=b3_ids, stages=stages,
                            host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM)
            rb3, _, _ = timed_replays(capture(roll3_binned).replay, MIN_SECTION_S)
            binned3 = rb3 * 1e6
            del b3_pts, b3_ids
        ws3 = ops.rollout_workspace(b3, EPISODE, (m3, d3))
        rho3 = us3(lambda i: ops.rollout_generated(b3, (m3, d3), EPISODE, SEED, max_value=MAX_VALUE, episodes=4,
                                                   defer_counts=True, workspace=ws3), reps=1) / 4

        def rho3_round3(i):
            ops.generate_points(b3, m3, d3, MAX_VALUE, seed=SEED + i, out=out3)
            ops.rollout(out3, EPISODE, SEED + i, defer_counts=True, workspace=ws3)
        rho3_two = us3(rho3_round3)
        ops.reduce_counts(ws3, dc3, b3, EPISODE, (m3, d3))
        # the other operators of the shape (SURVEY f-2, f-4, a15): observation features, Zeillinger's class
        feat3 = us3(lambda i: ops.get_features(fresh3, out=out3.reshape(b3, m3 * d3)))
        featt3 = us3(lambda i: ops.get_features_torch(fresh3))
        zeil3 = us3(lambda i: ops.zeillinger(fresh3))
        # list semantics (_list_ops.py:9-45: the state leaves the rollout sorted + compacted) and Zeillinger's host
        list_flags = ops.make_flags("list", noop_if_invalid=True)
        list3 = us3(lambda i: ops.rollout(state3, EPISODE, SEED, done_count=dc3, initial=fresh3, stages=stages,
                                          flags=list_flags, agent_policy=A.HK_AGENT_RANDOM_LEGAL), reps=1)
        zroll3 = us3(lambda i: ops.rollout(state3, EPISODE, SEED, done_count=dc3, initial=fresh3, stages=stages,
                                           host_policy=A.HK_HOST_ZEILLINGER), reps=1)
        config3 = {"workload": f"dim={d3}, max_points={m3}, batch={b3} (BASELINE configs[2])",
                   "hk_step_us": s3 / 5 * 1e6, "hk_step_env_steps_per_s": b3 * 5 / s3,
                   "roofline": hbm_roofline(s3 / 5, b3 * bs3, kernel="hk_step at (50,4) x 262144 from generate_pts states"),
                   "hk_step_dense_us": s3d / 2 * 1e6,
                   "hk_step_dense_note": "50 live rows per game: 1225 pair tests per game, VALU-bound",
                   "state_copy_us": c3 * 1e6,
                   "fused_rollout_us_per_episode": r3 * 1e6, "fused_env_steps_per_s": b3 * EPISODE / r3,
                   "fused_rollout_binned_by_live_rows_us_per_episode": binned3,
                   "fused_binned_env_steps_per_s": (b3 * EPISODE / binned3 * 1e6) if binned3 else None,
                   "generate_us_per_batch": gen3,
                   "generate_frac_of_hbm_peak": b3 * m3 * d3 * 4 / gen3 / 1e3 / HBM_PEAK_GBS,
                   "generate_binned_us_per_batch": genb3, "binning_us_per_batch": bin3,
                   "compute_rho_us_per_loop": rho3, "compute_rho_us_per_loop_generate_then_rollout": rho3_two,
                   "get_features_us": feat3, "get_features_torch_us": featt3, "zeillinger_us": zeil3,
                   "fused_rollout_list_semantics_us_per_episode": list3,
                   "fused_rollout_zeillinger_host_us_per_episode": zroll3,
                   "binned_note": "the order of hk_generate_points_binned (groups of 64 games, strata of 16) with the "
                                  "permutation as game ids: the default route for batches of this shape that are rolled "
                                  "out from memory -- generate + bin is one launch; the fused compute_rho loop draws its "
                                  "batches in the kernel",
                   "algorithmic_bytes_per_env_step": bs3}
        del fresh3, out3, state3, dense3

    # ---- BASELINE configs[4]: JAXTrainer.simulate-shaped MCTS self-play (8192 games, 32 simulations per move,
    # 20 moves) through hironaka_amd.trainer_api.HipTrainer; networks are stand-ins (out of scope) ------------
    if extras and not args.no_search:
        from hironaka_amd.trainer_api import HipTrainer, standin_mlp
        cfg = {"eval_batch_size": 8192, "max_num_points": m, "dimension": d, "max_length_game": EPISODE,
               "max_value": MAX_VALUE, "scale_observation": True, "reposition": True, "gumbel_scale": 0.3,
               "num_evaluations": 32, "num_evaluations_as_opponent": 8, "max_num_considered_actions": 10,
               "discount": 0.99}
        host_net, host_params = standin_mlp(m * d, 2 ** d - d - 1, 3)
        agent_net, agent_params = standin_mlp(m * d + d, d, 4)
        def simulate_seconds(role, fused, reps=3):
            trainer = HipTrainer(1, cfg, host_net=host_net, agent_net=agent_net, host_params=host_params,
                                 agent_params=agent_params, use_graph=True, fused_expand=fused)
            out = trainer.simulate(0, role)  # captures one hipGraph per search shape
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(reps):
                out = trainer.simulate(r + 1, role)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps, list(out[0].shape)

        dt5, shape5 = simulate_seconds("host", True)
        dt5_agent, _ = simulate_seconds("agent", True)
        dt5_generic, _ = simulate_seconds("host", False, reps=2)
        env5 = cfg["eval_batch_size"] * cfg["num_evaluations"] * EPISODE
        config5 = {"workload": "HipTrainer.simulate(key, 'host'): dim=3, max_points=20, batch=8192, 32 simulations "
                               "per move, 20 moves (BASELINE configs[4]); opponent = the agent network's argmax",
                   "policy_network": "stand-in: fixed random MLPs 60-256-5 / 63-256-4 behind the HIP feature "
                                     "transform (the reference's flax networks are out of scope)",
                   "launches": "one hipGraph per search (32 simulations), replayed per move; expansions through the "
                               "fused operators (hk_search_expand_gather, hk_step_features: the step with the agent's "
                               "logits as its axis and the features of its result; node-major tables: no scatter)",
                   "seconds_per_simulate": dt5, "env_steps_in_search_per_s": env5 / dt5,
                   "searches_per_s": cfg["eval_batch_size"] * EPISODE / dt5, "samples": shape5,
                   "seconds_per_simulate_agent_role": dt5_agent,
                   "seconds_per_simulate_generic_expansion": dt5_generic,
                   "generic_expansion_note": "the same call with the expansions as tensor-library glue around hk_step "
                                             "(HipTrainer(fused_expand=False)): identical rollouts",
                   "hip_kernel_share_of_gpu_time": prof.get("search_hip_kernel_share"),
                   "hip_kernel_share_source": prof_src if prof.get("search_hip_kernel_share") is not None else None}

    if rank == 0:
        launch_s = median_region_s / max(1, launches_per_region)  # per rollout launch (+ its share of the reduce)
        steps_per_launch = K / max(1, launches_per_region)
        alg_bytes = b * bytes_step * steps_per_launch
        full = b == BATCH and rem == 0
        traffic = prof.get("rollout_T20_bytes_per_launch") if full else None
        valu = prof.get("rollout_T20_valu_insts_per_launch") if full else None
        roofline = {
            # the fused kernel keeps the state in registers for 20 steps: what binds it is instruction issue
            "bound": "valu_issue",
            "kernel": f"fused rollout kernel, {b} games x {EPISODE} steps per launch, two lanes per game",
            "achieved": (valu / launch_s) if valu else None,
            "peak": VALU_PEAK_WAVE_INSTS,
            "unit": "wave64 VALU instructions/s",
            "frac": (valu / launch_s / VALU_PEAK_WAVE_INSTS) if valu else None,
            "valu_insts_per_launch": valu,
            "traffic": traffic,
            "traffic_source": prof_src,
            "hbm_frac_measured": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "algorithmic_GBps": alg_bytes / launch_s / 1e9,
            "algorithmic_frac": alg_bytes / launch_s / 1e9 / HBM_PEAK_GBS,
            "mean_launch_us": launch_s * 1e6,
            "note": "peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction; valu_insts_per_launch "
                    "(SQ_INSTS_VALU) and traffic (2 x FETCH_SIZE + WRITE_SIZE) come from the committed rocprofv3 "
                    "PMC passes; algorithmic_* = SURVEY 8(d)'s 501 B/env-step x games x steps (may exceed 1: the "
                    "state stays in registers between steps, real traffic is ~2 x 240 B per game per launch); the "
                    "HBM-bound kernel of this path is hk_step: see boundary_step.roofline",
        }
        out = {
            "metric": "env-steps/sec at dim=3, max_pts=20, batch=65536; 1/2/4/8 GPUs",
            "value": world * b * K * repeats / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / (K * repeats) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "repeats": repeats,
            "timed_steps": K * repeats,
            "timed_seconds": elapsed,
            "ms_per_step_median": median_region_s / K * 1e3,
            "value_from_median": world * b * K / median_region_s,
            "config": {
                "workload": f"dim={d}, max_points={m}, batch={b} games per GPU (BASELINE configs[1]), random "
                            f"host+agent policies sampled in-kernel, reposition=True, rescale=False, episodes of "
                            f"{EPISODE} steps from generate_pts states (max_value={MAX_VALUE}"
                            + ("; games ordered by live rows at generate time, game ids keep every game's policy stream"
                               if args.binning else "") + f"); fused rollout: "
                            f"{EPISODE} env steps per launch",
                "parallelism": f"{world} x independent game shards" + (
                    f"; after EVERY {K}-step region one all-gather of the final states of all ranks "
                    f"({os.environ.get('HK_BENCH_GATHER', 'rccl')}), pipelined behind the next region on a side "
                    f"stream, inside the timed region" if distributed else ""),
                "launch": f"hipGraph replays: one rollout kernel per episode, one counter reduce per "
                          f"{reduce_every * launches_per_region} episodes, {group} region(s) per graph; the K-step "
                          f"region repeated {repeats}x inside one barrier+synchronize bracket",
            },
            "roofline": roofline,
            "games_finished_per_episode": finished,
        }
        if gather_us is not None:
            out["gather_us"] = gather_us
            out["gather_bytes_per_rank"] = int(world * b * m * d * 4)
            out["value_without_gather"] = value_without_gather
            out["overlap_hidden_frac"] = overlap_hidden
            out["gather_check"] = "the pipelined gather's result equals the serial gather of the same states"
            out["gather_note"] = ("`value`: after EVERY region its final states are all-gathered over the ranks, on a "
                                  "side stream behind the next region (hkdist.GatherPipeline, two state buffers); "
                                  "`value_without_gather`: the same regions alone; `gather_us`: the gather serialised "
                                  "on the launch stream, HIP-event time, median, max over ranks; overlap_hidden_frac = "
                                  "(region + gather - pipelined region) / min(region, gather).  A 20 us episode cannot "
                                  "hide a gather of (N - 1) x 15.7 MB: at N > 1 `value` is the gather's rate")
        for key, val in (("single_step", single), ("boundary_step", api), ("single_step_dense", dense),
                         ("generate", generate), ("compute_rho", compute_rho), ("persistent_episodes", persistent),
                         ("binned_by_live_rows", unbinned), ("overlapped_episodes", overlapped), ("large_batch", large),
                         ("legal_axis_torch_list_semantics", legal),
                         ("config3_dim4_50points", config3), ("config5_mcts_simulate", config5)):
            if val is not None:
                out[key] = val
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_array_formulation"] = cpu_array_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(seconds: float = 12.0):
    """C/OpenMP oracle on the host cores: the same episodes (64k games x 20 steps), repeated
    until `seconds` of CPU work were timed."""
    from hironaka_amd import _abi as A
    from oracle import c_oracle as CO
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


def cpu_array_baseline(budget_s: float = 10.0):
    """SURVEY 8(d) CPU path (i): the reference's own formulation of the step -- broadcast [B,m,m,d] difference
    tensors, masks, any/all, as in _jax_ops.py / _torch_ops.py -- on torch-CPU with all cores
    (oracle/torch_cpu_oracle.py), the headline batch, step by step until the budget is spent (at most one
    episode)."""
    import torch
    from oracle import np_oracle as NO
    from oracle import torch_cpu_oracle as TO
    threads = torch.get_num_threads()
    p = torch.from_numpy(NO.generate_points(BATCH, MAX_POINTS, DIM, MAX_VALUE, 42))
    table = torch.from_numpy(NO.decode_table(DIM)).float()
    TO.step(p[:4096], table[torch.zeros(4096, dtype=torch.long)], torch.zeros(4096, dtype=torch.long))  # warm-up
    steps, spent = 0, 0.0
    while steps < EPISODE and spent < budget_s:
        cls, ax = NO.policy_actions(p.numpy(), steps, SEED, 0, NO.HOST_RANDOM, NO.AGENT_RANDOM)
        coords, axis = table[torch.from_numpy(cls).long()], torch.from_numpy(ax).long()
        t0 = time.perf_counter()
        p = TO.step(p, coords, axis)
        spent += time.perf_counter() - t0
        steps += 1
    return {"value": BATCH * steps / spent, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "os_cpu_count": os.cpu_count(), "torch_num_threads": threads,
            "sample": f"the first {steps} steps of one episode of {BATCH} games ({spent:.1f} s) with "
                      f"oracle/torch_cpu_oracle.py (torch CPU tensor ops, the reference's broadcast-tensor "
                      f"formulation, {threads} threads)"}


if __name__ == "__main__":
    main()
