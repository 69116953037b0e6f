#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched Hironaka-game step on MI355X.

Workload (BASELINE.json configs[1]): dim=3, max_points=20, batch=65 536 games per GPU, float32,
max_value=20, reposition on / rescale off (hironaka/jax/jax_config.yml), uniformly random host
class and agent axis sampled INSIDE the kernel (Philox), episodes of T=20 steps from freshly
generated states (generate_pts: randint -> newton -> reposition).

One "step" = one pass of the hot path over the whole per-GPU batch = ONE kernel launch that reads
the state from HBM, chooses the host subset and the agent axis, does shift -> reposition ->
Newton polytope -> done, counts finished games and writes the state back to HBM.  Every 20 steps
the episode restarts from the resident fresh states (a device-to-device copy, inside the timed
region).  The launches of an episode are captured once into a hipGraph and replayed, so the
python interpreter is not in the timed loop.  K steps are timed exactly (K//20 replays of the
episode graph + one graph of K%20 steps).

N>1: one process per GPU (torch.distributed, backend nccl = RCCL), games sharded by rank
(game_offset = rank*batch), no collective in the data path; one all-gather of the final states at
the end of the timed region (the trainer boundary).  value = steps of ALL ranks / max time.

Also reported on the same JSON line:
  roofline      dominant kernel (fast_kernel<20,3>): algorithmic bytes per launch / mean launch
                time from HIP events around the timed region
  cpu_baseline  the scalar C/OpenMP oracle (oracle/hironaka_oracle.c) on this box's host cores,
                same workload, bounded sample (rank 0, N=1 only)
  fused_rollout the same 20-step episodes as ONE launch each (state stays in registers)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

DIM, MAX_POINTS, BATCH, MAX_VALUE, EPISODE = 3, 20, 65536, 20, 20
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_step(m: int, d: int) -> int:
    """read state + write state + action pair + reward + done."""
    return 2 * m * d * 4 + 4 * d + 4 + 4 + 1


def capture_episode(ops, A, state, fresh, done_count, n_steps, seed, game_offset, stages):
    """hipGraph of: state <- fresh ; n_steps x (one-step launch with in-kernel policies)."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        state.copy_(fresh)
        for t in range(n_steps):
            ops.rollout(state, 1, seed, game_offset=game_offset, step_offset=t, stages=stages,
                        host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM,
                        done_count=done_count[t:t + 2])
    return g


def capture_fused(ops, A, state, fresh, done_count, n_steps, seed, game_offset, stages):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        state.copy_(fresh)
        ops.rollout(state, n_steps, seed, game_offset=game_offset, stages=stages,
                    host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM,
                    done_count=done_count[: n_steps + 1])
    return g


def cpu_baseline(seconds: float = 12.0):
    """C/OpenMP oracle on the host cores: same episodes (64k games x 20 steps), repeated until
    `seconds` of CPU work were timed."""
    from oracle import c_oracle as CO
    from hironaka_amd import _abi as A
    threads = CO.set_threads(0)
    fresh = CO.generate_points(BATCH, MAX_POINTS, DIM, MAX_VALUE, 42)
    CO.rollout(fresh[:4096], EPISODE, 7, record=False)  # warm-up
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        CO.rollout(fresh, EPISODE, 7, record=False,
                   stages=A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON)
        done += BATCH * EPISODE
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{done // (BATCH * EPISODE)} episodes of {BATCH} games x {EPISODE} steps "
                      f"({dt:.1f} s) with oracle/hironaka_oracle.c (scalar C + OpenMP, {threads} threads "
                      f"of {os.cpu_count()} cores)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=BATCH, help="games per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from hironaka_amd import _abi as A
    from hironaka_amd import ops

    b, m, d = args.batch, MAX_POINTS, DIM
    stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    game_offset = rank * b
    fresh = ops.generate_points(b, m, d, MAX_VALUE, seed=42, game_offset=game_offset)
    state = torch.empty_like(fresh)
    done_count = torch.zeros(EPISODE + 1, dtype=torch.int64, device="cuda")
    gathered = [torch.empty_like(state) for _ in range(world)] if distributed else None
    K, W = args.steps, args.warmup
    n_full, rem = divmod(K, EPISODE)

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g_episode = capture_episode(ops, A, state, fresh, done_count, EPISODE, 7, game_offset, stages)
        g_rem = capture_episode(ops, A, state, fresh, done_count, rem, 7, game_offset, stages) if rem else None
        g_fused = capture_fused(ops, A, state, fresh, done_count, EPISODE, 7, game_offset, stages)
    torch.cuda.synchronize()

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def run(n_full_, g_rem_):
        for _ in range(n_full_):
            g_episode.replay()
        if g_rem_ is not None:
            g_rem_.replay()

    # ---- warm-up (untimed) -------------------------------------------------------------------
    run(max(1, W // EPISODE), None)
    if distributed:
        dist.all_gather(gathered, state)
    barrier()

    # ---- exactly K timed steps ---------------------------------------------------------------
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    done_count.zero_()
    barrier()
    t0 = time.perf_counter()
    ev0.record()
    run(n_full, g_rem)
    ev1.record()
    if distributed:
        dist.all_gather(gathered, state)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)  # HIP events on the launch stream, around the K launches
    if distributed:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finished = int(done_count[EPISODE].item()) if n_full else 0

    # ---- fused-rollout variant (same episodes, one launch each), timed separately ---------
    g_fused.replay()
    barrier()
    f0 = time.perf_counter()
    for _ in range(max(1, n_full)):
        g_fused.replay()
    torch.cuda.synchronize()
    fused_elapsed = time.perf_counter() - f0
    fused_steps = max(1, n_full) * EPISODE

    if rank == 0:
        bytes_step = algorithmic_bytes_per_step(m, d)
        launch_s = (kernel_ms / 1e3) / K
        achieved = b * bytes_step / launch_s / 1e9
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
                "workload": f"dim={d}, max_points={m}, batch={b} games per GPU (configs[1]), random host+agent "
                            f"policies sampled in-kernel, reposition=True, rescale=False, episodes of {EPISODE} steps "
                            f"from generate_pts states (max_value={MAX_VALUE}); 1 step = 1 launch, state through HBM",
                "parallelism": f"{world} x independent game shards, all-gather of final states",
                "launch": "hipGraph replay of 20-step episodes",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": f"hk::fast_kernel<{m},{d}> (one env step of {b} games per launch)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": b * bytes_step,
                "mean_launch_us": launch_s * 1e6,
            },
            "fused_rollout": {
                "value": world * b * fused_steps / fused_elapsed,
                "unit": "env-steps/s",
                "note": f"{EPISODE} steps per launch, state in registers; per-GPU shard timed on rank 0",
            },
            "games_finished_per_episode": finished // max(1, n_full),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
