"""BASELINE configs[4]: the MCTS `simulate()` loop (dim 3, 20 points, 32 simulations per move, 8192 games,
20 moves) driving the HIP environment on one MI355X.  Networks are out of scope of this build: the policy /
value function is a fixed random two-layer MLP (stand-in, stated in the output); the opponent is the
uniformly random agent.  Prints one JSON line: env steps inside the search per second, and the time split
between the tree kernels, the environment step and the rest (torch glue + the stand-in network).

    python scripts/bench_search.py [--batch 8192] [--sims 32] [--moves 20]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hironaka_amd.functional import generate_pts, get_reward_fn
from hironaka_amd.players import random_agent_fn
from hironaka_amd.simulation_fn import get_evaluation_loop, get_simulation


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--sims", type=int, default=32)
    ap.add_argument("--moves", type=int, default=20)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--eager", action="store_true", help="issue every launch from Python instead of replaying a hipGraph")
    args = ap.parse_args()
    m, d = 20, 3
    spec = (m, d)
    ncls = 2 ** d - d - 1
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn(m * d, 256, generator=g) / (m * d) ** 0.5).cuda()
    w2 = (torch.randn(256, ncls + 1, generator=g) / 16.0).cuda()

    def policy_fn(obs, *a, key=None, **kw):
        h = torch.relu((obs.clamp(min=-1.0) / 20.0) @ w1) @ w2
        return h[:, :ncls].contiguous(), torch.tanh(h[:, ncls]).contiguous()

    opponent = lambda obs, *a, key=0, **kw: random_agent_fn(obs, spec, key=None)  # default generator: capturable
    ev = get_evaluation_loop("host", policy_fn, opponent, get_reward_fn("host"), spec, num_evaluations=args.sims,
                             max_depth=args.moves, max_num_considered_actions=ncls, discount=0.99,
                             rescale_points=False, reposition=True, use_graph=not args.eager)
    sim = get_simulation("host", ev, args.batch, m, d, args.moves)
    root = generate_pts(1, (args.batch, m, d), 20, torch.float32, False, True).reshape(args.batch, m * d)
    sim(0, root)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(args.reps):
        obs, logp, value = sim(r + 1, root)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    env_steps = args.batch * args.sims * args.moves
    print(json.dumps({
        "workload": f"simulate(): dim={d}, max_points={m}, batch={args.batch}, {args.sims} simulations/move, "
                    f"{args.moves} moves (BASELINE configs[4]); host role vs random agent",
        "policy_network": "stand-in: fixed random MLP 60-256-5 (the reference's networks are out of scope)",
        "launches": "eager" if args.eager else "one hipGraph per search (32 simulations)",
        "seconds_per_simulate": dt,
        "env_steps_in_search_per_s": env_steps / dt,
        "searches_per_s": args.batch * args.moves / dt,
        "samples": list(obs.shape),
    }))


if __name__ == "__main__":
    main()
