"""Dev probe: BASELINE configs[4] (HipTrainer.simulate, 8192 games x 32 simulations x 20 moves, one hipGraph per
search) with the expansions through the fused operators and through the generic tensor-library glue."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd.trainer_api import HipTrainer, standin_mlp

m, d = 20, 3
cfg = {"eval_batch_size": 8192, "max_num_points": m, "dimension": d, "max_length_game": 20, "max_value": 20,
       "scale_observation": True, "reposition": True, "gumbel_scale": 0.3, "num_evaluations": 32,
       "num_evaluations_as_opponent": 8, "max_num_considered_actions": 10, "discount": 0.99}
for fused in (True, False):
    host_net, host_params = standin_mlp(m * d, 2 ** d - d - 1, 3)
    agent_net, agent_params = standin_mlp(m * d + d, d, 4)
    t = HipTrainer(1, cfg, host_net=host_net, agent_net=agent_net, host_params=host_params, agent_params=agent_params,
                   use_graph=True, fused_expand=fused)
    for role in ("host", "agent"):
        t.simulate(0, role)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            t0 = time.perf_counter()
            t.simulate(r + 1, role)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print(f"fused_expand={fused} role={role}: simulate() median {ts[2]*1e3:.1f} ms  min {ts[0]*1e3:.1f} ms "
              f"({8192*20*32/ts[2]/1e6:.1f} M search env-steps/s)", flush=True)
