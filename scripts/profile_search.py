"""Workload for the search profile (scripts/profile_round.sh): BASELINE configs[4] -- HipTrainer.simulate with 8192
games, 32 simulations per move, 20 moves, stand-in networks -- a warm-up call (graph capture) and two timed calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hironaka_amd.trainer_api import HipTrainer, standin_mlp

m, d = 20, 3
cfg = {"eval_batch_size": 8192, "max_num_points": m, "dimension": d, "max_length_game": 20, "max_value": 20,
       "scale_observation": True, "reposition": True, "gumbel_scale": 0.3, "num_evaluations": 32,
       "num_evaluations_as_opponent": 8, "max_num_considered_actions": 10, "discount": 0.99}
host_net, host_params = standin_mlp(m * d, 2 ** d - d - 1, 3)
agent_net, agent_params = standin_mlp(m * d + d, d, 4)
trainer = HipTrainer(1, cfg, host_net=host_net, agent_net=agent_net, host_params=host_params,
                     agent_params=agent_params, use_graph=True)
for r in range(3):
    trainer.simulate(r, "host")
torch.cuda.synchronize()
