"""``HipTrainer`` -- the environment side of ``hironaka.jax.JAXTrainer`` behind the reference's own entry
points and config keys (hironaka/jax/jax_trainer.py:150-320, 398-592, 626-876):

    trainer = HipTrainer(key, config, host_net=..., agent_net=..., host_params=..., agent_params=...)
    obs, target_policy, target_value = trainer.simulate(key, role, use_mcts_policy=False, use_unified_tree=False)
    rho, details = trainer.compute_rho(host, agent, batch_size=..., num_of_loops=10, max_length=...)

What is NOT here is the neural-network side (flax models, optax optimisers, ``train``, checkpoints, wandb) and the
trainer's control plane around it (``validate``'s battle schedule, the ``get_* / update_*`` accessor layer of
jax_trainer.py:626-669,735-858): out of scope.  The two networks therefore arrive as callables
``net(features, params) -> (policy_logits [B, A], value [B])`` together with their parameter objects
(``host_params`` / ``agent_params``: opaque, handed back to the callables) -- the role ``model.apply`` plays in
``get_apply_fn`` (hironaka/jax/net.py:110-133); the feature transform in front of them (rescale + row sort,
jax/util.py:172-214) is the HIP operator ``hk_get_features``.  Everything around the networks -- root-state
generation, the agent-role observation build (host policy -> decode -> concat), the zero tail of the unified
tree, the Gumbel-MuZero search per move over the HIP environment, ``rollout_postprocess`` and the cross-rank
gather -- runs here on the device.

Differences forced by the platform (same as ``functional.py``): PRNG keys are integer seeds; there is no
``pmap``: one process per GPU, ``eval_batch_size`` games per process, the Philox counters carry the global game
index (``rank * eval_batch_size + i``), and ``simulate`` ends with one all-gather of the rollout over the ranks
(``distributed.all_gather_rollout`` -- the reference keeps a leading device axis instead,
jax_trainer.py:281-319).
"""
from __future__ import annotations

import time
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch

from . import distributed as hkdist
from .functional import (action_wrapper, apply_agent_action_mask, flatten, generate_pts, get_feature_fn,
                         get_reward_fn, mcts_wrapper)
from .host_action_preprocess import get_batch_decode_from_one_hot
from .players import get_host_with_flattened_obs
from .rollout import compute_rho as _compute_rho
from .recurrent_fn import AgentExpander, HostExpander
from .rollout import rollout_postprocess as _rollout_postprocess
from .simulation_fn import get_evaluation_loop, get_simulation

# jax_trainer.py:162-190 -- the keys JAXTrainer reads from its config
CONFIG_KEYS = ["eval_batch_size", "max_num_points", "dimension", "max_length_game", "max_value", "max_grad_norm",
               "scale_observation", "reposition", "gumbel_scale", "use_cuda", "version_string", "net_type",
               "num_evaluations", "num_evaluations_as_opponent", "eval_on_cpu", "max_num_considered_actions",
               "discount"]
# keys that only configure the network / optimiser side: accepted, stored, not used here
_NN_ONLY_KEYS = {"max_grad_norm": 1.0, "use_cuda": True, "version_string": "", "net_type": "dense", "eval_on_cpu": False}


def _split(key: int, n: int = 2):
    return tuple((int(key) * 6364136223846793005 + 1442695040888963407 * (i + 1)) % (1 << 63) for i in range(n))


def get_apply_fn(role: str, model: Callable, spec: Tuple[int, int], feature_fn=None) -> Callable:
    """hironaka/jax/net.py:110-133 -- `apply_fn(x, params, *args, **kwargs) -> (policy_prior, value_prior)`:
    the feature function (default: jax/util.py:172-214 through hk_get_features) then the network
    ``model(features, params)``."""
    if feature_fn is None:
        feature_fn = get_feature_fn(role, spec)

    def apply_fn(x: torch.Tensor, params, *args, **kwargs):
        policy_prior, value_prior = model(feature_fn(x), params)
        return policy_prior, value_prior.reshape(policy_prior.shape[0])

    apply_fn.__name__ = f"{role}_apply_fn"
    return apply_fn


def standin_mlp(in_dim: int, out_dim: int, seed: int, width: int = 256, device="cuda") -> Tuple[Callable, Any]:
    """A fixed random two-layer network ``(net, params)`` with ``net(features, params) -> (logits, value)`` -- a
    stand-in for the reference's flax models (out of scope) so that the search / simulate path can be exercised
    and timed."""
    g = torch.Generator().manual_seed(seed)
    params = ((torch.randn(in_dim, width, generator=g) / in_dim ** 0.5).to(device),
              (torch.randn(width, out_dim + 1, generator=g) / width ** 0.5).to(device))

    def net(features, params):
        w1, w2 = params
        h = torch.relu(features.clamp(min=-1.0) @ w1) @ w2
        return h[:, :out_dim].contiguous(), torch.tanh(h[:, out_dim]).contiguous()

    net.__name__ = f"standin_mlp_{in_dim}_{out_dim}"
    return net, params


class PendingRollout:
    """simulate(..., overlap_gather=True): the gathered (obs, target_policy, target_value), still travelling"""

    def __init__(self, pipes, tickets):
        self._pipes, self._tickets = pipes, tickets

    def result(self):
        return tuple(pipe.result(t) for pipe, t in zip(self._pipes, self._tickets))


class HipTrainer:
    """See the module docstring.  ``simulate`` / ``compute_rho`` / ``rollout_postprocess`` and the config keys follow
    JAXTrainer; the search loops behind ``simulate`` are built on first use, one per (role, kind of tree)."""

    def __init__(self, key: int, config: Union[dict, str], dtype=torch.float32,
                 host_net: Optional[Callable] = None, agent_net: Optional[Callable] = None,
                 host_params: Any = None, agent_params: Any = None,
                 host_feature_fn=None, agent_feature_fn=None, device=None, use_graph: bool = False,
                 fused_expand: bool = True):
        if isinstance(config, str):
            import yaml
            with open(config, "r") as stream:
                self.config = yaml.safe_load(stream)
        elif isinstance(config, dict):
            self.config = config
        else:
            raise TypeError(f"config must be either a string or a dict. Got {type(config)}.")
        for config_key in CONFIG_KEYS:
            if config_key in _NN_ONLY_KEYS:
                setattr(self, config_key, self.config.get(config_key, _NN_ONLY_KEYS[config_key]))
            else:
                setattr(self, config_key, self.config[config_key])  # KeyError like the reference
        self.dtype = dtype
        self.device = torch.device("cuda" if device is None else device)
        if self.device.type != "cuda":
            raise TypeError("HipTrainer lives on a HIP device (there is no CPU path)")
        if host_net is None or agent_net is None:
            raise ValueError("HipTrainer needs host_net and agent_net: net(features, params) -> (logits [B, A], "
                             "value [B]).  The reference builds flax networks here (net_type / net_arch); networks "
                             "are outside this package (trainer_api.standin_mlp gives a fixed random stand-in).")
        self.use_graph = use_graph
        # trees of the trainer's own policies expand through the fused operators (recurrent_fn.HostExpander /
        # AgentExpander) -- possible while the standard feature functions sit in front of the networks
        self.fused_expand = bool(fused_expand) and host_feature_fn is None and agent_feature_fn is None
        self.key = int(key)
        self.spec = (self.max_num_points, self.dimension)
        m, d = self.spec
        self.input_dim = {"host": m * d, "agent": m * d + d}
        self.output_dim = {"host": 2 ** d - d - 1, "agent": d}
        self.device_num = hkdist.world()  # one process per GPU
        self.models = {"host": host_net, "agent": agent_net}
        self.host_params, self.agent_params = host_params, agent_params
        feature_fns = {"host": host_feature_fn, "agent": agent_feature_fn}
        self.policy_fns: Dict[str, Callable] = {}
        for role in ("host", "agent"):
            feature_fn = feature_fns[role] if feature_fns[role] is not None else \
                get_feature_fn(role, self.spec, scale_observation=self.scale_observation)
            fn = get_apply_fn(role, self.models[role], self.spec, feature_fn=feature_fn)
            # (the agent's logits go through its action mask, jax_trainer.py:836-855, in its NaN-free form: the
            # reference's `policy * mask - inf * (~mask)`, jax/util.py:302, is inf * 0 = NaN on the allowed logits)
            self.policy_fns[role] = apply_agent_action_mask(fn, d, nan_free=True) if role == "agent" else fn
        self.reward_fns = {role: get_reward_fn(role) for role in ("host", "agent")}
        self._sim_fns: Dict[Any, Callable] = {}   # (role, mcts opponent, unified tree) -> simulation
        self._fn_args: Dict[Any, Any] = {}        # (role, unified tree) -> (params objects, argument tuples)
        self._gather_pipes = None                 # simulate(overlap_gather=True): one GatherPipeline per output tensor
        self.log: Dict[str, Any] = {}

    @staticmethod
    def _opponent(role: str) -> str:
        if role not in ("host", "agent"):
            raise ValueError(f"role must be either host or agent. Got {role}.")
        return "agent" if role == "host" else "host"

    def params(self, role: str):
        return self.host_params if role == "host" else self.agent_params

    # ---- the search loops (jax/simulation_fn.py:16-213 as JAXTrainer configures them, jax_trainer.py:735-834) ----------
    def _eval_loop(self, role: str, *, gumbel_scale: float, num_evaluations: int, opponent_fn: Callable,
                   own_policies: bool, unified: bool = False) -> Callable:
        common = dict(reward_fn=self.reward_fns[role], num_evaluations=num_evaluations, spec=self.spec,
                      max_depth=self.max_length_game, max_num_considered_actions=self.max_num_considered_actions,
                      discount=self.discount, rescale_points=False,  # rescaling lives in the feature functions
                      reposition=self.reposition, dtype=self.dtype, gumbel_scale=gumbel_scale)
        if unified:  # one tree for both players: host policy on the points, agent policy below it; the agent's reward
            host_on_points = lambda pts, *a, dtype=None, **k: self.policy_fns["host"](pts.reshape(pts.shape[0], -1), *a, **k)
            return get_evaluation_loop(role="host", role_agnostic=True, use_graph=False, expander=None,
                                       policy_fn=get_host_with_flattened_obs(self.spec, host_on_points, truncate_input=True),
                                       opponent_fn=self.policy_fns["agent"],
                                       **{**common, "reward_fn": self.reward_fns["agent"]})
        expander = None
        if own_policies and self.fused_expand and self.dtype == torch.float32:
            expander = (HostExpander if role == "host" else AgentExpander)(
                self.models["host"], self.models["agent"], self.spec, self.discount, self.scale_observation,
                self.reposition, rescale_points=False, reward_sign=getattr(self.reward_fns[role], "hk_reward_sign"))
        # one hipGraph per search: only where nothing inside synchronises with the host (an opponent that is itself a
        # search does)
        return get_evaluation_loop(role=role, policy_fn=self.policy_fns[role], opponent_fn=opponent_fn,
                                   use_graph=self.use_graph and own_policies, expander=expander, **common)

    def _sim_fn(self, role: str, use_mcts_policy: bool, use_unified_tree: bool) -> Callable:
        sim_key = (role, bool(use_mcts_policy), bool(use_unified_tree))
        if sim_key not in self._sim_fns:
            opponent = self._opponent(role)
            sim_cfg = dict(eval_batch_size=self.eval_batch_size, max_num_points=self.max_num_points,
                           dimension=self.dimension, max_length_game=self.max_length_game, dtype=self.dtype)
            if use_mcts_policy:  # the opponent answers with a (smaller) search of its own
                opp_loop = self._eval_loop(opponent, gumbel_scale=0.0, num_evaluations=self.num_evaluations_as_opponent,
                                           opponent_fn=action_wrapper(self.policy_fns[role], None), own_policies=True)
                opp_policy = mcts_wrapper(opp_loop)
                if opponent == "agent":
                    opp_policy = apply_agent_action_mask(opp_policy, self.dimension, nan_free=True)
                loop = self._eval_loop(role, gumbel_scale=self.gumbel_scale, num_evaluations=self.num_evaluations,
                                       opponent_fn=action_wrapper(opp_policy, None), own_policies=False)
            elif use_unified_tree:
                loop = self._eval_loop(role, gumbel_scale=self.gumbel_scale, num_evaluations=self.num_evaluations,
                                       opponent_fn=None, own_policies=False, unified=True)
            else:  # the opponent's network gives its definitive action (argmax as a one-hot array)
                loop = self._eval_loop(role, gumbel_scale=self.gumbel_scale, num_evaluations=self.num_evaluations,
                                       opponent_fn=action_wrapper(self.policy_fns[opponent], None), own_policies=True)
            self._sim_fns[sim_key] = get_simulation("host" if use_unified_tree else role, loop, **sim_cfg)
        return self._sim_fns[sim_key]

    # ---- jax_trainer.py:247-320 ------------------------------------------------------------------------------
    def simulate(self, key: int, role: str, use_mcts_policy=False, use_unified_tree=False, overlap_gather=False):
        """One batch of self-play: ``eval_batch_size`` games per process, ``max_length_game`` moves, every move
        chosen by a Gumbel-MuZero search of ``num_evaluations`` simulations over the HIP environment.
        Returns (obs [B*T, input_dim], target_policy [B*T, A], target_value [B*T]) with B the games of ALL
        ranks (gathered at the end), values replaced by the ground truth of the finished games
        (``rollout_postprocess``).

        A search captured into a hipGraph (``use_graph``) replays with the parameter OBJECTS it was captured with:
        update ``host_params`` / ``agent_params`` in place (``tensor.copy_``) to keep the capture; assigning new
        objects drops it and captures again on the next call (one capture per role and kind of tree is kept).

        overlap_gather (more than one rank): return a `PendingRollout` instead -- the all-gather of the three tensors
        runs on a side stream behind the next simulate's searches; `.result()` is the tuple above (valid until two
        further simulate calls)."""
        opponent = self._opponent(role)
        sim_fn = self._sim_fn(role, use_mcts_policy, use_unified_tree)
        if use_unified_tree:
            role_params, opp_params = self.host_params, self.agent_params
        else:
            role_params, opp_params = self.params(role), self.params(opponent)

        shard = hkdist.shard_range(self.eval_batch_size * hkdist.world())
        root_key, sim_key, host_key = _split(key, 3)
        # rescale is off: rescaling is part of the feature functions (jax_trainer.py:279-288)
        root_state = generate_pts(root_key, (self.eval_batch_size, self.max_num_points, self.dimension), self.max_value,
                                  self.dtype, False, self.reposition, game_offset=shard.start, device=self.device)
        if role == "agent":
            # host coordinates from the host network, decoded and appended: agent observations
            coords, _ = self.policy_fns["host"](flatten(root_state), self.host_params, key=host_key)
            coordinate_mask = get_batch_decode_from_one_hot(self.dimension)(coords, self.dtype)
            root_state = torch.cat([flatten(root_state), coordinate_mask], dim=-1)
        else:
            root_state = flatten(root_state)
            if use_unified_tree:  # pad zeros to the length of an agent observation
                root_state = torch.cat([root_state, torch.zeros((self.eval_batch_size, self.dimension),
                                                                dtype=self.dtype, device=self.device)], dim=-1)
        # one pair of argument tuples per (role, kind of tree): a captured search is keyed by their identity, so the
        # tuples live as long as the parameter objects do and are replaced (not accumulated) when those change
        args_key = (role, bool(use_unified_tree))
        cached = self._fn_args.get(args_key)
        if cached is None or cached[0] is not role_params or cached[1] is not opp_params:
            cached = (role_params, opp_params, (role_params,),
                      (opp_params,) if use_unified_tree else (opp_params, role_params))
            self._fn_args[args_key] = cached
        simulate_output = sim_fn(sim_key, root_state, cached[2], cached[3])
        out = self.rollout_postprocess(simulate_output, role, use_unified_tree)
        # (the rollout is flattened to [B * T, ...] rows: max_length_game rows per game)
        if not overlap_gather or hkdist.world() == 1:
            return hkdist.all_gather_rollout(out, shard, rows_per_game=self.max_length_game)
        # the trainer-boundary gather on a side stream (hkdist.GatherPipeline): this call returns at once, the NEXT
        # simulate's searches run while the three tensors travel; `.result()` makes the current stream wait for them
        if self._gather_pipes is None:
            self._gather_pipes = [hkdist.GatherPipeline(shard, rows_per_game=self.max_length_game, depth=2)
                                  for _ in range(3)]
        tickets = [pipe.submit(x.contiguous()) for pipe, x in zip(self._gather_pipes, out)]
        return PendingRollout(self._gather_pipes, tickets)

    # ---- jax_trainer.py:398-465 ----------------------------------------------------------------------------------
    def validate(self, metric_fn: Optional[Callable] = None, verbose=0, batch_size=50, num_of_loops=10, max_length=None,
                 write_wandb=False, key=None) -> Tuple[List, List]:
        """The reference's battle schedule over `compute_rho`: the host network against the agent network and the three
        fixed agents (random, choose_first, choose_last), then the three fixed hosts (random, all_coord, zeillinger)
        against the agent network -- seven metrics (rho) and seven game-length histograms, in the reference's order.
        The fixed strategies go BY NAME (fixed-vs-network pairs run the reference-shaped step loop over `hk_step`); the
        networks as `action_wrapper(partial(policy_fns[role], params=...))`.  Logging / wandb stay with the caller
        (control plane: out of scope)."""
        from functools import partial
        key = time.time_ns() % (1 << 62) if key is None else int(key)
        metric_fn = self.compute_rho if metric_fn is None else metric_fn
        host_net = action_wrapper(partial(self.policy_fns["host"], params=self.host_params), None)
        agent_net = action_wrapper(partial(self.policy_fns["agent"], params=self.agent_params), None)
        from . import players as _players
        hosts = [host_net] + [get_host_with_flattened_obs(self.spec, getattr(_players, n))
                              for n in ("random_host_fn", "all_coord_host_fn", "zeillinger_fn")]
        agents = [agent_net] + [partial(getattr(_players, n), spec=self.spec)
                                for n in ("random_agent_fn", "choose_first_agent_fn", "choose_last_agent_fn")]
        schedule = [(0, i) for i in range(len(agents))] + [(i, 0) for i in range(1, len(hosts))]
        rhos, details = [], []
        for hi, ai in schedule:
            key, _ = _split(key, 2)
            rho, detail = metric_fn(hosts[hi], agents[ai], batch_size=batch_size, num_of_loops=num_of_loops,
                                    max_length=max_length, write_wandb=False, key=key)
            rhos.append(rho)
            details.append(detail)
        return rhos, details

    def rollout_postprocess(self, rollouts, role: str, use_unified_tree=True):
        """jax_trainer.py:558-592"""
        return _rollout_postprocess(rollouts, role, self.dimension, self.discount, use_unified_tree)

    # ---- jax_trainer.py:467-556 ----------------------------------------------------------------------------------
    def compute_rho(self, host: Union[str, Callable], agent: Union[str, Callable], batch_size=None, num_of_loops=10,
                    max_length=None, write_wandb=False, key=None) -> Tuple[float, List]:
        """rho and the game-length histogram of host vs agent.  `host` / `agent`: callables as in the reference
        (observations -> one-hot actions; the networks: ``action_wrapper(partial(trainer.policy_fns[role],
        params=...), None)``) or the NAME of a fixed strategy ("random", "zeillinger", "all_coord" / "random",
        "choose_first", "choose_last", "random_legal"), which runs as one fused kernel per batch."""
        key = time.time_ns() % (1 << 62) if key is None else int(key)
        max_length = self.max_length_game if max_length is None else max_length
        batch_size = self.eval_batch_size if batch_size is None else batch_size
        world = hkdist.world()
        return _compute_rho(host, agent, spec=self.spec, batch_size=batch_size,
                            max_value=self.max_value, max_length=max_length, num_of_loops=num_of_loops,
                            reposition=self.reposition, key=key, dtype=self.dtype, device=self.device,
                            game_offset=hkdist.rank() * batch_size, world_batch=batch_size * world)
