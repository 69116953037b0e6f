"""``HipTrainer`` -- the environment side of ``hironaka.jax.JAXTrainer`` behind the reference's own entry
points and config keys (hironaka/jax/jax_trainer.py:150-320, 398-592, 626-876):

    trainer = HipTrainer(key, config, host_net=..., agent_net=..., host_params=..., agent_params=...)
    obs, target_policy, target_value = trainer.simulate(key, role, use_mcts_policy=False, use_unified_tree=False)
    rho, details = trainer.compute_rho(host, agent, batch_size=..., num_of_loops=10, max_length=...)
    rhos, details = trainer.validate(...)

What is NOT here is the neural-network side (flax models, optax optimisers, ``train``, checkpoints, wandb):
out of scope.  The two networks therefore arrive as callables
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

import functools
import time
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch

from . import distributed as hkdist
from .functional import (action_wrapper, apply_agent_action_mask, flatten, generate_pts, get_feature_fn,
                         get_reward_fn, get_value_est_fn, mcts_wrapper)
from .host_action_preprocess import get_batch_decode_from_one_hot
from .players import (all_coord_host_fn, choose_first_agent_fn, choose_last_agent_fn, get_host_with_flattened_obs,
                      get_name, random_agent_fn, random_host_fn, zeillinger_fn)
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

# the fixed strategies `validate` pits the networks against (jax_trainer.py:707-719); given BY NAME to
# compute_rho they run as one fused rollout kernel per batch
_FIXED_HOSTS = ("random", "zeillinger", "all_coord")
_FIXED_AGENTS = ("random", "choose_first", "choose_last")


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


class HipTrainer:
    """See the module docstring.  Attribute and method names follow JAXTrainer."""

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
        self.use_graph = use_graph
        # host-role trees of the trainer's own policies expand through the fused operators (recurrent_fn.HostExpander)
        # -- possible while the standard feature functions sit in front of the networks
        self.fused_expand = bool(fused_expand) and host_feature_fn is None and agent_feature_fn is None
        self.key = int(key)
        spec = (self.max_num_points, self.dimension)
        self.host_feature_fn = host_feature_fn if host_feature_fn is not None else \
            get_feature_fn("host", spec, scale_observation=self.scale_observation)
        self.agent_feature_fn = agent_feature_fn if agent_feature_fn is not None else \
            get_feature_fn("agent", spec, scale_observation=self.scale_observation)
        self.device_num = hkdist.world()  # one process per GPU
        m, d = spec
        self.input_dim = {"host": m * d, "agent": m * d + d}
        self.output_dim = {"host": 2 ** d - d - 1, "agent": d}
        # the two networks (callables + opaque parameters)
        if host_net is None or agent_net is None:
            raise ValueError("HipTrainer needs host_net and agent_net: net(features, params) -> (logits [B, A], "
                             "value [B]).  The reference builds flax networks here (net_type / net_arch); networks "
                             "are outside this package (trainer_api.standin_mlp gives a fixed random stand-in).")
        self.host_model, self.agent_model = host_net, agent_net
        self.host_params, self.agent_params = host_params, agent_params
        self._fn_args: Dict[Any, Any] = {}
        for role in ("host", "agent"):
            self.update_policy_fn(role)
        for role in ("host", "agent"):
            setattr(self, f"{role}_reward_fn", get_reward_fn(role))
            setattr(self, f"{role}_value_est_fn", get_value_est_fn(role))
        for role in ("host", "agent"):
            self.update_eval_sim_and_mcts_policy(role)
        self.cached_hosts_agents_for_validation: Dict[int, Any] = {}
        self.log: Dict[str, Any] = {}

    # ---- accessors (jax_trainer.py:626-669) ------------------------------------------------------------------
    def get_fns(self, role: str, name: str) -> Callable:
        if role not in ("host", "agent"):
            raise ValueError(f"role must be either host or agent. Got {role}.")
        return getattr(self, f"{role}_{name}")

    def get_policy_fn(self, role: str) -> Callable:
        return self.get_fns(role, "policy_fn")

    def get_eval_loop(self, role: str) -> Callable:
        return self.get_fns(role, "eval_loop")

    def get_sim_fn(self, role: str) -> Callable:
        return self.get_fns(role, "sim_fn")

    def get_params(self, role: str):
        return self.get_fns(role, "params")

    def get_mcts_sim_fn(self, role: str) -> Callable:
        if getattr(self, f"{role}_mcts_sim_fn", None) is None:
            opponent = self._get_opponent(role)
            _, _, sim_fn, _, _, _ = self.update_eval_sim_and_mcts_policy(
                role, self.get_policy_fn(role), getattr(self, f"{opponent}_mcts_policy_fn"), return_function=True)
            setattr(self, f"{role}_mcts_sim_fn", sim_fn)
        return getattr(self, f"{role}_mcts_sim_fn")

    @staticmethod
    def _get_opponent(role: str) -> str:
        if role == "host":
            return "agent"
        elif role == "agent":
            return "host"
        raise ValueError(f"role must be either host or agent. Got {role}.")

    # ---- jax_trainer.py:735-834 ------------------------------------------------------------------------------
    def update_eval_sim_and_mcts_policy(self, role: str, policy_fn: Optional[Callable] = None,
                                        opp_policy_fn: Optional[Callable] = None, return_function=False) -> Any:
        opponent = self._get_opponent(role)
        spec = (self.max_num_points, self.dimension)
        simulation_config = {"eval_batch_size": self.eval_batch_size, "max_num_points": self.max_num_points,
                             "dimension": self.dimension, "max_length_game": self.max_length_game, "dtype": self.dtype}
        mcts_opponent = opp_policy_fn is not None
        own_policies = policy_fn is None and opp_policy_fn is None
        policy_fn = getattr(self, f"{role}_policy_fn") if policy_fn is None else policy_fn
        opp_policy_fn = getattr(self, f"{opponent}_policy_fn") if opp_policy_fn is None else opp_policy_fn
        eval_loop_config = {
            "role": role, "policy_fn": policy_fn,
            "opponent_fn": action_wrapper(opp_policy_fn, None),  # definitive actions as one-hot arrays
            "reward_fn": getattr(self, f"{role}_reward_fn"), "num_evaluations": self.num_evaluations, "spec": spec,
            "max_depth": self.max_length_game, "max_num_considered_actions": self.max_num_considered_actions,
            "discount": self.discount, "rescale_points": False,  # rescaling lives in the feature functions
            "reposition": self.reposition, "dtype": self.dtype,
            # one hipGraph per search: only where nothing inside synchronises with the host (an opponent that is
            # itself a search does)
            "use_graph": self.use_graph and not mcts_opponent,
            "expander": ((HostExpander if role == "host" else AgentExpander)(
                             self.host_model, self.agent_model, spec, self.discount, self.scale_observation,
                             self.reposition, rescale_points=False,
                             reward_sign=getattr(getattr(self, f"{role}_reward_fn"), "hk_reward_sign"))
                         if own_policies and self.fused_expand and self.dtype == torch.float32 else None),
        }
        eval_loop_with_gumbel = get_evaluation_loop(gumbel_scale=self.gumbel_scale, **eval_loop_config)
        eval_loop = get_evaluation_loop(gumbel_scale=0.0, **eval_loop_config)
        eval_loop_as_opp = get_evaluation_loop(
            gumbel_scale=0.0, **{**eval_loop_config, "num_evaluations": self.num_evaluations_as_opponent})
        unified_eval_loop_config = {
            **eval_loop_config, "role": "host", "use_graph": False, "expander": None,
            "policy_fn": get_host_with_flattened_obs(spec, self._host_policy_on_points, truncate_input=True),
            "opponent_fn": self.agent_policy_fn,
            "reward_fn": self.agent_reward_fn,  # agent: the off-by-one-step convention of the unified tree
            "role_agnostic": True}
        unified_eval_loop_with_gumbel = get_evaluation_loop(gumbel_scale=self.gumbel_scale, **unified_eval_loop_config)
        unified_eval_loop = get_evaluation_loop(gumbel_scale=0.0, **unified_eval_loop_config)
        sim_fn = get_simulation(role, eval_loop_with_gumbel, **simulation_config)
        unified_sim_fn = get_simulation("host", unified_eval_loop_with_gumbel, **simulation_config)
        mcts_policy_fn = mcts_wrapper(eval_loop_as_opp)
        if role == "agent":
            mcts_policy_fn = apply_agent_action_mask(mcts_policy_fn, self.dimension, nan_free=True)
        if return_function:
            return eval_loop, eval_loop_as_opp, sim_fn, mcts_policy_fn, unified_eval_loop, unified_sim_fn
        setattr(self, f"{role}_eval_loop", eval_loop)
        setattr(self, f"{role}_eval_loop_as_opp", eval_loop_as_opp)
        setattr(self, f"{role}_sim_fn", sim_fn)
        setattr(self, f"{role}_mcts_policy_fn", mcts_policy_fn)
        setattr(self, "unified_eval_loop", unified_eval_loop)
        setattr(self, "unified_sim_fn", unified_sim_fn)
        setattr(self, f"{role}_mcts_sim_fn", None)

    def update_policy_fn(self, role: str, return_function=False) -> Any:
        """jax_trainer.py:836-855: feature function + network; the agent's logits go through its action mask.
        (The mask keeps the allowed logits: the reference's expression `policy * mask - inf * (~mask)`,
        jax/util.py:302, evaluates inf * 0 = NaN on them -- functional.apply_agent_action_mask reproduces that
        by default and documents it; a trainer cannot work with NaN logits.)"""
        policy_fn = get_apply_fn(role, getattr(self, f"{role}_model"), (self.max_num_points, self.dimension),
                                 feature_fn=getattr(self, f"{role}_feature_fn"))
        if role == "agent":
            policy_fn = apply_agent_action_mask(policy_fn, self.dimension, nan_free=True)
        if return_function:
            return policy_fn
        setattr(self, f"{role}_policy_fn", policy_fn)

    def update_fns(self, role: str):
        self.update_eval_sim_and_mcts_policy(role)
        self.update_policy_fn(role)

    def _host_policy_on_points(self, pts: torch.Tensor, *args, dtype=None, **kwargs):
        """the host network on [B, m, d] points (get_host_with_flattened_obs reshapes to that)"""
        return self.host_policy_fn(pts.reshape(pts.shape[0], -1), *args, **kwargs)

    # ---- jax_trainer.py:247-320 ------------------------------------------------------------------------------
    def simulate(self, key: int, role: str, use_mcts_policy=False, use_unified_tree=False):
        """One batch of self-play: ``eval_batch_size`` games per process, ``max_length_game`` moves, every move
        chosen by a Gumbel-MuZero search of ``num_evaluations`` simulations over the HIP environment.
        Returns (obs [B*T, input_dim], target_policy [B*T, A], target_value [B*T]) with B the games of ALL
        ranks (gathered at the end), values replaced by the ground truth of the finished games
        (``rollout_postprocess``)."""
        if role not in ("host", "agent"):
            raise ValueError(f"role must be either host or agent. Got {role}.")
        if use_mcts_policy:
            sim_fn = self.get_mcts_sim_fn(role)
        elif use_unified_tree:
            sim_fn = self.unified_sim_fn
        else:
            sim_fn = self.get_sim_fn(role)
        opponent = self._get_opponent(role)
        if use_unified_tree:
            role_params, opp_params = self.host_params, self.agent_params
        else:
            role_params, opp_params = self.get_params(role), self.get_params(opponent)

        shard = hkdist.shard_range(self.eval_batch_size * hkdist.world())
        root_key, sim_key, host_key = _split(key, 3)
        # rescale is off: rescaling is part of the feature functions (jax_trainer.py:279-288)
        root_state = generate_pts(root_key, (self.eval_batch_size, self.max_num_points, self.dimension), self.max_value,
                                  self.dtype, False, self.reposition, game_offset=shard.start, device=self.device)
        if role == "agent":
            # host coordinates from the host network, decoded and appended: agent observations
            coords, _ = self.get_policy_fn("host")(flatten(root_state), self.host_params, key=host_key)
            coordinate_mask = get_batch_decode_from_one_hot(self.dimension)(coords, self.dtype)
            root_state = torch.cat([flatten(root_state), coordinate_mask], dim=-1)
        else:
            root_state = flatten(root_state)
            if use_unified_tree:  # pad zeros to the length of an agent observation
                root_state = torch.cat([root_state, torch.zeros((self.eval_batch_size, self.dimension),
                                                                dtype=self.dtype, device=self.device)], dim=-1)
        # (the tuples are cached: a captured search replays with the argument objects it was captured with)
        args_key = (role, bool(use_unified_tree), id(role_params), id(opp_params))
        if args_key not in self._fn_args:
            self._fn_args[args_key] = ((role_params,),
                                       (opp_params,) if use_unified_tree else (opp_params, role_params))
        role_fn_args, opp_fn_args = self._fn_args[args_key]
        simulate_output = sim_fn(sim_key, root_state, role_fn_args, opp_fn_args)
        out = self.rollout_postprocess(simulate_output, role, use_unified_tree)
        return hkdist.all_gather_rollout(out, shard)

    def rollout_postprocess(self, rollouts, role: str, use_unified_tree=True):
        """jax_trainer.py:558-592"""
        return _rollout_postprocess(rollouts, role, self.dimension, self.discount, use_unified_tree)

    # ---- jax_trainer.py:467-556 / 398-465 / 686-725 ------------------------------------------------------------
    def compute_rho(self, host: Union[str, Callable], agent: Union[str, Callable], batch_size=None, num_of_loops=10,
                    max_length=None, write_wandb=False, key=None) -> Tuple[float, List]:
        """rho and the game-length histogram of host vs agent.  `host` / `agent`: callables as in the reference
        (observations -> one-hot actions) or the NAME of a fixed strategy ("random", "zeillinger", "all_coord" /
        "random", "choose_first", "choose_last", "random_legal"), which runs as one fused kernel per batch."""
        key = time.time_ns() % (1 << 62) if key is None else int(key)
        max_length = self.max_length_game if max_length is None else max_length
        batch_size = self.eval_batch_size if batch_size is None else batch_size
        world = hkdist.world()
        return _compute_rho(host, agent, spec=(self.max_num_points, self.dimension), batch_size=batch_size,
                            max_value=self.max_value, max_length=max_length, num_of_loops=num_of_loops,
                            reposition=self.reposition, key=key, dtype=self.dtype, device=self.device,
                            game_offset=hkdist.rank() * batch_size, world_batch=batch_size * world)

    def get_cached_hosts_agents_for_validation(self, batch_size: int, force_update=False):
        """the network players as action functions + the fixed strategies (by name where both sides of a battle
        are fixed, as callables against a network)"""
        if batch_size not in self.cached_hosts_agents_for_validation or force_update:
            spec = (self.max_num_points, self.dimension)
            hosts = [None, get_host_with_flattened_obs(spec, random_host_fn),
                     get_host_with_flattened_obs(spec, zeillinger_fn),
                     get_host_with_flattened_obs(spec, all_coord_host_fn)]
            agents = [None, functools.partial(random_agent_fn, spec=spec),
                      functools.partial(choose_first_agent_fn, spec=spec),
                      functools.partial(choose_last_agent_fn, spec=spec)]
            for f, n in zip(agents[1:], ("random_agent_fn", "choose_first_agent_fn", "choose_last_agent_fn")):
                f.__name__ = n
            self.cached_hosts_agents_for_validation[batch_size] = hosts, agents
        hosts, agents = self.cached_hosts_agents_for_validation[batch_size]
        host_net = action_wrapper(functools.partial(self.host_policy_fn, params=self.host_params), None)
        agent_net = action_wrapper(functools.partial(self.agent_policy_fn, params=self.agent_params), None)
        return [host_net, *hosts[1:]], [agent_net, *agents[1:]]

    def validate(self, metric_fn: Optional[Callable] = None, verbose=0, batch_size=50, num_of_loops=10,
                 max_length=None, write_wandb=False, key=None) -> Tuple[List, List]:
        """jax_trainer.py:398-465: the host network against every agent, every other host against the agent
        network; returns (rhos, details) in the reference's battle order."""
        key = time.time_ns() % (1 << 62) if key is None else int(key)
        max_length = self.max_length_game if max_length is None else max_length
        metric_fn = self.compute_rho if metric_fn is None else metric_fn
        hosts, agents = self.get_cached_hosts_agents_for_validation(batch_size)
        battle_schedule = [(0, i) for i in range(len(agents))] + [(i, 0) for i in range(1, len(hosts))]
        rhos, details = [], []
        for pair_idx in battle_schedule:
            key, _ = _split(key)
            host, agent = hosts[pair_idx[0]], agents[pair_idx[1]]
            rho, detail = metric_fn(host, agent, batch_size=batch_size, num_of_loops=num_of_loops,
                                    max_length=max_length, key=key)
            rhos.append(rho)
            details.append(detail)
            if verbose:
                print(f"{get_name(host)} vs {get_name(agent)}: rho = {rho}")
        return rhos, details
