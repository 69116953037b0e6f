"""Functional environment API on device tensors -- the counterpart of the env part of
``hironaka/jax/util.py`` (same names, same argument meaning, same error behaviour), backed by the
HIP kernels through the C ABI.  What `recurrent_fn`, `compute_rho` and `search.py` call in the
reference is what is exported here:

    get_take_actions(role, spec, rescale_points=False, reposition=True) -> take_actions
    take_actions(observations, actions, axis) -> [B, m*d]          one fused launch (hk_step)
    get_dones, get_done_from_flatten, get_reward_fn, get_preprocess_fns, make_agent_obs, flatten,
    get_feature_fn, generate_pts

Differences forced by the platform: arrays are torch tensors on a HIP device; PRNG keys are integer
seeds (Philox) instead of JAX threefry keys; there is no pmap -- one process per
GPU, see ``hironaka_amd.distributed``.
"""
from __future__ import annotations

import functools
from typing import Callable, Optional, Tuple

import torch

from . import _abi as A
from . import ops

Rollout = Tuple[torch.Tensor, torch.Tensor, torch.Tensor]  # observations, policy logits, values


def flatten(x: torch.Tensor) -> torch.Tensor:
    """jax/util.py:19 -- [B, ...] -> [B, prod(...)]"""
    return x.reshape(x.shape[0], -1)


def make_agent_obs(pts: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """jax/util.py:22-31 -- points flattened and concatenated with the host's subset mask."""
    return torch.cat([flatten(pts), coords.to(pts.dtype)], dim=1)


def get_dones(pts: torch.Tensor) -> torch.Tensor:
    """jax/util.py:34-35 -- [B, m, d] -> [B] bool."""
    return ops.get_dones(pts)


def get_done_from_flatten(obs: torch.Tensor, role: str, dimension: int) -> torch.Tensor:
    """jax/util.py:38-39 -- number of non-negative scalars <= d (+d for an agent observation)."""
    return (obs >= 0).sum(dim=-1) <= dimension + (role == "agent") * dimension


@functools.lru_cache()
def get_preprocess_fns(role: str, spec: Tuple[int, int]) -> Tuple[Callable, Callable]:
    """jax/util.py:43-79 -- (obs_preprocess, coords_preprocess) for flattened observations."""
    m, d = spec
    if role == "host":

        def obs_preprocess(observations):
            return observations.reshape(-1, m, d)

        def coords_preprocess(observations, actions):
            return actions

    elif role == "agent":

        def obs_preprocess(observations):
            return observations[:, : m * d].reshape(-1, m, d)

        def coords_preprocess(observations, actions):
            return observations[:, m * d: m * d + d]

    else:
        raise ValueError(f"role must be either host or agent. Got {role}.")
    return obs_preprocess, coords_preprocess


@functools.lru_cache()
def get_take_actions(role: str, spec: Tuple[int, int], rescale_points: bool = False,
                     reposition: bool = True) -> Callable:
    """jax/util.py:83-125.  role == 'host': observations = points ([B, m*d] or [B, m, d]), coords =
    actions ([B, d] mask, or [B] class ids), axis = axis.  role == 'agent': observations =
    concat(points, coords) [B, m*d + d], `actions` ignored, axis = axis.  Returns the new points
    flattened to [B, m*d] (NOT the concatenated agent observation).  One hk_step launch."""
    if role not in ("host", "agent"):
        raise ValueError(f"role must be either host or agent. Got {role}.")
    m, d = spec
    stages = ops.make_stages(shift=True, reposition=reposition, newton=True, rescale=rescale_points)

    def take_actions(observations: torch.Tensor, actions: Optional[torch.Tensor], axis: torch.Tensor,
                     want=(), reward_sign: float = 1.0):
        """want (not in the reference): also return "done" / "prev_done" / "reward" / "num_points" of the
        same launch -- {"points": [B, m*d], ...} instead of the bare points"""
        if role == "host":
            obs = observations if observations.dim() == 3 else observations.reshape(-1, m * d)
            res = ops.step(obs, actions, axis, stages=stages, spec=None if obs.dim() == 3 else (m, d), want=want,
                           reward_sign=reward_sign)
        else:
            res = ops.step(observations, None, axis, stages=stages, spec=(m, d), coords_in_record=True,
                           want=want, reward_sign=reward_sign)
        out = res["points"].reshape(-1, m * d)
        if want:
            res["points"] = out
            return res
        return out

    return take_actions


@functools.lru_cache()
def get_reward_fn(role: str) -> Callable:
    """jax/util.py:129-149 -- host: f32(done & ~prev_done); agent: its negative."""
    if role == "host":

        def reward_fn(dones: torch.Tensor, prev_dones: torch.Tensor) -> torch.Tensor:
            return (dones & (~prev_dones)).to(torch.float32)

        reward_fn.hk_reward_sign = 1.0  # lets recurrent_fn take the reward from the step launch itself
    elif role == "agent":

        def reward_fn(dones: torch.Tensor, prev_dones: torch.Tensor) -> torch.Tensor:
            return -(dones & (~prev_dones)).to(torch.float32)

        reward_fn.hk_reward_sign = -1.0
    else:
        raise ValueError(f"role must be either host or agent. Got {role}.")
    return reward_fn


def get_value_est_fn(role: str) -> Callable:
    """jax/util.py:153-169 -- the value of an unfinished game: +-1 / max(number of remaining points, 1)"""
    sign = 1 if role == "host" else -1

    def est_fn(last_values: torch.Tensor, num_points: torch.Tensor) -> torch.Tensor:
        return 1 / torch.clamp(num_points, min=1) * sign

    return est_fn


@functools.lru_cache()
def get_feature_fn(role: str, spec: Tuple[int, int], scale_observation: bool = True) -> Callable:
    """jax/util.py:172-214 -- [rescale] + rows ordered descending (last coordinate primary); the
    agent variant keeps the coordinate tail."""
    assert len(spec) == 2
    m, d = spec
    if role == "host":

        def feature_fn(observations: torch.Tensor) -> torch.Tensor:
            obs = observations if observations.dim() == 3 else observations.reshape(-1, m * d)
            return ops.get_features(obs, scale_observation, spec=None if obs.dim() == 3 else (m, d))

    elif role == "agent":

        def feature_fn(observations: torch.Tensor) -> torch.Tensor:
            feats = ops.get_features(observations, scale_observation, spec=(m, d))
            return torch.cat([feats, observations[:, m * d: m * d + d]], dim=1)

    else:
        raise ValueError(f"role must be either host or agent. Got {role}.")
    return feature_fn


def generate_pts(key: int, shape: Tuple[int, int, int], max_value: int, dtype=torch.float32,
                 rescale: bool = True, reposition: bool = True, *, game_offset: int = 0,
                 device=None) -> torch.Tensor:
    """jax/util.py:385-392 -- randint[0, max_value) -> newton -> [reposition] -> [rescale].
    `key` is an integer seed; `game_offset` is the global index of the first game (sharding)."""
    batch, m, d = shape
    return ops.generate_points(batch, m, d, max_value, int(key), game_offset=game_offset, dtype=dtype,
                               device=device, newton=True, reposition=reposition, rescale=rescale)


def rollout_sanity_tests(rollout: Rollout, spec: Tuple[int, int]) -> bool:
    """jax/util.py:395-423 -- masks applied where the observation carries one; policy not a softmax."""
    obs, policy, value = rollout
    m, d = spec
    if obs.shape[-1] == (m + 1) * d:
        mask = obs[..., -d:] > 0.5
        is_host = torch.isclose(mask.float(), torch.zeros((), device=obs.device)).all(dim=-1, keepdim=True)
        full = torch.cat([mask | is_host, is_host.expand(*is_host.shape[:-1], policy.shape[-1] - d)], dim=-1)
        masked_out = policy[~full]
        if masked_out.numel() and not torch.isinf(masked_out).all():
            return False
    sums = policy.sum(dim=-1)
    if torch.isclose(sums, torch.ones_like(sums)).all() and ((policy <= 1.0) & (policy >= 0.0)).all():
        return False
    return True


def get_dynamic_policy_fn(spec: Tuple[int, int], host_fn: Callable, agent_fn: Callable) -> Callable:
    """jax/util.py:217-258 -- the policy of a role-agnostic tree: states are [B, (m+1)*d]; a host state
    has a zero tail, an agent state carries its subset mask there.  As in the reference the choice is made
    ONCE per batch (host if any state of the batch is a host state); agent logits are padded with -inf to
    the host's action count."""
    _, coords_preprocess = get_preprocess_fns("agent", spec)
    extra_action_dim = 2 ** spec[1] - 2 * spec[1] - 1

    def dynamic_policy_fn(state, host_and_agent_args, *args, **kwargs):
        host_args, agent_args = host_and_agent_args
        coord = coords_preprocess(state, None)
        use_host = bool(torch.isclose(coord, torch.zeros((), device=coord.device, dtype=coord.dtype)).all(dim=-1).any())
        if use_host:
            return host_fn(state, *host_args, *args, **kwargs)
        policy, value = agent_fn(state, *agent_args, *args, **kwargs)
        return torch.nn.functional.pad(policy, (0, extra_action_dim), value=float("-inf")), value

    return dynamic_policy_fn


def _name_of(obj):
    if hasattr(obj, "__name__"):
        return obj.__name__
    if hasattr(obj, "func"):  # functools.partial
        return _name_of(obj.func)
    return None


def apply_agent_action_mask(agent_policy: Callable, dimension: int, nan_free: bool = False) -> Callable:
    """jax/util.py:287-305 -- restrict an agent's (policy, value) function to the host's subset, read from the
    last `dimension` entries of the flattened observation (cut at 0.5).

    The reference computes ``policy * mask - inf * (~mask)``; ``inf * 0`` is NaN, so with that expression the
    ALLOWED entries come out NaN and the excluded ones -inf (an argmax then lands on the first allowed axis).
    The default reproduces the expression as written; ``nan_free=True`` keeps the allowed logits instead."""

    def masked_agent_policy(x: torch.Tensor, *args, **kwargs):
        mask = x[..., x.shape[-1] - dimension:] > 0.5
        policy_prior, value_prior = agent_policy(x, *args, **kwargs)
        if nan_free:
            return torch.where(mask, policy_prior, torch.full_like(policy_prior, float("-inf"))), value_prior
        return policy_prior * mask - float("inf") * (~mask).to(policy_prior.dtype), value_prior

    masked_agent_policy.__name__ = _name_of(agent_policy)
    return masked_agent_policy


def action_wrapper(policy_value_fn: Callable, dimension: Optional[int] = None) -> Callable:
    """jax/util.py:308-327 -- (policy, value) function -> one-hot argmax actions, behind the agent's action mask
    if `dimension` is given"""
    masked_action = policy_value_fn if dimension is None else apply_agent_action_mask(policy_value_fn, dimension)

    def wrapped_action_fn(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        out, _ = masked_action(x, *args, **kwargs)
        return torch.nn.functional.one_hot(torch.argmax(out, dim=-1), out.shape[-1]).to(torch.float32)

    def index_fn(x: torch.Tensor, *args, **kwargs) -> torch.Tensor:
        """the chosen action as an index [B] -- what `recurrent_fn` would recover from the one-hot array with another
        argmax (it uses this shortcut when the opponent offers it: three launches per simulation less)"""
        out, _ = masked_action(x, *args, **kwargs)
        return torch.argmax(out, dim=-1)

    wrapped_action_fn.__name__ = _name_of(policy_value_fn)
    wrapped_action_fn.index_fn = index_fn
    return wrapped_action_fn


def mcts_wrapper(eval_loop: Callable) -> Callable:
    """jax/util.py:330-341 -- an evaluation loop (simulation_fn.get_evaluation_loop) as a (policy, value)
    function: log of the search's action weights (clipped at 1e-8, jax/loss.py:27-28) and the root values"""

    def mcts_wrapped_policy(x: torch.Tensor, params, opp_params, key):
        policy_output = eval_loop(key, x, (params,), (opp_params,))
        return (torch.log(torch.clamp(policy_output.action_weights, min=1e-8)),
                policy_output.search_tree.node_values[:, 0])

    return mcts_wrapped_policy
