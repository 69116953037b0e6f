"""Self-play with search -- the counterpart of ``hironaka/jax/simulation_fn.py`` with the same factory
names and arguments: ``get_evaluation_loop`` (one batched Gumbel-MuZero search from a batch of root
states, simulation_fn.py:16-122) and ``get_simulation`` (play ``max_length_game`` moves, each chosen by a
search, and collect (observation, improved policy, root value) for every move, simulation_fn.py:125-213).

The search itself (third-party ``mctx`` in the reference) is ``hironaka_amd.search`` over HIP tree
kernels; the environment inside ``recurrent_fn`` is the HIP step.  Randomness: ``key`` is an int seed (the
reference's JAX PRNG keys cannot be reproduced); sub-keys are derived by fixed arithmetic.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from .functional import get_dynamic_policy_fn
from .recurrent_fn import get_recurrent_fn_for_role, get_unified_recurrent_fn
from .search import CapturedSearch, PolicyOutput, RootFnOutput, gumbel_muzero_policy


def _split(key: int, n: int = 2):
    """deterministic sub-keys (stand-in for jax.random.split)"""
    return tuple((int(key) * 6364136223846793005 + 1442695040888963407 * (i + 1)) % (1 << 63) for i in range(n))


def get_evaluation_loop(role: str, policy_fn: Callable, opponent_fn: Callable, reward_fn: Callable,
                        spec: Tuple[int, int], num_evaluations: int, max_depth: int,
                        max_num_considered_actions: int, discount: float, rescale_points: bool, reposition: bool,
                        role_agnostic: Optional[bool] = None, gumbel_scale: Optional[float] = 0.3,
                        dtype=torch.float32, use_graph: bool = False, expander=None) -> Callable:
    """simulation_fn.py:16-122.  policy_fn(observations, *args, key=) -> (policy_prior, value_prior);
    opponent_fn(observations, *args, key=) -> one-hot actions; reward_fn(dones, prev_dones) -> rewards.
    use_graph (not in the reference): capture the search of each distinct batch shape into a hipGraph
    (search.CapturedSearch: its requirements apply -- role-specific trees only, callables without host
    synchronisation, randomness from torch's default generator, same argument objects on every call).
    expander (not in the reference): a recurrent_fn.HostExpander -- the expansions of a host-role tree through the
    fused operators."""
    if role_agnostic:
        policy_fn_on_root = get_dynamic_policy_fn(spec, policy_fn, opponent_fn)
        recurrent_fn = get_unified_recurrent_fn(policy_fn, opponent_fn, reward_fn, spec, discount=discount,
                                                dtype=dtype, rescale_points=rescale_points, reposition=reposition)
    else:
        def policy_fn_on_root(state, role_and_opponent_params, *args, **kwargs):
            params, _ = role_and_opponent_params
            return policy_fn(state, *params, *args, **kwargs)

        recurrent_fn = get_recurrent_fn_for_role(role, policy_fn, opponent_fn, reward_fn, spec, discount=discount,
                                                 dtype=dtype, rescale_points=rescale_points, reposition=reposition,
                                                 expander=expander)

    if use_graph and role_agnostic:
        raise ValueError("the role-agnostic tree decides host/agent on the host per call: not capturable")
    captured = {}

    def evaluation_loop(key: int, root_states: torch.Tensor, role_fn_args=(), opponent_fn_args=(),
                        invalid_actions=None) -> PolicyOutput:
        key, subkey = _split(key)
        policy_prior, value_prior = policy_fn_on_root(root_states, (role_fn_args, opponent_fn_args), key=subkey)
        root = RootFnOutput(prior_logits=policy_prior, value=value_prior, embedding=root_states)
        key, subkey = _split(key)
        if use_graph:
            # one capture per launch shape; it is tied to the identity of the parameter OBJECTS inside the argument
            # tuples (a hipGraph replays with the tensors it was captured with) -- not of the tuples themselves, which
            # callers such as functional.mcts_wrapper build afresh on every call: keyed by the tuples' ids a search was
            # re-captured on every opponent move and never replayed.  Other parameter objects REPLACE the capture -- no
            # unbounded growth when a trainer hands over fresh ones every optimiser step (updating the tensors in place
            # keeps it); the cache entry keeps the objects alive, so an id cannot be recycled while it is the key.
            sig = (tuple(root_states.shape), root_states.dtype, root_states.device, invalid_actions is not None)
            ids = (tuple(map(id, role_fn_args)), tuple(map(id, opponent_fn_args)))
            if sig not in captured or captured[sig][0] != ids:
                captured.pop(sig, None)
                captured[sig] = (ids, CapturedSearch((role_fn_args, opponent_fn_args), subkey, root, recurrent_fn,
                                                     num_evaluations, invalid_actions, max_depth,
                                                     max_num_considered_actions=max_num_considered_actions,
                                                     gumbel_scale=gumbel_scale), (role_fn_args, opponent_fn_args))
            return captured[sig][1](subkey, root, invalid_actions)
        return gumbel_muzero_policy(params=(role_fn_args, opponent_fn_args), rng_key=subkey, root=root,
                                    recurrent_fn=recurrent_fn, num_simulations=num_evaluations,
                                    invalid_actions=invalid_actions, max_depth=max_depth,
                                    max_num_considered_actions=max_num_considered_actions,
                                    gumbel_scale=gumbel_scale)

    evaluation_loop.role_agnostic = role_agnostic
    return evaluation_loop


def get_simulation(role: str, evaluation_loop: Callable, eval_batch_size: int, max_num_points: int, dimension: int,
                   max_length_game: int, dtype=torch.float32, **kwargs) -> Callable:
    """simulation_fn.py:125-213.  Returns simulation(key, root_state, role_fn_args=(), opponent_fn_args=())
    -> (obs [B, T, input_dim], log policy [B, T, action_num], value [B, T])."""
    if getattr(evaluation_loop, "role_agnostic", None):
        input_dim = (max_num_points + 1) * dimension
        action_num = 2 ** dimension - dimension - 1
    else:
        input_dim = max_num_points * dimension if role == "host" else (max_num_points + 1) * dimension
        action_num = 2 ** dimension - dimension - 1 if role == "host" else dimension

    def simulation(key: int, root_state: torch.Tensor, role_fn_args=(), opponent_fn_args=()):
        state = root_state.to(dtype)
        dev = state.device
        obs = torch.zeros((eval_batch_size, max_length_game, input_dim), dtype=dtype, device=dev)
        policy = torch.zeros((eval_batch_size, max_length_game, action_num), dtype=dtype, device=dev)
        value = torch.zeros((eval_batch_size, max_length_game), dtype=dtype, device=dev)
        rows = torch.arange(eval_batch_size, device=dev)
        key, _, loop_key = _split(key, 3)
        for i in range(max_length_game):
            out = evaluation_loop(loop_key, state, role_fn_args=role_fn_args, opponent_fn_args=opponent_fn_args)
            obs[:, i] = state.reshape(eval_batch_size, input_dim)
            policy[:, i] = out.action_weights.reshape(eval_batch_size, action_num).to(dtype)
            value[:, i] = out.search_tree.node_values[:, 0].to(dtype)
            # the next state is the embedding of the child the improved policy chose (always a visited one)
            child = out.search_tree.children_index[rows, 0, out.action].long()
            state = out.search_tree.embeddings[rows, child]
            key, _, loop_key = _split(key, 3)
        return obs, torch.log(policy), value

    return simulation
