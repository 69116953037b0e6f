"""Rollouts without a search: the counterparts of ``JAXTrainer.compute_rho``
(hironaka/jax/jax_trainer.py:467-556) and of the shape of ``JAXTrainer.simulate``'s output
(jax_trainer.py:247-320, 558-592).

* ``compute_rho(host, agent, ...)`` -- policy-vs-policy games from freshly generated states; returns
  (rho, details) with the reference's definitions: ``details[s]`` = #games that finished at exactly s
  steps (the last bin collects what is left), ``rho = sum(details[1:]) / sum(s * details[s])``.
  With the fixed policies of ``players.py`` given BY NAME the whole loop is ONE fused kernel launch
  per batch (``hk_rollout``); with arbitrary callables it is the reference's step-by-step loop over
  ``take_actions`` (one fused launch per step).
* ``simulate_fixed_policies`` -- (obs, policy, value) tensors shaped like ``simulate``'s output
  ``(B*T, obs_dim) / (B*T, A) / (B*T,)`` for fixed policies: observations before each step, one-hot
  policy targets of the actions taken, and discounted-reward value targets (value of a finished game:
  +-discount^k to the finishing step, the convention pinned by test/testJAXTrainer.py:91-389).
  (Rollouts WITH the Gumbel-MuZero search -- third-party mctx in the reference -- are
  ``hironaka_amd.simulation_fn`` / ``hironaka_amd.trainer_api.HipTrainer.simulate``.)
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple, Union

import torch

from . import _abi as A
from . import ops
from .functional import flatten, generate_pts, get_dones, get_take_actions, make_agent_obs
from .host_action_preprocess import get_batch_decode_from_one_hot, num_classes

_HOSTS = {"random": A.HK_HOST_RANDOM, "all_coord": A.HK_HOST_ALL_COORD, "zeillinger": A.HK_HOST_ZEILLINGER}
_AGENTS = {"random": A.HK_AGENT_RANDOM, "random_legal": A.HK_AGENT_RANDOM_LEGAL,
           "choose_first": A.HK_AGENT_CHOOSE_FIRST, "choose_last": A.HK_AGENT_CHOOSE_LAST}


def details_from_done_counts(done_count: torch.Tensor, total_games: int) -> List[int]:
    """done_count[s] = #games finished after s steps (s = 0..L-1)  ->  the reference's `details`
    histogram of length L, exactly as its loop leaves it (jax_trainer.py:501,519-540): details[s] =
    done_count[s] - done_count[s-1] for s < L-1 (added at the top of loop iteration s, before that
    iteration's move), and after the last of the L-1 moves details[L-1] = total - done_count[L-1].
    Games that finish on the very last move are therefore in NO bin (the reference drops them: neither
    the numerator nor the denominator of rho sees them); sum(details) <= total."""
    dc = [int(v) for v in done_count.tolist()]
    length = len(dc)
    details = [0] * length
    prev = 0
    for s in range(length - 1):
        details[s] = dc[s] - prev
        prev = dc[s]
    details[length - 1] = total_games - dc[length - 1]
    return details


def rho_from_details(details: List[int]) -> float:
    denom = sum(i * n for i, n in enumerate(details))
    return sum(details[1:]) / denom if denom else float("nan")


def compute_rho(host: Union[str, Callable], agent: Union[str, Callable], *, spec: Tuple[int, int],
                batch_size: int, max_value: int, max_length: int, num_of_loops: int = 10,
                reposition: bool = True, key: int = 0, dtype=torch.float32, device=None,
                game_offset: int = 0, world_batch: Optional[int] = None) -> Tuple[float, List[int]]:
    """jax_trainer.py:467-556.  `host` / `agent`: a name from players.py ("random", "all_coord",
    "zeillinger" / "random", "random_legal", "choose_first", "choose_last") -> fused kernel; or
    callables host(pts_flat, key=) -> one-hot [B, A], agent(agent_obs, key=) -> one-hot [B, d].
    `game_offset` / `world_batch` place this process' shard inside a larger sharded batch: when
    `world_batch` exceeds `batch_size` (one process per GPU, torch.distributed initialised) the per-step
    finished-game counts are summed over the ranks (`distributed.all_reduce_counts` -- the reference sums them
    over its device axis, jax_trainer.py:513,533-534) and every rank returns the histogram of all
    `world_batch * num_of_loops` games."""
    m, d = spec
    stages = ops.make_stages(True, reposition, True, False)
    fused = isinstance(host, str) and isinstance(agent, str)
    world_batch = batch_size if world_batch is None else world_batch
    totals = None
    if fused:
        # The reference's loop draws a fresh batch per iteration (jax/util.py:385-392) and reads nothing back but the
        # histogram (jax_trainer.py:513,533-534).  ONE launch does all `num_of_loops` of them: the initial states are
        # drawn inside the kernel (hk_rollout_desc.gen_max_value), episode e with seeds key + e, no state is stored
        # (points = NULL), the per-step finished-game counts of all episodes accumulate in the launch's workspace and
        # are reduced once.  A wave starts its next episode as soon as its own games are finished, so nothing waits for
        # a launch's slowest waves.  Requests the fused kernel does not serve (other shapes / dtypes, Zeillinger's
        # host on the small shapes) run as generate + rollout per episode inside the library, on a state buffer.
        from ._lib import HironakaHipError
        dev = torch.device("cuda" if device is None else device)
        common = dict(max_value=max_value, gen_seed=key, reposition=reposition, episodes=num_of_loops,
                      game_offset=game_offset, host_policy=_HOSTS[host], agent_policy=_AGENTS[agent], stages=stages,
                      dtype=dtype, device=dev)
        try:
            res = ops.rollout_generated(batch_size, spec, max_length - 1, key, **common)
        except HironakaHipError as e:
            if e.status != A.HK_ERR_UNSUPPORTED:
                raise
            buf = torch.empty((batch_size, m, d), dtype=dtype, device=dev)
            res = ops.rollout_generated(batch_size, spec, max_length - 1, key, out=buf, **common)
        totals = res["done_count"]
        details = details_from_done_counts(_world_counts(totals, batch_size, world_batch), world_batch * num_of_loops)
        return rho_from_details(details), details
    # arbitrary callables: the reference's step-by-step loop over take_actions (one fused launch per step)
    take_action = get_take_actions("host", spec, rescale_points=False, reposition=reposition)
    batch_decode = get_batch_decode_from_one_hot(d)
    for loop in range(num_of_loops):
        pts = generate_pts(key + loop, (batch_size, m, d), max_value, dtype, False, reposition,
                           game_offset=game_offset, device=device)
        counts = torch.zeros(max_length, dtype=torch.int64, device=pts.device)
        counts[0] = get_dones(pts).sum()
        flat = flatten(pts)
        for step in range(max_length - 1):
            host_action = host(flat, key=(key * 7919 + loop * 131 + 2 * step) % (1 << 62))
            coords = batch_decode(host_action, dtype)
            agent_obs = make_agent_obs(flat, coords)
            axis = torch.argmax(agent(agent_obs, key=(key * 7919 + loop * 131 + 2 * step + 1) % (1 << 62)), dim=-1)
            flat = take_action(flat, coords, axis)
            counts[step + 1] = get_dones(flat.reshape(-1, m, d)).sum()
        totals = counts.clone() if totals is None else totals + counts
    details = details_from_done_counts(_world_counts(totals, batch_size, world_batch), world_batch * num_of_loops)
    return rho_from_details(details), details


def _world_counts(totals: torch.Tensor, batch_size: int, world_batch: int) -> torch.Tensor:
    """this rank's per-step finished-game counts -> those of the whole sharded batch"""
    if world_batch == batch_size:
        return totals
    from . import distributed as hkdist
    if hkdist.world() == 1:
        raise ValueError(f"world_batch={world_batch} != batch_size={batch_size} needs an initialised "
                         f"torch.distributed process group (one process per GPU)")
    return hkdist.all_reduce_counts(totals)


def simulate_fixed_policies(key: int, role: str, *, spec: Tuple[int, int], batch_size: int, max_value: int,
                            max_length_game: int, host: str = "random", agent: str = "random",
                            reposition: bool = True, discount: float = 0.99, dtype=torch.float32,
                            device=None, game_offset: int = 0):
    """(obs, policy, value) with the shapes `JAXTrainer.simulate` returns for `role`:
    host:  obs [B*T, m*d],     policy [B*T, 2^d-d-1] (one-hot of the class played)
    agent: obs [B*T, m*d + d], policy [B*T, d]       (one-hot of the axis played)
    value [B*T]: sign * discount^(steps until the game finishes) for games that finish inside the
    rollout, sign/num_points for those that do not (jax/util.py:152-169,261-284); sign = +1 host, -1 agent."""
    if role not in ("host", "agent"):
        raise ValueError(f"role must be either host or agent. Got {role}.")
    m, d = spec
    T = max_length_game
    pts = generate_pts(key, (batch_size, m, d), max_value, dtype, False, reposition, game_offset=game_offset,
                       device=device)
    stages = ops.make_stages(True, reposition, True, False)
    res = ops.rollout(pts, T, key, game_offset=game_offset, host_policy=_HOSTS[host], agent_policy=_AGENTS[agent],
                      stages=stages, record=("obs", "host_class", "axis", "done", "game_length"))
    obs = res["obs"].reshape(T, batch_size, m * d)
    sign = 1.0 if role == "host" else -1.0
    if role == "host":
        policy = torch.nn.functional.one_hot(res["host_class"].long(), num_classes(d)).to(dtype)
    else:
        coords = ops.decode_host_class(res["host_class"].reshape(-1), d, dtype).reshape(T, batch_size, d)
        obs = torch.cat([obs, coords], dim=-1)
        policy = torch.nn.functional.one_hot(res["axis"].long(), d).to(dtype)
    # value targets exactly as rollout_postprocess computes them from the observed point counts
    num_points = ops.get_num_points(res["obs"].reshape(T * batch_size, m, d)).reshape(T, batch_size).transpose(0, 1)
    value = calculate_value_using_reward_fn(num_points, discount, role, use_unified_tree=False).to(dtype)
    # [T, B, .] -> [B*T, .] in the reference's (batch-major) order
    to_bt = lambda x: x.transpose(0, 1).reshape(batch_size * T, *x.shape[2:])
    return to_bt(obs), to_bt(policy), value.reshape(batch_size * T)


def calculate_value_using_reward_fn(num_points: torch.Tensor, discount: float, role: str,
                                    use_unified_tree: bool) -> torch.Tensor:
    """jax/util.py:261-284 with the reward / estimate functions rollout_postprocess selects
    (jax_trainer.py:583-584).  num_points [B, T] (points alive at each recorded state) -> values [B, T]:
    discounted reward of the finishing move for games that end inside the rollout (constant, with the
    clipped discount table, after the end) + sign/num_points * discount^(T-1-t) for those that do not."""
    if role not in ("host", "agent"):
        raise ValueError(f"role must be either host or agent. Got {role}.")
    b, t = num_points.shape
    dev = num_points.device
    done = num_points <= 1
    next_done = torch.cat([done[:, 1:], torch.zeros((b, 1), dtype=torch.bool, device=dev)], dim=1)
    rew = (next_done & ~done).to(torch.float32)
    if use_unified_tree or role == "agent":
        rew = -rew  # agent reward (util.py:141-144); the unified tree always uses the agent's
    steps = torch.arange(t, device=dev, dtype=torch.float32)
    disc = torch.tensor(-discount if use_unified_tree else discount, dtype=torch.float32, device=dev)
    table = torch.clamp(torch.pow(disc, steps.reshape(1, -1) - steps.reshape(-1, 1)), -1.0, 1.0)
    discounted = rew @ table.T
    sign = (-1) ** (t + 1) if use_unified_tree else 1
    role_sign = 1.0 if role == "host" else -1.0
    est = (role_sign / num_points[:, -1].clamp(min=1).to(torch.float32)).unsqueeze(1) * sign
    unfinished = (~done[:, -1:]).to(torch.float32) * est * torch.pow(disc, steps.flip(0)).unsqueeze(0)
    return discounted + unfinished


def rollout_postprocess(rollouts, role: str, dimension: int, discount: float = 0.99, use_unified_tree: bool = True):
    """JAXTrainer.rollout_postprocess (jax_trainer.py:558-592): replace the value prior by the ground-truth
    value from win/lose.  (obs [B, T, obs_dim], policy [B, T, A], value [B, T]) -> ([B*T, obs_dim],
    [B*T, A], [B*T])."""
    obs, policy, value = rollouts
    offset = 1 if (use_unified_tree or role == "agent") else 0
    b, t, width = obs.shape
    if obs.is_cuda and obs.dtype == torch.float32 and t <= 64:
        # one launch (hk_rollout_values: a wave per game) instead of ~15 tensor-library launches
        if role not in ("host", "agent"):
            raise ValueError(f"role must be either host or agent. Got {role}.")
        import ctypes as C

        from ._lib import check, lib
        o = obs.contiguous()
        out = torch.empty((b, t), dtype=torch.float32, device=obs.device)
        agent_like = use_unified_tree or role == "agent"
        est = (1.0 if role == "host" else -1.0) * ((-1.0) ** (t + 1) if use_unified_tree else 1.0)
        with torch.cuda.device(obs.device):
            check(lib().hk_rollout_values(o.data_ptr(), out.data_ptr(), b, t, width, dimension, offset,
                                          -discount if use_unified_tree else discount, -1.0 if agent_like else 1.0,
                                          est, C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)),
                  "hk_rollout_values")
        new_value = out.to(value.dtype)
    else:
        num_points = (obs >= 0).sum(dim=-1) // dimension - offset
        new_value = calculate_value_using_reward_fn(num_points, discount, role, use_unified_tree).to(value.dtype)
    return obs.reshape(-1, obs.shape[2]), policy.reshape(-1, policy.shape[2]), new_value.reshape(-1)


def select_sample_after_sim(role: str, rollout, dimension: int, mix_random_terminal_states: bool = True,
                            key: Optional[int] = None) -> torch.Tensor:
    """jax/util.py:351-382 -- mask [sample_size] of the samples kept for training: every state of an
    unfinished game, plus (optionally) as many uniformly chosen samples as there are unfinished ones
    (a random permutation of the indices, those ranked below the number of unfinished states), so that
    terminal states are mixed in at no more than 1:1.  `key`: int seed of the permutation (None: time)."""
    obs = rollout[0]
    size = obs.shape[0]
    offset = dimension if role == "agent" else 0
    undone_idx = (obs >= 0).sum(dim=-1) > (dimension + offset)
    if not mix_random_terminal_states:
        return undone_idx
    import time
    gen = torch.Generator(device=obs.device).manual_seed(int(time.time_ns() if key is None else key) % (1 << 63))
    random_idx = torch.randperm(size, generator=gen, device=obs.device)
    return undone_idx | (random_idx < undone_idx.sum())
