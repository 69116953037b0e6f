"""Vectorised gym-style environments -- the counterparts of ``hironaka/gym_env``
(`HironakaBase`, `HironakaHostEnv`, `HironakaAgentEnv`) with the same constructor keywords,
observation / action / reward conventions and list semantics (state sorted
descending-lexicographically and compacted, ``_list_ops.py:25-41``), for ``num_envs`` independent games
stepped by ONE fused kernel launch.

    HironakaHostEnv   fixes a Host; the learner is the agent: action = axis in Discrete(dim)
        obs = {"points": [N, m, d] float32, "coords": [N, d]},  reward +1 per legal non-final move,
        0 on the final one, invalid_move_penalty on an axis outside the subset (hironaka_host_env.py:41-74)
    HironakaAgentEnv  fixes an Agent; the learner is the host: action = MultiBinary(dim) subset (or a
        discrete code, decoded as the RAW binary expansion like the reference does,
        hironaka_agent_env.py:49-50 / src/_fn.py:156-170);  obs = [N, m, d] float32; reward +1 when the
        game ends, -step_threshold (or the fixed penalty) when a threshold trips, optionally
        +(points removed)  (hironaka_agent_env.py:44-80)

``num_envs=None`` gives the reference's single-game interface (numpy observations without the batch
axis, python scalars).  The state is float64 on the device like the reference's python floats; the
observation is cast to float32 at the boundary (hironaka_base.py:143-150).  ``gym`` is not required:
``spaces`` below is a minimal stand-in with the attributes the reference sets.
"""
from __future__ import annotations

import abc
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

from . import _abi as A
from . import ops
from .agent import Agent
from .host import Host


class spaces:  # minimal stand-ins for gym.spaces
    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete:
        def __init__(self, n):
            self.n = int(n)

    class MultiBinary:
        def __init__(self, n):
            self.n = int(n)

    class Dict(dict):
        pass


class HironakaBase(abc.ABC):
    metadata = {"render_modes": ["ansi"]}

    def __init__(self, dimension: Optional[int] = 3, max_num_points: Optional[int] = 10,
                 max_value: Optional[int] = 10, padding_value: Optional[float] = -1.0,
                 value_threshold: Optional[float] = None, step_threshold: Optional[int] = 1000,
                 fixed_penalty_crossing_threshold: Optional[int] = None, stop_at_threshold: Optional[bool] = True,
                 improve_efficiency: Optional[bool] = False, scale_observation: Optional[bool] = True,
                 reward_based_on_point_reduction: Optional[bool] = False, num_envs: Optional[int] = None,
                 device: Union[str, torch.device] = "cuda", seed: int = 0, **kwargs):
        self.dimension = dimension
        self.max_num_points = max_num_points
        self.max_value = max_value
        self.padding_value = padding_value
        self.value_threshold = value_threshold
        self.step_threshold = step_threshold
        self.fixed_penalty_crossing_threshold = fixed_penalty_crossing_threshold
        self.stop_at_threshold = stop_at_threshold
        self.improve_efficiency = improve_efficiency
        self.scale_observation = scale_observation
        self.reward_based_on_point_reduction = reward_based_on_point_reduction
        self.num_envs = num_envs
        self._n = 1 if num_envs is None else int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TypeError("the environments run on a HIP device")
        self._seed = int(seed)
        self._episode = 0
        high = np.inf if self.scale_observation else 1.0  # (bounds are swapped in the reference too)
        self.point_observation_space = spaces.Box(low=-1.0, high=high, shape=(max_num_points, dimension),
                                                  dtype=np.float32)
        self._points: Optional[torch.Tensor] = None  # [N, m, d] float64, list semantics
        self._coords = torch.zeros((self._n, dimension), dtype=torch.int32, device=self.device)
        self.current_step = 0
        self.exceed_threshold = self._false()
        self.last_action_taken = None

    # ---- helpers ---------------------------------------------------------------------------
    def _false(self):
        return torch.zeros(self._n, dtype=torch.bool, device=self.device)

    def _list_flags(self):
        return ops.make_flags("list", noop_if_invalid=True)

    def _ended(self) -> torch.Tensor:
        return ops.get_num_points(self._points) <= 1

    def _exceeds(self) -> torch.Tensor:
        """ListPoints.exceed_threshold (list_points.py:64-75): any coordinate > value_threshold"""
        if self.value_threshold is None:
            return self._false()
        return self._points.amax(dim=(1, 2)) > self.value_threshold

    def _squeeze(self, x):
        """single-game interface: drop the batch axis and hand numpy / python values back"""
        if self.num_envs is not None:
            return x
        if isinstance(x, dict):
            return {k: self._squeeze(v) for k, v in x.items()}
        if isinstance(x, torch.Tensor):
            v = x[0].detach().cpu().numpy()
            return v.item() if v.ndim == 0 else v
        return x

    def _get_info(self) -> Dict:
        if self.improve_efficiency:
            return {}
        return {"step_threshold": self.step_threshold, "current_step": self.current_step,
                "exceed_threshold": self._squeeze(self.exceed_threshold),
                "last_action_taken": self._squeeze(self.last_action_taken)
                if isinstance(self.last_action_taken, torch.Tensor) else self.last_action_taken}

    def _get_padded_points(self) -> torch.Tensor:
        return self._points.to(torch.float32)

    def _get_coords_multi_bin(self) -> torch.Tensor:
        """hironaka_base.py:152-163: zero when the game has ended or fewer than 2 coordinates"""
        ok = (~self._ended()) & (self._coords.sum(dim=1) >= 2)
        return (self._coords * ok.unsqueeze(1).to(torch.int32)).to(torch.float64)

    # ---- gym protocol ------------------------------------------------------------------------
    def reset(self, points=None, seed=None, return_info=False, options=None) -> Any:
        """hironaka_base.py:86-114: random ints in [0, max_value) (or the given points) -> newton ->
        [rescale] -> newton -> _post_reset_update."""
        if seed is not None:
            self._seed = int(seed)
            self._episode = 0
        m, d = self.max_num_points, self.dimension
        if points is None:
            raw = ops.generate_points(self._n, m, d, self.max_value, self._seed, game_offset=self._episode * self._n,
                                      dtype=torch.float64, device=self.device, newton=False, reposition=False)
            self._episode += 1
        else:
            raw = torch.as_tensor(np.asarray(points, dtype=np.float64) if not isinstance(points, torch.Tensor)
                                  else points, dtype=torch.float64, device=self.device)
            if raw.dim() == 2:
                raw = raw.unsqueeze(0)
            if raw.shape[1] < m:  # ragged / shorter input: pad like get_padded_array
                padrows = torch.full((raw.shape[0], m - raw.shape[1], d), self.padding_value, dtype=torch.float64,
                                     device=self.device)
                raw = torch.cat([raw, padrows], dim=1)
            assert raw.shape == (self._n, m, d), f"points must have shape {(self._n, m, d)}"
        st = A.HK_STAGE_NEWTON | (A.HK_STAGE_RESCALE if self.scale_observation else 0)
        self._points = ops.step(raw.contiguous(), stages=st, flags=self._list_flags(),
                                padding_value=self.padding_value)["points"]
        self.current_step = 0
        self.exceed_threshold = self._false()
        self.last_action_taken = None
        if not self.improve_efficiency:
            self._points = ops.get_newton_polytope(self._points, self.padding_value, sem="list")
        self._post_reset_update()
        observation = self._squeeze(self._get_obs())
        return (observation, self._get_info()) if return_info else observation

    @abc.abstractmethod
    def _post_reset_update(self):
        ...

    @abc.abstractmethod
    def step(self, action):
        self.current_step += 1

    @abc.abstractmethod
    def _get_obs(self):
        ...

    def render(self, mode="ansi"):
        print(self._points)
        print(self._coords)

    def close(self):
        pass


class HironakaHostEnv(HironakaBase):
    """The environment fixes a Host; it receives axes from an agent (hironaka_host_env.py)."""

    def __init__(self, host: Host, invalid_move_penalty: float = -1e-3, stop_after_invalid_move: bool = False,
                 config_kwargs: Optional[Dict[str, Any]] = None, **kwargs):
        config_kwargs = dict() if config_kwargs is None else config_kwargs
        super().__init__(**{**config_kwargs, **kwargs})
        self.observation_space = spaces.Dict({"points": self.point_observation_space,
                                              "coords": spaces.MultiBinary(self.dimension)})
        self.action_space = spaces.Discrete(self.dimension)
        self.host = host
        self.invalid_move_penalty = invalid_move_penalty
        self.stop_after_invalid_move = stop_after_invalid_move

    def _post_reset_update(self):
        self.step(action=None)

    def step(self, action):
        super().step(action)
        n, d = self._n, self.dimension
        if action is None:
            act = torch.full((n,), -1, dtype=torch.int32, device=self.device)
        else:
            act = torch.as_tensor(action, device=self.device).reshape(n).to(torch.int32)
        in_range = (act >= 0) & (act < d)
        legal = in_range & (self._coords.gather(1, act.clamp(0, d - 1).long().unsqueeze(1)).squeeze(1) > 0)
        # shift + newton in one launch; an axis outside the subset leaves the (already reduced) game as is
        self._points = ops.step(self._points, self._coords, act, stages=A.HK_STAGE_SHIFT | A.HK_STAGE_NEWTON,
                                flags=self._list_flags(), padding_value=self.padding_value, out=self._points)["points"]
        ended = self._ended()
        reward = torch.where(legal, (~ended).to(torch.float64),
                             torch.full((n,), float(self.invalid_move_penalty), dtype=torch.float64, device=self.device))
        stopped = (~legal) & bool(self.stop_after_invalid_move)
        stopped = stopped | ended
        self.exceed_threshold = self._exceeds()
        stopped = stopped | self.exceed_threshold
        chosen = self.host.select_coord(self._points).to(torch.int32)
        self._coords = chosen * (~stopped).unsqueeze(1).to(torch.int32)
        if self.scale_observation:
            self._points = ops.rescale(self._points, self.padding_value, sem="list")
        self.last_action_taken = self._coords
        obs = self._get_obs()
        return self._squeeze(obs), self._squeeze(reward), self._squeeze(stopped), self._get_info()

    def _get_obs(self):
        return {"points": self._get_padded_points(), "coords": self._get_coords_multi_bin()}


class HironakaAgentEnv(HironakaBase):
    """The environment fixes an Agent; it receives coordinate subsets from a host (hironaka_agent_env.py)."""

    def __init__(self, agent: Agent, use_discrete_actions_for_host: Optional[bool] = False,
                 compressed_host_output: Optional[bool] = True, config_kwargs: Optional[Dict[str, Any]] = None,
                 **kwargs):
        config = kwargs if config_kwargs is None else {**kwargs, **config_kwargs}
        config = dict(config)
        self.use_discrete_actions_for_host = config.pop("use_discrete_actions_for_host", use_discrete_actions_for_host)
        super().__init__(**config)
        self.agent = agent
        self.compressed_host_output = compressed_host_output
        self.observation_space = self.point_observation_space
        if self.use_discrete_actions_for_host:
            n = 2 ** self.dimension - self.dimension - 1 if compressed_host_output else 2 ** self.dimension
            self.action_space = spaces.Discrete(n)
        else:
            self.action_space = spaces.MultiBinary(self.dimension)

    def _post_reset_update(self):
        pass

    def step(self, action):
        super().step(action)
        n, d = self._n, self.dimension
        action = torch.as_tensor(action, device=self.device)
        if self.use_discrete_actions_for_host:
            # decode_action (src/_fn.py:156-170): the raw binary expansion of the integer
            code = action.reshape(n).to(torch.int64)
            mask = ((code.unsqueeze(1) >> torch.arange(d, device=self.device)) & 1).to(torch.int32)
        else:
            mask = (action.reshape(n, d) == 1).to(torch.int32)
        before = ops.get_num_points(self._points)

        class _State:  # what Agent.move needs
            pass

        state = _State()
        state.points, state.padding_value = self._points, self.padding_value
        self.last_action_taken = self.agent.move(state, mask)
        self._points = state.points
        ended = self._ended()
        stopped = ended.clone()
        reward = torch.zeros(n, dtype=torch.float64, device=self.device)
        self.exceed_threshold = self._exceeds()
        if self.stop_at_threshold:
            trip = self.exceed_threshold | (self.current_step >= self.step_threshold)
            stopped = stopped | trip
            penalty = -float(self.step_threshold) if self.fixed_penalty_crossing_threshold is None \
                else float(self.fixed_penalty_crossing_threshold)
            reward = reward + trip.to(torch.float64) * penalty
        if self.scale_observation:
            self._points = ops.rescale(self._points, self.padding_value, sem="list")
        obs = self._get_obs()
        if self.reward_based_on_point_reduction:
            reward = reward + (before - ops.get_num_points(self._points)).to(torch.float64)
        reward = reward + ended.to(torch.float64)
        return self._squeeze(obs), self._squeeze(reward), self._squeeze(stopped), self._get_info()

    def _get_obs(self):
        return self._get_padded_points()
