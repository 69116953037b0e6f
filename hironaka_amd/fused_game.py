"""``FusedGame`` -- the DQN trainers' roll-out step over a large batch of games, the counterpart of
``hironaka/trainer/fused_game.py:10-193`` (same constructor, ``step`` / ``host_move`` / ``agent_move``
signatures, outputs and reward convention), over a ``HipPoints`` container.

Per ``step`` the reference runs the features (argsort + gather) four to five times and the point operations
as three tensor programs (shift, Newton polytope, rescale: fused_game.py:150-163).  Here the features of a
state are one launch of ``hk_get_features_torch`` and are computed once per state, and the move is ONE fused
launch (``HipPoints.step``).  The two networks are the caller's ``torch.nn.Module``s and run as they are.
"""
import time
from copy import deepcopy
from typing import Callable, Optional, Tuple, Type, Union

import torch

from .core import HipPoints
from .host_action_preprocess import HostActionEncoder


class Timer:
    """trainer/timer.py: accumulates milliseconds into an external dict"""

    def __init__(self, name: str, log_dict: dict, active=True, use_cuda=False):
        self.name, self._log, self._active, self._cuda = name, log_dict, active, use_cuda

    def __enter__(self):
        if self._active:
            if self._cuda:
                torch.cuda.current_stream().synchronize()
            self._start = time.perf_counter()
            self._log.setdefault(self.name, 0.0)

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self._active:
            if self._cuda:
                torch.cuda.current_stream().synchronize()
            self._log[self.name] += (time.perf_counter() - self._start) * 1000


class FusedGame:
    def __init__(self, host_net: torch.nn.Module, agent_net: torch.nn.Module,
                 device: Optional[Union[str, torch.device]] = "cuda", log_time: Optional[bool] = True,
                 reward_func: Optional[Callable] = None,
                 dtype: Optional[Union[Type, torch.dtype]] = torch.float32):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise TypeError(f"FusedGame steps HipPoints on a HIP device (there is no CPU path). Got {self.device}.")
        self.use_cuda = True
        self.host_net = host_net.to(self.device)
        self.agent_net = agent_net.to(self.device)
        self.log_time = log_time
        if reward_func is None:
            self._rewards = self._default_reward
        else:
            assert callable(reward_func), f"reward_function must be callable. Got {type(reward_func)}."
            self._rewards = reward_func
        self.dtype = dtype
        self._make_type_for_nets(self.dtype)
        self.host_action_encoder = None
        self.time_log = dict()

    def _timer(self, name):
        return Timer(name, self.time_log, active=self.log_time, use_cuda=self.use_cuda)

    def step(self, points: HipPoints, sample_for: str, masked=True, scale_observation=True, exploration_rate=0.2):
        """observations, actions (of `sample_for`), rewards, dones, next_observations of the games that were
        not finished before the move (fused_game.py:54-102)"""
        assert sample_for in ["host", "agent"], f"sample_for must be one of 'host' and 'agent'. Got {sample_for}."
        if points.dtype != self.dtype:
            points.type(self.dtype)
        with self._timer("step-get_features_total"):
            observations = points.get_features()
        done = points.ended_batch_in_tensor
        # only the sampled side explores
        with self._timer("step-host_move"):
            host_move, chosen_actions = self.host_move(
                points, exploration_rate=exploration_rate if sample_for == "host" else 0.0, features=observations)
        with self._timer("step-agent_move"):
            agent_move = self.agent_move(
                points, host_move, masked=masked, scale_observation=scale_observation, inplace=True,
                exploration_rate=exploration_rate if sample_for == "agent" else 0.0, features=observations)
        next_done = points.ended_batch_in_tensor
        with self._timer("step-get_features_total"):
            next_observations = points.get_features()
        keep = ~done
        if sample_for == "host":
            with self._timer("step-host_postprocess_exps"):
                output_obs = observations[keep]
                output_actions = chosen_actions[keep]
                next_out = next_observations[keep]
        else:
            with self._timer("step-agent_extra_host_move"):
                next_host_move, _ = self.host_move(points, exploration_rate=exploration_rate,
                                                   features=next_observations)
            with self._timer("step-agent_postprocess_exps"):
                output_obs = {"points": observations[keep], "coords": host_move[keep]}
                output_actions = agent_move[keep]
                next_out = {"points": next_observations[keep], "coords": next_host_move[keep]}
        next_done = next_done[keep]
        return (output_obs, output_actions.reshape(-1, 1),
                self._rewards(sample_for, output_obs, next_out, next_done).reshape(-1, 1),
                next_done.reshape(-1, 1), next_out)

    def host_move(self, points: HipPoints, masked=True, exploration_rate=0.0,
                  features: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """the host net's argmax class (uniform noise instead with probability `exploration_rate`) and its
        multi-binary mask (fused_game.py:104-124); `features`: the state's features if already computed"""
        if features is None:
            features = points.get_features()
        with self._timer("host_move-host_net_inference"):
            with torch.inference_mode():
                output = self.host_net(features.to(self.device))
        if self.host_action_encoder is None:
            self.host_action_encoder = HostActionEncoder(points.dimension)
        noise = torch.rand(output.shape, device=self.device, dtype=output.dtype)
        random_mask = torch.rand(output.shape[0], 1, device=self.device).le(exploration_rate)
        output = output * ~random_mask + noise * random_mask
        with self._timer("host_move-decode_tensor"):
            # noise goes through the decoder too: explored moves are never illegal
            chosen_actions = torch.argmax(output, dim=1).type(torch.int32)
            host_move_binary = self.host_action_encoder.decode_tensor(chosen_actions, dtype=self.dtype)
        return host_move_binary, chosen_actions

    def agent_move(self, points: HipPoints, host_moves: torch.Tensor, masked: Optional[bool] = True,
                   scale_observation: Optional[bool] = True, inplace: Optional[bool] = True,
                   exploration_rate: Optional[float] = 0.0, features: Optional[torch.Tensor] = None) -> torch.Tensor:
        """the agent net's argmax axis (restricted to the host's subset if `masked`; a uniform axis with
        probability `exploration_rate`), and -- `inplace` -- the move itself: shift -> Newton polytope ->
        [rescale] in one launch (fused_game.py:126-163)"""
        if features is None:
            features = points.get_features()
        with self._timer("agent_move-agent_net_inference"):
            with torch.inference_mode():
                action_prob = self.agent_net({"points": features.to(self.device),
                                              "coords": host_moves.to(self.device)})
        if masked:
            minimum = torch.finfo(action_prob.dtype).min
            action_prob = action_prob * host_moves + (1 - host_moves) * minimum
        actions = torch.argmax(action_prob, dim=1)
        noise = torch.randint(0, action_prob.shape[1], actions.shape, device=actions.device, dtype=actions.dtype)
        random_mask = torch.rand(actions.shape[0], device=actions.device).le(exploration_rate)
        actions = actions * ~random_mask + noise * random_mask
        with self._timer("agent_move-point_operations"):
            if inplace:
                points.step(host_moves, actions, rescale=bool(scale_observation))
        return actions

    def _make_type_for_nets(self, dtype: torch.dtype):
        """a net whose parameters have another dtype is copied and recast (the caller's stays untouched)"""
        for role in ["host", "agent"]:
            net = getattr(self, f"{role}_net")
            param = next(net.parameters(), None)
            if param is not None and param.dtype != dtype:
                setattr(self, f"{role}_net", deepcopy(net).type(dtype))

    @staticmethod
    def _default_reward(sample_for: str, obs, next_obs, next_done: torch.Tensor) -> torch.Tensor:
        if sample_for == "host":
            return next_done.type(torch.float32).clone()
        elif sample_for == "agent":
            return (-next_done.type(torch.float32)).clone()
