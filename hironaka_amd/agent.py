"""Fixed agents for the gym / game surfaces, vectorised -- the counterparts of ``hironaka/agent.py``
(`Agent.move`, `RandomAgent`, `ChooseFirstAgent`).

``move(points, coords)`` takes a ``HipPoints``-like container (``.points`` [B, m, d]) and the host's
subsets as a multi-binary mask [B, d]; it chooses one axis per game (-1 = "None": the host offered
fewer than two coordinates, agent.py:89,97), and -- with ``inplace`` -- applies
shift -> [reposition] -> Newton polytope to the container in ONE fused launch (agent.py:69-72).
"""
import abc
from typing import Optional, Union

import torch

from . import _abi as A
from . import ops


class Agent(abc.ABC):
    USE_REPOSITION: bool = False

    def move(self, points, coords: torch.Tensor, inplace: bool = True, sem: str = "list") -> torch.Tensor:
        pts = points.points if hasattr(points, "points") else points
        coords = coords.to(pts.device)
        actions = self._get_actions(pts, coords)
        actions = torch.where(coords.sum(dim=1) > 1, actions, torch.full_like(actions, -1))
        if not inplace:
            return actions
        flags = ops.make_flags(getattr(points, "semantics", sem), noop_if_invalid=True)
        stages = ops.make_stages(True, self.USE_REPOSITION, True, False)
        pad = getattr(points, "padding_value", -1.0)
        res = ops.step(pts, coords, actions, stages=stages, flags=flags, padding_value=pad, out=pts)
        if hasattr(points, "points"):
            points.points = res["points"]
        return actions

    @abc.abstractmethod
    def _get_actions(self, points: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
        ...


class RandomAgent(Agent):
    """agent.py:85-90 -- uniform over the host's subset."""

    def __init__(self, seed: Optional[Union[int, torch.Generator]] = None):
        self._gen = seed if isinstance(seed, torch.Generator) else None
        self._seed = seed if isinstance(seed, int) else None

    def _get_actions(self, points, coords):
        if self._gen is None and self._seed is not None:
            self._gen = torch.Generator(device=points.device)
            self._gen.manual_seed(self._seed)
        noise = torch.rand(coords.shape, device=points.device, generator=self._gen) + 1e-6
        return torch.argmax(noise * (coords > 0), dim=1).to(torch.int32)


class ChooseFirstAgent(Agent):
    """agent.py:93-98 -- the lowest coordinate of the subset."""

    def _get_actions(self, points, coords):
        return torch.argmax((coords > 0).to(torch.int32), dim=1).to(torch.int32)


class PolicyAgent(Agent):
    """agent.py:101-111 -- an agent that asks a policy object: ``policy.predict((features, coords))`` returns
    one axis per game."""

    def __init__(self, policy, **kwargs):
        self._policy = policy

    def move(self, points, coords: torch.Tensor, inplace: bool = True, sem: str = "list") -> torch.Tensor:
        self._features = points.get_features() if hasattr(points, "get_features") else None
        return super().move(points, coords, inplace, sem)

    def _get_actions(self, points, coords):
        features = points if self._features is None else self._features
        return torch.as_tensor(self._policy.predict((features, coords)), device=points.device).to(torch.int32)
