"""Fixed hosts for the gym / game surfaces, vectorised over a batch of games -- the counterparts of
``hironaka/host.py`` (`Host.select_coord`, `RandomHost`, `AllCoordHost`, `Zeillinger`).

``select_coord(points)`` takes the padded state ``[B, m, d]`` (device tensor or a container with a
``.points`` attribute) and returns the chosen coordinate subsets as a multi-binary mask ``[B, d]``
(int32).  A game with fewer than two points gets the empty subset (all zeros), the vectorised form of
the reference's ``[]`` (host.py:78-80).
"""
import abc
from typing import Optional, Union

import torch

from . import ops


def _as_points(points) -> torch.Tensor:
    return points.points if hasattr(points, "points") else points


class Host(abc.ABC):
    def select_coord(self, points, debug=False) -> torch.Tensor:
        pts = _as_points(points)
        mask = self._select_coord(pts)
        alive = ops.get_num_points(pts) >= 2
        return mask * alive.unsqueeze(1).to(mask.dtype)

    @abc.abstractmethod
    def _select_coord(self, points: torch.Tensor) -> torch.Tensor:
        ...


class RandomHost(Host):
    """host.py:42-45 -- two distinct coordinates, uniformly."""

    def __init__(self, seed: Optional[Union[int, torch.Generator]] = None):
        self._gen = seed if isinstance(seed, torch.Generator) or seed is None else None
        self._seed = seed if isinstance(seed, int) else None

    def _generator(self, device):
        if self._gen is None and self._seed is not None:
            self._gen = torch.Generator(device=device)
            self._gen.manual_seed(self._seed)
        return self._gen

    def _select_coord(self, points: torch.Tensor) -> torch.Tensor:
        b, _, d = points.shape
        keys = torch.rand((b, d), device=points.device, generator=self._generator(points.device))
        pick = keys.argsort(dim=1)[:, :2]
        mask = torch.zeros((b, d), dtype=torch.int32, device=points.device)
        return mask.scatter_(1, pick, 1)


class AllCoordHost(Host):
    """host.py:48-51"""

    def _select_coord(self, points: torch.Tensor) -> torch.Tensor:
        b, _, d = points.shape
        return torch.ones((b, d), dtype=torch.int32, device=points.device)


class Zeillinger(Host):
    """host.py:54-95 -- characteristic vector (L, S) over the pairs of points, smallest first;
    the subset is {argmin, argmax} of that pair's difference (hk_zeillinger, list semantics)."""

    @staticmethod
    def get_char_vector(vt):
        mx, mn = max(vt), min(vt)
        return mx - mn, sum(v == mx for v in vt) + sum(v == mn for v in vt)

    def _select_coord(self, points: torch.Tensor) -> torch.Tensor:
        d = points.shape[2]
        cls = ops.zeillinger(points, sem="list")
        mask = ops.decode_host_class(cls.clamp(min=0), d, torch.int32)
        return mask * (cls >= 0).unsqueeze(1).to(torch.int32)


class PolicyHost(Host):
    """host.py:98-113 -- a host that asks a policy object: ``policy.predict(features)`` on the container's
    features (``get_features()``) returns the multi-binary subsets [B, d]."""

    def __init__(self, policy, use_discrete_actions_for_host: Optional[bool] = False, **kwargs):
        self._policy = policy
        self.use_discrete_actions_for_host = kwargs.get("use_discrete_actions_for_host", use_discrete_actions_for_host)

    def select_coord(self, points, debug=False) -> torch.Tensor:
        self._features = points.get_features() if hasattr(points, "get_features") else _as_points(points)
        return super().select_coord(points, debug)

    def _select_coord(self, points: torch.Tensor) -> torch.Tensor:
        coords = torch.as_tensor(self._policy.predict(self._features), device=points.device)
        return (coords == 1).to(torch.int32)
