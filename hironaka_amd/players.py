"""Fixed host and agent policies on device tensors -- the counterpart of
``hironaka/jax/players.py`` (same names, same one-hot outputs).

Hosts:  random_host_fn, all_coord_host_fn, zeillinger_fn      (pts [B, m, d] -> one-hot [B, A])
Agents: random_agent_fn, choose_first_agent_fn, choose_last_agent_fn
        (flattened agent observation [B, m*d + d] -> one-hot [B, d])

`key` is an integer seed or a torch.Generator (JAX PRNG keys do not exist here).  The same policies
are also fused into ``hk_rollout`` (``ops.rollout(host_policy=..., agent_policy=...)``), which is
what a rollout should use; these functions exist for callers that drive the environment step by
step (recurrent_fn, compute_rho with arbitrary callables).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple, Union

import torch

from . import ops
from .host_action_preprocess import num_classes

Key = Optional[Union[int, torch.Generator]]


def _generator(key: Key, device) -> Optional[torch.Generator]:
    if key is None or isinstance(key, torch.Generator):
        return key
    g = torch.Generator(device=device)
    g.manual_seed(int(key) % (1 << 63))
    return g


def get_name(obj):
    if hasattr(obj, "__name__"):
        return obj.__name__
    if hasattr(obj, "func"):  # functools.partial
        return get_name(obj.func)
    return type(obj).__name__


# ---------- hosts ---------- #

def random_host_fn(pts: torch.Tensor, key: Key = 0, dtype=torch.float32, **kwargs) -> torch.Tensor:
    """players.py:28-39 -- uniform class id."""
    batch_size, _, dimension = pts.shape
    n = num_classes(dimension)
    cls = torch.randint(0, n, (batch_size,), device=pts.device, generator=_generator(key, pts.device))
    return torch.nn.functional.one_hot(cls, n).to(dtype)


def all_coord_host_fn(pts: torch.Tensor, dtype=torch.float32, **kwargs) -> torch.Tensor:
    """players.py:42-52 -- always the full coordinate set (the last class)."""
    batch_size, _, dimension = pts.shape
    n = num_classes(dimension)
    cls = torch.full((batch_size,), n - 1, device=pts.device, dtype=torch.long)
    return torch.nn.functional.one_hot(cls, n).to(dtype)


def zeillinger_fn(pts: torch.Tensor, dtype=torch.float32, **kwargs) -> torch.Tensor:
    """players.py:84-109 -- Zeillinger's choice per game (hk_zeillinger)."""
    n = num_classes(pts.shape[-1])
    return torch.nn.functional.one_hot(ops.zeillinger(pts).long(), n).to(dtype)


def zeillinger_fn_slice(pts: torch.Tensor) -> torch.Tensor:
    """players.py:84-105 -- one game without batch axis."""
    return zeillinger_fn(pts.unsqueeze(0))[0]


def get_host_with_flattened_obs(spec: Tuple[int, int], func: Callable, truncate_input: bool = False,
                                dtype=torch.float32) -> Callable:
    """players.py:112-137 -- pre-compose a reshape (and the removal of a padded coordinate tail)."""
    m, d = spec

    def func_flatten(pts, *args, dtype=dtype, **kwargs):
        if truncate_input:
            pts = pts[..., :-d]
        return func(pts.reshape(*pts.shape[:-1], m, d), *args, dtype=dtype, **kwargs)

    func_flatten.__name__ = get_name(func)
    return func_flatten


# ---------- agents ---------- #

def random_agent_fn(pts: torch.Tensor, spec: Tuple[int, int], key: Key = 0, dtype=torch.float32,
                    **kwargs) -> torch.Tensor:
    """players.py:142-153 -- uniform over ALL `dim` axes (not restricted to the host's subset)."""
    (_, dimension), batch_size = spec, pts.shape[0]
    ax = torch.randint(0, dimension, (batch_size,), device=pts.device, generator=_generator(key, pts.device))
    return torch.nn.functional.one_hot(ax, dimension).to(dtype)


def choose_first_agent_fn(pts: torch.Tensor, spec: Tuple[int, int], dtype=torch.float32, **kwargs) -> torch.Tensor:
    """players.py:156-183 -- lowest coordinate of the host's subset (the observation's tail)."""
    m, d = spec
    host_action = pts[:, m * d: m * d + d]
    return torch.nn.functional.one_hot(torch.argmax(host_action, dim=1), d).to(dtype)


def choose_first_agent_fn_slice(pts: torch.Tensor, spec: Tuple[int, int]) -> torch.Tensor:
    """players.py:156-170 -- one observation without batch axis"""
    return choose_first_agent_fn(pts.unsqueeze(0), spec)[0]


def choose_last_agent_fn_slice(pts: torch.Tensor, spec: Tuple[int, int]) -> torch.Tensor:
    """players.py:185-199 -- one observation without batch axis"""
    return choose_last_agent_fn(pts.unsqueeze(0), spec)[0]


def char_vector(v1: torch.Tensor, v2: torch.Tensor) -> torch.Tensor:
    """players.py:55-77 -- Zeillinger's characteristic vector (L, S) of a pair of points: L = max - min of
    v1 - v2, S = #max + #min; a pair with a negative entry (an unavailable point) or with max ~ min gets
    (inf, inf).  (hk_zeillinger computes the same thing over all pairs in registers.)"""
    diff = v1 - v2
    mx, mn = diff.max(), diff.min()
    bad = bool((v1 < 0).any() | (v2 < 0).any() | torch.isclose(mx, mn))
    if bad:
        return torch.full((2,), float("inf"), dtype=diff.dtype, device=diff.device)
    s = (diff == mx).sum() + (diff == mn).sum()
    return torch.stack([mx - mn, s.to(diff.dtype)])


def choose_last_agent_fn(pts: torch.Tensor, spec: Tuple[int, int], dtype=torch.float32, **kwargs) -> torch.Tensor:
    """players.py:186-212 -- highest coordinate of the host's subset."""
    m, d = spec
    host_action = pts[:, m * d: m * d + d].to(torch.float32)
    eps = 1e-5
    bumped = host_action + torch.arange(d, device=pts.device, dtype=torch.float32) * eps
    return torch.nn.functional.one_hot(torch.argmax(bumped, dim=1), d).to(dtype)
