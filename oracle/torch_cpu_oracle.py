"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Nothing under ``hironaka_amd/`` may import this file.

The JAX trainer's step (shift -> reposition -> Newton polytope, JAX semantics: hironaka/src/_jax_ops.py
and jax/util.py:117-123) restated as torch CPU tensor ops in the reference's own *array* formulation --
broadcast [B, m, m, d] difference tensors, boolean masks, any / all, multiply-add masking -- equal in
structure to hironaka/src/_torch_ops.py / _jax_ops.py.  It exists for ONE purpose: ``bench.py``'s
``cpu_baseline_array_formulation`` leg, "the reference's CPU algorithm" on all host cores (SURVEY.md 8(d),
CPU path (i)).  ``tests/test_oracle.py`` pins it to ``np_oracle`` (hence to the golden vectors).
"""
from __future__ import annotations

import numpy as np
import torch

from . import np_oracle as NO


def _mask_mix(values: torch.Tensor, keep: torch.Tensor, fill: float) -> torch.Tensor:
    """values*keep + (~keep)*fill (_jax_ops.py:40,70,90,111)"""
    k = keep.to(values.dtype)
    return values * k + (~keep).to(values.dtype) * fill


def shift(points: torch.Tensor, coords: torch.Tensor, axis: torch.Tensor, padding_value: float = -1.0):
    """shift_jax (_jax_ops.py:76-90)"""
    b, m, d = points.shape
    onehot = torch.arange(d)[None, :] == axis[:, None]
    s = torch.zeros((b, m), dtype=points.dtype)
    for k in range(d):  # coordinate order 0..d-1
        s = s + points[:, :, k] * coords[:, None, k]
    moved = s[:, :, None] * onehot[:, None, :].to(points.dtype) + points * (~onehot)[:, None, :].to(points.dtype)
    avail = (points >= 0).any(dim=2)
    return _mask_mix(moved, avail[:, :, None].expand_as(points), padding_value)


def reposition(points: torch.Tensor, padding_value: float = -1.0):
    """reposition_jax (_jax_ops.py:114-123)"""
    avail = points >= 0
    col_max = points.max(dim=1, keepdim=True).values
    modified = points * avail.to(points.dtype) + (~avail).to(points.dtype) * col_max
    col_min = modified.min(dim=1, keepdim=True).values
    moved = _mask_mix(points - col_min, avail, padding_value)
    return torch.where(col_min <= 0, points, moved)


def get_newton_polytope(points: torch.Tensor, padding_value: float = -1.0):
    """get_newton_polytope_jax (_jax_ops.py:60-73): remove_repeated (fill -1.0, :65) then get_interior"""
    b, m, d = points.shape
    same = (points[:, :, None, :] == points[:, None, :, :]).all(dim=3)
    earlier = torch.tril(torch.ones((m, m), dtype=torch.bool), diagonal=-1)[None]
    repeated = (same & earlier).any(dim=2)
    p = _mask_mix(points, ~repeated[:, :, None].expand_as(points), -1.0)
    avail = (p >= 0).all(dim=2)
    both = avail[:, :, None] & avail[:, None, :]
    diff = p[:, :, None, :] - p[:, None, :, :]
    off_diag = ~torch.eye(m, dtype=torch.bool)[None]
    dominated = ((diff >= 0).all(dim=3) & off_diag & both).any(dim=2)
    return _mask_mix(p, ~dominated[:, :, None].expand_as(p), padding_value)


def step(points: torch.Tensor, coords: torch.Tensor, axis: torch.Tensor, padding_value: float = -1.0):
    """take_actions (jax/util.py:117-123) as the JAX trainer configures it: reposition on, rescale off"""
    return get_newton_polytope(reposition(shift(points, coords, axis, padding_value), padding_value), padding_value)


def get_dones(points: torch.Tensor) -> torch.Tensor:
    """jax/util.py:34-35"""
    return (points[:, :, 0] >= 0).sum(dim=1) < 2


def rollout(points: np.ndarray, steps: int, seed: int, game_offset: int = 0):
    """`steps` moves of every game with the Philox random policies of np_oracle.policy_actions (numpy: a few
    integer ops per game); returns (final states, finished games after each step)."""
    p = torch.from_numpy(np.ascontiguousarray(points))
    d = p.shape[2]
    table = torch.from_numpy(NO.decode_table(d)).to(p.dtype)
    done_count = [int(get_dones(p).sum())]
    for t in range(steps):
        cls, ax = NO.policy_actions(p.numpy(), t, seed, game_offset, NO.HOST_RANDOM, NO.AGENT_RANDOM)
        p = step(p, table[torch.from_numpy(cls).long()], torch.from_numpy(ax).long())
        done_count.append(int(get_dones(p).sum()))
    return p.numpy(), np.array(done_count, dtype=np.uint64)
