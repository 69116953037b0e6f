"""Host action codec on device tensors -- the counterpart of
``hironaka/jax/host_action_preprocess.py`` (same names and meaning).

A host action is a subset of >= 2 coordinates.  It travels either as a multi-binary mask
``[..., dim]`` or as a compressed class id in ``[0, 2**dim - dim - 1)``: the c-th integer in
``1 .. 2**dim - 1`` that is not a power of two, bit j <-> coordinate j
(host_action_preprocess.py:8-24).  Decoding runs in ``hk_decode_host_class``; the kernels of
``hk_step`` decode class ids themselves, so the mask never has to exist in HBM.
"""
from __future__ import annotations

import functools
from typing import Callable, Dict, Optional

import torch

from . import ops

_MAX_DIM = 11  # host_action_preprocess.py:27


def _check_dim(dimension: int) -> None:
    if dimension >= _MAX_DIM:
        raise ValueError(f"Dimension is capped at {_MAX_DIM}. Got {dimension}.")
    if dimension < 2:
        raise ValueError(f"A host needs at least 2 coordinates. Got {dimension}.")


def num_classes(dimension: int) -> int:
    return 2 ** dimension - dimension - 1


def decode_table(dimension: int, device="cuda", dtype=torch.int32) -> torch.Tensor:
    """The i-th row is the multi-binary vector of class i (host_action_preprocess.py:8-24)."""
    _check_dim(dimension)
    ids = torch.arange(num_classes(dimension), dtype=torch.int32, device=device)
    return ops.decode_host_class(ids, dimension, dtype)


class _LazyTables(dict):
    """``dec_table[d]`` of the reference, built on first use on the current device."""

    def __missing__(self, dimension):
        if dimension in (0, 1):
            return None
        table = decode_table(dimension)
        self[dimension] = table
        return table


dec_table: Dict[int, Optional[torch.Tensor]] = _LazyTables()


@functools.lru_cache()
def get_batch_decode(dimension: int) -> Callable:
    """class ids [B] -> masks [B, dim] (host_action_preprocess.py:59-65)."""
    _check_dim(dimension)

    def batch_decode(cls: torch.Tensor, dtype=torch.int32) -> torch.Tensor:
        return ops.decode_host_class(cls, dimension, dtype)

    return batch_decode


@functools.lru_cache()
def get_batch_decode_from_one_hot(dimension: int) -> Callable:
    """one-hot class vectors [B, A] -> masks [B, dim] (host_action_preprocess.py:69-75)."""
    _check_dim(dimension)

    def batch_decode_from_one_hot(one_hot: torch.Tensor, dtype=torch.int32) -> torch.Tensor:
        return ops.decode_host_class(torch.argmax(one_hot, dim=-1), dimension, dtype)

    return batch_decode_from_one_hot


def decode_from_one_hot(one_hot: torch.Tensor, lookup_dict: torch.Tensor) -> torch.Tensor:
    """one one-hot class vector -> its multi-binary mask, through a decode table (host_action_preprocess.py:38-45)"""
    return lookup_dict[torch.argmax(one_hot)]


def decode(cls: int, lookup_dict: torch.Tensor) -> torch.Tensor:
    """one class id -> its multi-binary mask (host_action_preprocess.py:48-52)"""
    return lookup_dict[cls]


def batch_encode(multi_binary: torch.Tensor) -> torch.Tensor:
    """masks [B, dim] -> class ids: v - floor(log2 v) - 2 with v = sum 2^j m_j
    (host_action_preprocess.py:78-87, src/_fn.py:282-292)."""
    dimension = multi_binary.shape[-1]
    weights = 2 ** torch.arange(dimension, device=multi_binary.device, dtype=torch.int64)
    v = (multi_binary.to(torch.int64) * weights).sum(dim=-1)
    # exact integer floor(log2 v): position of the highest set bit
    top = torch.zeros_like(v)
    for bit in range(1, dimension + 1):
        top = torch.where(v >= (1 << bit), torch.full_like(v, bit), top)
    return (v - top - 2).to(torch.int32)


def encode(multi_binary: torch.Tensor) -> torch.Tensor:
    return batch_encode(multi_binary.reshape(1, -1))[0]


def batch_encode_one_hot(multi_binary: torch.Tensor) -> torch.Tensor:
    """masks [B, dim] -> one-hot class vectors [B, A] float32 (host_action_preprocess.py:90-99)."""
    dimension = multi_binary.shape[-1]
    cls = batch_encode(multi_binary).long()
    return torch.nn.functional.one_hot(cls, num_classes(dimension)).to(torch.float32)


def encode_one_hot(multi_binary: torch.Tensor) -> torch.Tensor:
    return batch_encode_one_hot(multi_binary.reshape(1, -1))[0]


class HostActionEncoder:
    """The torch trainers' codec object (src/_fn.py:241-325): ``encode`` / ``decode`` between a list of chosen
    coordinates and the class id, ``encode_tensor`` / ``decode_tensor`` between [B, dim] 0/1 masks and [B]
    class ids on device tensors (decode_tensor runs ``hk_decode_host_class``)."""

    def __init__(self, dim: int = 3):
        _check_dim(dim)
        self.dim = dim
        # the class ids' integers: 1 .. 2^dim - 1 without the powers of two (src/_fn.py:256-259)
        self.action_translate = [v for v in range(1, 2 ** dim) if v & (v - 1)]

    def encode(self, coords) -> int:
        assert len(coords) > 1
        v = sum(1 << int(c) for c in coords)
        return v - (v.bit_length() - 1) - 2

    def encode_tensor(self, coords: torch.Tensor) -> torch.Tensor:
        assert coords.dim() == 2
        return batch_encode(coords)

    def decode(self, action: int):
        assert 0 <= action < num_classes(self.dim)
        v = self.action_translate[action]
        return [k for k in range(self.dim) if (v >> k) & 1]

    def decode_tensor(self, actions: torch.Tensor, dtype: torch.dtype = torch.float32) -> torch.Tensor:
        assert actions.dim() == 1
        return ops.decode_host_class(actions, self.dim, dtype)
