"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy-facing binding of ``libhironaka_oracle.so``
(``hironaka_oracle.c``).  Takes the same descriptors as the HIP library, with host pointers.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from hironaka_amd import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhironaka_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc the oracle (and, when /root/reference exists, oracle/_ref) via oracle/Makefile."""
    srcs = [os.path.join(_HERE, f) for f in ("hironaka_oracle.c", "hko_impl.inc")]
    stale = (not os.path.exists(_LIB_PATH)
             or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        protos = {}
        for name, (res, args) in A.PROTOTYPES.items():
            if name in ("hk_abi_version", "hk_strerror", "hk_has_fast_path", "hk_rollout_workspace_bytes",
                        "hk_rollout_reduce_counts", "hk_rollout_values", "hk_bin_group_games", "hk_bin_unit_games",
                        "hk_generate_points_binned", "hk_bin_by_live_rows") or name.startswith("hk_search_"):
                continue  # launch plumbing / restated in oracle/search_oracle.py and np_oracle.rollout_postprocess /
                          # the binning's group geometry is the kernels' own: hko_bin_by_live_rows takes it as arguments
            args = list(args[:-1])  # no stream on the CPU
            protos["hko_" + name[3:]] = (res, args)
        A.bind(_lib, protos)
        _lib.hko_set_threads.restype = C.c_int
        _lib.hko_set_threads.argtypes = [C.c_int]
        _lib.hko_bin_by_live_rows.restype = C.c_int
        _lib.hko_bin_by_live_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int]
    return _lib


def set_threads(n: int) -> int:
    return lib().hko_set_threads(n)


_NP2HK = {np.dtype(np.float32): A.HK_F32, np.dtype(np.float64): A.HK_F64,
          np.dtype(np.int32): A.HK_I32, np.dtype(np.int64): A.HK_I64, np.dtype(np.uint8): A.HK_U8}


def _hk_dtype(a: np.ndarray) -> int:
    return _NP2HK[a.dtype]


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def _check(st: int):
    if st != 0:
        raise RuntimeError(f"oracle status {st}: {A.STATUS_TEXT.get(st, '?')}")


def flags_of(sem: str = "jax", noop_if_invalid=False, ignore_ended=False, compact_sorted=False) -> int:
    f = A.SEMANTICS[sem]
    if noop_if_invalid:
        f |= A.HK_FLAG_AXIS_NOOP_IF_INVALID
    if ignore_ended:
        f |= A.HK_FLAG_IGNORE_ENDED
    if compact_sorted:
        f |= A.HK_FLAG_COMPACT_SORTED
    return f


def step(points: np.ndarray, coords=None, axis=None, *, stages: int, flags: int = 0,
         padding_value: float = -1.0, coords_kind: Optional[int] = None, reward_sign: float = 1.0,
         max_points: Optional[int] = None, dim: Optional[int] = None, axis_logits: bool = False,
         features: Optional[bool] = None):
    """Run hko_step.  `points` is [B, m, d] (or a [B, stride] record matrix when max_points/dim
    are given, e.g. an agent observation with the mask in its tail).
    axis_logits: `axis` is [B, d] float32 logits of the agent (HK_AXIS_MASKED_LOGITS: its move is the argmax over the
    host's subset).  features: None, or scale_observation -- hko_step_features, the result's observation features
    as `features` [B, m*d].
    Returns dict(points, done, prev_done, reward, num_points[, features])."""
    points = np.ascontiguousarray(points)
    if points.ndim == 3:
        b, m, d = points.shape
        in_stride = m * d
    else:
        b, in_stride = points.shape
        m, d = max_points, dim
    out = np.empty((b, m, d), dtype=points.dtype)
    done = np.empty(b, dtype=np.uint8)
    prev = np.empty(b, dtype=np.uint8)
    rew = np.empty(b, dtype=np.float32)
    npts = np.empty(b, dtype=np.int32)
    s = A.hk_step_desc()
    s.points_in, s.points_out = _ptr(points), _ptr(out)
    s.in_stride, s.out_stride = in_stride, m * d
    keep = []
    if coords_kind is None:
        coords_kind = A.HK_COORDS_NONE
        if coords is not None:
            coords = np.ascontiguousarray(coords)
            coords_kind = _hk_dtype(coords) if coords.ndim == 2 else (
                A.HK_COORDS_CLASS_I32 if coords.dtype == np.int32 else A.HK_COORDS_CLASS_I64)
    if coords is not None:
        coords = np.ascontiguousarray(coords)
        keep.append(coords)
        s.coords = _ptr(coords)
        s.coords_stride = d
    s.coords_kind = coords_kind
    if axis is not None:
        axis = np.ascontiguousarray(axis)
        keep.append(axis)
        s.axis = _ptr(axis)
        s.axis_dtype = A.HK_AXIS_MASKED_LOGITS if axis_logits else _hk_dtype(axis)
    s.done_out, s.prev_done_out = _ptr(done), _ptr(prev)
    s.reward_out, s.num_points_out = _ptr(rew), _ptr(npts)
    s.padding_value, s.reward_sign = padding_value, reward_sign
    s.batch, s.max_points, s.dim, s.dtype = b, m, d, _hk_dtype(points)
    s.stages, s.flags = stages, flags
    res = dict(points=out, done=done.astype(bool), prev_done=prev.astype(bool), reward=rew, num_points=npts)
    if features is None:
        _check(lib().hko_step(C.byref(s)))
    else:
        feat = np.empty((b, m * d), dtype=points.dtype)
        _check(lib().hko_step_features(C.byref(s), _ptr(feat), int(bool(features))))
        res["features"] = feat
    res["done"], res["prev_done"] = done.astype(bool), prev.astype(bool)
    return res


def shift(points, coords, axis, padding_value=-1.0, **kw):
    return step(points, coords, axis, stages=A.HK_STAGE_SHIFT, flags=flags_of(**kw),
                padding_value=padding_value)["points"]


def reposition(points, padding_value=-1.0, **kw):
    return step(points, stages=A.HK_STAGE_REPOSITION, flags=flags_of(**kw), padding_value=padding_value)["points"]


def get_newton_polytope(points, padding_value=-1.0, **kw):
    return step(points, stages=A.HK_STAGE_NEWTON, flags=flags_of(**kw), padding_value=padding_value)["points"]


def rescale(points, padding_value=-1.0, **kw):
    return step(points, stages=A.HK_STAGE_RESCALE, flags=flags_of(**kw), padding_value=padding_value)["points"]


def get_dones(points: np.ndarray) -> np.ndarray:
    points = np.ascontiguousarray(points)
    b, m, d = points.shape
    out = np.empty(b, dtype=np.uint8)
    _check(lib().hko_get_dones(_ptr(points), m * d, _ptr(out), b, m, d, _hk_dtype(points)))
    return out.astype(bool)


def get_num_points(points: np.ndarray) -> np.ndarray:
    points = np.ascontiguousarray(points)
    b, m, d = points.shape
    out = np.empty(b, dtype=np.int32)
    _check(lib().hko_get_num_points(_ptr(points), m * d, _ptr(out), b, m, d, _hk_dtype(points)))
    return out


def decode_host_class(cls: np.ndarray, dim: int, dtype=np.int32) -> np.ndarray:
    cls = np.ascontiguousarray(cls, dtype=np.int32)
    out = np.empty((len(cls), dim), dtype=dtype)
    _check(lib().hko_decode_host_class(_ptr(cls), _ptr(out), _NP2HK[np.dtype(dtype)], len(cls), dim))
    return out


def zeillinger(points: np.ndarray, sem: str = "jax") -> np.ndarray:
    points = np.ascontiguousarray(points)
    b, m, d = points.shape
    out = np.empty(b, dtype=np.int32)
    _check(lib().hko_zeillinger(_ptr(points), m * d, _ptr(out), b, m, d, _hk_dtype(points),
                                A.SEMANTICS[sem]))
    return out


def get_features(points: np.ndarray, scale_observation=True, padding_value=-1.0) -> np.ndarray:
    points = np.ascontiguousarray(points)
    b, m, d = points.shape
    out = np.empty((b, m * d), dtype=points.dtype)
    _check(lib().hko_get_features(_ptr(points), m * d, _ptr(out), m * d, b, m, d, _hk_dtype(points),
                                  int(scale_observation), padding_value))
    return out


def get_features_torch(points: np.ndarray, padding_value=-1.0) -> np.ndarray:
    """TensorPoints.get_features (core/tensor_points.py:72-74), ties in row order; [B, m, d]"""
    points = np.ascontiguousarray(points)
    b, m, d = points.shape
    out = np.empty((b, m, d), dtype=points.dtype)
    _check(lib().hko_get_features_torch(_ptr(points), m * d, _ptr(out), m * d, b, m, d, _hk_dtype(points),
                                        padding_value))
    return out


def generate_points(batch, max_points, dim, max_value, seed, game_offset=0, dtype=np.float32,
                    stages=A.HK_STAGE_NEWTON | A.HK_STAGE_REPOSITION, padding_value=-1.0, flags=0):
    out = np.empty((batch, max_points, dim), dtype=dtype)
    _check(lib().hko_generate_points(_ptr(out), batch, max_points, dim, _NP2HK[np.dtype(dtype)],
                                     max_value, seed, game_offset, stages, padding_value, flags))
    return out


def rollout(points: np.ndarray, steps: int, seed: int, *, game_offset=0, step_offset=0,
            host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM,
            stages=A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON, flags=0,
            padding_value=-1.0, reward_sign=1.0, record=True, game_ids=None):
    p = np.array(points, copy=True, order="C")
    b, m, d = p.shape
    r = A.hk_rollout_desc()
    if game_ids is not None:
        ids = np.ascontiguousarray(game_ids, dtype=np.int32)
        assert ids.shape == (b,)
        r.game_ids = _ptr(ids)
    counts = np.zeros(steps + 1, dtype=np.uint64)
    rec = dict(done_count=counts, game_length=np.empty(b, dtype=np.int32))
    r.points, r.done_count, r.game_length_out = _ptr(p), _ptr(counts), _ptr(rec["game_length"])
    if record:
        rec.update(obs=np.empty((steps, b, m, d), dtype=p.dtype),
                   host_class=np.empty((steps, b), dtype=np.int32),
                   axis=np.empty((steps, b), dtype=np.int32),
                   done=np.empty((steps, b), dtype=np.uint8),
                   reward=np.empty((steps, b), dtype=np.float32))
        r.obs_out, r.host_class_out, r.axis_out = _ptr(rec["obs"]), _ptr(rec["host_class"]), _ptr(rec["axis"])
        r.done_out, r.reward_out = _ptr(rec["done"]), _ptr(rec["reward"])
    r.seed, r.game_offset, r.step_offset = seed, game_offset, step_offset
    r.padding_value, r.reward_sign = padding_value, reward_sign
    r.batch, r.max_points, r.dim, r.dtype, r.steps = b, m, d, _hk_dtype(p), steps
    r.host_policy, r.agent_policy, r.stages, r.flags = host_policy, agent_policy, stages, flags
    _check(lib().hko_rollout(C.byref(r)))
    if record:
        rec["done"] = rec["done"].astype(bool)
    return p, rec


def rollout_generated(batch: int, spec, steps: int, seed: int, *, max_value: int, gen_seed=None,
                      gen_stages=A.HK_STAGE_NEWTON | A.HK_STAGE_REPOSITION, episodes=1, game_offset=0, step_offset=0,
                      host_policy=A.HK_HOST_RANDOM, agent_policy=A.HK_AGENT_RANDOM,
                      stages=A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON, flags=0,
                      padding_value=-1.0, dtype=np.float32, game_ids=None):
    """hk_rollout_desc.gen_max_value / episodes (ABI 4): the initial states are drawn per game (the generator stream,
    seed gen_seed + e) and rolled out (seed + e), e = 0 .. episodes - 1; returns (final state of the last episode,
    dict(done_count summed over the episodes, game_length of the last episode))."""
    m, d = spec
    p = np.empty((batch, m, d), dtype=dtype)
    r = A.hk_rollout_desc()
    if game_ids is not None:
        ids = np.ascontiguousarray(game_ids, dtype=np.int32)
        r.game_ids = _ptr(ids)
    counts = np.zeros(steps + 1, dtype=np.uint64)
    rec = dict(done_count=counts, game_length=np.empty(batch, dtype=np.int32))
    r.points, r.done_count, r.game_length_out = _ptr(p), _ptr(counts), _ptr(rec["game_length"])
    r.seed, r.game_offset, r.step_offset = seed, game_offset, step_offset
    r.padding_value, r.reward_sign = padding_value, 1.0
    r.batch, r.max_points, r.dim, r.dtype, r.steps = batch, m, d, _hk_dtype(p), steps
    r.host_policy, r.agent_policy, r.stages, r.flags = host_policy, agent_policy, stages, flags
    r.gen_max_value, r.gen_seed, r.gen_stages, r.episodes = max_value, seed if gen_seed is None else gen_seed, gen_stages, episodes
    _check(lib().hko_rollout(C.byref(r)))
    return p, rec


def bin_by_live_rows(points: np.ndarray, group: int, unit: int):
    """hk_bin_by_live_rows / hk_generate_points_binned's order (include/hironaka_hip.h): the games of every group of
    `group` consecutive games ranked widest first, equal games in their order; rank p of full group k goes to position
    (p // unit) * F * unit + k * unit + p % unit (F full groups: the k-th units of all groups lie together), a partial
    last group is ranked in place.  Returns (re-ordered points, game ids, live rows per position)."""
    p = np.ascontiguousarray(points)
    b, m, d = p.shape
    out = np.empty_like(p)
    ids = np.empty(b, dtype=np.int32)
    npts = np.empty(b, dtype=np.int32)
    _check(lib().hko_bin_by_live_rows(_ptr(p), _ptr(out), _ptr(ids), _ptr(npts), b, m, d, _hk_dtype(p), group, unit))
    return out, ids, npts

