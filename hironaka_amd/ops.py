"""Torch-tensor front end of the HIP operators (``include/hironaka_hip.h``).

Tensors are used for device memory and streams only; every function validates its arguments,
fills a C descriptor, calls the C ABI on the current HIP stream and raises on a non-zero status.
There is no CPU path: a non-CUDA tensor is a TypeError.

Operator names follow the reference's operator layer (hironaka/src/__init__.py:22-46):
``shift``, ``get_newton_polytope``, ``reposition``, ``rescale`` with ``sem`` selecting which
sibling's semantics is reproduced ("jax" = _jax_ops.py, "torch" = _torch_ops.py, "list" =
_list_ops.py on padded arrays).
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _abi as A
from ._lib import check, lib

_TORCH2HK = {torch.float32: A.HK_F32, torch.float64: A.HK_F64, torch.int32: A.HK_I32,
             torch.int64: A.HK_I64, torch.uint8: A.HK_U8, torch.bool: A.HK_U8}
ALL_STAGES = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON | A.HK_STAGE_RESCALE


# kernel-selection flags OR-ed into every step / rollout descriptor (tests: `with ops.forced(...)`)
_forced_flags = 0


@contextlib.contextmanager
def forced(flags: int):
    """Run the enclosed calls with kernel-selection flags (HK_FLAG_FORCE_ONE_LANE / _TEAM / _GENERIC) added to
    every step and rollout launch -- the results must not depend on them."""
    global _forced_flags
    before, _forced_flags = _forced_flags, _forced_flags | flags
    try:
        yield
    finally:
        _forced_flags = before


def make_flags(sem: str = "jax", noop_if_invalid: bool = False, ignore_ended: bool = False,
               compact_sorted: bool = False, force_generic: bool = False, force_team: bool = False) -> int:
    if sem not in A.SEMANTICS:
        raise ValueError(f"sem must be one of {sorted(A.SEMANTICS)}. Got {sem}.")
    f = A.SEMANTICS[sem]
    if noop_if_invalid:
        f |= A.HK_FLAG_AXIS_NOOP_IF_INVALID
    if ignore_ended:
        f |= A.HK_FLAG_IGNORE_ENDED
    if compact_sorted:
        f |= A.HK_FLAG_COMPACT_SORTED
    if force_generic:
        f |= A.HK_FLAG_FORCE_GENERIC
    if force_team:
        f |= A.HK_FLAG_FORCE_TEAM
    return f


def make_stages(shift=False, reposition=False, newton=False, rescale=False) -> int:
    return ((A.HK_STAGE_SHIFT if shift else 0) | (A.HK_STAGE_REPOSITION if reposition else 0)
            | (A.HK_STAGE_NEWTON if newton else 0) | (A.HK_STAGE_RESCALE if rescale else 0))


def _require_device(t: torch.Tensor, name: str) -> None:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor on a HIP device. Got {type(t)}.")
    if not t.is_cuda:
        raise TypeError(f"{name} must live on a HIP device (hironaka_amd has no CPU path). Got {t.device}.")


def _state(points: torch.Tensor, name="points") -> Tuple[torch.Tensor, torch.dtype]:
    """contiguous f32/f64 view of the state + the dtype to hand back"""
    _require_device(points, name)
    orig = points.dtype
    if orig in (torch.float16, torch.bfloat16):
        points = points.float()
    elif orig not in (torch.float32, torch.float64):
        raise TypeError(f"{name} must be a floating tensor. Got {orig}.")
    return points.contiguous(), orig


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _aux(t: Optional[torch.Tensor], like: torch.Tensor, name: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, device=like.device)
    _require_device(t, name)
    if t.device != like.device:
        raise ValueError(f"{name} is on {t.device}, points on {like.device}")
    if t.dtype in (torch.float16, torch.bfloat16):
        t = t.float()
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    if t.dtype not in _TORCH2HK:
        raise TypeError(f"{name}: unsupported dtype {t.dtype}")
    return t.contiguous()


def step(points: torch.Tensor, coords=None, axis=None, *, stages: int, flags: int = 0,
         padding_value: float = -1.0, reward_sign: float = 1.0, spec: Optional[Tuple[int, int]] = None,
         coords_in_record: bool = False, out: Optional[torch.Tensor] = None,
         want: Sequence[str] = (), features_out: Optional[torch.Tensor] = None,
         scale_observation: bool = True) -> Dict[str, torch.Tensor]:
    """One fused transition (hk_step).

    points: [B, m, d] state, or a [B, stride] record matrix with ``spec=(m, d)`` (flattened host
        observation, or agent observation whose last d columns are the subset mask --
        ``coords_in_record=True``; jax/util.py:58-74).
    coords: [B, d] multi-binary mask (any numeric dtype) or [B] class ids (int32/int64).
    axis: [B] int or float; or [B, d] float32 logits of the agent (with class-id coords): its move is the argmax over
        the subset's coordinates, decoded inside the kernel (shapes with a four-lane step kernel only).
    want: any of "done", "prev_done", "reward", "num_points".
    features_out: [B, m*d] contiguous float32 -- the observation features of the RESULT (get_features of the new
        points, rescaled first if `scale_observation`) written by the same launch (hk_step_features: shapes with a
        four-lane step kernel, class-id coords; HironakaHipError(HK_ERR_UNSUPPORTED) otherwise).
    Returns {"points": [B, m, d] (or `out`), ...}."""
    pts, orig = _state(points)
    if pts.dim() == 3:
        b, m, d = pts.shape
        in_stride = m * d
    elif pts.dim() == 2 and spec is not None:
        b, in_stride = pts.shape
        m, d = spec
    else:
        raise ValueError("points must be [B, m, d], or [B, stride] together with spec=(m, d)")
    dev = pts.device
    s = A.hk_step_desc()
    if out is None:
        out_t = torch.empty((b, m, d), dtype=pts.dtype, device=dev)
        out_stride = m * d
    else:
        _require_device(out, "out")
        if out.dtype != pts.dtype or not out.is_contiguous() or out.shape[0] != b:
            raise ValueError("out must be contiguous, of the state dtype and batch")
        out_t = out
        out_stride = out.numel() // max(b, 1)
    keep = [pts, out_t]
    s.points_in, s.points_out = pts.data_ptr(), out_t.data_ptr()
    s.in_stride, s.out_stride = in_stride, out_stride
    s.coords_kind = A.HK_COORDS_NONE
    if stages & A.HK_STAGE_SHIFT:
        ax = _aux(axis, pts, "axis")
        if ax is not None and ax.dim() == 2:
            # [B, d] float32 logits of the agent: its move is the argmax over the subset's coordinates
            # (HK_AXIS_MASKED_LOGITS; four-lane step kernel only -- HironakaHipError(HK_ERR_UNSUPPORTED) otherwise)
            if ax.shape != (b, d) or ax.dtype != torch.float32 or not ax.is_contiguous():
                raise ValueError(f"agent logits must be a contiguous float32 tensor of shape ({b}, {d})")
            keep.append(ax)
            s.axis, s.axis_dtype = ax.data_ptr(), A.HK_AXIS_MASKED_LOGITS
        else:
            if ax is None or ax.shape != (b,):
                raise ValueError(f"axis must have shape ({b},)")
            if ax.dtype == torch.uint8:
                ax = ax.to(torch.int32)
            keep.append(ax)
            s.axis, s.axis_dtype = ax.data_ptr(), _TORCH2HK[ax.dtype]
        if coords_in_record:
            s.coords_kind = A.HK_COORDS_IN_RECORD
        else:
            co = _aux(coords, pts, "coords")
            if co is None:
                raise ValueError("coords is required for the shift stage")
            keep.append(co)
            if co.dim() == 1:
                if co.shape != (b,) or co.dtype not in (torch.int32, torch.int64):
                    raise ValueError("class-id coords must be int32/int64 of shape (B,)")
                s.coords_kind = A.HK_COORDS_CLASS_I32 if co.dtype == torch.int32 else A.HK_COORDS_CLASS_I64
            elif co.shape == (b, d):
                s.coords_kind = _TORCH2HK[co.dtype]
                s.coords_stride = d
            else:
                raise ValueError(f"coords must have shape ({b}, {d}) or ({b},)")
            s.coords = co.data_ptr()
    res: Dict[str, torch.Tensor] = {}
    for key in want:
        if key in ("done", "prev_done"):
            res[key] = torch.empty(b, dtype=torch.bool, device=dev)  # the kernel stores 0/1 bytes
        elif key == "reward":
            res[key] = torch.empty(b, dtype=torch.float32, device=dev)
        elif key == "num_points":
            res[key] = torch.empty(b, dtype=torch.int32, device=dev)
        else:
            raise ValueError(f"unknown output {key}")
    s.done_out = res["done"].data_ptr() if "done" in res else None
    s.prev_done_out = res["prev_done"].data_ptr() if "prev_done" in res else None
    s.reward_out = res["reward"].data_ptr() if "reward" in res else None
    s.num_points_out = res["num_points"].data_ptr() if "num_points" in res else None
    s.padding_value, s.reward_sign = float(padding_value), float(reward_sign)
    s.batch, s.max_points, s.dim, s.dtype = b, m, d, _TORCH2HK[pts.dtype]
    s.stages, s.flags = stages, flags | _forced_flags
    with torch.cuda.device(dev):
        if features_out is not None:
            if (not features_out.is_cuda or features_out.dtype != torch.float32 or not features_out.is_contiguous()
                    or tuple(features_out.shape) != (b, m * d)):
                raise ValueError(f"features_out must be a contiguous float32 [{b}, {m * d}] tensor on the device")
            check(lib().hk_step_features(C.byref(s), features_out.data_ptr(), int(bool(scale_observation)),
                                         _stream(pts)), "hk_step_features")
            res["features"] = features_out
        else:
            check(lib().hk_step(C.byref(s), _stream(pts)), "hk_step")
    if out is None and orig != out_t.dtype:
        out_t = out_t.to(orig)
    res["points"] = out_t
    return res


def shift(points, coords, axis, padding_value: float = -1.0, sem: str = "jax", noop_if_invalid=False,
          ignore_ended=False, **kw) -> torch.Tensor:
    """shift_jax / shift_torch / shift_lst (see include/hironaka_hip.h)."""
    return step(points, coords, axis, stages=A.HK_STAGE_SHIFT, padding_value=padding_value,
                flags=make_flags(sem, noop_if_invalid, ignore_ended, **kw))["points"]


def reposition(points, padding_value: float = -1.0, sem: str = "jax", **kw) -> torch.Tensor:
    return step(points, stages=A.HK_STAGE_REPOSITION, padding_value=padding_value,
                flags=make_flags(sem, **kw))["points"]


def get_newton_polytope(points, padding_value: float = -1.0, sem: str = "jax", compact_sorted=False,
                        **kw) -> torch.Tensor:
    return step(points, stages=A.HK_STAGE_NEWTON, padding_value=padding_value,
                flags=make_flags(sem, compact_sorted=compact_sorted, **kw))["points"]


def rescale(points, padding_value: float = -1.0, sem: str = "jax", **kw) -> torch.Tensor:
    return step(points, stages=A.HK_STAGE_RESCALE, padding_value=padding_value,
                flags=make_flags(sem, **kw))["points"]


def _counts(points: torch.Tensor, spec, fn_name: str, out_dtype) -> torch.Tensor:
    pts, _ = _state(points)
    if pts.dim() == 3:
        b, m, d = pts.shape
        stride = m * d
    elif pts.dim() == 2 and spec is not None:
        b, stride = pts.shape
        m, d = spec
    else:
        raise ValueError("points must be [B, m, d], or [B, stride] together with spec=(m, d)")
    out = torch.empty(b, dtype=out_dtype, device=pts.device)
    with torch.cuda.device(pts.device):
        check(getattr(lib(), fn_name)(pts.data_ptr(), stride, out.data_ptr(), b, m, d,
                                      _TORCH2HK[pts.dtype], _stream(pts)), fn_name)
    return out


def get_dones(points: torch.Tensor, spec=None) -> torch.Tensor:
    """(#rows with x_0 >= 0) < 2 -- jax/util.py:34-35."""
    return _counts(points, spec, "hk_get_dones", torch.bool)  # 0/1 bytes


def get_num_points(points: torch.Tensor, spec=None) -> torch.Tensor:
    """core/tensor_points.py:65-70."""
    return _counts(points, spec, "hk_get_num_points", torch.int32)


def decode_host_class(cls: torch.Tensor, dim: int, dtype=torch.float32) -> torch.Tensor:
    """class ids -> multi-binary masks (jax/host_action_preprocess.py:59-65)."""
    _require_device(cls, "cls")
    ids = cls.to(torch.int32).contiguous()
    n_cls = 2 ** dim - dim - 1
    out = torch.empty((ids.numel(), dim), dtype=dtype, device=ids.device)
    if dtype not in _TORCH2HK:
        raise TypeError(f"unsupported mask dtype {dtype}")
    with torch.cuda.device(ids.device):
        check(lib().hk_decode_host_class(ids.data_ptr(), out.data_ptr(), _TORCH2HK[dtype], ids.numel(), dim,
                                         _stream(ids)), "hk_decode_host_class")
    return out.reshape(*cls.shape, dim)


def zeillinger(points: torch.Tensor, sem: str = "jax", spec=None, force_generic: bool = False,
               force_team: bool = False) -> torch.Tensor:
    """Zeillinger host: class id per game.  sem="jax": jax/players.py:84-109 (degenerate -> 0);
    sem="list": host.py:70-95 on padded rows (-1 for a game with fewer than 2 points).
    points: [B, m, d], or [B, stride] records with spec=(m, d) (e.g. agent observations)."""
    if sem not in ("jax", "list"):
        raise ValueError(f"sem must be 'jax' or 'list'. Got {sem}.")
    pts, _ = _state(points)
    if pts.dim() == 3:
        b, m, d = pts.shape
        stride = m * d
    elif pts.dim() == 2 and spec is not None:
        b, stride = pts.shape
        m, d = spec
    else:
        raise ValueError("points must be [B, m, d], or [B, stride] together with spec=(m, d)")
    out = torch.empty(b, dtype=torch.int32, device=pts.device)
    flags = (A.SEMANTICS[sem] | (A.HK_FLAG_FORCE_GENERIC if force_generic else 0)
             | (A.HK_FLAG_FORCE_TEAM if force_team else 0))
    with torch.cuda.device(pts.device):
        check(lib().hk_zeillinger(pts.data_ptr(), stride, out.data_ptr(), b, m, d, _TORCH2HK[pts.dtype],
                                  flags, _stream(pts)), "hk_zeillinger")
    return out


def get_features(points: torch.Tensor, scale_observation: bool = True, padding_value: float = -1.0,
                 spec=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """order_and_rescale (jax/util.py:186-197): [B, m*d] rows sorted descending, last coordinate
    primary, optionally rescaled first."""
    pts, orig = _state(points)
    if pts.dim() == 3:
        b, m, d = pts.shape
        in_stride = m * d
    elif pts.dim() == 2 and spec is not None:
        b, in_stride = pts.shape
        m, d = spec
    else:
        raise ValueError("points must be [B, m, d], or [B, stride] together with spec=(m, d)")
    if out is None:
        out = torch.empty((b, m * d), dtype=pts.dtype, device=pts.device)
    elif (not out.is_cuda or out.dtype != pts.dtype or tuple(out.shape) != (b, m * d) or not out.is_contiguous()):
        raise ValueError(f"out must be a contiguous [{b}, {m * d}] {pts.dtype} tensor on the device")
    with torch.cuda.device(pts.device):
        check(lib().hk_get_features(pts.data_ptr(), in_stride, out.data_ptr(), m * d, b, m, d,
                                    _TORCH2HK[pts.dtype], int(bool(scale_observation)), float(padding_value),
                                    _stream(pts)), "hk_get_features")
    return out if orig == out.dtype else out.to(orig)


def get_features_torch(points: torch.Tensor, padding_value: float = -1.0) -> torch.Tensor:
    """TensorPoints.get_features (core/tensor_points.py:72-74): [B, m, d] rows ordered by coordinate 0,
    descending (unavailable rows last); rows with equal coordinate 0 keep their order."""
    pts, orig = _state(points)
    if pts.dim() != 3:
        raise ValueError(f"points must be [B, m, d]. Got {tuple(pts.shape)}.")
    b, m, d = pts.shape
    out = torch.empty_like(pts)
    with torch.cuda.device(pts.device):
        check(lib().hk_get_features_torch(pts.data_ptr(), m * d, out.data_ptr(), m * d, b, m, d,
                                          _TORCH2HK[pts.dtype], float(padding_value), _stream(pts)),
              "hk_get_features_torch")
    return out if orig == out.dtype else out.to(orig)


def generate_points(batch: int, max_points: int, dim: int, max_value: int, seed: int, *,
                    game_offset: int = 0, dtype=torch.float32, device=None, newton=True, reposition=True,
                    rescale=False, padding_value: float = -1.0, flags: int = 0,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """generate_pts (jax/util.py:385-392): randint[0, max_value) -> newton -> [reposition] ->
    [rescale], Philox-keyed by (seed, game_offset + game index)."""
    if out is None:
        device = torch.device("cuda") if device is None else torch.device(device)
        if device.type != "cuda":
            raise TypeError("generate_points needs a HIP device")
        out = torch.empty((batch, max_points, dim), dtype=dtype, device=device)
    else:
        _require_device(out, "out")
    stages = make_stages(False, reposition, newton, rescale)
    with torch.cuda.device(out.device):
        check(lib().hk_generate_points(out.data_ptr(), batch, max_points, dim, _TORCH2HK[out.dtype], max_value,
                                       seed, game_offset, stages, float(padding_value), flags, _stream(out)),
              "hk_generate_points")
    return out


_WORKSPACES: Dict[Tuple[torch.device, int], torch.Tensor] = {}
_RETIRED_WORKSPACES: list = []  # outgrown buffers: a hipGraph captured earlier may still hold their address


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Per-(device, stream) memory for hk_rollout's per-workgroup counters, grown on demand (the C ABI
    never allocates).  Zero when created; every reduction leaves it zero again.  A buffer that a larger
    request outgrows is retired, not freed (a captured graph may replay launches that add to it), and growing
    DURING a capture is refused: allocate first (one eager call of the same shape, or the caller's own
    `rollout_workspace` + `defer_counts`)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("hk_rollout's counter workspace would have to be (re)allocated inside a stream "
                               "capture: run the same rollout once eagerly on this stream first, or pass "
                               "workspace=ops.rollout_workspace(...) with defer_counts=True")
        if ws is not None:
            _RETIRED_WORKSPACES.append(ws)
        ws = torch.zeros(max(nbytes, 1 << 16), dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = ws
    return ws


def _geometry_desc(batch: int, steps: int, spec: Tuple[int, int], dtype, flags: int) -> "A.hk_rollout_desc":
    r = A.hk_rollout_desc()
    r.batch, r.max_points, r.dim, r.dtype, r.steps = batch, spec[0], spec[1], _TORCH2HK[dtype], steps
    r.host_policy, r.agent_policy, r.flags = A.HK_HOST_RANDOM, A.HK_AGENT_RANDOM, flags | _forced_flags
    r.stages = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON
    return r


def rollout_workspace(batch: int, steps: int, spec: Tuple[int, int], dtype=torch.float32, device=None,
                      flags: int = 0) -> torch.Tensor:
    """A zeroed counter workspace of the caller's own, for rollouts with `defer_counts=True`."""
    dev = torch.device("cuda") if device is None else torch.device(device)
    r = _geometry_desc(batch, steps, spec, dtype, flags)
    probe = torch.empty(8, dtype=torch.uint8, device=dev)
    r.points = probe.data_ptr()
    need = lib().hk_rollout_workspace_bytes(C.byref(r))
    return torch.zeros(max(int(need), 8), dtype=torch.uint8, device=dev)


def reduce_counts(workspace: torch.Tensor, done_count: torch.Tensor, batch: int, steps: int,
                  spec: Tuple[int, int], dtype=torch.float32, flags: int = 0) -> torch.Tensor:
    """done_count += the partial counts that rollouts with `defer_counts=True` left in `workspace`
    (hk_rollout_reduce_counts); the workspace is zero afterwards."""
    _require_device(workspace, "workspace")
    _require_device(done_count, "done_count")
    if done_count.dtype != torch.int64 or done_count.numel() != steps + 1:
        raise ValueError("done_count must be an int64 device tensor of steps+1 elements")
    r = _geometry_desc(batch, steps, spec, dtype, flags)
    r.done_count = done_count.data_ptr()
    r.workspace, r.workspace_bytes = workspace.data_ptr(), workspace.numel()
    with torch.cuda.device(workspace.device):
        check(lib().hk_rollout_reduce_counts(C.byref(r), _stream(workspace)), "hk_rollout_reduce_counts")
    return done_count


def rollout(points: torch.Tensor, steps: int, seed: int, *, game_offset: int = 0, step_offset: int = 0,
            host_policy: int = A.HK_HOST_RANDOM, agent_policy: int = A.HK_AGENT_RANDOM,
            stages: int = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON, flags: int = 0,
            padding_value: float = -1.0, reward_sign: float = 1.0, record: Sequence[str] = (),
            done_count: Optional[torch.Tensor] = None,
            initial: Optional[torch.Tensor] = None, defer_counts: bool = False,
            workspace: Optional[torch.Tensor] = None,
            game_ids: Optional[torch.Tensor] = None, episodes: int = 1) -> Dict[str, torch.Tensor]:
    """T fused steps with in-kernel policies (hk_rollout); `points` is updated IN PLACE.
    record: any of "obs", "host_class", "axis", "done", "reward", "game_length".
    done_count: optional uint64-as-int64 [steps+1] accumulator (zeroed by the caller).
    initial: optional tensor like `points` holding the starting state; it is left untouched and
        `points` only receives the final state (an episode restart without a device copy).
    defer_counts: leave the finished-game counts as partial sums in `workspace` (from `rollout_workspace`;
        they accumulate over launches) and skip the reduce kernel; `reduce_counts` adds them to a
        done_count later -- one reduction for many rollouts.
    game_ids: optional int32 [B]: the policy stream of the game at position g is keyed by game_offset + game_ids[g]
        (a batch re-ordered by `bin_by_live_rows` rolls out game by game as the original order would).
    episodes: E > 1 (needs `initial`, no per-step records): E episodes back to back, each from `initial` with seed + e;
        the counts accumulate, `points` and game_length are the last episode's (hk_rollout_desc.episodes: one launch
        where the four-lane kernel's waves are all resident, else one launch per episode inside the library)."""
    _require_device(points, "points")
    if points.dtype not in (torch.float32, torch.float64) or not points.is_contiguous() or points.dim() != 3:
        raise ValueError("rollout updates a contiguous [B, m, d] float32/float64 tensor in place")
    b, m, d = points.shape
    dev = points.device
    if initial is not None:
        _require_device(initial, "initial")
        if (initial.shape != points.shape or initial.dtype != points.dtype or not initial.is_contiguous()
                or initial.device != dev):
            raise ValueError("initial must match points in shape, dtype, device and be contiguous")
    r = A.hk_rollout_desc()
    if game_ids is not None:
        _require_device(game_ids, "game_ids")
        if game_ids.dtype != torch.int32 or game_ids.shape != (b,) or not game_ids.is_contiguous() or game_ids.device != dev:
            raise ValueError("game_ids must be a contiguous int32 [B] tensor on the device of points")
        r.game_ids = game_ids.data_ptr()
    res: Dict[str, torch.Tensor] = {}
    if defer_counts:
        if workspace is None:
            raise ValueError("defer_counts=True needs the caller's own workspace (ops.rollout_workspace)")
        _require_device(workspace, "workspace")
        flags |= A.HK_FLAG_DEFER_COUNTS
    elif done_count is None:
        done_count = torch.zeros(steps + 1, dtype=torch.int64, device=dev)
    elif done_count.dtype != torch.int64 or done_count.numel() != steps + 1 or not done_count.is_cuda:
        raise ValueError("done_count must be an int64 device tensor of steps+1 elements")
    if done_count is not None:
        res["done_count"] = done_count
    for key in record:
        if key == "obs":
            res[key] = torch.empty((steps, b, m, d), dtype=points.dtype, device=dev)
        elif key in ("host_class", "axis"):
            res[key] = torch.empty((steps, b), dtype=torch.int32, device=dev)
        elif key == "done":
            res[key] = torch.empty((steps, b), dtype=torch.bool, device=dev)  # 0/1 bytes
        elif key == "reward":
            res[key] = torch.empty((steps, b), dtype=torch.float32, device=dev)
        elif key == "game_length":
            res[key] = torch.empty(b, dtype=torch.int32, device=dev)
        else:
            raise ValueError(f"unknown record {key}")
    ptr = lambda k: res[k].data_ptr() if k in res else None
    r.points = points.data_ptr()
    r.done_count = done_count.data_ptr() if (done_count is not None and not defer_counts) else None
    r.points_in = initial.data_ptr() if initial is not None else None
    r.obs_out, r.host_class_out, r.axis_out = ptr("obs"), ptr("host_class"), ptr("axis")
    r.done_out, r.reward_out, r.game_length_out = ptr("done"), ptr("reward"), ptr("game_length")
    r.seed, r.game_offset, r.step_offset = seed, game_offset, step_offset
    r.padding_value, r.reward_sign = float(padding_value), float(reward_sign)
    r.batch, r.max_points, r.dim, r.dtype, r.steps = b, m, d, _TORCH2HK[points.dtype], steps
    r.host_policy, r.agent_policy, r.stages, r.flags = host_policy, agent_policy, stages, flags | _forced_flags
    r.episodes = int(episodes)
    ws = workspace if defer_counts else _workspace(dev, lib().hk_rollout_workspace_bytes(C.byref(r)))
    r.workspace, r.workspace_bytes = ws.data_ptr(), ws.numel()
    with torch.cuda.device(dev):
        check(lib().hk_rollout(C.byref(r), _stream(points)), "hk_rollout")
    res["points"] = points
    return res


def rollout_generated(batch: int, spec: Tuple[int, int], steps: int, seed: int, *, max_value: int,
                      gen_seed: Optional[int] = None, newton: bool = True, reposition: bool = True, rescale: bool = False,
                      episodes: int = 1, game_offset: int = 0, step_offset: int = 0,
                      host_policy: int = A.HK_HOST_RANDOM, agent_policy: int = A.HK_AGENT_RANDOM,
                      stages: int = A.HK_STAGE_SHIFT | A.HK_STAGE_REPOSITION | A.HK_STAGE_NEWTON, flags: int = 0,
                      padding_value: float = -1.0, dtype=torch.float32, device=None,
                      out: Optional[torch.Tensor] = None, record: Sequence[str] = (),
                      done_count: Optional[torch.Tensor] = None, defer_counts: bool = False,
                      workspace: Optional[torch.Tensor] = None,
                      game_ids: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """hk_rollout from initial states drawn INSIDE the launch (hk_rollout_desc.gen_max_value, ABI 4): what
    `generate_points(batch, m, d, max_value, gen_seed, ...)` followed by `rollout(..., steps, seed)` computes, without
    the state ever touching memory -- the loop body of JAXTrainer.compute_rho (jax_trainer.py:502-555), `episodes` of them
    back to back (episode e: seed + e, gen_seed + e; the counts accumulate).  `out`: optional [B, m, d] tensor for the
    final state of the last episode (None: "counts only").  record: "game_length" (the last episode's).
    Requests the fused kernel does not serve (other shapes / dtypes / flags) run as generate + rollout per episode
    inside the library and then need `out`; without it they raise HironakaHipError(HK_ERR_UNSUPPORTED)."""
    m, d = spec
    dev = out.device if out is not None else (torch.device("cuda") if device is None else torch.device(device))
    if dev.type != "cuda":
        raise TypeError("rollout_generated needs a HIP device")
    if out is not None:
        _require_device(out, "out")
        if out.shape != (batch, m, d) or not out.is_contiguous():
            raise ValueError("out must be a contiguous [B, m, d] tensor")
        dtype = out.dtype
    r = A.hk_rollout_desc()
    res: Dict[str, torch.Tensor] = {}
    if game_ids is not None:
        _require_device(game_ids, "game_ids")
        if game_ids.dtype != torch.int32 or game_ids.shape != (batch,) or not game_ids.is_contiguous():
            raise ValueError("game_ids must be a contiguous int32 [B] tensor")
        r.game_ids = game_ids.data_ptr()
    if defer_counts:
        if workspace is None:
            raise ValueError("defer_counts=True needs the caller's own workspace (ops.rollout_workspace)")
        flags |= A.HK_FLAG_DEFER_COUNTS
    elif done_count is None:
        done_count = torch.zeros(steps + 1, dtype=torch.int64, device=dev)
    elif done_count.dtype != torch.int64 or done_count.numel() != steps + 1 or not done_count.is_cuda:
        raise ValueError("done_count must be an int64 device tensor of steps+1 elements")
    if done_count is not None:
        res["done_count"] = done_count
    for key in record:
        if key != "game_length":
            raise ValueError(f"rollout_generated records game_length only. Got {key}.")
        res[key] = torch.empty(batch, dtype=torch.int32, device=dev)
    r.points = out.data_ptr() if out is not None else None
    r.done_count = done_count.data_ptr() if (done_count is not None and not defer_counts) else None
    r.game_length_out = res["game_length"].data_ptr() if "game_length" in res else None
    r.seed, r.game_offset, r.step_offset = seed, game_offset, step_offset
    r.padding_value, r.reward_sign = float(padding_value), 1.0
    r.batch, r.max_points, r.dim, r.dtype, r.steps = batch, m, d, _TORCH2HK[dtype], steps
    r.host_policy, r.agent_policy, r.stages, r.flags = host_policy, agent_policy, stages, flags | _forced_flags
    r.gen_max_value, r.gen_seed = int(max_value), int(seed if gen_seed is None else gen_seed)
    r.gen_stages, r.episodes = make_stages(False, reposition, newton, rescale), int(episodes)
    with torch.cuda.device(dev):
        ws = workspace if defer_counts else _workspace(dev, lib().hk_rollout_workspace_bytes(C.byref(r)))
        r.workspace, r.workspace_bytes = ws.data_ptr(), ws.numel()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib().hk_rollout(C.byref(r), stream), "hk_rollout")
    if out is not None:
        res["points"] = out
    return res


def bin_group(max_points: int, dim: int, dtype=torch.float32) -> Tuple[int, int]:
    """(games per group, games per unit) of the on-device binning for a shape; (0, 0): no binning kernel."""
    code = _TORCH2HK.get(dtype, -1)
    return int(lib().hk_bin_group_games(max_points, dim, code)), int(lib().hk_bin_unit_games(max_points, dim, code))


def bin_by_live_rows(points: torch.Tensor, out: Optional[torch.Tensor] = None,
                     want_num_points: bool = False):
    """Games re-ordered by their number of live rows and the permutation as int32 `game_ids` (position -> original
    index) for `rollout(..., game_ids=)`: a wave of the rollout kernels then holds games of one size -- its slots per lane
    are the widest game's -- and every game keeps its policy stream, so the re-ordered batch rolls out exactly as the
    original one, game by game (the reference's batches carry no order: jax/util.py:385-392 draws them at random).
    ONE launch (hk_bin_by_live_rows, ABI 4) on the shapes with a four-lane kernel: the order is local to groups of
    `bin_group(...)[0]` consecutive games -- widest first, equal games in their order --, and the k-th sixteen games of
    all groups lie together in the output, the widest stratum first (include/hironaka_hip.h).  Other shapes / dtypes: a
    global stable sort with the tensor library.
    Returns (binned points, game_ids) or, with want_num_points, (binned points, game_ids, live rows per position)."""
    _require_device(points, "points")
    if points.dim() != 3:
        raise ValueError("points must be [B, m, d]")
    b, m, d = points.shape
    if points.is_contiguous() and bin_group(m, d, points.dtype)[0]:
        out = torch.empty_like(points) if out is None else out
        if out.data_ptr() == points.data_ptr():
            raise ValueError("bin_by_live_rows cannot work in place: a group's games leave for every stratum")
        ids = torch.empty(b, dtype=torch.int32, device=points.device)
        npts = torch.empty(b, dtype=torch.int32, device=points.device) if want_num_points else None
        with torch.cuda.device(points.device):
            check(lib().hk_bin_by_live_rows(points.data_ptr(), out.data_ptr(), ids.data_ptr(),
                                            npts.data_ptr() if npts is not None else None, b, m, d,
                                            _TORCH2HK[points.dtype], _stream(points)), "hk_bin_by_live_rows")
        return (out, ids, npts) if want_num_points else (out, ids)
    counts = get_num_points(points)
    order = torch.argsort(counts, descending=True, stable=True)
    binned = points.index_select(0, order).contiguous()
    if out is not None:
        out.copy_(binned)
        binned = out
    ids = order.to(torch.int32)
    return (binned, ids, counts.index_select(0, order)) if want_num_points else (binned, ids)


def generate_points_binned(batch: int, max_points: int, dim: int, max_value: int, seed: int, *, game_offset: int = 0,
                           device=None, newton=True, reposition=True, rescale=False, padding_value: float = -1.0,
                           flags: int = 0, out: Optional[torch.Tensor] = None, want_num_points: bool = False):
    """generate_points + bin_by_live_rows as ONE launch (hk_generate_points_binned): the generator has every game's
    rows in registers when it knows their count.  float32, the shapes with a four-lane kernel (HironakaHipError
    otherwise: generate_points + bin_by_live_rows).  Returns (points, game_ids[, live rows per position])."""
    dev = out.device if out is not None else (torch.device("cuda") if device is None else torch.device(device))
    if out is None:
        out = torch.empty((batch, max_points, dim), dtype=torch.float32, device=dev)
    else:
        _require_device(out, "out")
    ids = torch.empty(batch, dtype=torch.int32, device=dev)
    npts = torch.empty(batch, dtype=torch.int32, device=dev) if want_num_points else None
    stages = make_stages(False, reposition, newton, rescale)
    with torch.cuda.device(dev):
        check(lib().hk_generate_points_binned(out.data_ptr(), ids.data_ptr(), npts.data_ptr() if npts is not None else None,
                                              batch, max_points, dim, _TORCH2HK[out.dtype], max_value, seed, game_offset,
                                              stages, float(padding_value), flags | _forced_flags, _stream(out)),
              "hk_generate_points_binned")
    return (out, ids, npts) if want_num_points else (out, ids)


def has_fast_path(max_points: int, dim: int, dtype=torch.float32) -> bool:
    return bool(lib().hk_has_fast_path(max_points, dim, _TORCH2HK.get(dtype, -1)))
