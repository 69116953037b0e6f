"""``HipPoints`` -- a ``PointsBase`` container whose operators are the HIP kernels: the drop-in
for ``TensorPoints`` (hironaka/core/tensor_points.py:11-126).  Same constructor arguments, same
attributes the trainers read (``points, batch_size, dimension, max_num_points, dtype, device,
ended, ended_batch_in_tensor, get_num_points(), get_features(), type()``), same semantics as the
torch sibling: an axis outside the host's subset and (by default) a finished game are not shifted
(_torch_ops.py:90-93).

    FusedGame.agent_move (trainer/fused_game.py:150-163):
        points.shift(host_moves, actions); points.get_newton_polytope(); points.rescale()
    costs three launches here; ``points.step(host_moves, actions, rescale=True)`` fuses them.

``semantics="list"`` turns the container into the padded form of ``ListPoints``
(hironaka/core/list_points.py:20-135): the Newton polytope comes out sorted (descending, coordinate 0
primary) and compacted to the front (_list_ops.py:25-41), which is what ``GameHironaka`` and the gym
environments step.
"""
from typing import List, Optional, Type, Union

import numpy as np
import torch

from .. import _abi as A
from .. import ops
from .points_base import PointsBase


def _pad_ragged(points, new_length: int, constant_value: float) -> np.ndarray:
    """nested lists of ragged games -> [B, new_length, d] (src/_fn.py:69-89)"""
    d = len(points[0][0])
    out = np.full((len(points), new_length, d), constant_value, dtype=np.float64)
    for b, rows in enumerate(points):
        if len(rows):
            out[b, : len(rows)] = np.asarray(rows, dtype=np.float64)
    return out


class HipPoints(PointsBase):
    subcls_config_keys = ["value_threshold", "device", "padding_value", "dtype", "semantics"]
    running_attributes = ["distinguished_points"]

    def __init__(self, points: Union[torch.Tensor, List[List[List[float]]], np.ndarray],
                 value_threshold: Optional[float] = 1e8, device: Optional[Union[str, torch.device]] = "cuda",
                 padding_value: Optional[float] = -1.0, distinguished_points: Optional[List[int]] = None,
                 dtype: Optional[Union[Type, torch.dtype]] = torch.float32, semantics: str = "torch", **kwargs):
        if semantics not in ("torch", "list"):
            raise ValueError(f"semantics must be 'torch' or 'list'. Got {semantics}.")
        self.semantics = semantics
        assert padding_value <= 0.0, f"'padding_value' must be a non-positive number. Got {padding_value} instead."
        self.value_threshold = value_threshold
        self.dtype = dtype
        self.device = torch.device(device) if isinstance(device, str) else device
        if self.device.type != "cuda":
            raise TypeError(f"HipPoints lives on a HIP device (there is no CPU path). Got {self.device}.")
        if isinstance(points, list):
            points = torch.tensor(_pad_ragged(points, kwargs["max_num_points"], padding_value),
                                  device=self.device, dtype=self.dtype)
        elif isinstance(points, np.ndarray):
            points = torch.tensor(points, device=self.device, dtype=self.dtype)
        elif isinstance(points, torch.Tensor):
            points = points.type(self.dtype).to(self.device)
        else:
            raise Exception(f"Input must be a Tensor, a numpy array or a nested list. Got {type(points)}.")
        self.padding_value = padding_value
        self.distinguished_points = distinguished_points
        super().__init__(points.contiguous(), **kwargs)

    # ---- what the trainers read -----------------------------------------------------------
    def exceed_threshold(self) -> bool:
        """ListPoints / TensorPoints.exceed_threshold (core/tensor_points.py:57-63): a python bool, i.e. one host
        synchronisation -- loops that must stay on the device use `exceed_threshold_tensor`"""
        if self.value_threshold is not None:
            return bool(self.exceed_threshold_tensor())
        return False

    def exceed_threshold_tensor(self) -> torch.Tensor:
        """0-d bool tensor on the device (no synchronisation): any coordinate >= value_threshold"""
        if self.value_threshold is None:
            return torch.zeros((), dtype=torch.bool, device=self.points.device)
        return torch.max(self.points) >= self.value_threshold

    def get_num_points(self) -> torch.Tensor:
        return ops.get_num_points(self.points).to(torch.int64)

    def get_features(self) -> torch.Tensor:
        """rows sorted by coordinate 0, descending (tensor_points.py:72-74)"""
        return ops.get_features_torch(self.points, self.padding_value)

    def type(self, t: Union[Type, torch.dtype]):
        self.dtype = t
        self.points = self.points.type(t)

    @property
    def ended_batch_in_tensor(self) -> torch.Tensor:
        return ops.get_dones(self.points)

    @property
    def ended(self) -> bool:
        # one reduction on the device instead of a python `all` over B elements
        return bool(ops.get_dones(self.points).all())

    # ---- fused move (not in the reference interface) ------------------------------------------
    def step(self, coords, axis, reposition: bool = False, rescale: bool = False,
             ignore_ended_games: bool = True, want=()):
        """shift -> [reposition] -> newton -> [rescale] in ONE launch, in place."""
        coords, axis = self._actions(coords, axis)
        flags = ops.make_flags(self.semantics, noop_if_invalid=True, ignore_ended=ignore_ended_games)
        stages = ops.make_stages(True, reposition, True, rescale)
        work, back = self._work()
        res = ops.step(work, coords, axis, stages=stages, flags=flags, padding_value=self.padding_value,
                       out=work, want=want)
        self._commit(work, back)
        return res

    # ---- PointsBase hooks ----------------------------------------------------------------------
    def _work(self):
        """f32/f64 run natively; other float dtypes compute in f32 and are cast back"""
        if self.points.dtype in (torch.float32, torch.float64):
            return self.points, False
        return self.points.float(), True

    def _commit(self, work, back):
        if back:
            self.points = work.to(self.dtype)

    def _actions(self, coords, axis):
        if isinstance(coords, list):
            mask = np.zeros((len(coords), self.dimension), dtype=np.float32)
            for b, chosen in enumerate(coords):
                mask[b, list(chosen)] = 1
            coords = torch.tensor(mask, device=self.device)
        elif not isinstance(coords, torch.Tensor):
            raise Exception(f"unsupported input type for coord. Got {type(coords)}.")
        if isinstance(axis, list):
            axis = torch.tensor(axis, device=self.device)
        elif not isinstance(axis, torch.Tensor):
            raise Exception(f"unsupported input type for axis. Got {type(axis)},")
        assert coords.shape == (self.batch_size, self.dimension)
        assert axis.shape == (self.batch_size,)
        return coords.to(self.device), axis.to(self.device)

    def _op(self, points, inplace, stages, coords=None, axis=None, flags=None):
        flags = ops.make_flags(self.semantics) if flags is None else flags
        if inplace and points is self.points:
            work, back = self._work()
            ops.step(work, coords, axis, stages=stages, flags=flags, padding_value=self.padding_value, out=work)
            self._commit(work, back)
            return None
        return ops.step(points, coords, axis, stages=stages, flags=flags, padding_value=self.padding_value)["points"]

    def _shift(self, points, coords, axis, inplace: Optional[bool] = True, ignore_ended_games: Optional[bool] = True,
               **kwargs):
        coords, axis = self._actions(coords, axis)
        flags = ops.make_flags(self.semantics, noop_if_invalid=True, ignore_ended=ignore_ended_games)
        return self._op(points, inplace, A.HK_STAGE_SHIFT, coords, axis, flags)

    def _get_newton_polytope(self, points, inplace: Optional[bool] = True, **kwargs):
        return self._op(points, inplace, A.HK_STAGE_NEWTON)

    def _reposition(self, points, inplace: Optional[bool] = True, **kwargs):
        return self._op(points, inplace, A.HK_STAGE_REPOSITION)

    def _rescale(self, points, inplace: Optional[bool] = True, **kwargs):
        return self._op(points, inplace, A.HK_STAGE_RESCALE)

    def _get_shape(self, points: torch.Tensor):
        return points.shape

    @staticmethod
    def _points_copy(points: torch.Tensor) -> torch.Tensor:
        return points.clone().detach()

    def _add_batch_axis(self, points: torch.Tensor) -> torch.Tensor:
        return points.unsqueeze(0)

    def _get_batch_ended(self, points: torch.Tensor) -> torch.Tensor:
        return ops.get_dones(points)

    def _get_max_num_points(self) -> int:
        return int(self.points.shape[1])

    def __repr__(self) -> str:
        return str(self.points)

    def __hash__(self) -> int:
        return hash(self.points.detach().cpu().numpy().round(8).tobytes())
