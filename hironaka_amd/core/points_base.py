"""The abstract point-container interface a backend plugs into -- the same contract as
``hironaka/core/points_base.py:7-279`` (hook names, class attributes, in-place / copy behaviour),
so that ``Trainer(point_cls=...)`` (trainer/trainer.py:80,114,596) and ``FusedGame`` accept a
HIP-backed container unchanged.

A subclass implements the hooks
    _get_shape, _get_newton_polytope, _shift, _reposition, _rescale, _get_batch_ended
(optionally _add_batch_axis, _points_copy, get_features, _get_max_num_points) and sets the class
attributes ``subcls_config_keys`` / ``running_attributes`` BEFORE calling ``super().__init__``.
``op(inplace=True)`` mutates ``self.points`` and returns ``self``; ``op(inplace=False)`` returns a new
object built as ``self.__class__(new_points, **self.config)`` carrying the running attributes.
"""
import abc
import logging
from copy import deepcopy
from typing import Any, List, Optional, Tuple


class PointsBase(abc.ABC):
    base_config_keys = ["max_num_points"]
    subcls_config_keys: List[str]
    running_attributes: List[str]

    def __init__(self, points: Any, **kwargs):
        self.logger = logging.getLogger(type(self).__name__)
        for key in ("subcls_config_keys", "running_attributes"):
            if not hasattr(self, key):
                raise NotImplementedError(f"{key} must be initialized when subclassing.")
        self.points = points
        shape = self._check_points_shape()
        self.batch_size, _, self.dimension = shape
        self.max_num_points = self._get_max_num_points()
        if "max_num_points" in kwargs:
            if kwargs["max_num_points"] < self.max_num_points:
                self.logger.warning("Specified max_num_points is smaller than the one in input. Ignored.")
            else:
                self.max_num_points = kwargs["max_num_points"]
        self.config = {}
        for key in self.subcls_config_keys + self.base_config_keys:
            if not hasattr(self, key):
                raise Exception("Must initialize keys in 'subcls_config_keys' before calling super().__init__.")
            self.config[key] = getattr(self, key)

    # ---- public operations ---------------------------------------------------------------
    def copy(self, points=None) -> "PointsBase":
        src = self._points_copy(self.points) if points is None else points
        new = self.__class__(src, **self.config)
        for key in self.running_attributes:
            if not hasattr(self, key):
                raise Exception(f"Attribute {key} is not initialized.")
            setattr(new, key, deepcopy(getattr(self, key)))
        return new

    def _apply(self, hook, inplace, *args, **kwargs):
        result = hook(self.points, *args, inplace=inplace, **kwargs)
        return self if inplace else self.copy(points=result)

    def shift(self, coords, axis, inplace=True, **kwargs) -> "PointsBase":
        return self._apply(self._shift, inplace, coords, axis, **kwargs)

    def reposition(self, inplace=True, **kwargs) -> "PointsBase":
        return self._apply(self._reposition, inplace, **kwargs)

    def get_newton_polytope(self, inplace=True, **kwargs) -> "PointsBase":
        return self._apply(self._get_newton_polytope, inplace, **kwargs)

    def rescale(self, inplace=True, **kwargs) -> "PointsBase":
        return self._apply(self._rescale, inplace, **kwargs)

    @property
    def ended(self) -> bool:
        return bool(all(self._get_batch_ended(self.points)))

    @property
    def ended_batch(self) -> Any:
        return self._get_batch_ended(self.points)

    def get_features(self) -> Any:
        return self.points

    def __getitem__(self, item: int):
        return self.points[item]

    # ---- hooks -------------------------------------------------------------------------------
    @staticmethod
    def _points_copy(points):
        return deepcopy(points)

    @abc.abstractmethod
    def _get_shape(self, points: Any) -> Tuple:
        ...

    @abc.abstractmethod
    def _get_newton_polytope(self, points: Any, inplace: Optional[bool] = True, **kwargs):
        ...

    @abc.abstractmethod
    def _shift(self, points: Any, coords, axis, inplace: Optional[bool] = True, **kwargs):
        ...

    @abc.abstractmethod
    def _reposition(self, points: Any, inplace: Optional[bool] = True, **kwargs):
        ...

    @abc.abstractmethod
    def _rescale(self, points: Any, inplace: Optional[bool] = True, **kwargs):
        ...

    @abc.abstractmethod
    def _get_batch_ended(self, points: Any):
        ...

    def _add_batch_axis(self, points: Any):
        raise NotImplementedError

    def _get_max_num_points(self) -> int:
        return max((len(self[b]) for b in range(self.batch_size)), default=0)

    def _check_points_shape(self) -> Tuple[int, int, int]:
        shape = tuple(self._get_shape(self.points))
        if len(shape) == 2:
            try:
                self.points = self._add_batch_axis(self.points)
            except NotImplementedError:
                raise ValueError("Points must be 3-dimensional: batch, max_num_points, coordinates.")
            self.logger.warning("Points are 3-dimensional: batch, max_num_points, coordinates. "
                                "A batch dimension is automatically added.")
            shape = (1, *shape)
        if len(shape) != 3:
            raise ValueError("Input dimension must be 2 or 3.")
        return shape
