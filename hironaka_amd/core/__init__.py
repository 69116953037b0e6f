from .points_base import PointsBase
from .hip_points import HipPoints

__all__ = ["PointsBase", "HipPoints"]
