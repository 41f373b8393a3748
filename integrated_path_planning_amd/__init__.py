"""MI355X-native Frenet optimal-trajectory planner.

A from-scratch gfx950 implementation of one hot path of
mnhrk15/integrated_path_planning -- ``FrenetPlanner.plan()`` -- behind the C ABI
of ``include/fot.h`` (``libfot.so``), with the reference's Python call surface
kept on top (``FrenetPlanner``) and a batched, multi-GPU entry (``BatchPlanner``,
``distributed``).  Importing the package is cheap; the first planner object
loads ``libfot.so`` and fails loudly if it is missing -- there is no CPU path.
"""
from .batch import PackedBatch, PlanRequest
from .data_structures import EgoVehicleState, FrenetPath, FrenetState
from .footprint import EgoFootprint
from .planner import BatchPlanner, BatchResult, FrenetPlanner

__all__ = ["BatchPlanner", "BatchResult", "FrenetPlanner", "PackedBatch", "PlanRequest", "EgoVehicleState",
           "FrenetPath", "FrenetState", "EgoFootprint", "CubicSpline2D"]


def __getattr__(name):
    if name == "CubicSpline2D":
        from .cubic_spline import CubicSpline2D
        return CubicSpline2D
    raise AttributeError(name)
