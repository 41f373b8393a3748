"""Shared helpers of the parity tests (oracle = checker only)."""
import numpy as np

from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.batch import PlanRequest, request_from_instance  # noqa: F401
from oracle.check import (NORTH_STAR_TOL, TIGHT, assert_record_matches_oracle,  # noqa: F401
                          oracle_plan_for_request, wrap_angle)


def request_from_golden(g) -> PlanRequest:
    m = g.meta
    e = m["ego"]
    return PlanRequest(x=e[0], y=e[1], yaw=e[2], v=e[3], a=e[4], target_speed=m["target_speed"],
                       last_kappa=m["last_kappa"], prev_s=m["prev_s"], overrides=m["overrides"],
                       max_stop_distance=m["max_stop"], static=g.static, dyn=g.dyn, dist=g.dist)


