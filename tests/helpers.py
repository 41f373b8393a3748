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




# The evaluation kernels a plan call can take (csrc/fot_kernels.hip launch_evaluate).  Every parity test that asserts a
# per-candidate table runs under each of them: "auto" is what a caller gets (k_evaluate_split for a handful of egos,
# k_evaluate_group for batches); "group" / "wave" walk every candidate in one piece under the grouped / per-wave cut
# (k_evaluate_group / k_evaluate); "split-wave" cuts the per-wave tiles into time segments.
EVAL_PATHS = ("auto", "group", "wave", "split-wave")


def set_eval_path(bp, path):
    cut, seg = {"auto": (0, 0), "group": (2, 1), "wave": (1, 1), "split-wave": (1, 4)}[path]
    bp.set_tile_cut(cut)
    bp.set_eval_segments(seg)
