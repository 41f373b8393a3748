"""The production launch at scale against the oracle, every instance: whole 256-instance batches of config 3 seeds (the
launch bench.py times: k_evaluate_group at full occupancy, the last-arriver selection under real concurrency), ALL
records and ALL per-candidate tables -- status, kept length, cost of 573 440 candidates per batch -- against the oracle
(which runs on a thread pool: the C call releases the GIL).  Default: one batch of fresh seeds, under the grouped and
the per-wave cut; FOT_FULLSIZE_BATCHES=N sweeps N batches (profiles/r03_*_fullsize_sweep.log)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import TIGHT, assert_record_matches_oracle, oracle_plan_for_request, request_from_instance, set_eval_path
from integrated_path_planning_amd import synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.planner import BatchPlanner
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

N_BATCHES = int(os.environ.get("FOT_FULLSIZE_BATCHES", "1"))
SEED0 = int(os.environ.get("FOT_FULLSIZE_SEED0", "100000"))
WP = (syn.STRAIGHT_WX, syn.STRAIGHT_WY)


def _rounded(r, dt):
    c = lambda a: None if a is None else np.asarray(a).astype(dt).astype(np.float64)
    return PlanRequest(x=r.x, y=r.y, yaw=r.yaw, v=r.v, a=r.a, target_speed=r.target_speed, last_kappa=r.last_kappa,
                       prev_s=r.prev_s, overrides=r.overrides, max_stop_distance=r.max_stop_distance,
                       static=c(r.static), dyn=c(r.dyn), dist=c(r.dist))


@pytest.mark.parametrize("batch", range(N_BATCHES))
def test_every_instance_of_a_full_launch(batch):
    kw = syn.CONFIG3_PLANNER
    params, sp = orc.make_params(**kw), orc.Spline(*WP)
    bp = BatchPlanner(waypoints=WP, **kw)
    reqs = [request_from_instance(syn.config3_instance(SEED0 + 256 * batch + s)) for s in range(256)]
    pb = PackedBatch(reqs, np.float32)
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        wants = list(ex.map(lambda r: oracle_plan_for_request(orc, params, sp, _rounded(r, np.float32), table=True), reqs))
    n_cand = 0
    for path in ("group", "wave"):
        set_eval_path(bp, path)
        res = bp.plan_packed(pb)
        for i, want in enumerate(wants):
            label = f"batch {batch} inst {i} [{path}]"
            assert_record_matches_oracle(res.records[i], want, label=label)
            cost, status, keep, nt = bp.candidates(i)
            np.testing.assert_array_equal(status, want.cand_status, err_msg=label + " status table")
            np.testing.assert_array_equal(keep, want.cand_keep, err_msg=label)
            np.testing.assert_allclose(cost, want.cand_cost, rtol=TIGHT, atol=TIGHT, err_msg=label)
            n_cand += len(status)
    print(f"batch {batch}: 256 instances x 2 kernels, {n_cand} candidate rows equal to the oracle's; "
          f"{sum(w.status == 0 for w in wants)} instances with a path")
    bp.close()
