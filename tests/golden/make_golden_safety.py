#!/usr/bin/env python3
"""Golden vectors for SURVEY 8(f3), compute_safety_metrics_static (build container only).

Runs the REFERENCE function (src/core/data_structures.py:301-388) read-only on the scenes of the reference's own tests
(tests/test_footprint.py:52-102, tests/test_smooth_braking.py:137-163) and on seeded random scenes, with and without a
multi-circle footprint.  Writes tests/golden/safety/cases.npz -- inputs and expected outputs only.
"""
import argparse
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules["loguru"] = lg
    sys.path.insert(0, args.ref)
    from src.core.data_structures import EgoVehicleState, PedestrianState, compute_safety_metrics_static
    from src.core.footprint import EgoFootprint

    scenes = []          # (ego xyyawv, pos, vel, ego_r, ped_r, footprint spec or None)
    fp3 = (4.5, 2.0, 3)
    e0 = (0.0, 0.0, 0.0, 5.0)
    scenes += [(e0, [[3.0, 0.0]], [[-1.0, 0.0]], 1.0, 0.2, None),
               (e0, [[2.2, 0.0]], None, 1.0, 0.2, None),
               (e0, [[2.2, 0.0]], None, 1.0, 0.2, fp3),
               (e0, [[0.0, 1.5]], None, 1.0, 0.2, fp3),
               ((0.0, 0.0, np.pi / 2, 5.0), [[2.2, 0.0]], None, 1.0, 0.2, fp3),
               (e0, np.empty((0, 2)), None, 1.0, 0.2, fp3),
               (e0, np.empty((0, 2)), None, 1.0, 0.2, None)]
    e2 = (0.0, 0.0, 0.0, 2.0)
    scenes += [(e2, [[-1.5, 0.0], [3.0, 0.0]], None, 1.0, 0.2, None),
               (e2, [[-1.5, 0.0], [-3.0, 1.0]], None, 1.0, 0.2, None),
               ((0.0, 0.0, np.pi / 2, 2.0), [[3.0, 0.0], [0.0, 2.0]], None, 1.0, 0.2, None)]
    rng = np.random.default_rng(7)
    for k in range(60):
        P = int(rng.choice([1, 2, 5, 17, 64, 65, 130, 300]))
        ego = (rng.normal(0, 20), rng.normal(0, 20), rng.uniform(-np.pi, np.pi), float(rng.choice([0.0, rng.uniform(0, 12)])))
        spread = float(rng.choice([2.0, 8.0, 30.0]))
        pos = np.array(ego[:2]) + rng.normal(0, spread, (P, 2))
        vel = rng.normal(0, 1.3, (P, 2)) * (rng.random((P, 1)) < 0.85)
        fp = None if k % 3 == 0 else (float(rng.uniform(3.5, 5.2)), float(rng.uniform(1.6, 2.2)), int(rng.integers(1, 7)))
        scenes.append((ego, pos, vel, float(rng.uniform(0.8, 1.6)), float(rng.uniform(0.15, 0.4)), fp))

    out, meta = {}, []
    for i, (ego, pos, vel, er, pr, fp) in enumerate(scenes):
        pos = np.asarray(pos, dtype=float).reshape(-1, 2)
        vel = np.zeros_like(pos) if vel is None else np.asarray(vel, dtype=float).reshape(-1, 2)
        es = EgoVehicleState(x=ego[0], y=ego[1], yaw=ego[2], v=ego[3], a=0.0)
        ps = PedestrianState(positions=pos, velocities=vel, goals=np.zeros_like(pos), timestamp=0.0)
        foot = None if fp is None else EgoFootprint.multi_circle(*fp)
        m = compute_safety_metrics_static(es, ps, er, pr, footprint=foot)
        out[f"c{i}_ego"] = np.array(ego, dtype=float)
        out[f"c{i}_pos"], out[f"c{i}_vel"] = pos, vel
        out[f"c{i}_want"] = np.array([m["min_distance"], float(m["collision"]), m["ttc"], m["clearance"],
                                      m["clearance_ahead"]])
        meta.append({"ego_radius": er, "ped_radius": pr, "footprint": None if fp is None else
                     {"length": fp[0], "width": fp[1], "n": fp[2], "radius": float(foot.radius),
                      "offsets": [float(o) for o in foot.offsets]}})
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "safety", "cases.npz"), **out)
    print(f"wrote {len(scenes)} safety-metric cases")


if __name__ == "__main__":
    main()
