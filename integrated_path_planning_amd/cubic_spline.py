"""CubicSpline2D with the reference's interface (src/planning/cubic_spline.py:190-288).

The fit (natural cubic spline over the chord-length parameter) is done by
libfot's native builder and every ``calc_*`` query is evaluated by the gfx950
spline kernel -- the same coefficients and the same device code the planner
uses.  The object exposes ``s``, ``sx`` and ``sy`` (with ``a, b, c, d, x``) like
the reference's, so either class can be handed to ``FrenetPlanner``.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from .planner import BatchPlanner


class CubicSpline2D:
    def __init__(self, x, y, device: int = -1):
        wx = np.asarray(x, dtype=np.float64)
        wy = np.asarray(y, dtype=np.float64)
        self._engine = BatchPlanner(waypoints=(wx, wy), device=device)
        s, ax, bx, cx, dx, ay, by, cy, dy = self._engine.path_coeffs()
        self.s = s.tolist()
        self.ds = np.diff(s)
        self.sx = SimpleNamespace(x=s, y=wx, a=ax, b=bx, c=cx, d=dx, nx=len(s))
        self.sy = SimpleNamespace(x=s, y=wy, a=ay, b=by, c=cy, d=dy, nx=len(s))

    def _eval(self, s):
        scalar = np.ndim(s) == 0
        out = self._engine.spline_eval(s)
        return [o[0] if scalar else o for o in out]

    def calc_position(self, s):
        x, y, _, _, _ = self._eval(s)
        return x, y

    def calc_yaw(self, s):
        return self._eval(s)[2]

    def calc_curvature(self, s):
        return self._eval(s)[3]

    def calc_curvature_rate(self, s):
        return self._eval(s)[4]
