"""Swing-foot trajectory - counterpart of the reference's ctrl/Foot_Trajectory.py:5-43.

The reference builds scipy CubicSpline objects: two knots for x / y / yaw (a straight line), three
knots for z when rise_ratio == 0.5 (the not-a-knot spline through three points is one parabola) and
four knots otherwise (one cubic).  Here each coordinate is a single polynomial of degree <= 3 in
(t - t0), stored as coefficients so that a whole batch of swing phases evaluates as one tensor
expression on the device.  Pinned against the reference by tests/golden/planners.json.
"""
from typing import List

import numpy as np


def _poly_through(ts, ys):
    """Coefficients c0..c3 (ascending, in t - ts[0]) of the lowest-degree polynomial through the knots."""
    ts = np.asarray(ts, dtype=np.float64)
    x = ts - ts[0]
    V = np.vander(x, len(x), increasing=True)
    c = np.linalg.solve(V, np.asarray(ys, dtype=np.float64))
    out = np.zeros(4)
    out[:len(c)] = c
    return out


class FootTrajectory:
    def __init__(self, t: List, start, target, step_height: float, rise_ratio: float = 0.5, reference_quirks=True):
        self.t = t
        self.reference_quirks = reference_quirks
        start, target = np.asarray(start, dtype=np.float64), np.asarray(target, dtype=np.float64)
        t0, t1 = float(t[0]), float(t[1])
        duration = t1 - t0
        self.t0 = t0
        self.cx = _poly_through([t0, t1], [start[0], target[0]])
        self.cy = _poly_through([t0, t1], [start[1], target[1]])
        self.cyaw = _poly_through([t0, t1], [start[3], target[3]]) if len(start) > 3 else None
        if rise_ratio != 0.5:  # Foot_Trajectory.py:14-17
            rise = duration * rise_ratio
            self.cz = _poly_through([t0, t0 + rise, t1 - rise, t1],
                                    [start[2], start[2] + step_height, target[2] + step_height, target[2]])
        else:                  # Foot_Trajectory.py:19
            self.cz = _poly_through([t0, t0 + duration * rise_ratio, t1], [start[2], start[2] + step_height, target[2]])

    def coefficients(self):
        """[4 (x, y, z, yaw), 4] ascending coefficients in (t - t0); yaw row is zero when absent."""
        return np.stack([self.cx, self.cy, self.cz, self.cyaw if self.cyaw is not None else np.zeros(4)])

    @staticmethod
    def _eval(c, s, deriv=0):
        if deriv == 0:
            return c[0] + s * (c[1] + s * (c[2] + s * c[3]))
        if deriv == 1:
            return c[1] + s * (2 * c[2] + s * 3 * c[3])
        if deriv == 2:
            return 2 * c[2] + 6 * c[3] * s
        if deriv == 3:
            return 6 * c[3] + 0 * s
        return 0 * s

    def _vec(self, t, deriv):
        s = np.asarray(t, dtype=np.float64) - self.t0
        return np.array([self._eval(self.cx, s, deriv), self._eval(self.cy, s, deriv), self._eval(self.cz, s, deriv)])

    def yaw(self, t, deriv=0):
        return self._eval(self.cyaw, np.asarray(t, dtype=np.float64) - self.t0, deriv)

    def get_position(self, t):
        return self._vec(t, 0)

    def get_velocity(self, t):
        """Foot_Trajectory.py:29-35 evaluates the SECOND derivative here; with
        reference_quirks=False this is the first derivative."""
        return self._vec(t, 2 if self.reference_quirks else 1)

    def get_acceleration(self, t):
        """Foot_Trajectory.py:37-43 evaluates the THIRD derivative (F6f); otherwise the second."""
        return self._vec(t, 3 if self.reference_quirks else 2)
