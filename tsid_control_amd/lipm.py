"""Linear-inverted-pendulum CoM motion.

The reference sketches this in ctrl/LIPM.py:5-49 with sampled frames from ctrl/Trajectory.py:4-15; neither
module can be imported there (`from Trajectory import ...`; `from ctrl.conf import dt` names an attribute that
does not exist - SURVEY.md F5), so there are no golden vectors and the tests check the recurrence and the
DCM / ZMP identities instead.  Two forms live here:

  * the reference's sampled rollout about a fixed ZMP (`LIPM.make_trajectory`, same public names), kept as
    arrays rather than lists of frames;
  * the closed form of that motion, `x(s) = zmp + d/2 e^{w s} + c e^{-w s}` (`segment` / `eval_segment`),
    which is what WalkSchedule tabulates per step and the k_walk kernel evaluates on the device.
"""
import math

import numpy as np

GRAVITY = 9.80665  # LIPM.py:15


class Trajectory:
    """Samples [K, 3] = (pos, vel, acc) at a fixed period; `get_frame(t, diff)` is the lookup of
    Trajectory.py:4-15 with its bounds checks."""

    def __init__(self, dt):
        self.dt = dt
        self.traj = []          # rows [pos, vel, acc]; a list so that callers can len() / append like the reference

    def get_frame(self, t, diff):
        k = math.floor(t / self.dt)
        if not 0 <= k < len(self.traj):
            raise IndexError("Time index out of bounds")
        row = self.traj[k]
        if not 0 <= diff < len(row):   # the reference indexes traj[t] here (a type error for float t)
            raise IndexError("Difference index out of bounds")
        return row[diff]


def segment(omega, zmp, x0, v0=None, dcm0=None):
    """Coefficients (d, c) of the LIPM motion about a fixed ZMP that starts at x0 with velocity v0 (or with
    divergent component dcm0 = x0 + v0 / omega):  x(s) = zmp + d/2 e^{omega s} + c e^{-omega s}."""
    zmp, x0 = np.asarray(zmp, dtype=np.float64), np.asarray(x0, dtype=np.float64)
    if dcm0 is None:
        dcm0 = x0 + np.asarray(v0, dtype=np.float64) / omega
    d = np.asarray(dcm0, dtype=np.float64) - zmp
    return d, (x0 - zmp) - 0.5 * d


def eval_segment(omega, zmp, d, c, s):
    """(pos, vel, acc) of a segment at time s since its start."""
    ep, em = np.exp(omega * s), np.exp(-omega * s)
    u = 0.5 * d * ep + c * em
    return zmp + u, omega * (0.5 * d * ep - c * em), omega * omega * u


class LIPM:
    def __init__(self, h0, dt=0.002):
        self.w = np.sqrt(GRAVITY / h0)
        self.x, self.y = Trajectory(dt), Trajectory(dt)

    def _xy(self, t, diff):
        return np.array([self.x.get_frame(t, diff), self.y.get_frame(t, diff)])

    def pos(self, t):
        return self._xy(t, 0)

    def vel(self, t):
        return self._xy(t, 1)

    def acc(self, t):
        return self._xy(t, 2)

    def dcm(self, t):
        """Divergent component of motion (LIPM.py:28-29)."""
        return self.pos(t) + self.vel(t) / self.w

    def zmp(self, t):
        """Zero-moment point implied by the samples (LIPM.py:31-32)."""
        return self.pos(t) - self.acc(t) / self.w ** 2

    def make_trajectory(self, t, dt, pos0, vel0, acc0, zmp):
        """The reference's rollout about a fixed ZMP (LIPM.py:34-49), sign and update order as written there:
        acc = (zmp - pos) w^2, then vel += acc dt, then pos += vel dt; one sample per step."""
        state = np.array([pos0, vel0], dtype=np.float64)        # rows pos, vel; columns x, y
        target = np.asarray(zmp, dtype=np.float64)
        w2 = self.w ** 2
        for _ in range(math.floor((t[1] - t[0]) / dt)):
            acc = (target - state[0]) * w2
            state[1] = state[1] + acc * dt
            state[0] = state[0] + state[1] * dt
            for axis, tr in enumerate((self.x, self.y)):
                tr.traj.append([state[0, axis], state[1, axis], acc[axis]])
