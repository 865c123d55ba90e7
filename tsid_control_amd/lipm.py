"""Linear-inverted-pendulum CoM rollout - restated from the text of the reference's ctrl/LIPM.py:5-49
and ctrl/Trajectory.py:4-15 (neither module can be imported: `from Trajectory import ...`, and
`from ctrl.conf import dt` names a module attribute that does not exist).  No golden
fixtures exist for them; tests check the recurrences and the dcm/zmp identities instead.
"""
import math

import numpy as np


class Trajectory:
    """Sampled [pos, vel, acc] frames at a fixed period (Trajectory.py:4-15)."""

    def __init__(self, dt):
        self.dt = dt
        self.traj = []

    def get_frame(self, t, diff):
        k = math.floor(t / self.dt)
        if k < 0 or k >= len(self.traj):
            raise IndexError("Time index out of bounds")
        if diff < 0 or diff >= len(self.traj[k]):  # the reference indexes traj[t] here (a type error)
            raise IndexError("Difference index out of bounds")
        return self.traj[k][diff]


class LIPM:
    def __init__(self, h0, dt=0.002):
        self.w = np.sqrt(9.80665 / h0)  # LIPM.py:15
        self.x = Trajectory(dt)
        self.y = Trajectory(dt)

    def pos(self, t):
        return np.array([self.x.get_frame(t, 0), self.y.get_frame(t, 0)])

    def vel(self, t):
        return np.array([self.x.get_frame(t, 1), self.y.get_frame(t, 1)])

    def acc(self, t):
        return np.array([self.x.get_frame(t, 2), self.y.get_frame(t, 2)])

    def dcm(self, t):
        return self.pos(t) + self.vel(t) / self.w  # LIPM.py:28-29

    def zmp(self, t):
        return self.pos(t) - self.acc(t) / self.w ** 2  # LIPM.py:31-32

    def make_trajectory(self, t, dt, pos0, vel0, acc0, zmp):
        """Symplectic-Euler rollout about a fixed ZMP (LIPM.py:34-49): acc = (zmp - pos) w^2 evaluated
        as written (sign as in the reference), vel += acc dt, pos += vel dt."""
        duration = t[1] - t[0]
        pos = np.array(pos0, dtype=np.float64)
        vel = np.array(vel0, dtype=np.float64)
        zmp = np.asarray(zmp, dtype=np.float64)
        for _ in range(math.floor(duration / dt)):
            acc = (zmp - pos) * self.w ** 2
            vel = vel + acc * dt
            pos = pos + vel * dt
            self.x.traj.append([pos[0], vel[0], acc[0]])
            self.y.traj.append([pos[1], vel[1], acc[1]])
