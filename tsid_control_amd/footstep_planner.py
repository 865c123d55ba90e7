"""2-D footstep placement along a polyline path - host side of config 3 (once per episode).

Same public names and call signatures as the reference's ctrl/Footstep_Planner.py:4-125 (Footstep,
Support, FootstepPlanner.add_step / .plan) without its module-level demo (:127-186).  Results are
pinned against the reference by tests/golden/planners.json.
"""
from typing import List

import numpy as np


class Footstep:
    """A planar foot placement: position [x, y], orientation [.., .., yaw], side 0 = left, 1 = right
    (Footstep_Planner.py:4-38)."""

    def __init__(self, position, orientation, side):
        self.position = np.array(position, dtype=np.float64)
        self.orientation = np.array(orientation, dtype=np.float64)
        self.side = side
        self.frame = np.eye(3)
        self.update_frame()

    def rotation_matrix(self, orientation):
        c, s = np.cos(orientation[2]), np.sin(orientation[2])
        return np.array([[c, -s], [s, c]])

    def update_frame(self):
        self.frame[:2, :2] = self.rotation_matrix(self.orientation)
        self.frame[:2, 2] = self.position

    def transform(self, point):
        """Foot-frame point -> world (Footstep_Planner.py:29-35)."""
        return self.frame[:2, :2] @ np.asarray(point, dtype=np.float64) + self.frame[:2, 2]

    def __repr__(self):
        return f"Footstep(position={self.position}, orientation={self.orientation})"


class Support:
    """Single/double support and its polygon (Footstep_Planner.py:40-66)."""

    def __init__(self, contacts: List[Footstep], foot_width: float, foot_length: float, start_time: float = 0.0):
        self.contacts = contacts
        self.is_double_support = len(contacts) == 2
        self.foot_width = foot_width
        self.foot_length = foot_length
        self.start_time = start_time

    def get_support_polygon(self):
        hl, hw = self.foot_length / 2, self.foot_width / 2
        # corner order per side as the reference lists them (:55-64)
        left = [(-hl, hw), (-hl, -hw), (hl, -hw), (hl, hw)]
        right = [(hl, -hw), (hl, hw), (-hl, hw), (-hl, -hw)]
        poly = []
        for c in self.contacts:
            poly.extend(c.transform(p) for p in (left if c.side == 0 else right))
        return poly


class FootstepPlanner:
    """Emit a step every `step_length` of accumulated path length, offset half a step width to the
    stepping side, yaw along the local path tangent (Footstep_Planner.py:69-125)."""

    def __init__(self, step_width, step_length):
        self.step_width = step_width
        self.step_length = step_length

    def add_step(self, dx, dy, side, pos) -> Footstep:
        tangent = np.array([dx, dy], dtype=np.float64)
        tangent = tangent / np.linalg.norm(tangent)
        normal = np.array([-tangent[1], tangent[0]])
        sign = 1.0 if side == 0 else -1.0  # the reference compares with == 0, so False counts as left
        position = np.asarray(pos, dtype=np.float64) + tangent * (self.step_length / 2) + normal * (self.step_width / 2 * sign)
        return Footstep(position=position, orientation=np.array([0.0, 0.0, np.arctan2(dy, dx)]), side=side)

    def plan(self, path, init_supports: List[Footstep]) -> List[Footstep]:
        steps = list(init_supports)
        side = init_supports[-1].side
        travelled = 0.0
        dx = dy = 0.0
        for i in range(len(path) - 1):
            dx, dy = np.asarray(path[i + 1]) - np.asarray(path[i])
            travelled += float(np.hypot(dx, dy))
            if travelled >= self.step_length:
                side = not side
                steps.append(self.add_step(dx, dy, side, path[i]))
                travelled = 0.0
        # closing step at the end of the path, plus one more if the last stretch was partial (:114-123)
        side = not side
        steps.append(self.add_step(dx, dy, side, path[-1]))
        if travelled > 0:
            side = not side
            steps.append(self.add_step(dx, dy, side, path[-1]))
        return steps


def unicycle_path(v=0.5, w=0.1, dt=0.1, n=100, scale=1.0):
    """The demo path of Footstep_Planner.py:131-141 (forward-Euler unicycle), optionally scaled."""
    x = y = th = 0.0
    pts = []
    for _ in range(n):
        x += v * dt * np.cos(th)
        y += v * dt * np.sin(th)
        th += w * dt
        pts.append(np.array([x, y]) * scale)
    return pts


def resample_path(path, ds):
    """Polyline with vertices at most `ds` apart (linear interpolation).  FootstepPlanner.plan emits a
    step at the first vertex past each step_length of travel, so the vertex spacing quantises the stride;
    the demo path's 5 cm spacing is as long as an OP3 step."""
    pts = [np.asarray(path[0], dtype=np.float64)]
    for a, b in zip(path[:-1], path[1:]):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        m = max(1, int(np.ceil(np.linalg.norm(b - a) / ds)))
        pts.extend(a + (b - a) * (i / m) for i in range(1, m + 1))
    return pts
