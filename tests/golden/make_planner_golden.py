"""Generates tests/golden/planners.json by importing the reference's planner modules.

Run in the build container only (needs /root/reference, numpy, scipy, matplotlib):
    MPLBACKEND=Agg python tests/golden/make_planner_golden.py
ctrl/Footstep_Planner.py and ctrl/Foot_Trajectory.py import with numpy/scipy/matplotlib alone (their
module-level demos run on import and are harmless under the Agg backend).  ctrl/LIPM.py,
ctrl/Trajectory.py and ctrl/Walk_Planner.py cannot be imported (broken imports), so
they have no fixtures.  Only numbers are written; no reference source travels.
"""
import importlib.util
import json
import os
import sys
from pathlib import Path

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")


def load(name):
    spec = importlib.util.spec_from_file_location(name, REF / "ctrl" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def unicycle_path(v, w, dt, n, x=0.0, y=0.0, theta=0.0):
    path = []
    for _ in range(n):
        x += v * dt * np.cos(theta)
        y += v * dt * np.sin(theta)
        theta += w * dt
        path.append(np.array([x, y]))
    return path


def main():
    fp = load("Footstep_Planner")
    ft = load("Foot_Trajectory")
    out = {"footsteps": [], "support": [], "foot_traj": []}

    # ---- footstep plans
    cases = [
        dict(step_width=0.2, step_length=0.3, v=0.5, w=0.1, dt=0.1, n=100),   # the module's own demo
        dict(step_width=0.2, step_length=0.3, v=0.25, w=0.0, dt=0.1, n=100),  # straight line, half speed
        dict(step_width=0.1, step_length=0.15, v=0.5, w=-0.3, dt=0.05, n=60),
        dict(step_width=0.2, step_length=0.3, v=0.3, w=0.05, dt=0.1, n=10),   # exactly one step, no leftover
        dict(step_width=0.2, step_length=0.3, v=0.5, w=0.1, dt=0.1, n=7),     # short path with leftover
    ]
    for c in cases:
        planner = fp.FootstepPlanner(step_width=c["step_width"], step_length=c["step_length"])
        path = unicycle_path(c["v"], c["w"], c["dt"], c["n"])
        init = [fp.Footstep(position=np.array([0, 0.1]), orientation=np.array([0, 0, 0]), side=0),
                fp.Footstep(position=np.array([0, -0.1]), orientation=np.array([0, 0, 0]), side=1)]
        steps = planner.plan(path, init)
        out["footsteps"].append(dict(
            params=c, path=[p.tolist() for p in path],
            steps=[dict(pos=[float(s.position[0]), float(s.position[1])], yaw=float(s.orientation[2]),
                        side=int(bool(s.side))) for s in steps]))

    # ---- support polygons
    for (w, l) in ((0.1, 0.25), (0.055, 0.13)):
        for contacts in ([(0.0, 0.1, 0.0, 0), (0.0, -0.1, 0.0, 1)], [(0.3, 0.05, 0.4, 0)], [(1.0, -0.2, -0.7, 1)]):
            fs = [fp.Footstep(position=np.array([x, y]), orientation=np.array([0, 0, yaw]), side=s) for x, y, yaw, s in contacts]
            sup = fp.Support(fs, foot_width=w, foot_length=l)
            out["support"].append(dict(foot_width=w, foot_length=l, contacts=[list(c) for c in contacts],
                                       double=bool(sup.is_double_support),
                                       polygon=[np.asarray(p).tolist() for p in sup.get_support_polygon()]))

    # ---- swing-foot trajectories
    tcases = [
        dict(t=[0.0, 1.0], start=[0, 0, 0], target=[1, 1, 0], h=0.2, rise=0.5),
        dict(t=[0.0, 0.5], start=[0, 0.1, 0, 0], target=[0.3, 0.1, 0, 0.2], h=0.2, rise=0.5),   # conf.py:24-28
        dict(t=[0.0, 0.5], start=[0, 0.1, 0, 0], target=[0.3, 0.1, 0, 0.2], h=0.2, rise=0.1),
        dict(t=[1.5, 2.0], start=[0.2, -0.1, 0.01, -0.3], target=[0.55, -0.12, 0.0, 0.1], h=0.05, rise=0.3),
        dict(t=[0.0, 1.0], start=[0, 0, 0], target=[1, 1, 0], h=0.2, rise=0.1),
    ]
    for c in tcases:
        tr = ft.FootTrajectory(c["t"], np.array(c["start"], dtype=float), np.array(c["target"], dtype=float), c["h"], c["rise"])
        ts = np.linspace(c["t"][0], c["t"][1], 21)
        rec = dict(params=c, ts=ts.tolist(),
                   pos=[tr.get_position(t).tolist() for t in ts],
                   vel=[tr.get_velocity(t).tolist() for t in ts],   # 2nd derivatives (quirk F6f)
                   acc=[tr.get_acceleration(t).tolist() for t in ts])  # 3rd derivatives
        if tr.yaw is not None:
            rec["yaw"] = [float(tr.yaw(t)) for t in ts]
        out["foot_traj"].append(rec)

    dst = Path(__file__).parent / "planners.json"
    dst.write_text(json.dumps(out))
    print("wrote", dst, dst.stat().st_size, "bytes;", [len(f["steps"]) for f in out["footsteps"]], "footsteps")


if __name__ == "__main__":
    main()
