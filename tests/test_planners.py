"""Planners vs golden vectors captured from the reference's importable modules
(tests/golden/make_planner_golden.py -> tests/golden/planners.json): ctrl/Footstep_Planner.py and
ctrl/Foot_Trajectory.py are PINNED.  LIPM / Trajectory / WalkPlanner cannot be imported in the
reference: recurrence/identity checks only (parity unpinned)."""
import json
from pathlib import Path

import numpy as np
import pytest

from tsid_control_amd.foot_trajectory import FootTrajectory
from tsid_control_amd.footstep_planner import Footstep, FootstepPlanner, Support
from tsid_control_amd.lipm import LIPM

GOLD = json.loads((Path(__file__).parent / "golden" / "planners.json").read_text())


@pytest.mark.parametrize("case", GOLD["footsteps"], ids=lambda c: f"n{c['params']['n']}_w{c['params']['w']}")
def test_footstep_plan_matches_reference(case):
    p = case["params"]
    planner = FootstepPlanner(step_width=p["step_width"], step_length=p["step_length"])
    init = [Footstep(np.array([0, 0.1]), np.array([0, 0, 0]), 0), Footstep(np.array([0, -0.1]), np.array([0, 0, 0]), 1)]
    steps = planner.plan([np.array(x) for x in case["path"]], init)
    assert len(steps) == len(case["steps"])
    for s, g in zip(steps, case["steps"]):
        assert int(bool(s.side)) == g["side"]
        assert np.allclose(s.position, g["pos"], rtol=0, atol=1e-12)
        assert abs(s.orientation[2] - g["yaw"]) < 1e-12


def test_survey_known_footsteps():
    demo = GOLD["footsteps"][0]["steps"]   # SURVEY.md 8c quotes these
    assert len(demo) == 19
    assert np.allclose(demo[2]["pos"], [0.443596201, 0.11631278], atol=1e-8) and abs(demo[2]["yaw"] - 0.06) < 1e-12
    assert np.allclose(demo[-1]["pos"], [4.217513186, 2.457705425], atol=1e-8)


@pytest.mark.parametrize("case", GOLD["support"], ids=lambda c: f"{len(c['contacts'])}c_{c['foot_width']}")
def test_support_polygon_matches_reference(case):
    fs = [Footstep(np.array([x, y]), np.array([0, 0, yaw]), s) for x, y, yaw, s in case["contacts"]]
    sup = Support(fs, foot_width=case["foot_width"], foot_length=case["foot_length"])
    assert sup.is_double_support == case["double"]
    assert np.allclose(np.array(sup.get_support_polygon()), np.array(case["polygon"]), atol=1e-13)


@pytest.mark.parametrize("case", GOLD["foot_traj"], ids=lambda c: f"rise{c['params']['rise']}_h{c['params']['h']}")
def test_foot_trajectory_matches_reference(case):
    p = case["params"]
    tr = FootTrajectory(p["t"], np.array(p["start"], float), np.array(p["target"], float), p["h"], p["rise"])
    for i, t in enumerate(case["ts"]):
        assert np.allclose(tr.get_position(t), case["pos"][i], atol=1e-11)
        assert np.allclose(tr.get_velocity(t), case["vel"][i], atol=1e-8)       # 2nd derivative (quirk F6f)
        assert np.allclose(tr.get_acceleration(t), case["acc"][i], atol=1e-7)   # 3rd derivative
        if "yaw" in case:
            assert abs(tr.yaw(t) - case["yaw"][i]) < 1e-12


def test_foot_trajectory_survey_values_and_true_derivatives():
    tr = FootTrajectory([0, 1], np.array([0., 0, 0]), np.array([1., 1, 0]), 0.2)
    assert np.allclose(tr.get_position(0.25), [0.25, 0.25, 0.15]) and np.allclose(tr.get_position(0.9), [0.9, 0.9, 0.072])
    assert np.allclose(tr.get_velocity(0.3), [0, 0, -1.6])
    true = FootTrajectory([0, 1], np.array([0., 0, 0]), np.array([1., 1, 0]), 0.2, reference_quirks=False)
    eps = 1e-6
    fd = (true.get_position(0.3 + eps) - true.get_position(0.3 - eps)) / (2 * eps)
    assert np.allclose(true.get_velocity(0.3), fd, atol=1e-8)
    assert tr.coefficients().shape == (4, 4)


def test_lipm_recurrence_and_identities():
    lipm = LIPM(h0=0.24, dt=0.002)
    assert abs(lipm.w - np.sqrt(9.80665 / 0.24)) < 1e-15
    pos0, vel0, zmp = np.array([0.0, 0.0]), np.array([0.1, 0.0]), np.array([0.02, 0.05])
    lipm.make_trajectory([0.0, 0.1], 0.002, pos0, vel0, np.zeros(2), zmp)
    assert len(lipm.x.traj) == 50
    p, v = pos0.copy(), vel0.copy()
    for k in range(50):
        a = (zmp - p) * lipm.w ** 2
        v = v + a * 0.002
        p = p + v * 0.002
        assert np.allclose(lipm.pos(k * 0.002 + 1e-9), p) and np.allclose(lipm.acc(k * 0.002 + 1e-9), a)
    t = 0.05
    assert np.allclose(lipm.dcm(t), lipm.pos(t) + lipm.vel(t) / lipm.w)
    assert np.allclose(lipm.zmp(t), lipm.pos(t) - lipm.acc(t) / lipm.w ** 2)
    with pytest.raises(IndexError):
        lipm.x.get_frame(10.0, 0)


def test_walk_planner_swings_i_to_i_plus_2():
    from tsid_control_amd.conf import RobotConfig
    from tsid_control_amd.footstep_planner import unicycle_path
    from tsid_control_amd.walk_planner import WalkPlanner
    conf = RobotConfig()
    init = [Footstep(np.array([0, 0.1]), np.zeros(3), 0), Footstep(np.array([0, -0.1]), np.zeros(3), 1)]
    steps = FootstepPlanner(conf.step_width, conf.step_length).plan(unicycle_path(), init)
    wp = WalkPlanner(conf)
    swings = wp.plan(steps)
    assert len(swings) == len(steps) - 2 and abs(wp.t - conf.step_duration * len(swings)) < 1e-12
    for i, sw in enumerate(swings):
        assert np.allclose(sw.get_position(0.0)[:2], steps[i].position)
        assert np.allclose(sw.get_position(conf.step_duration)[:2], steps[i + 2].position)
        assert abs(sw.get_position(conf.step_duration / 2)[2] - conf.step_height) < 1e-12
