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


def _schedule(n=3, seed=4, **kw):
    import torch
    from tsid_control_amd.conf import RobotConfig
    from tsid_control_amd.walk_planner import WalkSchedule, op3_walking_conf
    conf = op3_walking_conf(RobotConfig())
    lf, rf = np.array([0.0424, 0.0446]), np.array([-0.0424, 0.0446])   # the OP3's soles at the standing pose
    com0 = np.array([0.001, 0.052, 0.2414])
    return WalkSchedule.from_demo_paths(n, conf, "cpu", torch.float64, seed=seed, q0_feet=(lf, rf), com0=com0, **kw), conf, lf, rf, com0


def test_walk_schedule_com_reference_is_a_lipm_motion():
    """com_ref(t): starts at rest at the initial CoM, is C1 across every phase boundary, obeys the LIPM
    equation acc = w^2 (pos - zmp) with the ZMP on the stance foot of each step (LIPM.py:44), and comes to
    rest between the last two footsteps."""
    sched, conf, lf, rf, com0 = _schedule()
    T, t0, w = conf.step_duration, sched.t_start, sched.omega
    r = sched.com_ref(0.0).numpy()
    assert np.allclose(r[:, :2], com0[:2], atol=1e-12) and np.abs(r[:, 3:5]).max() < 1e-12
    assert np.allclose(r[:, 2], com0[2]) and np.allclose(sched.com_ref(t0 + 1.0).numpy()[:, 2], com0[2] - sched.dz)
    ns = sched.nsteps.numpy()
    for k in range(0, int(ns.max()) + 2):
        tb = t0 + k * T
        a, b = sched.com_ref(tb - 1e-9).numpy(), sched.com_ref(tb + 1e-9).numpy()
        assert np.abs(a[:, :6] - b[:, :6]).max() < 1e-6, k          # position and velocity continuous
    # LIPM equation inside the steps: the ZMP is where the stance foot rests
    for k in (0, 1, 5, 11):
        t = t0 + (k + 0.37) * T
        r = sched.com_ref(t).numpy()
        zmp = r[:, 0:2] - r[:, 6:8] / w ** 2
        sLF, sRF, cLF, cRF = sched.sample(t)
        stance = np.where(cLF.numpy()[:, None], sLF.numpy()[:, :2], sRF.numpy()[:, :2])
        assert (cLF ^ cRF).all() and np.abs(zmp - stance).max() < 1e-9, k
    # after the plan: both feet down, CoM settles between them
    t_end = t0 + (int(ns.max()) + 4) * T
    r = sched.com_ref(t_end).numpy()
    sLF, sRF, cLF, cRF = sched.sample(t_end)
    assert cLF.all() and cRF.all()
    mid = 0.5 * (sLF.numpy()[:, :2] + sRF.numpy()[:, :2])
    assert np.abs(r[:, :2] - mid).max() < 1e-3 and np.abs(r[:, 3:5]).max() < 1e-2


def test_walk_schedule_heading_and_phases():
    """The demo path is laid along the robot's own heading (left foot stays on the left), both feet
    stay down until t_start, then the feet alternate, each swing starting and ending on its footsteps."""
    sched, conf, lf, rf, com0 = _schedule(t_start=0.4)
    sLF, sRF, cLF, cRF = sched.sample(0.2)
    assert cLF.all() and cRF.all()
    assert np.allclose(sLF.numpy()[:, :2], lf) and np.allclose(sRF.numpy()[:, :2], rf)
    assert np.allclose(sLF.numpy()[:, 3:12], np.eye(3).reshape(-1))   # yaw relative to the initial heading
    T = conf.step_duration
    prev = None
    for k in range(6):
        sLF, sRF, cLF, cRF = sched.sample(0.4 + (k + 0.5) * T)
        assert (cLF ^ cRF).all()
        if prev is not None:
            assert (cLF == ~prev).all()
        prev = cLF
        swing = np.where(cLF.numpy()[:, None], sRF.numpy(), sLF.numpy())
        assert np.allclose(swing[:, 2], conf.step_height, atol=0.0021)   # apex of the parabola (a12), above a pressed foot
    # walking direction: the left foot (at +x of the right one) is to the left of the direction of travel
    end = sched.rest[:, 6].numpy()                                    # feet before step 6
    fwd = 0.5 * (end[:, 0, :2] + end[:, 1, :2]) - 0.5 * (lf + rf)
    left = lf - rf
    assert (left[0] * fwd[:, 1] - left[1] * fwd[:, 0] < 0).all() and (np.linalg.norm(fwd, axis=1) > 0.1).all()


def test_lipm_closed_form_segment_matches_the_rollout():
    """segment / eval_segment (what WalkSchedule tabulates and k_walk evaluates) is the continuous-time limit of
    the reference's sampled rollout about a fixed ZMP (LIPM.py:34-49) and satisfies x'' = w^2 (x - zmp)."""
    from tsid_control_amd.lipm import eval_segment, segment
    lipm = LIPM(h0=0.226, dt=1e-5)
    x0, v0, zmp = np.array([0.01, -0.02]), np.array([0.08, 0.03]), np.array([0.03, 0.0])
    d, c = segment(lipm.w, zmp, x0, v0=v0)
    p, v, a = eval_segment(lipm.w, zmp, d, c, 0.0)
    assert np.allclose(p, x0) and np.allclose(v, v0) and np.allclose(a, lipm.w ** 2 * (x0 - zmp))
    lipm.make_trajectory([0.0, 0.2], 1e-5, x0, v0, np.zeros(2), zmp)
    for t in (0.05, 0.1, 0.19):
        p, v, a = eval_segment(lipm.w, zmp, d, c, t)
        # the reference's rollout uses acc = (zmp - pos) w^2, i.e. the opposite sign of the pendulum
        # equation it names; the closed form follows the pendulum (LIPM.py:31-32's zmp identity holds for it)
        assert np.allclose(a, lipm.w ** 2 * (p - zmp), rtol=1e-12)
        h = 1e-6
        fd = (eval_segment(lipm.w, zmp, d, c, t + h)[0] - 2 * p + eval_segment(lipm.w, zmp, d, c, t - h)[0]) / h ** 2
        assert np.allclose(fd, a, rtol=1e-4, atol=1e-5)
        assert np.allclose(p + v / lipm.w, zmp + d * np.exp(lipm.w * t))           # the DCM grows as e^{w t}
