"""Oracle MuJoCo-subset step vs closed-form cases: free fall, joint servo
recurrence, resting contact force balance.  mujoco is not available to compare against (parity
unpinned)."""
import numpy as np

NQ, NV = 27, 26


def test_free_fall(oracle):
    qpos = np.zeros(NQ); qpos[2] = 2.0; qpos[3] = 1.0
    qvel, ws = np.zeros(NV), np.zeros(NV)
    r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
    assert r["ncon"] == 0 and r["nefc"] == 20
    assert np.allclose(r["qacc"][:3], [0, 0, -9.81], atol=1e-12) and np.abs(r["qacc"][3:]).max() < 1e-10
    assert np.allclose(qvel[:3], [0, 0, -9.81 * 0.002]) and abs(qpos[2] - (2.0 - 9.81 * 0.002 ** 2)) < 1e-15
    M = r["M"]
    assert np.abs(M - M.T).max() == 0 and abs(M[0, 0] - 2.873639) < 1e-9


def test_angular_momentum_free_spin(oracle):
    """Torque-free tumbling in the air (no gravity coupling to rotation): world angular momentum about
    the CoM is conserved by the bias terms; checked over a few steps to O(dt)."""
    rng = np.random.default_rng(0)
    qpos = np.zeros(NQ); qpos[2] = 5.0
    quat = rng.normal(size=4); qpos[3:7] = quat / np.linalg.norm(quat)
    qvel = np.zeros(NV); qvel[3:6] = [1.0, -2.0, 0.5]
    ws = np.zeros(NV)
    # ctrl = joint angles held at zero; frictionloss + stiff servo keep the joints nearly locked
    w0 = qvel[3:6].copy()
    r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
    # rigid-ish body: I w' + w x I w = 0 in the body frame; check that |dw| is O(dt * |w|^2 * anisotropy)
    assert np.linalg.norm(qvel[3:6] - w0) < 0.05
    assert abs(r["qacc"][2] + 9.81) < 0.5


def test_single_joint_servo_closed_form(oracle, blob):
    """One elbow displaced in free fall: its acceleration follows kp/kv/M0 and the frictionloss row."""
    qpos = np.zeros(NQ); qpos[2] = 2.0; qpos[3] = 1.0
    d = int(blob["mj_act_dof"][14])  # left_elbow
    qpos[d + 1] = 0.2
    qvel, ws = np.zeros(NV), np.zeros(NV)
    r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
    assert abs(r["qfrc_actuator"][d] + 50.0 * 0.2) < 1e-12               # kp (0 - q)
    # smooth acceleration is opposed by at most the frictionloss torque (0.1 N m)
    tau_fric = r["M"] @ (r["qacc"] - r["qacc_smooth"])
    assert abs(abs(tau_fric[d]) - 0.1) < 1e-6 and np.sign(tau_fric[d]) > 0
    assert r["qacc"][d] < 0


def test_resting_contact_balances_weight(oracle):
    qpos = np.zeros(NQ); qpos[2] = 0.3319677531 - 0.0005; qpos[3] = 1.0
    qvel, ws = np.zeros(NV), np.zeros(NV)
    for i in range(600):
        r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
        assert r["rc"] == 0
    fn = r["efc_force"][20:].sum()                                       # sum of pyramid forces = normal force
    assert abs(fn - 2.873639 * 9.81) < 0.5
    assert np.abs(qvel).max() < 0.05 and 0.3315 < qpos[2] < 0.3325
    assert set(r["con_geom"].tolist()) == {6, 12}                        # both feet, nothing else
    assert r["efc_force"][20:].min() >= 0                                # pyramid forces are non-negative
    assert r["iters"] <= 10


def test_contact_indexing_is_deterministic(oracle):
    qpos = np.zeros(NQ); qpos[2] = 0.3319677531 - 0.0005; qpos[3] = 1.0
    a = oracle.sim_step(qpos.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV), plane_mesh="all")
    b = oracle.sim_step(qpos.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV), plane_mesh="all")
    assert a["ncon"] == b["ncon"] == 16
    c = oracle.sim_step(qpos.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV))    # the default rule: at most 4 per sole
    assert c["ncon"] == 8 and set(c["con_vert"].tolist()) <= set(a["con_vert"].tolist())
    assert np.array_equal(a["con_geom"], b["con_geom"]) and np.array_equal(a["con_vert"], b["con_vert"])
    assert np.all(a["con_dist"] <= 0) and np.all(np.diff(a["con_geom"]) >= 0)


def test_reference_loop_never_touches_the_floor(oracle, params, standing):
    """What the reference's own loop does with this model: the sim's foot hull bottom sits 2.1e-7 m
    above the floor when the base is teleported to TSID's pose, so no contact forms and the sim
    velocity integrates gravity (main.py:192 resets position, not velocity)."""
    from conftest import oracle_state
    st = oracle_state(1, standing)
    for _ in range(20):
        oracle.env_step_batch(params, st)
    assert st["status"][0] == 0 and st["ncon"][0] == 0
    assert abs(st["qvel"][0, 2] + 20 * 0.002 * 9.81) < 1e-9
    assert abs(st["q"][0, 2] - standing["q"][2]) < 1e-6


def test_randomised_env_params(oracle):
    """BASELINE config 5 knobs (no reference counterpart): mass scale, contact friction, tilted floor."""
    base = np.zeros(NQ); base[2] = 0.3319677531 - 0.0005; base[3] = 1.0
    nominal = oracle.sim_step(base.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV))
    same = oracle.sim_step(base.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV), envp=[1, 1, 0, 0, 1, 0, 0, 0])
    assert np.array_equal(nominal["qacc"], same["qacc"]) and np.array_equal(nominal["con_vert"], same["con_vert"])
    # mass scale: M scales (armature does not), bias scales
    heavy = oracle.sim_step(base.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV), envp=[1.2, 1, 0, 0, 1, 0, 0, 0])
    assert abs(heavy["M"][0, 0] - 1.2 * nominal["M"][0, 0]) < 1e-12
    assert np.allclose(heavy["qfrc_bias"], 1.2 * nominal["qfrc_bias"], atol=1e-12)
    # floor raised by 1 cm along a tilted normal: every contact distance is measured to that plane
    th = np.deg2rad(5.0)
    n = np.array([np.sin(th), 0.0, np.cos(th)])
    r = oracle.sim_step(base.copy(), np.zeros(NV), np.zeros(20), np.zeros(NV), envp=[1, 0.6, *n, 0.0, 0, 0])
    assert r["ncon"] >= 1
    for pos, dist in zip(r["con_pos"], r["con_dist"]):
        assert abs((n @ pos) - 0.5 * dist) < 1e-12       # contact point sits half-way between the surfaces
    # friction enters the pyramid: with mu = 0.6 the tangential share of the contact force is bounded by it
    f = r["efc_force"][20:].reshape(-1, 4)
    assert (f >= 0).all()
    # settle on the tilted floor with high friction: the robot comes to rest
    qpos, qvel, ws = base.copy(), np.zeros(NV), np.zeros(NV)
    for _ in range(400):
        out = oracle.sim_step(qpos, qvel, np.zeros(20), ws, envp=[1, 1.0, *n, 0.0, 0, 0])
    assert out["rc"] == 0 and np.isfinite(qpos).all()


def test_non_finite_state_skips_the_step(oracle):
    qpos = np.zeros(NQ); qpos[2] = 0.5; qpos[3] = 1.0
    qvel, ws = np.zeros(NV), np.zeros(NV)
    qpos[9] = np.nan
    before = qpos.copy()
    r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
    assert r["rc"] == 4 and np.array_equal(before, qpos, equal_nan=True) and r["ncon"] == 0


def test_diverged_state_skips_the_step(oracle):
    """a finite but absurd state (the reference's teleported sim reaches 1e150 within 700 ticks of the standing loop: its
    base velocity accumulates until the contact forces explode) is treated like a non-finite one: sum |qpos| + |qvel| > 1e6"""
    for bad in (("qvel", 8, 2e6), ("qpos", 0, -3e6), ("qvel", 20, 1e150)):
        qpos = np.zeros(NQ); qpos[2] = 0.5; qpos[3] = 1.0
        qvel, ws = np.zeros(NV), np.zeros(NV)
        (qpos if bad[0] == "qpos" else qvel)[bad[1]] = bad[2]
        b4 = (qpos.copy(), qvel.copy())
        r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
        assert r["rc"] == 4 and np.array_equal(b4[0], qpos) and np.array_equal(b4[1], qvel)
    qpos = np.zeros(NQ); qpos[2] = 0.5; qpos[3] = 1.0
    qvel, ws = np.zeros(NV), np.zeros(NV)
    qvel[2] = -9e5                                            # fast, but inside the bound: stepped
    r = oracle.sim_step(qpos, qvel, np.zeros(20), ws)
    assert r["rc"] == 0 and qpos[2] < -1000


def test_plane_mesh_rule_switch(oracle, blob, standing):
    """a9 fidelity switch: 'all' takes the support vertex and EVERY hull-graph neighbour within the margin; 'mujoco' (upstream's
    plane-convex rule as far as it is known here) at most 3 more per geom, each at least 0.3 x the geom's bounding radius from
    the first contact.  Both see the same support vertices; flat feet pressed into the floor show the difference."""
    rb = np.asarray(blob["mj_rbound"]).reshape(-1, 4)
    res = {}
    for rule in ("all", "mujoco"):
        qpos = np.concatenate([standing["q"][:3], standing["q"][[6, 3, 4, 5]], np.zeros(20)])
        qpos[2] -= 0.002                       # soles 2 mm into the floor: every sole vertex is inside the margin
        qvel, ws = np.zeros(26), np.zeros(26)
        res[rule] = oracle.sim_step(qpos, qvel, np.zeros(20), ws, plane_mesh=rule)
    a, m = res["all"], res["mujoco"]
    assert a["flags"] == 0 and m["flags"] == 0
    feet = sorted(set(a["con_geom"].tolist()))
    assert len(feet) == 2 and sorted(set(m["con_geom"].tolist())) == feet
    for g in feet:
        ia, im = np.nonzero(a["con_geom"] == g)[0], np.nonzero(m["con_geom"] == g)[0]
        assert len(ia) > 4 and 1 <= len(im) <= 4
        assert a["con_vert"][ia[0]] == m["con_vert"][im[0]]                       # same support vertex first
        assert set(m["con_vert"][im].tolist()) <= set(a["con_vert"][ia].tolist())  # a subset, in the same graph order
        d = np.linalg.norm(m["con_pos"][im[1:]] - m["con_pos"][im[0]], axis=1)
        assert (d >= 0.3 * rb[g, 3] - 1e-12).all()
    # the robot is held up either way: total normal force = weight within the solver's soft-contact compliance
    for r in (a, m):
        assert abs(r["qacc"][2]) < 50.0 and r["rc"] == 0
