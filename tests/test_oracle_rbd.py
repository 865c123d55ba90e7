"""Oracle rigid-body terms vs known answers and invariants.  The reference holds
no golden vectors for these (parity unpinned); these pins are solver-independent."""
import numpy as np

NV = 26


def rand_state(rng, standing):
    q = standing["q"].copy()
    q[7:] += rng.uniform(-0.6, 0.6, 20)
    quat = rng.normal(size=4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[:3] += rng.normal(size=3)
    return q, rng.normal(size=NV)


def test_known_answers(blob, oracle, standing):
    t0 = oracle.terms(blob.q0, np.zeros(NV))
    assert abs(t0["mass"] - 2.893639) < 1e-9
    # sole frames at q = 0 in the torso frame
    assert np.allclose(t0["oMf"][0][9:] - blob.q0[:3], [0.04236, 0.044586, -0.331968], atol=2e-6)
    assert np.allclose(t0["oMf"][1][9:] - blob.q0[:3], [-0.04236, 0.044586, -0.331968], atol=2e-6)
    assert np.allclose(t0["oMf"][0][:9].reshape(3, 3), np.eye(3), atol=2e-5)
    assert abs(standing["q"][2] - 0.331968) < 1e-6                       # WalkController.py:74
    assert np.allclose(standing["terms"]["com"] - standing["q"][:3], [-0.000079, 0.052064, -0.090264], atol=1e-6)
    assert abs(standing["terms"]["oMf"][0][11]) < 1e-12                   # left sole on z = 0


def test_mass_matrix_invariants(oracle, standing):
    rng = np.random.default_rng(0)
    for _ in range(5):
        q, v = rand_state(rng, standing)
        t = oracle.terms(q, v)
        M = t["M"]
        assert np.abs(M - M.T).max() == 0.0
        assert np.linalg.eigvalsh(M).min() > 0
        b0 = oracle.rnea(q, np.zeros(NV), np.zeros(NV))
        Mr = np.stack([oracle.rnea(q, np.zeros(NV), np.eye(NV)[i]) - b0 for i in range(NV)], 1)
        assert np.abs(M - Mr).max() < 1e-12                               # CRBA == RNEA columns
        assert np.abs(t["h"] - oracle.rnea(q, v, np.zeros(NV))).max() < 1e-13
        assert abs(M[:3, :3].trace() / 3 - t["mass"]) < 1e-12             # base linear block = m I (rotated)


def test_jacobians_by_finite_differences(oracle, standing):
    rng = np.random.default_rng(1)
    q, v = rand_state(rng, standing)
    t = oracle.terms(q, v)
    eps = 1e-6

    def fd(fun):
        cols = []
        for i in range(NV):
            d = np.zeros(NV); d[i] = eps
            cols.append((fun(oracle.integrate(q, d)) - fun(oracle.integrate(q, -d))) / (2 * eps))
        return np.stack(cols, 1)

    assert np.abs(fd(lambda qq: oracle.terms(qq, v)["com"]) - t["Jcom"]).max() < 1e-8
    assert np.abs(t["Jcom"] @ v - t["vcom"]).max() < 1e-13
    for f in range(2):
        R = t["oMf"][f][:9].reshape(3, 3)
        Jp = fd(lambda qq: oracle.terms(qq, v)["oMf"][f][9:])
        assert np.abs(R.T @ Jp - t["Jf"][f][:3]).max() < 1e-8
        assert np.abs(t["Jf"][f] @ v - t["vf"][f]).max() < 1e-13


def test_drift_accelerations(oracle, standing):
    """Jdot v terms: d/dt of (J v) along the flow of v with zero joint acceleration."""
    rng = np.random.default_rng(2)
    q, v = rand_state(rng, standing)
    t = oracle.terms(q, v)
    dt = 1e-6
    tp, tm = oracle.terms(oracle.integrate(q, v * dt), v), oracle.terms(oracle.integrate(q, -v * dt), v)
    assert np.abs((tp["Jcom"] @ v - tm["Jcom"] @ v) / (2 * dt) - t["acom"]).max() < 1e-7
    for f in range(2):
        R = t["oMf"][f][:9].reshape(3, 3)
        Rp, Rm = tp["oMf"][f][:9].reshape(3, 3), tm["oMf"][f][:9].reshape(3, 3)
        a_lin = R.T @ (Rp @ (tp["Jf"][f][:3] @ v) - Rm @ (tm["Jf"][f][:3] @ v)) / (2 * dt)
        a_ang = R.T @ (Rp @ (tp["Jf"][f][3:] @ v) - Rm @ (tm["Jf"][f][3:] @ v)) / (2 * dt)
        assert np.abs(a_lin - t["af"][f][:3]).max() < 1e-6 and np.abs(a_ang - t["af"][f][3:]).max() < 1e-6


def test_se3_exp_log_roundtrip(oracle):
    rng = np.random.default_rng(3)
    for _ in range(20):
        nu = rng.normal(size=6) * rng.choice([1e-9, 1e-3, 1.0])
        if np.linalg.norm(nu[3:]) > 3.0:  # log6 returns the principal angle (< pi)
            nu[3:] *= 3.0 / np.linalg.norm(nu[3:])
        q0 = np.zeros(27); q0[6] = 1.0
        q1 = oracle.integrate(q0, np.concatenate([nu, np.zeros(20)]))
        x, y, z, w = q1[3:7]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        assert np.abs(oracle.log6(R, q1[:3]) - nu).max() < 1e-9 * max(1.0, np.abs(nu).max()) + 1e-12


def _quat_R(q):
    x, y, z, w = q[3:7] / np.linalg.norm(q[3:7])
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_centroidal_angular_momentum(oracle, standing):
    """Angular-momentum task terms (legacy/biped.py:82-87).  A_G,ang against the mass matrix
    (the base rows of M v are the robot's spatial momentum in the base frame, so L_G follows from CRBA
    alone), its drift against the change of L_G along a motion with zero generalized acceleration."""
    rng = np.random.default_rng(7)
    for _ in range(5):
        q, v = rand_state(rng, standing)
        t = oracle.terms(q, v)
        R = _quat_R(q)
        hb = t["M"][:6] @ v                                   # momentum in the base frame, about the base origin
        p, LO = R @ hb[:3], R @ hb[3:]
        LG = LO - np.cross(t["com"] - q[:3], p)
        assert np.abs(t["Aam"] @ v - LG).max() < 1e-12 and np.abs(t["Lam"] - LG).max() < 1e-12
        assert np.abs(p - t["mass"] * t["vcom"]).max() < 1e-12  # same check on the linear part (Jcom)
        # every column the same way: unit velocities
        for c in (0, 4, 9, 17, 25):
            e = np.eye(NV)[c]
            hb = t["M"][:6] @ e
            assert np.abs(t["Aam"][:, c] - (R @ hb[3:] - np.cross(t["com"] - q[:3], R @ hb[:3]))).max() < 1e-12
        # drift = d/dt L_G at zero acceleration (central difference along the geodesic)
        h = 1e-6
        Lp = oracle.terms(oracle.integrate(q, v * h), v)["Lam"]
        Lm = oracle.terms(oracle.integrate(q, -v * h), v)["Lam"]
        assert np.abs((Lp - Lm) / (2 * h) - t["dLam"]).max() < 1e-6
