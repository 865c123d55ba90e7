"""The oracle's convex-hull <-> convex-hull narrow phase (oracle/or_collide.c, MPR) and stepped terrain, pinned by
solver-independent facts: intersection decisions agree with an LP feasibility test of the two vertex sets, the
reported (depth, direction) is the exact separating translation along that direction, the contact point lies
between the hulls, and a penetrating pair is pushed apart by the sim step.  PARITY UNPINNED against MuJoCo
(not vendored)."""
import numpy as np
import pytest
from scipy.optimize import linprog


def quat_R(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def kinematics(blob, qj):
    par, pos, quat = blob["mj_parent"], blob["mj_pos"].reshape(-1, 3), blob["mj_quat"].reshape(-1, 4)
    R, p = [None] * 21, [None] * 21
    for b in range(21):
        if par[b] < 0:
            R[b], p[b] = np.eye(3), np.zeros(3)
            continue
        c, s = np.cos(qj[b - 1]), np.sin(qj[b - 1])
        R[b] = R[par[b]] @ quat_R(quat[b]) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
        p[b] = p[par[b]] + R[par[b]] @ pos[b]
    return R, p


def lp_intersect(A, B):
    na, nb = len(A), len(B)
    Aeq = np.zeros((5, na + nb)); beq = np.zeros(5)
    Aeq[:3, :na] = A.T; Aeq[:3, na:] = -B.T
    Aeq[3, :na] = 1; beq[3] = 1; Aeq[4, na:] = 1; beq[4] = 1
    return linprog(np.zeros(na + nb), A_eq=Aeq, b_eq=beq, bounds=(0, None), method="highs").status == 0


def hull(blob, b):
    adr = blob["mj_hull_adr"]
    return blob["mj_hull_vert"].reshape(-1, 3)[adr[b]:adr[b + 1]]


def near_pairs(blob, R, p):
    rb = blob["mj_rbound"].reshape(-1, 4)
    for i, j in blob["mj_pairs"].reshape(-1, 2):
        ci, cj = R[i] @ rb[i, :3] + p[i], R[j] @ rb[j, :3] + p[j]
        if np.linalg.norm(ci - cj) <= rb[i, 3] + rb[j, 3]:
            yield int(i), int(j)


def test_new_blob_sections(blob):
    c, bx = blob["mj_hull_center"].reshape(21, 3), blob["mj_hull_box"].reshape(21, 6)
    for b in range(21):
        v = hull(blob, b)
        assert np.all(np.abs(v - bx[b, :3]) <= bx[b, 3:] + 1e-12)            # the box bounds the hull
        assert np.allclose(bx[b, :3] + bx[b, 3:], v.max(0)) and np.allclose(bx[b, :3] - bx[b, 3:], v.min(0))
        assert lp_intersect(v, c[b][None])                                     # the centre is inside the hull
    assert blob["mj_pairs"].reshape(-1, 2).shape == (170, 2)


def test_mpr_agrees_with_lp_and_separates_exactly(blob, oracle):
    rng = np.random.default_rng(3)
    n_hit = n_miss = 0
    for trial in range(12):
        qj = rng.uniform(-0.6, 0.6, 20)
        R, p = kinematics(blob, qj)
        for a, b in near_pairs(blob, R, p):
            A, B = hull(blob, a) @ R[a].T + p[a], hull(blob, b) @ R[b].T + p[b]
            truth = lp_intersect(A, B)
            res = oracle.mpr(a, R[a], p[a], b, R[b], p[b])
            if res is None:
                # MPR may miss a graze shallower than its tolerance, never a real overlap
                assert not truth or not lp_intersect(A, B + 1e-7 * (B.mean(0) - A.mean(0)) / np.linalg.norm(B.mean(0) - A.mean(0))), (a, b)
                n_miss += 1
                continue
            depth, d, pos = res
            assert truth, (a, b)
            assert abs(np.linalg.norm(d) - 1) < 1e-12 and depth >= 0
            assert not lp_intersect(A, B + (depth + 1e-7) * d), (a, b, depth)       # b moved by depth along d: apart
            if depth > 1e-5:
                assert lp_intersect(A, B + (depth - 1e-6) * d), (a, b, depth)       # a hair less: still overlapping
            # the contact point is within the penetration depth of both hulls' support planes along d
            assert (A @ d).max() + 1e-9 >= pos @ d >= (B @ d).min() - 1e-9
            assert np.all(pos <= np.maximum(A.max(0), B.max(0)) + 1e-9) and np.all(pos >= np.minimum(A.min(0), B.min(0)) - 1e-9)
            n_hit += 1
    assert n_hit >= 12 and n_miss >= 20


def test_mpr_is_symmetric_in_the_pair(blob, oracle):
    rng = np.random.default_rng(5)
    R, p = kinematics(blob, rng.uniform(-0.6, 0.6, 20))
    seen = 0
    for a, b in near_pairs(blob, R, p):
        r1, r2 = oracle.mpr(a, R[a], p[a], b, R[b], p[b]), oracle.mpr(b, R[b], p[b], a, R[a], p[a])
        assert (r1 is None) == (r2 is None)
        if r1 is not None and r1[0] > 1e-4:
            # not the same portal sequence, so not the same direction to rounding - but the same overlap region
            assert np.dot(r1[1], r2[1]) < 0 and abs(r1[0] - r2[0]) < 0.5 * max(r1[0], r2[0]) + 1e-3
            seen += 1
    assert seen >= 2


def arm_into_torso_state(blob, standing):
    """left shoulder rolled inwards so that the upper arm / forearm hulls penetrate the torso hull"""
    qpos = np.zeros(27); qpos[2] = 0.5; qpos[3] = 1.0
    return qpos


def test_sim_step_pushes_a_penetrating_pair_apart(blob, oracle):
    """a pose with robot<->robot penetration, held in the air (no floor contact): with self-collision the sim
    creates one contact per penetrating pair and the penetration shrinks; without it nothing happens."""
    rng = np.random.default_rng(0)
    base = None
    for trial in range(50):
        qj = rng.uniform(-0.6, 0.6, 20)
        R, p = kinematics(blob, qj)
        res = {(a, b): oracle.mpr(a, R[a], p[a], b, R[b], p[b]) for a, b in near_pairs(blob, R, p)}
        hits = [k for k, r in res.items() if r is not None]
        if 1 <= len(hits) <= 3 and all(res[k][0] > 2e-3 for k in hits):
            base = qj
            break
    assert base is not None
    qpos = np.zeros(27); qpos[2] = 1.0; qpos[3] = 1.0; qpos[7:] = base
    ctrl = base.copy()                                        # the servos hold the pose
    q1, v1, w1 = qpos.copy(), np.zeros(26), np.zeros(26)
    r = oracle.sim_step(q1, v1, ctrl, w1, self_collision=True)
    hh = r["con_body1"] >= 0
    assert r["rc"] == 0 and hh.sum() == len(hits) and r["ncon"] == len(hits)          # airborne: no floor contact
    assert sorted(zip(r["con_body1"][hh], r["con_geom"][hh])) == sorted(hits)
    assert np.all(r["con_vert"][hh] == (0x8000 | r["con_body1"][hh]))
    assert np.all(r["con_dist"][hh] < -2e-3)
    f = r["efc_force"][20:].reshape(-1, 4)
    assert np.all(f >= 0) and np.all(f.sum(1) > 0)                                  # every penetrating pair pushes back
    d0 = -r["con_dist"][hh].max()
    for _ in range(60):
        r = oracle.sim_step(q1, v1, ctrl, w1, self_collision=True)
    hh = r["con_body1"] >= 0
    assert hh.sum() == 0 or -r["con_dist"][hh].min() < 0.7 * d0                        # the overlap is being resolved
    q2, v2, w2 = qpos.copy(), np.zeros(26), np.zeros(26)
    r2 = oracle.sim_step(q2, v2, ctrl, w2, self_collision=False)
    assert r2["ncon"] == 0


def test_terrain_height_and_stepped_floor_contacts(blob, oracle, standing):
    import ctypes as C
    L = oracle.lib
    L.or_terrain_height.restype = C.c_double
    L.or_terrain_height.argtypes = [C.c_void_p, C.c_double, C.c_double]
    terr = np.zeros(20); terr[0], terr[1], terr[2], terr[3] = 0.6, 0.8, 0.013, 1.0 / 0.05
    terr[4:] = 0.01 * (np.arange(16) % 3 == 0)
    h = lambda X, Y: L.or_terrain_height(terr.ctypes.data, X, Y)
    for X, Y in ((0.0, 0.0), (0.31, -0.2), (-1.7, 0.4), (5.0, 5.0)):
        cell = int(np.floor((0.6 * X + 0.8 * Y - 0.013) * (1.0 / 0.05)))
        assert h(X, Y) == terr[4 + cell % 16]
    assert L.or_terrain_height(None, 1.0, 2.0) == 0.0
    # a robot standing on terrain: every contact's distance is measured from the step under it
    q = standing["q"].copy()
    qpos = q.copy(); qpos[3:7] = [1, 0, 0, 0]; qpos[2] -= 0.002
    flat = oracle.sim_step(qpos.copy(), np.zeros(26), np.zeros(20), np.zeros(26))
    terr2 = terr.copy(); terr2[4:] = 0.01                       # a uniformly raised floor = the flat floor 1 cm higher
    qz = qpos.copy(); qz[2] += 0.01
    up = oracle.sim_step(qz, np.zeros(26), np.zeros(20), np.zeros(26), terrain=terr2)
    assert flat["ncon"] == up["ncon"] > 0 and np.array_equal(flat["con_vert"], up["con_vert"])
    assert np.allclose(flat["con_dist"], up["con_dist"], atol=1e-12) and np.allclose(flat["qacc"], up["qacc"], atol=1e-7)
    # real steps: contacts only where the sole is over a raised cell or below the base level
    st = oracle.sim_step(qpos.copy(), np.zeros(26), np.zeros(20), np.zeros(26), terrain=terr)
    assert st["ncon"] > 0
    for c in range(st["ncon"]):
        pos = st["con_pos"][c] + 0.5 * st["con_dist"][c] * np.array([0, 0, 1.0])
        assert abs(pos[2] - h(pos[0], pos[1]) - st["con_dist"][c]) < 1e-12
