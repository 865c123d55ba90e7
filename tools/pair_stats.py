"""Offline (build container): how many of the 170 robot<->robot candidate pairs survive each filter of the
broad/mid phase at typical poses - bounding spheres, body-frame boxes (15-axis SAT) - and how many really
intersect (LP feasibility).  Sizes the narrow-phase workload of the sim kernel."""
import sys
import numpy as np
from scipy.optimize import linprog
sys.path.insert(0, ".")
from tsid_control_amd.model import ModelBlob
from tsid_control_amd.model_compiler import quat_wxyz_to_R
from tsid_control_amd.walk_planner import op3_walking_posture

mb = ModelBlob()
par = mb["mj_parent"]; pos = mb["mj_pos"].reshape(-1, 3); quat = mb["mj_quat"].reshape(-1, 4)
adr = mb["mj_hull_adr"]; hv = mb["mj_hull_vert"].reshape(-1, 3); rb = mb["mj_rbound"].reshape(-1, 4)
pairs = mb["mj_pairs"].reshape(-1, 2)
ctrl_qidx = mb["mj_ctrl_qidx"]
box = []
for b in range(21):
    v = hv[adr[b]:adr[b + 1]]
    box.append((0.5 * (v.min(0) + v.max(0)), 0.5 * (v.max(0) - v.min(0))))

def kin(qj):
    R = [None] * 21; p = [None] * 21
    for b in range(21):
        Rq = quat_wxyz_to_R(quat[b])
        if par[b] < 0:
            R[b], p[b] = np.eye(3), np.zeros(3)
        else:
            c, s = np.cos(qj[b - 1]), np.sin(qj[b - 1])
            Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
            R[b] = R[par[b]] @ Rq @ Rz; p[b] = p[par[b]] + R[par[b]] @ pos[b]
    return R, p

def sat(Ra, ca, ha, Rb, cb, hb):
    Rm = Ra.T @ Rb; t = Ra.T @ (cb - ca); A = np.abs(Rm) + 1e-12
    for i in range(3):
        if abs(t[i]) > ha[i] + A[i] @ hb: return False
    for j in range(3):
        if abs(t @ Rm[:, j]) > ha @ A[:, j] + hb[j]: return False
    for i in range(3):
        for j in range(3):
            ra = ha[(i + 1) % 3] * A[(i + 2) % 3, j] + ha[(i + 2) % 3] * A[(i + 1) % 3, j]
            rbb = hb[(j + 1) % 3] * A[i, (j + 2) % 3] + hb[(j + 2) % 3] * A[i, (j + 1) % 3]
            if abs(t[(i + 2) % 3] * Rm[(i + 1) % 3, j] - t[(i + 1) % 3] * Rm[(i + 2) % 3, j]) > ra + rbb: return False
    return True

def intersect(A, B):
    na, nb = len(A), len(B)
    Aeq = np.zeros((5, na + nb)); beq = np.zeros(5)
    Aeq[:3, :na] = A.T; Aeq[:3, na:] = -B.T
    Aeq[3, :na] = 1; beq[3] = 1; Aeq[4, na:] = 1; beq[4] = 1
    r = linprog(np.zeros(na + nb), A_eq=Aeq, b_eq=beq, bounds=(0, None), method="highs")
    return r.status == 0

def stats(name, qj):
    R, p = kin(qj)
    ns = nb_ = nx = 0; hits = []
    for i, j in pairs:
        ci, cj = R[i] @ rb[i, :3] + p[i], R[j] @ rb[j, :3] + p[j]
        if np.linalg.norm(ci - cj) > rb[i, 3] + rb[j, 3]: continue
        ns += 1
        if not sat(R[i], R[i] @ box[i][0] + p[i], box[i][1], R[j], R[j] @ box[j][0] + p[j], box[j][1]): continue
        nb_ += 1
        H = lambda b: hv[adr[b]:adr[b + 1]] @ R[b].T + p[b]
        if intersect(H(i), H(j)): nx += 1; hits.append((int(i), int(j)))
    print(f"{name:28s} spheres {ns:3d}  boxes {nb_:3d}  intersect {nx:2d} {hits[:8]}")

rng = np.random.default_rng(0)
sim_of_tsid = np.zeros(20, int)
for a, qi in enumerate(ctrl_qidx): sim_of_tsid[qi - 7] = a
walk = np.zeros(20); walk[:] = op3_walking_posture()[np.argsort(sim_of_tsid)] if False else 0
wp = op3_walking_posture(); walk = np.array([wp[ctrl_qidx[a] - 7] for a in range(20)])
stats("standing q=0", np.zeros(20))
stats("walking posture", walk)
for k in range(6):
    stats(f"walking + U(+-0.15) #{k}", walk + rng.uniform(-0.15, 0.15, 20))
for k in range(3):
    stats(f"random U(+-0.6) #{k}", rng.uniform(-0.6, 0.6, 20))
