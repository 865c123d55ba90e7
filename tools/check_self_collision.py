"""Offline check (build container): do any of the 170 candidate robot<->robot hull pairs intersect at
a given sim pose?  LP feasibility of  A lam = B mu, lam,mu >= 0, sum = 1  after a bounding-sphere filter."""
import sys
import numpy as np
from scipy.optimize import linprog
sys.path.insert(0, ".")
from tsid_control_amd.model import ModelBlob
from tsid_control_amd.model_compiler import quat_wxyz_to_R

mb = ModelBlob()
par = mb["mj_parent"]; pos = mb["mj_pos"].reshape(-1, 3); quat = mb["mj_quat"].reshape(-1, 4)
adr = mb["mj_hull_adr"]; hv = mb["mj_hull_vert"].reshape(-1, 3); rb = mb["mj_rbound"].reshape(-1, 4)
pairs = mb["mj_pairs"].reshape(-1, 2)

def world_hulls(qj):
    R = [None] * 21; p = [None] * 21
    for b in range(21):
        Rq = quat_wxyz_to_R(quat[b])
        if par[b] < 0:
            R[b], p[b] = np.eye(3), np.zeros(3)
        else:
            c, s = np.cos(qj[b - 1]), np.sin(qj[b - 1])
            Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
            R[b] = R[par[b]] @ Rq @ Rz; p[b] = p[par[b]] + R[par[b]] @ pos[b]
    return [hv[adr[b]:adr[b + 1]] @ R[b].T + p[b] for b in range(21)], [(R[b] @ rb[b, :3] + p[b], rb[b, 3]) for b in range(21)]

def intersect(A, B):
    na, nb = len(A), len(B)
    Aeq = np.zeros((5, na + nb)); beq = np.zeros(5)
    Aeq[:3, :na] = A.T; Aeq[:3, na:] = -B.T
    Aeq[3, :na] = 1; beq[3] = 1; Aeq[4, na:] = 1; beq[4] = 1
    r = linprog(np.zeros(na + nb), A_eq=Aeq, b_eq=beq, bounds=(0, None), method="highs")
    return r.status == 0

rng = np.random.default_rng(0)
for name, qj in (("standing q=0", np.zeros(20)), ("random +-0.3", rng.uniform(-0.3, 0.3, 20)), ("random +-0.6", rng.uniform(-0.6, 0.6, 20))):
    H, S = world_hulls(qj)
    hits = []
    for i, j in pairs:
        (ci, ri), (cj, rj) = S[i], S[j]
        if np.linalg.norm(ci - cj) > ri + rj: continue
        if intersect(H[i], H[j]): hits.append((int(i), int(j)))
    print(name, "->", len(hits), "intersecting pairs", hits[:12])
