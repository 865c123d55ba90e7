// tsidb_sim.hpp - one MuJoCo-subset forward-dynamics + contact step for one env on one wavefront.
//
// Replaces, per env, main.py:192-195: base teleport (qpos[:7] = q[:7]), ctrl = map_tsid_to_mujoco(q),
// mujoco.mj_step.  Feature subset: SURVEY.md 3.4 (free joint + 20 z-hinges, armature, frictionloss,
// position actuators, floor-plane <-> convex-hull contacts, soft constraints, pyramidal cones,
// Newton solver with warm start, semi-implicit Euler).
//
// MI355X mapping: one wavefront per env, everything in LDS/registers.  Constraint Jacobians are
// never materialised: a contact row is (direction, point, body), so J x is a body twist evaluated at
// the point, J^T f is a wrench pushed up the tree, and the Newton Hessian M + J^T D J is the
// composite-rigid-body recursion run on body inertia plus a per-body 6x6 "contact inertia"
// sum_rows D w w^T.  The 26x26 factorisations run in registers (lane i = row i, v_readlane).
// Spatial vectors: [lin; ang], world axes, about the base origin O (z distances use absolute O.z).
#pragma once
#include "tsidb_common.hpp"
#include "tsidb_tick.hpp" // rdlane
#include "tsidb_topology.hpp"

namespace tsidb {

constexpr int LDM = 27;

// floor plane n.x = d with its contact frame (mju_makeFrame: t1 from y unless |n_y| >= 0.5, t2 = n x t1)
template <typename T>
struct Floor {
  T n[3], t1[3], t2[3], d;
};

template <typename T>
struct SimLds {
  Floor<T> fl; // per-env floor frame: read where needed instead of ten live registers per lane
  T S[NV][6];
  union { // tree-pass scratch is dead once bias forces and M exist; the Newton loop reuses the space
    struct { T V[NB][6], A[NB][6], f[NB][6], Yc[NB][10]; };
    struct { T K[NB][21]; }; // per-body contact inertia (packed sym 6x6), composite over subtrees
  };
  T M[NV * LDM];
  union { // body frames are needed until the contacts exist; per-contact inertias are folded into K
          // before the Hessian is assembled
    struct { T R[NB][9], p[NB][3]; };
    T H[NV * LDM];
    T Wc[MAXCON][21];
  };
  T qpos[NQ], qvel[NV];
  T xv[NV];
  int cbody[MAXCON], cvert[MAXCON];
  unsigned anc[NB]; // ancestor bitmask per body (copied from the model: LDS latency, not global)
  T cr[MAXCON][3], cdist[MAXCON], cfv[MAXCON][3]; // contact point (rel O), distance, force vector
};

__device__ __forceinline__ int sym_idx(int i, int j) { // packed upper index of a symmetric 6x6
  const int a = i < j ? i : j, b = i < j ? j : i;
  return a * 6 - a * (a - 1) / 2 + (b - a);
}

// Solve A x = rhs for the 26x26 SPD matrix whose row `lane` is in a[] (lanes >= 26 hold zeros), where A
// has the sparsity of a kinematic tree (A[i][j] != 0 only if dof i is an ancestor of dof j or vice
// versa) - true for the joint-space inertia M and for M + J^T D J with contact/friction rows.
// Eliminating leaves first (k = 25 .. 0) gives A = U U^T with U upper triangular and NO fill-in, so
// only ancestor pairs are touched; MJ_DOFANC is a compile-time table, so the unrolled code simply
// does not contain the zero updates.  U[i][k] ends up in lane i's a[k] (k >= i).
template <typename T>
__device__ __forceinline__ T chol26_solve(T (&a)[NV], T rhs, int lane, bool &spd) {
  spd = true;
  T rd[NV]; // 1 / U[k][k], wave-uniform
#pragma unroll
  for (int k = NV - 1; k >= 0; k--) {
    const T akk = rdlane(a[k], k);
    if (!(akk > 0)) spd = false;
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    rd[k] = rk;
    const T uik = lane < k ? a[k] * rk : (lane == k ? akk * rk : T(0));
    a[k] = uik;
#pragma unroll
    for (int j = 0; j < k; j++)
      if ((MJ_DOFANC[k] >> j) & 1u) a[j] -= uik * rdlane(uik, j);
  }
  // U y = rhs (k descending; lane k contributes y_k), then U^T x = y (k ascending, wave-uniform)
  T acc = rhs, y[NV];
#pragma unroll
  for (int k = NV - 1; k >= 0; k--) {
    y[k] = rdlane(acc, k) * rd[k];
    acc -= a[k] * y[k];
  }
  T xs[NV], x = 0;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    T s0 = y[k];
#pragma unroll
    for (int i = 0; i < k; i++)
      if ((MJ_DOFANC[k] >> i) & 1u) s0 -= rdlane(a[k], i) * xs[i];
    xs[k] = s0 * rd[k];
    if (lane == k) x = xs[k];
  }
  return x;
}

// out = M * x for the lane's dof (x in LDS)
template <typename T> __device__ __forceinline__ T mulM(const SimLds<T> &L, const T *x, int lane) {
  T s = 0;
  if (lane < NV)
    for (int j = 0; j < NV; j++) s += L.M[lane * LDM + j] * x[j];
  return s;
}

// the 4 pyramid rows of contact `c` applied to generalized vector x (LDS): J_row x
template <typename T>
__device__ __forceinline__ void contact_rows(const DevModel<T> &m, const SimLds<T> &L, const Floor<T> &fl, int c, const T *x,
                                             T mu, T *out) {
  T tw[6] = {0, 0, 0, 0, 0, 0};
  for (unsigned mk = L.anc[L.cbody[c]]; mk; mk &= mk - 1) {
    const int a = __ffs(mk) - 1;
    if (a == 0) {
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const T xk = x[k];
#pragma unroll
        for (int i = 0; i < 6; i++) tw[i] += L.S[k][i] * xk;
      }
    } else {
      const T xk = x[5 + a];
#pragma unroll
      for (int i = 0; i < 6; i++) tw[i] += L.S[5 + a][i] * xk;
    }
  }
  T wxr[3];
  cross3(tw + 3, L.cr[c], wxr);
  const T u[3] = {tw[0] + wxr[0], tw[1] + wxr[1], tw[2] + wxr[2]};
  const T un = dot3(fl.n, u), u1 = dot3(fl.t1, u), u2 = dot3(fl.t2, u);
  out[0] = un + mu * u1; out[1] = un - mu * u1; out[2] = un + mu * u2; out[3] = un - mu * u2;
}

// per-lane constraint bookkeeping (friction row of dof `lane`, contact `lane`)
template <typename T>
struct RowState {
  // friction dof row
  bool has_f;
  T fD, fR, floss, faref, fjar, fJv;
  // contact rows
  bool has_c;
  T cD, caref[4], cjar[4], cJv[4];
};

// cost / first / second derivative contribution of this lane's rows at jar + alpha*Jv
template <typename T>
__device__ __forceinline__ void rows_eval(const RowState<T> &rs, T alpha, T &c, T &g, T &h) {
  c = 0; g = 0; h = 0;
  if (rs.has_f) {
    const T x = rs.fjar + alpha * rs.fJv, f = rs.floss, r = rs.fR;
    if (x <= -r * f) { c += -T(0.5) * r * f * f - f * x; g += -f * rs.fJv; }
    else if (x >= r * f) { c += -T(0.5) * r * f * f + f * x; g += f * rs.fJv; }
    else { c += T(0.5) * rs.fD * x * x; g += rs.fD * x * rs.fJv; h += rs.fD * rs.fJv * rs.fJv; }
  }
  if (rs.has_c) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const T x = rs.cjar[i] + alpha * rs.cJv[i];
      if (x < 0) { c += T(0.5) * rs.cD * x * x; g += rs.cD * x * rs.cJv[i]; h += rs.cD * rs.cJv[i] * rs.cJv[i]; }
    }
  }
}

template <typename T>
__device__ __forceinline__ void sim_step_env(const DevModel<T> &m, SimLds<T> &L, int lane, const T *q_tsid, const T *v_tsid, T *qpos_g, T *qvel_g,
                             T *qacc_ws_g, const T *envp, const T *motor_tau, T *qacc_out, int *ncon_out, int *con_out,
                             int *info) {
  // per-env randomisation (BASELINE config 5), NULL = nominal: mass scale, contact friction, floor plane
  const T mscale = envp ? envp[0] : T(1);
  Floor<T> &fl = L.fl;
  if (lane == 0) {
    fl.n[0] = envp ? envp[2] : T(0); fl.n[1] = envp ? envp[3] : T(0); fl.n[2] = envp ? envp[4] : T(1);
    fl.d = envp ? envp[5] : T(0);
    T t[3] = {0, 0, 0};
    if (fabs(fl.n[1]) < T(0.5)) t[1] = 1; else t[2] = 1;
    const T dn = dot3(fl.n, t);
    T nn = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { fl.t1[i] = t[i] - dn * fl.n[i]; nn += fl.t1[i] * fl.t1[i]; }
    nn = T(1) / sqrt(nn);
#pragma unroll
    for (int i = 0; i < 3; i++) fl.t1[i] *= nn;
    cross3(fl.n, fl.t1, fl.t2);
  }
  const T dt = m.opt[0], gz = m.opt[1], tol = m.opt[2];
  const int maxiter = (int)m.opt[3], ls_iter = (int)m.opt[4];
  const T ls_tol = m.opt[5];
  const T INF = Eps<T>::inf;
  const T MINVAL = T(1e-15);
  const bool quirks = m.params[P_QUIRKS] != 0;

  TSIDB_STAMP(16);
  // ---- stage state; teleport the base and map joint targets (main.py:192-194)
  if (lane < NQ) {
    T val = qpos_g[lane];
    if (q_tsid) {
      if (lane < 3) val = q_tsid[lane];
      else if (lane < 7) val = quirks ? q_tsid[lane] : (lane == 3 ? q_tsid[6] : q_tsid[lane - 1]);
    }
    L.qpos[lane] = val;
  }
  if (lane < NV) {
    T val = qvel_g[lane];
    if (q_tsid && v_tsid && !quirks && lane < 6) {
      // the base is kinematic in this mode: its velocity comes with its pose (the reference writes
      // qpos[:7] only, main.py:192, which is harmless only while the robot stands still).  TSID: linear
      // and angular velocity in the body frame; sim: linear in the world frame, angular in the body frame.
      if (lane < 3) {
        const T x = q_tsid[3], y = q_tsid[4], z = q_tsid[5], w = q_tsid[6];
        const T r0 = lane == 0 ? 1 - 2 * (y * y + z * z) : lane == 1 ? 2 * (x * y + w * z) : 2 * (x * z - w * y);
        const T r1 = lane == 0 ? 2 * (x * y - w * z) : lane == 1 ? 1 - 2 * (x * x + z * z) : 2 * (y * z + w * x);
        const T r2 = lane == 0 ? 2 * (x * z + w * y) : lane == 1 ? 2 * (y * z - w * x) : 1 - 2 * (x * x + y * y);
        val = r0 * v_tsid[0] + r1 * v_tsid[1] + r2 * v_tsid[2];
      } else {
        val = v_tsid[lane];
      }
    }
    L.qvel[lane] = val;
  }
  const T myctrl = (lane < NA && q_tsid) ? q_tsid[m.mj_ctrl_qidx[lane]] : T(0); // joint target of actuator `lane`
  for (int i = lane; i < NV * LDM; i += WAVE) L.M[i] = 0;
  __syncthreads();
  const T Oz = L.qpos[2];

  // ---- kinematics, velocities, bias accelerations: parent-independent work up front, the depth
  //      loop carries R, p, V, A down the tree, body inertias/forces in one parallel pass
  T Rb[9], pb[3], Vb[6], Ab[6], Sb[6], qd = 0;
  const int up0 = lane < NB ? m.mj_up[0][lane] : -1, up1 = lane < NB ? m.mj_up[1][lane] : -1,
            up2 = lane < NB ? m.mj_up[2][lane] : -1;
  const unsigned bodyanc = lane < NB ? m.mj_anc[lane] : 0u;
  const unsigned dofanc = lane < NV ? m.mj_anc[lane < 6 ? 0 : lane - 5] : 0u;
  if (lane < NB) L.anc[lane] = bodyanc;
#pragma unroll
  for (int i = 0; i < 6; i++) { Vb[i] = 0; Ab[i] = 0; Sb[i] = 0; }
#pragma unroll
  for (int i = 0; i < 3; i++) pb[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) Rb[i] = 0;
  if (lane == 0) {
    quat_to_R(L.qpos[4], L.qpos[5], L.qpos[6], L.qpos[3], Rb); // wxyz storage
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        L.S[k][i] = (i == k) ? T(1) : T(0); L.S[k][3 + i] = 0;        // world-frame linear dofs
        L.S[3 + k][i] = 0; L.S[3 + k][3 + i] = Rb[3 * i + k];          // body-frame angular dofs
      }
    }
    T wl[3] = {L.qvel[3], L.qvel[4], L.qvel[5]};
    Vb[0] = L.qvel[0]; Vb[1] = L.qvel[1]; Vb[2] = L.qvel[2];
    mat3vec(Rb, wl, Vb + 3);
    // dS/dt = V x S for the body-fixed angular axes, 0 for the world-fixed linear ones
#pragma unroll
    for (int k = 0; k < 3; k++) {
      T Sk[6] = {0, 0, 0, Rb[k], Rb[3 + k], Rb[6 + k]}, dS[6];
      cross_mm(Vb, Sk, dS);
#pragma unroll
      for (int i = 0; i < 6; i++) Ab[i] += dS[i] * wl[k];
    }
  } else if (lane < NB) {
    // the body's transform in its parent: body rotation times Rz(theta) of its hinge, body offset
    const T *Rq = m.mj_R[lane];
    const T th = L.qpos[6 + lane];
    T c, s;
    sincos_t(th, s, c);
    qd = L.qvel[5 + lane];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Rb[3 * i + 0] = c * Rq[3 * i] + s * Rq[3 * i + 1];
      Rb[3 * i + 1] = -s * Rq[3 * i] + c * Rq[3 * i + 1];
      Rb[3 * i + 2] = Rq[3 * i + 2];
      pb[i] = m.mj_pos[lane][i];
    }
  }
  tree_forward<T>(lane, NB, up0, up1, up2, bodyanc, &L.R[0][0], &L.V[0][0], &L.f[0][0], 6, &L.Yc[0][0], 10, Rb, pb, qd, Sb,
                  Vb, Ab);
  if (lane < NB) {
    const int b = lane;
    if (b > 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) L.S[5 + b][i] = Sb[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) L.R[b][i] = Rb[i];
#pragma unroll
    for (int i = 0; i < 3; i++) L.p[b][i] = pb[i];
#pragma unroll
    for (int i = 0; i < 6; i++) { L.V[b][i] = Vb[i]; L.A[b][i] = Ab[i]; }
  }
  {
    // body inertias / forces in one parallel pass, then composite inertias and subtree forces as
    // prefix-scan differences over the lanes (bodies are numbered depth-first)
    T fb[6] = {0, 0, 0, 0, 0, 0}, Y[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < NB) {
      const int b = lane;
      const T *Yb = m.mj_inertia[b];
      T cw[3], I[9] = {mscale * Yb[4], mscale * Yb[5], mscale * Yb[6], mscale * Yb[5], mscale * Yb[7], mscale * Yb[8],
                       mscale * Yb[6], mscale * Yb[8], mscale * Yb[9]}, Tm[9], RT[9];
      mat3vec(Rb, Yb + 1, cw);
#pragma unroll
      for (int i = 0; i < 3; i++) cw[i] += pb[i];
      mat3mul(Rb, I, Tm);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) RT[3 * i + k] = Rb[3 * k + i];
      mat3mul(Tm, RT, I);
      const T mass = mscale * Yb[0], c2 = dot3(cw, cw);
      Y[0] = mass; Y[1] = mass * cw[0]; Y[2] = mass * cw[1]; Y[3] = mass * cw[2];
      Y[4] = I[0] + mass * (c2 - cw[0] * cw[0]); Y[5] = I[1] - mass * cw[0] * cw[1]; Y[6] = I[2] - mass * cw[0] * cw[2];
      Y[7] = I[4] + mass * (c2 - cw[1] * cw[1]); Y[8] = I[5] - mass * cw[1] * cw[2];
      Y[9] = I[8] + mass * (c2 - cw[2] * cw[2]);
      T Ag[6] = {Ab[0], Ab[1], Ab[2] - gz, Ab[3], Ab[4], Ab[5]}, Ya[6], Yv[6], vx[6];
      yo_mul(Y, Ag, Ya);
      yo_mul(Y, Vb, Yv);
      cross_mf(Vb, Yv, vx);
#pragma unroll
      for (int i = 0; i < 6; i++) fb[i] = Ya[i] + vx[i];
    }
    const int mylast = lane < NB ? m.mj_last[lane] : lane;
#pragma unroll
    for (int i = 0; i < 6; i++) fb[i] = subtree_sum32(fb[i], mylast);
#pragma unroll
    for (int i = 0; i < 10; i++) Y[i] = subtree_sum32(Y[i], mylast);
    if (lane < NB) {
#pragma unroll
      for (int i = 0; i < 6; i++) L.f[lane][i] = fb[i];
#pragma unroll
      for (int i = 0; i < 10; i++) L.Yc[lane][i] = Y[i];
    }
  }
  __syncthreads();
  // ---- per dof: bias, mass-matrix column (+ armature), actuation
  T qfs = 0;
  if (lane < NV) {
    const int k = lane, bk = k < 6 ? 0 : k - 5;
    T Sk[6], Fk[6], hk = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { Sk[i] = L.S[k][i]; hk += Sk[i] * L.f[bk][i]; }
    yo_mul(L.Yc[bk], Sk, Fk);
    for (unsigned mk = dofanc; mk; mk &= mk - 1) {
      const int a = __ffs(mk) - 1;
      const int i0 = a == 0 ? 0 : 5 + a, i1 = a == 0 ? 5 : 5 + a;
      for (int i = i0; i <= i1; i++) {
        if (i > k) continue;
        T val = 0;
#pragma unroll
        for (int e = 0; e < 6; e++) val += L.S[i][e] * Fk[e];
        L.M[i * LDM + k] = val;
        L.M[k * LDM + i] = val;
      }
    }
    qfs = -hk;
  }
  __syncthreads();
  if (lane < NV) L.M[lane * LDM + lane] += m.mj_armature[lane];
  if (lane < NA) {
    const int d = m.mj_act_dof[lane];
    // closed loop: TSID's joint torques as motor forces; otherwise the reference's position servos
    L.xv[d] = motor_tau ? motor_tau[m.mj_ctrl_qidx[lane] - 7]
                        : m.mj_act_kp[lane] * (myctrl - L.qpos[d + 1]) - m.mj_act_kv[lane] * L.qvel[d];
  } else if (lane >= 32 && lane < 38) L.xv[lane - 32] = 0;
  __syncthreads();
  if (lane < NV) qfs += L.xv[lane];
  TSIDB_STAMP(17);
  // ---- qacc_smooth = M^-1 qfrc_smooth
  T arow[NV];
#pragma unroll
  for (int j = 0; j < NV; j++) arow[j] = lane < NV ? L.M[lane * LDM + j] : T(0);
  bool spd;
  const T qas = chol26_solve(arow, qfs, lane, spd);
  int fail = spd ? 0 : 1;

  TSIDB_STAMP(18);
  // ---- collision: floor plane (n.x = d; nominal z = 0) against each body's convex hull
  const T margin = 0, tie_tol = m.opt[6];
  const T Ow[3] = {L.qpos[0], L.qpos[1], L.qpos[2]};
  const T nO = dot3(fl.n, Ow) - fl.d; // signed distance of the base origin O to the floor
  int ncon = 0;
  // bounding-sphere pretest for all bodies at once (lane = body, whose rotation and position are still in
  // this lane's registers); only the bodies that can reach the floor enter the support search, in body order
  unsigned long long cand_bodies;
  {
    bool near = false;
    if (lane < NB) {
      const T c6 = fl.n[0] * Rb[0] + fl.n[1] * Rb[3] + fl.n[2] * Rb[6];
      const T c7 = fl.n[0] * Rb[1] + fl.n[1] * Rb[4] + fl.n[2] * Rb[7];
      const T c8 = fl.n[0] * Rb[2] + fl.n[1] * Rb[5] + fl.n[2] * Rb[8];
      const T pzl = dot3(fl.n, pb) + nO;
      const T zc = c6 * m.rbound[lane][0] + c7 * m.rbound[lane][1] + c8 * m.rbound[lane][2] + pzl;
      near = !(zc - m.rbound[lane][3] > margin);
    }
    cand_bodies = __ballot(near);
  }
  for (unsigned long long bm = cand_bodies; bm && ncon < MAXCON; bm &= bm - 1) {
    const int b = __ffsll((long long)bm) - 1;
    const T *Rb = L.R[b];
    // floor normal in the body frame; "z" below = signed distance to the floor
    const T r6 = fl.n[0] * Rb[0] + fl.n[1] * Rb[3] + fl.n[2] * Rb[6];
    const T r7 = fl.n[0] * Rb[1] + fl.n[1] * Rb[4] + fl.n[2] * Rb[7];
    const T r8 = fl.n[0] * Rb[2] + fl.n[1] * Rb[5] + fl.n[2] * Rb[8];
    const T pz = dot3(fl.n, L.p[b]) + nO;
    const int v0 = m.hull_adr[b], v1 = m.hull_adr[b + 1];
    // exact pruned support search: the hull's vertices are stored in k-d order, 64 per chunk, each
    // chunk with a bounding box.  A chunk can hold the lowest vertex (or one within the tie
    // tolerance of it) only if the box's lower bound along the floor normal does not exceed the best
    // z found so far; every such chunk is scanned, so the result equals the exhaustive search.
    const int c0 = m.chunk_adr[b], nch = m.chunk_adr[b + 1] - c0;
    T zlb = INF;
    if (lane < nch) {
      const T *bx = m.chunk_box + 6 * (c0 + lane);
      zlb = r6 * bx[0] + r7 * bx[1] + r8 * bx[2] + pz - (fabs(r6) * bx[3] + fabs(r7) * bx[4] + fabs(r8) * bx[5]);
      zlb -= fabs(zlb) * T(4) * Eps<T>::v; // rounding slack: never prune a chunk that could matter
    }
    unsigned long long scanned = 0;
    T zmin = INF;
    T zmine = 0; // this lane's vertex z in the chunk being scanned
    {
      T zl = zlb;
      int ci = lane;
      wave_argmin(zl, ci);
      unsigned long long pend = 1ull << ci;
      while (pend) {
        const int c = __ffsll((long long)pend) - 1;
        const int i = v0 + WAVE * c + lane;
        T z = INF;
        if (i < v1) z = r6 * m.hull_x[i] + r7 * m.hull_y[i] + r8 * m.hull_z[i] + pz;
        const T zc2 = wave_min(z);
        zmin = zc2 < zmin ? zc2 : zmin;
        scanned |= 1ull << c;
        pend = __ballot(lane < nch && zlb <= zmin + tie_tol) & ~scanned;
      }
    }
    (void)zmine;
    if (zmin > margin) continue;
    // lowest-index vertex within the tie tolerance of the minimum (chunks are in index order)
    int best = 0x7fffffff;
    {
      const T zt = zmin + tie_tol;
      for (unsigned long long sm = scanned; sm && best == 0x7fffffff; sm &= sm - 1) {
        const int c = __ffsll((long long)sm) - 1;
        const int i = v0 + WAVE * c + lane;
        int cand = 0x7fffffff;
        if (i < v1) {
          const T z = r6 * m.hull_x[i] + r7 * m.hull_y[i] + r8 * m.hull_z[i] + pz;
          if (z <= zt) cand = i;
        }
        best = wave_min_int(cand);
      }
    }
    const int e0 = m.hull_eadr[best];
    int nnb = m.hull_eadr[best + 1] - e0;
    nnb = nnb > WAVE - 1 ? WAVE - 1 : nnb;
    bool keep = false;
    T w[3] = {0, 0, 0}, wd = 0;
    int vid = best;
    if (lane <= nnb) {
      if (lane > 0) vid = v0 + m.hull_edge[e0 + lane - 1];
      const T vv[3] = {m.hull_x[vid], m.hull_y[vid], m.hull_z[vid]};
      mat3vec(Rb, vv, w);
      w[0] += L.p[b][0]; w[1] += L.p[b][1]; w[2] += L.p[b][2]; // relative to O
      wd = dot3(fl.n, w) + nO;
      keep = lane == 0 || wd <= margin;
    }
    const unsigned long long mask = __ballot(keep);
    const int slot = ncon + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep && slot < MAXCON) {
      const T dist = wd;
      L.cbody[slot] = b;
      L.cvert[slot] = vid - v0;
      L.cdist[slot] = dist;
#pragma unroll
      for (int i = 0; i < 3; i++) L.cr[slot][i] = w[i] - T(0.5) * dist * fl.n[i];
    }
    ncon += __popcll(mask);
    ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  __syncthreads();
  if (lane == 0 && ncon_out) ncon_out[0] = ncon;
  if (con_out && lane < MAXCON) con_out[lane] = lane < ncon ? ((L.cbody[lane] << 16) | L.cvert[lane]) : -1;

  TSIDB_STAMP(19);
  // ---- constraint rows: frictionloss (lane = dof), pyramidal contact rows (lane = contact)
  const T mu = envp ? envp[1] : m.contact[0];
  const T timeconst = m.contact[1] > 2 * dt ? m.contact[1] : 2 * dt, dampratio = m.contact[2];
  const T dmin = m.contact[3], dmax = m.contact[4], width = m.contact[5], mid = m.contact[6], power = m.contact[7];
  const T kk = T(1) / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), bb = T(2) / (dmax * timeconst);
  RowState<T> rs;
  rs.has_f = lane < NV && m.mj_frictionloss[lane < NV ? lane : 0] > 0;
  rs.fD = rs.fR = rs.floss = rs.faref = rs.fjar = rs.fJv = 0;
  if (rs.has_f) {
    const T r = (1 - dmin) / dmin * m.mj_dof_invw0[lane];
    rs.fR = r > MINVAL ? r : MINVAL;
    rs.fD = T(1) / rs.fR;
    rs.floss = m.mj_frictionloss[lane];
    rs.faref = -bb * L.qvel[lane];
  }
  rs.has_c = lane < ncon;
  rs.cD = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) rs.caref[i] = rs.cjar[i] = rs.cJv[i] = 0;
  if (rs.has_c) {
    const int c = lane;
    const T dist = L.cdist[c];
    T x = fabs(dist - margin) / width, imp;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      T y;
      if (power == T(2)) { // MuJoCo's default solimp exponent: x^2 and mid^1 are exact, no pow() needed
        if (x <= mid) y = x * x / mid;
        else y = 1 - (1 - x) * (1 - x) / (1 - mid);
      } else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = dmin + y * (dmax - dmin);
    }
    const T tran = m.mj_body_invw0[L.cbody[c]][0];
    const T diagA = tran + mu * mu * tran;
    T R0 = (1 - imp) / imp * diagA;
    R0 = R0 > MINVAL ? R0 : MINVAL;
    rs.cD = T(1) / (2 * mu * mu * R0);
    T vel[4];
    contact_rows(m, L, fl, c, L.qvel, mu, vel);
#pragma unroll
    for (int i = 0; i < 4; i++) rs.caref[i] = -bb * vel[i] - kk * imp * (dist - margin);
  }

  int solver_iter = 0;
  T qacc = qas;
  const int nefc = 20 + 4 * ncon; // frictionloss rows exist on all 20 hinges
  if (nefc > 0) {
    // helpers over a candidate qacc held per lane (value `xa`, also staged in L.xv)
    auto stage = [&](T xa) { __syncthreads(); if (lane < NV) L.xv[lane] = xa; __syncthreads(); };
    auto jar_of = [&](T xa) { // fills rs.fjar / rs.cjar from L.xv
      if (rs.has_f) rs.fjar = xa - rs.faref;
      if (rs.has_c) {
        T o[4];
        contact_rows(m, L, fl, lane, L.xv, mu, o);
#pragma unroll
        for (int i = 0; i < 4; i++) rs.cjar[i] = o[i] - rs.caref[i];
      }
    };
    // warm start: keep qacc_warmstart only if it is cheaper than qacc_smooth
    T xw = lane < NV ? qacc_ws_g[lane] : T(0);
    stage(xw);
    T Ma = mulM(L, L.xv, lane);
    jar_of(xw);
    T cc, gg, hh;
    rows_eval(rs, T(0), cc, gg, hh);
    T cost_w = wave_sum(cc + (lane < NV ? T(0.5) * (Ma - qfs) * (xw - qas) : T(0)));
    // keep the warm start's row residuals while the smooth solution is evaluated: whichever wins, its
    // M a and residuals are already there (same operations on the same data as evaluating the winner again)
    const T fjar_w = rs.fjar;
    T cjar_w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) cjar_w[i] = rs.cjar[i];
    stage(qas);
    jar_of(qas);
    rows_eval(rs, T(0), cc, gg, hh);
    T cost_s = wave_sum(cc);
    if (cost_w > cost_s) {
      qacc = qas;
      Ma = mulM(L, L.xv, lane); // L.xv still holds qacc_smooth
    } else {
      qacc = xw;
      rs.fjar = fjar_w;
#pragma unroll
      for (int i = 0; i < 4; i++) rs.cjar[i] = cjar_w[i];
    }

    TSIDB_STAMP(20);
    const T scale = T(1) / (m.meaninertia * NV);
    T cost = 0;
    int iter = 0;
    TSIDB_LAP_ZERO(24); TSIDB_LAP_ZERO(25); TSIDB_LAP_ZERO(26); TSIDB_LAP_ZERO(27); TSIDB_LAP_ZERO(28);
    TSIDB_LAP_INIT();
    while (true) {
      // ---- constraint state at the current point: forces, active rows, cost
      rows_eval(rs, T(0), cc, gg, hh);
      const T gauss = wave_sum(lane < NV ? T(0.5) * (Ma - qfs) * (qacc - qas) : T(0));
      const T newcost = gauss + wave_sum(cc);
      // force on the friction row and force vector of the contact
      T ff = 0;
      bool fact = false;
      if (rs.has_f) {
        const T f = rs.floss, r = rs.fR;
        if (rs.fjar <= -r * f) ff = f;
        else if (rs.fjar >= r * f) ff = -f;
        else { ff = -rs.fD * rs.fjar; fact = true; }
      }
      T Arow[6] = {0, 0, 0, 0, 0, 0}; // sum_rows D dir dir^T (sym 3x3: xx xy xz yy yz zz)
      if (rs.has_c) {
        T fr[4];
        T fv[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const bool act = rs.cjar[i] < 0;
          fr[i] = act ? -rs.cD * rs.cjar[i] : T(0);
          // direction of row i in world axes: n + s*mu*t_k
          const T sg = (i & 1) ? -mu : mu;
          const T *tk = i < 2 ? fl.t1 : fl.t2;
          const T dir[3] = {fl.n[0] + sg * tk[0], fl.n[1] + sg * tk[1], fl.n[2] + sg * tk[2]};
#pragma unroll
          for (int e = 0; e < 3; e++) fv[e] += fr[i] * dir[e];
          if (act) {
            Arow[0] += rs.cD * dir[0] * dir[0]; Arow[1] += rs.cD * dir[0] * dir[1]; Arow[2] += rs.cD * dir[0] * dir[2];
            Arow[3] += rs.cD * dir[1] * dir[1]; Arow[4] += rs.cD * dir[1] * dir[2]; Arow[5] += rs.cD * dir[2] * dir[2];
          }
        }
        L.cfv[lane][0] = fv[0]; L.cfv[lane][1] = fv[1]; L.cfv[lane][2] = fv[2];
      }
      __syncthreads();
      // ---- gradient: Ma - qfrc_smooth - J^T force
      T grad = 0;
      if (lane < NV) {
        const int k = lane, bk = k < 6 ? 0 : k - 5;
        T s = 0;
        for (int c = 0; c < ncon; c++) {
          if (!((L.anc[L.cbody[c]] >> bk) & 1u)) continue;
          T rxf[3];
          cross3(L.cr[c], L.cfv[c], rxf);
          s += L.S[k][0] * L.cfv[c][0] + L.S[k][1] * L.cfv[c][1] + L.S[k][2] * L.cfv[c][2] +
               L.S[k][3] * rxf[0] + L.S[k][4] * rxf[1] + L.S[k][5] * rxf[2];
        }
        grad = Ma - qfs - s - ff;
      }
      if (iter > 0) {
        const T gn = wave_sum(grad * grad);
        const T improvement = scale * (cost - newcost), gradient = scale * sqrt(gn);
        cost = newcost;
        if (improvement < tol || gradient < tol) break;
      }
      cost = newcost;
      if (iter >= maxiter) break;
      TSIDB_LAP(24);
      // ---- Newton Hessian H = M + J^T D J: CRB recursion on the per-body contact inertia
      if (rs.has_c) {
        // W = [A, -A [r]x ; [r]x A, -[r]x A [r]x] with [r]x the cross matrix of the contact point
        const T *r = L.cr[lane];
        const T A3[9] = {Arow[0], Arow[1], Arow[2], Arow[1], Arow[3], Arow[4], Arow[2], Arow[4], Arow[5]};
        const T X[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
        T XA[9], XAXt[9], Xt[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Xt[3 * i + j] = X[3 * j + i];
        mat3mul(X, A3, XA);
        mat3mul(XA, Xt, XAXt);
        // upper-left A, upper-right (X A)^T, lower-right X A X^T
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = i; j < 3; j++) { L.Wc[lane][sym_idx(i, j)] = A3[3 * i + j]; L.Wc[lane][sym_idx(3 + i, 3 + j)] = XAXt[3 * i + j]; }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) L.Wc[lane][sym_idx(i, 3 + j)] = XA[3 * j + i];
      }
      for (int i = lane; i < NB * 21; i += WAVE) (&L.K[0][0])[i] = 0;
      __syncthreads();
      // composite contact inertia: contacts arrive grouped by body; lane e sums entry e of the group
      // and pushes it to every ancestor of that body (subtree sums without a depth loop)
      unsigned touched = 0;
      {
        T accK = 0;
        for (int c = 0; c < ncon; c++) {
          const int b = L.cbody[c];
          if (lane < 21) accK += L.Wc[c][lane];
          if (c + 1 == ncon || L.cbody[c + 1] != b) {
            const unsigned am = L.anc[b];
            for (unsigned mk = am; mk; mk &= mk - 1) {
              const int a = __ffs(mk) - 1;
              if (lane < 21) L.K[a][lane] += accK;
            }
            touched |= am;
            accK = 0;
          }
        }
      }
      __syncthreads();
      for (int i = lane; i < NV * LDM; i += WAVE) L.H[i] = L.M[i];
      __syncthreads();
      if (lane < NV && ((touched >> (lane < 6 ? 0 : lane - 5)) & 1u)) {
        const int k = lane, bk = k < 6 ? 0 : k - 5;
        T G[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
          T s = 0;
#pragma unroll
          for (int j = 0; j < 6; j++) s += L.K[bk][sym_idx(i, j)] * L.S[k][j];
          G[i] = s;
        }
        for (unsigned mk = dofanc; mk; mk &= mk - 1) {
          const int a = __ffs(mk) - 1;
          const int i0 = a == 0 ? 0 : 5 + a, i1 = a == 0 ? 5 : 5 + a;
          for (int i = i0; i <= i1; i++) {
            if (i > k) continue;
            T val = 0;
#pragma unroll
            for (int e = 0; e < 6; e++) val += L.S[i][e] * G[e];
            L.H[i * LDM + k] += val;
            if (i != k) L.H[k * LDM + i] += val;
          }
        }
      }
      if (fact) L.H[lane * LDM + lane] += rs.fD;
      __syncthreads();
      TSIDB_LAP(25);
#pragma unroll
      for (int j = 0; j < NV; j++) arow[j] = lane < NV ? L.H[lane * LDM + j] : T(0);
      bool ok;
      const T search = -chol26_solve(arow, grad, lane, ok);
      if (!ok) { fail |= 2; break; }
      TSIDB_LAP(26);
      // ---- exact line search along `search`
      stage(search);
      const T Mv = mulM(L, L.xv, lane);
      if (rs.has_f) rs.fJv = search;
      if (rs.has_c) contact_rows(m, L, fl, lane, L.xv, mu, rs.cJv);
      T qg1 = wave_sum(lane < NV ? search * (Ma - qfs) : T(0));
      T qg2 = wave_sum(lane < NV ? T(0.5) * search * Mv : T(0));
      T snorm = sqrt(wave_sum(lane < NV ? search * search : T(0)));
      if (snorm < MINVAL) break;
      const T gtol = tol * ls_tol * snorm * m.meaninertia * NV;
      auto ls_eval = [&](T alpha, T &c, T &d1, T &d2) {
        T lc, lg, lh;
        rows_eval(rs, alpha, lc, lg, lh);
        c = alpha * alpha * qg2 + alpha * qg1 + gauss + wave_sum(lc);
        d1 = 2 * alpha * qg2 + qg1 + wave_sum(lg);
        d2 = 2 * qg2 + wave_sum(lh);
      };
      T c0, g1, g2, ca, alpha = 0, lo = 0, hi = INF;
      ls_eval(T(0), c0, g1, g2);
      ca = c0;
      for (int li = 0; li < ls_iter && fabs(g1) >= gtol; li++) {
        if (g1 < 0) lo = alpha; else hi = alpha;
        T an = alpha - g1 / g2;
        if (!(an > lo) || !(an < hi)) an = hi >= INF ? 2 * alpha + 1 : T(0.5) * (lo + hi);
        alpha = an;
        ls_eval(alpha, ca, g1, g2);
      }
      if (!(ca < c0) || alpha == 0) break;
      qacc += alpha * search;
      Ma += alpha * Mv;
      if (rs.has_f) rs.fjar += alpha * rs.fJv;
      if (rs.has_c) {
#pragma unroll
        for (int i = 0; i < 4; i++) rs.cjar[i] += alpha * rs.cJv[i];
      }
      iter++;
      TSIDB_LAP(27);
    }
    solver_iter = iter;
  }
  __syncthreads();
  TSIDB_STAMP(21);
  // ---- semi-implicit Euler, write back
  if (lane < NV) {
    const T vn = L.qvel[lane] + dt * qacc;
    L.qvel[lane] = vn;
    qvel_g[lane] = vn;
    qacc_ws_g[lane] = qacc;
    if (qacc_out) qacc_out[lane] = qacc;
  }
  __syncthreads();
  if (lane < 3) L.qpos[lane] += dt * L.qvel[lane];
  if (lane >= 6 && lane < NV) L.qpos[lane + 1] += dt * L.qvel[lane];
  if (lane == 3) {
    const T *w = &L.qvel[3];
    const T th = sqrt(dot3(w, w)) * dt;
    T dq[4] = {1, 0, 0, 0};
    if (th > 0) {
      T sh, ch;
      sincos_t(T(0.5) * th, sh, ch);
      const T s = sh * dt / th;
      dq[0] = ch; dq[1] = s * w[0]; dq[2] = s * w[1]; dq[3] = s * w[2];
    }
    const T a[4] = {L.qpos[3], L.qpos[4], L.qpos[5], L.qpos[6]};
    T r[4];
    r[0] = a[0] * dq[0] - a[1] * dq[1] - a[2] * dq[2] - a[3] * dq[3];
    r[1] = a[0] * dq[1] + a[1] * dq[0] + a[2] * dq[3] - a[3] * dq[2];
    r[2] = a[0] * dq[2] - a[1] * dq[3] + a[2] * dq[0] + a[3] * dq[1];
    r[3] = a[0] * dq[3] + a[1] * dq[2] - a[2] * dq[1] + a[3] * dq[0];
    const T nn = T(1) / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) L.qpos[3 + i] = r[i] * nn;
  }
  __syncthreads();
  if (lane < NQ) qpos_g[lane] = L.qpos[lane];
  if (lane == 0 && info) { info[2] = solver_iter; info[3] = fail; }
  TSIDB_STAMP(22);
}

} // namespace tsidb
