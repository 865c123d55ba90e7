// tsidb_tick.hpp - one TSID tick for one env on one 64-lane wavefront.
//
// Replaces, per env, the reference's per-tick calls (main.py:119-129):
//   formulation.computeProblemData -> rbd_terms() + assemble()      (tsid / pinocchio)
//   solver.solve                    -> qp_solve()                     (tsid SolverHQuadProgFast / eiquadprog)
//   getActuatorForces / getAccelerations / integrate_dv / get_cop    -> finish()
//
// MI355X mapping: the env's whole problem lives in LDS + registers of one wavefront; HBM traffic is
// the state/refs in and state/outputs out.  Rigid-body terms use world-aligned spatial vectors about
// the base origin (no per-joint frame changes; keeps magnitudes < 0.5 m for fp32).  The 26x26
// Hessian block is assembled, Cholesky-factorised and inverted entirely in registers with
// v_readlane broadcasts (lane i owns row i); the force blocks' factor is a model constant.  The
// dual active-set iterations keep J in registers too (lane i = row i, fixed column slots); only the small
// triangular factor of the active inequality normals, row values and flags are in LDS; constraint adds use one
// Householder reflection instead of eiquadprog's Givens chain (same iterates, no sequential sqrt/div chain).
#pragma once
#include "tsidb_common.hpp"

namespace tsidb {

constexpr int NAS = 34;   // room for the active inequalities (at most n - p = 32)

// LDS plan of k_tick (20.3 KB in float64 -> 8 workgroups per CU).  Three lifetimes share one region:
//   rigid-body passes:  S, frames, subtree forces/inertias, and per-joint R, p, V, A
//   task assembly:      the frame / CoM Jacobians take over R, p, V, A's space once the frames are known;
//                       the task right-hand sides take over the subtree forces' space
//   active set:         triangular factor, row values, flags take over all of it
template <typename T>
struct KinScratch {
  T S[NV][6], fR[2][9], fp[2][3];
  union {
    T f[NJ][6];
    struct { T arhs[4][6], acomr[3], apost[NA], aamr[3]; }; // a_des - drift: contact LF/RF, foot LF/RF; CoM; posture; AM
  };
  T Yc[NJ][10];
  union {
    struct { T R[NJ][9], p[NJ][3], V[NJ][6], A[NJ][6]; };
    struct { T Jf[12 * LDF], Jcom[3 * LDF], Jam[3 * LDF]; }; // frame Jacobians LOCAL (LF rows 0..5, RF 6..11), CoM
                                                              // Jacobian, centroidal angular-momentum rows
  };
};

template <typename T>
struct ActiveSetLds { // dual active-set bookkeeping; J itself lives in registers
  T Ra[NAS * (NAS + 1) / 2 + 2]; // packed upper-triangular factor of the active inequality normals
  T Rinv[NAS + 2];               // reciprocal diagonal
  T s[160];
  T actp[3][NA];                 // partial sums of the actuation rows [M_a | -J_a^T] x (three column ranges)
  int slot[NAS + 2], A[NAS + 2], Aold[NAS + 2];
  unsigned char cstate[160];     // bit0: in the active set, bit1: excluded for this outer iteration
};

template <typename T>
struct TickLds {
  union {
    KinScratch<T> k;
    ActiveSetLds<T> as;
  };
  T Dyn[NV * LDD]; // row r = [M[r][0:26] | -Jc[:, r]^T]  (rows 0..5: base dynamics, 6..25: actuation)
  T h[NV];
  T x[NVAR];
  T oMf[2][12]; // frame placement: R row-major, p (world)
  T vf[2][6], af[2][6];
  T com[3], vcom[3], acomd[3];
  T Lam[3], dLam[3]; // centroidal angular momentum and its drift (angular-momentum task)
  T qs[NQ], vs[NV];
  int act[2]; // contact flags of this tick (LF, RF): read where needed - kept in registers from the top of the tick they are
              // long-lived values the QP phases have no room for (they were spilled to scratch)
};
static_assert(sizeof(ActiveSetLds<double>) <= sizeof(KinScratch<double>), "active-set state must fit the scratch region");
static_assert(sizeof(TickLds<double>) <= 20480, "k_tick must fit 8 workgroups per CU");

template <typename T> __device__ __forceinline__ T bcast(T v, int src) { return __shfl(v, src, WAVE); }

// lane-uniform readlane for float / double (src must be wave-uniform; constant after unrolling)
__device__ __forceinline__ float rdlane(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ double rdlane(double v, int src) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}


// --------------------------------------------------------------------------- rigid-body terms
// Inputs L.qs, L.vs.  Outputs: L.Dyn (M part, rest zero), L.h, L.Jf, L.Jcom, L.oMf, L.vf, L.af,
// L.com, L.vcom, L.acomd.  Spatial vectors are [lin; ang] in world axes about the base origin O.
template <typename T>
__device__ __forceinline__ void rbd_terms(const DevModel<T> &m, TickLds<T> &L, int lane) {
  KinScratch<T> &K = L.k;
  const T GZ = T(9.81);
  TSIDB_LAP_ZERO(15); TSIDB_LAP_ZERO(23); TSIDB_LAP_ZERO(29); TSIDB_LAP_ZERO(30); TSIDB_LAP_ZERO(31);
  TSIDB_LAP_INIT();
  for (int i = lane; i < NV * LDD; i += WAVE) L.Dyn[i] = 0;
  // ---- forward pass (tree_forward: pointer jumping for the transforms, ancestor sums for V and A);
  //      body inertias and forces follow in one parallel pass.
  T Rj[9], pj[3], Vj[6], Aj[6], Sj[6], qd = 0;
  // topology into registers up front: no dependent global loads inside the tree passes
  const int up0 = lane < NJ ? m.pin_up[0][lane] : -1, up1 = lane < NJ ? m.pin_up[1][lane] : -1,
            up2 = lane < NJ ? m.pin_up[2][lane] : -1;
  int jchn[7];
#pragma unroll
  for (int d = 0; d < 7; d++) jchn[d] = lane < NJ ? m.pin_chain[lane][d] : -1;
  // (the frames' parent joints and their ancestor masks too: a dependent pair of loads, needed after the tree passes)
  const int fpar0 = m.frame_parent[0], fpar1 = m.frame_parent[1];
  const unsigned fanc0 = m.pin_anc[fpar0], fanc1 = m.pin_anc[fpar1];
#pragma unroll
  for (int i = 0; i < 6; i++) { Vj[i] = 0; Aj[i] = 0; Sj[i] = 0; }
#pragma unroll
  for (int i = 0; i < 3; i++) pj[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) Rj[i] = 0;
  if (lane == 0) {
    quat_to_R(L.qs[3], L.qs[4], L.qs[5], L.qs[6], Rj);
    mat3vec(Rj, &L.vs[0], Vj);
    mat3vec(Rj, &L.vs[3], Vj + 3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        K.S[k][i] = Rj[3 * i + k]; K.S[k][3 + i] = 0;
        K.S[3 + k][i] = 0; K.S[3 + k][3 + i] = Rj[3 * i + k];
      }
    }
  } else if (lane < NJ) {
    // the joint's transform in its parent: placement rotation times Rz(theta), placement offset
    const T *PR = m.pin_place[lane];
    const T th = L.qs[6 + lane];
    T c, s;
    sincos_t(th, s, c);
    qd = L.vs[5 + lane];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Rj[3 * i + 0] = c * PR[3 * i] + s * PR[3 * i + 1];
      Rj[3 * i + 1] = -s * PR[3 * i] + c * PR[3 * i + 1];
      Rj[3 * i + 2] = PR[3 * i + 2];
      pj[i] = PR[9 + i];
    }
  }
  TSIDB_LAP(15);
  tree_forward<T>(lane, NJ, up0, up1, up2, jchn, &K.R[0][0], &K.V[0][0], &K.f[0][0], 6, &K.Yc[0][0], 10, Rj, pj, qd, Sj,
                  Vj, Aj);
  if (lane < NJ) {
    const int j = lane;
    if (j > 0) {
#pragma unroll
      for (int i = 0; i < 6; i++) K.S[5 + j][i] = Sj[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) K.R[j][i] = Rj[i];
#pragma unroll
    for (int i = 0; i < 3; i++) K.p[j][i] = pj[i];
#pragma unroll
    for (int i = 0; i < 6; i++) { K.V[j][i] = Vj[i]; K.A[j][i] = Aj[i]; }
  }
  TSIDB_LAP(23);
  {
    // body inertia about O in world axes, RNEA body force (gravity as +g base acceleration); then the
    // subtree sums (composite inertias, subtree forces) as prefix-scan differences over the lanes
    T fb[6] = {0, 0, 0, 0, 0, 0}, Y[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < NJ) {
      const int j = lane;
      const T *Yb = m.pin_inertia[j];
      T cw[3], I[9] = {Yb[4], Yb[5], Yb[6], Yb[5], Yb[7], Yb[8], Yb[6], Yb[8], Yb[9]}, Tm[9], RT[9];
      mat3vec(Rj, Yb + 1, cw);
#pragma unroll
      for (int i = 0; i < 3; i++) cw[i] += pj[i];
      mat3mul(Rj, I, Tm);
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) RT[3 * i + k] = Rj[3 * k + i];
      mat3mul(Tm, RT, I);
      T mass = Yb[0], c2 = dot3(cw, cw);
      Y[0] = mass; Y[1] = mass * cw[0]; Y[2] = mass * cw[1]; Y[3] = mass * cw[2];
      Y[4] = I[0] + mass * (c2 - cw[0] * cw[0]); Y[5] = I[1] - mass * cw[0] * cw[1]; Y[6] = I[2] - mass * cw[0] * cw[2];
      Y[7] = I[4] + mass * (c2 - cw[1] * cw[1]); Y[8] = I[5] - mass * cw[1] * cw[2];
      Y[9] = I[8] + mass * (c2 - cw[2] * cw[2]);
      T Ag[6] = {Aj[0], Aj[1], Aj[2] + GZ, Aj[3], Aj[4], Aj[5]}, Ya[6], Yv[6], vx[6];
      yo_mul(Y, Ag, Ya);
      yo_mul(Y, Vj, Yv);
      cross_mf(Vj, Yv, vx);
#pragma unroll
      for (int i = 0; i < 6; i++) fb[i] = Ya[i] + vx[i];
    }
    TSIDB_LAP(29);
    const int mylast = lane < NJ ? m.pin_last[lane] : lane;
#pragma unroll
    for (int i = 0; i < 6; i++) fb[i] = subtree_sum32(fb[i], mylast);
#pragma unroll
    for (int i = 0; i < 10; i++) Y[i] = subtree_sum32(Y[i], mylast);
    if (lane < NJ) {
#pragma unroll
      for (int i = 0; i < 6; i++) K.f[lane][i] = fb[i];
#pragma unroll
      for (int i = 0; i < 10; i++) K.Yc[lane][i] = Y[i];
    }
  }
  TSIDB_SYNC1();
  TSIDB_LAP(30);
  // ---- per dof: bias, F = Yc S, mass-matrix entries, CoM Jacobian column
  const T invm = T(1) / m.mass;
  T jc[3] = {0, 0, 0}, ja[3] = {0, 0, 0};
  if (lane < NV) {
    const int k = lane, jk = k < 6 ? 0 : k - 5;
    T Sk[6], Fk[6];
#pragma unroll
    for (int i = 0; i < 6; i++) Sk[i] = K.S[k][i];
    T hk = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) hk += Sk[i] * K.f[jk][i];
    L.h[k] = hk;
    yo_mul(K.Yc[jk], Sk, Fk);
    jc[0] = Fk[0] * invm; jc[1] = Fk[1] * invm; jc[2] = Fk[2] * invm; // CoM Jacobian column (written below)
    // centroidal angular-momentum column: the subtree's momentum about the CoM, n_O - c x f
    const T cx = K.Yc[0][1] * invm, cy = K.Yc[0][2] * invm, cz = K.Yc[0][3] * invm;
    ja[0] = Fk[3] - (cy * Fk[2] - cz * Fk[1]);
    ja[1] = Fk[4] - (cz * Fk[0] - cx * Fk[2]);
    ja[2] = Fk[5] - (cx * Fk[1] - cy * Fk[0]);
    // M[i][k] = S_i . F_k for every dof i on the path root..k: a static walk over the chain table (the joint's non-root
    // ancestors, then the root's six dofs) - every load is independent and issued up front; the bit-mask loop it replaces
    // waited out one LDS round trip per ancestor
    int chn[7];
#pragma unroll
    for (int d = 0; d < 7; d++) chn[d] = m.pin_chain[jk][d];
#pragma unroll
    for (int d = 0; d < 7 + 6; d++) {
      const int i = d < 7 ? (chn[d < 7 ? d : 0] > 0 ? 5 + chn[d < 7 ? d : 0] : -1) : d - 7;
      if (i < 0 || i > k) continue;
      T val = 0;
#pragma unroll
      for (int e = 0; e < 6; e++) val += K.S[i][e] * Fk[e];
      L.Dyn[i * LDD + k] = val;
      L.Dyn[k * LDD + i] = val;
    }
  }
  TSIDB_LAP(31);
  // ---- frames (lanes 0,1): placement, velocity, classical drift acceleration (LOCAL)
  if (lane < 2) {
    const int f = lane, jf = f == 0 ? fpar0 : fpar1;
    const T *P = m.frame_place[f];
    T Rf[9], pf[3];
    mat3mul(K.R[jf], P, Rf);
    mat3vec(K.R[jf], P + 9, pf);
#pragma unroll
    for (int i = 0; i < 3; i++) pf[i] += K.p[jf][i];
    const T *V = K.V[jf], *A = K.A[jf];
    T wxp[3], vP[3], axp[3], wxv[3], aP[3];
    cross3(V + 3, pf, wxp);
#pragma unroll
    for (int i = 0; i < 3; i++) vP[i] = V[i] + wxp[i];
    cross3(A + 3, pf, axp);
    cross3(V + 3, vP, wxv);
#pragma unroll
    for (int i = 0; i < 3; i++) aP[i] = A[i] + axp[i] + wxv[i];
    mat3Tvec(Rf, vP, L.vf[f]);
    mat3Tvec(Rf, V + 3, L.vf[f] + 3);
    mat3Tvec(Rf, aP, L.af[f]);
    mat3Tvec(Rf, A + 3, L.af[f] + 3);
#pragma unroll
    for (int i = 0; i < 9; i++) { L.oMf[f][i] = Rf[i]; K.fR[f][i] = Rf[i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) { L.oMf[f][9 + i] = pf[i] + L.qs[i]; K.fp[f][i] = pf[i]; }
  }
  if (lane == 2) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      L.com[i] = K.Yc[0][1 + i] * invm + L.qs[i];
      L.acomd[i] = K.f[0][i] * invm - (i == 2 ? GZ : T(0));
    }
  }
  if (lane == 3) { // rate of the centroidal angular momentum at zero acceleration (gravity has no moment about the CoM)
    const T cx = K.Yc[0][1] * invm, cy = K.Yc[0][2] * invm, cz = K.Yc[0][3] * invm;
    const T *f0 = K.f[0];
    L.dLam[0] = f0[3] - (cy * f0[2] - cz * f0[1]);
    L.dLam[1] = f0[4] - (cz * f0[0] - cx * f0[2]);
    L.dLam[2] = f0[5] - (cx * f0[1] - cy * f0[0]);
  }
  TSIDB_SYNC1();
  // ---- frame Jacobian columns (LOCAL) and CoM velocity
  if (lane < NV) {
    const int k = lane, jk = k < 6 ? 0 : k - 5;
#pragma unroll
    for (int f = 0; f < 2; f++) {
      T col[6] = {0, 0, 0, 0, 0, 0};
      if (((f == 0 ? fanc0 : fanc1) >> jk) & 1u) {
        T wxp[3], lin[3];
        cross3(&K.S[k][3], K.fp[f], wxp);
#pragma unroll
        for (int i = 0; i < 3; i++) lin[i] = K.S[k][i] + wxp[i];
        mat3Tvec(K.fR[f], lin, col);
        mat3Tvec(K.fR[f], &K.S[k][3], col + 3);
      }
#pragma unroll
      for (int i = 0; i < 6; i++) K.Jf[(6 * f + i) * LDF + k] = col[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { K.Jcom[i * LDF + k] = jc[i]; K.Jam[i * LDF + k] = ja[i]; }
  }
  {
    T vk = lane < NV ? L.vs[lane] : T(0);
#pragma unroll
    for (int i = 0; i < 3; i++) {
      T tot = wave_sum(lane < NV ? jc[i] * vk : T(0));
      if (lane == 0) L.vcom[i] = tot;
    }
    if (m.params[P_W_AM] != 0) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        T tot = wave_sum(lane < NV ? ja[i] * vk : T(0));
        if (lane == 0) L.Lam[i] = tot;
      }
    }
  }
  TSIDB_SYNC1();
}

// --------------------------------------------------------------------------- task right-hand sides
// TaskSE3Equality in the local frame: kp*log6(M^-1 Mref) + kd*(R^T vref - v) + R^T aref - drift
template <typename T>
__device__ __forceinline__ void se3_rhs(const TickLds<T> &L, int f, const T *ref, int nref, T kp, T kd, T *rhs) {
  const T *R = L.oMf[f], *p = L.oMf[f] + 9;
  T rel[9], d[3], pr[3], err[6];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      T s = 0;
#pragma unroll
      for (int k = 0; k < 3; k++) s += R[3 * k + i] * ref[3 + 3 * j + k]; // ref rotation is column-major
      rel[3 * i + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 3; i++) d[i] = ref[i] - p[i];
  mat3Tvec(R, d, pr);
  log6(rel, pr, err);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    T vr[3] = {0, 0, 0}, ar[3] = {0, 0, 0};
    if (nref >= 24) {
      mat3Tvec(R, ref + 12 + 3 * h, vr);
      mat3Tvec(R, ref + 18 + 3 * h, ar);
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
      rhs[3 * h + i] = kp * err[3 * h + i] + kd * (vr[i] - L.vf[f][3 * h + i]) + ar[i] - L.af[f][3 * h + i];
  }
}

#define ORDER_PIN2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#define ORDER_PIN4(x, y, z, w) asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)) // (two steps at a time: hipcc pads an
// asm statement's outputs with one wait state before the next VALU that touches them, so a pin is not entirely free)

// --------------------------------------------------------------------------- QP pieces
__device__ __forceinline__ int rcol(int c) { return c * (c + 1) / 2; } // packed upper-triangular column start

template <typename T>
struct QpCtx {
  int n, nslot, nin, p, iq;
  T R_norm;
};
// foot carried by contact slot s of a tick with NS feet in contact (slots are LF, RF in that order)
template <typename T, int NS> __device__ __forceinline__ int slot_foot(const TickLds<T> &L, int s) {
  if constexpr (NS == 2) return s;
  else if constexpr (NS == 1) return L.act[0] ? 0 : 1;
  else return -1;
}

// One-sided inequality rows (CI x + ci0 >= 0), ordered as tsid stacks them: per contact slot 17 cone
// lower + 17 cone upper rows, then 20 + 20 actuation rows, then 26 + 26 acceleration-bound rows.
// Each lane decodes its (up to three) rows ONCE into registers; a sweep then only touches x.
//   s = cst + sgn * val ;  kind 0: val = cf . x[idx..idx+2]         (friction pyramid row)
//                          kind 1: val = sum_pt cf . x[idx+3pt ..]   (normal-force row)
//                          kind 2: val = [M_a | -J_a^T]_idx . x      (actuation row)
//                          kind 3: val = x[idx]                      (acceleration bound)
template <typename T>
struct RowDesc {
  int kind, idx;
  T sgn, cst, cf[3];
};

template <typename T>
__device__ __forceinline__ RowDesc<T> row_desc(const DevModel<T> &m, const TickLds<T> &L, const QpCtx<T> &c, int r) {
  RowDesc<T> d;
  d.kind = -1; d.idx = 0; d.sgn = 0; d.cst = 0; d.cf[0] = d.cf[1] = d.cf[2] = 0;
  if (r >= c.nin) return d;
  const int nc = 34 * c.nslot;
  if (r < nc) {
    const int s = r / 34, rr = r % 34, up = rr >= 17, b = rr % 17;
    d.sgn = up ? T(-1) : T(1);
    d.cst = up ? m.cone_ub[b] : -m.cone_lb[b];
    if (b < 16) {
      d.kind = 0; d.idx = NV + 12 * s + 3 * (b >> 2);
#pragma unroll
      for (int e = 0; e < 3; e++) d.cf[e] = m.Bcone[b][3 * (b >> 2) + e];
    } else {
      d.kind = 1; d.idx = NV + 12 * s;
#pragma unroll
      for (int e = 0; e < 3; e++) d.cf[e] = m.Bcone[16][e];
    }
    return d;
  }
  r -= nc;
  if (r < 2 * NA) {
    const int up = r >= NA, j = r % NA;
    const T tm = m.params[P_TAU_MAX + j], hj = L.h[6 + j];
    d.kind = 2; d.idx = j;
    d.sgn = up ? T(-1) : T(1);
    d.cst = up ? tm - hj : tm + hj;
    return d;
  }
  r -= 2 * NA;
  const int up = r >= NV, k = r % NV;
  T lo = T(-1e10), hi = T(1e10);
  if (k >= 6) {
    const T dt2 = 2 * m.params[P_DT], vmax = m.params[P_V_MAX + k - 6], vk = L.vs[k];
    const T amax = (vmax - vk) / dt2, amin = (-vmax - vk) / dt2;
    hi = amax < T(1e10) ? amax : T(1e10);
    lo = amin > T(-1e10) ? amin : T(-1e10);
  }
  d.kind = 3; d.idx = k;
  d.sgn = up ? T(-1) : T(1);
  d.cst = up ? hi : -lo;
  return d;
}

// partial sums of the actuation rows at L.x: lane = row + 20 * part, part = one third of the columns
template <typename T>
__device__ __forceinline__ void act_partials(TickLds<T> &L, int n, int lane) {
  if (lane < 3 * NA) {
    const int j = lane % NA, part = lane / NA, e0 = part * 17, e1 = e0 + 17 < n ? e0 + 17 : n;
    T a0 = 0, a1 = 0;
    int e = e0;
    for (; e + 1 < e1; e += 2) {
      a0 += L.Dyn[(6 + j) * LDD + e] * L.x[e];
      a1 += L.Dyn[(6 + j) * LDD + e + 1] * L.x[e + 1];
    }
    if (e < e1) a0 += L.Dyn[(6 + j) * LDD + e] * L.x[e];
    L.as.actp[part][j] = a0 + a1;
  }
}

template <typename T>
__device__ __forceinline__ T row_eval(const RowDesc<T> &d, const TickLds<T> &L) {
  T val = 0;
  if (d.kind == 0) {
    val = d.cf[0] * L.x[d.idx] + d.cf[1] * L.x[d.idx + 1] + d.cf[2] * L.x[d.idx + 2];
  } else if (d.kind == 1) {
#pragma unroll
    for (int pt = 0; pt < 4; pt++)
      val += d.cf[0] * L.x[d.idx + 3 * pt] + d.cf[1] * L.x[d.idx + 3 * pt + 1] + d.cf[2] * L.x[d.idx + 3 * pt + 2];
  } else if (d.kind == 2) {
    val = L.as.actp[0][d.idx] + L.as.actp[1][d.idx] + L.as.actp[2][d.idx];
  } else if (d.kind == 3) {
    val = L.x[d.idx];
  }
  return d.cst + d.sgn * val;
}

// v_readlane with a wave-uniform but run-time lane index
__device__ __forceinline__ float rdlane_dyn(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane(src)));
}
__device__ __forceinline__ double rdlane_dyn(double v, int src) {
  const int sl = __builtin_amdgcn_readfirstlane(src);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), sl), hi = __builtin_amdgcn_readlane((int)(b >> 32), sl);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// jr[cidx] for a wave-uniform run-time column index (registers cannot be indexed dynamically)
template <typename T, int PP, int NN> __device__ __forceinline__ T col_select(const T (&jr)[NN], int cidx) {
  T r = 0;
#pragma unroll
  for (int j = PP; j < NN; j++) r = (j == cidx) ? jr[j] : r;
  return r;
}

// Goldfarb-Idnani dual active set (eiquadprog-fast's control flow and tie-breaking) with the factor
// J = L^-T Q held in REGISTERS: lane i owns row i (jr[0..49]).  Columns keep fixed register slots; a
// 64-bit mask marks the free ones (initially p..n-1) and slot[k] names the column that carries the
// k-th active inequality, so that no register is ever indexed by a run-time value.  A constraint add
// is one Householder reflection of the free columns onto the lowest free slot.  Cross-lane pieces use
// v_readlane (sparse normals) or DPP wave sums (dense normals).  Only the small triangular factor of
// the active inequality normals, the row values and flags live in LDS.  The multipliers of the
// equality constraints are not tracked (nothing downstream reads them).
template <typename T, int NS>
__device__ __forceinline__ int qp_active_regs(const DevModel<T> &m, TickLds<T> &L, QpCtx<T> &c, int lane, T (&jr)[NV + 12 * NS],
                                              T &xl, const RowDesc<T> (&rd)[3], T c1, T c2, int max_iter, int &iter_out) {
  ActiveSetLds<T> &S = L.as;
  constexpr int NN = NV + 12 * NS, PP = 6 + 6 * NS;
  const int n = NN, p = PP;
  const T INF = Eps<T>::inf;
  unsigned long long freem = ((1ull << n) - 1ull) & ~((1ull << p) - 1ull);
  int na = 0;
  T ul = 0; // lane k < na: multiplier of the k-th active inequality; lane na: the candidate's
  T R_norm = c.R_norm;
  for (int r = lane; r < c.nin; r += WAVE) S.cstate[r] = 0;
  TSIDB_SYNC1();

  int iter = 0, status = -1;
  TSIDB_LAP_ZERO(10); TSIDB_LAP_ZERO(11); TSIDB_LAP_ZERO(12); TSIDB_LAP_ZERO(13); TSIDB_LAP_ZERO(14);
  TSIDB_LAP_INIT();
  while (status < 0) {
    // ---------------- l1: new outer iteration
    iter++;
    if (iter >= max_iter) { status = 3; break; }
    TSIDB_SYNC1();
    if (lane < n) L.x[lane] = xl;
    TSIDB_SYNC1();
    act_partials(L, n, lane);
    TSIDB_SYNC1();
    T psi = 0;
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
      const int r = lane + WAVE * rr;
      if (r < c.nin) {
        const T sv = row_eval(rd[rr], L);
        S.s[r] = sv;
        S.cstate[r] &= 1;
        psi += sv < 0 ? sv : T(0);
      }
    }
    psi = wave_sum(psi);
    if (fabs(psi) <= T(c.nin) * T(2.220446049250313e-16) * c1 * c2 * T(100)) { status = 0; break; }
    const T xold = xl, uold = ul;
    if (lane < na) S.Aold[lane] = S.A[lane];
    TSIDB_SYNC1();
    TSIDB_LAP(10);

    bool outer_done = false;
    while (!outer_done && status < 0) {
      // ---------------- l2: most violated eligible constraint (lowest index among equals)
      T best = 0;
      int ip = 0x7fffffff;
      for (int r = lane; r < c.nin; r += WAVE) {
        const T sv = S.s[r];
        if (S.cstate[r] == 0 && sv < best) { best = sv; ip = r; }
      }
      wave_argmin(best, ip);
      if (!(best < 0)) { status = 0; break; }
      if (lane == na) ul = 0;
      if (lane == 0) S.A[na] = ip;
      // structure of the normal of row ip, broadcast from the lane that owns the row's descriptor:
      // up to three nonzeros (pyramid / bound rows) or dense (normal-force / actuation rows)
      const int owner = ip & (WAVE - 1), oround = ip >> 6;
      int okind = 0, oidx = 0;
      T osgn = 0, ocf[3] = {0, 0, 0};
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
        if (rr == oround) {
          okind = __builtin_amdgcn_readlane(rd[rr].kind, __builtin_amdgcn_readfirstlane(owner));
          oidx = __builtin_amdgcn_readlane(rd[rr].idx, __builtin_amdgcn_readfirstlane(owner));
          osgn = rdlane_dyn(rd[rr].sgn, owner);
#pragma unroll
          for (int e = 0; e < 3; e++) ocf[e] = rdlane_dyn(rd[rr].cf[e], owner);
        }
      const bool sparse = okind == 0 || okind == 3;
      int si[3] = {0, 0, 0};
      T sc[3] = {0, 0, 0};
      T npl = 0; // dense case: this lane's coefficient
      if (okind == 0) {
#pragma unroll
        for (int e = 0; e < 3; e++) { si[e] = oidx + e; sc[e] = osgn * ocf[e]; }
      } else if (okind == 3) {
        si[0] = oidx; sc[0] = osgn;
      } else if (okind == 1) {
        if (lane >= oidx && lane < oidx + 12) {
          const int e = (lane - oidx) % 3;
          npl = osgn * (e == 0 ? ocf[0] : (e == 1 ? ocf[1] : ocf[2]));
        }
      } else {
        if (lane < n) npl = osgn * L.Dyn[(6 + oidx) * LDD + lane];
      }

      TSIDB_LAP(11);
      while (true) {
        // ---------------- l2a: d = J^T np over the column slots (lane j keeps d_j)
        T dl = 0;
        if (sparse) {
#pragma unroll
          for (int j = PP; j < NN; j++) {
            const T tmp = sc[0] * rdlane_dyn(jr[j], si[0]) + sc[1] * rdlane_dyn(jr[j], si[1]) + sc[2] * rdlane_dyn(jr[j], si[2]);
            if (lane == j) dl = tmp;
          }
        } else {
#pragma unroll
          for (int j = PP; j < NN; j++) {
            const T tmp = wave_sum(npl * jr[j]);
            if (lane == j) dl = tmp;
          }
        }
        TSIDB_LAP(12);
        // Householder: fold the free columns' part of d onto the lowest free slot
        const int cstar = freem ? __ffsll((long long)freem) - 1 : -1;
        T z = 0, zz = 0, znp = 0, dnew = 0;
        if (cstar >= 0) {
          const T dfree = ((freem >> lane) & 1ull) ? dl : T(0);
          const T sigma = wave_sum(lane != cstar ? dfree * dfree : T(0));
          const T alpha = rdlane_dyn(dl, cstar);
          dnew = alpha;
          if (sigma > 0) {
            const T nrm = sqrt(alpha * alpha + sigma);
            const T v0 = alpha + (alpha >= 0 ? nrm : -nrm);
            const T beta = T(2) / (v0 * v0 + sigma);
            T vl = lane == cstar ? v0 : dfree;
            T w = 0;
#pragma unroll
            for (int j = PP; j < NN; j++) w += jr[j] * rdlane(vl, j);
            w *= beta;
            asm volatile("" : "+v"(vl)); // broadcast again rather than keep NN - PP SGPR pairs (spilled to VGPR lanes) alive
#pragma unroll
            for (int j = PP; j < NN; j++) jr[j] -= w * rdlane(vl, j);
            dnew = alpha >= 0 ? -nrm : nrm;
          }
          z = lane < n ? col_select<T, PP, NN>(jr, cstar) * dnew : T(0);
          zz = wave_sum(z * z);
          znp = dnew * dnew;
        }
        TSIDB_LAP(13);
        // r = R^-1 d over the active inequalities (lane k <-> k-th active constraint)
        const int myslot = lane < na ? S.slot[lane] : 0;
        const T dk = __shfl(dl, myslot, WAVE);
        T rl = lane < na ? dk : T(0);
        for (int j = na - 1; j >= 0; j--) {
          const T rj = bcast(rl, j) * S.Rinv[j];
          if (lane == j) rl = rj;
          else if (lane < j) rl -= S.Ra[rcol(j) + lane] * rj;
        }
        // ---------------- step lengths
        T t1 = INF;
        int kmin = 0x7fffffff;
        if (lane < na && rl > 0) { t1 = ul / rl; kmin = lane; }
        wave_argmin(t1, kmin);
        const int l = kmin < na ? S.A[kmin] : 0;
        const T sip = S.s[ip];
        const T t2 = fabs(zz) > Eps<T>::v ? -sip / znp : INF;
        const T t = t1 < t2 ? t1 : t2;
        if (t >= INF) { status = 1; break; }
        bool drop = false;
        if (t2 >= INF) { // dual step only
          if (lane < na) ul -= t * rl;
          if (lane == na) ul += t;
          drop = true;
        } else {
          xl += t * z;
          if (lane < na) ul -= t * rl;
          if (lane == na) ul += t;
          if (t == t2) { // full step: ip joins the active set on slot cstar
            const T ad = fabs(dnew);
            if (na >= NAS - 2 || ad <= Eps<T>::v * R_norm) {
              // numerically dependent on the active set: exclude it for this outer iteration and
              // fall back to the state saved at l1 (eiquadprog's recovery path)
              TSIDB_SYNC1();
              if (lane == 0) S.cstate[ip] |= 2;
              for (int r = lane; r < c.nin; r += WAVE) S.cstate[r] &= 2;
              TSIDB_SYNC1();
              if (lane < na) { S.A[lane] = S.Aold[lane]; S.cstate[S.Aold[lane]] |= 1; }
              ul = lane < na ? uold : T(0);
              xl = xold;
              TSIDB_SYNC1();
              break; // back to l2
            }
            if (ad > R_norm) R_norm = ad;
            if (lane < na) S.Ra[rcol(na) + lane] = dk;
            if (lane == 0) { S.Ra[rcol(na) + na] = dnew; S.Rinv[na] = T(1) / dnew; S.slot[na] = cstar; S.cstate[ip] |= 1; }
            freem &= ~(1ull << cstar);
            na++;
            TSIDB_SYNC1();
            TSIDB_LAP(14);
            outer_done = true;
            break; // back to l1
          }
          drop = true; // partial step
        }
        if (drop) {
          // ---------------- remove active constraint l, then recompute the direction for ip
          TSIDB_SYNC1();
          if (lane == 0) S.cstate[l] &= ~1;
          const int qq = wave_min_int((lane < na && S.A[lane] == l) ? lane : 9999);
          for (int j = qq; j < na - 1; j++) { // Givens sweep restoring the triangular factor
            const int co = j + 1;
            T cc = S.Ra[rcol(co) + j], ss = S.Ra[rcol(co) + j + 1];
            const T h = hypot(cc, ss);
            if (h == 0) continue;
            cc /= h; ss /= h;
            T diag = h;
            if (cc < 0) { cc = -cc; ss = -ss; diag = -h; }
            const T xny = ss / (T(1) + cc);
            TSIDB_SYNC1();
            for (int k = co + 1 + lane; k < na; k += WAVE) {
              const T a1 = S.Ra[rcol(k) + j], a2 = S.Ra[rcol(k) + j + 1];
              const T n1 = a1 * cc + a2 * ss;
              S.Ra[rcol(k) + j] = n1;
              S.Ra[rcol(k) + j + 1] = xny * (a1 + n1) - a2;
            }
            if (lane == 0) { S.Ra[rcol(co) + j] = diag; S.Ra[rcol(co) + j + 1] = 0; }
            const int sa = S.slot[j], sb = S.slot[j + 1];
            const T ta = col_select<T, PP, NN>(jr, sa), tb = col_select<T, PP, NN>(jr, sb);
            const T n1 = ta * cc + tb * ss, n2 = xny * (n1 + ta) - tb;
#pragma unroll
            for (int jj = PP; jj < NN; jj++) jr[jj] = (jj == sa) ? n1 : ((jj == sb) ? n2 : jr[jj]);
            TSIDB_SYNC1();
          }
          for (int k = qq + 1; k < na; k++) { // shift the packed columns left
            const T val = lane < k ? S.Ra[rcol(k) + lane] : T(0);
            TSIDB_SYNC1();
            if (lane < k) S.Ra[rcol(k - 1) + lane] = val;
            TSIDB_SYNC1();
          }
          if (lane >= qq && lane < na - 1) S.Rinv[lane] = T(1) / S.Ra[rcol(lane) + lane];
          int Av = 0;
          if (lane >= qq && lane < na) Av = S.A[lane + 1];
          const T unext = __shfl_down(ul, 1, WAVE);
          TSIDB_SYNC1();
          if (lane >= qq && lane < na) S.A[lane] = Av;
          if (lane >= qq && lane <= na) ul = unext;
          const int freed = S.slot[na - 1];
          freem |= 1ull << freed;
          na--;
          if (lane == na + 1) ul = 0;
          // refresh s[ip] at the new point
          TSIDB_SYNC1();
          if (lane < n) L.x[lane] = xl;
          TSIDB_SYNC1();
          if (okind == 2) { act_partials(L, n, lane); TSIDB_SYNC1(); }
#pragma unroll
          for (int rr = 0; rr < 3; rr++)
            if (rr == oround && lane == owner) S.s[ip] = row_eval(rd[rr], L);
          TSIDB_SYNC1();
        }
      }
    }
  }
  TSIDB_SYNC1();
  if (lane < n) L.x[lane] = xl;
  TSIDB_SYNC1();
  c.iq = p + na;
  iter_out = iter;
  return status;
}

// QP of one contact configuration (NS = number of feet in contact: variables 26 + 12 NS, equalities
// 6 + 6 NS) from the Cholesky factor of the dv block onwards; sized at compile time so that the
// register-resident rows/columns and every unrolled loop carry no padding for absent contacts.
template <typename T, int NS, bool COP>
__device__ __forceinline__ void tick_qp(const DevModel<T> &m, TickLds<T> &L, QpCtx<T> &c, int lane, const T (&a)[NV],
                                        T rdv, T gi, T c1, bool spd, const T *cop_ref, int &qp_status, int &qp_iters, bool fast_eq) {
  // ORDER_PIN2(x, y): an empty volatile asm that "modifies" x and y.  The unrolled loops below use every broadcast value
  // (v_readlane -> an SGPR pair) twice, in two FMA chains; the compiler likes to run one chain for the whole loop and
  // then the other, keeping all the loop's broadcast values alive in between - ~100 SGPRs it does not have, so it spills
  // each to a VGPR lane (v_writelane + wait states) and reloads it (v_readlane): 3 800 spill sites in this kernel, whose
  // spill VGPRs in turn pushed the vector registers over budget.  Volatile asms keep their order, so pinning both
  // accumulators after each step keeps the two uses adjacent.  Costs no instruction (one wait state at most).
  // ---- three forward substitutions with L share its broadcast entries:
  //   y  = L^-1 (-g)                      (uniform, lane i contributes y_i)
  //   xr = L^-1 e_lane                    (column `lane` of L^-1 = row `lane` of J0 = L^-T)
  //   bc = L^-1 CE[lane][0:26]^T          (column `lane` of B = J0^T CE^T, lanes < p)
  constexpr int NN = NV + 12 * NS, PP = 6 + 6 * NS; // variables / equality constraints of this contact configuration
  constexpr int NC = 6 * NS;                        // contact-motion equalities (they come first, see below)
  const int p = PP, n = NN;
  // B = J^T CE^T (NN x PP) is held by COLUMN, each column split over G lanes: lane c + PP g keeps rows
  // [g GS, (g + 1) GS) of column c in bs[0..GS).  (One whole column per lane - the first version - costs NN doubles in
  // every lane for PP useful lanes: with J's row of NN doubles beside it the double-support body spilled ~130 VGPRs.)
  // (Measured and dropped, round 2: with <= 16 equalities the columns fit one 16-lane DPP row; every row of the wavefront
  //  keeping a full copy, a reflector entry reaches all lanes with one v_mov_b64_dpp row_newbcast instead of a
  //  v_readlane pair - but a full column per lane is 76 VGPRs again, the single-support body then spills 104, and
  //  k_tick went from 0.154 to 0.174 ms on the walking workload.)
  constexpr int G = 3, GS = (NN + G - 1) / G;
  static_assert(G * PP <= WAVE, "the column groups of B must fit the wavefront");
  // Fast equality solve (round 4, float64, reference stack): most ticks end with NO active inequality, and for those the
  // equality-constrained optimum is all that is needed.  It is x0 + J0 z with z = -B (B^T B)^-1 c, the least-norm solution
  // of B^T z + c = 0 - the same point the QR below reaches, by a PP x PP Cholesky instead of PP reflectors applied to the
  // NN x NN factor J (a fifth of the tick).  Only when the feasibility sweep at that point fails does the env go on to the
  // QR and the dual active-set iterations, exactly as before (B and J0 are untouched).  Scratch in the dead kinematics region
  // of LDS: B's dv rows [NV][PP] at its start, B^T B packed behind them, both in front of B's force rows.
  constexpr bool FASTEQ = sizeof(T) == 8 && !COP;
  constexpr int NPAIR = PP * (PP + 1) / 2;
  T *const Bl = reinterpret_cast<T *>(&L.k);
  T *const Sl = Bl + NV * PP;
  constexpr int FCL0 = 16; // B's force rows start this far into the active-set row buffer (they need 72 NS of its 160 entries)
  T *const fcl = L.as.s + FCL0;
  static_assert(FCL0 + 6 * 12 * NS <= 160, "force rows are staged in the active-set row buffer");
  static_assert(!FASTEQ || ((NV * PP + NPAIR) * sizeof(T) <= offsetof(ActiveSetLds<T>, s) + FCL0 * sizeof(T)), "B's dv rows and B^T B must end before B's force rows");
  static_assert(NV % 2 == 0, "the pair sums of B^T B take two rows at a time");
  const int grp = lane / PP, col = lane - grp * PP;
  const bool bl = lane < G * PP; // this lane holds a part of column `col`
  T jr[NN], bs[GS];
  int ln = lane; // (re-read per phase: see tsid_tick_env)
  asm volatile("" : "+v"(ln));
  {
    T x0 = 0, c2 = 0, ck = 0;
    {
      T bc[NV]; // rows 0..NV-1 of the column, computed alike by the G lanes that share it
      T acc = -gi;
      T yv = 0;
      // CE row `col`: base dynamics rows come from Dyn, contact motion rows from the frame Jacobians
      // column order: the 6 NS contact-motion rows FIRST, then the six base-dynamics rows.  (tsid stacks the base dynamics
      // first; the solution does not depend on the order the equalities enter the factorisation, only its rounding does.)
      // A contact row has no force variables, so its column of B is zero below row NV and stays so while only contact
      // columns are reflected: their reflectors are NV long instead of NN - a third of the QR's work in double support.
      const bool isbase = col >= NC;
      const int crow = isbase ? col - NC : 6 * slot_foot<T, NS>(L, col / 6) + col % 6;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const T rdi = rdlane(rdv, i); // 1 / L[i][i] (lane i keeps it: 26 wave-uniform doubles held across these loops
                                      // are 52 SGPRs the kernel does not have)
        // two partial sums per substitution: four independent FMA chains instead of two
        T xs = ln == i ? T(1) : T(0), xs1 = 0;
        T bsum = bl ? (isbase ? L.Dyn[crow * LDD + i] : L.k.Jf[crow * LDF + i]) : T(0), bsum1 = 0;
#pragma unroll
        for (int k0 = 0; k0 < i; k0 += 2) { // two broadcasts READ, then their four FMAs (see tick_qp's header)
          const bool p1 = k0 + 1 < i;
          const T l0 = rdlane(a[k0], i), l1 = p1 ? rdlane(a[p1 ? k0 + 1 : 0], i) : T(0);
          asm volatile("" : "+v"(xs), "+v"(bsum), "+v"(xs1), "+v"(bsum1) : "s"(l0), "s"(l1));
          xs -= l0 * jr[k0]; bsum -= l0 * bc[k0];
          if (p1) { xs1 -= l1 * jr[k0 + 1]; bsum1 -= l1 * bc[k0 + 1]; }
        }
        jr[i] = (xs + xs1) * rdi;
        bc[i] = (bsum + bsum1) * rdi;
        const T yi = rdlane(acc, i) * rdi;
        if (ln == i) yv = yi;
        acc -= a[i] * yi;
      }
      // x0 = L^-T y ; c_k = ce0_k + B[:,k] . y ; trace(J0)
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const T yi = rdlane(yv, i);
        x0 += jr[i] * yi;
        ck += bc[i] * yi;
        if (ln == i) c2 = jr[i];
      }
      c2 = wave_sum(c2) + T(c.nslot) * m.Jf0_trace;
      if (bl) {
        if (isbase) ck += L.h[col - NC];
        else ck -= L.k.arhs[slot_foot<T, NS>(L, col / 6)][col % 6];
      }
      if constexpr (FASTEQ) { // the fast equality solve below reads B row by row: rows 0..NV-1 of column `col` (group 0's copy)
        if (bl && grp == 0) {
#pragma unroll
          for (int i = 0; i < NV; i++) Bl[i * PP + col] = bc[i];
        }
      }
      // rows of the dv block go to their group's lane
#pragma unroll
      for (int j = 0; j < GS; j++) {
        const T v0 = j < NV ? bc[j] : T(0), v1 = GS + j < NV ? bc[GS + j < NV ? GS + j : 0] : T(0),
                v2 = 2 * GS + j < NV ? bc[2 * GS + j < NV ? 2 * GS + j : 0] : T(0);
        bs[j] = grp == 0 ? v0 : (grp == 1 ? v1 : v2);
      }
    }
    // force rows: J0 row of the constant block; B rows 26.. = L_f^-1 (-Jc)^T for the base-dynamics columns
#pragma unroll
    for (int j = NV; j < NN; j++) jr[j] = 0;
    if (lane >= NV && lane < n) {
#pragma unroll
      for (int i = 0; i < NV; i++) jr[i] = 0;
      const int e = (lane - NV) % 12, sl = (lane - NV) / 12;
#pragma unroll
      for (int s2 = 0; s2 < NS; s2++)
#pragma unroll
        for (int b = 0; b < 12; b++)
          if (s2 == sl && b >= e) jr[NV + 12 * s2 + b] = m.Jf0[e][b];
    }
    if (lane >= n) {
#pragma unroll
      for (int i = 0; i < NV; i++) jr[i] = 0;
    }
    if constexpr (NS > 0) {
      // rows NV.. of B (non-zero for the base-dynamics columns only): fc[col][e] = sum_{b <= e} Jf0[b][e] Dyn[col][NV + b],
      // 6 x 12 NS dot products spread over the wavefront and handed over through LDS (the Jacobian scratch is dead now;
      // six lanes doing it alone read the 78 constants of Jf0 as wave-uniform scalars: 156 SGPRs at once)
      TSIDB_SYNC1();
      for (int idx = lane; idx < 6 * 12 * NS; idx += WAVE) {
        const int cl = idx / (12 * NS), rem = idx % (12 * NS), s2 = rem / 12, e = rem % 12;
        T sacc = 0;
        // (fixed trip count, predicated: with the lane-dependent bound every term waited for its own global load of Jf0)
#pragma unroll
        for (int b = 0; b < 12; b++) {
          const T term = m.Jf0[b][e] * L.Dyn[cl * LDD + NV + 12 * s2 + b];
          sacc = b <= e ? sacc + term : sacc;
        }
        fcl[idx] = sacc;
      }
      TSIDB_SYNC1();
      if constexpr (COP) {
        const T w_cop = m.params[P_W_COP];
        if (w_cop != 0 && cop_ref) {
          // CoP force task (legacy/biped.py:79-80, tsid TaskCopEquality): cost w |t x sum_i (x_i - p_ref) x f_i|^2 over the
          // two tangents t of the contact normal, x_i the world position of contact point i, f_i its force - the
          // tangential moment of the contact forces about the reference CoP.  Its Hessian w A^T A (A: 2 x 12 NS) is a
          // rank-2 term on top of the constant force-regularisation block L0 L0^T, so the factor J with J J^T = H_f^-1
          // is J0 (I - b1 a1 a1^T)(I - b2 b b^T) with a_k = rows of A J0: two rank-1 corrections of the constant rows.
          const int fcol = lane - NV; // force variable of this lane
          const bool isf = lane >= NV && lane < n;
          const T tiny = sizeof(T) == 8 ? T(1e-30) : T(1e-20);
          T A0 = 0, A1 = 0;
          if (isf) {
            const int sl = fcol / 12, pt = (fcol % 12) / 3, j = fcol % 3, f = slot_foot<T, NS>(L, sl);
            const T *R = L.oMf[f], *pp = L.oMf[f] + 9;
            const T *r = &m.params[P_CPOINTS + 3 * pt];
            T d[3], x1[3], x2[3];
#pragma unroll
            for (int i = 0; i < 3; i++) d[i] = pp[i] + R[3 * i] * r[0] + R[3 * i + 1] * r[1] + R[3 * i + 2] * r[2] - cop_ref[i];
            cross3(m.cop_t[0], d, x1);
            cross3(m.cop_t[1], d, x2);
            A0 = R[j] * x1[0] + R[3 + j] * x1[1] + R[6 + j] * x1[2]; // component j of R^T (t x d)
            A1 = R[j] * x2[0] + R[3 + j] * x2[1] + R[6 + j] * x2[2];
          }
          TSIDB_SYNC1();
          if (isf) { L.x[fcol] = A0; L.x[24 + fcol] = A1; }
          TSIDB_SYNC1();
          T t0 = 0, t1 = 0; // this lane's column of A J0 (J0 upper triangular within each foot's block)
          if (isf) {
            const int sl = fcol / 12, cb = fcol % 12;
            for (int a2 = 0; a2 <= cb; a2++) {
              const T jf = m.Jf0[a2][cb];
              t0 += L.x[12 * sl + a2] * jf;
              t1 += L.x[24 + 12 * sl + a2] * jf;
            }
          }
          const T al1 = wave_sum(t0 * t0), a12 = wave_sum(t0 * t1);
          const T be1 = al1 > tiny ? (1 - T(1) / sqrt(1 + w_cop * al1)) / al1 : T(0);
          const T bq = t1 - be1 * a12 * t0;
          const T al2 = wave_sum(bq * bq);
          const T be2 = al2 > tiny ? (1 - T(1) / sqrt(1 + w_cop * al2)) / al2 : T(0);
          TSIDB_SYNC1();
          if (isf) { L.x[fcol] = t0; L.x[24 + fcol] = bq; }
          TSIDB_SYNC1();
          if (isf) { // row of J: j0 (I - b1 a1 a1^T)(I - b2 b b^T)
            T s1 = 0, s2 = 0;
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) s1 += jr[NV + cc] * L.x[cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) jr[NV + cc] -= be1 * s1 * L.x[cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) s2 += jr[NV + cc] * L.x[24 + cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) jr[NV + cc] -= be2 * s2 * L.x[24 + cc];
          }
          if (lane < 6) { // column of B = J^T CE^T: the same two (symmetric) factors, in the same order
            T fc[12 * NS];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) fc[cc] = fcl[lane * 12 * NS + cc];
            T s1 = 0, s2 = 0;
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) s1 += fc[cc] * L.x[cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) fc[cc] -= be1 * s1 * L.x[cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) s2 += fc[cc] * L.x[24 + cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) fc[cc] -= be2 * s2 * L.x[24 + cc];
#pragma unroll
            for (int cc = 0; cc < 12 * NS; cc++) fcl[lane * 12 * NS + cc] = fc[cc];
          }
          c1 += w_cop * wave_sum(A0 * A0 + A1 * A1); // trace of the Hessian (tolerance scale only)
          TSIDB_SYNC1();
        }
      }
      // rows of the force block go to their group's lane
      if (bl && col >= NC) {
#pragma unroll
        for (int j = 0; j < GS; j++) {
          const int row = grp * GS + j;
          if (row >= NV && row < NN) bs[j] = fcl[(col - NC) * 12 * NS + row - NV];
        }
      }
      TSIDB_SYNC1(); // (the sweep below writes the row buffer)
    }
    TSIDB_STAMP(5);
    bool fast_done = false, swept = false; // swept: the fast attempt got as far as its sweep and found a violated inequality
    if constexpr (FASTEQ) {
      if (spd && fast_eq) {
        // S = B^T B, one pair (c <= d) per lane and round: 26 dv rows from LDS, the force rows for base-dynamics pairs
        TSIDB_SYNC1();
        for (int pidx = lane; pidx < NPAIR; pidx += WAVE) {
          int d = (int)((sqrtf((float)(8 * pidx + 1)) - 1.0f) * 0.5f); // (pair index -> (c, d), d >= c; float is exact enough
          d += (d + 1) * (d + 2) / 2 <= pidx ? 1 : 0;                   //  for < 200 pairs, one correction step either way)
          d -= d * (d + 1) / 2 > pidx ? 1 : 0;
          const int cc = pidx - d * (d + 1) / 2;
          T s0 = 0, s1 = 0;
#pragma unroll
          for (int r = 0; r < NV; r += 2) {
            s0 += Bl[r * PP + cc] * Bl[r * PP + d];
            s1 += Bl[(r + 1) * PP + cc] * Bl[(r + 1) * PP + d];
          }
          if constexpr (NS > 0) {
            if (cc >= NC) { // (d >= cc): both are base-dynamics columns
              const T *fa = fcl + (cc - NC) * 12 * NS, *fb = fcl + (d - NC) * 12 * NS;
#pragma unroll
              for (int e = 0; e < 12 * NS; e += 2) { s0 += fa[e] * fb[e]; s1 += fa[e + 1] * fb[e + 1]; }
            }
          }
          Sl[pidx] = s0 + s1;
        }
        TSIDB_SYNC1();
        TSIDB_STAMP(10);
        // lane r < PP: row r of S in registers; Cholesky S = Ls Ls^T; then forward substitutions that share Ls's broadcasts:
        // u = Ls^-1 c and column `lane` of Ls^-1 (as for J0 above), so that lambda = Ls^-T u needs no transposed access
        T sr[PP];
#pragma unroll
        for (int j = 0; j < PP; j++) {
          const int rr = lane < PP ? lane : 0, hi = rr > j ? rr : j, lo = rr > j ? j : rr;
          sr[j] = lane < PP ? Sl[hi * (hi + 1) / 2 + lo] : T(0);
        }
        // Conditioning guard.  B^T B squares B's condition number; what is lost is governed by rho_k = Ls[k][k]^2 / S[k][k],
        // the share of column k of B that is NOT in the span of the columns before it (scale free).  The v1 robot's
        // equality blocks have rho >= 2e-3 (error 1e-12 against the QR); the v0 robot's two rigid 6-D contacts on five-joint
        // legs are nearly dependent (DESIGN.md section 4 "Second robot") and lose seven digits here: below 1e-4 the env
        // takes the QR path, which does not square anything.
        int bad = 0;
        T srd = 0; // lane k: 1 / Ls[k][k]
        T sd0 = 0; // lane k: S[k][k]
#pragma unroll
        for (int k = 0; k < PP; k++) sd0 = ln == k ? sr[k] : sd0;
#pragma unroll
        for (int k = 0; k < PP; k++) {
          const T skk = rdlane(sr[k], k);
          bad = skk > T(1e-4) * rdlane(sd0, k) ? bad : 1;
          const T rk = rsqrt_t(skk > 0 ? skk : T(1));
          if (ln == k) srd = rk;
          const T lik = ln == k ? skk * rk : sr[k] * rk;
          sr[k] = lik;
          T lk = lik; // (broadcasts four at a time, read ahead of their FMAs and pinned there: see the Cholesky of H_dv)
#pragma unroll
          for (int j0 = k + 1; j0 < PP; j0 += 4) {
            const bool p1 = j0 + 1 < PP, p2 = j0 + 2 < PP, p3 = j0 + 3 < PP;
            const T u0 = rdlane(lk, j0), u1 = p1 ? rdlane(lk, p1 ? j0 + 1 : 0) : T(0), u2 = p2 ? rdlane(lk, p2 ? j0 + 2 : 0) : T(0),
                    u3 = p3 ? rdlane(lk, p3 ? j0 + 3 : 0) : T(0);
            if (j0 > k + 1) asm volatile("" : "+v"(lk), "+v"(sr[j0 - 1]) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
            else asm volatile("" : "+v"(lk) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
            sr[j0] -= lk * u0;
            if (p1) sr[j0 + 1] -= lk * u1;
            if (p2) sr[j0 + 2] -= lk * u2;
            if (p3) sr[j0 + 3] -= lk * u3;
          }
        }
#pragma unroll
        for (int k = 0; k < PP; k++) asm volatile("" : "+v"(sr[k]));
        asm volatile("" : "+v"(bad));
        TSIDB_STAMP(11);
        if (!__ballot(bad != 0)) {
          T li[PP]; // column `lane` of Ls^-1
          T acc = (bl && grp == 0) ? ck : T(0), uv = 0;
#pragma unroll
          for (int i = 0; i < PP; i++) {
            const T rdi = rdlane(srd, i);
            T xs = ln == i ? T(1) : T(0), xs1 = 0;
#pragma unroll
            for (int k0 = 0; k0 < i; k0 += 2) { // two broadcasts READ, then their FMAs (pinned: no SGPR pile-up)
              const bool p1 = k0 + 1 < i;
              const T l0 = rdlane(sr[k0], i), l1 = p1 ? rdlane(sr[p1 ? k0 + 1 : 0], i) : T(0);
              asm volatile("" : "+v"(xs), "+v"(xs1) : "s"(l0), "s"(l1));
              xs -= l0 * li[k0];
              if (p1) xs1 -= l1 * li[k0 + 1];
            }
            li[i] = (xs + xs1) * rdi;
            const T ui = rdlane(acc, i) * rdi;
            if (ln == i) uv = ui;
            acc -= sr[i] * ui;
          }
          T lam = 0; // lane j < PP: lambda_j = sum_k (Ls^-1)[k][j] u_k
          {
            T lam1 = 0;
#pragma unroll
            for (int k = 0; k < PP; k += 2) {
              const bool p1 = k + 1 < PP;
              const T u0 = rdlane(uv, k), u1 = p1 ? rdlane(uv, p1 ? k + 1 : 0) : T(0);
              asm volatile("" : "+v"(lam), "+v"(lam1) : "s"(u0), "s"(u1));
              lam += li[k] * u0;
              if (p1) lam1 += li[k + 1] * u1;
            }
            lam += lam1;
          }
          TSIDB_STAMP(12);
          // z = -B lambda (lane r = row r of B), x = x0 + J0 z
          if (lane < PP) L.x[lane] = lam;
          TSIDB_SYNC1();
          T z = 0;
          if (lane < NV) {
#pragma unroll
            for (int cc = 0; cc < PP; cc++) z -= Bl[lane * PP + cc] * L.x[cc];
          } else if (lane < NN) {
            if constexpr (NS > 0) {
#pragma unroll
              for (int cb = 0; cb < 6; cb++) z -= fcl[cb * 12 * NS + (lane - NV)] * L.x[NC + cb];
            }
          }
          T xf = lane < NV ? x0 : T(0);
          {
            T xf1 = 0;
#pragma unroll
            for (int i = 0; i < NN; i += 2) {
              const bool p1 = i + 1 < NN;
              const T z0 = rdlane(z, i), z1 = p1 ? rdlane(z, p1 ? i + 1 : 0) : T(0);
              asm volatile("" : "+v"(xf), "+v"(xf1) : "s"(z0), "s"(z1));
              xf += jr[i] * z0;
              if (p1) xf1 += jr[i + 1] * z1;
            }
            xf += xf1;
          }
          TSIDB_SYNC1();
          if (lane < n) L.x[lane] = xf;
          TSIDB_SYNC1();
          TSIDB_STAMP(6);
          // feasibility sweep at the equality-constrained optimum
          c.iq = p;
          RowDesc<T> rdf[3];
#pragma unroll
          for (int rr = 0; rr < 3; rr++) rdf[rr] = row_desc(m, L, c, lane + WAVE * rr);
          act_partials(L, n, lane);
          TSIDB_SYNC1();
          T psi = 0;
#pragma unroll
          for (int rr = 0; rr < 3; rr++)
            if (lane + WAVE * rr < c.nin) {
              const T sv = row_eval(rdf[rr], L);
              psi += sv < 0 ? sv : T(0);
            }
          psi = wave_sum(psi);
          if (fabs(psi) <= T(c.nin) * T(2.220446049250313e-16) * c1 * c2 * T(100)) {
            fast_done = true;
            qp_status = 0;
            qp_iters = 1;
          }
          swept = true;
          TSIDB_SYNC1();
        }
      }
    }
    if (!fast_done) {
    // ---- Householder QR of B (column c on lanes c + PP g) applied to J (rows in lanes 0..n-1)
    T R_norm = 1;    // (R_norm and the degeneracy flag are wave-uniform and live across the whole unrolled QR: pinned in
    int degen = 0;   //  VGPRs - as SGPRs they were spilled to VGPR lanes and reloaded around every column)
    T xeq = lane < NV ? x0 : T(0);
#pragma unroll
    for (int k = 0; k < PP; k++) {
      const int gk = k / GS, sk = k % GS; // the group and slot that hold row k (compile-time after unrolling)
      const int RL = k < NC ? NV : NN;    // rows the reflector of column k reaches (exact zeros beyond, for a contact column)
      {
        // tail norm of column k: each of its lanes sums its own rows > k
        T sig_hi = 0, sig_all = 0;
#pragma unroll
        for (int j = 0; j < GS; j++) {
          const T sq = bs[j] * bs[j];
          sig_all += sq;
          if (j > sk) sig_hi += sq;
        }
        T sigma = rdlane(sig_hi, k + PP * gk);
#pragma unroll
        for (int g = 0; g < G; g++)
          if (g > gk) sigma += rdlane(sig_all, k + PP * g);
        const T alpha = rdlane(bs[sk], k + PP * gk);
        T dkk = alpha;
        if (sigma > 0) {
          const T nrm = sqrt(alpha * alpha + sigma);
          const T v0 = alpha + (alpha >= 0 ? nrm : -nrm);
          const T beta = T(2) / (v0 * v0 + sigma);
          dkk = alpha >= 0 ? -nrm : nrm;
          // s = v . b_col: lane (c, g) sums the rows of its group (sbg[g] is meaningful on group g's lanes only);
          // w = J_row . v (all lanes)
          T sbg[G], wj = v0 * jr[k], wj1 = 0;
#pragma unroll
          for (int g = 0; g < G; g++) sbg[g] = 0;
          sbg[gk] = v0 * bs[sk];
#pragma unroll
          for (int i0 = k + 1; i0 < RL; i0 += 2) { // two reflector entries READ, then their four FMAs
            const bool p1 = i0 + 1 < RL;
            const int i1 = p1 ? i0 + 1 : i0, g0 = i0 / GS, s0 = i0 % GS, g1 = i1 / GS, s1 = i1 % GS;
            const T u0 = rdlane(bs[s0], k + PP * g0), u1 = p1 ? rdlane(bs[s1], k + PP * g1) : T(0);
            asm volatile("" : "+v"(wj), "+v"(wj1), "+v"(sbg[g0]), "+v"(sbg[g1]) : "s"(u0), "s"(u1));
            sbg[g0] += u0 * bs[s0];
            wj += u0 * jr[i0];
            if (p1) { sbg[g1] += u1 * bs[s1]; wj1 += u1 * jr[i1]; }
          }
          const T sbm = grp == 0 ? sbg[0] : (grp == 1 ? sbg[1] : sbg[2]);
          T sb = bcast(sbm, col) + bcast(sbm, col + PP) + bcast(sbm, col + 2 * PP);
          sb *= beta;
          wj = (wj + wj1) * beta;
          const bool upd = bl && col > k; // columns <= k are final (their rows >= k are already zero)
          T sg[G];
#pragma unroll
          for (int g = 0; g < G; g++) sg[g] = (upd && grp == g) ? sb : T(0);
          jr[k] -= wj * v0;
          bs[sk] -= sg[gk] * v0;
          // the reflector is READ AGAIN from lane k's registers: left to itself the compiler keeps the first loop's
          // NN broadcast values (2 SGPRs each) for this loop, i.e. spills them to VGPR lanes with v_writelane and
          // reloads them with v_readlane - three times the instructions of reading them again, and the spill lanes
          // were what pushed the kernel over its VGPR budget (3 800 SGPR spills -> see DESIGN.md section 4).
          // (Round 3, measured and dropped: the reflector staged in LDS by the lanes that hold it and read back at wave-uniform
          //  addresses, 8 entries per chunk with the next chunk's reads in flight - 1560 v_readlane less on paper; the unrolled
          //  k loop outgrew the pragma-unroll threshold, and with the threshold raised the kernel spilled 2 600 VGPRs.)
#pragma unroll
          for (int j = 0; j < GS; j++) asm volatile("" : "+v"(bs[j]));
#pragma unroll
          for (int i0 = k + 1; i0 < RL; i0 += 2) { // (column k's lanes are untouched until the loop ends: sg = 0 there)
            const bool p1 = i0 + 1 < RL;
            const int i1 = p1 ? i0 + 1 : i0, g0 = i0 / GS, s0 = i0 % GS, g1 = i1 / GS, s1 = i1 % GS;
            const T u0 = rdlane(bs[s0], k + PP * g0), u1 = p1 ? rdlane(bs[s1], k + PP * g1) : T(0);
            // this pair's targets and the previous pair's last result are "changed": the previous FMAs come before, these after
            asm volatile("" : "+v"(jr[i0]), "+v"(jr[i1]), "+v"(jr[i0 - 1]), "+v"(wj) : "s"(u0), "s"(u1));
            jr[i0] -= wj * u0;
            bs[s0] -= sg[g0] * u0;
            if (p1) { jr[i1] -= wj * u1; bs[s1] -= sg[g1] * u1; }
          }
          if (bl && col == k) { // column k of R: the diagonal, zeros below
#pragma unroll
            for (int j = 0; j < GS; j++)
              if (grp > gk || (grp == gk && j > sk)) bs[j] = 0;
            if (grp == gk) bs[sk] = dkk;
          }
        }
        const T ad = fabs(dkk);
        degen = ad <= Eps<T>::v * R_norm ? 1 : degen;
        R_norm = ad > R_norm ? ad : R_norm;
        asm volatile("" : "+v"(degen), "+v"(R_norm));
      }
    }
    const bool degenerate = degen != 0;
    // ---- R^T t = -c (forward), x = x0 + J[:, :p] t  (the equality multipliers u = R^-1 t are not needed).
    //      R[i][c] sits on lane c + PP g(i): tg[g] is column c's running sum over the rows of group g
    T tg[G], tv[PP];
    tg[0] = -ck;
#pragma unroll
    for (int g = 1; g < G; g++) tg[g] = 0;
#pragma unroll
    for (int i = 0; i < PP; i++) {
      const int gi = i / GS, si = i % GS;
      T num = rdlane(tg[0], i);
#pragma unroll
      for (int g = 1; g < G; g++) num += rdlane(tg[g], i + PP * g);
      tv[i] = num / rdlane(bs[si], i + PP * gi);
      tg[gi] -= bs[si] * tv[i];
    }
#pragma unroll
    for (int k = 0; k < PP; k++) xeq += jr[k] * tv[k];
    TSIDB_STAMP(6);
    // ---- first feasibility sweep straight from registers' result; J / R go to LDS only if the
    //      active-set iterations are actually needed
    if (lane < n) L.x[lane] = xeq;
    TSIDB_SYNC1();
    c.iq = p;
    c.R_norm = R_norm;
    int iters = 1;
    int status = -1;
    if (!spd) status = 2;
    else if (degenerate) { status = 4; iters = 0; }
    RowDesc<T> rdesc[3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) rdesc[rr] = row_desc(m, L, c, lane + WAVE * rr);
    if (status < 0 && !swept) { // (after a failed fast attempt the answer is known: the active-set loop's own first sweep follows)
      act_partials(L, n, lane);
      TSIDB_SYNC1();
      T psi = 0;
#pragma unroll
      for (int rr = 0; rr < 3; rr++)
        if (lane + WAVE * rr < c.nin) {
          const T sv = row_eval(rdesc[rr], L);
          psi += sv < 0 ? sv : T(0);
        }
      psi = wave_sum(psi);
      if (fabs(psi) <= T(c.nin) * T(2.220446049250313e-16) * c1 * c2 * T(100)) status = 0;
    }
    if (status < 0) status = qp_active_regs<T, NS>(m, L, c, lane, jr, xeq, rdesc, c1, c2, (int)m.params[P_MAX_ITER], iters);
    qp_status = status;
    qp_iters = iters;
    } // (!fast_done)
  }
}

// --------------------------------------------------------------------------- the tick
template <typename T, int NS, bool COP>
__device__ __forceinline__ void tsid_tick_env(const DevModel<T> &m, TickLds<T> &L, int lane, T *q, T *v, const T *com_ref,
                              const T *posture_ref, const T *foot_ref, const T *contact_ref,
                              const uint8_t *cact, const T *cop_frames, T *tau, T *dv, T *fout, int *status_out,
                              T *obs, T *rowx, int *info, const T *qpos_sim, const T *qvel_sim, const T *cop_ref, bool fast_eq) {
  TSIDB_STAMP(0);
  // ---- stage state.  Closed loop: the TSID state is read from the sim state each
  //      tick - quat wxyz -> xyzw, world-frame base linear velocity -> body frame, sim joint order ->
  //      TSID joint order; otherwise TSID integrates its own state as the reference does (main.py:128).
  if (qpos_sim) {
    if (lane < NQ) {
      T val;
      if (lane < 3) val = qpos_sim[lane];
      else if (lane < 6) val = qpos_sim[lane + 1];
      else if (lane == 6) val = qpos_sim[3];
      else val = qpos_sim[7 + m.tsid2sim[lane - 7]];
      L.qs[lane] = val;
    }
    if (lane < NV) {
      T val;
      if (lane < 3) {
        T R[9];
        quat_to_R(qpos_sim[4], qpos_sim[5], qpos_sim[6], qpos_sim[3], R);
        val = R[lane] * qvel_sim[0] + R[3 + lane] * qvel_sim[1] + R[6 + lane] * qvel_sim[2];
      } else if (lane < 6) val = qvel_sim[lane];
      else val = qvel_sim[6 + m.tsid2sim[lane - 6]];
      L.vs[lane] = val;
    }
  } else {
    if (lane < NQ) L.qs[lane] = q[lane];
    if (lane < NV) L.vs[lane] = v[lane];
  }
  if (lane >= 32 && lane < 34) L.act[lane - 32] = cact[lane - 32] != 0; // (the caller dispatched on act[0] + act[1] == NS)
  TSIDB_SYNC1();
  rbd_terms(m, L, lane);
  if (m.params[P_TSID_ARMATURE] != 0) { // closed-loop knob: rotor inertia of the actuated joints in TSID's model
    if (lane >= 6 && lane < NV) L.Dyn[lane * LDD + lane] += m.params[P_TSID_ARMATURE];
    TSIDB_SYNC1();
  }
  TSIDB_STAMP(1);

  QpCtx<T> c;
  c.nslot = NS;
  c.n = NV + 12 * NS;
  c.p = 6 + 6 * NS;
  c.nin = 34 * NS + 2 * NA + 2 * NV;
  const int n = c.n;

  // ---- task right-hand sides
  if (lane < 4) { // contact LF / RF, foot LF / RF: ONE copy of se3_rhs for the four lanes (two copies in two branches ran one after the other)
    const bool ct = lane < 2;
    se3_rhs(L, lane & 1, ct ? contact_ref + 12 * lane : foot_ref + 24 * (lane - 2), ct ? 12 : 24,
            ct ? m.params[P_KP_CONTACT] : m.params[P_KP_FOOT], ct ? m.params[P_KD_CONTACT] : m.params[P_KD_FOOT], L.k.arhs[lane]);
  } else if (lane < 7) {
    const int i = lane - 4;
    L.k.acomr[i] = -m.params[P_KP_COM] * (L.com[i] - com_ref[i]) - m.params[P_KD_COM] * (L.vcom[i] - com_ref[3 + i]) +
                 com_ref[6 + i] - L.acomd[i];
  } else if (lane < 10) { // angular-momentum task, zero reference (legacy/biped.py:82-87): -Kp L - drift
    const int i = lane - 7;
    L.k.aamr[i] = -m.params[P_KP_AM + i] * L.Lam[i] - L.dLam[i];
  } else if (lane >= 32 && lane < 32 + NA) {
    const int r = lane - 32;
    L.k.apost[r] = -m.params[P_KP_POSTURE + r] * (L.qs[7 + r] - posture_ref[r]) - m.params[P_KD_POSTURE + r] * L.vs[6 + r];
  }
  // ---- right block of the dynamics rows: Dyn[r][26 + 12 s + cc] = -sum_i T[i][cc] Jf[6 f + i][r]
  for (int idx = lane; idx < NV * 12 * c.nslot; idx += WAVE) {
    const int r = idx / (12 * c.nslot), cc = idx % (12 * c.nslot), s = cc / 12, e = cc % 12, f = slot_foot<T, NS>(L, s);
    T a = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) a += m.Tgen[i][e] * L.k.Jf[(6 * f + i) * LDF + r];
    L.Dyn[r * LDD + NV + cc] = -a;
  }
  TSIDB_SYNC1(); // kinematics scratch is dead from here on
  TSIDB_STAMP(2);

  // ---- Hessian block of dv in registers: lane i owns row i
  const T w_foot = m.params[P_W_FOOT], w_com = m.params[P_W_COM], w_post = m.params[P_W_POSTURE], reg = m.params[P_HESS_REG];
  T jt[15], a[NV];
  {
    const int i = lane < NV ? lane : 0;
#pragma unroll
    for (int r = 0; r < 12; r++) jt[r] = lane < NV ? L.k.Jf[r * LDF + i] : T(0);
#pragma unroll
    for (int r = 0; r < 3; r++) jt[12 + r] = lane < NV ? L.k.Jcom[r * LDF + i] : T(0);
  }
  T gi = 0;
#pragma unroll
  for (int r = 0; r < 12; r++) gi -= w_foot * jt[r] * L.k.arhs[2 + r / 6][r % 6];
#pragma unroll
  for (int r = 0; r < 3; r++) gi -= w_com * jt[12 + r] * L.k.acomr[r];
  if (lane >= 6 && lane < NV) gi -= w_post * L.k.apost[lane - 6];
  // H[i][j] = sum_r w_r J_r[i] J_r[j]: lane i keeps w_r J_r[i]; J_r[j] is read back from the Jacobians in LDS at a
  // wave-uniform address (one ds_read2_b64 per two values, in the LDS pipe beside the FMAs) instead of a v_readlane
  // pair + wait states per value from the neighbour's registers: 390 broadcast values per tick
  {
    T wjt[15];
#pragma unroll
    for (int r = 0; r < 12; r++) wjt[r] = w_foot * jt[r];
#pragma unroll
    for (int r = 0; r < 3; r++) wjt[12 + r] = w_com * jt[12 + r];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      T acc = 0, acc1 = 0;
#pragma unroll
      for (int r = 0; r < 12; r++) {
        if (r & 1) acc1 += wjt[r] * L.k.Jf[r * LDF + j];
        else acc += wjt[r] * L.k.Jf[r * LDF + j];
      }
#pragma unroll
      for (int r = 0; r < 3; r++) {
        if (r & 1) acc1 += wjt[12 + r] * L.k.Jcom[r * LDF + j];
        else acc += wjt[12 + r] * L.k.Jcom[r * LDF + j];
      }
      a[j] = acc + acc1;
    }
  }
  const T w_am = m.params[P_W_AM];
  if (w_am != 0) { // optional angular-momentum rows; wave-uniform branch
    T ja[3];
#pragma unroll
    for (int r = 0; r < 3; r++) ja[r] = lane < NV ? L.k.Jam[r * LDF + (lane < NV ? lane : 0)] : T(0);
#pragma unroll
    for (int r = 0; r < 3; r++) gi -= w_am * ja[r] * L.k.aamr[r];
#pragma unroll
    for (int j = 0; j < NV; j++) {
      T acc = 0;
#pragma unroll
      for (int r = 0; r < 3; r++) acc += w_am * ja[r] * rdlane(ja[r], j);
      a[j] += acc;
    }
  }
  if (lane >= NV) gi = 0;
  // (`lane` is re-read through an empty asm at each phase: the lane == j masks of one unrolled phase, 26 SGPR pairs,
  //  are otherwise kept for the next phase's comparisons, i.e. spilled to VGPR lanes and reloaded - a v_cmp is cheaper)
  int ln = lane;
  asm volatile("" : "+v"(ln));
  T dg = 0; // this lane's diagonal entry
#pragma unroll
  for (int j = 0; j < NV; j++)
    if (ln == j) { a[j] += reg + (j >= 6 ? w_post : T(0)); dg = a[j]; }
  T c1 = wave_sum(dg) + T(c.nslot) * m.Hf_trace; // trace (one DPP reduction instead of 26 broadcasts)

  int qp_status = -1, qp_iters = 0;
  TSIDB_STAMP(3);
  // ---- Cholesky in registers (right-looking; lane i holds row i of L in a[0..i]); rd[k] = 1/L[k][k]
  T rdv = 0; // lane k: 1 / L[k][k]
  int notspd = 0; // (a VGPR flag, pinned per pivot: 26 compare masks kept to be and-ed at the end are 52 SGPRs)
  asm volatile("" : "+v"(ln));
#pragma unroll
  for (int k = 0; k < NV; k++) {
    const T akk = rdlane(a[k], k);
    notspd = akk > 0 ? notspd : 1;
    asm volatile("" : "+v"(notspd));
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    if (ln == k) rdv = rk;
    const T lik = ln == k ? akk * rk : a[k] * rk;
    a[k] = lik;
    // broadcasts four at a time, read ahead of their FMAs (left alone: v_readlane x2 - s_nop 1 - v_fma per entry, every
    // FMA waiting out the v_readlane -> VALU hazard).  The empty asm takes the four values as SGPR inputs, so they exist
    // before it, and "changes" lik and the previous group's last result, so that group's FMAs come before it and this
    // group's after.
    T lk = lik;
#pragma unroll
    for (int j0 = k + 1; j0 < NV; j0 += 4) {
      const bool p1 = j0 + 1 < NV, p2 = j0 + 2 < NV, p3 = j0 + 3 < NV;
      const T u0 = rdlane(lk, j0), u1 = p1 ? rdlane(lk, p1 ? j0 + 1 : 0) : T(0), u2 = p2 ? rdlane(lk, p2 ? j0 + 2 : 0) : T(0),
              u3 = p3 ? rdlane(lk, p3 ? j0 + 3 : 0) : T(0);
      if (j0 > k + 1) asm volatile("" : "+v"(lk), "+v"(a[j0 - 1]) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      else asm volatile("" : "+v"(lk) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      a[j0] -= lk * u0;
      if (p1) a[j0 + 1] -= lk * u1;
      if (p2) a[j0 + 2] -= lk * u2;
      if (p3) a[j0 + 3] -= lk * u3;
    }
  }
  TSIDB_STAMP(4);
  // (the substitutions broadcast L's entries again: carried over from the factorisation they would be 650 SGPRs,
  //  i.e. spilled to VGPR lanes and reloaded - more instructions than reading them again)
#pragma unroll
  for (int k = 0; k < NV; k++) asm volatile("" : "+v"(a[k]));
  const bool spd = notspd == 0;
  tick_qp<T, NS, COP>(m, L, c, lane, a, rdv, gi, c1, spd, cop_ref, qp_status, qp_iters, fast_eq);
  int status = qp_status, iters = qp_iters;

  TSIDB_STAMP(8);
  // ---- decode: dv, f, tau = M_a dv + h_a - J_a^T f.  A failed QP (the reference stops its loop there, main.py:122-124,
  //      before it reads the solution) leaves dv = f = tau = 0: nothing of the solver's last iterate is handed on, and in
  //      the closed loop the env's motors go limp instead of being driven by it
  const bool solved = status == 0;
  if (lane < NV) dv[lane] = solved ? L.x[lane] : T(0);
  if (lane < 24) {
    const int fo = lane / 12, e = lane % 12;
    T val = 0;
    for (int s = 0; s < c.nslot; s++)
      if (solved && slot_foot<T, NS>(L, s) == fo) val = L.x[NV + 12 * s + e];
    fout[lane] = val;
    L.as.s[lane] = val; // staged for the CoP
  }
  T tau2 = 0;
  if (lane < NA) {
    // (compile-time length, four partial sums: the run-time loop with one accumulator waited out an LDS round trip and a
    //  dependent float64 FMA per term - 38 to 50 of them)
    constexpr int NNc = NV + 12 * NS;
    const T *Dr = &L.Dyn[(6 + lane) * LDD];
    T t = L.h[6 + lane], t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
    for (int e = 0; e < NNc; e += 4) {
      t += Dr[e] * L.x[e];
      if (e + 1 < NNc) t1 += Dr[e + 1] * L.x[e + 1];
      if (e + 2 < NNc) t2 += Dr[e + 2] * L.x[e + 2];
      if (e + 3 < NNc) t3 += Dr[e + 3] * L.x[e + 3];
    }
    t = (t + t1) + (t2 + t3);
    if (m.params[P_FRICTION_COMP] != 0) { // Coulomb-friction feed-forward along the commanded joint velocity
      const T vn = L.vs[6 + lane] + m.params[P_DT] * L.x[6 + lane], sat = vn * T(20);
      t += m.params[P_FRICTION_COMP] * (sat > 1 ? T(1) : (sat < -1 ? T(-1) : sat));
    }
    t = solved ? t : T(0);
    tau[lane] = t;
    tau2 = t * t;
  }
  if (rowx) tau2 = wave_sum(tau2);
  if (lane == 0) {
    status_out[0] = status;
    if (info) { info[0] = iters; info[1] = c.iq; }
  }
  TSIDB_SYNC1();
  // ---- observations from this tick's data (main.py:132-142 reads data() before recomputing)
  if (obs) {
    // contact wrenches in the sole frames: component i of foot fo on lane 6 fo + i (twelve 12-term sums side by side
    // instead of one lane doing all 144 terms), then the small centre-of-pressure arithmetic on the broadcast values
    T wl = 0;
    if (lane < 12) {
      const int fo = lane / 6, i = lane % 6;
#pragma unroll
      for (int e = 0; e < 12; e++) wl += m.Tgen[i][e] * L.as.s[12 * fo + e];
    }
    T fz[2], copw[2][3];
#pragma unroll
    for (int fo = 0; fo < 2; fo++) {
      const T w2 = rdlane(wl, 6 * fo + 2), w3 = rdlane(wl, 6 * fo + 3), w4 = rdlane(wl, 6 * fo + 4);
      T cl[3] = {0, 0, 0};
      const bool on = L.act[fo] != 0;
      if (on && w2 > T(1e-3)) { cl[0] = w4 / w2; cl[1] = w3 / w2; }
      const T *F = (m.params[P_QUIRKS] != 0 && cop_frames) ? cop_frames + 12 * fo : L.oMf[fo];
#pragma unroll
      for (int i = 0; i < 3; i++) copw[fo][i] = F[3 * i] * cl[0] + F[3 * i + 1] * cl[1] + F[3 * i + 2] * cl[2] + F[9 + i];
      fz[fo] = w2;
    }
    T cop[3] = {0, 0, 0};
    if (NS == 2 && fz[0] + fz[1] != 0) {
      cop[0] = (copw[0][0] * fz[0] + copw[1][0] * fz[1]) / (fz[0] + fz[1]);
      cop[1] = (copw[0][1] * fz[0] + copw[1][1] * fz[1]) / (fz[0] + fz[1]);
    }
    if (lane < 3) obs[NQ + NV + 3 + lane] = lane == 0 ? cop[0] : (lane == 1 ? cop[1] : cop[2]);
    if (lane >= 8 && lane < 11) obs[NQ + NV + lane - 8] = L.com[lane - 8];
    if (lane >= 16 && lane < 19) obs[NQ + NV + 6 + lane - 16] = L.oMf[0][9 + lane - 16];
    if (lane >= 24 && lane < 27) obs[NQ + NV + 9 + lane - 24] = L.oMf[1][9 + lane - 24];
  }
  // ---- integrate_dv (WalkController.py:291-295); a failed QP leaves the state untouched
  if (status == 0) {
    const T dt = m.params[P_DT];
    T vm = 0;
    if (lane < NV) {
      const T vv = L.vs[lane], dd = L.x[lane];
      vm = dt * (vv + T(0.5) * dt * dd);
      L.vs[lane] = vv + dt * dd;
      L.as.Ra[lane] = vm;
    }
    TSIDB_SYNC1();
    if (lane >= 6 && lane < NV) L.qs[lane + 1] += vm;
    if (lane == 0) {
      const T *vl = &L.as.Ra[0], *w = &L.as.Ra[3];
      const T th2 = dot3(w, w), th = sqrt(th2);
      const T small = sizeof(T) == 8 ? T(1e-8) : T(1e-4);
      T b, cc, sh, ch;
      if (th < small) { b = T(0.5) - th2 / 24; cc = T(1.0 / 6) - th2 / 120; sh = T(0.5) - th2 / 48; ch = 1 - th2 / 8; }
      else {
        T s1, c1, s2, c2;
        sincos_t(th, s1, c1);
        sincos_t(T(0.5) * th, s2, c2);
        b = (1 - c1) / th2; cc = (th - s1) / (th2 * th); sh = s2 / th; ch = c2;
      }
      T wxv[3], wxwxv[3], pd[3], R0[9], rp[3];
      cross3(w, vl, wxv); cross3(w, wxv, wxwxv);
#pragma unroll
      for (int i = 0; i < 3; i++) pd[i] = vl[i] + b * wxv[i] + cc * wxwxv[i];
      quat_to_R(L.qs[3], L.qs[4], L.qs[5], L.qs[6], R0);
      mat3vec(R0, pd, rp);
      const T dq[4] = {sh * w[0], sh * w[1], sh * w[2], ch};
      const T aq[4] = {L.qs[3], L.qs[4], L.qs[5], L.qs[6]};
      T r[4];
      r[0] = aq[3] * dq[0] + aq[0] * dq[3] + aq[1] * dq[2] - aq[2] * dq[1];
      r[1] = aq[3] * dq[1] - aq[0] * dq[2] + aq[1] * dq[3] + aq[2] * dq[0];
      r[2] = aq[3] * dq[2] + aq[0] * dq[1] - aq[1] * dq[0] + aq[2] * dq[3];
      r[3] = aq[3] * dq[3] - aq[0] * dq[0] - aq[1] * dq[1] - aq[2] * dq[2];
      const T nn = T(1) / sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
#pragma unroll
      for (int i = 0; i < 3; i++) L.qs[i] += rp[i];
#pragma unroll
      for (int i = 0; i < 4; i++) L.qs[3 + i] = r[i] * nn;
    }
    TSIDB_SYNC1();
    if (lane < NQ) q[lane] = L.qs[lane];
    if (lane < NV) v[lane] = L.vs[lane];
  }
  if (obs) {
    if (lane < NQ) obs[lane] = L.qs[lane];
    if (lane < NV) obs[NQ + lane] = L.vs[lane];
  }
  // ---- reward / done (SURVEY.md 8d write list; no reference counterpart): tracking reward on this tick's CoM
  //      error minus a torque penalty; done = failed QP or fallen (base height / tilt of the new state)
  if (rowx && lane == 0) {
    T rew = 0, dn = 1;
    if (status == 0) {
      const T up = 1 - 2 * (L.qs[3] * L.qs[3] + L.qs[4] * L.qs[4]);
      const bool fall = L.qs[2] < m.params[P_DONE_HEIGHT] || up < m.params[P_DONE_TILT];
      const T e0 = L.com[0] - com_ref[0], e1 = L.com[1] - com_ref[1], e2 = L.com[2] - com_ref[2];
      const T sg = m.params[P_REW_SIGMA];
      dn = fall ? T(1) : T(0);
      rew = fall ? T(0) : exp(-(e0 * e0 + e1 * e1 + e2 * e2) / (sg * sg)) - m.params[P_REW_CTAU] * tau2;
    }
    rowx[0] = rew;
    rowx[1] = dn;
  }
  TSIDB_STAMP(9);
}

} // namespace tsidb
