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

namespace tsidb {

constexpr int LDM = 27;
#ifndef TSIDB_NEWTON_INCR_MAX
#define TSIDB_NEWTON_INCR_MAX 8
#endif
constexpr int NEWTON_INCR_MAX = TSIDB_NEWTON_INCR_MAX; // Newton: at most this many changed rows are applied to the factor as
                                                       // rank-1 updates; more, and the Hessian is rebuilt (< 0: never incremental)

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
    struct {
      T K[NB][21];      // per-body contact inertia (packed sym 6x6), composite over subtrees (Newton loop)
      T hn[MAXHH][3];   // robot<->robot contacts: normal (geom1 -> geom2); copied here from the staging below once the
      int hb1[MAXHH];   //                        collision phase is over; body of geom1
    };
  };
  T M[NV * LDM];
  union { // body frames are needed until the contacts exist; per-contact inertias are folded into K
          // before the Hessian is assembled
    struct {
      T R[NB][9], p[NB][3];
      // collision-time scratch BEHIND the body frames (dead before the Newton loop writes Wc / H).  It used to overlay the
      // tree-pass scratch; here it also survives the two-wavefront variant, where the collision phase runs beside the
      // bias / mass-matrix phase that still reads f and Yc.
      int fcand[64];    // candidate pairs that passed the sphere and the box test: the narrow phase's work list
      int pcand[128];   // candidate pairs that passed the bounding-sphere test (a batch list)
      T terr[20];       // stepped-terrain table of this env
      T scen[NG][4];    // bounding spheres of the geoms in the world (relative to O): centre, radius
      T hn_s[MAXHH][3]; // narrow-phase output (staging of hn / hb1)
      int hb1_s[MAXHH];
    };
    T H[NV * LDM];
    T Wc[MAXCON][21];
  };
  T qpos[NQ], qvel[NV];
  T xv[NV];
  int cbody[MAXCON], cvert[MAXCON], cgeom[MAXCON]; // contact: body and geom of the hull (geom2), hull vertex
  T ctq[CONDIM > 3 ? MAXCON : 1][3]; // contact torque vector about the contact point (torsional friction rows)
  unsigned anc[NB]; // ancestor bitmask per body (copied from the model: LDS latency, not global)
  T cr[MAXCON][3], cdist[MAXCON], cfv[MAXCON][3]; // contact point (rel O), distance, force vector
  int xch[4]; // two-wavefront variant: ncon, nfl, cross-branch flag, flag bits handed from the collision wavefront to the other
  unsigned char cchain[MAXCON][8]; // floor contacts: the contact body and its non-root ancestors, deepest first, 0 padded
                                   // (the chain table's row of the body, kept beside the contact: static chain walks)
};
static_assert(sizeof(SimLds<double>) <= 20480, "k_sim must fit 8 workgroups per CU");

// LDS hand-over between the lanes of ONE wavefront.  With one wavefront per workgroup (NW = 1) that is what __syncthreads()
// is; in the two-wavefront variant the phases that run on different wavefronts at the same time must not meet at a
// workgroup barrier, and need none: a wavefront's LDS instructions execute in order, only the compiler has to be told.
template <int NW> __device__ __forceinline__ void wsync() {
  if constexpr (NW == 1) TSIDB_SYNC1();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

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
// DENSE: a robot<->robot contact between two branches of the tree couples dofs that are not ancestor-related;
// the same elimination then runs over all pairs, in index order.
// Two variations were measured and dropped (round 2, 4096 walkers, k_sim back to back 0.238 ms): pivots in an
// interleaved order (children before parents, branches alternating, so that consecutive pivots are independent):
// 0.242 ms - the routine is VALU-issue bound, not latency bound; broadcasts through LDS (each pivot's column written
// once, ancestors' entries read back at uniform addresses: one LDS instruction instead of the two v_readlane a
// float64 broadcast costs): 0.260 ms - the values then sit in VGPRs the kernel does not have (86 spilled) and the
// per-pivot LDS round trip is exposed.
template <typename T, bool DENSE>
__device__ __forceinline__ T chol26_solve(T (&a)[NV], T rhs, int lane, bool &spd) {
  int ln = lane; // (re-read here: the lane < k / lane == k masks are not kept from one inlined copy of this routine to the next)
  asm volatile("" : "+v"(ln));
  int notspd = 0; // (a VGPR flag pinned per pivot: the 26 compare masks kept to be and-ed at the end are 52 SGPRs)
  T rd[NV]; // 1 / U[k][k], wave-uniform
#pragma unroll
  for (int t = 0; t < NV; t++) {
    const int k = NV - 1 - t;
    const T akk = rdlane(a[k], k);
    notspd = akk > 0 ? notspd : 1;
    asm volatile("" : "+v"(notspd));
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    rd[k] = rk;
    T uik = ln < k ? a[k] * rk : (ln == k ? akk * rk : T(0));
    a[k] = uik;
    // Broadcasts four at a time: READ four (eight v_readlane), then the four FMAs.  Left alone the compiler either
    // issues the whole row's broadcasts as one burst (2 SGPRs each, parked in VGPR lanes with v_writelane / v_readlane
    // until their turn) or one at a time, v_readlane x2 - s_nop 1 - v_fma, where every FMA waits out the
    // v_readlane -> VALU hazard.  The empty asm takes the four values as SGPR inputs (so they exist before it) and
    // "changes" uik (so the FMAs and the next reads come after it) and the previous group's last result (so that group's
    // FMAs come before it).
    int jl = -1; // (compile-time after unrolling)
#pragma unroll
    for (int j0 = 0; j0 < k; j0 += 4) {
      const bool p0 = DENSE || ((MJ_DOFANC[k] >> j0) & 1u), p1 = j0 + 1 < k && (DENSE || ((MJ_DOFANC[k] >> (j0 + 1)) & 1u)),
                 p2 = j0 + 2 < k && (DENSE || ((MJ_DOFANC[k] >> (j0 + 2)) & 1u)), p3 = j0 + 3 < k && (DENSE || ((MJ_DOFANC[k] >> (j0 + 3)) & 1u));
      if (!(p0 || p1 || p2 || p3)) continue;
      const T u0 = p0 ? rdlane(uik, j0) : T(0), u1 = p1 ? rdlane(uik, j0 + 1) : T(0), u2 = p2 ? rdlane(uik, j0 + 2) : T(0),
              u3 = p3 ? rdlane(uik, j0 + 3) : T(0);
      if (jl >= 0) asm volatile("" : "+v"(uik), "+v"(a[jl]) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      else asm volatile("" : "+v"(uik) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      if (p0) { a[j0] -= uik * u0; jl = j0; }
      if (p1) { a[j0 + 1] -= uik * u1; jl = j0 + 1; }
      if (p2) { a[j0 + 2] -= uik * u2; jl = j0 + 2; }
      if (p3) { a[j0 + 3] -= uik * u3; jl = j0 + 3; }
    }
  }
  spd = notspd == 0;
  // the solves broadcast U's entries again: carried over from the factorisation (the compiler's choice) they are ~350
  // SGPRs, i.e. spilled to VGPR lanes with v_writelane and reloaded - more instructions than reading them again
#pragma unroll
  for (int k = 0; k < NV; k++) asm volatile("" : "+v"(a[k]));
  // U y = rhs (descendants first; lane k contributes y_k), then U^T x = y (ancestors first, wave-uniform)
  T acc = rhs, y[NV];
#pragma unroll
  for (int t = 0; t < NV; t++) {
    const int k = NV - 1 - t;
    y[k] = rdlane(acc, k) * rd[k];
    acc -= a[k] * y[k];
  }
  T xs[NV], x = 0;
#pragma unroll
  for (int t = NV - 1; t >= 0; t--) {
    const int k = NV - 1 - t;
    T s0 = y[k], s1 = 0, ak = a[k]; // two chains: a dependent f64 FMA waits for its predecessor
#pragma unroll
    for (int i0 = 0; i0 < k; i0 += 4) { // (broadcasts four at a time, as above)
      const bool p0 = DENSE || ((MJ_DOFANC[k] >> i0) & 1u), p1 = i0 + 1 < k && (DENSE || ((MJ_DOFANC[k] >> (i0 + 1)) & 1u)),
                 p2 = i0 + 2 < k && (DENSE || ((MJ_DOFANC[k] >> (i0 + 2)) & 1u)), p3 = i0 + 3 < k && (DENSE || ((MJ_DOFANC[k] >> (i0 + 3)) & 1u));
      if (!(p0 || p1 || p2 || p3)) continue;
      const T u0 = p0 ? rdlane(ak, i0) : T(0), u1 = p1 ? rdlane(ak, i0 + 1) : T(0), u2 = p2 ? rdlane(ak, i0 + 2) : T(0),
              u3 = p3 ? rdlane(ak, i0 + 3) : T(0);
      asm volatile("" : "+v"(ak), "+v"(s0), "+v"(s1) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      if (p0) s0 -= u0 * xs[i0];
      if (p1) s1 -= u1 * xs[i0 + 1];
      if (p2) s0 -= u2 * xs[i0 + 2];
      if (p3) s1 -= u3 * xs[i0 + 3];
    }
    xs[k] = (s0 + s1) * rd[k];
    if (ln == k) x = xs[k];
  }
  return x;
}

// the dense variant out of line: it runs only when a robot<->robot contact couples two branches of the tree, and
// inlined next to the sparse one it doubles the Newton loop's code for nothing
template <typename T>
__device__ __noinline__ T chol26_dense(T (&a)[NV], T rhs, int lane, bool &spd) { return chol26_solve<T, true>(a, rhs, lane, spd); }

// The same factorisation in three pieces, for the Newton loop, which keeps the factor from one iteration to the next:
//   chol26_factor  A = U U^T (tree-sparse, as above); 1 / U[k][k] ends up in lane k of rdv (a VGPR: 26 wave-uniform
//                  reciprocals are 52 SGPRs)
//   chol26_subst   x = A^-1 rhs from the factor
//   chol26_rank1   U U^T <- U U^T + sigma x x^T for a vector x supported on ONE root path of the tree (a constraint row's
//                  Jacobian: the dofs of the contact body's ancestors) - pivots outside the path are untouched, and the
//                  path's own pairs are all ancestor pairs, so the update stays inside the sparsity pattern.  Returns
//                  false when a downdate (sigma = -1) loses positive definiteness; the caller then rebuilds the factor.
template <typename T>
__device__ __forceinline__ void chol26_factor(T (&a)[NV], T &rdv, int lane, bool &spd) {
  int ln = lane;
  asm volatile("" : "+v"(ln));
  int notspd = 0;
  rdv = 0;
#pragma unroll
  for (int t = 0; t < NV; t++) {
    const int k = NV - 1 - t;
    const T akk = rdlane(a[k], k);
    notspd = akk > 0 ? notspd : 1;
    asm volatile("" : "+v"(notspd));
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    rdv = ln == k ? rk : rdv;
    T uik = ln < k ? a[k] * rk : (ln == k ? akk * rk : T(0));
    a[k] = uik;
    int jl = -1; // (broadcasts four at a time, read ahead of their FMAs: see chol26_solve)
#pragma unroll
    for (int j0 = 0; j0 < k; j0 += 4) {
      const bool p0 = (MJ_DOFANC[k] >> j0) & 1u, p1 = j0 + 1 < k && ((MJ_DOFANC[k] >> (j0 + 1)) & 1u),
                 p2 = j0 + 2 < k && ((MJ_DOFANC[k] >> (j0 + 2)) & 1u), p3 = j0 + 3 < k && ((MJ_DOFANC[k] >> (j0 + 3)) & 1u);
      if (!(p0 || p1 || p2 || p3)) continue;
      const T u0 = p0 ? rdlane(uik, j0) : T(0), u1 = p1 ? rdlane(uik, j0 + 1) : T(0), u2 = p2 ? rdlane(uik, j0 + 2) : T(0),
              u3 = p3 ? rdlane(uik, j0 + 3) : T(0);
      if (jl >= 0) asm volatile("" : "+v"(uik), "+v"(a[jl]) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      else asm volatile("" : "+v"(uik) : "s"(u0), "s"(u1), "s"(u2), "s"(u3));
      if (p0) { a[j0] -= uik * u0; jl = j0; }
      if (p1) { a[j0 + 1] -= uik * u1; jl = j0 + 1; }
      if (p2) { a[j0 + 2] -= uik * u2; jl = j0 + 2; }
      if (p3) { a[j0 + 3] -= uik * u3; jl = j0 + 3; }
    }
  }
  spd = notspd == 0;
#pragma unroll
  for (int k = 0; k < NV; k++) asm volatile("" : "+v"(a[k]));
}

// x = A^-1 rhs from the factor A = U U^T: `a` = row `lane` of U in registers, `Up` = the same factor parked in LDS (row stride
// LDM).  Both substitutions are column-oriented - one broadcast and one FMA per pivot, all lanes at once:
//   U y = rhs     descendants first: y_k = acc_k / U_kk, then acc_i -= U[i][k] y_k on every lane i (U[i][k] = a[k])
//   U^T x = y     ancestors first:   x_i = acc_i / U_ii, then acc_k -= U[i][k] x_i on every lane k - which needs COLUMN k
//                 of U on lane k: read back transposed from the parked copy (26 conflict-free ds_read_b64 per lane;
//                 entries below the diagonal are exact zeros).
// (The first version ran the second pass row-oriented from `a` alone: a dot product across lanes per pivot, i.e. two
//  v_readlane + one FMA per ancestor PAIR - 780 instructions against 160.  Measured and dropped, round 3: the branches'
//  pivots dealt to two accumulators, alternating in program order - two independent mul -> v_readlane -> FMA chains instead of
//  one through all 26 pivots: the Newton loop's factor + solves 39.5 k -> 38.5 k cycles over 3.3 iterations, not worth the
//  compile-time order tables.)
template <typename T>
__device__ __forceinline__ T chol26_subst(const T (&a)[NV], const T *Up, T rdv, T rhs, int lane) {
  int ln = lane;
  asm volatile("" : "+v"(ln));
  T b[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) b[i] = lane < NV ? Up[i * LDM + lane] : T(0);
  T acc = rhs, yv = 0;
#pragma unroll
  for (int t = 0; t < NV; t++) {
    const int k = NV - 1 - t;
    const T yl = acc * rdv;
    const T yk = rdlane(yl, k);
    yv = ln == k ? yl : yv;
    acc -= a[k] * yk;
  }
  acc = yv;
  T x = 0;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const T xl = acc * rdv;
    const T xi = rdlane(xl, i);
    x = ln == i ? xl : x;
    acc -= b[i] * xi;
  }
  return x;
}

template <typename T>
__device__ __forceinline__ bool chol26_rank1(T (&a)[NV], T &rdv, T xv, T sigma, unsigned long long chain, int lane) {
  // per pivot k on the path, with s = x_k / U_kk and c = sqrt(1 + sigma s^2) (the diagonal's growth factor):
  //   lanes i <= k: U[i][k] <- (U[i][k] + sigma s x_i) / c   (at i = k this is U_kk c, the new diagonal)
  //                 x_i     <- c x_i - s U[i][k]             (0 at i = k)
  // lanes > k hold zeros in both (x there was consumed by the earlier pivots, U is upper triangular)
  bool ok = true;
#pragma unroll
  for (int t = 0; t < NV; t++) {
    const int k = NV - 1 - t;
    if (ok && ((chain >> k) & 1ull)) { // (wave-uniform)
      const T s = rdlane(xv * rdv, k);
      const T q = T(1) + sigma * s * s;
      if (!(q > (sizeof(T) == 8 ? T(1e-10) : T(1e-5)))) ok = false; // downdate to (nearly) singular: rebuild instead
      else {
        const T ic = rsqrt_t(q), c = q * ic;
        const T un = (a[k] + sigma * s * xv) * ic;
        xv = lane == k ? T(0) : c * xv - s * un;
        a[k] = un;
        rdv = lane == k ? rdv * ic : rdv;
      }
    }
  }
  return ok;
}

// out = M * x for the lane's dof (x in LDS)
template <typename T> __device__ __forceinline__ T mulM(const SimLds<T> &L, const T *x, int lane) {
  T s = 0, s1 = 0, s2 = 0, s3 = 0; // (four partial sums: a dependent float64 FMA waits for its predecessor)
  if (lane < NV) {
    const T *Mr = &L.M[lane * LDM];
#pragma unroll
    for (int j = 0; j < NV; j += 4) {
      s += Mr[j] * x[j];
      if (j + 1 < NV) s1 += Mr[j + 1] * x[j + 1];
      if (j + 2 < NV) s2 += Mr[j + 2] * x[j + 2];
      if (j + 3 < NV) s3 += Mr[j + 3] * x[j + 3];
    }
  }
  return (s + s1) + (s2 + s3);
}

// mju_makeFrame: tangents of a contact frame from its normal (t1 from y unless |n_y| >= 0.5, t2 = n x t1)
template <typename T> __device__ __forceinline__ void make_frame(const T *n, T *t1, T *t2) {
  T t[3] = {0, 0, 0};
  if (fabs(n[1]) < T(0.5)) t[1] = 1; else t[2] = 1;
  const T dn = dot3(n, t);
  T nn = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) { t1[i] = t[i] - dn * n[i]; nn += t1[i] * t1[i]; }
  nn = T(1) / sqrt(nn);
#pragma unroll
  for (int i = 0; i < 3; i++) t1[i] *= nn;
  cross3(n, t1, t2);
}

// frame (normal, tangents) and geom1 body of contact c: floor contacts [0, nfl) share the floor frame and have
// no geom1 body (-1); robot<->robot contacts keep their normal in LDS and rebuild the tangents
template <typename T>
__device__ __forceinline__ int contact_frame(const SimLds<T> &L, int c, int nfl, T *n, T *t1, T *t2) {
  if (c < nfl) {
#pragma unroll
    for (int i = 0; i < 3; i++) { n[i] = L.fl.n[i]; t1[i] = L.fl.t1[i]; t2[i] = L.fl.t2[i]; }
    return -1;
  }
#pragma unroll
  for (int i = 0; i < 3; i++) n[i] = L.hn[c - nfl][i];
  make_frame(n, t1, t2);
  return L.hb1[c - nfl];
}

// the 4 pyramid rows of contact `c` applied to generalized vector x (LDS): J_row x, with J = point Jacobian of
// geom2's body minus that of geom1's body (the floor does not move)
template <typename T>
__device__ __forceinline__ void contact_rows(const DevModel<T> &m, const SimLds<T> &L, int nfl, int c, const T *x,
                                             T mu, T *out) {
  // rows 0..3: normal +- mu * tangent_k (sliding friction); CONDIM 4 adds rows 4, 5: normal +- mu_t * (relative
  // angular velocity about the normal) - torsional friction (robot/v0/robot.xml:4)
  T n[3], t1[3], t2[3];
  const int b1 = contact_frame(L, c, nfl, n, t1, t2);
  T tw[6] = {0, 0, 0, 0, 0, 0};
  if (b1 < 0) {
    // floor contact: the root's six dofs, then the body's chain from the shallowest ancestor down (the bytes stored with the
    // contact) - a static walk, every load independent; the same order of sums as the bit-mask loop below
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const T xk = x[k];
#pragma unroll
      for (int i = 0; i < 6; i++) tw[i] += L.S[k][i] * xk;
    }
#pragma unroll
    for (int d = 6; d >= 0; d--) {
      const int a = L.cchain[c][d];
      if (a > 0) {
        const T xk = x[5 + a];
#pragma unroll
        for (int i = 0; i < 6; i++) tw[i] += L.S[5 + a][i] * xk;
      }
    }
  } else {
  const unsigned m2 = L.anc[L.cbody[c]], m1 = b1 >= 0 ? L.anc[b1] : 0u;
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const T sg = side == 0 ? T(1) : T(-1);
    for (unsigned mk = side == 0 ? (m2 & ~m1) : (m1 & ~m2); mk; mk &= mk - 1) {
      const int a = __ffs(mk) - 1;
      if (a == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const T xk = sg * x[k];
#pragma unroll
          for (int i = 0; i < 6; i++) tw[i] += L.S[k][i] * xk;
        }
      } else {
        const T xk = sg * x[5 + a];
#pragma unroll
        for (int i = 0; i < 6; i++) tw[i] += L.S[5 + a][i] * xk;
      }
    }
  }
  }
  T wxr[3];
  cross3(tw + 3, L.cr[c], wxr);
  const T u[3] = {tw[0] + wxr[0], tw[1] + wxr[1], tw[2] + wxr[2]};
  const T un = dot3(n, u), u1 = dot3(t1, u), u2 = dot3(t2, u);
  out[0] = un + mu * u1; out[1] = un - mu * u1; out[2] = un + mu * u2; out[3] = un - mu * u2;
  if constexpr (CONDIM > 3) {
    const T ut = m.contact[9] * dot3(n, tw + 3);
    out[4] = un + ut; out[5] = un - ut;
  }
}

// the same for K generalized vectors at once: ONE walk over the contact's body chains (the motion vectors S are read once),
// K twists side by side.  Used where the rows are needed for several vectors from the same state: joint velocities (the
// reference acceleration), the warm start and the unconstrained acceleration - three walks before.
template <typename T, int K>
__device__ __forceinline__ void contact_rows_multi(const DevModel<T> &m, const SimLds<T> &L, int nfl, int c, const T *const (&x)[K],
                                                   T mu, T (&out)[K][NROWC]) {
  T n[3], t1[3], t2[3];
  const int b1 = contact_frame(L, c, nfl, n, t1, t2);
  T tw[K][6];
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int i = 0; i < 6; i++) tw[k][i] = 0;
  auto add_dof = [&](int d, T sg) {
    T Sd[6];
#pragma unroll
    for (int i = 0; i < 6; i++) Sd[i] = L.S[d][i];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const T xk = sg * x[k][d];
#pragma unroll
      for (int i = 0; i < 6; i++) tw[k][i] += Sd[i] * xk;
    }
  };
  if (b1 < 0) { // floor contact: static walk over the chain stored with the contact (see contact_rows)
#pragma unroll
    for (int d = 0; d < 6; d++) add_dof(d, T(1));
#pragma unroll
    for (int d = 6; d >= 0; d--) {
      const int a = L.cchain[c][d];
      if (a > 0) add_dof(5 + a, T(1));
    }
  } else {
    const unsigned m2 = L.anc[L.cbody[c]], m1 = L.anc[b1];
#pragma unroll
    for (int side = 0; side < 2; side++) {
      const T sg = side == 0 ? T(1) : T(-1);
      for (unsigned mk = side == 0 ? (m2 & ~m1) : (m1 & ~m2); mk; mk &= mk - 1) {
        const int a = __ffs(mk) - 1;
        const int d0 = a == 0 ? 0 : 5 + a, d1 = a == 0 ? 5 : 5 + a;
        for (int d = d0; d <= d1; d++) add_dof(d, sg);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    T wxr[3];
    cross3(tw[k] + 3, L.cr[c], wxr);
    const T u[3] = {tw[k][0] + wxr[0], tw[k][1] + wxr[1], tw[k][2] + wxr[2]};
    const T un = dot3(n, u), u1 = dot3(t1, u), u2 = dot3(t2, u);
    out[k][0] = un + mu * u1; out[k][1] = un - mu * u1; out[k][2] = un + mu * u2; out[k][3] = un - mu * u2;
    if constexpr (CONDIM > 3) {
      const T ut = m.contact[9] * dot3(n, tw[k] + 3);
      out[k][4] = un + ut; out[k][5] = un - ut;
    }
  }
}

// ------------------------------------------------------------------ collision helpers
// stepped terrain (BASELINE.json configs[4]; no reference counterpart): the floor surface is the plane n.x = d
// raised along n by heights[cell & 15], cell = floor((dir . x_world_xy - phase) * inv_len).
// Table: dir_x, dir_y, phase, inv_len, heights[16] (staged in LDS for the collision phase).
template <typename T> __device__ __forceinline__ T terrain_h(const T *terr, T X, T Y) {
  const T u = terr[0] * X + terr[1] * Y;
  const int cell = (int)floor((u - terr[2]) * terr[3]) & 15;
  return terr[4 + cell];
}

// Lowest-index vertex of geom b's hull within `tie` of the minimum of  r . v + pz  [- terrain height under the
// vertex], v in the body frame.  Exact pruned search: the hull's vertices are stored in k-d order, 64 per chunk,
// each chunk with a bounding box; a chunk can hold the minimum (or a vertex within the tie tolerance of it) only
// if its lower bound does not exceed the best value found so far, and every such chunk is scanned, so the result
// equals the exhaustive search.  TERR: Rw = the body's world rotation, (px, py) = its world position; the lower
// bound then subtracts the highest terrain cell the chunk's box can reach.
template <typename T, bool TERR>
__device__ __forceinline__ int hull_argmin(const DevModel<T> &m, int lane, int b, T r6, T r7, T r8, T pz, T tie, const T *terr,
                                           const T *Rw, T px, T py, T hmax_all, T &zmin_out) {
  const T INF = Eps<T>::inf;
  const int v0 = m.hull_adr[b], v1 = m.hull_adr[b + 1];
  const int c0 = m.chunk_adr[b], nch = m.chunk_adr[b + 1] - c0;
  T zlb = INF;
  if (lane < nch) {
    const T *bx = m.chunk_box + 6 * (c0 + lane);
    zlb = r6 * bx[0] + r7 * bx[1] + r8 * bx[2] + pz - (fabs(r6) * bx[3] + fabs(r7) * bx[4] + fabs(r8) * bx[5]);
    zlb -= fabs(zlb) * T(4) * Eps<T>::v; // rounding slack: never prune a chunk that could matter
    if constexpr (TERR) {
      // terrain cells the box can reach along the step direction
      const T a0 = terr[0] * Rw[0] + terr[1] * Rw[3], a1 = terr[0] * Rw[1] + terr[1] * Rw[4], a2 = terr[0] * Rw[2] + terr[1] * Rw[5];
      const T uc = a0 * bx[0] + a1 * bx[1] + a2 * bx[2] + terr[0] * px + terr[1] * py;
      const T ext = fabs(a0) * bx[3] + fabs(a1) * bx[4] + fabs(a2) * bx[5] + T(1e-6);
      const int k0 = (int)floor((uc - ext - terr[2]) * terr[3]), k1 = (int)floor((uc + ext - terr[2]) * terr[3]);
      T hm = hmax_all;
      if (k1 - k0 < 15) {
        hm = terr[4 + (k0 & 15)];
        for (int k = k0 + 1; k <= k1; k++) { const T hk = terr[4 + (k & 15)]; hm = hk > hm ? hk : hm; }
      }
      zlb -= hm;
    }
  }
  auto vert_val = [&](int i) -> T {
    const T x = m.hull_x[i], y = m.hull_y[i], z = m.hull_z[i];
    T val = r6 * x + r7 * y + r8 * z + pz;
    if constexpr (TERR) val -= terrain_h(terr, Rw[0] * x + Rw[1] * y + Rw[2] * z + px, Rw[3] * x + Rw[4] * y + Rw[5] * z + py);
    return val;
  };
  unsigned long long scanned = 0;
  T zmin = INF;
  T zk0 = INF, zk1 = INF; // this lane's values in the two chunks scanned last (the tie search below reads them again:
  int ck0 = -1, ck1 = -1; //  a foot is decided in one or two chunks, and the reload is a global-memory round trip)
  {
    T zl = zlb;
    int ci = lane;
    wave_argmin(zl, ci);
    unsigned long long pend = 1ull << ci;
    while (pend) {
      const int c = __ffsll((long long)pend) - 1;
      const int i = v0 + WAVE * c + lane;
      T z = INF;
      if (i < v1) z = vert_val(i);
      zk1 = zk0; ck1 = ck0; zk0 = z; ck0 = c;
      const T zc2 = wave_min(z);
      zmin = zc2 < zmin ? zc2 : zmin;
      scanned |= 1ull << c;
      pend = __ballot(lane < nch && zlb <= zmin + tie) & ~scanned;
    }
  }
  zmin_out = zmin;
  // lowest-index vertex within the tie tolerance of the minimum (chunks are in index order)
  int best = 0x7fffffff;
  const T zt = zmin + tie;
  for (unsigned long long sm = scanned; sm && best == 0x7fffffff; sm &= sm - 1) {
    const int c = __ffsll((long long)sm) - 1;
    const int i = v0 + WAVE * c + lane;
    T zv;
    if (c == ck0) zv = zk0;
    else if (c == ck1) zv = zk1;
    else zv = i < v1 ? vert_val(i) : INF;
    int cand = 0x7fffffff;
    if (i < v1 && zv <= zt) cand = i;
    best = wave_min_int(cand);
  }
  // non-finite values (a NaN placement) match nothing: hand back a valid index, the callers index global memory with it
  return best == 0x7fffffff ? v0 : best;
}

template <typename T> __device__ __forceinline__ bool normalize3(T *a) {
  const T n2 = dot3(a, a);
  if (!(n2 > 0)) return false;
  const T r = T(1) / sqrt(n2);
  a[0] *= r; a[1] *= r; a[2] *= r;
  return true;
}

// squared distance from the origin to triangle (a, b, c) and the closest point (Ericson 5.1.5)
template <typename T> __device__ __forceinline__ T origin_tri_closest(const T *a, const T *b, const T *c, T *q) {
  T ab[3], ac[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; }
  const T d1 = -dot3(ab, a), d2 = -dot3(ac, a);
  if (d1 <= 0 && d2 <= 0) { q[0] = a[0]; q[1] = a[1]; q[2] = a[2]; return dot3(q, q); }
  const T d3 = -dot3(ab, b), d4 = -dot3(ac, b);
  if (d3 >= 0 && d4 <= d3) { q[0] = b[0]; q[1] = b[1]; q[2] = b[2]; return dot3(q, q); }
  const T vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) {
    const T t = d1 / (d1 - d3);
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = a[i] + t * ab[i];
    return dot3(q, q);
  }
  const T d5 = -dot3(ab, c), d6 = -dot3(ac, c);
  if (d6 >= 0 && d5 <= d6) { q[0] = c[0]; q[1] = c[1]; q[2] = c[2]; return dot3(q, q); }
  const T vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) {
    const T t = d2 / (d2 - d6);
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = a[i] + t * ac[i];
    return dot3(q, q);
  }
  const T va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    const T t = (d4 - d3) / ((d4 - d3) + (d5 - d6));
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = b[i] + t * (c[i] - b[i]);
    return dot3(q, q);
  }
  const T den = T(1) / (va + vb + vc), v = vb * den, w = vc * den;
#pragma unroll
  for (int i = 0; i < 3; i++) q[i] = a[i] + ab[i] * v + ac[i] * w;
  return dot3(q, q);
}

// Penetration of the convex hulls of bodies a (geom1) and b (geom2): Minkowski Portal Refinement on A - B, what
// MuJoCo's mjc_Convex runs for mesh pairs (libccd ccdMPRPenetration; Snethen 2008) - portal discovery from the
// interior point centre(a) - centre(b), refinement until the portal stops advancing, depth = distance from the
// origin to the final portal, contact point from the portal's barycentric weights; one contact per pair.
// The scalar state is wave-uniform (every lane computes it); the wave's parallelism goes into the two hull
// support searches per step (hull_argmin: chunk bounds on the lanes, then 64 vertices per scan).
// Placements L.R / L.p are relative to the base origin O; so is pos.  Returns true with depth >= 0, unit dir
// (geom1 -> geom2) and pos.
// a, b: geom indices; their hulls are placed by the bodies that carry them.  margin > 0 (robot/v0/robot.xml:4): each
// hull is inflated by margin / 2 along the support direction (as mjc_Convex does), the caller takes dist = margin - depth.
template <typename T, bool TERR> // (the packed kernels' support search, tsidb_sim2.hpp)
__device__ __forceinline__ int hull_argmin_p(const DevModel<T> &m, int lane, int b, T r6, T r7, T r8, T pz, T tie, const T *terr,
                                             const T *Rw, T px, T py, T hmax_all, T &zmin_out);
template <typename T, bool PACK = false>
__device__ __forceinline__ bool mpr_penetration(const DevModel<T> &m, const SimLds<T> &L, int lane, int a, int b, T margin, T &depth, T *dir_out, T *pos) {
  const T TOL = sizeof(T) == 8 ? T(1e-10) : T(2e-6), TIE = sizeof(T) == 8 ? T(1e-12) : T(1e-7);
  const T TINY2 = sizeof(T) == 8 ? T(1e-30) : T(1e-20), SIDE = sizeof(T) == 8 ? T(1e-14) : T(1e-9);
  constexpr int MAXIT = 64;
  const int ba = m.geom_body[a], bb = m.geom_body[b];
  const T *Ra = L.R[ba], *Rb = L.R[bb], *pa = L.p[ba], *pb = L.p[bb];
  auto wvert = [&](const T *R, const T *p, int i, T *w) {
    const T v[3] = {m.hull_x[i], m.hull_y[i], m.hull_z[i]};
    mat3vec(R, v, w);
    w[0] += p[0]; w[1] += p[1]; w[2] += p[2];
  };
  // support point of A - B along d: vertex of A farthest along d minus vertex of B farthest along -d
  auto support = [&](const T *d, T *v, int &ia, int &ib) {
    T ra[3], rb[3], zz, wa[3], wb[3];
    mat3Tvec(Ra, d, ra);
    mat3Tvec(Rb, d, rb);
    if constexpr (PACK) {
      ia = hull_argmin_p<T, false>(m, lane, a, -ra[0], -ra[1], -ra[2], T(0), TIE, nullptr, nullptr, T(0), T(0), T(0), zz);
      ib = hull_argmin_p<T, false>(m, lane, b, rb[0], rb[1], rb[2], T(0), TIE, nullptr, nullptr, T(0), T(0), T(0), zz);
    } else {
    ia = hull_argmin<T, false>(m, lane, a, -ra[0], -ra[1], -ra[2], T(0), TIE, nullptr, nullptr, T(0), T(0), T(0), zz);
    ib = hull_argmin<T, false>(m, lane, b, rb[0], rb[1], rb[2], T(0), TIE, nullptr, nullptr, T(0), T(0), T(0), zz);
    }
    wvert(Ra, pa, ia, wa);
    wvert(Rb, pb, ib, wb);
    v[0] = wa[0] - wb[0]; v[1] = wa[1] - wb[1]; v[2] = wa[2] - wb[2];
    if (margin != 0) { // both hulls inflated by margin / 2 along d (d is a unit vector wherever support is called)
      v[0] += margin * d[0]; v[1] += margin * d[1]; v[2] += margin * d[2];
    }
  };
  T ca[3], cb[3], v0[3], v1[3], v2[3], v3[3], v4[3], dir[3], e1[3], e2[3];
  int i1a = 0, i1b = 0, i2a = 0, i2b = 0, i3a = 0, i3b = 0, i4a = 0, i4b = 0;
  mat3vec(Ra, m.hcen[a], ca);
  mat3vec(Rb, m.hcen[b], cb);
#pragma unroll
  for (int i = 0; i < 3; i++) { ca[i] += pa[i]; cb[i] += pb[i]; v0[i] = ca[i] - cb[i]; }
  if (dot3(v0, v0) < TINY2) { v0[0] = sizeof(T) == 8 ? T(1e-10) : T(1e-6); v0[1] = 0; v0[2] = 0; }
  // ---- portal discovery
#pragma unroll
  for (int i = 0; i < 3; i++) dir[i] = -v0[i];
  normalize3(dir);
  support(dir, v1, i1a, i1b);
  if (dot3(v1, dir) <= 0) return false;
  cross3(v0, v1, dir);
  if (dot3(dir, dir) < TINY2) { // the origin lies on the ray v0 -> v1: penetration along that ray
    const T d1 = sqrt(dot3(v1, v1));
    if (!(d1 > 0)) return false;
    depth = d1;
    T wa[3], wb[3];
    wvert(Ra, pa, i1a, wa);
    wvert(Rb, pb, i1b, wb);
#pragma unroll
    for (int i = 0; i < 3; i++) { dir_out[i] = v1[i] / d1; pos[i] = T(0.5) * (wa[i] + wb[i]); }
    return true;
  }
  normalize3(dir);
  support(dir, v2, i2a, i2b);
  if (dot3(v2, dir) <= 0) return false;
#pragma unroll
  for (int i = 0; i < 3; i++) { e1[i] = v1[i] - v0[i]; e2[i] = v2[i] - v0[i]; }
  cross3(e1, e2, dir);
  normalize3(dir);
  if (dot3(dir, v0) > 0) { // orient the portal normal away from v0
#pragma unroll
    for (int i = 0; i < 3; i++) { const T t = v1[i]; v1[i] = v2[i]; v2[i] = t; dir[i] = -dir[i]; }
    int t = i1a; i1a = i2a; i2a = t;
    t = i1b; i1b = i2b; i2b = t;
  }
  for (int it = 0;; it++) {
    if (it > MAXIT) return false;
    support(dir, v3, i3a, i3b);
    if (dot3(v3, dir) <= 0) return false;
    bool cont = false;
    cross3(v1, v3, e1);
    if (dot3(e1, v0) < -SIDE) { // origin outside (v1, v0, v3)
#pragma unroll
      for (int i = 0; i < 3; i++) v2[i] = v3[i];
      i2a = i3a; i2b = i3b;
      cont = true;
    }
    if (!cont) {
      cross3(v3, v2, e1);
      if (dot3(e1, v0) < -SIDE) { // origin outside (v3, v0, v2)
#pragma unroll
        for (int i = 0; i < 3; i++) v1[i] = v3[i];
        i1a = i3a; i1b = i3b;
        cont = true;
      }
    }
    if (!cont) break;
#pragma unroll
    for (int i = 0; i < 3; i++) { e1[i] = v1[i] - v0[i]; e2[i] = v2[i] - v0[i]; }
    cross3(e1, e2, dir);
    normalize3(dir);
  }
  // ---- portal refinement until the portal reaches the surface of A - B; the origin must end up inside it
  bool inside = false;
  for (int it = 0;; it++) {
#pragma unroll
    for (int i = 0; i < 3; i++) { e1[i] = v2[i] - v1[i]; e2[i] = v3[i] - v1[i]; }
    cross3(e1, e2, dir);
    normalize3(dir);
    if (dot3(dir, v1) >= 0) inside = true;
    support(dir, v4, i4a, i4b);
    const T d4 = dot3(v4, dir);
    T adv = d4 - dot3(v1, dir);
    const T adv2 = d4 - dot3(v2, dir), adv3 = d4 - dot3(v3, dir);
    adv = adv2 < adv ? adv2 : adv;
    adv = adv3 < adv ? adv3 : adv;
    if (!inside && d4 < 0) return false; // the origin is beyond the support plane: separated
    if (adv <= TOL || it >= MAXIT) {
      if (!inside) return false;
      break;
    }
    // expand the portal with v4: replace the vertex that keeps the origin's ray inside
    cross3(v4, v0, e1);
    if (dot3(v1, e1) > 0) {
      if (dot3(v2, e1) > 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) v1[i] = v4[i];
        i1a = i4a; i1b = i4b;
      } else {
#pragma unroll
        for (int i = 0; i < 3; i++) v3[i] = v4[i];
        i3a = i4a; i3b = i4b;
      }
    } else {
      if (dot3(v3, e1) > 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) v2[i] = v4[i];
        i2a = i4a; i2b = i4b;
      } else {
#pragma unroll
        for (int i = 0; i < 3; i++) v1[i] = v4[i];
        i1a = i4a; i1b = i4b;
      }
    }
  }
  // ---- penetration: closest point of the final portal to the origin
  T q[3];
  const T d2 = origin_tri_closest(v1, v2, v3, q);
  depth = sqrt(d2);
#pragma unroll
  for (int i = 0; i < 3; i++) { e1[i] = v2[i] - v1[i]; e2[i] = v3[i] - v1[i]; }
  cross3(e1, e2, dir);
  normalize3(dir);
  if (depth > SIDE) {
#pragma unroll
    for (int i = 0; i < 3; i++) dir_out[i] = q[i] / depth;
  } else {
#pragma unroll
    for (int i = 0; i < 3; i++) dir_out[i] = dir[i];
  }
  // ---- contact point: barycentric weights of the origin's ray in the portal
  T bw[4], t1[3];
  cross3(v1, v2, t1); bw[0] = dot3(t1, v3);
  cross3(v3, v2, t1); bw[1] = dot3(t1, v0);
  cross3(v0, v1, t1); bw[2] = dot3(t1, v3);
  cross3(v2, v1, t1); bw[3] = dot3(t1, v0);
  T sum = bw[0] + bw[1] + bw[2] + bw[3];
  if (sum <= 0) {
    bw[0] = 0;
    cross3(v2, v3, t1); bw[1] = dot3(t1, dir);
    cross3(v3, v1, t1); bw[2] = dot3(t1, dir);
    cross3(v1, v2, t1); bw[3] = dot3(t1, dir);
    sum = bw[1] + bw[2] + bw[3];
  }
  T p1[3], p2[3], wa[3], wb[3];
#pragma unroll
  for (int i = 0; i < 3; i++) { p1[i] = bw[0] * ca[i]; p2[i] = bw[0] * cb[i]; }
  wvert(Ra, pa, i1a, wa); wvert(Rb, pb, i1b, wb);
#pragma unroll
  for (int i = 0; i < 3; i++) { p1[i] += bw[1] * wa[i]; p2[i] += bw[1] * wb[i]; }
  wvert(Ra, pa, i2a, wa); wvert(Rb, pb, i2b, wb);
#pragma unroll
  for (int i = 0; i < 3; i++) { p1[i] += bw[2] * wa[i]; p2[i] += bw[2] * wb[i]; }
  wvert(Ra, pa, i3a, wa); wvert(Rb, pb, i3b, wb);
#pragma unroll
  for (int i = 0; i < 3; i++) { p1[i] += bw[3] * wa[i]; p2[i] += bw[3] * wb[i]; }
#pragma unroll
  for (int i = 0; i < 3; i++) pos[i] = T(0.5) * (p1[i] + p2[i]) / sum;
  return true;
}

// mid phase for one candidate pair (one lane per pair): bounding spheres, then the 15-axis separating-axis test
// on the hulls' body-frame boxes
template <typename T>
__device__ __forceinline__ bool spheres_overlap(const SimLds<T> &L, int a, int b, T margin) {
  const T d[3] = {L.scen[a][0] - L.scen[b][0], L.scen[a][1] - L.scen[b][1], L.scen[a][2] - L.scen[b][2]};
  const T rr = L.scen[a][3] + L.scen[b][3] + margin;
  return !(dot3(d, d) > rr * rr);
}
template <typename T>
__device__ __forceinline__ bool boxes_may_touch(const DevModel<T> &m, const SimLds<T> &L, int a, int b, T margin) {
  const int ba = m.geom_body[a], bb = m.geom_body[b]; // a, b: geoms
  const T *Ra = L.R[ba], *Rb = L.R[bb];
  T ca[3], cb[3], d[3];
  mat3vec(Ra, m.hbox[a], ca);
  mat3vec(Rb, m.hbox[b], cb);
#pragma unroll
  for (int i = 0; i < 3; i++) d[i] = (cb[i] + L.p[bb][i]) - (ca[i] + L.p[ba][i]);
  const T hm = T(0.5) * margin; // contacts are made within the margin: both boxes grow by half of it
  const T ha[3] = {m.hbox[a][3] + hm, m.hbox[a][4] + hm, m.hbox[a][5] + hm}, hb[3] = {m.hbox[b][3] + hm, m.hbox[b][4] + hm, m.hbox[b][5] + hm};
  T Rm[3][3], A[3][3], t[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    t[i] = Ra[i] * d[0] + Ra[3 + i] * d[1] + Ra[6 + i] * d[2];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Rm[i][j] = Ra[i] * Rb[j] + Ra[3 + i] * Rb[3 + j] + Ra[6 + i] * Rb[6 + j];
      A[i][j] = fabs(Rm[i][j]) + (sizeof(T) == 8 ? T(1e-12) : T(1e-6));
    }
  }
  bool sep = false;
#pragma unroll
  for (int i = 0; i < 3; i++) sep |= fabs(t[i]) > ha[i] + A[i][0] * hb[0] + A[i][1] * hb[1] + A[i][2] * hb[2];
#pragma unroll
  for (int j = 0; j < 3; j++)
    sep |= fabs(t[0] * Rm[0][j] + t[1] * Rm[1][j] + t[2] * Rm[2][j]) > ha[0] * A[0][j] + ha[1] * A[1][j] + ha[2] * A[2][j] + hb[j];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const T ra = ha[i1] * A[i2][j] + ha[i2] * A[i1][j], rb = hb[j1] * A[i][j2] + hb[j2] * A[i][j1];
      sep |= fabs(t[i2] * Rm[i1][j] - t[i1] * Rm[i2][j]) > ra + rb;
    }
  return !sep;
}

template <typename T> __device__ __forceinline__ unsigned bodyanc_of(const DevModel<T> &m, int b) { return m.mj_anc[b]; }

// The linear algebra of one Newton iteration, out of line:  search direction = H^-1 grad  from the LDS copy of H.
//   mode 0  L.H holds a freshly assembled Hessian: factor it (tree-sparse U U^T) and park the factor in its place
//           (column 26: 1 / diagonal);
//   mode 1  L.H holds the factor parked by an earlier iteration: apply the constraint rows whose state changed since
//           (bits of `chg`: bit 0 = this lane's friction row, bit 1 + i = row i of this lane's contact; `actbits` = the new
//           states) as rank-1 updates (row entered its quadratic zone) / downdates (left it) with sqrt(D_r) J_r, park it
//           again.  The row's Jacobian is not materialised elsewhere either: entry k = S_k,lin . dir + S_k,ang . (r x dir
//           [+ torsional part]) with the sign of the body chain dof k is on.
// then the two substitutions.  status: 0 = fine, 1 = the Hessian is not positive definite, 2 = a downdate lost
// definiteness (the caller assembles the Hessian and comes back with mode 0).
// Out of line on purpose: inlined, the factor's 26 rows and the row construction are register-allocated together with
// the Newton loop, and the kernel - at the 256-VGPR limit of two wavefronts per SIMD - spilled two dozen loop-carried
// values to scratch, which cost more than the row updates saved (measured: line search +40 %, every phase +5 %).
template <typename T> struct NewtonDir { T search; int status; };
template <typename T>
__device__ __noinline__ NewtonDir<T> newton_direction(const DevModel<T> &m, SimLds<T> &L, int nfl, int mode, unsigned chg, unsigned actbits,
                                                      T mu, T cD, T fD, T grad) {
  const int lane = threadIdx.x & (WAVE - 1);
  NewtonDir<T> out;
  out.search = 0;
  out.status = 0;
  T arow[NV], rdv = 0;
#pragma unroll
  for (int j = 0; j < NV; j++) arow[j] = lane < NV ? L.H[lane * LDM + j] : T(0);
  bool store = true;
  if (mode == 0) {
    bool spd;
    chol26_factor<T>(arow, rdv, lane, spd);
    if (!spd) { out.status = 1; return out; }
  } else {
    rdv = lane < NV ? L.H[lane * LDM + NV] : T(0);
    store = __ballot(chg != 0) != 0ull;
    while (true) {
      const unsigned long long mk = __ballot(chg != 0);
      if (!mk) break;
      const int src = __ffsll((long long)mk) - 1;
      const unsigned bits = (unsigned)__builtin_amdgcn_readlane((int)chg, src);
      const int bit = __ffs(bits) - 1;
      if (lane == src) chg &= ~(1u << bit);
      const T sigma = (((unsigned)__builtin_amdgcn_readlane((int)actbits, src) >> bit) & 1u) ? T(1) : T(-1);
      const int bk = lane < 6 ? 0 : lane - 5; // body of this lane's dof
      T xv = 0;
      unsigned pathm; // bodies on the root path that carries the row's Jacobian
      if (bit == 0) { // friction row of dof `src`: sqrt(D) e_src
        const T d = rdlane_dyn(fD, src);
        xv = lane == src ? d * rsqrt_t(d) : T(0);
        pathm = L.anc[src < 6 ? 0 : src - 5];
        if (lane > src) pathm = 0;
      } else { // row bit - 1 of contact `src`
        const int c = src, i = bit - 1;
        T cn[3], ct1[3], ct2[3];
        const int b1 = contact_frame(L, c, nfl, cn, ct1, ct2);
        const T muc = rdlane_dyn(mu, c), d = rdlane_dyn(cD, c);
        const T sg = i >= 4 ? T(0) : ((i & 1) ? -muc : muc);
        const T *tk = i < 2 ? ct1 : ct2;
        const T dl[3] = {cn[0] + sg * tk[0], cn[1] + sg * tk[1], cn[2] + sg * tk[2]};
        T rxd[3];
        cross3(L.cr[c], dl, rxd);
        if constexpr (CONDIM > 3) {
          if (i >= 4) {
            const T mt = (i & 1) ? -m.contact[9] : m.contact[9];
            rxd[0] += mt * cn[0]; rxd[1] += mt * cn[1]; rxd[2] += mt * cn[2];
          }
        }
        const unsigned m2 = L.anc[L.cbody[c]], m1 = b1 >= 0 ? L.anc[b1] : 0u;
        pathm = m2 | m1;
        if (lane < NV) {
          const int sgn = (int)((m2 >> bk) & 1u) - (int)((m1 >> bk) & 1u);
          if (sgn != 0) {
            const T *Sk = L.S[lane];
            const T jv = Sk[0] * dl[0] + Sk[1] * dl[1] + Sk[2] * dl[2] + Sk[3] * rxd[0] + Sk[4] * rxd[1] + Sk[5] * rxd[2];
            const T sd = d * rsqrt_t(d);
            xv = sgn > 0 ? sd * jv : -(sd * jv);
          }
        }
      }
      const unsigned long long chain = __ballot(lane < NV && ((pathm >> bk) & 1u));
      if (!chol26_rank1<T>(arow, rdv, xv, sigma, chain, lane)) { out.status = 2; return out; }
    }
  }
  if (store && lane < NV) {
#pragma unroll
    for (int j = 0; j < NV; j++) L.H[lane * LDM + j] = arow[j];
    L.H[lane * LDM + NV] = rdv;
  }
  wsync<2>(); // (wavefront-local: orders the parked rows before the transposed read; only one wavefront is ever in here)
  out.search = chol26_subst<T>(arow, L.H, rdv, grad, lane);
  return out;
}

// per-lane constraint bookkeeping (friction row of dof `lane`, contact `lane`)
template <typename T>
struct RowState {
  // friction dof row
  bool has_f;
  T fD, fR, floss, faref, fjar, fJv;
  // contact rows
  bool has_c;
  T cD, caref[NROWC], cjar[NROWC], cJv[NROWC];
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
    for (int i = 0; i < NROWC; i++) {
      const T x = rs.cjar[i] + alpha * rs.cJv[i];
      if (x < 0) { c += T(0.5) * rs.cD * x * x; g += rs.cD * x * rs.cJv[i]; h += rs.cD * rs.cJv[i] * rs.cJv[i]; }
    }
  }
}

// NW = wavefronts per env.  1: the batch fills the GPU, one wavefront per env is the throughput-optimal shape.  2 (small
// batches, where a step costs one wavefront's LATENCY and half the SIMDs idle): after the kinematics - which both wavefronts
// run, redundantly and in lockstep - wavefront 1 does the collision phase while wavefront 0 builds bias forces and the mass
// matrix, factors it and solves for the unconstrained acceleration; they join before the constraint rows, and wavefront 0
// finishes the step alone.  Same operations on the same data: results are bit-identical to NW = 1.
template <typename T, int NW>
__device__ __forceinline__ void sim_step_env(const DevModel<T> &m, SimLds<T> &L, int lane, int wv, const T *q_tsid, const T *v_tsid, T *qpos_g, T *qvel_g,
                             T *qacc_ws_g, const T *envp, const T *terr_g, const T *motor_tau, T *qacc_out, int *ncon_out, int *con_out,
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
  const T gz = m.opt[1];
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
  wg_sync<NW>();
  const T Oz = L.qpos[2];

  // ---- kinematics, velocities, bias accelerations: parent-independent work up front, the depth
  //      loop carries R, p, V, A down the tree, body inertias/forces in one parallel pass
  T Rb[9], pb[3], Vb[6], Ab[6], Sb[6], qd = 0;
  const int up0 = lane < NB ? m.mj_up[0][lane] : -1, up1 = lane < NB ? m.mj_up[1][lane] : -1,
            up2 = lane < NB ? m.mj_up[2][lane] : -1;
  const unsigned bodyanc = lane < NB ? m.mj_anc[lane] : 0u;
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
  int bchn[7];
#pragma unroll
  for (int d = 0; d < 7; d++) bchn[d] = lane < NB ? m.mj_chain[lane][d] : -1;
  tree_forward<T, NW>(lane, NB, up0, up1, up2, bchn, &L.R[0][0], &L.V[0][0], &L.f[0][0], 6, &L.Yc[0][0], 10, Rb, pb, qd, Sb,
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
  wg_sync<NW>();
  const bool w_dyn = NW == 1 || wv == 0, w_col = NW == 1 || wv == NW - 1; // which phases this wavefront runs
  // ---- per dof: bias, mass-matrix column (+ armature), actuation
  T qfs = 0;
  if (w_dyn) {
  if (lane < NV) {
    const int k = lane, bk = k < 6 ? 0 : k - 5;
    T Sk[6], Fk[6], hk = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { Sk[i] = L.S[k][i]; hk += Sk[i] * L.f[bk][i]; }
    yo_mul(L.Yc[bk], Sk, Fk);
    // (a static walk over the chain table - the body's non-root ancestors, then the root's six dofs: every load independent
    //  and issued up front, where the bit-mask loop waited out one LDS round trip per ancestor)
    int chn[7];
#pragma unroll
    for (int d = 0; d < 7; d++) chn[d] = m.mj_chain[bk][d];
#pragma unroll
    for (int d = 0; d < 7 + 6; d++) {
      const int i = d < 7 ? (chn[d < 7 ? d : 0] > 0 ? 5 + chn[d < 7 ? d : 0] : -1) : d - 7;
      if (i < 0 || i > k) continue;
      T val = 0;
#pragma unroll
      for (int e = 0; e < 6; e++) val += L.S[i][e] * Fk[e];
      L.M[i * LDM + k] = val;
      L.M[k * LDM + i] = val;
    }
    qfs = -hk;
  }
  wsync<NW>();
  if (lane < NV) L.M[lane * LDM + lane] += m.mj_armature[lane];
  if (lane < NA) {
    const int d = m.mj_act_dof[lane];
    // closed loop: TSID's joint torques as motor forces; otherwise the reference's position servos
    // position servo kp (ctrl - q) - kv qdot with the control clamped to its range and the force to its range
    // (robot/v0/robot.xml:5 ctrlrange / forcerange; the v1 robot has neither: +-1e300)
    const T cc = myctrl < m.act_range[lane][0] ? m.act_range[lane][0] : (myctrl > m.act_range[lane][1] ? m.act_range[lane][1] : myctrl);
    T servo = m.mj_act_kp[lane] * (cc - L.qpos[d + 1]) - m.mj_act_kv[lane] * L.qvel[d];
    servo = servo < m.act_range[lane][2] ? m.act_range[lane][2] : (servo > m.act_range[lane][3] ? m.act_range[lane][3] : servo);
    L.xv[d] = motor_tau ? motor_tau[m.mj_ctrl_qidx[lane] - 7] : servo;
  } else if (lane >= 32 && lane < 38) L.xv[lane - 32] = 0;
  wsync<NW>();
  if (lane < NV) qfs += L.xv[lane];
  if constexpr (EULERDAMP) { // passive joint damping (robot/v0/robot.xml:3)
    if (lane < NV) qfs -= m.mj_damping[lane] * L.qvel[lane];
  }
  } // (w_dyn)
  // ---- floor collision, first half: bounding-sphere pretest for all geoms at once (lane = geom, placed by its body).
  //      Done BEFORE the factorisation below (nothing of it stays live across it); only the geoms that can reach the
  //      floor enter the support search later, in geom order
  const T margin = m.contact[10], tie_tol = m.opt[6];
  const T Ow[3] = {L.qpos[0], L.qpos[1], L.qpos[2]};
  const T nO = dot3(fl.n, Ow) - fl.d; // signed distance of the base origin O to the floor plane
#ifdef TSIDB_NO_TERR
  const bool has_terr = false;
#else
  const bool has_terr = terr_g != nullptr;
#endif
  T hmax_all = 0;
  if (has_terr) {
#pragma unroll
    for (int i = 0; i < 16; i++) hmax_all = terr_g[4 + i] > hmax_all ? terr_g[4 + i] : hmax_all;
  }
  unsigned long long cand_geoms;
  {
    bool near = false;
    if (lane < NG) {
      const int gb = m.geom_body[lane];
      const T *Rg = L.R[gb];
      const T c6 = fl.n[0] * Rg[0] + fl.n[1] * Rg[3] + fl.n[2] * Rg[6];
      const T c7 = fl.n[0] * Rg[1] + fl.n[1] * Rg[4] + fl.n[2] * Rg[7];
      const T c8 = fl.n[0] * Rg[2] + fl.n[1] * Rg[5] + fl.n[2] * Rg[8];
      const T pzl = dot3(fl.n, L.p[gb]) + nO;
      const T zc = c6 * m.rbound[lane][0] + c7 * m.rbound[lane][1] + c8 * m.rbound[lane][2] + pzl;
      near = !(zc - m.rbound[lane][3] - hmax_all > margin);
      // second, tighter bound: the hull's body-frame box (centre, half extents) projected on the floor normal.  A swinging
      // foot a few centimetres up is inside its bounding sphere's reach of the floor but not its box's: no support search
      // for it (a whole pruned hull search, 5 % of a step, for most of every swing).  Exact: a lower bound of every vertex.
      const T *hb = m.hbox[lane];
      const T zb = c6 * hb[0] + c7 * hb[1] + c8 * hb[2] + pzl - (fabs(c6) * hb[3] + fabs(c7) * hb[4] + fabs(c8) * hb[5]);
      near = near && !(zb - fabs(zb) * T(8) * Eps<T>::v - hmax_all > margin);
    }
    cand_geoms = __ballot(near);
  }
  TSIDB_STAMP(17);
  // ---- qacc_smooth = M^-1 qfrc_smooth
  T arow[NV];
  T qas = 0;
  int fail = 0;
  // park the two per-lane values that live across the collision phase in LDS (the contact-force scratch is free until
  // the Newton loop): left in registers they are what the compiler spills to scratch around the narrow phase
  T *park = &L.cfv[0][0];
  if (w_dyn) {
#pragma unroll
    for (int j = 0; j < NV; j++) arow[j] = lane < NV ? L.M[lane * LDM + j] : T(0);
    bool spd;
    qas = chol26_solve<T, false>(arow, qfs, lane, spd);
    fail = spd ? 0 : 1;
    if (lane < NV) { park[lane] = qas; park[NV + lane] = qfs; }
  }

  TSIDB_STAMP(18);
  int ncon = 0, nfl = 0, cfail = 0; // contacts, floor contacts among them, flag bits of the collision phase
  bool hh_cross = false; // some robot<->robot contact couples two branches of the tree (dense Newton Hessian)
  if (w_col) {
  // ---- collision: floor (plane n.x = d, nominal z = 0, optionally with terrain steps) against each body's hull
  if (has_terr) { // stage this env's terrain table
    if (lane < 20) L.terr[lane] = terr_g[lane];
    wsync<NW>();
  }
  const bool pm_rule = m.params[P_PLANE_MESH] != 0; // upstream's plane <-> mesh rule instead of "every neighbour in the margin"
  for (unsigned long long bm = cand_geoms; bm; bm &= bm - 1) {
    const int g = __ffsll((long long)bm) - 1, b = m.geom_body[g];
    const T *Rb = L.R[b];
    // floor normal in the body frame; "z" below = signed distance to the floor
    const T r6 = fl.n[0] * Rb[0] + fl.n[1] * Rb[3] + fl.n[2] * Rb[6];
    const T r7 = fl.n[0] * Rb[1] + fl.n[1] * Rb[4] + fl.n[2] * Rb[7];
    const T r8 = fl.n[0] * Rb[2] + fl.n[1] * Rb[5] + fl.n[2] * Rb[8];
    const T pz = dot3(fl.n, L.p[b]) + nO;
    const int v0 = m.hull_adr[g];
    T zmin;
    const int best = has_terr ? hull_argmin<T, true>(m, lane, g, r6, r7, r8, pz, tie_tol, L.terr, Rb, L.p[b][0] + Ow[0],
                                                     L.p[b][1] + Ow[1], hmax_all, zmin)
                              : hull_argmin<T, false>(m, lane, g, r6, r7, r8, pz, tie_tol, nullptr, nullptr, T(0), T(0), T(0), zmin);
    if (!(zmin <= margin)) continue; // (also when zmin is NaN)
    // the support vertex, then its hull-graph neighbours within the margin
    const int e0 = m.hull_eadr[best];
    int nnb = m.hull_eadr[best + 1] - e0;
    if (nnb > WAVE - 1) { nnb = WAVE - 1; cfail |= 16; }
    bool keep = false;
    T w[3] = {0, 0, 0}, wd = 0;
    int vid = best;
    if (lane <= nnb) {
      if (lane > 0) vid = v0 + m.hull_edge[e0 + lane - 1];
      const T vv[3] = {m.hull_x[vid], m.hull_y[vid], m.hull_z[vid]};
      mat3vec(Rb, vv, w);
      w[0] += L.p[b][0]; w[1] += L.p[b][1]; w[2] += L.p[b][2]; // relative to O
      wd = dot3(fl.n, w) + nO;
      if (has_terr) wd -= terrain_h(L.terr, w[0] + Ow[0], w[1] + Ow[1]);
      keep = lane == 0 || wd <= margin;
    }
    T cp[3] = {w[0] - T(0.5) * wd * fl.n[0], w[1] - T(0.5) * wd * fl.n[1], w[2] - T(0.5) * wd * fl.n[2]}; // contact position
    if (pm_rule) {
      // in graph order (= lane order): at most 3 more contacts, each at least 0.3 rbound from the FIRST one (lane 0's)
      const T d3[3] = {cp[0] - rdlane(cp[0], 0), cp[1] - rdlane(cp[1], 0), cp[2] - rdlane(cp[2], 0)};
      const T thr = T(0.3) * m.rbound[g][3];
      const bool far = lane > 0 && keep && !(dot3(d3, d3) < thr * thr);
      const unsigned long long fm = __ballot(far);
      keep = lane == 0 || (far && __popcll(fm & ((1ull << lane) - 1ull)) < 3);
    }
    const unsigned long long mask = __ballot(keep);
    if (ncon + __popcll(mask) > MAXCON) cfail |= 8; // a contact is dropped at the cap
    const int slot = ncon + __popcll(mask & ((1ull << lane) - 1ull));
    if (keep && slot < MAXCON) {
      const T dist = wd;
      L.cbody[slot] = b;
      L.cgeom[slot] = g;
      L.cvert[slot] = vid - v0;
      L.cdist[slot] = dist;
#pragma unroll
      for (int i = 0; i < 3; i++) L.cr[slot][i] = cp[i];
#pragma unroll
      for (int d = 0; d < 7; d++) { const int a = m.mj_chain[b][d]; L.cchain[slot][d] = (unsigned char)(a > 0 ? a : 0); }
    }
    ncon += __popcll(mask);
    ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  nfl = ncon; // contacts [0, nfl) are floor contacts (shared frame), [nfl, ncon) robot<->robot ones
  // ---- collision: robot<->robot convex-hull pairs (robot.xml:13-15 after the excludes of :18-52 and the
  //      parent-child filter): mid phase one lane per pair, narrow phase (MPR) one pair at a time on the wave
#ifndef TSIDB_NO_HH
  if (m.params[P_SELF_COLLISION] != 0) {
    // bounding spheres of all bodies in the world (lane = body)
    if (lane < NG) {
      const int gb = m.geom_body[lane];
      T c[3];
      mat3vec(L.R[gb], m.rbound[lane], c);
      L.scen[lane][0] = c[0] + L.p[gb][0]; L.scen[lane][1] = c[1] + L.p[gb][1]; L.scen[lane][2] = c[2] + L.p[gb][2];
      L.scen[lane][3] = m.rbound[lane][3];
    }
    wsync<NW>();
    // broad phase, one lane per pair: sphere test, survivors compacted into a list; the box test then runs on the list
    // 64 entries at a time (once, at the end, for the v1 robot's ~30 survivors; the v0 robot has 1044 pairs)
    int nsph = 0, ncand = 0;
    bool over = false, over64 = false;
    auto box_pass = [&](int cnt) { // pcand[0, cnt) -> survivors appended to fcand, in pair order
      const int k = lane < cnt ? L.pcand[lane] : 0;
      const bool may = lane < cnt && boxes_may_touch(m, L, m.pair_a[k], m.pair_b[k], margin);
      const unsigned long long mk = __ballot(may);
      const int pos = ncand + __popcll(mk & ((1ull << lane) - 1ull));
      if (may && pos < WAVE) L.fcand[pos] = k;
      ncand += __popcll(mk);
    };
    for (int k0 = 0; k0 < m.npair; k0 += WAVE) {
      const int k = k0 + lane;
      const bool may = k < m.npair && spheres_overlap(L, m.pair_a[k], m.pair_b[k], margin);
      const unsigned long long mk = __ballot(may);
      if (may) L.pcand[nsph + __popcll(mk & ((1ull << lane) - 1ull))] = k; // nsph <= 64 here: the list holds 128
      nsph += __popcll(mk);
      if (nsph > WAVE) {
        wsync<NW>();
        box_pass(WAVE);
        const int rest = lane < nsph - WAVE ? L.pcand[WAVE + lane] : 0;
        wsync<NW>();
        if (lane < nsph - WAVE) L.pcand[lane] = rest;
        nsph -= WAVE;
        wsync<NW>();
      }
    }
    wsync<NW>();
    box_pass(nsph);
    if (ncand > WAVE) { ncand = WAVE; over64 = true; }
    wsync<NW>();
    for (int ci = 0; ci < ncand; ci++) {
      const int k = L.fcand[ci];
      const int ga = m.pair_a[k], gb = m.pair_b[k], a = m.geom_body[ga], b = m.geom_body[gb]; // geoms, their bodies
      T depth, dir[3], pos[3];
      if (!mpr_penetration(m, L, lane, ga, gb, margin, depth, dir, pos)) continue;
      if (ncon - nfl >= MAXHH || ncon >= MAXCON) { over = true; continue; }
      if (lane == 0) {
        L.cbody[ncon] = b;
        L.cgeom[ncon] = gb;
        L.cvert[ncon] = 0x8000 | ga;
        L.cdist[ncon] = margin - depth;
        L.hb1_s[ncon - nfl] = a;
#pragma unroll
        for (int i = 0; i < 3; i++) { L.cr[ncon][i] = pos[i]; L.hn_s[ncon - nfl][i] = dir[i]; }
      }
      if (!((bodyanc_of(m, b) >> a) & 1u) && !((bodyanc_of(m, a) >> b) & 1u)) hh_cross = true;
      ncon++;
    }
    if (over) cfail |= 8;
    if (over64) cfail |= 32;
  }
#endif
  } // (w_col)
  if constexpr (NW > 1) { // join: the collision wavefront hands over its counts and is done
    if (wv == NW - 1 && lane == 0) { L.xch[0] = ncon; L.xch[1] = nfl; L.xch[2] = hh_cross ? 1 : 0; L.xch[3] = cfail; }
    __syncthreads();
    if (wv != 0) return; // (from here on only wavefront 0 works on this env's step: its LDS hand-overs are wavefront-local)
    ncon = L.xch[0]; nfl = L.xch[1]; hh_cross = L.xch[2] != 0; cfail = L.xch[3];
  }
  fail |= cfail;
  { // the robot<->robot contacts' normals and geom1 bodies move from the collision scratch to where the Newton loop keeps them
    T hv = 0;
    int hb = 0;
    if (lane < 3 * MAXHH) hv = (&L.hn_s[0][0])[lane];
    if (lane < MAXHH) hb = L.hb1_s[lane];
    wsync<NW>();
    if (lane < 3 * MAXHH) (&L.hn[0][0])[lane] = hv;
    if (lane < MAXHH) L.hb1[lane] = hb;
  }
  wsync<NW>();
  qas = lane < NV ? park[lane] : T(0);
  qfs = lane < NV ? park[NV + lane] : T(0);
  asm volatile("" ::"v"(qas), "v"(qfs)); // (the reload is the definition the rest of the kernel uses)
  if (lane == 0 && ncon_out) ncon_out[0] = ncon;
  if (con_out && lane < MAXCON) con_out[lane] = lane < ncon ? ((L.cgeom[lane] << 16) | L.cvert[lane]) : -1;

  TSIDB_STAMP(19);
  // ---- constraint rows: frictionloss (lane = dof), pyramidal contact rows (lane = contact)
  // friction: floor contacts take the per-env value (config 5), robot<->robot contacts the model's
  const T mu = (lane >= nfl || !envp) ? m.contact[0] : envp[1];
  // (solver constants are read where they are used: wave-uniform values held from the top of the kernel cost SGPRs,
  //  and SGPRs spilled to VGPR lanes are what pushes this kernel over its VGPR budget)
  asm volatile("" ::: "memory");
  const T dt = m.opt[0];
  const T timeconst = m.contact[1] > 2 * dt ? m.contact[1] : 2 * dt, dampratio = m.contact[2];
  const T dmin = m.contact[3], dmax = m.contact[4], width = m.contact[5], mid = m.contact[6], power = m.contact[7];
  const T kk = T(1) / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), bb = T(2) / (dmax * timeconst);
  // frictionloss rows take the joint's solreffriction, which neither MJCF sets: MuJoCo's default (0.02, 1)
  const T tcf = T(0.02) > 2 * dt ? T(0.02) : 2 * dt, bbf = T(2) / (dmax * tcf);
  RowState<T> rs;
  rs.has_f = lane < NV && m.mj_frictionloss[lane < NV ? lane : 0] > 0;
  rs.fD = rs.fR = rs.floss = rs.faref = rs.fjar = rs.fJv = 0;
  if (rs.has_f) {
    const T r = (1 - dmin) / dmin * m.mj_dof_invw0[lane];
    rs.fR = r > MINVAL ? r : MINVAL;
    rs.fD = T(1) / rs.fR;
    rs.floss = m.mj_frictionloss[lane];
    rs.faref = -bbf * L.qvel[lane];
  }
  rs.has_c = lane < ncon;
  rs.cD = 0;
#pragma unroll
  for (int i = 0; i < NROWC; i++) rs.caref[i] = rs.cjar[i] = rs.cJv[i] = 0;
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
    T tran = m.mj_body_invw0[L.cbody[c]][0];
    if (c >= nfl) tran += m.mj_body_invw0[L.hb1[c - nfl]][0]; // geom1's body (the floor contributes 0)
    const T diagA = tran + mu * mu * tran;
    T R0 = (1 - imp) / imp * diagA;
    R0 = R0 > MINVAL ? R0 : MINVAL;
    rs.cD = T(1) / (2 * mu * mu * R0);
    // (reference acceleration of the rows: - bb J v is added below, in the walk that also evaluates the warm start)
#pragma unroll
    for (int i = 0; i < NROWC; i++) rs.caref[i] = -kk * imp * (dist - margin);
  }

  int solver_iter = 0;
  T qacc = qas;
  const int nefc = NA + NROWC * ncon; // frictionloss rows exist on all hinges
  T Ma = 0; // M qacc at the current point
  if (nefc > 0) {
    // helpers over a candidate qacc held per lane (value `xa`, also staged in L.xv)
    auto stage = [&](T xa) { wsync<NW>(); if (lane < NV) L.xv[lane] = xa; wsync<NW>(); };
    // warm start: keep qacc_warmstart only if it is cheaper than qacc_smooth.  The contact rows of the joint velocities
    // (reference acceleration), of the warm start and of qacc_smooth come out of ONE walk over each contact's body chain
    // (qacc_smooth is still in LDS where it was parked for the collision phase)
    T xw = lane < NV ? qacc_ws_g[lane] : T(0);
    stage(xw);
    Ma = mulM(L, L.xv, lane);
    T cjar_w[NROWC], cjar_s[NROWC];
#pragma unroll
    for (int i = 0; i < NROWC; i++) cjar_w[i] = cjar_s[i] = 0;
    if (rs.has_c) {
      const T *const xs[3] = {L.qvel, L.xv, park};
      T o[3][NROWC];
      contact_rows_multi<T, 3>(m, L, nfl, lane, xs, mu, o);
#pragma unroll
      for (int i = 0; i < NROWC; i++) {
        rs.caref[i] -= bb * o[0][i];
        cjar_w[i] = o[1][i] - rs.caref[i];
        cjar_s[i] = o[2][i] - rs.caref[i];
      }
    }
    const T fjar_w = rs.has_f ? xw - rs.faref : T(0), fjar_s = rs.has_f ? qas - rs.faref : T(0);
    T cc, gg, hh;
    rs.fjar = fjar_w;
#pragma unroll
    for (int i = 0; i < NROWC; i++) rs.cjar[i] = cjar_w[i];
    rows_eval(rs, T(0), cc, gg, hh);
    T cost_w = wave_sum(cc + (lane < NV ? T(0.5) * (Ma - qfs) * (xw - qas) : T(0)));
    rs.fjar = fjar_s;
#pragma unroll
    for (int i = 0; i < NROWC; i++) rs.cjar[i] = cjar_s[i];
    rows_eval(rs, T(0), cc, gg, hh);
    T cost_s = wave_sum(cc);
    if (cost_w > cost_s) {
      qacc = qas;
      Ma = mulM(L, park, lane); // M qacc_smooth
    } else {
      qacc = xw;
      rs.fjar = fjar_w;
#pragma unroll
      for (int i = 0; i < NROWC; i++) rs.cjar[i] = cjar_w[i];
    }

    TSIDB_STAMP(20);
    asm volatile("" ::: "memory");
    // (float32: the solver tolerance 1e-8 is below the rounding noise of the cost, so a converged env used to pass the
    //  improvement test by noise, build and factor a second Hessian and then find no step: the tolerance is floored at
    //  64 ulp there - no effect in float64)
    // (read where they are used, every iteration: kept from here they are loop-invariant values in registers the loop
    //  does not have - the compiler parked them in scratch memory and reloaded them inside the loop)
    auto solver_tol = [&]() { asm volatile("" ::: "memory"); const T t0 = m.opt[2]; return t0 > 64 * Eps<T>::v ? t0 : 64 * Eps<T>::v; };
    T cost = 0;
    int iter = 0;
    // Newton Hessian H = M + J^T D J over the rows in their quadratic zone: built and factored in full on the first
    // iteration; afterwards only the rows whose state changed are applied to the FACTOR, as rank-1 updates / downdates
    // with sqrt(D_r) J_r (MuJoCo's Newton solver does the same for pyramidal cones: HessianIncremental, mju_cholUpdate) -
    // unless more than NEWTON_INCR_MAX rows changed, a downdate loses definiteness, or a robot<->robot contact couples
    // two branches of the tree (dense factor).  In a touch-down window 80 % of the later iterations change <= 2 rows
    // (tools/newton_stats.py).  Between iterations the factor is parked in the LDS copy of H (column 26: 1 / diagonal).
    bool have_fac = false;
    // the floor contacts arrive grouped by body: bit c = contact c is the first of its body's group (fixed for the step)
    const unsigned long long gfirst = __ballot(lane < nfl && (lane == 0 || L.cbody[lane] != L.cbody[lane > 0 ? lane - 1 : 0]));
    const bool grouped = nfl > 2;
    unsigned prevbits = 0; // bit 0: this lane's friction row was in its quadratic zone, bit 1 + i: contact row i was active
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
      T tors_e = 0, tors_f = 0;      // CONDIM 4: mu_t (D4 - D5), mu_t^2 (D4 + D5) over the active torsional rows
      if (rs.has_c) {
        T fr[NROWC];
        T fv[3] = {0, 0, 0};
        T cn[3], ct1[3], ct2[3];
        contact_frame(L, lane, nfl, cn, ct1, ct2);
        if constexpr (CONDIM > 3) { // torsional rows: force along the normal, torque +- mu_t * force about it
          const T mut = m.contact[9];
          const bool a4 = rs.cjar[4] < 0, a5 = rs.cjar[5] < 0;
          const T f4 = a4 ? -rs.cD * rs.cjar[4] : T(0), f5 = a5 ? -rs.cD * rs.cjar[5] : T(0);
          const T tq = mut * (f4 - f5);
          L.ctq[lane][0] = tq * cn[0]; L.ctq[lane][1] = tq * cn[1]; L.ctq[lane][2] = tq * cn[2];
          tors_e = mut * ((a4 ? rs.cD : T(0)) - (a5 ? rs.cD : T(0)));
          tors_f = mut * mut * ((a4 ? rs.cD : T(0)) + (a5 ? rs.cD : T(0)));
        }
#pragma unroll
        for (int i = 0; i < NROWC; i++) {
          const bool act = rs.cjar[i] < 0;
          fr[i] = act ? -rs.cD * rs.cjar[i] : T(0);
          // direction of row i in world axes: n + s*mu*t_k (sliding rows), n (torsional rows)
          const T sg = i >= 4 ? T(0) : ((i & 1) ? -mu : mu);
          const T *tk = i < 2 ? ct1 : ct2;
          const T dir[3] = {cn[0] + sg * tk[0], cn[1] + sg * tk[1], cn[2] + sg * tk[2]};
#pragma unroll
          for (int e = 0; e < 3; e++) fv[e] += fr[i] * dir[e];
          if (act) {
            Arow[0] += rs.cD * dir[0] * dir[0]; Arow[1] += rs.cD * dir[0] * dir[1]; Arow[2] += rs.cD * dir[0] * dir[2];
            Arow[3] += rs.cD * dir[1] * dir[1]; Arow[4] += rs.cD * dir[1] * dir[2]; Arow[5] += rs.cD * dir[2] * dir[2];
          }
        }
        L.cfv[lane][0] = fv[0]; L.cfv[lane][1] = fv[1]; L.cfv[lane][2] = fv[2];
        if (grouped && lane < nfl) { // floor contact: its wrench about O, for the per-body sums below (the composite-inertia scratch is free)
          T rxf[3];
          cross3(L.cr[lane], fv, rxf);
          if constexpr (CONDIM > 3) { rxf[0] += L.ctq[lane][0]; rxf[1] += L.ctq[lane][1]; rxf[2] += L.ctq[lane][2]; }
          T *gw = &L.K[0][0] + 6 * lane;
          gw[0] = fv[0]; gw[1] = fv[1]; gw[2] = fv[2]; gw[3] = rxf[0]; gw[4] = rxf[1]; gw[5] = rxf[2];
        }
      }
      wsync<NW>();
      // ---- gradient: Ma - qfrc_smooth - J^T force.  With more than two floor contacts their wrenches are summed per contact
      //      BODY first (lane e sums component e over each group, in place), then dof k takes S_k . (sum of the groups below
      //      it) - a handful of groups instead of a loop over every contact per dof; robot<->robot contacts, and one or two
      //      floor contacts, keep the per-contact form (the extra exchange costs more than it saves there)
      if (grouped) {
        for (unsigned long long gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffsll((long long)gm) - 1;
          const unsigned long long rest = gm & (gm - 1);
          const int ce = rest ? __ffsll((long long)rest) - 1 : nfl;
          if (lane < 6) {
            T acc = 0;
            for (int c = cf; c < ce; c++) acc += (&L.K[0][0])[6 * c + lane];
            (&L.K[0][0])[6 * cf + lane] = acc;
          }
        }
        wsync<NW>();
      }
      T grad = 0;
      if (lane < NV) {
        const int k = lane, bk = k < 6 ? 0 : k - 5;
        T s = 0;
        if (grouped)
          for (unsigned long long gm = gfirst; gm; gm &= gm - 1) {
            const int cf = __ffsll((long long)gm) - 1;
            if ((L.anc[L.cbody[cf]] >> bk) & 1u) {
              const T *gw = &L.K[0][0] + 6 * cf;
              s += L.S[k][0] * gw[0] + L.S[k][1] * gw[1] + L.S[k][2] * gw[2] + L.S[k][3] * gw[3] + L.S[k][4] * gw[4] + L.S[k][5] * gw[5];
            }
          }
        for (int c = grouped ? nfl : 0; c < ncon; c++) {
          // +1 on geom2's chain, -1 on geom1's, 0 above their common ancestor
          int sgn = (int)((L.anc[L.cbody[c]] >> bk) & 1u);
          if (c >= nfl) sgn -= (int)((L.anc[L.hb1[c - nfl]] >> bk) & 1u);
          if (sgn == 0) continue;
          T rxf[3];
          cross3(L.cr[c], L.cfv[c], rxf);
          if constexpr (CONDIM > 3) { rxf[0] += L.ctq[c][0]; rxf[1] += L.ctq[c][1]; rxf[2] += L.ctq[c][2]; }
          const T js = L.S[k][0] * L.cfv[c][0] + L.S[k][1] * L.cfv[c][1] + L.S[k][2] * L.cfv[c][2] +
                       L.S[k][3] * rxf[0] + L.S[k][4] * rxf[1] + L.S[k][5] * rxf[2];
          s += sgn > 0 ? js : -js;
        }
        grad = Ma - qfs - s - ff;
      }
      if (iter > 0) {
        const T gn = wave_sum(grad * grad);
        const T tol = solver_tol(), scale = T(1) / (m.meaninertia * NV);
        const T improvement = scale * (cost - newcost), gradient = scale * sqrt(gn);
        cost = newcost;
        if (improvement < tol || gradient < tol) break;
      }
      cost = newcost;
      if (iter >= (int)m.opt[3]) break;
      TSIDB_LAP(24);
      unsigned actbits = fact ? 1u : 0u;
      if (rs.has_c) {
#pragma unroll
        for (int i = 0; i < NROWC; i++) actbits |= rs.cjar[i] < 0 ? (2u << i) : 0u;
      }
      bool full = NEWTON_INCR_MAX < 0 || !have_fac || hh_cross;
      bool ok = true;
      T search = 0;
      if (!full) {
        unsigned chg = actbits ^ prevbits;
        int nchange = __popc(chg);
        nchange = wave_sum_int(nchange);
        if (nchange > NEWTON_INCR_MAX) full = true;
        else {
          const NewtonDir<T> nd = newton_direction<T>(m, L, nfl, 1, chg, actbits, mu, rs.cD, rs.fD, grad);
          search = -nd.search;
          if (nd.status != 0) full = true; // a downdate lost definiteness: rebuild
        }
      }
      prevbits = actbits;
      if (full) {
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
        if constexpr (CONDIM > 3) {
          // torsional rows w = [n; r x n +- mu_t n]: on top of the above, e n n^T in the lin-ang block and
          // e ((r x n) n^T + n (r x n)^T) + f n n^T in the ang-ang block
          T cn[3], ct1[3], ct2[3], xn[3];
          contact_frame(L, lane, nfl, cn, ct1, ct2);
          cross3(r, cn, xn);
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              L.Wc[lane][sym_idx(i, 3 + j)] += tors_e * cn[i] * cn[j];
              if (j >= i) L.Wc[lane][sym_idx(3 + i, 3 + j)] += tors_e * (xn[i] * cn[j] + cn[i] * xn[j]) + tors_f * cn[i] * cn[j];
            }
        }
      }
      wsync<NW>();
      // composite contact inertia K[a] = sum of the contact inertias of every body in a's subtree.  Contacts arrive grouped
      // by body.  Stage 1: lane e sums entry e over each group, in place (the group's first contact keeps the sum).  Stage 2:
      // the 21 NB entries of K are dealt to the lanes, each adds up the groups whose body has `a` among its ancestors -
      // plain loads and one store per entry (the first version pushed every group sum to every ancestor with dependent
      // LDS read-modify-writes, after zeroing K: 3-4 k cycles of latency per build; same sums in the same order)
      unsigned touched = 0;
      {
        for (unsigned long long gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffsll((long long)gm) - 1;
          const unsigned long long rest = gm & (gm - 1);
          const int ce = rest ? __ffsll((long long)rest) - 1 : nfl;
          touched |= L.anc[L.cbody[cf]];
          if (lane < 21) {
            T acc = 0;
            for (int c = cf; c < ce; c++) acc += L.Wc[c][lane];
            L.Wc[cf][lane] = acc;
          }
        }
      }
      wsync<NW>();
      for (int id = lane; id < NB * 21; id += WAVE) {
        const int a = id / 21, e = id - 21 * a;
        T acc = 0;
        for (unsigned long long gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffsll((long long)gm) - 1;
          if ((L.anc[L.cbody[cf]] >> a) & 1u) acc += L.Wc[cf][e];
        }
        L.K[a][e] = acc;
      }
      // (K sits in the tree-pass scratch, Wc in the H region: H is written only after the barrier below)
      wsync<NW>();
      T Gk[6] = {0, 0, 0, 0, 0, 0};
      const bool mine = lane < NV && ((touched >> (lane < 6 ? 0 : lane - 5)) & 1u);
      if (mine) { // G = K[body of dof k] S_k, before Wc's space becomes H
        const int k = lane, bk = k < 6 ? 0 : k - 5;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          T sacc = 0;
#pragma unroll
          for (int j = 0; j < 6; j++) sacc += L.K[bk][sym_idx(i, j)] * L.S[k][j];
          Gk[i] = sacc;
        }
      }
      wsync<NW>();
      for (int i = lane; i < NV * LDM; i += WAVE) L.H[i] = L.M[i];
      wsync<NW>();
      if (mine) { // H[i][k] = H[k][i] = M[i][k] + S_i . G_k for the dofs i <= k on k's root path: written once, by lane k alone
        const int k = lane, bk = k < 6 ? 0 : k - 5;
        int chn[7];
#pragma unroll
        for (int d = 0; d < 7; d++) chn[d] = m.mj_chain[bk][d];
#pragma unroll
        for (int d = 0; d < 7 + 6; d++) { // (static walk over the chain table, as for M)
          const int i = d < 7 ? (chn[d < 7 ? d : 0] > 0 ? 5 + chn[d < 7 ? d : 0] : -1) : d - 7;
          if (i < 0 || i > k) continue;
          T val = 0;
#pragma unroll
          for (int e = 0; e < 6; e++) val += L.S[i][e] * Gk[e];
          const T hv = L.M[i * LDM + k] + val;
          L.H[i * LDM + k] = hv;
          L.H[k * LDM + i] = hv;
        }
      }
      wsync<NW>();
      if (fact) L.H[lane * LDM + lane] += rs.fD;
      wsync<NW>();
      // robot<->robot contacts (rare): H += J^T A J with J = the contact point's relative velocity per unit dof
      // rate (3 x 26, column k on lane k) and A = sum over the active rows of D dir dir^T (3 x 3, held by the
      // contact's lane).  Done on the LDS copy of H, before its rows go to registers: g_k = A j_k is staged in the
      // (now dead) composite-inertia scratch, then lane i adds j_i . g_k to its row for every k.
      for (int c = nfl; c < ncon; c++) {
        T A6[6];
#pragma unroll
        for (int e = 0; e < 6; e++) A6[e] = rdlane_dyn(Arow[e], c);
        T jk[3] = {0, 0, 0};
        if (lane < NV) {
          const int bk = lane < 6 ? 0 : lane - 5;
          const int sgn = (int)((L.anc[L.cbody[c]] >> bk) & 1u) - (int)((L.anc[L.hb1[c - nfl]] >> bk) & 1u);
          if (sgn != 0) {
            T wxr[3];
            cross3(&L.S[lane][3], L.cr[c], wxr);
#pragma unroll
            for (int i = 0; i < 3; i++) jk[i] = sgn > 0 ? L.S[lane][i] + wxr[i] : -(L.S[lane][i] + wxr[i]);
          }
          T *g = &L.K[0][0] + 5 * lane;
          g[0] = A6[0] * jk[0] + A6[1] * jk[1] + A6[2] * jk[2];
          g[1] = A6[1] * jk[0] + A6[3] * jk[1] + A6[4] * jk[2];
          g[2] = A6[2] * jk[0] + A6[4] * jk[1] + A6[5] * jk[2];
          if constexpr (CONDIM > 3) { // a_k = n . j_k (relative linear), b_k = n . (relative angular) for the torsional rows
            const T *hn = L.hn[c - nfl];
            g[3] = hn[0] * jk[0] + hn[1] * jk[1] + hn[2] * jk[2];
            g[4] = sgn == 0 ? T(0) : (sgn > 0 ? T(1) : T(-1)) * (hn[0] * L.S[lane][3] + hn[1] * L.S[lane][4] + hn[2] * L.S[lane][5]);
          }
        }
        wsync<NW>();
        const T te = rdlane_dyn(tors_e, c), tf = rdlane_dyn(tors_f, c);
        if (lane < NV) {
          const T *g = &L.K[0][0];
          for (int k = 0; k < NV; k++) L.H[lane * LDM + k] += jk[0] * g[5 * k] + jk[1] * g[5 * k + 1] + jk[2] * g[5 * k + 2];
          if constexpr (CONDIM > 3) {
            const T ai = g[5 * lane + 3], bi = g[5 * lane + 4];
            for (int k = 0; k < NV; k++) L.H[lane * LDM + k] += te * (ai * g[5 * k + 4] + bi * g[5 * k + 3]) + tf * bi * g[5 * k + 4];
          }
        }
        wsync<NW>();
      }
      TSIDB_LAP(25);
      if (hh_cross) { // rare: the dense variant (out of line as well)
        T dense_rows[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) dense_rows[j] = lane < NV ? L.H[lane * LDM + j] : T(0);
        search = -chol26_dense(dense_rows, grad, lane, ok);
      } else {
        const NewtonDir<T> nd = newton_direction<T>(m, L, nfl, 0, 0u, actbits, mu, rs.cD, rs.fD, grad);
        search = -nd.search;
        ok = nd.status == 0;
        have_fac = true;
      }
      } // (full)
      if (!ok) { fail |= 2; break; }
      TSIDB_LAP(26);
      // ---- exact line search along `search`
      stage(search);
      const T Mv = mulM(L, L.xv, lane);
      if (rs.has_f) rs.fJv = search;
      if (rs.has_c) contact_rows(m, L, nfl, lane, L.xv, mu, rs.cJv);
      T qg1 = lane < NV ? search * (Ma - qfs) : T(0), qg2 = lane < NV ? T(0.5) * search * Mv : T(0),
        snorm = lane < NV ? search * search : T(0);
      wave_sum3(qg1, qg2, snorm);
      snorm = sqrt(snorm);
      if (snorm < MINVAL) break;
      const T gtol = solver_tol() * m.opt[5] * snorm * m.meaninertia * NV;
      const int ls_iter = (int)m.opt[4];
      auto ls_eval = [&](T alpha, T &c, T &d1, T &d2) {
        T lc, lg, lh;
        rows_eval(rs, alpha, lc, lg, lh);
        wave_sum3(lc, lg, lh);
        c = alpha * alpha * qg2 + alpha * qg1 + gauss + lc;
        d1 = 2 * alpha * qg2 + qg1 + lg;
        d2 = 2 * qg2 + lh;
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
        for (int i = 0; i < NROWC; i++) rs.cjar[i] += alpha * rs.cJv[i];
      }
      iter++;
      TSIDB_LAP(27);
    }
    solver_iter = iter;
  }
  wsync<NW>();
  TSIDB_STAMP(21);
  // ---- semi-implicit Euler, write back
  asm volatile("" ::: "memory");
  const T dte = m.opt[0];
  T qacc_int = qacc; // the acceleration the velocity is advanced with
  if constexpr (EULERDAMP) {
    // MuJoCo's Euler integrates joint damping implicitly: v+ = v + h (M + h B)^-1 (M qacc), with M qacc = the total
    // force the solver ended on (tracked in Ma); qacc itself (reported, warm start) stays the solver's
    T Mq = Ma;
    if (nefc == 0) { wsync<NW>(); if (lane < NV) L.xv[lane] = qacc; wsync<NW>(); Mq = mulM(L, L.xv, lane); }
#pragma unroll
    for (int j = 0; j < NV; j++) arow[j] = lane < NV ? L.M[lane * LDM + j] + (j == lane ? dte * m.mj_damping[j] : T(0)) : T(0);
    bool okd;
    qacc_int = chol26_solve<T, false>(arow, Mq, lane, okd);
    if (!okd) fail |= 1;
  }
  if (lane < NV) {
    const T vn = L.qvel[lane] + dte * qacc_int;
    L.qvel[lane] = vn;
    qvel_g[lane] = vn;
    qacc_ws_g[lane] = qacc;
    if (qacc_out) qacc_out[lane] = qacc;
  }
  wsync<NW>();
  if (lane < 3) L.qpos[lane] += dte * L.qvel[lane];
  if (lane >= 6 && lane < NV) L.qpos[lane + 1] += dte * L.qvel[lane];
  if (lane == 3) {
    const T *w = &L.qvel[3];
    const T th = sqrt(dot3(w, w)) * dte;
    T dq[4] = {1, 0, 0, 0};
    if (th > 0) {
      T sh, ch;
      sincos_t(T(0.5) * th, sh, ch);
      const T s = sh * dte / th;
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
  wsync<NW>();
  if (lane < NQ) qpos_g[lane] = L.qpos[lane];
  if (lane == 0 && info) { info[2] = solver_iter; info[3] = fail; }
  TSIDB_STAMP(22);
}

} // namespace tsidb
