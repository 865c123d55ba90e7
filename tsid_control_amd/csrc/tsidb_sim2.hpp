// tsidb_sim2.hpp - the sim step (tsidb_sim.hpp: main.py:192-195, mujoco.mj_step) in the PACKED layout: two envs per
// wavefront, env h on lanes [32 h, 32 h + 32), each with its own SimLds.
//
// Same operations on the same data in the same order as sim_step_env<T, 1> - per env the results are bit-identical to the
// one-env-per-wavefront kernel (tested) - but every vector instruction serves two envs (the one-env kernel has 26 dofs /
// 21 bodies / <= 32 contacts on 64 lanes) and the float64 broadcasts of the register factorisations are DPP moves instead
// of v_readlane pairs through the scalar unit (tsidb_pack.hpp).  What is wave-uniform there (contact counts, Newton
// iteration state, candidate lists) is uniform per env here: plain divergent SIMT code, the two envs of a wavefront take
// their own branches and loop counts.  Limits: NB, NV, NG, MAXCON <= 32 (the v1 robot; the v0 robot's 52 geoms do not fit).
#pragma once
#include <type_traits>
#include "tsidb_sim.hpp"
#include "tsidb_pack.hpp"

namespace tsidb {

template <int K0, int K1, typename F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (K0 < K1) {
    f(std::integral_constant<int, K0>{});
    static_for<K0 + 1, K1>(f);
  }
}

constexpr bool SIM_PACKABLE = NB <= pk::LPE && NV <= pk::LPE && NG <= pk::LPE && MAXCON <= pk::LPE && NA + 6 <= pk::LPE && !EULERDAMP && CONDIM == 3;

// chol26_solve (tsidb_sim.hpp) on 32 lanes per env: lane hl = row hl, broadcasts by DPP
template <typename T, bool DENSE>
__device__ __forceinline__ T chol26_solve_p(T (&a)[NV], T rhs, int hl, bool &spd) {
  int notspd = 0;
  T rd[NV]; // 1 / U[k][k], uniform per env
  static_for<0, NV>([&](auto tt) {
    constexpr int k = NV - 1 - decltype(tt)::value;
    const T akk = pk::bc1<k>(a[k]);
    notspd = akk > 0 ? notspd : 1;
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    rd[k] = rk;
    const T uik = hl < k ? a[k] * rk : (hl == k ? akk * rk : T(0));
    a[k] = uik;
    if constexpr (k > 0 && (DENSE || MJ_DOFANC[k] != 0u)) {
      const pk::Dup<T> du = pk::dup(uik);
      static_for<0, k>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if constexpr (DENSE || ((MJ_DOFANC[k] >> j) & 1u)) a[j] -= uik * pk::bc<j>(du);
      });
    }
  });
  spd = notspd == 0;
  // U y = rhs (descendants first; lane k contributes y_k), then U^T x = y (ancestors first, uniform per env)
  T acc = rhs, y[NV];
  static_for<0, NV>([&](auto tt) {
    constexpr int k = NV - 1 - decltype(tt)::value;
    y[k] = pk::bc1<k>(acc) * rd[k];
    acc -= a[k] * y[k];
  });
  T xs[NV], x = 0;
  static_for<0, NV>([&](auto kk) {
    constexpr int k = decltype(kk)::value;
    T s0 = y[k], s1 = 0; // (the two chains of the one-env routine: positions 0, 2 / 1, 3 of each group of four)
    if constexpr (k > 0 && (DENSE || MJ_DOFANC[k] != 0u)) {
      const pk::Dup<T> dk = pk::dup(a[k]);
      static_for<0, k>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        if constexpr (DENSE || ((MJ_DOFANC[k] >> i) & 1u)) {
          if constexpr ((i & 1) == 0) s0 -= pk::bc<i>(dk) * xs[i];
          else s1 -= pk::bc<i>(dk) * xs[i];
        }
      });
    }
    xs[k] = (s0 + s1) * rd[k];
    if (hl == k) x = xs[k];
  });
  return x;
}
template <typename T>
__device__ __noinline__ T chol26_dense_p(T (&a)[NV], T rhs, int hl, bool &spd) { return chol26_solve_p<T, true>(a, rhs, hl, spd); }

// chol26_factor / chol26_subst / chol26_rank1 (tsidb_sim.hpp) on 32 lanes per env
template <typename T>
__device__ __forceinline__ void chol26_factor_p(T (&a)[NV], T &rdv, int hl, bool &spd) {
  int notspd = 0;
  rdv = 0;
  static_for<0, NV>([&](auto tt) {
    constexpr int k = NV - 1 - decltype(tt)::value;
    const T akk = pk::bc1<k>(a[k]);
    notspd = akk > 0 ? notspd : 1;
    const T rk = rsqrt_t(akk > 0 ? akk : T(1));
    rdv = hl == k ? rk : rdv;
    const T uik = hl < k ? a[k] * rk : (hl == k ? akk * rk : T(0));
    a[k] = uik;
    if constexpr (MJ_DOFANC[k] != 0u) {
      const pk::Dup<T> du = pk::dup(uik);
      static_for<0, k>([&](auto jj) {
        constexpr int j = decltype(jj)::value;
        if constexpr ((MJ_DOFANC[k] >> j) & 1u) a[j] -= uik * pk::bc<j>(du);
      });
    }
  });
  spd = notspd == 0;
}
template <typename T>
__device__ __forceinline__ T chol26_subst_p(const T (&a)[NV], const T *Up, T rdv, T rhs, int hl) {
  T b[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) b[i] = hl < NV ? Up[i * LDM + hl] : T(0);
  T acc = rhs, yv = 0;
  static_for<0, NV>([&](auto tt) {
    constexpr int k = NV - 1 - decltype(tt)::value;
    const T yl = acc * rdv;
    const T yk = pk::bc1<k>(yl);
    yv = hl == k ? yl : yv;
    acc -= a[k] * yk;
  });
  acc = yv;
  T x = 0;
  static_for<0, NV>([&](auto ii) {
    constexpr int i = decltype(ii)::value;
    const T xl = acc * rdv;
    const T xi = pk::bc1<i>(xl);
    x = hl == i ? xl : x;
    acc -= b[i] * xi;
  });
  return x;
}
template <typename T>
__device__ __forceinline__ bool chol26_rank1_p(T (&a)[NV], T &rdv, T xv, T sigma, unsigned chain, int hl) {
  bool ok = true;
  static_for<0, NV>([&](auto tt) {
    constexpr int k = NV - 1 - decltype(tt)::value;
    if (ok && ((chain >> k) & 1u)) { // (uniform per env)
      const T s = pk::bc1<k>(xv * rdv);
      const T q = T(1) + sigma * s * s;
      if (!(q > (sizeof(T) == 8 ? T(1e-10) : T(1e-5)))) ok = false;
      else {
        const T ic = rsqrt_t(q), c = q * ic;
        const T un = (a[k] + sigma * s * xv) * ic;
        xv = hl == k ? T(0) : c * xv - s * un;
        a[k] = un;
        rdv = hl == k ? rdv * ic : rdv;
      }
    }
  });
  return ok;
}

// hull_argmin (tsidb_sim.hpp) with 32 lanes per env: the chunk bounds take two rounds of lanes (a foot has 45 chunks), a
// 64-vertex chunk is scanned as two half chunks side by side.  Same pruning rule, same result (the minimum and the lowest
// index within the tie tolerance do not depend on the order of the scan).
template <typename T, bool TERR>
__device__ __forceinline__ int hull_argmin_p(const DevModel<T> &m, int lane, int b, T r6, T r7, T r8, T pz, T tie, const T *terr,
                                             const T *Rw, T px, T py, T hmax_all, T &zmin_out) {
  const int hl = lane & (pk::LPE - 1);
  const T INF = Eps<T>::inf;
  const int v0 = m.hull_adr[b], v1 = m.hull_adr[b + 1];
  const int c0 = m.chunk_adr[b], nch = m.chunk_adr[b + 1] - c0;
  T zlb[2] = {INF, INF};
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int ch = hl + pk::LPE * r;
    if (ch < nch) {
      const T *bx = m.chunk_box + 6 * (c0 + ch);
      T zl = r6 * bx[0] + r7 * bx[1] + r8 * bx[2] + pz - (fabs(r6) * bx[3] + fabs(r7) * bx[4] + fabs(r8) * bx[5]);
      zl -= fabs(zl) * T(4) * Eps<T>::v;
      if constexpr (TERR) {
        const T a0 = terr[0] * Rw[0] + terr[1] * Rw[3], a1 = terr[0] * Rw[1] + terr[1] * Rw[4], a2 = terr[0] * Rw[2] + terr[1] * Rw[5];
        const T uc = a0 * bx[0] + a1 * bx[1] + a2 * bx[2] + terr[0] * px + terr[1] * py;
        const T ext = fabs(a0) * bx[3] + fabs(a1) * bx[4] + fabs(a2) * bx[5] + T(1e-6);
        const int k0 = (int)floor((uc - ext - terr[2]) * terr[3]), k1 = (int)floor((uc + ext - terr[2]) * terr[3]);
        T hm = hmax_all;
        if (k1 - k0 < 15) {
          hm = terr[4 + (k0 & 15)];
          for (int k = k0 + 1; k <= k1; k++) { const T hk = terr[4 + (k & 15)]; hm = hk > hm ? hk : hm; }
        }
        zl -= hm;
      }
      zlb[r] = zl;
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
  T zk0[2] = {INF, INF}, zk1[2] = {INF, INF}; // this lane's values in the two chunks scanned last
  int ck0 = -1, ck1 = -1;
  {
    // first chunk: lowest bound, lowest chunk index among equals
    T zl = zlb[0];
    int ci = hl;
    if (zlb[1] < zl) { zl = zlb[1]; ci = hl + pk::LPE; }
    pk::argmin(zl, ci);
    unsigned long long pend = 1ull << ci;
    while (pend) {
      const int c = __ffsll((long long)pend) - 1;
      const int i0 = v0 + WAVE * c + hl, i1 = i0 + pk::LPE;
      T z0 = INF, z1 = INF;
      if (i0 < v1) z0 = vert_val(i0);
      if (i1 < v1) z1 = vert_val(i1);
      zk1[0] = zk0[0]; zk1[1] = zk0[1]; ck1 = ck0; zk0[0] = z0; zk0[1] = z1; ck0 = c;
      const T zc2 = pk::min(z1 < z0 ? z1 : z0);
      zmin = zc2 < zmin ? zc2 : zmin;
      scanned |= 1ull << c;
      const unsigned p0 = pk::ballot(hl < nch && zlb[0] <= zmin + tie, lane), p1 = pk::ballot(hl + pk::LPE < nch && zlb[1] <= zmin + tie, lane);
      pend = (((unsigned long long)p1 << 32) | p0) & ~scanned;
    }
  }
  zmin_out = zmin;
  int best = 0x7fffffff;
  const T zt = zmin + tie;
  for (unsigned long long sm = scanned; sm && best == 0x7fffffff; sm &= sm - 1) {
    const int c = __ffsll((long long)sm) - 1;
    const int i0 = v0 + WAVE * c + hl, i1 = i0 + pk::LPE;
    T za, zb;
    if (c == ck0) { za = zk0[0]; zb = zk0[1]; }
    else if (c == ck1) { za = zk1[0]; zb = zk1[1]; }
    else { za = i0 < v1 ? vert_val(i0) : INF; zb = i1 < v1 ? vert_val(i1) : INF; }
    int cand = 0x7fffffff;
    if (i1 < v1 && zb <= zt) cand = i1;
    if (i0 < v1 && za <= zt) cand = i0;
    best = pk::min<int>(cand);
  }
  return best == 0x7fffffff ? v0 : best;
}

// newton_direction (tsidb_sim.hpp) for the packed layout.  `mode` is per env: an env whose factor is rebuilt (0) and one
// that applies row updates (1) share the row loads, the parking of the factor and the two substitutions.
template <typename T>
__device__ __noinline__ NewtonDir<T> newton_direction_p(const DevModel<T> &m, SimLds<T> &L, int lane, int nfl, int mode, unsigned chg, unsigned actbits,
                                                        T mu, T cD, T fD, T grad) {
  const int hl = lane & (pk::LPE - 1);
  NewtonDir<T> out;
  out.search = 0;
  out.status = 0;
  T arow[NV], rdv = 0;
#pragma unroll
  for (int j = 0; j < NV; j++) arow[j] = hl < NV ? L.H[hl * LDM + j] : T(0);
  bool store = true;
  if (mode == 0) {
    bool spd;
    chol26_factor_p<T>(arow, rdv, hl, spd);
    if (!spd) out.status = 1;
  } else {
    rdv = hl < NV ? L.H[hl * LDM + NV] : T(0);
    store = pk::ballot(chg != 0, lane) != 0u;
    while (out.status == 0) {
      const unsigned mk = pk::ballot(chg != 0, lane);
      if (!mk) break;
      const int src = __ffs(mk) - 1;
      const unsigned bits = (unsigned)pk::bc_dyn((int)chg, src, lane);
      const int bit = __ffs(bits) - 1;
      if (hl == src) chg &= ~(1u << bit);
      const T sigma = (((unsigned)pk::bc_dyn((int)actbits, src, lane) >> bit) & 1u) ? T(1) : T(-1);
      const int bk = hl < 6 ? 0 : hl - 5; // body of this lane's dof
      T xv = 0;
      unsigned pathm;
      if (bit == 0) { // friction row of dof `src`
        const T d = pk::bc_dyn(fD, src, lane);
        xv = hl == src ? d * rsqrt_t(d) : T(0);
        pathm = L.anc[src < 6 ? 0 : src - 5];
        if (hl > src) pathm = 0;
      } else { // row bit - 1 of contact `src`
        const int c = src, i = bit - 1;
        T cn[3], ct1[3], ct2[3];
        const int b1 = contact_frame(L, c, nfl, cn, ct1, ct2);
        const T muc = pk::bc_dyn(mu, c, lane), d = pk::bc_dyn(cD, c, lane);
        const T sg = i >= 4 ? T(0) : ((i & 1) ? -muc : muc);
        const T *tk = i < 2 ? ct1 : ct2;
        const T dl[3] = {cn[0] + sg * tk[0], cn[1] + sg * tk[1], cn[2] + sg * tk[2]};
        T rxd[3];
        cross3(L.cr[c], dl, rxd);
        const unsigned m2 = L.anc[L.cbody[c]], m1 = b1 >= 0 ? L.anc[b1] : 0u;
        pathm = m2 | m1;
        if (hl < NV) {
          const int sgn = (int)((m2 >> bk) & 1u) - (int)((m1 >> bk) & 1u);
          if (sgn != 0) {
            const T *Sk = L.S[hl];
            const T jv = Sk[0] * dl[0] + Sk[1] * dl[1] + Sk[2] * dl[2] + Sk[3] * rxd[0] + Sk[4] * rxd[1] + Sk[5] * rxd[2];
            const T sd = d * rsqrt_t(d);
            xv = sgn > 0 ? sd * jv : -(sd * jv);
          }
        }
      }
      const unsigned chain = pk::ballot(hl < NV && ((pathm >> bk) & 1u), lane);
      if (!chol26_rank1_p<T>(arow, rdv, xv, sigma, chain, hl)) out.status = 2;
    }
  }
  if (out.status == 0) {
    if (store && hl < NV) {
#pragma unroll
      for (int j = 0; j < NV; j++) L.H[hl * LDM + j] = arow[j];
      L.H[hl * LDM + NV] = rdv;
    }
    pk::sync();
    out.search = chol26_subst_p<T>(arow, L.H, rdv, grad, hl);
  }
  return out;
}

// sim_step_env<T, 1> (tsidb_sim.hpp) for the env on the caller's half of the wavefront.  lane = 0..63; L = the env's LDS.
template <typename T>
__device__ __forceinline__ void sim_step_pair(const DevModel<T> &m, SimLds<T> &L, int lane, const T *q_tsid, const T *v_tsid, T *qpos_g, T *qvel_g,
                                              T *qacc_ws_g, const T *envp, const T *terr_g, const T *motor_tau, T *qacc_out, int *ncon_out,
                                              int *con_out, int *info) {
  static_assert(SIM_PACKABLE, "two envs per wavefront need the robot's bodies, dofs, geoms and contacts on 32 lanes");
  constexpr int LPE = pk::LPE;
  const int hl = lane & (LPE - 1), hbase = lane & LPE;
  const T mscale = envp ? envp[0] : T(1);
  Floor<T> &fl = L.fl;
  if (hl == 0) {
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
  if (hl < NQ) {
    T val = qpos_g[hl];
    if (q_tsid) {
      if (hl < 3) val = q_tsid[hl];
      else if (hl < 7) val = quirks ? q_tsid[hl] : (hl == 3 ? q_tsid[6] : q_tsid[hl - 1]);
    }
    L.qpos[hl] = val;
  }
  if (hl < NV) {
    T val = qvel_g[hl];
    if (q_tsid && v_tsid && !quirks && hl < 6) {
      if (hl < 3) {
        const T x = q_tsid[3], y = q_tsid[4], z = q_tsid[5], w = q_tsid[6];
        const T r0 = hl == 0 ? 1 - 2 * (y * y + z * z) : hl == 1 ? 2 * (x * y + w * z) : 2 * (x * z - w * y);
        const T r1 = hl == 0 ? 2 * (x * y - w * z) : hl == 1 ? 1 - 2 * (x * x + z * z) : 2 * (y * z + w * x);
        const T r2 = hl == 0 ? 2 * (x * z + w * y) : hl == 1 ? 2 * (y * z - w * x) : 1 - 2 * (x * x + y * y);
        val = r0 * v_tsid[0] + r1 * v_tsid[1] + r2 * v_tsid[2];
      } else {
        val = v_tsid[hl];
      }
    }
    L.qvel[hl] = val;
  }
  const T myctrl = (hl < NA && q_tsid) ? q_tsid[m.mj_ctrl_qidx[hl]] : T(0);
  for (int i = hl; i < NV * LDM; i += LPE) L.M[i] = 0;
  pk::sync();

  // ---- kinematics, velocities, bias accelerations
  T Rb[9], pb[3], Vb[6], Ab[6], Sb[6], qd = 0;
  const int up0 = hl < NB ? m.mj_up[0][hl] : -1, up1 = hl < NB ? m.mj_up[1][hl] : -1, up2 = hl < NB ? m.mj_up[2][hl] : -1;
  const unsigned bodyanc = hl < NB ? m.mj_anc[hl] : 0u;
  if (hl < NB) L.anc[hl] = bodyanc;
#pragma unroll
  for (int i = 0; i < 6; i++) { Vb[i] = 0; Ab[i] = 0; Sb[i] = 0; }
#pragma unroll
  for (int i = 0; i < 3; i++) pb[i] = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) Rb[i] = 0;
  if (hl == 0) {
    quat_to_R(L.qpos[4], L.qpos[5], L.qpos[6], L.qpos[3], Rb);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        L.S[k][i] = (i == k) ? T(1) : T(0); L.S[k][3 + i] = 0;
        L.S[3 + k][i] = 0; L.S[3 + k][3 + i] = Rb[3 * i + k];
      }
    }
    T wl[3] = {L.qvel[3], L.qvel[4], L.qvel[5]};
    Vb[0] = L.qvel[0]; Vb[1] = L.qvel[1]; Vb[2] = L.qvel[2];
    mat3vec(Rb, wl, Vb + 3);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      T Sk[6] = {0, 0, 0, Rb[k], Rb[3 + k], Rb[6 + k]}, dS[6];
      cross_mm(Vb, Sk, dS);
#pragma unroll
      for (int i = 0; i < 6; i++) Ab[i] += dS[i] * wl[k];
    }
  } else if (hl < NB) {
    const T *Rq = m.mj_R[hl];
    const T th = L.qpos[6 + hl];
    T c, s;
    sincos_t(th, s, c);
    qd = L.qvel[5 + hl];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Rb[3 * i + 0] = c * Rq[3 * i] + s * Rq[3 * i + 1];
      Rb[3 * i + 1] = -s * Rq[3 * i] + c * Rq[3 * i + 1];
      Rb[3 * i + 2] = Rq[3 * i + 2];
      pb[i] = m.mj_pos[hl][i];
    }
  }
  int bchn[7];
#pragma unroll
  for (int d = 0; d < 7; d++) bchn[d] = hl < NB ? m.mj_chain[hl][d] : -1;
  tree_forward<T, 0>(hl, NB, up0, up1, up2, bchn, &L.R[0][0], &L.V[0][0], &L.f[0][0], 6, &L.Yc[0][0], 10, Rb, pb, qd, Sb, Vb, Ab);
  if (hl < NB) {
    const int b = hl;
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
    T fb[6] = {0, 0, 0, 0, 0, 0}, Y[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hl < NB) {
      const int b = hl;
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
    const int mylast = (hl < NB ? m.mj_last[hl] : hl) | hbase; // (a lane of this env's half)
#pragma unroll
    for (int i = 0; i < 6; i++) fb[i] = subtree_sum32(fb[i], mylast);
#pragma unroll
    for (int i = 0; i < 10; i++) Y[i] = subtree_sum32(Y[i], mylast);
    if (hl < NB) {
#pragma unroll
      for (int i = 0; i < 6; i++) L.f[hl][i] = fb[i];
#pragma unroll
      for (int i = 0; i < 10; i++) L.Yc[hl][i] = Y[i];
    }
  }
  pk::sync();
  // ---- per dof: bias, mass-matrix column (+ armature), actuation
  T qfs = 0;
  if (hl < NV) {
    const int k = hl, bk = k < 6 ? 0 : k - 5;
    T Sk[6], Fk[6], hk = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) { Sk[i] = L.S[k][i]; hk += Sk[i] * L.f[bk][i]; }
    yo_mul(L.Yc[bk], Sk, Fk);
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
  pk::sync();
  if (hl < NV) L.M[hl * LDM + hl] += m.mj_armature[hl];
  if (hl < NA) {
    const int d = m.mj_act_dof[hl];
    const T cc = myctrl < m.act_range[hl][0] ? m.act_range[hl][0] : (myctrl > m.act_range[hl][1] ? m.act_range[hl][1] : myctrl);
    T servo = m.mj_act_kp[hl] * (cc - L.qpos[d + 1]) - m.mj_act_kv[hl] * L.qvel[d];
    servo = servo < m.act_range[hl][2] ? m.act_range[hl][2] : (servo > m.act_range[hl][3] ? m.act_range[hl][3] : servo);
    L.xv[d] = motor_tau ? motor_tau[m.mj_ctrl_qidx[hl] - 7] : servo;
  } else if (hl < NA + 6) L.xv[hl - NA] = 0;
  pk::sync();
  if (hl < NV) qfs += L.xv[hl];
  // ---- floor collision, first half: bounding-sphere / box pretest for all geoms at once (lane = geom)
  const T margin = m.contact[10], tie_tol = m.opt[6];
  const T Ow[3] = {L.qpos[0], L.qpos[1], L.qpos[2]};
  const T nO = dot3(fl.n, Ow) - fl.d;
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
  unsigned cand_geoms;
  {
    bool near = false;
    if (hl < NG) {
      const int gb = m.geom_body[hl];
      const T *Rg = L.R[gb];
      const T c6 = fl.n[0] * Rg[0] + fl.n[1] * Rg[3] + fl.n[2] * Rg[6];
      const T c7 = fl.n[0] * Rg[1] + fl.n[1] * Rg[4] + fl.n[2] * Rg[7];
      const T c8 = fl.n[0] * Rg[2] + fl.n[1] * Rg[5] + fl.n[2] * Rg[8];
      const T pzl = dot3(fl.n, L.p[gb]) + nO;
      const T zc = c6 * m.rbound[hl][0] + c7 * m.rbound[hl][1] + c8 * m.rbound[hl][2] + pzl;
      near = !(zc - m.rbound[hl][3] - hmax_all > margin);
      const T *hb = m.hbox[hl];
      const T zb = c6 * hb[0] + c7 * hb[1] + c8 * hb[2] + pzl - (fabs(c6) * hb[3] + fabs(c7) * hb[4] + fabs(c8) * hb[5]);
      near = near && !(zb - fabs(zb) * T(8) * Eps<T>::v - hmax_all > margin);
    }
    cand_geoms = pk::ballot(near, lane);
  }
  TSIDB_STAMP(17);
  // ---- qacc_smooth = M^-1 qfrc_smooth
  T arow[NV];
  T qas = 0;
  int fail = 0;
  T *park = &L.cfv[0][0];
  {
#pragma unroll
    for (int j = 0; j < NV; j++) arow[j] = hl < NV ? L.M[hl * LDM + j] : T(0);
    bool spd;
    qas = chol26_solve_p<T, false>(arow, qfs, hl, spd);
    fail = spd ? 0 : 1;
    if (hl < NV) { park[hl] = qas; park[NV + hl] = qfs; }
  }

  TSIDB_STAMP(18);
  int ncon = 0, nfl = 0, cfail = 0;
  bool hh_cross = false;
  // ---- collision: floor against each candidate geom's hull
  if (has_terr) {
    if (hl < 20) L.terr[hl] = terr_g[hl];
    pk::sync();
  }
  const bool pm_rule = m.params[P_PLANE_MESH] != 0;
  for (unsigned bm = cand_geoms; bm; bm &= bm - 1) {
    const int g = __ffs(bm) - 1, b = m.geom_body[g];
    const T *Rbb = L.R[b];
    const T r6 = fl.n[0] * Rbb[0] + fl.n[1] * Rbb[3] + fl.n[2] * Rbb[6];
    const T r7 = fl.n[0] * Rbb[1] + fl.n[1] * Rbb[4] + fl.n[2] * Rbb[7];
    const T r8 = fl.n[0] * Rbb[2] + fl.n[1] * Rbb[5] + fl.n[2] * Rbb[8];
    const T pz = dot3(fl.n, L.p[b]) + nO;
    const int v0 = m.hull_adr[g];
    T zmin;
    const int best = has_terr ? hull_argmin_p<T, true>(m, lane, g, r6, r7, r8, pz, tie_tol, L.terr, Rbb, L.p[b][0] + Ow[0],
                                                       L.p[b][1] + Ow[1], hmax_all, zmin)
                              : hull_argmin_p<T, false>(m, lane, g, r6, r7, r8, pz, tie_tol, nullptr, nullptr, T(0), T(0), T(0), zmin);
    if (!(zmin <= margin)) continue;
    // the support vertex (entry 0), then its hull-graph neighbours (entries 1 .. nnb) within the margin; entry j sits on
    // lane j % 32, round j / 32 (a second round only for vertices with 32 and more neighbours)
    const int e0 = m.hull_eadr[best];
    int nnb = m.hull_eadr[best + 1] - e0;
    if (nnb > WAVE - 1) { nnb = WAVE - 1; cfail |= 16; }
    bool keep[2] = {false, false};
    T wd[2] = {0, 0}, cp[2][3] = {{0, 0, 0}, {0, 0, 0}};
    int vid[2] = {best, best};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int j = hl + LPE * r;
      if ((r == 0 || nnb >= LPE) && j <= nnb) {
        if (j > 0) vid[r] = v0 + m.hull_edge[e0 + j - 1];
        const T vv[3] = {m.hull_x[vid[r]], m.hull_y[vid[r]], m.hull_z[vid[r]]};
        T w[3];
        mat3vec(Rbb, vv, w);
        w[0] += L.p[b][0]; w[1] += L.p[b][1]; w[2] += L.p[b][2];
        wd[r] = dot3(fl.n, w) + nO;
        if (has_terr) wd[r] -= terrain_h(L.terr, w[0] + Ow[0], w[1] + Ow[1]);
        keep[r] = j == 0 || wd[r] <= margin;
#pragma unroll
        for (int i = 0; i < 3; i++) cp[r][i] = w[i] - T(0.5) * wd[r] * fl.n[i];
      } else if (r == 0) {
        // (lanes beyond the list: the one-env kernel's contact position of an idle lane is w = 0, wd = 0)
      }
    }
    if (pm_rule) {
      const T c00 = pk::bc1<0>(cp[0][0]), c01 = pk::bc1<0>(cp[0][1]), c02 = pk::bc1<0>(cp[0][2]);
      const T thr = T(0.3) * m.rbound[g][3];
      bool far[2];
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const T d3[3] = {cp[r][0] - c00, cp[r][1] - c01, cp[r][2] - c02};
        far[r] = (hl + LPE * r) > 0 && keep[r] && !(dot3(d3, d3) < thr * thr);
      }
      const unsigned long long fm = ((unsigned long long)pk::ballot(far[1], lane) << 32) | pk::ballot(far[0], lane);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int j = hl + LPE * r;
        keep[r] = j == 0 || (far[r] && __popcll(fm & ((1ull << j) - 1ull)) < 3);
      }
    }
    const unsigned long long mask = ((unsigned long long)pk::ballot(keep[1], lane) << 32) | pk::ballot(keep[0], lane);
    if (ncon + __popcll(mask) > MAXCON) cfail |= 8;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int j = hl + LPE * r;
      const int slot = ncon + __popcll(mask & ((1ull << j) - 1ull));
      if (keep[r] && slot < MAXCON) {
        L.cbody[slot] = b;
        L.cgeom[slot] = g;
        L.cvert[slot] = vid[r] - v0;
        L.cdist[slot] = wd[r];
#pragma unroll
        for (int i = 0; i < 3; i++) L.cr[slot][i] = cp[r][i];
#pragma unroll
        for (int d = 0; d < 7; d++) { const int a = m.mj_chain[b][d]; L.cchain[slot][d] = (unsigned char)(a > 0 ? a : 0); }
      }
    }
    ncon += __popcll(mask);
    ncon = ncon > MAXCON ? MAXCON : ncon;
  }
  nfl = ncon;
  // ---- collision: robot<->robot convex-hull pairs: mid phase one lane per pair (rounds of 32), MPR one pair at a time
#ifndef TSIDB_NO_HH
  if (m.params[P_SELF_COLLISION] != 0) {
    if (hl < NG) {
      const int gb = m.geom_body[hl];
      T c[3];
      mat3vec(L.R[gb], m.rbound[hl], c);
      L.scen[hl][0] = c[0] + L.p[gb][0]; L.scen[hl][1] = c[1] + L.p[gb][1]; L.scen[hl][2] = c[2] + L.p[gb][2];
      L.scen[hl][3] = m.rbound[hl][3];
    }
    pk::sync();
    int nsph = 0, ncand = 0;
    bool over = false, over64 = false;
    auto box_pass = [&](int cnt) { // pcand[0, cnt <= 32) -> survivors appended to fcand, in pair order
      const int k = hl < cnt ? L.pcand[hl] : 0;
      const bool may = hl < cnt && boxes_may_touch(m, L, m.pair_a[k], m.pair_b[k], margin);
      const unsigned mk = pk::ballot(may, lane);
      const int pos = ncand + __popc(mk & ((1u << hl) - 1u));
      if (may && pos < WAVE) L.fcand[pos] = k;
      ncand += __popc(mk);
    };
    for (int k0 = 0; k0 < m.npair; k0 += LPE) {
      const int k = k0 + hl;
      const bool may = k < m.npair && spheres_overlap(L, m.pair_a[k], m.pair_b[k], margin);
      const unsigned mk = pk::ballot(may, lane);
      if (may) L.pcand[nsph + __popc(mk & ((1u << hl) - 1u))] = k; // nsph <= 32 here: the list holds 128
      nsph += __popc(mk);
      if (nsph > LPE) {
        pk::sync();
        box_pass(LPE);
        const int rest = hl < nsph - LPE ? L.pcand[LPE + hl] : 0;
        pk::sync();
        if (hl < nsph - LPE) L.pcand[hl] = rest;
        nsph -= LPE;
        pk::sync();
      }
    }
    pk::sync();
    box_pass(nsph);
    if (ncand > WAVE) { ncand = WAVE; over64 = true; }
    pk::sync();
    for (int ci = 0; ci < ncand; ci++) {
      const int k = L.fcand[ci];
      const int ga = m.pair_a[k], gb = m.pair_b[k], a = m.geom_body[ga], b = m.geom_body[gb];
      T depth, dir[3], pos[3];
      if (!mpr_penetration<T, true>(m, L, lane, ga, gb, margin, depth, dir, pos)) continue;
      if (ncon - nfl >= MAXHH || ncon >= MAXCON) { over = true; continue; }
      if (hl == 0) {
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
  fail |= cfail;
  { // the robot<->robot contacts' normals and geom1 bodies move from the collision scratch to where the Newton loop keeps them
    T hv[2] = {0, 0};
    int hb = 0;
#pragma unroll
    for (int r = 0; r < 2; r++)
      if (hl + LPE * r < 3 * MAXHH) hv[r] = (&L.hn_s[0][0])[hl + LPE * r];
    if (hl < MAXHH) hb = L.hb1_s[hl];
    pk::sync();
#pragma unroll
    for (int r = 0; r < 2; r++)
      if (hl + LPE * r < 3 * MAXHH) (&L.hn[0][0])[hl + LPE * r] = hv[r];
    if (hl < MAXHH) L.hb1[hl] = hb;
  }
  pk::sync();
  qas = hl < NV ? park[hl] : T(0);
  qfs = hl < NV ? park[NV + hl] : T(0);
  if (hl == 0 && ncon_out) ncon_out[0] = ncon;
  if (con_out && hl < MAXCON) con_out[hl] = hl < ncon ? ((L.cgeom[hl] << 16) | L.cvert[hl]) : -1;

  TSIDB_STAMP(19);
  // ---- constraint rows: frictionloss (lane = dof), pyramidal contact rows (lane = contact)
  const T mu = (hl >= nfl || !envp) ? m.contact[0] : envp[1];
  const T dt = m.opt[0];
  const T timeconst = m.contact[1] > 2 * dt ? m.contact[1] : 2 * dt, dampratio = m.contact[2];
  const T dmin = m.contact[3], dmax = m.contact[4], width = m.contact[5], mid = m.contact[6], power = m.contact[7];
  const T kk = T(1) / (dmax * dmax * timeconst * timeconst * dampratio * dampratio), bb = T(2) / (dmax * timeconst);
  const T tcf = T(0.02) > 2 * dt ? T(0.02) : 2 * dt, bbf = T(2) / (dmax * tcf);
  RowState<T> rs;
  rs.has_f = hl < NV && m.mj_frictionloss[hl < NV ? hl : 0] > 0;
  rs.fD = rs.fR = rs.floss = rs.faref = rs.fjar = rs.fJv = 0;
  if (rs.has_f) {
    const T r = (1 - dmin) / dmin * m.mj_dof_invw0[hl];
    rs.fR = r > MINVAL ? r : MINVAL;
    rs.fD = T(1) / rs.fR;
    rs.floss = m.mj_frictionloss[hl];
    rs.faref = -bbf * L.qvel[hl];
  }
  rs.has_c = hl < ncon;
  rs.cD = 0;
#pragma unroll
  for (int i = 0; i < NROWC; i++) rs.caref[i] = rs.cjar[i] = rs.cJv[i] = 0;
  if (rs.has_c) {
    const int c = hl;
    const T dist = L.cdist[c];
    T x = fabs(dist - margin) / width, imp;
    if (x >= 1) imp = dmax;
    else if (x <= 0) imp = dmin;
    else {
      T y;
      if (power == T(2)) {
        if (x <= mid) y = x * x / mid;
        else y = 1 - (1 - x) * (1 - x) / (1 - mid);
      } else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
      else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
      imp = dmin + y * (dmax - dmin);
    }
    T tran = m.mj_body_invw0[L.cbody[c]][0];
    if (c >= nfl) tran += m.mj_body_invw0[L.hb1[c - nfl]][0];
    const T diagA = tran + mu * mu * tran;
    T R0 = (1 - imp) / imp * diagA;
    R0 = R0 > MINVAL ? R0 : MINVAL;
    rs.cD = T(1) / (2 * mu * mu * R0);
#pragma unroll
    for (int i = 0; i < NROWC; i++) rs.caref[i] = -kk * imp * (dist - margin);
  }

  int solver_iter = 0;
  T qacc = qas;
  const int nefc = NA + NROWC * ncon;
  T Ma = 0;
  if (nefc > 0) {
    auto stage = [&](T xa) { pk::sync(); if (hl < NV) L.xv[hl] = xa; pk::sync(); };
    T xw = hl < NV ? qacc_ws_g[hl] : T(0);
    stage(xw);
    Ma = mulM(L, L.xv, hl);
    T cjar_w[NROWC], cjar_s[NROWC];
#pragma unroll
    for (int i = 0; i < NROWC; i++) cjar_w[i] = cjar_s[i] = 0;
    if (rs.has_c) {
      const T *const xs[3] = {L.qvel, L.xv, park};
      T o[3][NROWC];
      contact_rows_multi<T, 3>(m, L, nfl, hl, xs, mu, o);
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
    T cost_w = pk::sum(cc + (hl < NV ? T(0.5) * (Ma - qfs) * (xw - qas) : T(0)));
    rs.fjar = fjar_s;
#pragma unroll
    for (int i = 0; i < NROWC; i++) rs.cjar[i] = cjar_s[i];
    rows_eval(rs, T(0), cc, gg, hh);
    T cost_s = pk::sum(cc);
    if (cost_w > cost_s) {
      qacc = qas;
      Ma = mulM(L, park, hl);
    } else {
      qacc = xw;
      rs.fjar = fjar_w;
#pragma unroll
      for (int i = 0; i < NROWC; i++) rs.cjar[i] = cjar_w[i];
    }

    TSIDB_STAMP(20);
    auto solver_tol = [&]() { const T t0 = m.opt[2]; return t0 > 64 * Eps<T>::v ? t0 : 64 * Eps<T>::v; };
    T cost = 0;
    int iter = 0;
    bool have_fac = false;
    const unsigned gfirst = pk::ballot(hl < nfl && (hl == 0 || L.cbody[hl] != L.cbody[hl > 0 ? hl - 1 : 0]), lane);
    const bool grouped = nfl > 2;
    unsigned prevbits = 0;
    TSIDB_LAP_ZERO(24); TSIDB_LAP_ZERO(25); TSIDB_LAP_ZERO(26); TSIDB_LAP_ZERO(27); TSIDB_LAP_ZERO(28);
    TSIDB_LAP_INIT();
    while (true) {
      // ---- constraint state at the current point: forces, active rows, cost
      rows_eval(rs, T(0), cc, gg, hh);
      const T gauss = pk::sum(hl < NV ? T(0.5) * (Ma - qfs) * (qacc - qas) : T(0));
      const T newcost = gauss + pk::sum(cc);
      T ff = 0;
      bool fact = false;
      if (rs.has_f) {
        const T f = rs.floss, r = rs.fR;
        if (rs.fjar <= -r * f) ff = f;
        else if (rs.fjar >= r * f) ff = -f;
        else { ff = -rs.fD * rs.fjar; fact = true; }
      }
      T Arow[6] = {0, 0, 0, 0, 0, 0};
      if (rs.has_c) {
        T fr[NROWC];
        T fv[3] = {0, 0, 0};
        T cn[3], ct1[3], ct2[3];
        contact_frame(L, hl, nfl, cn, ct1, ct2);
#pragma unroll
        for (int i = 0; i < NROWC; i++) {
          const bool act = rs.cjar[i] < 0;
          fr[i] = act ? -rs.cD * rs.cjar[i] : T(0);
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
        L.cfv[hl][0] = fv[0]; L.cfv[hl][1] = fv[1]; L.cfv[hl][2] = fv[2];
        if (grouped && hl < nfl) {
          T rxf[3];
          cross3(L.cr[hl], fv, rxf);
          T *gw = &L.K[0][0] + 6 * hl;
          gw[0] = fv[0]; gw[1] = fv[1]; gw[2] = fv[2]; gw[3] = rxf[0]; gw[4] = rxf[1]; gw[5] = rxf[2];
        }
      }
      pk::sync();
      // ---- gradient: Ma - qfrc_smooth - J^T force
      if (grouped) {
        for (unsigned gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffs(gm) - 1;
          const unsigned rest = gm & (gm - 1);
          const int ce = rest ? __ffs(rest) - 1 : nfl;
          if (hl < 6) {
            T acc = 0;
            for (int c = cf; c < ce; c++) acc += (&L.K[0][0])[6 * c + hl];
            (&L.K[0][0])[6 * cf + hl] = acc;
          }
        }
        pk::sync();
      }
      T grad = 0;
      if (hl < NV) {
        const int k = hl, bk = k < 6 ? 0 : k - 5;
        T s = 0;
        if (grouped)
          for (unsigned gm = gfirst; gm; gm &= gm - 1) {
            const int cf = __ffs(gm) - 1;
            if ((L.anc[L.cbody[cf]] >> bk) & 1u) {
              const T *gw = &L.K[0][0] + 6 * cf;
              s += L.S[k][0] * gw[0] + L.S[k][1] * gw[1] + L.S[k][2] * gw[2] + L.S[k][3] * gw[3] + L.S[k][4] * gw[4] + L.S[k][5] * gw[5];
            }
          }
        for (int c = grouped ? nfl : 0; c < ncon; c++) {
          int sgn = (int)((L.anc[L.cbody[c]] >> bk) & 1u);
          if (c >= nfl) sgn -= (int)((L.anc[L.hb1[c - nfl]] >> bk) & 1u);
          if (sgn == 0) continue;
          T rxf[3];
          cross3(L.cr[c], L.cfv[c], rxf);
          const T js = L.S[k][0] * L.cfv[c][0] + L.S[k][1] * L.cfv[c][1] + L.S[k][2] * L.cfv[c][2] +
                       L.S[k][3] * rxf[0] + L.S[k][4] * rxf[1] + L.S[k][5] * rxf[2];
          s += sgn > 0 ? js : -js;
        }
        grad = Ma - qfs - s - ff;
      }
      if (iter > 0) {
        const T gn = pk::sum(grad * grad);
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
      unsigned chg = 0;
      if (!full) {
        chg = actbits ^ prevbits;
        const int nchange = pk::sum_int(__popc(chg));
        if (nchange > NEWTON_INCR_MAX) full = true;
      }
      prevbits = actbits;
      // Two passes at most: an env whose row updates lose definiteness (status 2) comes round again with the full build.
      // An env that rebuilds and its neighbour that only updates rows share one call of newton_direction_p.
      for (int pass = 0; pass < 2; pass++) {
      if (full) {
      // ---- Newton Hessian H = M + J^T D J: CRB recursion on the per-body contact inertia
      if (rs.has_c) {
        const T *r = L.cr[hl];
        const T A3[9] = {Arow[0], Arow[1], Arow[2], Arow[1], Arow[3], Arow[4], Arow[2], Arow[4], Arow[5]};
        const T X[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
        T XA[9], XAXt[9], Xt[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) Xt[3 * i + j] = X[3 * j + i];
        mat3mul(X, A3, XA);
        mat3mul(XA, Xt, XAXt);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = i; j < 3; j++) { L.Wc[hl][sym_idx(i, j)] = A3[3 * i + j]; L.Wc[hl][sym_idx(3 + i, 3 + j)] = XAXt[3 * i + j]; }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) L.Wc[hl][sym_idx(i, 3 + j)] = XA[3 * j + i];
      }
      pk::sync();
      unsigned touched = 0;
      {
        for (unsigned gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffs(gm) - 1;
          const unsigned rest = gm & (gm - 1);
          const int ce = rest ? __ffs(rest) - 1 : nfl;
          touched |= L.anc[L.cbody[cf]];
          if (hl < 21) {
            T acc = 0;
            for (int c = cf; c < ce; c++) acc += L.Wc[c][hl];
            L.Wc[cf][hl] = acc;
          }
        }
      }
      pk::sync();
      for (int id = hl; id < NB * 21; id += LPE) {
        const int a = id / 21, e = id - 21 * a;
        T acc = 0;
        for (unsigned gm = gfirst; gm; gm &= gm - 1) {
          const int cf = __ffs(gm) - 1;
          if ((L.anc[L.cbody[cf]] >> a) & 1u) acc += L.Wc[cf][e];
        }
        L.K[a][e] = acc;
      }
      pk::sync();
      T Gk[6] = {0, 0, 0, 0, 0, 0};
      const bool mine = hl < NV && ((touched >> (hl < 6 ? 0 : hl - 5)) & 1u);
      if (mine) {
        const int k = hl, bk = k < 6 ? 0 : k - 5;
#pragma unroll
        for (int i = 0; i < 6; i++) {
          T sacc = 0;
#pragma unroll
          for (int j = 0; j < 6; j++) sacc += L.K[bk][sym_idx(i, j)] * L.S[k][j];
          Gk[i] = sacc;
        }
      }
      pk::sync();
      for (int i = hl; i < NV * LDM; i += LPE) L.H[i] = L.M[i];
      pk::sync();
      if (mine) {
        const int k = hl, bk = k < 6 ? 0 : k - 5;
        int chn[7];
#pragma unroll
        for (int d = 0; d < 7; d++) chn[d] = m.mj_chain[bk][d];
#pragma unroll
        for (int d = 0; d < 7 + 6; d++) {
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
      pk::sync();
      if (fact) L.H[hl * LDM + hl] += rs.fD;
      pk::sync();
      // robot<->robot contacts (rare): H += J^T A J on the LDS copy of H
      for (int c = nfl; c < ncon; c++) {
        T A6[6];
#pragma unroll
        for (int e = 0; e < 6; e++) A6[e] = pk::bc_dyn(Arow[e], c, lane);
        T jk[3] = {0, 0, 0};
        if (hl < NV) {
          const int bk = hl < 6 ? 0 : hl - 5;
          const int sgn = (int)((L.anc[L.cbody[c]] >> bk) & 1u) - (int)((L.anc[L.hb1[c - nfl]] >> bk) & 1u);
          if (sgn != 0) {
            T wxr[3];
            cross3(&L.S[hl][3], L.cr[c], wxr);
#pragma unroll
            for (int i = 0; i < 3; i++) jk[i] = sgn > 0 ? L.S[hl][i] + wxr[i] : -(L.S[hl][i] + wxr[i]);
          }
          T *g = &L.K[0][0] + 5 * hl;
          g[0] = A6[0] * jk[0] + A6[1] * jk[1] + A6[2] * jk[2];
          g[1] = A6[1] * jk[0] + A6[3] * jk[1] + A6[4] * jk[2];
          g[2] = A6[2] * jk[0] + A6[4] * jk[1] + A6[5] * jk[2];
        }
        pk::sync();
        if (hl < NV) {
          const T *g = &L.K[0][0];
          for (int k = 0; k < NV; k++) L.H[hl * LDM + k] += jk[0] * g[5 * k] + jk[1] * g[5 * k + 1] + jk[2] * g[5 * k + 2];
        }
        pk::sync();
      }
      } // (full)
      TSIDB_LAP(25);
      if (full && hh_cross) { // rare: the dense variant
        T dense_rows[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) dense_rows[j] = hl < NV ? L.H[hl * LDM + j] : T(0);
        search = -chol26_dense_p(dense_rows, grad, hl, ok);
        break;
      }
      const NewtonDir<T> nd = newton_direction_p<T>(m, L, lane, nfl, full ? 0 : 1, chg, actbits, mu, rs.cD, rs.fD, grad);
      if (nd.status == 2 && !full) { full = true; continue; } // a downdate lost definiteness: rebuild
      search = -nd.search;
      ok = nd.status == 0;
      have_fac = true;
      break;
      } // (pass)
      TSIDB_LAP(26);
      if (!ok) { fail |= 2; break; }
      // ---- exact line search along `search`
      stage(search);
      const T Mv = mulM(L, L.xv, hl);
      if (rs.has_f) rs.fJv = search;
      if (rs.has_c) contact_rows(m, L, nfl, hl, L.xv, mu, rs.cJv);
      T qg1 = hl < NV ? search * (Ma - qfs) : T(0), qg2 = hl < NV ? T(0.5) * search * Mv : T(0), snorm = hl < NV ? search * search : T(0);
      pk::sum3(qg1, qg2, snorm);
      snorm = sqrt(snorm);
      if (snorm < MINVAL) break;
      const T gtol = solver_tol() * m.opt[5] * snorm * m.meaninertia * NV;
      const int ls_iter = (int)m.opt[4];
      auto ls_eval = [&](T alpha, T &c, T &d1, T &d2) {
        T lc, lg, lh;
        rows_eval(rs, alpha, lc, lg, lh);
        pk::sum3(lc, lg, lh);
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
      TSIDB_LAP(27);
      iter++;
    }
    solver_iter = iter;
  }
  pk::sync();
  TSIDB_STAMP(21);
  // ---- semi-implicit Euler, write back
  const T dte = m.opt[0];
  if (hl < NV) {
    const T vn = L.qvel[hl] + dte * qacc;
    L.qvel[hl] = vn;
    qvel_g[hl] = vn;
    qacc_ws_g[hl] = qacc;
    if (qacc_out) qacc_out[hl] = qacc;
  }
  pk::sync();
  if (hl < 3) L.qpos[hl] += dte * L.qvel[hl];
  if (hl >= 6 && hl < NV) L.qpos[hl + 1] += dte * L.qvel[hl];
  if (hl == 3) {
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
  pk::sync();
  if (hl < NQ) qpos_g[hl] = L.qpos[hl];
  if (hl == 0 && info) { info[2] = solver_iter; info[3] = fail; }
  TSIDB_STAMP(22);
}

} // namespace tsidb
