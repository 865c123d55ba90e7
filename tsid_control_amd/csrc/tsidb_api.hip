// tsidb_api.hip - kernels and the C-ABI of libtsidb.so (include/tsidb.h).  gfx950 only.
#include "../../include/tsidb.h"
#include "tsidb_common.hpp"
#include "tsidb_sim.hpp"
#include "tsidb_sim2.hpp"
#include "tsidb_tick.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace tsidb;

// ============================================================================ kernels
// walking reference update (Walk_Planner.py:23-31 samples -> WalkController.py:189-253): 16 lanes per env (r = 0..15).  Runs
// either as its own kernel (k_walk, tsidb_walk_update) or in the prologue of k_tick (tsidb_tick_walk: one launch less on
// the tick stream, which is what bounds small batches).
template <typename T>
struct WalkArgs {
  const T *coef; const int *side; const int *nsteps; const T *rest; const T *com; int K;
  T t_now; const T *t_off; T Tstep, t_start, omega, z0, dz;
  const T *frames; T *foot_ref; T *contact_ref; uint8_t *cact; T *com_ref;
  const int *ncon; const int *con; int *latch; unsigned long long fgeoms0, fgeoms1; T td_frac; const double *t_dev;
};
template <typename T>
__device__ __forceinline__ void walk_update_env(const WalkArgs<T> &wa, int e, int r) {
  const T *coef = wa.coef, *rest = wa.rest, *com = wa.com, *t_off = wa.t_off, *frames = wa.frames;
  const int *side = wa.side, *nsteps = wa.nsteps, *ncon = wa.ncon, *con = wa.con;
  int *latch = wa.latch;
  T *foot_ref = wa.foot_ref, *contact_ref = wa.contact_ref, *com_ref = wa.com_ref;
  uint8_t *cact = wa.cact;
  const int K = wa.K;
  const T t_now = wa.t_now, Tstep = wa.Tstep, t_start = wa.t_start, omega = wa.omega, z0 = wa.z0, dz = wa.dz, td_frac = wa.td_frac;
  const unsigned long long fgeoms0 = wa.fgeoms0, fgeoms1 = wa.fgeoms1;
  const double *t_dev = wa.t_dev;
  // 16 lanes per env: every lane evaluates the (cheap) polynomials, each writes its share of the rows, so
  // the table reads hit one line per env and the reference rows are written as contiguous runs
  const size_t E = (size_t)e;
  // per-env start delay (de-phased schedules): the env's own clock starts at t_off[e]
  T t = t_dev ? (T)t_dev[0] : t_now; // device clock (always float64: a float32 clock advanced by dt per tick drifts by
                                     // ~one dt over a few thousand ticks): a captured graph replays with the time it finds there
  if (t_off) { t -= t_off[e]; t = t > 0 ? t : T(0); }
  // timeline: [0, t_start) both feet down; step k in [t_start + k T, t_start + (k+1) T); then the final stand
  const int k = t < t_start ? -1 : (int)floor((t - t_start) / Tstep);
  const T s = k < 0 ? t : (t - t_start) - k * Tstep;
  const int ns = nsteps[e];
  const bool walking = k >= 0 && k < ns;
  const int kk = k < 0 ? 0 : k;
  const int kc = kk < (ns > 0 ? ns - 1 : 0) ? kk : (ns > 0 ? ns - 1 : 0);
  const int sd = side[E * K + kc];
  const int kr = kk < ns ? kk : ns;
  // contact-timing feedback (closed loop): if the sim reported the swing foot on the floor in the last part of its
  // swing, the step's touch-down is taken now - the foot counts as a stance foot for the rest of step k (latched per
  // env), so the contact is added at the placement the foot really has
  bool early = false;
  if (latch) {
    const int lk = latch[e];
    early = walking && lk == k;
    if (walking && !early && s > td_frac * Tstep) {
      const unsigned long long fg = sd == 0 ? fgeoms0 : fgeoms1; // the collision geoms of the swing foot's body
      const int nc = ncon[e];
      for (int c = 0; c < nc; c++) {
        const int cp = con[E * MAXCON + c];
        if (((fg >> (cp >> 16)) & 1ull) && !(cp & 0x8000)) early = true;
      }
    }
  }
  const bool act[2] = {cact[E * 2] != 0, cact[E * 2 + 1] != 0}; // read by every lane before lane 0 rewrites them
  const T *c = coef + (E * K + kc) * 16;
  const T pw[4] = {T(1), s, s * s, s * s * s}, d1[4] = {T(0), T(1), 2 * s, 3 * s * s}, d2[4] = {T(0), T(0), T(2), 6 * s};
  T pos[4], vel[4], acc[4];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    pos[a] = vel[a] = acc[a] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { pos[a] += c[4 * a + i] * pw[i]; vel[a] += c[4 * a + i] * d1[i]; acc[a] += c[4 * a + i] * d2[i]; }
  }
#pragma unroll
  for (int f = 0; f < 2; f++) {
    const bool swing = walking && sd == f && !early;
    const T *rs = rest + ((E * (K + 1) + kr) * 2 + f) * 4;
    const T x = swing ? pos[0] : rs[0], y = swing ? pos[1] : rs[1], z = swing ? pos[2] : rs[3], yaw = swing ? pos[3] : rs[2];
    const T cy = cos(yaw), sy = sin(yaw);
    const bool active = act[f];
    const T *fr = frames + E * 24 + 12 * f; // R row-major, p
    T *fo = foot_ref + E * 48 + 24 * f;
    // element i of the tsid SE3 sample (p, R column-major, v6, a6) / of the current placement as an SE3 vector
    auto smp_at = [&](int i) -> T {
      if (i < 3) return i == 0 ? x : (i == 1 ? y : z);
      if (i < 12) {
        const int q = i - 3;
        return q == 0 ? cy : q == 1 ? sy : q == 3 ? -sy : q == 4 ? cy : q == 8 ? T(1) : T(0);
      }
      if (!swing) return T(0);
      const int q = i - 12, a = q < 6 ? q : q - 6;
      const T *src = q < 6 ? vel : acc;
      return a < 3 ? (a == 0 ? src[0] : a == 1 ? src[1] : src[2]) : (a == 5 ? src[3] : T(0));
    };
    auto cur_at = [&](int i) -> T { return i < 3 ? fr[9 + i] : fr[((i - 3) % 3) * 3 + (i - 3) / 3]; };
    if (!swing && !active && r < 12) contact_ref[E * 24 + 12 * f + r] = cur_at(r); // add_contact at the current placement
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = r + 16 * h;
      if (i < 24) fo[i] = (swing && active) ? (i < 12 ? cur_at(i) : T(0)) // remove_contact: the foot task restarts here
                                            : smp_at(i);
    }
    if (r == 0) {
      if (!swing && !active) cact[E * 2 + f] = 1;
      if (swing && active) cact[E * 2 + f] = 0;
    }
  }
  if (latch && r == 0 && early) latch[e] = k; // (every lane of the env read the old value above: same wavefront)
  // CoM reference: LIPM segment (zmp, d, c) of the current phase in the plane, quintic descent in height
  if (r < 9) {
    const int ph = k + 1 < ns + 1 ? k + 1 : ns + 1;
    const T sc = (k >= 0 && ph > ns) ? (t - t_start) - ns * Tstep : s;
    const int a = r % 3, d = r / 3; // axis, derivative order
    T val;
    if (a < 2) {
      const T ep = exp(omega * sc), em = exp(-omega * sc);
      const T *sg = com + ((E * (K + 2) + ph) * 2 + a) * 3;
      const T u = T(0.5) * sg[1] * ep + sg[2] * em;
      val = d == 0 ? sg[0] + u : d == 1 ? omega * (T(0.5) * sg[1] * ep - sg[2] * em) : omega * omega * u;
    } else {
      const T q = t_start > 0 ? (t < t_start ? t / t_start : T(1)) : T(1);
      const T sz = q * q * q * (10 - 15 * q + 6 * q * q);
      const T dsz = t_start > 0 ? 30 * q * q * (1 - q) * (1 - q) / t_start : T(0);
      const T ddsz = t_start > 0 ? 60 * q * (1 - q) * (1 - 2 * q) / (t_start * t_start) : T(0);
      val = d == 0 ? z0 - dz * sz : d == 1 ? -dz * dsz : -dz * ddsz;
    }
    com_ref[E * 9 + 3 * d + a] = val;
  }
}

// One launch, three bodies: the tick is compiled for every contact configuration (NS = 2, 1, 0 feet in
// contact: 50 / 38 / 26 variables, every loop bound a compile-time constant) and each workgroup branches
// to the body of its env.  Three separate kernels (tried first) allocate registers a little better, but
// a variant with nothing to do still queues 4096 workgroups that each need the full LDS allocation, and
// in one stream it holds the working variant back behind whatever else occupies the GPU (the sim of
// the previous step): 11.0 M -> 12.9 M env-steps/s from merging them.
// COP: the variant with the CoP force task rows (legacy/biped.py:79-80) compiled in; the reference's
// ctrl/WalkController.py stack (w_cop = 0) runs the variant without them.
#ifndef TSIDB_WPE
#define TSIDB_WPE 2 // wavefronts per SIMD the float64 kernels are register-allocated for (LDS allows 2: DESIGN.md section 5)
#endif
template <typename T, bool COP>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(TSIDB_WPE))) void k_tick(const DevModel<T> *__restrict__ mp, int n, T *q, T *v, const T *com_ref,
                                                  const T *posture_ref, const T *foot_ref, const T *contact_ref,
                                                  const uint8_t *cact, const T *cop_frames, T *tau, T *dv, T *f,
                                                  int *status, T *obs, int obs_ld, T *frames, int *info, const T *qpos_sim,
                                                  const T *qvel_sim, const T *cop_ref, WalkArgs<T> wa, T *q_snap, T *v_snap, int fast_eq) {
  __shared__ TickLds<T> L;
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n) return;
  const int e = env_of_block(blockIdx.x, n);
  const size_t E = (size_t)e;
  if (wa.coef) { // this tick's walking reference update first (tsidb_tick_walk): it rewrites the references read below
    if (lane < 16) walk_update_env<T>(wa, e, lane);
    __syncthreads();
  }
  const int ns = (cact[E * 2] != 0) + (cact[E * 2 + 1] != 0);
  {
    // a non-finite state or reference never enters the solver: the env is flagged HQP_STATUS_ERROR (4) and
    // left untouched (the dual active-set loop's exit tests are comparisons, which NaN makes meaningless)
    const T *qs = qpos_sim ? qpos_sim + E * NQ : q + E * NQ, *vs = qvel_sim ? qvel_sim + E * NV : v + E * NV;
    T chk = 0;
    if (lane < NQ) chk += fabs(qs[lane]);
    if (lane < NV) chk += fabs(vs[lane]);
    if (lane < 9) chk += fabs(com_ref[E * 9 + lane]);
    if (lane < NA) chk += fabs(posture_ref[E * NA + lane]);
    if (lane < 48) chk += fabs(foot_ref[E * 48 + lane]);
    if (lane < 24) chk += fabs(contact_ref[E * 24 + lane]);
    if (__ballot(!(chk <= Eps<T>::inf))) {
      // nothing of an earlier tick is handed on either: tau = dv = f = 0, as after a failed solve (in the closed loop
      // the env's motors go limp instead of being driven by stale torques)
      if (lane < NA) tau[E * NA + lane] = 0;
      if (lane < NV) dv[E * NV + lane] = 0;
      if (lane < 24) f[E * 24 + lane] = 0;
      if (lane == 0) {
        status[e] = 4;
        if (info) { info[E * 4] = 0; info[E * 4 + 1] = 0; }
        if (obs && obs_ld >= NROW) { obs[E * obs_ld + NOBS] = 0; obs[E * obs_ld + NOBS + 1] = 1; }
      }
      if (q_snap && lane < NQ) q_snap[E * NQ + lane] = q[E * NQ + lane]; // (the snapshot is the state as it stands)
      if (v_snap && lane < NV) v_snap[E * NV + lane] = v[E * NV + lane];
      return;
    }
  }
#ifdef TSIDB_ONLY_NS // (diagnostic builds: one body, to read its resource usage alone)
  if (ns != TSIDB_ONLY_NS) return;
#endif
  if (ns == 2) {
    tsid_tick_env<T, 2, COP>(*mp, L, lane, q + E * NQ, v + E * NV, com_ref + E * 9, posture_ref + E * NA, foot_ref + E * 48,
                         contact_ref + E * 24, cact + E * 2, cop_frames ? cop_frames + E * 24 : nullptr, tau + E * NA,
                         dv + E * NV, f + E * 24, status + e, obs ? obs + E * obs_ld : nullptr, (obs && obs_ld >= NROW) ? obs + E * obs_ld + NOBS : nullptr, info ? info + E * 4 : nullptr,
                         qpos_sim ? qpos_sim + E * NQ : nullptr, qvel_sim ? qvel_sim + E * NV : nullptr, cop_ref ? cop_ref + E * 3 : nullptr, fast_eq != 0);
  } else if (ns == 1) {
    tsid_tick_env<T, 1, COP>(*mp, L, lane, q + E * NQ, v + E * NV, com_ref + E * 9, posture_ref + E * NA, foot_ref + E * 48,
                         contact_ref + E * 24, cact + E * 2, cop_frames ? cop_frames + E * 24 : nullptr, tau + E * NA,
                         dv + E * NV, f + E * 24, status + e, obs ? obs + E * obs_ld : nullptr, (obs && obs_ld >= NROW) ? obs + E * obs_ld + NOBS : nullptr, info ? info + E * 4 : nullptr,
                         qpos_sim ? qpos_sim + E * NQ : nullptr, qvel_sim ? qvel_sim + E * NV : nullptr, cop_ref ? cop_ref + E * 3 : nullptr, fast_eq != 0);
  } else {
    tsid_tick_env<T, 0, COP>(*mp, L, lane, q + E * NQ, v + E * NV, com_ref + E * 9, posture_ref + E * NA, foot_ref + E * 48,
                         contact_ref + E * 24, cact + E * 2, cop_frames ? cop_frames + E * 24 : nullptr, tau + E * NA,
                         dv + E * NV, f + E * 24, status + e, obs ? obs + E * obs_ld : nullptr, (obs && obs_ld >= NROW) ? obs + E * obs_ld + NOBS : nullptr, info ? info + E * 4 : nullptr,
                         qpos_sim ? qpos_sim + E * NQ : nullptr, qvel_sim ? qvel_sim + E * NV : nullptr, cop_ref ? cop_ref + E * 3 : nullptr, fast_eq != 0);
  }
  if (frames && lane < 24) frames[E * 24 + lane] = L.oMf[lane / 12][lane % 12];
  // a second copy of the TSID state this tick ends on (what q / v hold now): the sim stage of the pipelined step reads it
  // from there while the next tick already overwrites q / v (two copy kernels per step otherwise)
  if (q_snap && lane < NQ) q_snap[E * NQ + lane] = L.qs[lane];
  if (v_snap && lane < NV) v_snap[E * NV + lane] = L.vs[lane];
}

// NW = wavefronts per env: 1 when the batch fills the GPU, 2 for small batches (tsidb_sim.hpp: sim_step_env).
// B consecutive sim steps per launch (tsidb_sim_batch; the pipelined open-loop step hands over a batch of TSID state snapshots):
// envs do not interact, so each workgroup simply steps its env B times - no launch gaps between the steps, which at small
// batches were 10 % of a step.  B = 1 is the plain tsidb_sim.
// (the snapshots of a batch are slots of one [K, N, NQ] / [K, N, NV] ring: two base pointers and the slot numbers packed four
//  bits each - sixteen separate pointers were 32 SGPRs live across the whole step body)
template <typename T> struct SimRing { const T *q, *v; unsigned long long slots; };
// WPE = wavefronts per SIMD the kernel is register-allocated for.  float64 needs 256 registers for its matrix rows: two.
// float32 (10.6 KB of LDS per env: 15 workgroups per CU by LDS) has a second build for THREE (168 VGPRs, 65 spilled): alone
// that kernel is 5 % slower, but the third slot per SIMD lets the tick kernel of the next step (146 VGPRs) run beside it -
// pipelined 18.9 -> 23.6 M env-steps/s at 4096 walkers, 21.3 -> 24.9 M at 16 384; below ~3000 envs, where a step is one
// wavefront's latency, and in the closed loop (tick and sim back to back) the spills cost 2-9 %, so launch_sim picks it for
// open-loop steps of 3072 envs and more (profiles/r04_f32_wpe3.txt, r04_f32_wpe3_sizes.txt).  Same arithmetic: bit-identical.
template <typename T, int NW, bool MULTI, int WPE = TSIDB_WPE>
__global__ __launch_bounds__(WAVE * NW) __attribute__((amdgpu_waves_per_eu(WPE))) void k_sim(const DevModel<T> *__restrict__ mp, int n, int B, SimRing<T> ring, T *qpos, T *qvel,
                                              T *qacc_ws, const T *env_params, const T *terrain, const T *motor_tau, T *qacc, int *ncon,
                                              int *con, int *info) {
  __shared__ SimLds<T> L;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if ((int)blockIdx.x >= n) return;
  const int e = env_of_block(blockIdx.x, n);
  const size_t E = (size_t)e;
  for (int b = 0; b < (MULTI ? B : 1); b++) { // (MULTI = false: one step, no loop in the code at all)
    // (the model pointer is laundered per step: otherwise every load of a model constant in the 1200-line step body is
    //  loop-invariant, gets hoisted in front of the loop and stays live across the whole step - the kernel, at its register
    //  limit already, then spilled a hundred values to scratch and ran 20 % slower)
    const DevModel<T> *mq = mp;
    if constexpr (MULTI) {
      asm volatile("" : "+s"(mq), "+s"(qpos), "+s"(qvel), "+s"(qacc_ws), "+s"(env_params), "+s"(terrain), "+s"(motor_tau) : : "memory");
      asm volatile("" : "+s"(qacc), "+s"(ncon), "+s"(con), "+s"(info) : : "memory");
    }
    const size_t slot = (size_t)((ring.slots >> (4 * b)) & 15ull);
    const T *q_tsid = ring.q ? ring.q + slot * (size_t)n * NQ : nullptr, *v_tsid = ring.v ? ring.v + slot * (size_t)n * NV : nullptr;
    bool skip;
    {
      // non-finite sim state / targets, or a sim state that has diverged (sum of |qpos| + |qvel| beyond SIM_STATE_BOUND:
      // the reference's own loop gets there, its teleported sim accumulates velocity until the contact forces explode -
      // and products of such values overflow to inf / NaN inside the step): skip the step, failure bit 4 in info[3]
      T chk = 0, big = 0;
      if (lane < NQ) { big += fabs(qpos[E * NQ + lane]); chk += q_tsid ? fabs(q_tsid[E * NQ + lane]) : T(0); }
      if (lane < NV) { big += fabs(qvel[E * NV + lane]); chk += fabs(qacc_ws[E * NV + lane]) + (v_tsid ? fabs(v_tsid[E * NV + lane]) : T(0)); }
      if (lane < NA && motor_tau) chk += fabs(motor_tau[E * NA + lane]);
      big = wave_sum(big);
      skip = __ballot(!(chk <= Eps<T>::inf)) || !(big <= T(SIM_STATE_BOUND));
      if (skip && wv == 0) {
        if (lane == 0) {
          if (info) { info[E * 4 + 2] = 0; info[E * 4 + 3] = 4; }
          if (ncon) ncon[e] = 0;
        }
        if (con && lane < MAXCON) con[E * MAXCON + lane] = -1;
      }
    }
    if (!skip)
      sim_step_env<T, NW>(*mq, L, lane, wv, q_tsid ? q_tsid + E * NQ : nullptr, v_tsid ? v_tsid + E * NV : nullptr, qpos + E * NQ, qvel + E * NV, qacc_ws + E * NV,
                          env_params ? env_params + E * 8 : nullptr, terrain ? terrain + E * 20 : nullptr, motor_tau ? motor_tau + E * NA : nullptr,
                          qacc ? qacc + E * NV : nullptr, ncon ? ncon + e : nullptr, con ? con + E * MAXCON : nullptr,
                          info ? info + E * 4 : nullptr);
    if constexpr (MULTI) __syncthreads(); // both wavefronts; the step's state is written before the next step reads it
  }
}

// Two envs per wavefront (tsidb_sim2.hpp; TSIDB_OPT_SIM_PACK): env 2 p + h on lanes [32 h, 32 h + 32) of workgroup p's one
// wavefront, each half with its own SimLds (2 x 20 KB: four workgroups = eight envs per CU, one wavefront per SIMD with the
// whole register file).  One step per launch.  Per env bit-identical to k_sim<T, 1, false>.
template <typename T>
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_sim2(const DevModel<T> *__restrict__ mp, int n, SimRing<T> ring, T *qpos, T *qvel,
                                              T *qacc_ws, const T *env_params, const T *terrain, const T *motor_tau, T *qacc, int *ncon,
                                              int *con, int *info) {
  if constexpr (SIM_PACKABLE) {
    __shared__ SimLds<T> Ls[2];
    const int lane = threadIdx.x, hf = lane >> 5, hl = lane & (pk::LPE - 1);
    const int npair = (n + 1) / 2;
    if ((int)blockIdx.x >= npair) return;
    const int e2 = 2 * env_of_block(blockIdx.x, npair) + hf;
    const bool live = e2 < n; // (odd n: the last wavefront's second half has no env - it reads env n - 1 and writes nothing)
    const int e = live ? e2 : n - 1;
    const size_t E = (size_t)e;
    const size_t slot = (size_t)(ring.slots & 15ull);
    const T *q_tsid = ring.q ? ring.q + slot * (size_t)n * NQ : nullptr, *v_tsid = ring.v ? ring.v + slot * (size_t)n * NV : nullptr;
    T chk = 0, big = 0;
    if (hl < NQ) { big += fabs(qpos[E * NQ + hl]); chk += q_tsid ? fabs(q_tsid[E * NQ + hl]) : T(0); }
    if (hl < NV) { big += fabs(qvel[E * NV + hl]); chk += fabs(qacc_ws[E * NV + hl]) + (v_tsid ? fabs(v_tsid[E * NV + hl]) : T(0)); }
    if (hl < NA && motor_tau) chk += fabs(motor_tau[E * NA + hl]);
    big = pk::sum(big);
    const bool skip = pk::ballot(!(chk <= Eps<T>::inf), lane) != 0u || !(big <= T(SIM_STATE_BOUND));
    if (!live) return;
    if (skip) {
      if (hl == 0) {
        if (info) { info[E * 4 + 2] = 0; info[E * 4 + 3] = 4; }
        if (ncon) ncon[e] = 0;
      }
      if (con && hl < MAXCON) con[E * MAXCON + hl] = -1;
      return;
    }
    sim_step_pair<T>(*mp, Ls[hf], lane, q_tsid ? q_tsid + E * NQ : nullptr, v_tsid ? v_tsid + E * NV : nullptr, qpos + E * NQ, qvel + E * NV, qacc_ws + E * NV,
                     env_params ? env_params + E * 8 : nullptr, terrain ? terrain + E * 20 : nullptr, motor_tau ? motor_tau + E * NA : nullptr,
                     qacc ? qacc + E * NV : nullptr, ncon ? ncon + e : nullptr, con ? con + E * MAXCON : nullptr, info ? info + E * 4 : nullptr);
  }
}

template <typename T>
__global__ __launch_bounds__(WAVE) void k_rbd(const DevModel<T> *__restrict__ mp, int n, const T *q, const T *v, T *M, T *hb,
                                              T *Jcom, T *Jf, T *oMf, T *com) {
  __shared__ TickLds<T> L;
  const int lane = threadIdx.x;
  if ((int)blockIdx.x >= n) return;
  const int e = env_of_block(blockIdx.x, n);
  const size_t E = (size_t)e;
  if (lane < NQ) L.qs[lane] = q[E * NQ + lane];
  if (lane < NV) L.vs[lane] = v[E * NV + lane];
  __syncthreads();
  rbd_terms<T>(*mp, L, lane);
  for (int i = lane; i < NV * NV; i += WAVE) M[E * NV * NV + i] = L.Dyn[(i / NV) * LDD + i % NV];
  if (lane < NV) hb[E * NV + lane] = L.h[lane];
  for (int i = lane; i < 3 * NV; i += WAVE) Jcom[E * 3 * NV + i] = L.k.Jcom[(i / NV) * LDF + i % NV];
  for (int i = lane; i < 12 * NV; i += WAVE) Jf[E * 12 * NV + i] = L.k.Jf[(i / NV) * LDF + i % NV];
  if (lane < 24) oMf[E * 24 + lane] = L.oMf[lane / 12][lane % 12];
  if (lane < 3) com[E * 3 + lane] = L.com[lane];
}

// reset: standing state + references (WalkController.py:22-26,72-79,81,122,151-152,164-165; main.py:57-64)
// done_rows (may be NULL): the [N, rows_ld] rows k_tick writes - only envs whose done flag (column NOBS + 1) is set are reset
// (episode lifecycle on the device: no host round trip between `done` and the restart).  posture_bias (may be NULL, [NA]):
// added to the captured posture reference (a walking workload's bent-knee posture).  frames (may be NULL): receives the
// sole placements, as the host facade copies them after a reset.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_reset(const DevModel<T> *__restrict__ mp, int n, const int *env_ids, int n_ids, T *q,
                                                T *v, T *qpos, T *qvel, T *qacc_ws, T *com_ref, T *posture_ref,
                                                T *foot_ref, T *contact_ref, uint8_t *cact, T *cop_frames, T *cop_ref,
                                                const T *done_rows, int rows_ld, const T *posture_bias, T *frames) {
  __shared__ TickLds<T> L;
  const DevModel<T> &m = *mp;
  const int lane = threadIdx.x;
  int e = blockIdx.x;
  if (env_ids) {
    if (e >= n_ids) return;
    e = env_ids[e];
  }
  if (e < 0 || e >= n) return;
  const size_t E = (size_t)e;
  if (done_rows && !(done_rows[E * rows_ld + NOBS + 1] != T(0))) return;
  if (lane < NQ) L.qs[lane] = m.q0[lane];
  if (lane < NV) L.vs[lane] = 0;
  __syncthreads();
  rbd_terms<T>(m, L, lane);
  const T zlf = L.oMf[0][11];
  __syncthreads();
  if (lane == 0) L.qs[2] -= zlf; // WalkController.py:74
  __syncthreads();
  rbd_terms<T>(m, L, lane);
  if (lane < NQ) q[E * NQ + lane] = L.qs[lane];
  if (lane < NV) { v[E * NV + lane] = 0; qvel[E * NV + lane] = 0; qacc_ws[E * NV + lane] = 0; }
  if (lane < NQ) {
    T val = L.qs[lane]; // main.py:64: raw copy (quirks F6a/F6b); joints are all zero in "standing"
    if (m.params[P_QUIRKS] == 0 || m.params[P_CLOSED_LOOP] != 0) {
      if (lane == 3) val = L.qs[6];
      else if (lane > 3 && lane < 7) val = L.qs[lane - 1];
      else if (lane >= 7) val = L.qs[m.mj_ctrl_qidx[lane - 7]];
    }
    qpos[E * NQ + lane] = val;
  }
  if (lane < 24) {
    const int f = lane / 12, i = lane % 12;
    // SE3ToVector layout: p(3), R column-major(9)
    T val = i < 3 ? L.oMf[f][9 + i] : L.oMf[f][3 * ((i - 3) % 3) + (i - 3) / 3];
    contact_ref[E * 24 + lane] = val;
    cop_frames[E * 24 + lane] = L.oMf[f][i];
    if (frames) frames[E * 24 + lane] = L.oMf[f][i];
  }
  if (lane < 48) {
    const int i = lane % 24;
    // foot tasks never get a reference in the reference (quirk F6c): identity placement
    foot_ref[E * 48 + lane] = (i == 3 || i == 7 || i == 11) ? T(1) : T(0);
  }
  if (lane < 9) com_ref[E * 9 + lane] = lane < 3 ? L.com[lane] : T(0);
  if (lane < NA) posture_ref[E * NA + lane] = L.qs[7 + lane] + (posture_bias ? posture_bias[lane] : T(0));
  if (lane < 2) cact[E * 2 + lane] = 1;
  // CoP task reference: between the soles, on the floor
  if (cop_ref && lane < 3) cop_ref[E * 3 + lane] = lane < 2 ? T(0.5) * (L.oMf[0][9 + lane] + L.oMf[1][9 + lane]) : T(0);
}

template <typename T>
__global__ __launch_bounds__(256) void k_walk(int n, WalkArgs<T> wa) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = gid >> 4, r = gid & 15;
  if (e >= n) return;
  walk_update_env<T>(wa, e, r);
}

// ---------------------------------------------------------------------------- episode plan on the device
// One thread per env: everything a walking episode needs from a path - footsteps (ctrl/Footstep_Planner.py:92-125: a step
// every step_length of accumulated path length at pos + t L/2 +- n W/2, yaw along the tangent, closing steps :114-123),
// swing polynomials from footstep k to k + 2 (ctrl/Walk_Planner.py:23-31, ctrl/Foot_Trajectory.py:5-27: x, y, yaw linear,
// z the parabola / cubic through the knots), rest placements, and the CoM plan: DCM end points backwards from the final
// stand, then one LIPM segment (ctrl/LIPM.py:34-49 in closed form) per phase - the tables WalkSchedule.__init__ builds on
// the host, so that a reset with a new path never leaves the GPU.  Arithmetic in float64 whatever the path's type.
__device__ __forceinline__ unsigned long long plan_hash(unsigned long long seed, unsigned long long env, unsigned long long episode) {
  unsigned long long x = seed ^ (env * 0x9E3779B97F4A7C15ull) ^ (episode * 0xD1B54A32D192ED03ull);
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ void plan_poly(int nk, const double *x, const double *f, double *c) { // Newton's divided differences
  double d1[3] = {0, 0, 0}, d2[2] = {0, 0}, d3 = 0;
  for (int i = 0; i + 1 < nk; i++) d1[i] = (f[i + 1] - f[i]) / (x[i + 1] - x[i]);
  for (int i = 0; i + 2 < nk; i++) d2[i] = (d1[i + 1] - d1[i]) / (x[i + 2] - x[i]);
  if (nk == 4) d3 = (d2[1] - d2[0]) / (x[3] - x[0]);
  const double x1 = nk > 2 ? x[1] : 0.0, x2 = nk > 3 ? x[2] : 0.0;
  c[0] = f[0];
  c[1] = d1[0] - d2[0] * x1 + d3 * x1 * x2;
  c[2] = d2[0] - d3 * (x1 + x2);
  c[3] = d3;
}
struct PlanParams { double v[16]; };
template <typename T>
__global__ __launch_bounds__(64) void k_plan(int n, const int *env_ids, int n_ids, const T *done_rows, int rows_ld, PlanParams PP,
                                             const T *cop_frames, const T *com_ref, const double *path, const int *npts, int P,
                                             const double *scale, int *episode, int bump, int K, double *steps, T *coef, int *side,
                                             int *nsteps, T *rest, T *com, int *flags, T *t_offset, int *latch, double t_now,
                                             const double *t_dev) {
  const double *pp = PP.v;
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (env_ids) {
    if (e >= n_ids) return;
    e = env_ids[e];
  }
  if (e < 0 || e >= n) return;
  const size_t E = (size_t)e;
  if (done_rows && !(done_rows[E * rows_ld + NOBS + 1] != T(0))) return;
  const double L = pp[0], Tst = pp[3], rise_ratio = pp[4], t_start = pp[5], foot_press = pp[7], ds = pp[8];
  const T *fr = cop_frames + E * 24;
  const double lf[2] = {(double)fr[9], (double)fr[10]}, rf[2] = {(double)fr[21], (double)fr[22]};
  const double com0[3] = {(double)com_ref[E * 9], (double)com_ref[E * 9 + 1], (double)com_ref[E * 9 + 2]};
  const double heading = atan2(-(lf[0] - rf[0]), lf[1] - rf[1]);
  const double ch = cos(heading), sh = sin(heading), mid[2] = {0.5 * (lf[0] + rf[0]), 0.5 * (lf[1] + rf[1])};
  double *st = steps + E * (K + 2) * 4; // x, y, yaw, side per footstep
  int nst = 0, cut = 0;
  auto put = [&](double x, double y, double yaw, int sd) { st[4 * nst] = x; st[4 * nst + 1] = y; st[4 * nst + 2] = yaw; st[4 * nst + 3] = sd; nst++; };
  auto add_step = [&](double dx, double dy, int sd, const double *pos) {
    if (nst >= K + 2) { cut = 1; return; }
    const double nrm = sqrt(dx * dx + dy * dy), tx = dx / nrm, ty = dy / nrm, sign = sd == 0 ? 1.0 : -1.0;
    put(pos[0] + tx * (pp[0] / 2) + (-ty) * (pp[1] / 2 * sign), pos[1] + ty * (pp[0] / 2) + tx * (pp[1] / 2 * sign), atan2(dy, dx), sd);
  };
  put(lf[0], lf[1], heading, 0);
  put(rf[0], rf[1], heading, 1);
  int sd = 1;
  double travelled = 0, dx = 0, dy = 0, prev[2] = {0, 0}, a[2] = {0, 0}, sc = 1.0;
  if (!path) {
    if (episode && bump) episode[e] += 1;
    if (scale) sc = scale[e];
    else if (episode) sc = pp[13] + (pp[14] - pp[13]) * ((double)(plan_hash((unsigned long long)pp[15], (unsigned long long)e, (unsigned long long)episode[e]) >> 11) * (1.0 / 9007199254740992.0));
  }
  int nv = path ? npts[e] : (int)pp[12];
  if (path && nv > P) nv = P; // (never read past the env's row of the path table)
  double ux = 0, uy = 0, uth = 0;
  for (int i = 0; i < nv; i++) {
    double b[2];
    if (path) { b[0] = path[(E * P + i) * 2]; b[1] = path[(E * P + i) * 2 + 1]; }
    else {
      ux += pp[9] * pp[11] * cos(uth); uy += pp[9] * pp[11] * sin(uth); uth += pp[10] * pp[11];
      const double px = ux * sc, py = uy * sc;
      b[0] = (ch * px - sh * py) + mid[0]; b[1] = (sh * px + ch * py) + mid[1];
    }
    if (i == 0) { prev[0] = a[0] = b[0]; prev[1] = a[1] = b[1]; continue; }
    const double seg = sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]));
    int mres = ds > 0 && seg / ds < 4096.0 ? (int)ceil(seg / ds) : (ds > 0 ? 4096 : 1);
    if (ds > 0 && !(seg / ds < 4096.0)) cut |= 4; // a piece of more than 4096 resample intervals (or a non-finite vertex): resampled coarser, flag bit 4
    if (mres < 1) mres = 1;
    for (int j = 1; j <= mres; j++) {
      const double qf = (double)j / mres, p[2] = {a[0] + (b[0] - a[0]) * qf, a[1] + (b[1] - a[1]) * qf};
      const double ddx = p[0] - prev[0], ddy = p[1] - prev[1];
      if (ddx != 0 || ddy != 0) { dx = ddx; dy = ddy; } // (a repeated vertex keeps the direction of the piece before it)
      travelled += hypot(ddx, ddy);
      if (travelled >= L) { sd = !sd; add_step(dx, dy, sd, prev); travelled = 0; }
      prev[0] = p[0]; prev[1] = p[1];
    }
    a[0] = b[0]; a[1] = b[1];
  }
  // a path without a direction (fewer than two distinct vertices; the reference's planner raises on it) plans no step: the env
  // stands, flag bit 1
  // the LIPM needs a positive CoM height above the feet after the descent (plan() before a reset, or com_drop >= that height,
  // would put NaN into every table): such an env plans no step and stands, flag bit 8
  const bool no_height = !(com0[2] - pp[6] > 1e-3);
  const bool degenerate = !(dx != 0 || dy != 0) || no_height;
  if (no_height) nst = 2;
  if (!degenerate) {
    sd = !sd;
    add_step(dx, dy, sd, prev);
    if (travelled > 0) { sd = !sd; add_step(dx, dy, sd, prev); }
  }
  const int ns = nst - 2;
  nsteps[e] = ns;
  if (flags) flags[e] = cut | (degenerate ? 2 : 0) | (no_height ? 8 : 0);
  for (int k = nst; k < K + 2; k++) st[4 * k] = st[4 * k + 1] = st[4 * k + 2] = st[4 * k + 3] = 0;
  // swing polynomials and rest placements; yaw relative to the initial heading
  double cur[2][4];
  for (int s2 = 0; s2 < 2; s2++) {
    const int f = (int)st[4 * s2 + 3];
    cur[f][0] = st[4 * s2]; cur[f][1] = st[4 * s2 + 1]; cur[f][2] = st[4 * s2 + 2] - heading; cur[f][3] = 0;
  }
  for (int k = 0; k <= K; k++) {
    T *ro = rest + (E * (K + 1) + k) * 8;
    for (int i = 0; i < 8; i++) ro[i] = (T)cur[i / 4][i % 4];
    if (k >= K) break;
    double c[16];
    for (int i = 0; i < 16; i++) c[i] = 0;
    int sw = 0;
    if (k < ns) {
      sw = (int)st[4 * k + 3];
      const double nxt[4] = {st[4 * (k + 2)], st[4 * (k + 2) + 1], st[4 * (k + 2) + 2] - heading, -foot_press};
      const double x2[2] = {0, Tst};
      double f2[2];
      f2[0] = cur[sw][0]; f2[1] = nxt[0]; plan_poly(2, x2, f2, c + 0);
      f2[0] = cur[sw][1]; f2[1] = nxt[1]; plan_poly(2, x2, f2, c + 4);
      f2[0] = cur[sw][2]; f2[1] = nxt[2]; plan_poly(2, x2, f2, c + 12);
      if (rise_ratio != 0.5) {
        const double rise = Tst * rise_ratio, x4[4] = {0, rise, Tst - rise, Tst}, f4[4] = {cur[sw][3], cur[sw][3] + pp[2], nxt[3] + pp[2], nxt[3]};
        plan_poly(4, x4, f4, c + 8);
      } else {
        const double x3[3] = {0, Tst * rise_ratio, Tst}, f3[3] = {cur[sw][3], cur[sw][3] + pp[2], nxt[3]};
        plan_poly(3, x3, f3, c + 8);
      }
      for (int i = 0; i < 4; i++) cur[sw][i] = nxt[i];
    }
    T *co = coef + (E * K + k) * 16;
    for (int i = 0; i < 16; i++) co[i] = (T)c[i];
    side[E * K + k] = sw;
  }
  // CoM plan: DCM end points backwards (kept in the table's d slot), then forwards one LIPM segment per phase
  const double w = sqrt(9.80665 / (no_height ? 1e-3 : com0[2] - pp[6])), ewT = exp(-w * Tst);
  T *cm = com + E * (K + 2) * 6;
  double fin[2] = {com0[0], com0[1]}, xi[2];
  if (ns > 0) { fin[0] = 0.5 * (st[4 * ns] + st[4 * (ns + 1)]); fin[1] = 0.5 * (st[4 * ns + 1] + st[4 * (ns + 1) + 1]); }
  xi[0] = fin[0]; xi[1] = fin[1];
  for (int k = ns - 1; k >= 0; k--) // xi_k waits in the d slot of phase k + 1 for the forward pass (rounded to T there)
    for (int ax = 0; ax < 2; ax++) { const double z = st[4 * (k + 1) + ax]; xi[ax] = z + (xi[ax] - z) * ewT; cm[((k + 1) * 2 + ax) * 3 + 1] = (T)xi[ax]; }
  double x[2] = {com0[0], com0[1]};
  const double ep0 = exp(w * t_start), em0 = exp(-w * t_start), epT = exp(w * Tst), emT = exp(-w * Tst);
  for (int ax = 0; ax < 2; ax++) {
    const double x0k = ns > 0 ? (sizeof(T) == 8 ? (double)cm[(1 * 2 + ax) * 3 + 1] : xi[ax]) : fin[ax];
    const double z = (x0k - x[ax] * ep0) / (1.0 - ep0), d = x[ax] - z, c2 = (x[ax] - z) - 0.5 * d;
    cm[ax * 3] = (T)z; cm[ax * 3 + 1] = (T)d; cm[ax * 3 + 2] = (T)c2;
    x[ax] = z + (0.5 * d * ep0 + c2 * em0);
  }
  auto xi_at = [&](int k0, int ax) { // float32 tables cannot park a float64 DCM end point: run the recursion again from the end
    double vv = fin[ax];
    for (int k = ns - 1; k >= k0; k--) { const double z = st[4 * (k + 1) + ax]; vv = z + (vv - z) * ewT; }
    return vv;
  };
  for (int k = 0; k < ns; k++)
    for (int ax = 0; ax < 2; ax++) {
      T *sg = cm + ((k + 1) * 2 + ax) * 3;
      const double z = st[4 * (k + 1) + ax], d = (sizeof(T) == 8 ? (double)sg[1] : xi_at(k, ax)) - z, c2 = (x[ax] - z) - 0.5 * d;
      sg[0] = (T)z; sg[1] = (T)d; sg[2] = (T)c2;
      x[ax] = z + (0.5 * d * epT + c2 * emT);
    }
  for (int k = ns + 1; k < K + 2; k++)
    for (int ax = 0; ax < 2; ax++) { T *sg = cm + (k * 2 + ax) * 3; sg[0] = (T)fin[ax]; sg[1] = 0; sg[2] = (T)(x[ax] - fin[ax]); }
  // the env's clock restarts: its timeline runs on t - t_offset
  if (t_offset) t_offset[e] = (T)(t_dev ? t_dev[0] : t_now);
  if (latch) latch[e] = -1;
}

// ============================================================================ host side
namespace {

struct Sect {
  char name[24];
  uint32_t dtype, count;
  uint64_t offset;
};

struct Blob {
  std::vector<uint8_t> raw;
  // every section must lie inside the blob (a truncated or corrupt file must not make the host read past it)
  void validate() const {
    const size_t nb = raw.size();
    if (nb < 16) throw std::string("model blob: truncated header");
    uint32_t n;
    memcpy(&n, raw.data() + 8, 4);
    if (n > 4096 || 16 + (size_t)n * sizeof(Sect) > nb) throw std::string("model blob: bad section table");
    const Sect *s = (const Sect *)(raw.data() + 16);
    for (uint32_t i = 0; i < n; i++) {
      const size_t esz = s[i].dtype == 0 ? 8 : 4;
      if (s[i].dtype > 1 || (s[i].offset & 7) || s[i].offset > nb || (size_t)s[i].count * esz > nb - s[i].offset)
        throw std::string("model blob: bad section ") + std::string(s[i].name, strnlen(s[i].name, 24));
    }
  }
  const Sect *find(const char *name) const {
    uint32_t n;
    memcpy(&n, raw.data() + 8, 4);
    const Sect *s = (const Sect *)(raw.data() + 16);
    for (uint32_t i = 0; i < n; i++)
      if (strncmp(s[i].name, name, 24) == 0) return &s[i];
    return nullptr;
  }
  const double *f64(const char *name, uint32_t cnt) const {
    const Sect *s = find(name);
    if (!s || s->dtype != 0 || (cnt && s->count != cnt)) throw std::string("model blob: bad section ") + name;
    return (const double *)(raw.data() + s->offset);
  }
  const int *i32(const char *name, uint32_t cnt) const {
    const Sect *s = find(name);
    if (!s || s->dtype != 1 || (cnt && s->count != cnt)) throw std::string("model blob: bad section ") + name;
    return (const int *)(raw.data() + s->offset);
  }
  uint32_t count(const char *name) const {
    const Sect *s = find(name);
    return s ? s->count : 0;
  }
};

void quat_wxyz_to_R_host(const double *q, double *R) {
  double w = q[0], x = q[1], y = q[2], z = q[3], n = 1.0 / std::sqrt(w * w + x * x + y * y + z * z);
  w *= n; x *= n; y *= n; z *= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

template <int N>
void tree_tables(const int *parent, int n, int *depth, int *nchild, int (*child)[MAXCHILD], unsigned *anc, int *maxdepth,
                 int *last, int (*up)[N]) {
  // subtree sums are taken as differences of a prefix scan over the lanes, which needs the numbering to be
  // a depth-first pre-order (every subtree a contiguous index range j..last[j]); both pinocchio's joint
  // order and MuJoCo's body order are
  if (n > 32) throw std::string("model blob: more than 32 tree nodes");
  for (int j = 0; j < n; j++) last[j] = j;
  for (int j = n - 1; j > 0; j--) {
    if (parent[j] < 0 || parent[j] >= j) throw std::string("model blob: tree nodes are not numbered parent-first");
    if (last[j] > last[parent[j]]) last[parent[j]] = last[j];
  }
  for (int j = 0; j < n; j++) { // every index in (j, last[j]] must descend from j
    for (int i = j + 1; i <= last[j]; i++) {
      int a = i;
      while (a > j) a = parent[a];
      if (a != j) throw std::string("model blob: tree numbering is not a depth-first pre-order");
    }
  }
  *maxdepth = 0;
  for (int j = 0; j < n; j++) {
    nchild[j] = 0;
    depth[j] = parent[j] < 0 ? 0 : depth[parent[j]] + 1;
    anc[j] = (parent[j] < 0 ? 0u : anc[parent[j]]) | (1u << j);
    if (depth[j] > *maxdepth) *maxdepth = depth[j];
  }
  for (int j = 0; j < n; j++)
    if (parent[j] >= 0) {
      int p = parent[j];
      if (nchild[p] >= MAXCHILD) throw std::string("model blob: too many children per body");
      child[p][nchild[p]++] = j;
    }
  // ancestors 1, 2, 4 levels up for the pointer-jumping forward pass (three rounds cover depth <= 7)
  if (*maxdepth > 7) throw std::string("model blob: kinematic tree deeper than 7 levels");
  for (int j = 0; j < n; j++) up[0][j] = parent[j];
  for (int r = 1; r < 3; r++)
    for (int j = 0; j < n; j++) up[r][j] = up[r - 1][j] < 0 ? -1 : up[r - 1][up[r - 1][j]];
}

// chain[j] = j and its ancestors other than the root, deepest first, -1 padded (tree depth <= 7, checked by tree_tables)
static void chain_table(const int *parent, int n, int (*chain)[8]) {
  for (int j = 0; j < n; j++) {
    int k = 0;
    for (int i = 0; i < 8; i++) chain[j][i] = -1;
    for (int a = j; a > 0 && k < 8; a = parent[a]) chain[j][k++] = a;
  }
}

} // namespace

struct tsidb_ctx {
  int device = 0, dtype = 0, num_envs = 0;
  int sim_waves = 1; // wavefronts per env in k_sim (tsidb_set_option)
  std::vector<hipStream_t> used_streams; // streams this handle has launched model-reading kernels on (tsidb_set_params waits for them)
  void note_stream(hipStream_t s) {
    for (hipStream_t x : used_streams) if (x == s) return;
    if (used_streams.size() < 32) used_streams.push_back(s);
  }
  int qp_fast_eq = 1; // the tick tries the equality-constrained optimum by a PP x PP Cholesky before the QR (TSIDB_OPT_QP_FAST_EQ)
  int sim_pack = 0;  // two envs per wavefront in the sim kernel (k_sim2; one step per launch, robots that fit 32 lanes)
  unsigned lds_pad = 0; // diagnostic: unused dynamic LDS per workgroup of k_tick / k_sim (occupancy experiments)
  int cu_split = -1;    // tsidb_stream_create: tick and sim streams on disjoint halves of the CUs (-1 = up to 512 envs)
  Blob blob;
  std::vector<double> params;
  void *d_model = nullptr, *d_hull = nullptr, *d_box = nullptr;
  int *d_eadr = nullptr, *d_edge = nullptr;
  const void *com_ref = nullptr, *posture_ref = nullptr, *foot_ref = nullptr, *contact_ref = nullptr, *cop_frames = nullptr;
  const uint8_t *contact_active = nullptr;
  const void *env_params = nullptr, *terrain = nullptr, *cop_ref = nullptr, *posture_bias = nullptr;
  int foot_body[2] = {-1, -1}; // sim bodies that carry the left / right sole frame
  unsigned long long foot_geoms[2] = {0, 0}; // bit g: collision geom g is on that body
  std::string err;
};

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) throw std::string(#call " failed: ") + hipGetErrorString(e_); \
  } while (0)

template <typename T>
static void build_model(tsidb_ctx *h, DevModel<T> &m) {
  const Blob &b = h->blob;
  memset(&m, 0, sizeof m);
  // ---- TSID side
  memcpy(m.pin_parent, b.i32("pin_parent", NJ), sizeof m.pin_parent);
  tree_tables(m.pin_parent, NJ, m.pin_depth, m.pin_nchild, m.pin_child, m.pin_anc, &m.pin_maxdepth, m.pin_last, m.pin_up);
  chain_table(m.pin_parent, NJ, m.pin_chain);
  const double *pl = b.f64("pin_place", NJ * 12), *in = b.f64("pin_inertia", NJ * 10);
  double mass = 0;
  for (int j = 0; j < NJ; j++) {
    for (int i = 0; i < 12; i++) m.pin_place[j][i] = (T)pl[12 * j + i];
    for (int i = 0; i < 10; i++) m.pin_inertia[j][i] = (T)in[10 * j + i];
    mass += in[10 * j];
  }
  m.mass = (T)mass;
  memcpy(m.frame_parent, b.i32("pin_frame_parent", 2), sizeof m.frame_parent);
  const double *fp = b.f64("pin_frame_place", 24), *q0 = b.f64("pin_q0", NQ);
  for (int i = 0; i < 24; i++) m.frame_place[i / 12][i % 12] = (T)fp[i];
  for (int i = 0; i < NQ; i++) m.q0[i] = (T)q0[i];
  // ---- parameters and the constant QP blocks derived from them
  const double *P = h->params.data();
  for (int i = 0; i < P_COUNT; i++) m.params[i] = (T)P[i];
  double Tg[6][12] = {{0}};
  for (int i = 0; i < 4; i++) {
    const double *p = P + P_CPOINTS + 3 * i;
    for (int k = 0; k < 3; k++) Tg[k][3 * i + k] = 1.0;
    Tg[3][3 * i + 1] = -p[2]; Tg[3][3 * i + 2] = p[1];
    Tg[4][3 * i + 0] = p[2];  Tg[4][3 * i + 2] = -p[0];
    Tg[5][3 * i + 0] = -p[1]; Tg[5][3 * i + 1] = p[0];
  }
  for (int i = 0; i < 6; i++) for (int c = 0; c < 12; c++) m.Tgen[i][c] = (T)Tg[i][c];
  { // friction pyramid (Contact6d::updateForceInequalityConstraints)
    const double *n = P + P_NORMAL, mu = P[P_MU];
    auto cr = [](const double *a, const double *c, double *o) {
      o[0] = a[1] * c[2] - a[2] * c[1]; o[1] = a[2] * c[0] - a[0] * c[2]; o[2] = a[0] * c[1] - a[1] * c[0];
    };
    double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, t1[3], t2[3];
    cr(n, ex, t1);
    if (std::sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]) < 1e-5) cr(n, ey, t1);
    cr(n, t1, t2);
    double n1 = std::sqrt(t1[0] * t1[0] + t1[1] * t1[1] + t1[2] * t1[2]), n2 = std::sqrt(t2[0] * t2[0] + t2[1] * t2[1] + t2[2] * t2[2]);
    for (int i = 0; i < 3; i++) { t1[i] /= n1; t2[i] /= n2; }
    for (int i = 0; i < 4; i++)
      for (int k = 0; k < 3; k++) {
        m.Bcone[4 * i + 0][3 * i + k] = (T)(-t1[k] - mu * n[k]);
        m.Bcone[4 * i + 1][3 * i + k] = (T)(t1[k] - mu * n[k]);
        m.Bcone[4 * i + 2][3 * i + k] = (T)(-t2[k] - mu * n[k]);
        m.Bcone[4 * i + 3][3 * i + k] = (T)(t2[k] - mu * n[k]);
        m.Bcone[16][3 * i + k] = (T)n[k];
      }
    for (int i = 0; i < 16; i++) { m.cone_lb[i] = (T)-1e10; m.cone_ub[i] = 0; }
    for (int k = 0; k < 3; k++) { m.cop_t[0][k] = (T)t1[k]; m.cop_t[1][k] = (T)t2[k]; }
    m.cone_lb[16] = (T)P[P_FMIN];
    m.cone_ub[16] = (T)P[P_FMAX];
  }
  { // force-regularisation Hessian block: w (W T)^T (W T) + reg I, its Cholesky factor inverse-transposed
    const double wreg[6] = {1, 1, 1e-3, 2, 2, 2};
    double Hf[12][12] = {{0}}, Lf[12][12] = {{0}}, X[12][12] = {{0}};
    for (int a = 0; a < 12; a++)
      for (int c = 0; c < 12; c++) {
        double s = 0;
        for (int i = 0; i < 6; i++) s += wreg[i] * Tg[i][a] * wreg[i] * Tg[i][c];
        Hf[a][c] = P[P_W_FORCEREF] * s + (a == c ? P[P_HESS_REG] : 0.0);
      }
    double tr = 0;
    for (int a = 0; a < 12; a++) tr += Hf[a][a];
    for (int j = 0; j < 12; j++) {
      double s = Hf[j][j];
      for (int k = 0; k < j; k++) s -= Lf[j][k] * Lf[j][k];
      if (!(s > 0)) throw std::string("force-regularisation block is not positive definite");
      Lf[j][j] = std::sqrt(s);
      for (int i = j + 1; i < 12; i++) {
        double t = Hf[i][j];
        for (int k = 0; k < j; k++) t -= Lf[i][k] * Lf[j][k];
        Lf[i][j] = t / Lf[j][j];
      }
    }
    for (int c = 0; c < 12; c++) // X = L^-1 by forward substitution
      for (int i = 0; i < 12; i++) {
        double s = i == c ? 1.0 : 0.0;
        for (int k = 0; k < i; k++) s -= Lf[i][k] * X[k][c];
        X[i][c] = s / Lf[i][i];
      }
    double trJ = 0;
    for (int a = 0; a < 12; a++) {
      for (int c = 0; c < 12; c++) m.Jf0[a][c] = (T)X[c][a]; // L^-T
      trJ += X[a][a];
    }
    m.Hf_trace = (T)tr;
    m.Jf0_trace = (T)trJ;
  }
  // ---- sim side (absent in a TSID-only build)
  if constexpr (!TOPO_HAS_SIM) return;
  memcpy(m.mj_parent, b.i32("mj_parent", NB), sizeof m.mj_parent);
  for (int j = 0; j < NB; j++)
    if (m.mj_parent[j] != TOPO_PARENT[j])
      throw std::string("model blob's sim tree differs from the topology this library was compiled for "
                        "(regenerate csrc/tsidb_topology.hpp with model_compiler.py and rebuild)");
  tree_tables(m.mj_parent, NB, m.mj_depth, m.mj_nchild, m.mj_child, m.mj_anc, &m.mj_maxdepth, m.mj_last, m.mj_up);
  chain_table(m.mj_parent, NB, m.mj_chain);
  const double *mp = b.f64("mj_pos", NB * 3), *mq = b.f64("mj_quat", NB * 4), *mi = b.f64("mj_inertia", NB * 10);
  for (int j = 0; j < NB; j++) {
    double R[9];
    quat_wxyz_to_R_host(mq + 4 * j, R);
    for (int i = 0; i < 9; i++) m.mj_R[j][i] = (T)R[i];
    for (int i = 0; i < 3; i++) m.mj_pos[j][i] = (T)mp[3 * j + i];
    for (int i = 0; i < 10; i++) m.mj_inertia[j][i] = (T)mi[10 * j + i];
  }
  const double *arm = b.f64("mj_armature", NV), *fl = b.f64("mj_frictionloss", NV), *iw = b.f64("mj_dof_invw0", NV);
  const double *bw = b.f64("mj_body_invw0", NB * 2), *M0 = b.f64("mj_dof_M0", NV);
  double mean = 0;
  for (int i = 0; i < NV; i++) {
    m.mj_armature[i] = (T)arm[i]; m.mj_frictionloss[i] = (T)(fl[i] * P[P_SIM_FLOSS_SCALE]); m.mj_dof_invw0[i] = (T)iw[i];
    mean += M0[i];
  }
  m.meaninertia = (T)(mean / NV);
  for (int i = 0; i < NB * 2; i++) m.mj_body_invw0[i / 2][i % 2] = (T)bw[i];
  memcpy(m.mj_act_dof, b.i32("mj_act_dof", NA), sizeof m.mj_act_dof);
  memcpy(m.mj_ctrl_qidx, b.i32("mj_ctrl_qidx", NA), sizeof m.mj_ctrl_qidx);
  for (int a2 = 0; a2 < NA; a2++) m.tsid2sim[m.mj_ctrl_qidx[a2] - 7] = a2;
  const double *kp = b.f64("mj_act_kp", NA), *kv = b.f64("mj_act_kv", NA);
  for (int i = 0; i < NA; i++) { m.mj_act_kp[i] = (T)kp[i]; m.mj_act_kv[i] = (T)kv[i]; }
  memcpy(m.hull_adr, b.i32("mj_hull_adr", NG + 1), sizeof m.hull_adr);
  memcpy(m.geom_body, b.i32("mj_geom_body", NG), sizeof m.geom_body);
  for (int g = 0; g < NG; g++)
    if (m.geom_body[g] < 0 || m.geom_body[g] >= NB || (g > 0 && m.geom_body[g] < m.geom_body[g - 1]))
      throw std::string("model blob: geoms must be listed in body order");
  const double *rb = b.f64("mj_rbound", NG * 4), *op = b.f64("mj_opt", 7), *ct = b.f64("mj_contact", 12);
  for (int i = 0; i < NG * 4; i++) m.rbound[i / 4][i % 4] = (T)rb[i];
  if ((int)ct[8] != CONDIM) throw std::string("model blob: contact dimension differs from this library's");
  const double *dmp = b.f64("mj_damping", NV), *ar = b.f64("mj_act_range", NA * 4);
  bool anyd = false;
  for (int i = 0; i < NV; i++) { m.mj_damping[i] = (T)dmp[i]; anyd |= dmp[i] != 0.0; }
  if (anyd != EULERDAMP) throw std::string("model blob: joint damping differs from what this library was built for");
  for (int i = 0; i < NA * 4; i++) {
    const double lim = sizeof(T) == 8 ? 1e300 : 1e30;
    m.act_range[i / 4][i % 4] = (T)(ar[i] > lim ? lim : (ar[i] < -lim ? -lim : ar[i]));
  }
  for (int i = 0; i < 7; i++) m.opt[i] = (T)op[i];
  m.opt[0] = (T)P[P_DT];                         // main.py:52 mj_model.opt.timestep = conf.dt
  m.opt[6] = sizeof(T) == 8 ? (T)1e-9 : (T)2e-6; // support-vertex tie tolerance
  for (int i = 0; i < 12; i++) m.contact[i] = (T)ct[i];
  const uint32_t nvert = h->blob.count("mj_hull_vert") / 3;
  m.hull_x = (const T *)h->d_hull;
  m.hull_y = m.hull_x + nvert;
  m.hull_z = m.hull_y + nvert;
  {
    const uint32_t np2 = b.count("mj_pairs");
    if (np2 % 2 || np2 / 2 > MAXPAIR) throw std::string("model blob: too many candidate body pairs");
    const int *pp = b.i32("mj_pairs", 0);
    m.npair = (int)(np2 / 2);
    for (int k = 0; k < m.npair; k++) {
      m.pair_a[k] = pp[2 * k]; m.pair_b[k] = pp[2 * k + 1];
      if (pp[2 * k] < 0 || pp[2 * k] >= NG || pp[2 * k + 1] < 0 || pp[2 * k + 1] >= NG) throw std::string("model blob: bad geom pair");
    }
    const double *hc = b.f64("mj_hull_center", NG * 3), *hb = b.f64("mj_hull_box", NG * 6);
    for (int i = 0; i < NG * 3; i++) m.hcen[i / 3][i % 3] = (T)hc[i];
    for (int i = 0; i < NG * 6; i++) m.hbox[i / 6][i % 6] = (T)hb[i];
    if (sizeof(T) == 4) for (int j = 0; j < NG; j++) for (int i = 3; i < 6; i++) m.hbox[j][i] = m.hbox[j][i] * (T)1.00001 + (T)1e-7;
  }
  memcpy(m.chunk_adr, b.i32("mj_chunk_adr", NG + 1), sizeof m.chunk_adr);
  m.chunk_box = (const T *)h->d_box;
  m.hull_eadr = h->d_eadr;
  m.hull_edge = h->d_edge;
}

// The kernels index the hull arrays with addresses taken from the blob and pack (geom << 16 | vertex) with bit 15 as
// the robot<->robot flag: check every address table against the section it indexes before anything is uploaded
// (Blob::validate only knows section bounds).
static void validate_geometry(const Blob &b) {
  if constexpr (!TOPO_HAS_SIM) return;
  const int *ha = b.i32("mj_hull_adr", NG + 1), *ca = b.i32("mj_chunk_adr", NG + 1);
  const int64_t nvert = b.count("mj_hull_vert") / 3, nchunk = b.count("mj_chunk_box") / 6, nedge = b.count("mj_hull_edge");
  if (b.count("mj_hull_vert") % 3 || b.count("mj_chunk_box") % 6) throw std::string("model blob: hull vertex / chunk box sections are not whole records");
  if (ha[0] != 0 || ha[NG] != nvert || ca[0] != 0 || ca[NG] != nchunk) throw std::string("model blob: hull / chunk address tables do not span their sections");
  for (int g = 0; g < NG; g++) {
    const int64_t nv = (int64_t)ha[g + 1] - ha[g], nc = (int64_t)ca[g + 1] - ca[g];
    if (nv < 1 || nc < 1) throw std::string("model blob: hull or chunk addresses are not increasing");
    if (nv >= 0x8000) throw std::string("model blob: a hull has 32768 or more vertices (contact ids keep the vertex in 15 bits)");
    if (nc > WAVE || nc * WAVE < nv) throw std::string("model blob: a hull's chunks do not cover it (at most 64 chunks of 64 vertices)");
  }
  if ((int64_t)b.count("mj_hull_eadr") != nvert + 1) throw std::string("model blob: hull graph address table has the wrong length");
  const int *ea = b.i32("mj_hull_eadr", 0), *ed = b.i32("mj_hull_edge", 0);
  if (ea[0] != 0 || ea[nvert] > nedge) throw std::string("model blob: hull graph addresses exceed the edge list");
  for (int g = 0; g < NG; g++)
    for (int v = ha[g]; v < ha[g + 1]; v++) {
      if (ea[v + 1] < ea[v]) throw std::string("model blob: hull graph addresses are not increasing");
      for (int e = ea[v]; e < ea[v + 1]; e++)
        if (ed[e] < 0 || ed[e] >= ha[g + 1] - ha[g]) throw std::string("model blob: hull graph edge leaves its hull");
    }
}

template <typename T>
static void upload_model(tsidb_ctx *h) {
  const Blob &b = h->blob;
  if (!h->d_model && !TOPO_HAS_SIM) HIP_OK(hipMalloc(&h->d_model, sizeof(DevModel<T>)));
  if (!h->d_hull && TOPO_HAS_SIM) {
    const uint32_t nvert3 = b.count("mj_hull_vert"), nedge = b.count("mj_hull_edge"), neadr = b.count("mj_hull_eadr");
    const double *hv = b.f64("mj_hull_vert", 0);
    std::vector<T> hvt(nvert3);
    for (uint32_t i = 0; i < nvert3; i++) hvt[(i % 3) * (nvert3 / 3) + i / 3] = (T)hv[i]; // AoS -> SoA
    HIP_OK(hipMalloc(&h->d_hull, nvert3 * sizeof(T)));
    HIP_OK(hipMemcpy(h->d_hull, hvt.data(), nvert3 * sizeof(T), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void **)&h->d_eadr, neadr * sizeof(int)));
    HIP_OK(hipMemcpy(h->d_eadr, b.i32("mj_hull_eadr", 0), neadr * sizeof(int), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc((void **)&h->d_edge, nedge * sizeof(int)));
    HIP_OK(hipMemcpy(h->d_edge, b.i32("mj_hull_edge", 0), nedge * sizeof(int), hipMemcpyHostToDevice));
    const uint32_t nbox6 = b.count("mj_chunk_box");
    const double *bx = b.f64("mj_chunk_box", 0);
    std::vector<T> bxt(nbox6);
    for (uint32_t i = 0; i < nbox6; i++) bxt[i] = (T)bx[i];
    // boxes must bound the vertices AFTER conversion to T: widen the half extents by one rounding step
    if (sizeof(T) == 4) for (uint32_t i = 0; i < nbox6; i++) if (i % 6 >= 3) bxt[i] = bxt[i] * (T)1.00001 + (T)1e-7;
    HIP_OK(hipMalloc(&h->d_box, nbox6 * sizeof(T)));
    HIP_OK(hipMemcpy(h->d_box, bxt.data(), nbox6 * sizeof(T), hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&h->d_model, sizeof(DevModel<T>)));
  }
  static thread_local DevModel<T> m;
  build_model<T>(h, m);
  HIP_OK(hipMemcpy(h->d_model, &m, sizeof m, hipMemcpyHostToDevice));
}

#define GUARD_BEGIN                      \
  if (!h) return -1;                     \
  try {                                  \
    HIP_OK(hipSetDevice(h->device));
#define GUARD_END                        \
  }                                      \
  catch (const std::string &s) {         \
    h->err = s;                          \
    return 1;                            \
  }                                      \
  return 0;

extern "C" int tsidb_set_env_params(tsidb_handle h, const void *env_params, const void *terrain) {
  if (!h) return -1;
  h->env_params = env_params; // NULL restores the nominal model
  h->terrain = terrain;       // NULL = no terrain steps
  return 0;
}

static void need_refs(tsidb_ctx *h) {
  if (!h->com_ref) throw std::string("reference buffers not registered (call tsidb_set_refs first)");
  if (h->params[P_W_COP] != 0.0 && !h->cop_ref) throw std::string("w_cop != 0 needs a CoP reference (tsidb_set_cop_ref)");
}

template <typename T>
static void launch_tick(tsidb_ctx *h, void *q, void *v, void *tau, void *dv, void *f, int32_t *status, void *obs, int obs_ld,
                        void *frames, int32_t *info, hipStream_t s, const void *qpos_sim = nullptr,
                        const void *qvel_sim = nullptr, const WalkArgs<T> *walk = nullptr, void *q_snap = nullptr, void *v_snap = nullptr) {
  if (obs && obs_ld < NOBS) throw std::string("obs row stride must be at least TSIDB_NOBS");
  h->note_stream(s);
  WalkArgs<T> wa;
  memset(&wa, 0, sizeof wa);
  if (walk) wa = *walk;
#define TSIDB_LAUNCH_TICK(COP)                                                                                             \
  hipLaunchKernelGGL((k_tick<T, COP>), dim3(h->num_envs), dim3(WAVE), h->lds_pad, s, (const DevModel<T> *)h->d_model, h->num_envs,   \
                     (T *)q, (T *)v, (const T *)h->com_ref, (const T *)h->posture_ref, (const T *)h->foot_ref,             \
                     (const T *)h->contact_ref, h->contact_active, (const T *)h->cop_frames, (T *)tau, (T *)dv, (T *)f,    \
                     status, (T *)obs, obs_ld, (T *)frames, info, (const T *)qpos_sim, (const T *)qvel_sim, (const T *)h->cop_ref, wa, \
                     (T *)q_snap, (T *)v_snap, h->qp_fast_eq)
  if (h->params[P_W_COP] != 0.0) TSIDB_LAUNCH_TICK(true);
  else TSIDB_LAUNCH_TICK(false);
#undef TSIDB_LAUNCH_TICK
  HIP_OK(hipGetLastError());
}
template <typename T>
static void launch_sim(tsidb_ctx *h, int B, const void *q_ring, const void *v_ring, const int32_t *slots, void *qpos, void *qvel, void *qacc_ws, void *qacc,
                       int32_t *ncon, int32_t *con, int32_t *info, hipStream_t s, const void *motor_tau = nullptr) {
  if constexpr (!TOPO_HAS_SIM) throw std::string("this library was built without the sim stage");
  else {
    if (B < 1 || B > TSIDB_MAX_SIM_BATCH) throw std::string("sim batch must be 1 .. TSIDB_MAX_SIM_BATCH steps");
    h->note_stream(s);
    SimRing<T> ring;
    ring.q = (const T *)q_ring; ring.v = (const T *)v_ring; ring.slots = 0;
    for (int b = 0; b < B; b++) {
      const int sl = slots ? slots[b] : 0;
      if (sl < 0 || sl > 15) throw std::string("sim batch: slot numbers must be 0 .. 15");
      ring.slots |= (unsigned long long)sl << (4 * b);
    }
    // Two wavefronts per env (collision beside the unconstrained dynamics, bit-identical) shorten the step only while every
    // wavefront has a SIMD to itself: 2 N sim wavefronts + N of the tick running beside them on 1024 SIMDs, or 2 N on the
    // 512 SIMDs of the sim stream's half when the streams are CU-split (<= 512 envs) - the library picks NW = 2 up to 512
    // envs (tsidb_create; k_sim 5-9 % shorter; at 448 / 512 envs it pays since round 4's shorter tick made the sim the longer
    // stream there: 7.8 -> 8.0 M env-steps/s).  Beyond that it loses, and the round-4 traces show how
    // (profiles/r04_trace_pipeline_*.txt, DESIGN.md section 5 "Streams"): at 1024 envs the 2048 wavefronts of a sim batch
    // take every wave slot of the GPU (2 per SIMD) for the whole batch, and the tick launched beside it "runs" 338-526 us
    // instead of 58 waiting for a slot; at 512 envs on half the CUs the two wavefronts of an env share SIMDs with their
    // neighbours', the sim becomes the slower stream (up to 95 us per step) and the tick stream stalls on the snapshot ring.
#define TSIDB_LAUNCH_SIM(NW, MULTI)                                                                                                    \
    hipLaunchKernelGGL((k_sim<T, NW, MULTI>), dim3(h->num_envs), dim3(WAVE * NW), h->lds_pad, s, (const DevModel<T> *)h->d_model, h->num_envs, B, ring, \
                       (T *)qpos, (T *)qvel, (T *)qacc_ws, (const T *)h->env_params, (const T *)h->terrain, (const T *)motor_tau,     \
                       (T *)qacc, ncon, con, info)
    if (sizeof(T) == 4 && B == 1 && h->sim_waves == 1 && !h->sim_pack && !motor_tau && h->num_envs >= 3072 && !h->lds_pad)
      hipLaunchKernelGGL((k_sim<T, 1, false, (sizeof(T) == 4 ? 3 : TSIDB_WPE)>), dim3(h->num_envs), dim3(WAVE), 0, s, (const DevModel<T> *)h->d_model, h->num_envs, B, ring,
                         (T *)qpos, (T *)qvel, (T *)qacc_ws, (const T *)h->env_params, (const T *)h->terrain, (const T *)motor_tau, (T *)qacc, ncon, con, info);
    else if (B == 1 && h->sim_pack && SIM_PACKABLE)
      hipLaunchKernelGGL((k_sim2<T>), dim3((h->num_envs + 1) / 2), dim3(WAVE), 0, s, (const DevModel<T> *)h->d_model, h->num_envs, ring, (T *)qpos, (T *)qvel,
                         (T *)qacc_ws, (const T *)h->env_params, (const T *)h->terrain, (const T *)motor_tau, (T *)qacc, ncon, con, info);
    else if (B > 1) { if (h->sim_waves == 2) TSIDB_LAUNCH_SIM(2, true); else TSIDB_LAUNCH_SIM(1, true); }
    else { if (h->sim_waves == 2) TSIDB_LAUNCH_SIM(2, false); else TSIDB_LAUNCH_SIM(1, false); }
#undef TSIDB_LAUNCH_SIM
  }
  HIP_OK(hipGetLastError());
}
template <typename T>
static void launch_sim(tsidb_ctx *h, const void *q_tsid, const void *v_tsid, void *qpos, void *qvel, void *qacc_ws, void *qacc,
                       int32_t *ncon, int32_t *con, int32_t *info, hipStream_t s, const void *motor_tau = nullptr) {
  launch_sim<T>(h, 1, q_tsid, v_tsid, nullptr, qpos, qvel, qacc_ws, qacc, ncon, con, info, s, motor_tau);
}

// tsidb_walk_args (include/tsidb.h) -> the kernel's argument block, with the registered reference buffers
template <typename T>
static WalkArgs<T> walk_args(tsidb_ctx *h, const tsidb_walk_args *a) {
  if (!a->coef || !a->side || !a->nsteps || !a->rest || !a->com || !a->frames || a->K <= 0) throw std::string("walking reference update: null table or K <= 0");
  if (!(a->step_duration > 0) || !(a->omega > 0) || a->t_start < 0) throw std::string("walking reference update: bad timing");
  if (a->td_latch && (!a->ncon || !a->con_pairs)) throw std::string("walking reference update: touch-down feedback needs ncon and con_pairs");
  WalkArgs<T> w;
  w.coef = (const T *)a->coef; w.side = a->side; w.nsteps = a->nsteps; w.rest = (const T *)a->rest; w.com = (const T *)a->com; w.K = a->K;
  w.t_now = (T)a->t; w.t_off = (const T *)a->t_offset; w.Tstep = (T)a->step_duration; w.t_start = (T)a->t_start; w.omega = (T)a->omega;
  w.z0 = (T)a->com_z0; w.dz = (T)a->com_drop; w.frames = (const T *)a->frames; w.foot_ref = (T *)h->foot_ref; w.contact_ref = (T *)h->contact_ref;
  w.cact = (uint8_t *)h->contact_active; w.com_ref = (T *)h->com_ref; w.ncon = a->ncon; w.con = a->con_pairs; w.latch = a->td_latch;
  w.fgeoms0 = h->foot_geoms[0]; w.fgeoms1 = h->foot_geoms[1]; w.td_frac = (T)a->td_fraction; w.t_dev = a->t_device;
  return w;
}

template <typename T>
static void launch_reset(tsidb_ctx *h, const int32_t *env_ids, int n_ids, void *q, void *v, void *qpos, void *qvel, void *qacc_ws,
                         const void *done_rows, int rows_ld, void *frames, hipStream_t s) {
  const int grid = env_ids ? n_ids : h->num_envs;
  if (grid <= 0) return;
  h->note_stream(s);
  hipLaunchKernelGGL(k_reset<T>, dim3(grid), dim3(WAVE), 0, s, (const DevModel<T> *)h->d_model, h->num_envs, env_ids, n_ids, (T *)q,
                     (T *)v, (T *)qpos, (T *)qvel, (T *)qacc_ws, (T *)h->com_ref, (T *)h->posture_ref, (T *)h->foot_ref,
                     (T *)h->contact_ref, (uint8_t *)h->contact_active, (T *)h->cop_frames, (T *)h->cop_ref, (const T *)done_rows,
                     rows_ld, (const T *)h->posture_bias, (T *)frames);
  HIP_OK(hipGetLastError());
}

extern "C" {

int tsidb_create(const void *model_blob, size_t nbytes, const double *params, int n_params, int num_envs, int device,
                 int dtype, tsidb_handle *out) {
  if (!out) return -1;
  *out = nullptr;
  tsidb_ctx *h = new tsidb_ctx();
  *out = h; // returned even on failure so that tsidb_last_error can be read; caller destroys it
  try {
    if (!model_blob || nbytes < 16 || memcmp(model_blob, "TSIDBM01", 8) != 0) throw std::string("not a TSIDBM01 model blob");
    if (!params || n_params != P_COUNT) throw std::string("params must hold TSIDB_P_COUNT doubles");
    if (num_envs <= 0) throw std::string("num_envs must be positive");
    if (dtype != TSIDB_F64 && dtype != TSIDB_F32) throw std::string("dtype must be TSIDB_F64 or TSIDB_F32");
    h->device = device; h->dtype = dtype; h->num_envs = num_envs;
    h->sim_waves = num_envs <= 512 ? 2 : 1; // (measured, DESIGN.md section 5 "Streams": up to the batch size the streams are CU-split for)
    h->blob.raw.assign((const uint8_t *)model_blob, (const uint8_t *)model_blob + nbytes);
    h->blob.validate();
    { // the blob must be for the robot this library was built for
      const int want[9] = {NJ, NQ, NV, NA, NB, TOPO_HAS_SIM, NG, CONDIM, TOPO_EULERDAMP};
      const int *got = h->blob.i32("model_dims", 9);
      for (int i = 0; i < 9; i++)
        if (got[i] != want[i])
          throw std::string("model blob is for another robot than this library (dimensions differ: build the library with "
                            "the blob's topology header, -DTSIDB_TOPOLOGY_HEADER)");
    }
    h->params.assign(params, params + n_params);
    if (!TOPO_HAS_SIM && h->params[P_SIM_ENABLED] != 0.0) throw std::string("this library was built without the sim stage: set sim_enabled = False");
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) throw std::string("no such HIP device (this library has no CPU path)");
    HIP_OK(hipSetDevice(device));
    validate_geometry(h->blob);
    if (dtype == TSIDB_F64) upload_model<double>(h); else upload_model<float>(h);
    if (TOPO_HAS_SIM) { // sim body of each sole frame: frame -> TSID joint -> sim joint (mj_sim2tsid) -> body
      const int *fp = h->blob.i32("pin_frame_parent", 2), *s2t = h->blob.i32("mj_sim2tsid", NA);
      for (int f = 0; f < 2; f++)
        for (int i = 0; i < NA; i++)
          if (s2t[i] == fp[f] - 1) h->foot_body[f] = 1 + i;
      if (h->foot_body[0] < 0 || h->foot_body[1] < 0) throw std::string("model blob: sole frames are not on sim bodies");
      static_assert(NG <= 64, "foot geoms are kept as a 64-bit mask");
      const int *gb = h->blob.i32("mj_geom_body", NG);
      for (int f = 0; f < 2; f++)
        for (int g = 0; g < NG; g++)
          if (gb[g] == h->foot_body[f]) h->foot_geoms[f] |= 1ull << g;
    }
  } catch (const std::string &s) {
    h->err = s;
    return 1;
  }
  return 0;
}

int tsidb_destroy(tsidb_handle h) {
  if (!h) return -1;
  if (h->d_model) {
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_model); (void)hipFree(h->d_hull); (void)hipFree(h->d_box); (void)hipFree(h->d_eadr); (void)hipFree(h->d_edge);
  }
  delete h;
  return 0;
}

const char *tsidb_last_error(tsidb_handle h) { return h ? h->err.c_str() : "null handle"; }

int tsidb_set_params(tsidb_handle h, const double *params, int n_params) {
  GUARD_BEGIN
  if (!params || n_params != P_COUNT) throw std::string("params must hold TSIDB_P_COUNT doubles");
  h->params.assign(params, params + n_params);
  // kernels in flight (the pipelined sim stage runs on a side stream) read the model constants: wait for the streams THIS
  // handle has launched on before the constants are replaced - not for the whole device (other handles, the caller's own
  // work and collectives keep running).  A stream the caller has destroyed since has nothing in flight: its error is dropped.
  for (hipStream_t st : h->used_streams)
    if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
  if (h->dtype == TSIDB_F64) upload_model<double>(h); else upload_model<float>(h);
  GUARD_END
}

int tsidb_set_refs(tsidb_handle h, const void *com_ref, const void *posture_ref, const void *foot_ref,
                   const void *contact_ref, const uint8_t *contact_active, const void *cop_frames) {
  if (!h) return -1;
  if (!com_ref || !posture_ref || !foot_ref || !contact_ref || !contact_active || !cop_frames) {
    h->err = "tsidb_set_refs: null reference buffer";
    return 1;
  }
  h->com_ref = com_ref; h->posture_ref = posture_ref; h->foot_ref = foot_ref;
  h->contact_ref = contact_ref; h->contact_active = contact_active; h->cop_frames = cop_frames;
  return 0;
}

int tsidb_set_option(tsidb_handle h, int option, int value) {
  if (!h) return -1;
  if (option == TSIDB_OPT_SIM_WAVES && (value == 1 || value == 2)) { h->sim_waves = value; return 0; }
  if (option == TSIDB_OPT_LDS_PAD && value >= 0 && value <= 40960) { h->lds_pad = (unsigned)value; return 0; }
  if (option == TSIDB_OPT_SIM_PACK && (value == 0 || (value == 1 && SIM_PACKABLE))) { h->sim_pack = value; return 0; }
  if (option == TSIDB_OPT_QP_FAST_EQ && (value == 0 || value == 1)) { h->qp_fast_eq = value; return 0; }
  if (option == TSIDB_OPT_CU_SPLIT && value >= -1 && value <= 1) { h->cu_split = value; return 0; }
  h->err = "tsidb_set_option: unknown option or value";
  return 1;
}

int tsidb_get_option(tsidb_handle h, int option, int *value) {
  if (!h) return -1;
  if (!value) { h->err = "tsidb_get_option: null value"; return 1; }
  if (option == TSIDB_OPT_SIM_WAVES) { *value = h->sim_waves; return 0; }
  if (option == TSIDB_OPT_LDS_PAD) { *value = (int)h->lds_pad; return 0; }
  if (option == TSIDB_OPT_SIM_PACK) { *value = h->sim_pack; return 0; }
  if (option == TSIDB_OPT_QP_FAST_EQ) { *value = h->qp_fast_eq; return 0; }
  if (option == TSIDB_OPT_CU_SPLIT) { *value = (h->cu_split == 1 || (h->cu_split < 0 && h->num_envs <= 512)) ? 1 : 0; return 0; }
  h->err = "tsidb_get_option: unknown option";
  return 1;
}

int tsidb_stream_create(tsidb_handle h, int role, void **stream) {
  GUARD_BEGIN
  if (!stream || (role != TSIDB_STREAM_TICK && role != TSIDB_STREAM_SIM)) throw std::string("tsidb_stream_create: role must be TSIDB_STREAM_TICK or TSIDB_STREAM_SIM");
  hipStream_t s = nullptr;
  const bool split = h->cu_split == 1 || (h->cu_split < 0 && h->num_envs <= 512);
  if (split) {
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, h->device));
    const int ncu = prop.multiProcessorCount, half = ncu / 2;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int c = role == TSIDB_STREAM_TICK ? 0 : half; c < (role == TSIDB_STREAM_TICK ? half : ncu); c++) mask[(size_t)c / 32] |= 1u << (c % 32);
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
      (void)hipGetLastError(); // CU masking not available here: an ordinary stream, and no further attempts for this handle
      s = nullptr;
      h->cu_split = 0;
    }
  }
  if (!s) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  GUARD_END
}

int tsidb_stream_destroy(tsidb_handle h, void *stream) {
  GUARD_BEGIN
  if (stream) {
    for (size_t i = 0; i < h->used_streams.size(); i++)
      if (h->used_streams[i] == (hipStream_t)stream) { h->used_streams.erase(h->used_streams.begin() + i); break; }
    HIP_OK(hipStreamDestroy((hipStream_t)stream)); // (hipStreamDestroy lets the stream's pending work finish)
  }
  GUARD_END
}

int tsidb_set_cop_ref(tsidb_handle h, const void *cop_ref) {
  if (!h) return -1;
  h->cop_ref = cop_ref;
  return 0;
}

int tsidb_reset(tsidb_handle h, const int32_t *env_ids, int n_ids, void *q, void *v, void *qpos, void *qvel,
                void *qacc_ws, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!q || !v || !qpos || !qvel || !qacc_ws) throw std::string("tsidb_reset: null state buffer");
  if (h->dtype == TSIDB_F64) launch_reset<double>(h, env_ids, n_ids, q, v, qpos, qvel, qacc_ws, nullptr, 0, nullptr, (hipStream_t)stream);
  else launch_reset<float>(h, env_ids, n_ids, q, v, qpos, qvel, qacc_ws, nullptr, 0, nullptr, (hipStream_t)stream);
  GUARD_END
}

int tsidb_reset_done(tsidb_handle h, const void *rows, int rows_ld, void *q, void *v, void *qpos, void *qvel, void *qacc_ws,
                     void *frames, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!rows || rows_ld < NROW) throw std::string("tsidb_reset_done: needs the [N, >= TSIDB_NROW] rows tsidb_tick writes (done flag in column TSIDB_NOBS + 1)");
  if (!q || !v || !qpos || !qvel || !qacc_ws) throw std::string("tsidb_reset_done: null state buffer");
  if (h->dtype == TSIDB_F64) launch_reset<double>(h, nullptr, 0, q, v, qpos, qvel, qacc_ws, rows, rows_ld, frames, (hipStream_t)stream);
  else launch_reset<float>(h, nullptr, 0, q, v, qpos, qvel, qacc_ws, rows, rows_ld, frames, (hipStream_t)stream);
  GUARD_END
}

int tsidb_set_posture_bias(tsidb_handle h, const void *posture_bias) {
  if (!h) return -1;
  h->posture_bias = posture_bias;
  return 0;
}

int tsidb_walk_plan(tsidb_handle h, const int32_t *env_ids, int n_ids, const void *done_rows, int rows_ld,
                    const double *plan_params, int n_plan_params, const double *path, const int32_t *npts, int P,
                    const double *scale, int32_t *episode, int bump_episode, int K, double *steps, void *coef, int32_t *side,
                    int32_t *nsteps, void *rest, void *com, int32_t *flags, void *t_offset, int32_t *td_latch, double t,
                    const double *t_device, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!plan_params || n_plan_params != TSIDB_PLAN_NPARAMS) throw std::string("tsidb_walk_plan: plan_params must hold TSIDB_PLAN_NPARAMS doubles");
  if (!steps || !coef || !side || !nsteps || !rest || !com || K <= 0) throw std::string("tsidb_walk_plan: null table or K <= 0");
  if (path && (!npts || P < 2)) throw std::string("tsidb_walk_plan: an explicit path needs npts and P >= 2");
  if (!path && !(plan_params[12] >= 2)) throw std::string("tsidb_walk_plan: the unicycle path needs at least two vertices");
  if (!(plan_params[0] > 0) || !(plan_params[3] > 0) || !(plan_params[5] > 0) || !(plan_params[4] > 0 && plan_params[4] < 1))
    throw std::string("tsidb_walk_plan: step_length, step_duration, t_start must be positive and rise_ratio inside (0, 1)");
  if (done_rows && rows_ld < NROW) throw std::string("tsidb_walk_plan: the done mask needs rows of at least TSIDB_NROW values");
  // one thread per env walks its path vertex by vertex: every loop bound that comes from a parameter is checked here
  for (int i = 0; i < TSIDB_PLAN_NPARAMS; i++)
    if (!std::isfinite(plan_params[i])) throw std::string("tsidb_walk_plan: non-finite plan parameter");
  if (!(plan_params[1] > 0) || !(plan_params[2] >= 0) || !(plan_params[6] >= 0))
    throw std::string("tsidb_walk_plan: step_width must be positive, step_height and com_drop non-negative");
  if (plan_params[8] != 0 && !(plan_params[8] >= plan_params[0] / 1000))
    throw std::string("tsidb_walk_plan: resample_ds must be 0 (vertices as given) or at least step_length / 1000");
  if (!path) {
    if (!(plan_params[12] <= 65536)) throw std::string("tsidb_walk_plan: the unicycle path may have at most 65536 vertices");
    if (!scale && !(plan_params[13] > 0 && plan_params[14] >= plan_params[13])) throw std::string("tsidb_walk_plan: scale range must satisfy 0 < lo <= hi");
    if (!(plan_params[11] > 0)) throw std::string("tsidb_walk_plan: the unicycle path's time step must be positive");
  } else if (P > 65536) throw std::string("tsidb_walk_plan: an explicit path may have at most 65536 vertices per env");
  const int cnt = env_ids ? n_ids : h->num_envs;
  if (cnt <= 0) return 0;
  PlanParams PP;
  for (int i = 0; i < 16; i++) PP.v[i] = plan_params[i];
  hipStream_t s = (hipStream_t)stream;
  const int grid = (cnt + 63) / 64;
  if (h->dtype == TSIDB_F64)
    hipLaunchKernelGGL(k_plan<double>, dim3(grid), dim3(64), 0, s, h->num_envs, env_ids, n_ids, (const double *)done_rows, rows_ld, PP,
                       (const double *)h->cop_frames, (const double *)h->com_ref, path, npts, P, scale, episode, bump_episode, K, steps,
                       (double *)coef, side, nsteps, (double *)rest, (double *)com, flags, (double *)t_offset, td_latch, t, t_device);
  else
    hipLaunchKernelGGL(k_plan<float>, dim3(grid), dim3(64), 0, s, h->num_envs, env_ids, n_ids, (const float *)done_rows, rows_ld, PP,
                       (const float *)h->cop_frames, (const float *)h->com_ref, path, npts, P, scale, episode, bump_episode, K, steps,
                       (float *)coef, side, nsteps, (float *)rest, (float *)com, flags, (float *)t_offset, td_latch, t, t_device);
  HIP_OK(hipGetLastError());
  GUARD_END
}

int tsidb_tick(tsidb_handle h, void *q, void *v, void *tau, void *dv, void *f, int32_t *status, void *obs, int obs_ld,
               void *frames, int32_t *info, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!q || !v || !tau || !dv || !f || !status) throw std::string("tsidb_tick: null buffer");
  if (h->dtype == TSIDB_F64) launch_tick<double>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, (hipStream_t)stream);
  else launch_tick<float>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, (hipStream_t)stream);
  GUARD_END
}

int tsidb_sim(tsidb_handle h, const void *q_tsid, const void *v_tsid, void *qpos, void *qvel, void *qacc_ws, void *qacc,
              int32_t *ncon, int32_t *con_pairs, int32_t *info, void *stream) {
  GUARD_BEGIN
  if (!qpos || !qvel || !qacc_ws) throw std::string("tsidb_sim: null state buffer");
  if (h->dtype == TSIDB_F64) launch_sim<double>(h, q_tsid, v_tsid, qpos, qvel, qacc_ws, qacc, ncon, con_pairs, info, (hipStream_t)stream);
  else launch_sim<float>(h, q_tsid, v_tsid, qpos, qvel, qacc_ws, qacc, ncon, con_pairs, info, (hipStream_t)stream);
  GUARD_END
}

int tsidb_sim_batch(tsidb_handle h, int n_steps, const void *q_ring, const void *v_ring, const int32_t *slots, void *qpos, void *qvel,
                    void *qacc_ws, void *qacc, int32_t *ncon, int32_t *con_pairs, int32_t *info, void *stream) {
  GUARD_BEGIN
  if (!qpos || !qvel || !qacc_ws || !q_ring || !slots) throw std::string("tsidb_sim_batch: null buffer");
  if (h->dtype == TSIDB_F64) launch_sim<double>(h, n_steps, q_ring, v_ring, slots, qpos, qvel, qacc_ws, qacc, ncon, con_pairs, info, (hipStream_t)stream);
  else launch_sim<float>(h, n_steps, q_ring, v_ring, slots, qpos, qvel, qacc_ws, qacc, ncon, con_pairs, info, (hipStream_t)stream);
  GUARD_END
}

int tsidb_step(tsidb_handle h, void *q, void *v, void *qpos, void *qvel, void *qacc_ws, void *tau, void *dv, void *f,
               int32_t *status, void *obs, int obs_ld, void *frames, int32_t *ncon, int32_t *con_pairs, int32_t *info,
               int n_substeps, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!q || !v || !tau || !dv || !f || !status) throw std::string("tsidb_step: null buffer");
  const bool sim = h->params[P_SIM_ENABLED] != 0.0, closed = h->params[P_CLOSED_LOOP] != 0.0;
  if (sim && (!qpos || !qvel || !qacc_ws)) throw std::string("tsidb_step: null sim state buffer");
  hipStream_t s = (hipStream_t)stream;
  for (int it = 0; it < n_substeps; it++) {
    // closed loop: the tick reads the sim state, the sim is driven by tau and keeps its own base pose
    if (h->dtype == TSIDB_F64) {
      launch_tick<double>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, s, closed ? qpos : nullptr, closed ? qvel : nullptr);
      if (sim) launch_sim<double>(h, closed ? nullptr : q, closed ? nullptr : v, qpos, qvel, qacc_ws, nullptr, ncon, con_pairs, info, s, closed ? tau : nullptr);
    } else {
      launch_tick<float>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, s, closed ? qpos : nullptr, closed ? qvel : nullptr);
      if (sim) launch_sim<float>(h, closed ? nullptr : q, closed ? nullptr : v, qpos, qvel, qacc_ws, nullptr, ncon, con_pairs, info, s, closed ? tau : nullptr);
    }
  }
  GUARD_END
}

int tsidb_walk_update(tsidb_handle h, const void *coef, const int32_t *side, const int32_t *nsteps, const void *rest,
                      const void *com, int K, double t, double step_duration, double t_start, double omega, double com_z0,
                      double com_drop, const void *frames, const void *t_offset, const int32_t *ncon, const int32_t *con_pairs,
                      int32_t *td_latch, double td_fraction, const void *t_device, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  const tsidb_walk_args a = {coef, side, nsteps, rest, com, K, t, step_duration, t_start, omega, com_z0, com_drop, frames, t_offset,
                             ncon, con_pairs, td_latch, td_fraction, (const double *)t_device};
  hipStream_t s = (hipStream_t)stream;
  const int grid = (h->num_envs * 16 + 255) / 256;
  if (h->dtype == TSIDB_F64) hipLaunchKernelGGL(k_walk<double>, dim3(grid), dim3(256), 0, s, h->num_envs, walk_args<double>(h, &a));
  else hipLaunchKernelGGL(k_walk<float>, dim3(grid), dim3(256), 0, s, h->num_envs, walk_args<float>(h, &a));
  HIP_OK(hipGetLastError());
  GUARD_END
}

int tsidb_tick_walk(tsidb_handle h, const tsidb_walk_args *walk, void *q, void *v, void *tau, void *dv, void *f, int32_t *status,
                    void *obs, int obs_ld, void *frames, int32_t *info, void *q_snapshot, void *v_snapshot, void *stream) {
  GUARD_BEGIN
  need_refs(h);
  if (!q || !v || !tau || !dv || !f || !status) throw std::string("tsidb_tick_walk: null buffer");
  if (h->dtype == TSIDB_F64) {
    WalkArgs<double> wa;
    if (walk) wa = walk_args<double>(h, walk);
    launch_tick<double>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, (hipStream_t)stream, nullptr, nullptr, walk ? &wa : nullptr,
                        q_snapshot, v_snapshot);
  } else {
    WalkArgs<float> wa;
    if (walk) wa = walk_args<float>(h, walk);
    launch_tick<float>(h, q, v, tau, dv, f, status, obs, obs_ld, frames, info, (hipStream_t)stream, nullptr, nullptr, walk ? &wa : nullptr,
                       q_snapshot, v_snapshot);
  }
  GUARD_END
}

int tsidb_rbd_terms(tsidb_handle h, const void *q, const void *v, void *M, void *hbias, void *Jcom, void *Jf, void *oMf,
                    void *com, void *stream) {
  GUARD_BEGIN
  if (!q || !v || !M || !hbias || !Jcom || !Jf || !oMf || !com) throw std::string("tsidb_rbd_terms: null buffer");
  hipStream_t s = (hipStream_t)stream;
  h->note_stream(s);
  if (h->dtype == TSIDB_F64)
    hipLaunchKernelGGL(k_rbd<double>, dim3(h->num_envs), dim3(WAVE), 0, s, (const DevModel<double> *)h->d_model, h->num_envs,
                       (const double *)q, (const double *)v, (double *)M, (double *)hbias, (double *)Jcom, (double *)Jf,
                       (double *)oMf, (double *)com);
  else
    hipLaunchKernelGGL(k_rbd<float>, dim3(h->num_envs), dim3(WAVE), 0, s, (const DevModel<float> *)h->d_model, h->num_envs,
                       (const float *)q, (const float *)v, (float *)M, (float *)hbias, (float *)Jcom, (float *)Jf,
                       (float *)oMf, (float *)com);
  HIP_OK(hipGetLastError());
  GUARD_END
}

#ifdef TSIDB_STAMPS
int tsidb_debug_stamps(unsigned long long *out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 32 * (size_t)n);
}
#endif

/* dimensions of the robot this library was built for, in the order of the blob's model_dims section: NJ, NQ, NV, NA,
 * NB (sim bodies), 1 if the sim stage is built, NG (collision geoms), contact dimension, 1 if joints are damped */
int tsidb_dims(int *out9) {
  if (!out9) return -1;
  const int d[9] = {NJ, NQ, NV, NA, NB, TOPO_HAS_SIM, NG, CONDIM, TOPO_EULERDAMP};
  for (int i = 0; i < 9; i++) out9[i] = d[i];
  return 0;
}

int tsidb_lds_bytes(int dtype, int which) {
  if (dtype == TSIDB_F64) return which == 0 ? (int)sizeof(TickLds<double>) : (int)sizeof(SimLds<double>);
  return which == 0 ? (int)sizeof(TickLds<float>) : (int)sizeof(SimLds<float>);
}

} // extern "C"
