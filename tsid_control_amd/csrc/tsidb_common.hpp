// tsidb_common.hpp - shared device-side definitions for the batched TSID + contact-dynamics path.
// gfx950 only: 64-lane wavefronts, one wavefront (= one workgroup) per env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef TSIDB_STAMPS
// diagnostic build only (tools/stamp_profile.py): shader-clock stamps per phase, never in the product .so
__device__ unsigned long long g_stamp[8192][32];
#define TSIDB_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamp[blockIdx.x][k] = __builtin_amdgcn_s_memtime(); } while (0)
// accumulate the time since the previous TSIDB_LAP into slot k (for phases inside loops)
#define TSIDB_LAP_INIT() unsigned long long lap_t_ = __builtin_amdgcn_s_memtime()
#define TSIDB_LAP(k) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamp[blockIdx.x][k] += n_ - lap_t_; lap_t_ = n_; } while (0)
#define TSIDB_LAP_ZERO(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamp[blockIdx.x][k] = 0; } while (0)
#else
#define TSIDB_STAMP(k) do { } while (0)
#define TSIDB_LAP_INIT() do { } while (0)
#define TSIDB_LAP(k) do { } while (0)
#define TSIDB_LAP_ZERO(k) do { } while (0)
#endif

// One library is built per robot: the generated header (model_compiler.py) carries its dimensions and sim tree.
#ifndef TSIDB_TOPOLOGY_HEADER
#define TSIDB_TOPOLOGY_HEADER "tsidb_topology.hpp"
#endif
#include TSIDB_TOPOLOGY_HEADER

namespace tsidb {

constexpr int NJ = TOPO_NJ;   // TSID joints incl. the free-flyer root (pinocchio order); 21 for the v1 robot
constexpr int NQ = TOPO_NQ;   // 27
constexpr int NV = TOPO_NV;   // 26
constexpr int NA = TOPO_NA;   // 20
constexpr int NB = TOPO_NB;   // sim bodies (MuJoCo document order, world excluded); 21
constexpr int NVAR = NV + 24; // [dv(NV); f_slot0(12); f_slot1(12)]; 50
constexpr int LDD = NVAR + 1; // leading dimension of the dynamics rows [M | -Jc^T]
constexpr int LDF = NV + 1;   // leading dimension of frame / CoM Jacobians
constexpr int NOBS = NQ + NV + 12; // q, v, com, cop, LF, RF; 65
constexpr int NROW = NOBS + 2;     // obs row + reward + done when the caller's row stride has room for them
static_assert(NJ == NV - 5 && NQ == NV + 1 && NA == NV - 6, "one free-flyer + hinges");
static_assert(NJ <= 32 && NB <= 32 && NVAR <= 64 && NA <= 20 && 32 + NA <= 64, "lane / bitmask / parameter-vector limits");
constexpr int MAXCON = 32;
constexpr int MAXHH = 12;    // robot<->robot contacts per env (the last slots of the contact list)
constexpr int MAXPAIR = TOPO_MAXPAIR; // candidate geom pairs (rounds of one lane per pair); 192 for the v1 robot
constexpr int NG = TOPO_NG;           // collision geoms = convex hulls (one per body in the v1 robot)
constexpr int CONDIM = TOPO_CONDIM;   // contact dimension: 3 (robot/v1) or 4 (+ torsional friction, robot/v0/robot.xml:4)
constexpr int NROWC = 2 * (CONDIM - 1); // pyramidal rows per contact
constexpr bool EULERDAMP = TOPO_EULERDAMP != 0; // joint damping, integrated implicitly as MuJoCo's Euler does
static_assert(NG <= 64 && NG >= NB && (CONDIM == 3 || CONDIM == 4) && MAXPAIR % 64 == 0, "sim stage limits");
constexpr int MAXCHILD = 6;
constexpr int WAVE = 64;

// parameter vector indices: mirror of include/tsidb.h TSIDB_P_*
enum {
  P_DT = 0, P_MU, P_FMIN, P_FMAX, P_W_FORCEREF, P_KP_CONTACT, P_KD_CONTACT, P_W_FOOT, P_KP_FOOT, P_KD_FOOT,
  P_W_COM, P_KP_COM, P_KD_COM, P_W_POSTURE, P_HESS_REG, P_QUIRKS, P_NORMAL, P_CPOINTS = P_NORMAL + 3,
  P_KP_POSTURE = P_CPOINTS + 12, P_KD_POSTURE = P_KP_POSTURE + 20, P_TAU_MAX = P_KD_POSTURE + 20,
  P_V_MAX = P_TAU_MAX + 20, P_MAX_ITER = P_V_MAX + 20, P_SIM_ENABLED, P_CLOSED_LOOP, P_W_AM, P_KP_AM /*3*/,
  P_REW_SIGMA = P_KP_AM + 3, P_REW_CTAU, P_DONE_HEIGHT, P_DONE_TILT, P_SELF_COLLISION, P_W_COP, P_SIM_FLOSS_SCALE, P_TSID_ARMATURE, P_FRICTION_COMP, P_PLANE_MESH, P_COUNT = 128
};

// Model constants in the arithmetic type of the path; one copy in HBM, read by every workgroup
// (scalar/broadcast loads, L2 resident).
template <typename T>
struct DevModel {
  // ---- TSID side
  int pin_parent[NJ], pin_depth[NJ], pin_nchild[NJ], pin_child[NJ][MAXCHILD];
  unsigned pin_anc[NJ]; // bit a set <=> joint a is an ancestor of (or is) joint j
  int pin_last[NJ];     // last index of joint j's subtree (depth-first pre-order numbering)
  int pin_up[3][NJ];    // ancestor 1, 2, 4 levels up (-1 beyond the root)
  int pin_chain[NJ][8]; // the joint itself and its non-root ancestors, deepest first, -1 padded (the root is on every chain):
                        // static, fully unrolled walks over a dof's root path instead of bit-mask loops with dependent loads
  int pin_maxdepth;
  T pin_place[NJ][12];   // R row-major, p
  T pin_inertia[NJ][10]; // m, c(3), Ixx Ixy Ixz Iyy Iyz Izz about c
  int frame_parent[2];
  T frame_place[2][12];
  T q0[NQ];
  T mass;
  // ---- derived from the parameter vector at create / set_params time
  T params[P_COUNT];
  T Tgen[6][12];   // contact force generator
  T Bcone[17][12]; // friction pyramid rows + normal-force row
  T cone_lb[17], cone_ub[17];
  T Jf0[12][12];   // L_f^-T of the (constant) force-regularisation Hessian block
  T Hf_trace, Jf0_trace;
  T cop_t[2][3];   // tangents of the contact normal (CoP task rows)
  // ---- sim side
  int mj_parent[NB], mj_depth[NB], mj_nchild[NB], mj_child[NB][MAXCHILD];
  unsigned mj_anc[NB];
  int mj_last[NB];
  int mj_up[3][NB];
  int mj_chain[NB][8];
  int mj_maxdepth;
  T mj_pos[NB][3], mj_R[NB][9]; // body frame in parent (rotation from body_quat)
  T mj_inertia[NB][10];
  T mj_armature[NV], mj_frictionloss[NV], mj_dof_invw0[NV], mj_body_invw0[NB][2];
  int mj_act_dof[NA];
  T mj_act_kp[NA], mj_act_kv[NA];
  int mj_ctrl_qidx[NA];
  int tsid2sim[NA]; // inverse map: TSID actuated joint -> sim joint/actuator index
  int geom_body[NG];   // body carrying each collision geom
  int hull_adr[NG + 1];
  T rbound[NG][4];
  T mj_damping[NV];    // joint damping (passive force -b v; robot/v0/robot.xml:3)
  T act_range[NA][4];  // control range lo / hi, force range lo / hi of each position actuator (+-1e300 = none)
  T opt[7];
  T contact[12];       // sliding friction, solref (2), solimp (5), condim, torsional friction, margin, spare
  T meaninertia;
  const T *hull_x, *hull_y, *hull_z; // hull vertices, body frame, struct-of-arrays [nvert] each (device)
  int chunk_adr[NG + 1];             // 64-vertex spatial chunks per hull (k-d order) ...
  const T *chunk_box;                // ... with boxes [nchunk][6] = centre xyz, half extent xyz (device)
  const int *hull_eadr, *hull_edge;
  // robot<->robot collision: candidate body pairs (after excludes and the parent-child filter), the hulls'
  // centres of mass and body-frame bounding boxes (centre, half extents)
  int npair, pair_a[MAXPAIR], pair_b[MAXPAIR]; // geom pairs
  T hcen[NG][3], hbox[NG][6];
};

// ------------------------------------------------------------------ small vector helpers
template <typename T> __device__ __forceinline__ void cross3(const T *a, const T *b, T *c) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
template <typename T> __device__ __forceinline__ T dot3(const T *a, const T *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> __device__ __forceinline__ void mat3vec(const T *R, const T *v, T *o) {
  T x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  T y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  T z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
template <typename T> __device__ __forceinline__ void mat3Tvec(const T *R, const T *v, T *o) {
  T x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
  T y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
  T z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
template <typename T> __device__ __forceinline__ void mat3mul(const T *A, const T *B, T *C) {
  T t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) C[i] = t[i];
}

// 1/sqrt(x) and 1/x for positive normal x: hardware seed + Newton steps (full precision to ~1 ulp)
// instead of the IEEE sqrt + divide expansions (~35 instructions each in float64).
__device__ __forceinline__ float rsqrt_t(float x) { return 1.0f / sqrtf(x); }
__device__ __forceinline__ double rsqrt_t(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  y = y * (1.5 - hx * y * y);
  y = y * (1.5 - hx * y * y);
  return y;
}
__device__ __forceinline__ float rcp_t(float x) { return 1.0f / x; }
__device__ __forceinline__ double rcp_t(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  y = y * (2.0 - x * y);
  return y;
}

// sin and cos of a joint angle together.  Joint angles are bounded (|x| < 1e3 rad is plenty), so the
// argument reduction is two-constant Cody-Waite by pi/2 and the kernels are the classic degree-13 / 14
// minimax polynomials on [-pi/4, pi/4] (< 1 ulp); the library sincos pays for a 1e300-proof reduction.
__device__ __forceinline__ void sincos_t(float x, float &s, float &c) { s = sinf(x); c = cosf(x); }
__device__ __forceinline__ void sincos_t(double x, double &s, double &c) {
  const double k = __builtin_rint(x * 0.63661977236758134308);
  double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
  const double z = r * r;
  // sin kernel: r + r z (S1 + z (S2 + ...))
  double ps = 1.58969099521155010221e-10;
  ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
  ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
  ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
  ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
  ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(r * z, ps, r);
  // cos kernel: 1 - z/2 + z^2 (C1 + z (C2 + ...))
  double pc = -1.13596475577881948265e-11;
  pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
  pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
  pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
  pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
  pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double cr = w + (((1.0 - w) - hz) + z * z * pc);
  const int q = (int)k & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// Workgroup -> env.  The hardware deals consecutive workgroups round-robin to the 8 XCDs (one L2 each); with env =
// blockIdx, neighbouring envs - whose 160..536-byte rows share 64 / 128-byte lines at their ends - are written from
// different L2s, and each writes its part of the shared line back on its own (k_tick WRITE_SIZE 2.4 x the bytes stored).
// Give XCD x the contiguous env range [start_x, start_x + count_x) instead: lines are shared only at the 7 range borders.
__device__ __forceinline__ int env_of_block(int b, int n) {
  constexpr int NXCD = 8;
  const int x = b % NXCD, i = b / NXCD, q = n / NXCD, r = n % NXCD;
  return x * q + (x < r ? x : r) + i;
}

// a sim state with sum |qpos| + sum |qvel| beyond this (m, rad, m/s, rad/s) has diverged: the step is skipped and flagged
constexpr double SIM_STATE_BOUND = 1e6;
template <typename T> struct Eps;
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; static constexpr double inf = 1e300; };
template <> struct Eps<float> { static constexpr float v = 1.1920929e-07f; static constexpr float inf = 1e30f; };

// wavefront reductions over 64 lanes on the VALU (DPP row ops + row broadcasts; no LDS crossbar):
// quad xor-1, quad xor-2, row_half_mirror, row_mirror give every lane its 16-lane row total; row_bcast15
// / row_bcast31 fold the four rows into lane 63, which v_readlane broadcasts.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float lane63(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ double lane63(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
  v += dpp_mov<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xf>(v); // row_half_mirror
  v += dpp_mov<0x140, 0xf>(v); // row_mirror
  v += dpp_mov<0x142, 0xa>(v); // row_bcast15 into rows 1, 3
  v += dpp_mov<0x143, 0xc>(v); // row_bcast31 into rows 2, 3
  return lane63(v);
}
// three sums at once: the DPP steps of the three values are interleaved, so that a step's v_mov_dpp pair does not wait
// out the VALU-write -> DPP-read hazard on the sum the previous step just produced (one reduction alone: s_nop 1 per step)
template <typename T> __device__ __forceinline__ void wave_sum3(T &a, T &b, T &c) {
#define TSIDB_SUM3_STEP(CTRL, MASK)                                                                  \
  {                                                                                                  \
    const T ta = dpp_mov<CTRL, MASK>(a), tb = dpp_mov<CTRL, MASK>(b), tc = dpp_mov<CTRL, MASK>(c);   \
    a += ta; b += tb; c += tc;                                                                       \
  }
  TSIDB_SUM3_STEP(0xB1, 0xf) TSIDB_SUM3_STEP(0x4E, 0xf) TSIDB_SUM3_STEP(0x141, 0xf) TSIDB_SUM3_STEP(0x140, 0xf)
  TSIDB_SUM3_STEP(0x142, 0xa) TSIDB_SUM3_STEP(0x143, 0xc)
#undef TSIDB_SUM3_STEP
  a = lane63(a); b = lane63(b); c = lane63(c);
}
// Subtree sums of a tree numbered depth-first pre-order on lanes 0..31 (lane j = node j, `last` = last index
// of j's subtree, v = 0 on unused lanes): an inclusive prefix scan over the lanes (row_shr 1, 2, 4, 8 within
// the 16-lane rows, row_bcast15 into row 1), then S_j = (P[last_j] - P[j]) + v_j.  Replaces a depth loop of
// LDS round trips; a leaf returns its own value exactly.
template <typename T> __device__ __forceinline__ T bperm(T v, int src_lane) {
  if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned int)lo);
  }
}
template <typename T> __device__ __forceinline__ T subtree_sum32(T v, int last) {
  T p = v;
  p += dpp_mov<0x111, 0xf>(p); // row_shr:1 (zero fill)
  p += dpp_mov<0x112, 0xf>(p); // row_shr:2
  p += dpp_mov<0x114, 0xf>(p); // row_shr:4
  p += dpp_mov<0x118, 0xf>(p); // row_shr:8
  p += dpp_mov<0x142, 0xa>(p); // row_bcast15 into row 1 (and 3, unused)
  return (bperm(p, last) - p) + v;
}

__device__ __forceinline__ int wave_sum_int(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);
  return __builtin_amdgcn_readlane(v, 63);
}

template <int CTRL, int ROW_MASK, typename T> __device__ __forceinline__ T dpp_min_step(T v) {
  // lanes a disabled row would leave at 0 must not win the min: feed the lane's own value instead
  T w;
  if constexpr (sizeof(T) == 4) {
    w = __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
  } else {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)(b & 0xffffffffLL), (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    w = __builtin_bit_cast(T, ((long long)hi << 32) | (unsigned int)lo);
  }
  return w < v ? w : v;
}
template <typename T> __device__ __forceinline__ T wave_min(T v) {
  v = dpp_min_step<0xB1, 0xf>(v);
  v = dpp_min_step<0x4E, 0xf>(v);
  v = dpp_min_step<0x141, 0xf>(v);
  v = dpp_min_step<0x140, 0xf>(v);
  v = dpp_min_step<0x142, 0xa>(v);
  v = dpp_min_step<0x143, 0xc>(v);
  return lane63(v);
}
// (value, index) lexicographic minimum: smallest value, lowest index among equals (two DPP reductions)
template <typename T> __device__ __forceinline__ void wave_argmin(T &v, int &i) {
  const T vmin = wave_min(v);
  i = wave_min<int>(v == vmin ? i : 0x7fffffff);
  v = vmin;
}
__device__ __forceinline__ int wave_min_int(int v) { return wave_min<int>(v); }

// quaternion (x,y,z,w order given as separate scalars) -> rotation, normalising
template <typename T> __device__ __forceinline__ void quat_to_R(T x, T y, T z, T w, T *R) {
  T n = T(1) / sqrt(x * x + y * y + z * z + w * w);
  x *= n; y *= n; z *= n; w *= n;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// composite inertia about the reference origin O, world axes: (m, h = m c, I_O sym6)
// Y * [lin; ang] -> [force; moment]
template <typename T> __device__ __forceinline__ void yo_mul(const T *Y, const T *X, T *f) {
  T wxh[3], hxl[3];
  cross3(X + 3, Y + 1, wxh);
  cross3(Y + 1, X, hxl);
  f[0] = Y[0] * X[0] + wxh[0];
  f[1] = Y[0] * X[1] + wxh[1];
  f[2] = Y[0] * X[2] + wxh[2];
  f[3] = Y[4] * X[3] + Y[5] * X[4] + Y[6] * X[5] + hxl[0];
  f[4] = Y[5] * X[3] + Y[7] * X[4] + Y[8] * X[5] + hxl[1];
  f[5] = Y[6] * X[3] + Y[8] * X[4] + Y[9] * X[5] + hxl[2];
}
// spatial cross products ([lin; ang] about a common origin)
template <typename T> __device__ __forceinline__ void cross_mm(const T *a, const T *b, T *o) {
  T t1[3], t2[3], t3[3];
  cross3(a + 3, b, t1); cross3(a, b + 3, t2); cross3(a + 3, b + 3, t3);
#pragma unroll
  for (int i = 0; i < 3; i++) { o[i] = t1[i] + t2[i]; o[3 + i] = t3[i]; }
}
template <typename T> __device__ __forceinline__ void cross_mf(const T *v, const T *f, T *o) {
  T t1[3], t2[3], t3[3];
  cross3(v + 3, f, t1); cross3(v + 3, f + 3, t2); cross3(v, f, t3);
#pragma unroll
  for (int i = 0; i < 3; i++) { o[i] = t1[i]; o[3 + i] = t2[i] + t3[i]; }
}

// Forward pass of a kinematic tree, lane j = node j (n <= 32 nodes, depth <= 7), world-aligned spatial
// vectors about the common origin O.  In: (R, p) = the node's transform in its parent (root: in the world),
// qd = joint rate, root only: V, A = its spatial velocity / bias acceleration.  Out: (R, p) in the world,
// the joint's motion vector S (non-root), V, A of every node.
//   transforms : pointer jumping - round r composes each node with its ancestor 2^r levels up (tables
//                up0/up1/up2, -1 beyond the root), three rounds instead of one LDS round trip per level
//   V, A       : sums over the node's root path (chain table: the node and its non-root ancestors) of S qd and of
//                V x (S qd), which only add in this representation
// bufA / bufB: two n*12 scratch areas, sq / cb: n rows of >= 6 (strides given), all in LDS.
// LDS hand-over between the lanes of a ONE-wavefront workgroup.  A wavefront's LDS instructions execute in order, so nothing
// has to be waited for - only the compiler has to keep the order; __syncthreads() additionally drains every outstanding LDS
// AND global access (s_waitcnt vmcnt(0) lgkmcnt(0)) at each of the ~60 hand-overs of a step.  Measured (round 3,
// -DTSIDB_WAVE_SYNC, tools/r03_ab.sh): no difference - 16.5 M env-steps/s at 4096 envs and 6.6 M at 512 either way, nothing is
// in flight at those points that the next phase does not need - so the default stays the plain barrier.  With the fences
// restricted to the LDS address space (-DTSIDB_WAVE_SYNC_LOCAL: global loads may then be scheduled across the hand-overs)
// + 0.3-0.5 %, consistently but too little to change the synchronisation the tests have run on.
__device__ __forceinline__ void wave_sync() {
#ifdef TSIDB_WAVE_SYNC_LOCAL // fences for the LDS address space only: global loads may be scheduled across the hand-over
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
#ifdef TSIDB_WAVE_SYNC
#define TSIDB_SYNC1() wave_sync()
#else
#define TSIDB_SYNC1() __syncthreads()
#endif
template <int NW> __device__ __forceinline__ void wg_sync() { // every wavefront of the workgroup
  if constexpr (NW == 0) wave_sync(); // (NW = 0: the packed kernels - one wavefront, two envs, hand-overs inside divergent code)
  else if constexpr (NW == 1) TSIDB_SYNC1();
  else __syncthreads();
}

template <typename T, int NW = 1>
__device__ __forceinline__ void tree_forward(int lane, int n, int up0, int up1, int up2, const int (&chn)[7], T *bufA, T *bufB,
                                             T *sq, int sq_stride, T *cb, int cb_stride, T (&R)[9], T (&p)[3], T qd,
                                             T (&S)[6], T (&V)[6], T (&A)[6]) {
  const bool on = lane < n;
  if (on) {
#pragma unroll
    for (int i = 0; i < 9; i++) bufA[lane * 12 + i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; i++) bufA[lane * 12 + 9 + i] = p[i];
  }
  wg_sync<NW>();
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const int u = r == 0 ? up0 : (r == 1 ? up1 : up2);
    const T *src = (r & 1) ? bufB : bufA;
    T *dst = (r & 1) ? bufA : bufB;
    if (on && u >= 0) {
      T Ru[9], pu[3], Rn[9], pn[3];
#pragma unroll
      for (int i = 0; i < 9; i++) Ru[i] = src[u * 12 + i];
#pragma unroll
      for (int i = 0; i < 3; i++) pu[i] = src[u * 12 + 9 + i];
      mat3mul(Ru, R, Rn);
      mat3vec(Ru, p, pn);
#pragma unroll
      for (int i = 0; i < 9; i++) R[i] = Rn[i];
#pragma unroll
      for (int i = 0; i < 3; i++) p[i] = pu[i] + pn[i];
    }
    if (r < 2) {
      if (on) {
#pragma unroll
        for (int i = 0; i < 9; i++) dst[lane * 12 + i] = R[i];
#pragma unroll
        for (int i = 0; i < 3; i++) dst[lane * 12 + 9 + i] = p[i];
      }
      wg_sync<NW>();
    }
  }
  // joint motion vector (revolute about the node's z axis through p) and its rate; the root hands in V, A
  T Sq[6];
  if (on && lane > 0) {
    const T a[3] = {R[2], R[5], R[8]};
    cross3(p, a, S);
    S[3] = a[0]; S[4] = a[1]; S[5] = a[2];
#pragma unroll
    for (int i = 0; i < 6; i++) Sq[i] = S[i] * qd;
  } else {
#pragma unroll
    for (int i = 0; i < 6; i++) Sq[i] = V[i];
  }
  if (on) {
#pragma unroll
    for (int i = 0; i < 6; i++) sq[lane * sq_stride + i] = Sq[i];
  }
  wg_sync<NW>();
  if (on && lane > 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) V[i] = sq[i]; // the root, then the node's chain from the shallowest ancestor down to itself:
                                              // a static walk (chain table), every load independent - the ancestor
                                              // bit-mask loop waited out one LDS round trip per level; same order of sums
#pragma unroll
    for (int d = 6; d >= 0; d--) {
      const int a = chn[d];
      if (a > 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) V[i] += sq[a * sq_stride + i];
      }
    }
  }
  T cj[6];
  if (on && lane > 0) cross_mm(V, Sq, cj);
  else {
#pragma unroll
    for (int i = 0; i < 6; i++) cj[i] = A[i];
  }
  if (on) {
#pragma unroll
    for (int i = 0; i < 6; i++) cb[lane * cb_stride + i] = cj[i];
  }
  wg_sync<NW>();
  if (on && lane > 0) {
#pragma unroll
    for (int i = 0; i < 6; i++) A[i] = cb[i];
#pragma unroll
    for (int d = 6; d >= 0; d--) {
      const int a = chn[d];
      if (a > 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) A[i] += cb[a * cb_stride + i];
      }
    }
  }
  wg_sync<NW>(); // sq / cb / bufA / bufB are free again
}

// log6 of a relative placement (R row-major, p) -> [v; w]   (pinocchio::log6 semantics)
template <typename T> __device__ __forceinline__ void log6(const T *R, const T *p, T *out) {
  const T PI = T(3.14159265358979323846);
  T tr = R[0] + R[4] + R[8];
  T ct = T(0.5) * (tr - 1);
  ct = ct > 1 ? T(1) : (ct < -1 ? T(-1) : ct);
  T th = acos(ct);
  T sth, cth;
  sincos_t(th, sth, cth);
  T w[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  T k;
  const T small = sizeof(T) == 8 ? T(1e-8) : T(1e-4);
  if (th < small) k = T(0.5) * (1 + th * th / 6);
  else if (th > PI - T(1e-6) * (sizeof(T) == 8 ? 1 : 1000)) {
    T ax[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      T d = (R[4 * i] - ct) / (1 - ct);
      ax[i] = sqrt(d > 0 ? d : T(0));
      if (w[i] < 0) ax[i] = -ax[i];
      w[i] = ax[i] * th;
    }
    k = 0;
  } else k = T(0.5) * th / sth;
  if (k != 0) { w[0] *= k; w[1] *= k; w[2] *= k; }
  T th2 = th * th, beta;
  if (th < T(1e-4) * (sizeof(T) == 8 ? 1 : 100)) beta = T(1.0 / 12) + th2 / 720;
  else beta = (1 - th * sth / (2 * (1 - cth))) / th2;
  T wxp[3], wxwxp[3];
  cross3(w, p, wxp); cross3(w, wxp, wxwxp);
#pragma unroll
  for (int i = 0; i < 3; i++) { out[i] = p[i] - T(0.5) * wxp[i] + beta * wxwxp[i]; out[3 + i] = w[i]; }
}

} // namespace tsidb
