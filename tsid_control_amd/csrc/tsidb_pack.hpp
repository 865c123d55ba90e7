// tsidb_pack.hpp - cross-lane primitives of the PACKED layout: two envs per wavefront, env h on lanes [32 h, 32 h + 32).
//
// Why: the one-env-per-wavefront kernels use 26 (dofs) / 21 (bodies) / <= 32 (contacts) of the 64 lanes, so 59 % of every
// vector instruction is idle lanes, and a float64 broadcast (lane k -> all) costs two v_readlane through the scalar
// unit.  With two envs side by side every per-lane instruction serves two envs, and the broadcast becomes a DPP move that
// stays in the vector unit:
//   dup(x)     one v_permlane16_swap per dword (gfx950) makes two registers whose 16-lane DPP rows BOTH hold the env's lanes
//              0..15 (a) resp. 16..31 (b);
//   bc<K>(d)   v_mov_b32_dpp row_newbcast:K%16 per dword of a (K < 16) or b - lane K of the env in every lane of the env, one VALU
//              instruction for both envs, no SGPR, no v_readlane -> VALU hazard.
// Reductions are the 64-lane ones cut after the row step: quad_perm / row_half_mirror / row_mirror give every lane its row
// total, then dup() + one add give R0 + R1 to both rows of the env.  Same operands in the same order as tsidb_common.hpp's
// wave_sum on a vector that is zero on lanes >= 32 (there: (0 + 0) + (R1 + R0)), so the packed kernels reproduce the
// one-env kernels' sums bit for bit.
// Values that are wave-uniform in the one-env kernels (contact counts, iteration counters, Newton state) are uniform per
// ENV here and live in VGPRs; control flow on them is ordinary divergent SIMT code, and every cross-lane operation in this
// file only ever reads lanes of the caller's own env, so an env whose lanes are masked off does not disturb its neighbour.
#pragma once
#include "tsidb_common.hpp"

namespace tsidb {
namespace pk {

constexpr int LPE = 32; // lanes per env

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

// rows 1 / 3 of x0 <-> rows 0 / 2 of x1
__device__ __forceinline__ void swap16(unsigned &x0, unsigned &x1) {
  const v2u_t r = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
  x0 = r[0];
  x1 = r[1];
}

template <typename T> struct Dup { T a, b; };

template <typename T> __device__ __forceinline__ Dup<T> dup(T x) {
  Dup<T> d;
  if constexpr (sizeof(T) == 4) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    swap16(a, b);
    d.a = __builtin_bit_cast(T, a);
    d.b = __builtin_bit_cast(T, b);
  } else {
    const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
    unsigned al = (unsigned)v, ah = (unsigned)(v >> 32), bl = al, bh = ah;
    swap16(al, bl);
    swap16(ah, bh);
    d.a = __builtin_bit_cast(T, ((unsigned long long)ah << 32) | al);
    d.b = __builtin_bit_cast(T, ((unsigned long long)bh << 32) | bl);
  }
  return d;
}

template <int CTRL, typename T> __device__ __forceinline__ T dpp_all(T v) { return dpp_mov<CTRL, 0xf>(v); } // every row enabled
// (float64: two v_mov_b32_dpp - this compiler's __builtin_amdgcn_update_dpp returns int whatever its operands, so
//  v_mov_b64_dpp is only reachable through inline asm, whose DPP read hazard the compiler would not see)

// lane K (0..31, compile time) of the caller's env
template <int K, typename T> __device__ __forceinline__ T bc(const Dup<T> &d) {
  static_assert(K >= 0 && K < LPE, "lane of the env");
  return dpp_all<0x150 + (K & 15)>(K < 16 ? d.a : d.b);
}
// one value only: a row-local broadcast, then the row that holds it is copied over the other one
template <int K, typename T> __device__ __forceinline__ T bc1(T x) {
  const Dup<T> d = dup(dpp_all<0x150 + (K & 15)>(x));
  return K < 16 ? d.a : d.b;
}
// run-time lane (uniform per env): through the LDS crossbar (rare paths only)
template <typename T> __device__ __forceinline__ T bc_dyn(T x, int k, int lane) { return bperm(x, (lane & LPE) | k); }

template <typename T> __device__ __forceinline__ T sum(T v) {
  v += dpp_mov<0xB1, 0xf>(v);  // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xf>(v);  // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xf>(v); // row_half_mirror
  v += dpp_mov<0x140, 0xf>(v); // row_mirror
  const Dup<T> d = dup(v);
  return d.b + d.a; // (R1 + R0: the order wave_sum's row_bcast15 step adds in)
}
template <typename T> __device__ __forceinline__ void sum3(T &a, T &b, T &c) {
#define TSIDB_PK_STEP(CTRL)                                                                          \
  {                                                                                                  \
    const T ta = dpp_mov<CTRL, 0xf>(a), tb = dpp_mov<CTRL, 0xf>(b), tc = dpp_mov<CTRL, 0xf>(c);      \
    a += ta; b += tb; c += tc;                                                                       \
  }
  TSIDB_PK_STEP(0xB1) TSIDB_PK_STEP(0x4E) TSIDB_PK_STEP(0x141) TSIDB_PK_STEP(0x140)
#undef TSIDB_PK_STEP
  const Dup<T> da = dup(a), db = dup(b), dc = dup(c);
  a = da.b + da.a; b = db.b + db.a; c = dc.b + dc.a;
}
__device__ __forceinline__ int sum_int(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true);
  const Dup<int> d = dup(v);
  return d.a + d.b;
}
template <typename T> __device__ __forceinline__ T min(T v) {
  v = dpp_min_step<0xB1, 0xf>(v);
  v = dpp_min_step<0x4E, 0xf>(v);
  v = dpp_min_step<0x141, 0xf>(v);
  v = dpp_min_step<0x140, 0xf>(v);
  const Dup<T> d = dup(v);
  return d.b < d.a ? d.b : d.a;
}
template <typename T> __device__ __forceinline__ void argmin(T &v, int &i) {
  const T vmin = min(v);
  i = min<int>(v == vmin ? i : 0x7fffffff);
  v = vmin;
}
// the env's 32 ballot bits (bit j = lane j of the env)
__device__ __forceinline__ unsigned ballot(bool p, int lane) {
  const unsigned long long m = __ballot(p);
  return (lane & LPE) ? (unsigned)(m >> 32) : (unsigned)m;
}

// LDS hand-over between the lanes of the one wavefront (its LDS instructions execute in order: only the compiler is told)
__device__ __forceinline__ void sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace pk
} // namespace tsidb
