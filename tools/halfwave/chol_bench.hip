// Micro-benchmark (diagnostic, not part of the product): k_sim's tree-sparse 26 x 26 factorisation + the two substitutions
//   mode 0  the product's one-env-per-wavefront code (tsidb_sim.hpp: chol26_factor / chol26_subst; v_readlane broadcasts),
//           20 KB of LDS per workgroup = 8 workgroups per CU, two wavefronts per SIMD - the residency k_sim runs at
//   mode 1  two envs per wavefront (tsidb_pack.hpp: DPP row_newbcast broadcasts), 40 KB per workgroup = 4 per CU, one
//           wavefront per SIMD - the same 8 envs per CU
// Both produce the same bits (checked).  Time = one launch of NENV envs x REPS repetitions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tsid_control_amd/csrc -o tools/halfwave/chol_bench tools/halfwave/chol_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "tsidb_sim.hpp"
#include "tsidb_pack.hpp"
using namespace tsidb;

template <int K0, int K1, typename F> __device__ __forceinline__ void static_for(F &&f) {
  if constexpr (K0 < K1) { f(std::integral_constant<int, K0>{}); static_for<K0 + 1, K1>(f); }
}

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

constexpr int REPS = 32;
template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == 0 ? 2 : 1))) void k(const double *Ag, const double *rg, double *xg, int n) {
  extern __shared__ double lds[]; // [EPW][2][NV * LDM]: the matrix (as k_sim keeps M), the parked factor
  const int lane = threadIdx.x;
  constexpr int EPW = MODE == 0 ? 1 : 2;
  const int hl = MODE == 0 ? lane : (lane & 31), hf = MODE == 0 ? 0 : (lane >> 5);
  const int e = blockIdx.x * EPW + hf;
  double *Mx = lds + hf * 2 * NV * LDM, *H = Mx + NV * LDM;
  double arow[NV], rhs = hl < NV ? rg[(size_t)e * NV + hl] : 0.0, x = 0;
  if (hl < NV) {
    for (int j = 0; j < NV; j++) Mx[hl * LDM + j] = Ag[((size_t)e * NV + hl) * NV + j];
  }
  pk::sync();
  for (int rep = 0; rep < REPS; rep++) {
#pragma unroll
    for (int j = 0; j < NV; j++) arow[j] = hl < NV ? Mx[hl * LDM + j] : 0.0;
    double rdv;
    bool spd;
    if constexpr (MODE == 0) chol26_factor<double>(arow, rdv, lane, spd);
    else chol26_factor_p<double>(arow, rdv, hl, spd);
    if (hl < NV) {
#pragma unroll
      for (int j = 0; j < NV; j++) H[hl * LDM + j] = arow[j];
    }
    pk::sync();
    if constexpr (MODE == 0) x = chol26_subst<double>(arow, H, rdv, rhs + x, lane);
    else x = chol26_subst_p<double>(arow, H, rdv, rhs + x, hl);
    pk::sync();
  }
  if (hl < NV) xg[(size_t)e * NV + hl] = x;
}

int main() {
  const int n = 8192;
  std::vector<double> A((size_t)n * NV * NV), r((size_t)n * NV);
  srand(1);
  for (int e = 0; e < n; e++) {
    double U[NV][NV] = {};
    for (int k = 0; k < NV; k++) {
      U[k][k] = 1.0 + (rand() % 1000) * 1e-3;
      for (int i = 0; i < k; i++) if ((MJ_DOFANC[k] >> i) & 1u) U[i][k] = ((rand() % 2001) - 1000) * 3e-4;
    }
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) {
      double s = 0;
      for (int k = 0; k < NV; k++) s += U[i][k] * U[j][k];
      A[((size_t)e * NV + i) * NV + j] = s;
    }
    for (int i = 0; i < NV; i++) r[(size_t)e * NV + i] = ((rand() % 2001) - 1000) * 1e-3;
  }
  double *dA, *dr, *dx0, *dx1;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&dr, r.size() * 8); hipMalloc(&dx0, r.size() * 8); hipMalloc(&dx1, r.size() * 8);
  hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dr, r.data(), r.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[2] = {0, 0};
  // latency of one wavefront (few envs: every wavefront alone on its SIMD) up to the full-residency throughput
  for (int ne : {256, 1024, 2048, 4096, 8192}) {
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<0>, dim3(ne), dim3(64), 20480, 0, dA, dr, dx0, ne);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<1>, dim3(ne / 2), dim3(64), 40960, 0, dA, dr, dx1, ne);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    }
    printf("envs %5d: one env per wavefront %.4f ms, two envs per wavefront %.4f ms (x %.2f)\n", ne, ms[0], ms[1], ms[0] / ms[1]);
  }
  std::vector<double> x0(r.size()), x1(r.size());
  hipMemcpy(x0.data(), dx0, r.size() * 8, hipMemcpyDeviceToHost);
  hipMemcpy(x1.data(), dx1, r.size() * 8, hipMemcpyDeviceToHost);
  size_t ndiff = 0; double md = 0, mx = 0;
  for (size_t i = 0; i < x0.size(); i++) { if (x0[i] != x1[i]) ndiff++; md = fmax(md, fabs(x0[i] - x1[i])); mx = fmax(mx, fabs(x0[i])); }
  printf("envs %d reps %d: one env per wavefront %.3f ms, two envs per wavefront %.3f ms (x %.2f); differing values %zu of %zu, max |diff| %.3e, max |x| %.3e\n",
         n, REPS, ms[0], ms[1], ms[0] / ms[1], ndiff, x0.size(), md, mx);
  printf("per env per factor+solve: %.1f ns vs %.1f ns\n", ms[0] * 1e6 / ((double)n * REPS), ms[1] * 1e6 / ((double)n * REPS));
  return ndiff != 0;
}
