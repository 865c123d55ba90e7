// Micro-benchmark (diagnostic, not part of the product): the dv-block Hessian of the TSID tick, H = sum_r w_r J_r^T J_r over
// the 15 dense task rows (12 foot rows + 3 CoM rows, 26 columns), assembled into "lane i = row i" registers - what
// k_tick's register Cholesky consumes - in two ways:
//   valu   the product code's way: lane i keeps w_r J_r[i]; J_r[j] is read back from LDS at wave-uniform addresses
//          (ds_read2_b64 beside 390 v_fma_f64)
//   mfma   v_mfma_f64_16x16x4_f64: 2 x 2 output tiles x 4 k-steps = 16 MFMAs on operands read from LDS (8 loads), then the
//          16 x 16 accumulator layout (column on the lane, rows over 4 registers x 4 lane groups) is gathered into rows by
//          symmetry + ds_bpermute (there is no LDS left in k_tick for a 26 x 27 staging tile: 176 bytes spare of 20 480)
// One wavefront per workgroup, like k_tick; cycles from s_memtime around REPS repetitions; checks that both agree.
//   hipcc --offload-arch=gfx950 -O3 -o h_bench tools/mfma/h_assembly_bench.hip && ./h_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
constexpr int NV = 26, LDF = 27, NR = 15, REPS = 64;
typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bperm(double v, int src) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffLL)), hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k(const double *Jg, double *out, unsigned long long *cyc, double wf, double wc) {
  __shared__ double J[(NR + 1) * LDF + 64]; // row 15 = zeros (K padded to 16)
  const int lane = threadIdx.x;
  for (int i = lane; i < (NR + 1) * LDF + 64; i += 64) J[i] = i < NR * LDF ? Jg[(size_t)blockIdx.x * NR * LDF + i] : 0.0;
  __syncthreads();
  double a[NV];
  for (int j = 0; j < NV; j++) a[j] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < REPS; rep++) {
    if (MODE == 0) {
      double wjt[NR];
#pragma unroll
      for (int r = 0; r < NR; r++) wjt[r] = (lane < NV ? J[r * LDF + lane] : 0.0) * (r < 12 ? wf : wc);
#pragma unroll
      for (int j = 0; j < NV; j++) {
        double acc = 0, acc1 = 0;
#pragma unroll
        for (int r = 0; r < NR; r++) {
          if (r & 1) acc1 += wjt[r] * J[r * LDF + j];
          else acc += wjt[r] * J[r * LDF + j];
        }
        a[j] += acc + acc1;
      }
    } else {
      const int c = lane & 15, g = lane >> 4;
      double x[4][2];
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int r = 4 * ks + g, i = 16 * t + c;
          x[ks][t] = i < NV ? J[r * LDF + i] : 0.0; // (row 15 is zero)
        }
      v4d D[2][2];
#pragma unroll
      for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int tj = 0; tj < 2; tj++) D[ti][tj] = (v4d){0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        const double w = (4 * ks + g) < 12 ? wf : wc;
#pragma unroll
        for (int ti = 0; ti < 2; ti++)
#pragma unroll
          for (int tj = 0; tj < 2; tj++) D[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(w * x[ks][ti], x[ks][tj], D[ti][tj], 0, 0, 0);
      }
      // lane (c, g) holds H[16 ti + g + 4 reg][16 tj + c] = (symmetry) H[16 tj + c][16 ti + g + 4 reg]: row 16 tj + c of H is spread
      // over the four lanes c + 16 g.  Row i goes to lane i: lane i < 16 takes tj = 0, c = i; lane 16 + c takes tj = 1.
      // Entry j = 16 ti + q + 4 reg of that row sits on lane c + 16 q, register reg of tile (ti, tj).
      const int tjm = lane >= 16 ? 1 : 0;
#pragma unroll
      for (int ti = 0; ti < 2; ti++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int j = 16 * ti + q + 4 * reg;
            if (j >= NV) continue;
            const double v0 = bperm(D[ti][0][reg], (lane & 15) + 16 * q), v1 = bperm(D[ti][1][reg], (lane & 15) + 16 * q);
            a[j] += tjm ? v1 : v0;
          }
    }
#pragma unroll
    for (int j = 0; j < NV; j++) asm volatile("" : "+v"(a[j]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane < NV)
    for (int j = 0; j < NV; j++) out[((size_t)blockIdx.x * NV + lane) * NV + j] = a[j];
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int NB = 4096;
  std::vector<double> J((size_t)NB * NR * LDF);
  srand(1);
  for (auto &v : J) v = rand() / (double)RAND_MAX - 0.5;
  double *dJ, *o0, *o1;
  unsigned long long *c0, *c1;
  hipMalloc(&dJ, J.size() * 8); hipMalloc(&o0, (size_t)NB * NV * NV * 8); hipMalloc(&o1, (size_t)NB * NV * NV * 8);
  hipMalloc(&c0, NB * 8); hipMalloc(&c1, NB * 8);
  hipMemcpy(dJ, J.data(), J.size() * 8, hipMemcpyHostToDevice);
  for (int nb : {512, 4096}) {
    for (int w = 0; w < 2; w++) {
      hipLaunchKernelGGL(k<0>, dim3(nb), dim3(64), 0, 0, dJ, o0, c0, 1.0, 0.7);
      hipLaunchKernelGGL(k<1>, dim3(nb), dim3(64), 0, 0, dJ, o1, c1, 1.0, 0.7);
    }
    hipDeviceSynchronize();
    std::vector<double> h0((size_t)nb * NV * NV), h1(h0.size());
    std::vector<unsigned long long> y0(nb), y1(nb);
    hipMemcpy(h0.data(), o0, h0.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), o1, h1.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(y0.data(), c0, nb * 8, hipMemcpyDeviceToHost); hipMemcpy(y1.data(), c1, nb * 8, hipMemcpyDeviceToHost);
    double err = 0, mx = 0, s0 = 0, s1 = 0;
    for (size_t i = 0; i < h0.size(); i++) { err = fmax(err, fabs(h0[i] - h1[i])); mx = fmax(mx, fabs(h0[i])); }
    for (int i = 0; i < nb; i++) { s0 += y0[i]; s1 += y1[i]; }
    printf("%d workgroups: valu %.0f cycles per assembly, mfma + gather %.0f (s_memtime ticks / %d reps); max |diff| %.2e of %.2e\n", nb,
           s0 / nb / REPS, s1 / nb / REPS, REPS, err, mx);
  }
  return 0;
}
