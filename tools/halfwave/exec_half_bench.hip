// Micro-benchmark (diagnostic): does a SIMD-32 of gfx950 skip the half of a wave64 instruction whose 32 lanes are all
// masked off in EXEC?  Streams of independent v_fma_f64 / v_fma_f32 with EXEC = all 64 lanes against EXEC = lanes 0..31
// only, at 1 / 2 / 4 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/halfwave/exec_half_bench tools/halfwave/exec_half_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int HALF>
__global__ __launch_bounds__(64) void k(T *out, int reps, T a, T b) {
  T x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  if (!HALF || threadIdx.x < 32) {
    for (int r = 0; r < reps; r++) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <typename T, int HALF> float run(int waves_per_simd, T *d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 4 * waves_per_simd, reps = 2000;
  float ms = 0;
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, HALF>), dim3(grid), dim3(64), 0, 0, d, reps, (T)1.0000001, (T)1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e-3f * 2.1e9f / (reps * 128.0f); // cycles per instruction per wave at ~2.1 GHz
}
int main() {
  double *d; hipMalloc(&d, 8 * 64 * 256 * 4 * 8);
  for (int w : {1, 2, 4}) {
    printf("%d wavefront(s) per SIMD, wall cycles (at 2.1 GHz) per instruction of each wavefront:  f64 fma full %.2f  half-EXEC %.2f |  f32 fma full %.2f  half-EXEC %.2f\n", w,
           run<double, 0>(w, d), run<double, 1>(w, d), run<float, 0>(w, (float *)d), run<float, 1>(w, (float *)d));
  }
  return 0;
}
