// Do VALU instructions issue in the shadow of an executing MFMA — and does it matter whether the MFMA's accumulator
// lives in the architectural VGPRs or in the AGPR half of the register file, and whether one or two waves share the SIMD?
// Each wave runs [v_mfma_f32_32x32x16_f16 + NFILL independent VALU] x 8 per iteration (4 independent accumulators,
// operands in registers, no memory). Prints ns per MFMA slot per SIMD; the NFILL = 0 line is the bare matrix-pipe rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o coexec_probe coexec_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int AGPR, int NFILL, int NEXP, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
  const int t = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((t + i) & 63)); b[i] = (_Float16)(0.002f * ((t - i) & 63)); }
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.001f * (t + j);
  const float c1 = 0.999f, c2 = 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NFILL; ++j) {
        if (j < NEXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j & 7]));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) s += acc[u][i];
  for (int j = 0; j < 8; ++j) s += x[j];
  out[blockIdx.x * THREADS + t] = s;
}

template <int AGPR, int NFILL, int NEXP, int THREADS>
void run() {
  float* out;
  hipMalloc(&out, 256 * THREADS * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<AGPR, NFILL, NEXP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, 200);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<AGPR, NFILL, NEXP, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = THREADS / 256;
  const double slots = (double)iters * 8 * waves_per_simd;  // MFMAs one SIMD executed
  printf("acc in %s  waves/SIMD %d  fillers %d (%d v_exp)   %7.2f ns per MFMA slot   (%.0f TFLOP/s)\n", AGPR ? "AGPR" : "VGPR", waves_per_simd,
         NFILL, NEXP, ms * 1e6 / slots, 1024.0 * slots * 32768 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

#define ROW(A, T) run<A, 0, 0, T>(); run<A, 2, 0, T>(); run<A, 4, 0, T>(); run<A, 5, 0, T>(); run<A, 6, 0, T>(); run<A, 8, 0, T>(); \
                  run<A, 4, 1, T>(); run<A, 5, 2, T>(); run<A, 7, 2, T>();
int main() {
  ROW(0, 256) ROW(1, 256) ROW(0, 512) ROW(1, 512)
  return 0;
}
