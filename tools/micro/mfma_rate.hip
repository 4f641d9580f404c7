// MFMA issue-rate microbenchmark for the round-2 precision plan: fp16 16x16x32 vs block-scaled fp8 16x16x128
// (v_mfma_scale_f32_16x16x128_f8f6f4), 8 waves per CU, 4 independent accumulators per wave, operands in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int t = threadIdx.x;
  f16x8 a, b;
  i32x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (t + i)); b[i] = (_Float16)(0.002f * (t - i)); a8[i] = 0x38383838 + t; b8[i] = 0x3c3c3c3c - t; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if constexpr (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
      else acc[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[u], 0, 0, 0, 127, 0, 127);
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  out[blockIdx.x * 512 + t] = s[0] + s[1] + s[2] + s[3];
}

template <int KIND>
double run(const char* name, double flop_per_mfma) {
  float* out;
  hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
  const int iters = 20000, blocks = 256 * 2;  // 2 x 8 waves per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)blocks * 8 * iters * 4;
  const double tf = mfmas * flop_per_mfma / (ms * 1e-3) / 1e12;
  printf("%-34s %8.2f ms  %8.1f TFLOP/s\n", name, ms, tf);
  hipFree(out);
  return tf;
}

int main() {
  const double f16 = run<0>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32);
  const double f8 = run<1>("v_mfma_scale_f32_16x16x128_f8f6f4", 2.0 * 16 * 16 * 128);
  printf("fp8 / fp16 rate under load: %.2fx\n", f8 / f16);
  return 0;
}
