// Operand-layout, scale and issue-rate probe for v_mfma_scale_f32_32x32x64_f8f6f4 (gfx950), the instruction the
// 8-bit cross terms of the split-precision attention would run on: exact small values in e4m3 / e5m2 / e2m3 laid out as
// lane l (r = l & 31, h = l >> 5) holds A[r][32 h + j] / B[32 h + j][r] in element j of its operand (bytes for the 8-bit
// formats, 6-bit fields packed little-endian for fp6), checked against a host sum; then E8M0 scale bytes; then rates.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_scale_probe.bin mfma_scale_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// value set {0.5, 1, 1.5, 2} in each format
__host__ __device__ inline float val(int i) { return 0.5f * (float)(i + 1); }
__host__ __device__ inline uint32_t enc(int fmt, int i) {
  const uint8_t e4m3[4] = {0x30, 0x38, 0x3C, 0x40}, e5m2[4] = {0x38, 0x3C, 0x3E, 0x40}, e2m3[4] = {0x04, 0x08, 0x0C, 0x10};
  return fmt == 0 ? e4m3[i] : (fmt == 1 ? e5m2[i] : e2m3[i]);
}
__host__ __device__ inline int ai(int r, int k) { return (r * 7 + k * 3) & 3; }
__host__ __device__ inline int bi(int k, int c) { return (k * 5 + c + (k >> 4)) & 3; }

template <int FMT>
__global__ void probe(float* out, int sa, int sb) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const uint32_t ea = enc(FMT, ai(r, 32 * h + j)), eb = enc(FMT, bi(32 * h + j, r));
    if (FMT < 2) {
      a[j >> 2] |= ea << (8 * (j & 3));
      b[j >> 2] |= eb << (8 * (j & 3));
    } else {  // 6-bit fields, little-endian bit stream
      const int bit = 6 * j;
      a[bit >> 5] |= ea << (bit & 31);
      b[bit >> 5] |= eb << (bit & 31);
      if ((bit & 31) > 26) { a[(bit >> 5) + 1] |= ea >> (32 - (bit & 31)); b[(bit >> 5) + 1] |= eb >> (32 - (bit & 31)); }
    }
  }
  i32x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (int)a[i]; bv[i] = (int)b[i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, FMT, FMT, 0, sa, 0, sb);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    out[row * 32 + r] = c[i];
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  f32x16 acc[2];
  for (int u = 0; u < 2; ++u)
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
  const int t = threadIdx.x;
  f16x8 a, b;
  i32x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (t + i)); b[i] = (_Float16)(0.002f * (t - i)); a8[i] = 0x38383838 + t; b8[i] = 0x3c3c3c3c - t; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
      else if constexpr (KIND == 1) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 0, 0, 0, 127, 0, 127);
      else if constexpr (KIND == 2) acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 1, 1, 0, 127, 0, 117);
      else acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u], 2, 2, 0, 127, 0, 127);
    }
  }
  float s = 0.f;
  for (int u = 0; u < 2; ++u)
    for (int i = 0; i < 16; ++i) s += acc[u][i];
  out[blockIdx.x * 256 + t] = s;
}

template <int KIND>
double run_rate(const char* name, double flop) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  const int iters = 20000, blocks = 256 * 2;  // 2 x 4 waves per CU = 2 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tf = (double)blocks * 4 * iters * 2 * flop / (ms * 1e-3) / 1e12;
  printf("%-44s %8.2f ms  %8.1f TFLOP/s\n", name, ms, tf);
  hipFree(out);
  return tf;
}

template <int FMT>
int check(const char* name, int sa, int sb) {
  float* d;
  hipMalloc(&d, 1024 * sizeof(float));
  hipLaunchKernelGGL(probe<FMT>, dim3(1), dim3(64), 0, 0, d, sa, sb);
  float hst[1024];
  hipMemcpy(hst, d, sizeof(hst), hipMemcpyDeviceToHost);
  const float scale = ldexpf(1.f, (sa & 255) - 127 + (sb & 255) - 127);
  int bad = 0;
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      float ref = 0.f;
      for (int k = 0; k < 64; ++k) ref += val(ai(r, k)) * val(bi(k, c));
      ref *= scale;
      if (hst[r * 32 + c] != ref) { if (bad < 3) printf("  %s [%d][%d] got %g want %g\n", name, r, c, hst[r * 32 + c], ref); ++bad; }
    }
  printf("%-24s scale bytes (%d, %d): %s (%d of 1024 differ)\n", name, sa & 255, sb & 255, bad ? "MISMATCH" : "layout + scale OK", bad);
  hipFree(d);
  return bad;
}

int main() {
  int bad = 0;
  bad += check<0>("e4m3 x e4m3", 127, 127);
  bad += check<1>("e5m2 x e5m2", 127, 127);
  bad += check<1>("e5m2 x e5m2", 127, 117);
  bad += check<0>("e4m3 x e4m3", 120, 130);
  bad += check<2>("e2m3 x e2m3 (fp6)", 127, 127);
  bad += check<2>("e2m3 x e2m3 (fp6)", 125, 127);
  const double f16 = run_rate<0>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16);
  const double f8 = run_rate<1>("v_mfma_scale_f32_32x32x64_f8f6f4 e4m3", 2.0 * 32 * 32 * 64);
  const double b8 = run_rate<2>("v_mfma_scale_f32_32x32x64_f8f6f4 e5m2", 2.0 * 32 * 32 * 64);
  const double f6 = run_rate<3>("v_mfma_scale_f32_32x32x64_f8f6f4 e2m3 (fp6)", 2.0 * 32 * 32 * 64);
  printf("rate vs fp16: e4m3 %.2fx  e5m2 %.2fx  fp6 %.2fx\n", f8 / f16, b8 / f16, f6 / f16);
  return bad ? 1 : 0;
}
