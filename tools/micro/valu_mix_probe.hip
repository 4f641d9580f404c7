// What does the softmax VALU mix of the attention kernel cost, alone and in the shadow of MFMAs?
//  (a) VALU-only loops of one instruction kind (independent registers): ns per wave-instruction per SIMD, 1 and 2 waves/SIMD;
//  (b) per MFMA gap the work of one score pair in software-pipelined form (2 v_fma with an SGPR operand, 2 v_exp,
//      v_cvt_pkrtz, v_dot2c; the stages use different registers) behind a v_mfma_f32_32x32x16_f16.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix_probe.bin valu_mix_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND: 0 v_fma (vgpr operands), 1 v_fma with SGPR, 2 v_exp, 3 v_cvt_pkrtz, 4 v_dot2c literal, 5 v_max3, 6 mix of a pair
template <int KIND, int MFMA, int THREADS, int NACC = 4>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, float sc) {
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
  const int t = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((t + i) & 63)); b[i] = (_Float16)(0.002f * ((t - i) & 63)); }
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  i32x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = 0x38383838 + t; b8[i] = 0x3c3c3c3c - t; }
  float x[8], y[8], z[8];
  unsigned w[8];
  float ls0 = 0.f, ls1 = 0.f;
  for (int j = 0; j < 8; ++j) { x[j] = 0.001f * (t + j); y[j] = -0.5f - 0.01f * j; z[j] = 0.25f; w[j] = 0x3c003c00u; }
  const float c1 = 0.999f, c2 = 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (MFMA == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
      if constexpr (MFMA == 2) {  // the attention kernel's pattern: two chains (S, O) alternating, every 7th MFMA a block-scaled 8-bit one
        if (u == 3) acc[u % NACC] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[u % NACC], 1, 1, 0, 127, 0, 117);
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
      }
      if constexpr (KIND == 6) {  // stage F of pair u+2, X of pair u+1, C of pair u (registers of different pairs)
        asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %2, %5, %4" : "=v"(x[(u + 2) & 7]), "=v"(x[(u + 6) & 7]) : "s"(sc), "v"(y[u]), "v"(c2), "v"(y[(u + 1) & 7]));
        asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(z[(u + 1) & 7]), "=v"(z[(u + 5) & 7]) : "v"(x[(u + 1) & 7]), "v"(x[(u + 5) & 7]));
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w[u & 7]) : "v"(z[u & 7]), "v"(z[(u + 4) & 7]));
        if (u & 1) asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls1) : "v"(w[(u + 7) & 7]));
        else asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls0) : "v"(w[(u + 7) & 7]));
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(c1), "v"(c2));
          if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(x[j]) : "s"(sc), "v"(c2));
          if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(z[j]) : "v"(y[j]));
          if constexpr (KIND == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w[j]) : "v"(x[j]), "v"(y[j]));
          if constexpr (KIND == 4) { if (j & 1) asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls1) : "v"(w[j])); else asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls0) : "v"(w[j])); }
          if constexpr (KIND == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(y[(j + 1) & 7]));
        }
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = ls0 + ls1;
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) s += acc[u][i];
  for (int j = 0; j < 8; ++j) s += x[j] + z[j] + (float)w[j];
  out[blockIdx.x * THREADS + t] = s;
}

// The attention kernel's register traffic between the pipes: the VALU reads the accumulators the MFMAs of the PREVIOUS
// iteration wrote (S(t)) while this iteration's MFMAs write the other pair (S(t+1)), and MFMA B operands are registers the
// VALU converts wrote one iteration ago (P). MODE 0: VALU reads plain registers (control), 1: reads the other accumulator pair,
// 2: also the MFMA B operand comes from the converts' outputs.
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k2(float* out, int iters, float sc) {
  f32x16 acc[4];
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) acc[u][i] = -0.01f * i;
  const int t = threadIdx.x;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * ((t + i) & 63)); b[i] = (_Float16)(0.002f * ((t - i) & 63)); }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pw[2] = {{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}};
  float x[8], z[8], y[8];
  float ls0 = 0.f, ls1 = 0.f;
  for (int j = 0; j < 8; ++j) { x[j] = 0.f; z[j] = 0.25f; y[j] = -0.5f - 0.01f * j; }
  const float c2 = -1.0f;
#define XBODY(P)                                                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                                                  \
    if constexpr (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[2 * P + (u & 1)]) : "v"(a), "v"(pw[1 - P])); \
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[2 * P + (u & 1)]) : "v"(a), "v"(b));                     \
    const float s0 = MODE ? acc[2 * (1 - P) + (u >> 2)][2 * (u & 3)] : y[u], s1 = MODE ? acc[2 * (1 - P) + (u >> 2)][2 * (u & 3) + 1] : y[(u + 1) & 7]; \
    asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_fma_f32 %1, %2, %5, %4" : "=v"(x[(u + 2) & 7]), "=v"(x[(u + 6) & 7]) : "s"(sc), "v"(s0), "v"(c2), "v"(s1)); \
    asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=v"(z[(u + 1) & 7]), "=v"(z[(u + 5) & 7]) : "v"(x[(u + 1) & 7]), "v"(x[(u + 5) & 7])); \
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pw[P][u & 3]) : "v"(z[u & 7]), "v"(z[(u + 4) & 7]));                       \
    if (u & 1) asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls1) : "v"(pw[P][(u + 3) & 3]));                            \
    else asm volatile("v_dot2c_f32_f16 %0, 0x3c003c00, %1" : "+v"(ls0) : "v"(pw[P][(u + 3) & 3]));                                  \
  }
  for (int it = 0; it < iters; it += 2) {
    XBODY(0)
    XBODY(1)
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = ls0 + ls1;
  for (int u = 0; u < 4; ++u)
    for (int i = 0; i < 16; ++i) s += acc[u][i];
  for (int j = 0; j < 8; ++j) s += x[j] + z[j];
  out[blockIdx.x * THREADS + t] = s + (float)pw[0][0] + (float)pw[1][1];
}

template <int MODE, int THREADS>
void run2(const char* name) {
  float* out;
  (void)hipMalloc(&out, 256 * THREADS * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, 200, 0.18f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k2<MODE, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, iters, 0.18f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int wps = THREADS / 256;
  printf("%-58s waves/SIMD %d   %7.2f ns per gap (MFMA + pair)\n", name, wps, ms * 1e6 / ((double)iters * 8 * wps));
  (void)hipFree(out);
}

template <int KIND, int MFMA, int THREADS, int NACC = 4>
void run(const char* name) {
  float* out;
  (void)hipMalloc(&out, 256 * THREADS * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, MFMA, THREADS, NACC>), dim3(256), dim3(THREADS), 0, 0, out, 200, 0.18f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, MFMA, THREADS, NACC>), dim3(256), dim3(THREADS), 0, 0, out, iters, 0.18f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const int wps = THREADS / 256;
  const double slots = (double)iters * 8 * wps;  // gaps one SIMD executed (6 VALU each)
  printf("%-34s MFMA %d acc %d  waves/SIMD %d   %7.2f ns per gap of 6 VALU = %5.2f ns per VALU instruction\n", name, MFMA, NACC, wps, ms * 1e6 / slots, ms * 1e6 / slots / 6);
  (void)hipFree(out);
}
#define ALL(M, T) run<0, M, T>("v_fma_f32 (vgpr)"); run<1, M, T>("v_fma_f32 (sgpr operand)"); run<2, M, T>("v_exp_f32"); run<3, M, T>("v_cvt_pkrtz_f16_f32"); \
                  run<4, M, T>("v_dot2c_f32_f16 (literal)"); run<5, M, T>("v_max3_f32"); run<6, M, T>("pair: 2 fma 2 exp cvt dot2c");
#define DEP(T) run<6, 1, T, 1>("pair, chain on ONE accumulator"); run<6, 1, T, 2>("pair, two alternating chains"); run<6, 1, T, 4>("pair, four chains"); \
               run<6, 2, T, 2>("pair, two chains + 1/8 scaled MFMA"); run<0, 1, T, 2>("6 v_fma, two chains"); run<0, 1, T, 1>("6 v_fma, one chain");
int main() {
  if (getenv("PROBE_XDEP")) {
    run2<0, 256>("control: VALU reads plain registers"); run2<1, 256>("VALU reads the accumulators of the previous iteration");
    run2<2, 256>("... and the MFMA B operand is last iteration's converts");
    run2<0, 512>("control: VALU reads plain registers"); run2<1, 512>("VALU reads the accumulators of the previous iteration");
    run2<2, 512>("... and the MFMA B operand is last iteration's converts");
    return 0;
  }
  if (getenv("PROBE_DEP")) { DEP(256) DEP(512) return 0; }
  ALL(0, 256) ALL(0, 512) ALL(1, 256) ALL(1, 512)
  return 0;
}
