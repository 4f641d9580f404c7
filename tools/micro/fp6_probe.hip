// Probe for block-scaled 6-bit (e3m2 "bf6") cross-term planes on gfx950:
//  1. v_cvt_scalef32_2xpk16_bf6_f32 / v_cvt_scalef32_pk32_bf6_f16: element order of the 192-bit result, meaning of the
//     scale operand, rounding and saturation;
//  2. v_mfma_scale_f32_32x32x64_f8f6f4 with e3m2 operands and PER-LANE E8M0 scale bytes: which rows / K halves a lane's
//     scale byte applies to.
// Build: hipcc --offload-arch=gfx950 -O3 -o fp6_probe.bin fp6_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline float e3m2_value(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 2) & 7, m = c & 3;
  const float v = e ? ldexpf(1.f + 0.25f * m, e - 3) : ldexpf(0.25f * m, -2);
  return s ? -v : v;
}
__host__ __device__ inline unsigned field(const unsigned* w, int j) {
  const int bit = 6 * j;
  unsigned v = w[bit >> 5] >> (bit & 31);
  if ((bit & 31) > 26) v |= w[(bit >> 5) + 1] << (32 - (bit & 31));
  return v & 63;
}

__global__ void cvt_probe(const float* x, float scale, unsigned* out32, unsigned* out16) {
  const int l = threadIdx.x;
  f32x16 a, b;
  f16x32 hh;
  for (int j = 0; j < 16; ++j) { a[j] = x[l * 32 + j]; b[j] = x[l * 32 + 16 + j]; }
  for (int j = 0; j < 32; ++j) hh[j] = (_Float16)x[l * 32 + j];
  const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_bf6_f32(a, b, scale);
  const u32x6 q = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(hh, scale);
  for (int i = 0; i < 6; ++i) { out32[l * 6 + i] = r[i]; out16[l * 6 + i] = q[i]; }
}

// A[r][k] = 1, B[k][c] = 1 in e3m2 (code 0x0C); the first operand's scale byte varies per lane, the second's is 127
__global__ void mfma_probe(float* out, int vary_second) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  unsigned w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const int bit = 6 * j;
    w[bit >> 5] |= 0x0Cu << (bit & 31);
    if ((bit & 31) > 26) w[(bit >> 5) + 1] |= 0x0Cu >> (32 - (bit & 31));
  }
  i32x8 v;
  for (int i = 0; i < 8; ++i) v[i] = (int)w[i];
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int sv = 127 + (r % 3) + 4 * h;  // 2^(r % 3) for K half 0, 2^(r % 3 + 4) for K half 1
  if (vary_second) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, v, c, 3, 3, 0, 127, 0, sv);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v, v, c, 3, 3, 0, sv, 0, 127);
  // C layout: lane (r, h) holds column r (second operand's row), rows (i & 3) + 8 (i >> 2) + 4 h (first operand's rows)
  for (int i = 0; i < 16; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

int main() {
  // ---- 1. conversions
  float hx[64 * 32];
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 32; ++j) {
      float v = 0.37f * (j + 1) * ((j & 1) ? -1.f : 1.f);            // 0.37 .. 11.8, alternating sign
      if (l == 1) v = ldexpf(1.f, j - 12);                            // powers of two 2^-12 .. 2^19: subnormals, saturation
      if (l == 2) v = (j < 16 ? 1.f : 2.f) + 0.125f * (j & 15);       // rounding: steps of 1/8 around 1 and 2 (ties at odd eighths)
      hx[l * 32 + j] = v;
    }
  float* dx; unsigned *d32, *d16;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&d32, 64 * 6 * 4); hipMalloc(&d16, 64 * 6 * 4);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, scale, d32, d16);
    unsigned h32[64 * 6], h16[64 * 6];
    hipMemcpy(h32, d32, sizeof(h32), hipMemcpyDeviceToHost);
    hipMemcpy(h16, d16, sizeof(h16), hipMemcpyDeviceToHost);
    for (int l = 0; l < 3; ++l) {
      printf("scale %g lane %d: x -> bf6(f32 src) / bf6(f16 src), decoded * scale\n", scale, l);
      for (int j = 0; j < 32; ++j)
        printf("  [%2d] %12.6g -> %9.5g / %9.5g%s", j, hx[l * 32 + j], e3m2_value(field(h32 + l * 6, j)) * scale,
               e3m2_value(field(h16 + l * 6, j)) * scale, (j & 1) ? "\n" : "");
    }
  }
  // ---- 2. per-lane MFMA scales
  float* dout; hipMalloc(&dout, 1024 * 4);
  for (int second = 0; second < 2; ++second) {
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dout, second);
    float ho[1024];
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    printf("mfma e3m2, per-lane scale on the %s operand: out[row][col] / 32 for rows 0..5, cols 0..5\n", second ? "second" : "first");
    for (int i = 0; i < 6; ++i) { for (int j = 0; j < 6; ++j) printf(" %7.1f", ho[i * 32 + j] / 32.f); printf("\n"); }
    printf("  expected if the byte of lane (r, h) scales row r of ITS operand over K half h: 2^(r%%3) + 2^(r%%3+4) = 17, 34, 68 along %s\n",
           second ? "columns" : "rows");
  }
  return 0;
}
