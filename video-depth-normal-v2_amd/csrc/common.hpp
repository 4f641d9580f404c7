// Shared device helpers for the gfx950 kernels (wave = 64 lanes, MFMA fragments per
// /opt/skills/guides/cdna_hip_programming.md §3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vdn.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define VDN_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return -(1000 + (int)e__);        \
  } while (0)

template <int DT> struct Half;
template <> struct Half<VDN_F16> {
  using T = _Float16;
  using V8 = f16x8;
  using V4 = f16x4;
  using V2 = f16x2;
  static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Half<VDN_BF16> {
  using T = __bf16;
  using V8 = bf16x8;
  using V4 = bf16x4;
  using V2 = bf16x2;
  static __device__ __forceinline__ f32x4 mfma16(V8 a, V8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// ReLU on 8 packed 16-bit floats held as 4 dwords, ONE instruction per dword: as signed 16-bit integers every value
// with the sign bit set is negative, so v_pk_max_i16(x, 0) clears exactly those (fp16 and bf16 alike; -0 -> +0, +NaN and
// +inf pass through like in torch.relu, a NaN with the sign bit becomes 0). The ReLU-on-load of the convolutions runs
// beside the MFMAs of the main loop: 4 VALU per dword (and / shift / multiply / and-not) cost the 256 x 256 kernel 0.7 us
// of its 2.7 us per K step.
template <typename V8>
__device__ __forceinline__ V8 relu8(V8 v) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(V8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

__device__ __forceinline__ float gelu_erf(float x) {  // nn.GELU() default (exact erf form)
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Exact-form GELU, 4 values at a time, fp32-level accuracy (|error| <= 2.5e-7 on [-8, 8], i.e. the
// rounding of the result itself; measured against float64 erf in tools/fit_gelu.py, which also produced
// the coefficients). With t = |x|/sqrt2:  0.5(1+erf(x/sqrt2)) = 1 - 0.5 erfc(t) for x >= 0, 0.5 erfc(t) for
// x < 0, so  gelu(x) = max(x,0) - 0.5|x| erfc(t)  with no cancellation, and erfc(t) = 2^(t Q(t)) where
// Q(t) = log2(erfc(t))/t is smooth: a degree-7 polynomial on [0, 4.3] (minimax in the error it causes in
// gelu); beyond 4.3 erfc < 2e-9 and t is clamped. One v_exp_f32 and 12 multiply-adds per value, written
// on vectors so the compiler emits packed fp32 ops (v_pk_fma_f32: two values per instruction) — the fc1
// epilogue is VALU-bound (tools/gemm_ablate.sh: 31 of 295 us per launch with the rcp+exp form before).
// A literal operand forces the scalar v_fmaak form; a constant the compiler cannot see through lives in
// an SGPR and lets the multiply-adds pair up as v_pk_fma_f32.
__device__ __forceinline__ float sgpr_const(float c) { asm("" : "+s"(c)); return c; }
__device__ __forceinline__ f32x4 gelu4(f32x4 x) {
  const float c7 = sgpr_const(-9.663539458415471e-06f), c6 = sgpr_const(0.00011715106666088104f),
              c5 = sgpr_const(-0.00028722305432893336f), c4 = sgpr_const(-0.0030229499097913504f),
              c3 = sgpr_const(0.030544215813279152f), c2 = sgpr_const(-0.14973200857639313f),
              c1 = sgpr_const(-0.9180853962898254f), c0 = sgpr_const(-1.62794029712677f),
              kh = sgpr_const(0.70710678118654752440f);
  f32x4 t0, t, q, e, r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    t0[i] = fabsf(x[i]) * 0.70710678118654752440f;
    t[i] = __builtin_amdgcn_fmed3f(t0[i], 0.f, 4.3f);
    r[i] = __builtin_amdgcn_fmed3f(x[i], 0.f, 3.0e38f);
  }
  q = t * c7 + c6;
  q = q * t + c5;
  q = q * t + c4;
  q = q * t + c3;
  q = q * t + c2;
  q = q * t + c1;
  q = q * t + c0;
  q = q * t;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(q[i]);
  return r - (t0 * kh) * e;  // 0.5|x| = t0/sqrt2
}
__device__ __forceinline__ float gelu_fast(float x) {
  const f32x4 r = gelu4(f32x4{x, x, x, x});
  return r[0];
}

// 4 consecutive values of a residual / table operand as floats (16-byte or 8-byte vector load)
__device__ __forceinline__ f32x4 load4_as_float(const void* p, int dt, size_t i) {
  if (dt == VDN_F32) return *(const f32x4*)((const float*)p + i);
  if (dt == VDN_F16) {
    const f16x4 h = *(const f16x4*)((const _Float16*)p + i);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  }
  const bf16x4 h = *(const bf16x4*)((const __bf16*)p + i);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

__device__ __forceinline__ float load_as_float(const void* p, int dt, size_t i) {
  if (dt == VDN_F32) return ((const float*)p)[i];
  if (dt == VDN_F16) return (float)((const _Float16*)p)[i];
  return (float)((const __bf16*)p)[i];
}

__device__ __forceinline__ void store_from_float(void* p, int dt, size_t i, float v) {
  if (dt == VDN_F32) ((float*)p)[i] = v;
  else if (dt == VDN_F16) ((_Float16*)p)[i] = (_Float16)v;
  else ((__bf16*)p)[i] = (__bf16)v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Barrier closing a pipeline stage that (a) other waves will overwrite by LDS-DMA and (b) whose DMA for
// the next stage this wave has issued: the wave's own LDS reads AND its LDS-DMA must have completed
// before it arrives (hipcc alone may leave ds_reads in flight across the barrier: a WAR race against
// the next stage's DMA, and `s_barrier` itself waits for no counter).
__device__ __forceinline__ void stage_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Split a float into (hi, lo) 16-bit planes: hi = x rounded TOWARD ZERO, lo = nearest(x - hi).
// hi and lo share their sign (or lo == 0), so sign-based ops (ReLU) act per plane.
__device__ __forceinline__ void split_rtz(float x, _Float16& hi, _Float16& lo) {
  const f16x2 p = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x, 0.f));
  hi = p[0];
#ifdef VDN_SPLIT_CLASSIC
  lo = (_Float16)(x - (float)hi);
#else
  uint32_t lp;  // fp16(x - hi) by one mixed-precision FMA (see split2_rtz)
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(__builtin_bit_cast(uint32_t, p)), "v"(x));
  lo = __builtin_bit_cast(f16x2, lp)[0];
#endif
}
__device__ __forceinline__ void split_rtz(float x, __bf16& hi, __bf16& lo) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u;  // truncate mantissa
  const float h = __builtin_bit_cast(float, u);
  hi = __builtin_bit_cast(__bf16, (uint16_t)(u >> 16));
  lo = (__bf16)(x - h);
}
// two values at once: one packed RTZ convert for the hi pair, a packed fp32 subtract for the remainders
// (the VALU cost per value drops from ~5.5 to ~3 instructions: the attention softmax splits every P entry)
__device__ __forceinline__ void split2_rtz(float x0, float x1, _Float16& h0, _Float16& h1, _Float16& l0, _Float16& l1) {
  const f16x2 p = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  h0 = p[0];
  h1 = p[1];
  // lo = fp16(x - hi) in ONE instruction per value: the mixed-precision FMA reads the fp16 hi half directly
  // (no v_cvt_f32_f16), multiplies by -1, adds the fp32 x and rounds the result to fp16 (3 instead of 5 VALU
  // instructions per pair; the compiler's own choice was cvt + cvt + pk_add + cvt_pk)
#ifdef VDN_SPLIT_CLASSIC  // A/B builds only (tools/build_variant.sh)
  const f32x2 d = f32x2{x0, x1} - f32x2{(float)p[0], (float)p[1]};
  l0 = (_Float16)d[0];
  l1 = (_Float16)d[1];
#else
  const uint32_t hp = __builtin_bit_cast(uint32_t, p);
  uint32_t lp;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(x1));
  const f16x2 q = __builtin_bit_cast(f16x2, lp);
  l0 = q[0];
  l1 = q[1];
#endif
}
__device__ __forceinline__ void split2_rtz(float x0, float x1, __bf16& h0, __bf16& h1, __bf16& l0, __bf16& l1) {
  split_rtz(x0, h0, l0);
  split_rtz(x1, h1, l1);
}
// 8-bit planes of a split value for the attention cross terms (include/vdn.h: vdn_gemm_desc.dst8): e5m2 of the value
// and e5m2 of its fp16 remainder scaled by 2^10 (e5m2 shares fp16's exponent range, so neither needs a block scale;
// the attention kernel undoes the 2^10 with the MFMA's E8M0 scale operand).
__device__ __forceinline__ uint32_t pk4_bf8(float a, float b, float c, float d) {
  int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
  return (uint32_t)v;
}
constexpr float VDN_LO8_SCALE = 1024.0f;  // 2^10; E8M0 byte 127 - 10 on the other side
constexpr int VDN_LO8_E8M0 = 117;

// ---- 6-bit block-scaled operand rows of the cross-term GEMM (include/vdn.h: vdn_gemm_desc.A8 / W8 / out8, "x6 rows").
// Per row and 64 of K one 64-byte row-slab = two HALVES of 32 bytes, one per K half of the 32x32x64 MFMA:
//   [24 B: 32 e3m2 codes (sign, 3 exponent bits bias 3, 2 mantissa bits; 6-bit fields, little-endian bit stream)]
//   [ 1 B: E8M0 scale byte s of the half: value = code * 2^(s - 127)] [7 B unused]
// so that the two 16-byte LDS reads of a lane ARE its MFMA operand (registers 0..5) and its scale operand (register 6).
// Scale of a half: 2^(floor(log2 max|hi|) - 4), the largest hi lands in [16, 32) (e3m2 tops out at 28: the few values
// in (30, 32) saturate, 12.5 % at worst, the rounding error of the format). The remainder plane of the same half uses
// that scale - 10: |lo| < ulp(hi) <= 2^-10 of the largest hi, so it lands below 16.
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
__device__ __forceinline__ int x6_scale_byte(const f16x32& hi) {
  typedef unsigned short u16x32 __attribute__((ext_vector_type(32)));
  typedef unsigned short u16x16 __attribute__((ext_vector_type(16)));
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
  typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x32 b = __builtin_bit_cast(u16x32, hi) & (unsigned short)0x7FFF;   // magnitudes: fp16 patterns order like integers
  const u16x16 m16 = __builtin_elementwise_max(b.lo, b.hi);
  const u16x8 m8 = __builtin_elementwise_max(m16.lo, m16.hi);
  const u16x4 m4 = __builtin_elementwise_max(m8.lo, m8.hi);
  const u16x2 m2 = __builtin_elementwise_max(m4.lo, m4.hi);
  const unsigned m = m2[0] > m2[1] ? m2[0] : m2[1];
  return (int)(m >> 10) + 108;   // fp16 exponent field e: 2^(e - 15 - 4) -> E8M0 byte 127 + e - 19
}
// one half (32 fp16 values in stream order) -> 32 bytes at dst (16-byte aligned)
__device__ __forceinline__ void x6_store_half(uint8_t* dst, const f16x32& v, int scale_byte) {
  const u32x6 c = __builtin_amdgcn_cvt_scalef32_pk32_bf6_f16(v, __builtin_bit_cast(float, (unsigned)scale_byte << 23));
  *(u32x4*)dst = u32x4{c[0], c[1], c[2], c[3]};
  *(u32x4*)(dst + 16) = u32x4{c[4], c[5], (unsigned)scale_byte, 0u};
}
// Column of a 64-wide K slab that sits at stream position p (0..31) of half h, by the ORDER the producer of the activation
// planes writes (the weight planes of the consuming GEMM are packed in the same order, vdn_pack_x8):
//   0 natural        col = 32 h + p                                  vdn_layernorm, vdn_pack_x8 on activations
//   1 GEMM epilogue  col = 32 (p >> 4) + 16 ((p >> 3) & 1) + 8 h + (p & 7)     the bias + GELU flavour of the 8-bit kernel
//   2 attention      col = 32 (p >> 4) + 8 ((p >> 2) & 3) + 4 h + (p & 3)      vdn_flash_attn
__host__ __device__ __forceinline__ int x6_col(int order, int h, int p) {
  return order == 1 ? 32 * (p >> 4) + 16 * ((p >> 3) & 1) + 8 * h + (p & 7)
       : order == 2 ? 32 * (p >> 4) + 8 * ((p >> 2) & 3) + 4 * h + (p & 3)
                    : 32 * h + p;
}

// store 1 value: hi-only (round to nearest) or split planes
template <typename T>
__device__ __forceinline__ void store_half(T* hi, T* lo, size_t i, float v) {
  if (lo) { T a, b; split_rtz(v, a, b); hi[i] = a; lo[i] = b; }
  else hi[i] = (T)v;
}
// hi rounded to NEAREST, lo = the remainder (hi + lo as exact as the toward-zero split; the planes may differ in sign)
template <typename T>
__device__ __forceinline__ void store_half_nearest(T* hi, T* lo, size_t i, float v) {
  const T a = (T)v;
  hi[i] = a;
  if (lo) lo[i] = (T)(v - (float)a);
}
template <typename T>
__device__ __forceinline__ float load_half(const T* hi, const T* lo, size_t i) {
  return lo ? (float)hi[i] + (float)lo[i] : (float)hi[i];
}

// XCD-aware bijective block remap (guide §5 'XCD swizzle must be bijective'): blocks that share
// blockIdx % 8 share an XCD/L2, so give each XCD a contiguous run of the logical tile order.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int xcd = orig & 7;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}
