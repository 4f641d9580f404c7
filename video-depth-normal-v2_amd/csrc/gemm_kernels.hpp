// GEMM / implicit-GEMM convolution for gfx950:  out = epilogue(A[M,K] x W[N,K]^T)
//
// * half (fp16|bf16) operands, fp32 accumulate on v_mfma_f32_16x16x32_{f16,bf16}
// * BM x BN x 64 tiles, 4 waves (WM x WN), both operands staged global -> LDS by LDS-DMA
//   (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave-instruction) into a double buffer;
//   the LDS image is lane-linear, the bank-conflict swizzle (16-B chunk ^ ((row>>1)&7) on 128-B
//   rows) is applied on the per-lane SOURCE address and again on the ds_read_b128 address
//   (guide §5.4 rule 21)
// * the A row can be a plain row-major row or an on-the-fly 3x3 (stride 1|2, pad 1) NHWC gather:
//   the per-lane source pointer makes the DMA itself the im2col; padding taps read a zero page
// * everything the reference does around its Linear/Conv (bias, GELU/ReLU, LayerScale, pos-embed,
//   residual adds, head split + RoPE, pixel-shuffle for ConvTranspose, GEGLU) happens on the
//   fp32 accumulators before the single store.
//
// Roofline: MFMA-bound (>= 170 flop per HBM byte on every shape of the path).
#pragma once
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace vdn_gemm_impl {

// Kernel-selection knobs of a launch (include/vdn.h: vdn_gemm_tuning): the descriptor's own, or the defaults read from
// VDN_GEMM_* / VDN_SPLITK_* ONCE at the first use (immutable afterwards: no process-wide mutable state).
const vdn_gemm_tuning& tuning(const vdn_gemm_desc& d);  // gemm.hip

constexpr int BK = 64;

// ---------------------------------------------------------------------------------------------
// Fused fp32 epilogue. The MFMAs are issued with the WEIGHT fragment as the A operand and the
// activation fragment as B, so an accumulator tile is C^T:
//   acc[i][j][e]  ->  C[m = mw + 16 i + (lane & 15)][n = nw + 16 j + 4 (lane >> 4) + e]
// i.e. a lane always holds 4 CONSECUTIVE output columns of one row. emit4() finishes one such group:
// bias / row-add / activation / LayerScale / pos-embed table / residuals as single 16-byte loads,
// then one 8- or 16-byte store per plane in the layout the consumer reads (plain rows, per-head
// Q/K/V^T with RoPE, pixel-shuffle, GEGLU). `b` is the group 16 columns to the right (the GEGLU gate /
// the RoPE imaginary parts, which the weight packing put there).
// `bias_a` / `bias_b` / `gam` are the bias of this group, the bias of the group 16 columns to the right and the
// LayerScale factor, loaded ONCE per wave before the first store (0 / 0 / 1 when the operand is absent): a
// load inside the store loop waits on vmcnt, which on gfx9 also counts the stores issued before it, so
// every group paid a full store round trip (measured: 10 us of "math" per tile round that was latency).
// Implicit-GEMM 3x3 gather, fast path for the (ci/64, tap, ...) K order: per 16-row piece a pointer to the centre
// tap (plus this lane's 16-byte chunk) and a 9-bit "tap lies inside the image" mask are computed once; a K step
// then costs a wave-uniform offset, one bit test and one 64-bit add per piece instead of ~25 VALU of index math.
struct ConvTap {
  const void* center;
  unsigned ok9;
};
template <typename T>
__device__ __forceinline__ ConvTap conv_tap_setup(const T* image, int iy0, int ix0, int chunk_elems, const vdn_gemm_desc& p) {
  ConvTap c;
  c.center = image + ((ptrdiff_t)(iy0 + 1) * p.cW + (ix0 + 1)) * p.cC + chunk_elems;
  c.ok9 = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = iy0 + t / 3, ix = ix0 + t % 3;
    c.ok9 |= (unsigned)((iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW)) << t;
  }
  return c;
}
// element offset of tap `tap` (0..8, wave-uniform) relative to the centre
__device__ __forceinline__ int conv_tap_offset(int tap, const vdn_gemm_desc& p) {
  const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
  return ((ky - 1) * p.cW + (kx - 1)) * p.cC;
}

// internal store codes (never in a descriptor): specialised epilogues, see emit4 / epi_flavour
constexpr int VDN_STX_FC1 = 100, VDN_STX_RES = 101, VDN_STX_HEADS = 102, VDN_STX_HALF = 103, VDN_STX_RESHALF1 = 104,
              VDN_STX_RESHALF2 = 105, VDN_STX_SPLITK = 106;

// Which straight-line flavour (if any) computes exactly what descriptor `d` asks for.
__host__ __device__ inline int epi_flavour(const vdn_gemm_desc& d) {
  const bool half_out = d.out_dt != VDN_F32;
  const bool plain_rows = d.store == VDN_ST_PLAIN && !d.rowadd && !d.gamma && !d.tab && d.row_group <= 0;
  if (plain_rows && half_out && d.out_lo && d.act != VDN_ACT_GELU) {
    if (!d.res1 && !d.res2) return VDN_STX_HALF;  // [bias] [relu] -> split half planes (1x1 / 3x3 convolutions)
    if (d.act == VDN_ACT_NONE && d.res1 && d.res1_lo && d.res1_dt == d.out_dt) {
      if (!d.res2) return VDN_STX_RESHALF1;
      if (d.res2_lo && d.res2_dt == d.out_dt) return VDN_STX_RESHALF2;
    }
  }
  if (d.store == VDN_ST_PLAIN && d.bias && d.act == VDN_ACT_GELU && !d.rowadd && !d.gamma && !d.tab && !d.res1 && !d.res2 &&
      half_out && (d.out_lo || (d.out8 && d.A8 && d.W8)) && d.row_group <= 0)   // 8-bit kernel: the fp16 lo plane is optional next to out8
    return VDN_STX_FC1;
  if (d.store == VDN_ST_PLAIN && d.act == VDN_ACT_NONE && !d.rowadd && !d.tab && d.res1 && d.res1_dt == VDN_F32 && !d.res1_lo &&
      !d.res2 && d.out_dt == VDN_F32 && d.row_group <= 0)
    return VDN_STX_RES;
  if (d.store == VDN_ST_HEADS && !d.rope[0] && !d.rope[1] && !d.rope[2] && d.nsplit >= 1) {
    for (int i = 0; i < d.nsplit; ++i)
      if (!d.dst[i] || (!d.dst_lo[i] && !d.transposed[i] && !d.dst8[i])) return d.store;   // V^T, and Q / K with 8-bit planes, may come without lo
    return VDN_STX_HEADS;
  }
  return d.store;
}

template <int DT, int STORE>
__device__ __forceinline__ void emit4(const vdn_gemm_desc& p, int m, int n, f32x4 a, f32x4 b, f32x4 bias_a, f32x4 bias_b,
                                      f32x4 gam) {
  using H = Half<DT>;
  using T = typename H::T;
  if (m >= p.M || n >= p.N) return;
  // ---- straight-line flavours of the hot epilogues (chosen on the host by epi_flavour()): the generic code
  // below tests ~40 wave-uniform descriptor fields per group, and with 32 groups per lane and only two waves
  // per SIMD those scalar branches were 12 us of every 31 us tile round at K = 32 (tools/gemm_ablate.sh).
  if constexpr (STORE == VDN_STX_SPLITK) {  // raw partial sums of one K slice (p.out / p.ldc were redirected to the slice)
    *(f32x4*)((float*)p.out + (size_t)m * p.ldc + n) = a;
    return;
  } else if constexpr (STORE == VDN_STX_FC1) {  // bias + GELU -> split half planes, plain rows
    a = gelu4(a + bias_a);
    // out_kt: the planes are written K-tile-major ([N/32][M][32] halves, [N/64][M][64] bytes) for the next 8-bit GEMM
    const size_t o = p.out_kt ? ((size_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (size_t)m * p.ldc + n;
    typename H::V4 h, l;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      T h0, h1, l0, l1;
      split2_rtz(a[e], a[e + 1], h0, h1, l0, l1);
      h[e] = h0; h[e + 1] = h1; l[e] = l0; l[e + 1] = l1;
    }
    *(typename H::V4*)((T*)p.out + o) = h;
    if (p.out_lo) *(typename H::V4*)((T*)p.out_lo + o) = l;
    return;
  } else if constexpr (STORE == VDN_STX_HALF || STORE == VDN_STX_RESHALF1 || STORE == VDN_STX_RESHALF2) {
    // [bias] [relu] [+ split-half residual(s)] -> split half planes, plain rows (the DPT head's convolutions)
    a += bias_a;
    if constexpr (STORE == VDN_STX_HALF) {
      const float floor_v = p.act == VDN_ACT_RELU ? 0.f : -INFINITY;  // branch-free optional ReLU
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = (a[e] < floor_v) ? floor_v : a[e];  // NaN passes through (fmaxf would hide it)
    } else {
      const size_t r1 = (size_t)m * p.ldr1 + n;
      const typename H::V4 rh = *(const typename H::V4*)((const T*)p.res1 + r1), rl = *(const typename H::V4*)((const T*)p.res1_lo + r1);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] += (float)rh[e] + (float)rl[e];
      if constexpr (STORE == VDN_STX_RESHALF2) {
        const size_t r2 = (size_t)m * p.ldr2 + n;
        const typename H::V4 qh = *(const typename H::V4*)((const T*)p.res2 + r2), ql = *(const typename H::V4*)((const T*)p.res2_lo + r2);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += (float)qh[e] + (float)ql[e];
      }
    }
    const size_t o = (size_t)m * p.ldc + n;
    typename H::V4 h, l;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      T h0, h1, l0, l1;
      split2_rtz(a[e], a[e + 1], h0, h1, l0, l1);
      h[e] = h0; h[e + 1] = h1; l[e] = l0; l[e + 1] = l1;
    }
    *(typename H::V4*)((T*)p.out + o) = h;
    *(typename H::V4*)((T*)p.out_lo + o) = l;
    return;
  } else if constexpr (STORE == VDN_STX_RES) {  // (acc + bias) * gamma + f32 residual -> f32 rows
    a = (a + bias_a) * gam + *(const f32x4*)((const float*)p.res1 + (size_t)m * p.ldr1 + n);
    *(f32x4*)((float*)p.out + (size_t)m * p.ldc + n) = a;
    return;
  } else if constexpr (STORE == VDN_STX_HEADS) {  // bias -> per-head Q / K rows or V^T columns, split planes, no RoPE
    const int hc = p.heads * 64;
    const int bt = m / p.tokens, tl = m - bt * p.tokens;
    const int tk = tl + p.tok_off;
    const int split = n / hc;
    const int head = (n - split * hc) >> 6, e0 = n & 63;
    const size_t hb = (size_t)bt * p.heads + head;
    T* dst = (T*)p.dst[split];
    T* dlo = (T*)p.dst_lo[split];
    a += bias_a;
    typename H::V4 h, l;
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      T h0, h1, l0, l1;
      split2_rtz(a[e], a[e + 1], h0, h1, l0, l1);
      h[e] = h0; h[e + 1] = h1; l[e] = l0; l[e + 1] = l1;
    }
    if (p.transposed[split]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t o = (hb * 64 + e0 + e) * p.tpad + tk;
        store_half_nearest(dst, dlo, o, a[e]);  // V^T: hi to nearest (the one-product P V reads it alone), lo = remainder
      }
    } else {
      const size_t o = (hb * p.tpad + tk) * 64 + e0;
      *(typename H::V4*)(dst + o) = h;
      if (dlo) *(typename H::V4*)(dlo + o) = l;   // with 8-bit planes the fp16 lo plane is optional: the attention reads e5m2(lo 2^10) instead
      if (p.dst8[split]) {
        uint8_t* d8 = (uint8_t*)p.dst8[split] + (hb * p.tpad + tk) * 128 + e0;
        const float k = VDN_LO8_SCALE;
        *(uint32_t*)d8 = pk4_bf8(a[0], a[1], a[2], a[3]);
        *(uint32_t*)(d8 + 64) = pk4_bf8(k * (float)l[0], k * (float)l[1], k * (float)l[2], k * (float)l[3]);
      }
    }
    return;
  }
  if constexpr (STORE == VDN_ST_PLAIN || STORE == VDN_ST_CONVT) {
    a += bias_a;
    if (p.rowadd) a += p.rowadd[m];
    if (p.act == VDN_ACT_GELU) {
      a = gelu4(a);
    } else if (p.act == VDN_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] = (a[e] < 0.f) ? 0.f : a[e];
    }
    a *= gam;
    if (p.tab) a += *(const f32x4*)(p.tab + (size_t)(m % p.tab_mod + p.tab_off) * p.N + n);
    if (p.res1) {
      a += load4_as_float(p.res1, p.res1_dt, (size_t)m * p.ldr1 + n);
      if (p.res1_lo) a += load4_as_float(p.res1_lo, p.res1_dt, (size_t)m * p.ldr1 + n);
    }
    if (p.res2) {
      a += load4_as_float(p.res2, p.res2_dt, (size_t)m * p.ldr2 + n);
      if (p.res2_lo) a += load4_as_float(p.res2_lo, p.res2_dt, (size_t)m * p.ldr2 + n);
    }
    size_t o;
    if constexpr (STORE == VDN_ST_CONVT) {
      const int hw = p.cH * p.cW;
      const int cb = m / hw, rem = m - cb * hw;
      const int cy = rem / p.cW, cx = rem - cy * p.cW;
      const int kk = n / p.cout, co = n - kk * p.cout;  // cout % 4 == 0: the 4 columns share a tap
      const int ky = kk / p.ck, kx = kk - ky * p.ck;
      o = ((((size_t)cb * (p.cH * p.ck) + cy * p.ck + ky) * (p.cW * p.ck)) + cx * p.ck + kx) * p.cout + co;
    } else {
      o = (size_t)(p.row_group > 0 ? m + (m / p.row_group + 1) * p.row_skip : m) * p.ldc + n;
    }
    if (p.out_dt == VDN_F32) {
      *(f32x4*)((float*)p.out + o) = a;
    } else if (p.out_lo) {
      typename H::V4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { T x0, x1; split_rtz(a[e], x0, x1); h[e] = x0; l[e] = x1; }
      *(typename H::V4*)((T*)p.out + o) = h;
      *(typename H::V4*)((T*)p.out_lo + o) = l;
    } else {
      typename H::V4 h = {(T)a[0], (T)a[1], (T)a[2], (T)a[3]};
      *(typename H::V4*)((T*)p.out + o) = h;
    }
  } else if constexpr (STORE == VDN_ST_GEGLU) {
    // packed columns: 16-wide blocks alternate [h | gate]
    if ((n & 16) || n + 16 >= p.N) return;
    a += bias_a;
    b += bias_b;
    if (p.act == VDN_ACT_SILU) {  // SwiGLU (ViT-g FFN): gate * sigmoid(gate), sigmoid through one exp2 and one reciprocal
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= b[e] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * b[e]));
    } else {
      a = a * gelu4(b);
    }
    const size_t o = (size_t)m * p.ldc + ((n >> 5) << 4) + (n & 15);
    if (p.out_dt == VDN_F32) {
      *(f32x4*)((float*)p.out + o) = a;
    } else if (p.out_lo) {
      typename H::V4 h, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { T x0, x1; split_rtz(a[e], x0, x1); h[e] = x0; l[e] = x1; }
      *(typename H::V4*)((T*)p.out + o) = h;
      *(typename H::V4*)((T*)p.out_lo + o) = l;
    } else {
      typename H::V4 h = {(T)a[0], (T)a[1], (T)a[2], (T)a[3]};
      *(typename H::V4*)((T*)p.out + o) = h;
    }
  } else {  // VDN_ST_HEADS
    const int hc = p.heads * 64;
    const int bt = m / p.tokens, tl = m - bt * p.tokens;
    const int tk = tl + p.tok_off;
    const int split = n / hc;
    const int head = (n - split * hc) >> 6, e0 = n & 63;
    const size_t hb = (size_t)bt * p.heads + head;
    T* dst = (T*)p.dst[split];
    T* dlo = (T*)p.dst_lo[split];
    a += bias_a;
    if (p.rope[split]) {
      if (e0 & 16) return;  // imaginary group: consumed together with its real partner
      b += bias_b;
      const int pi = ((e0 >> 5) << 4) + (e0 & 15);  // first of 4 consecutive pair indices
      const float* cs = p.rope_cs + (size_t)(tl % p.rope_mod) * 64 + 2 * pi;
      float o8[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float cc = cs[2 * e], ss = cs[2 * e + 1];
        o8[2 * e] = a[e] * cc - b[e] * ss;
        o8[2 * e + 1] = a[e] * ss + b[e] * cc;
      }
      if (p.transposed[split]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) store_half_nearest(dst, dlo, (hb * 64 + 2 * pi + e) * p.tpad + tk, o8[e]);
      } else {
        const size_t o = (hb * p.tpad + tk) * 64 + 2 * pi;
        typename H::V8 h8, l8;
        const bool sp = dlo || p.dst8[split];   // split planes: hi toward zero + remainder (fp16 plane and / or the 8-bit one)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (sp) { T x0, x1; split_rtz(o8[e], x0, x1); h8[e] = x0; l8[e] = x1; }
          else h8[e] = (T)o8[e];
        }
        *(typename H::V8*)(dst + o) = h8;
        if (dlo) *(typename H::V8*)(dlo + o) = l8;
        if (p.dst8[split]) {
          uint8_t* d8 = (uint8_t*)p.dst8[split] + (hb * p.tpad + tk) * 128 + 2 * pi;
          const float k = VDN_LO8_SCALE;
          *(u32x2*)d8 = u32x2{pk4_bf8(o8[0], o8[1], o8[2], o8[3]), pk4_bf8(o8[4], o8[5], o8[6], o8[7])};
          *(u32x2*)(d8 + 64) = u32x2{pk4_bf8(k * (float)l8[0], k * (float)l8[1], k * (float)l8[2], k * (float)l8[3]),
                                     pk4_bf8(k * (float)l8[4], k * (float)l8[5], k * (float)l8[6], k * (float)l8[7])};
        }
      }
      return;
    }
    if (p.transposed[split]) {
#pragma unroll
      for (int e = 0; e < 4; ++e) store_half_nearest(dst, dlo, (hb * 64 + e0 + e) * p.tpad + tk, a[e]);
    } else {
      const size_t o = (hb * p.tpad + tk) * 64 + e0;
      typename H::V4 h, l;
      const bool sp = dlo || p.dst8[split];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (sp) { T x0, x1; split_rtz(a[e], x0, x1); h[e] = x0; l[e] = x1; }
        else h[e] = (T)a[e];
      }
      *(typename H::V4*)(dst + o) = h;
      if (dlo) *(typename H::V4*)(dlo + o) = l;
      if (p.dst8[split]) {
        uint8_t* d8 = (uint8_t*)p.dst8[split] + (hb * p.tpad + tk) * 128 + e0;
        const float k = VDN_LO8_SCALE;
        *(uint32_t*)d8 = pk4_bf8(a[0], a[1], a[2], a[3]);
        *(uint32_t*)(d8 + 64) = pk4_bf8(k * (float)l[0], k * (float)l[1], k * (float)l[2], k * (float)l[3]);
      }
    }
  }
}

// Epilogue straight from registers (4-wave kernels): a store instruction covers 16 rows x 64 (f32) or
// 32 (half) bytes.
// Flavours whose outputs are 16-bit planes store 8 consecutive columns per lane (one 16-byte store per plane, half
// the store instructions, 64-byte row segments): the 8-wave kernels then load W rows into LDS in the order
// pair8_col() so that a lane's accumulators j = 2u, 2u+1 hold columns 32u + 8 fq + {0..3}, {4..7} of the wave's
// 64-column slab (instead of 16j + 4 fq + {0..3}).
template <int STORE>
constexpr bool vdn_pair8 = (STORE == VDN_STX_FC1 || STORE == VDN_STX_HALF || STORE == VDN_STX_RESHALF1 ||
                            STORE == VDN_STX_RESHALF2 || STORE == VDN_STX_HEADS);
// W row (within a 64-row slab) held by LDS row 16 t + r under the paired mapping
__device__ __forceinline__ int pair8_col(int t, int r) { return (t >> 1) * 32 + (r >> 2) * 8 + (t & 1) * 4 + (r & 3); }

// 8 consecutive columns n..n+7 of row m (a0: n..n+3, a1: n+4..n+7) for the plane-output flavours
template <int DT, int STORE>
__device__ __forceinline__ void emit8(const vdn_gemm_desc& p, int m, int n, f32x4 a0, f32x4 a1, f32x4 b0, f32x4 b1) {
  using H = Half<DT>;
  using T = typename H::T;
  using V8 = typename H::V8;
  if (m >= p.M || n >= p.N) return;
  float a[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = a0[e] + b0[e]; a[4 + e] = a1[e] + b1[e]; }
  if constexpr (STORE == VDN_STX_FC1) {
    const f32x4 g0 = gelu4(f32x4{a[0], a[1], a[2], a[3]}), g1 = gelu4(f32x4{a[4], a[5], a[6], a[7]});
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = g0[e]; a[4 + e] = g1[e]; }
  } else if constexpr (STORE == VDN_STX_HALF) {
    const float floor_v = p.act == VDN_ACT_RELU ? 0.f : -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = (a[e] < floor_v) ? floor_v : a[e];  // NaN passes through
  } else if constexpr (STORE == VDN_STX_RESHALF1 || STORE == VDN_STX_RESHALF2) {
    const size_t r1 = (size_t)m * p.ldr1 + n;
    const V8 rh = *(const V8*)((const T*)p.res1 + r1), rl = *(const V8*)((const T*)p.res1_lo + r1);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += (float)rh[e] + (float)rl[e];
    if constexpr (STORE == VDN_STX_RESHALF2) {
      const size_t r2 = (size_t)m * p.ldr2 + n;
      const V8 qh = *(const V8*)((const T*)p.res2 + r2), ql = *(const V8*)((const T*)p.res2_lo + r2);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += (float)qh[e] + (float)ql[e];
    }
  }
  V8 h, l;
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    T h0, h1, l0, l1;
    split2_rtz(a[e], a[e + 1], h0, h1, l0, l1);
    h[e] = h0; h[e + 1] = h1; l[e] = l0; l[e + 1] = l1;
  }
  if constexpr (STORE == VDN_STX_HEADS) {
    const int hc = p.heads * 64;
    const int bt = m / p.tokens, tl = m - bt * p.tokens;
    const int tk = tl + p.tok_off;
    const int split = n / hc;
    const int head = (n - split * hc) >> 6, e0 = n & 63;
    const size_t hb = (size_t)bt * p.heads + head;
    T* dst = (T*)p.dst[split];
    T* dlo = (T*)p.dst_lo[split];
    if (p.transposed[split]) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const size_t o = (hb * 64 + e0 + e) * p.tpad + tk;
        store_half_nearest(dst, dlo, o, a[e]);  // V^T: hi to nearest (the one-product P V reads it alone), lo = remainder
      }
    } else {
      const size_t o = (hb * p.tpad + tk) * 64 + e0;
      *(V8*)(dst + o) = h;
      if (dlo) *(V8*)(dlo + o) = l;
      if (p.dst8[split]) {  // e5m2 planes for the attention cross terms
        uint8_t* d8 = (uint8_t*)p.dst8[split] + (hb * p.tpad + tk) * 128 + e0;
        *(u32x2*)d8 = u32x2{pk4_bf8(a[0], a[1], a[2], a[3]), pk4_bf8(a[4], a[5], a[6], a[7])};
        const float k = VDN_LO8_SCALE;
        *(u32x2*)(d8 + 64) = u32x2{pk4_bf8(k * (float)l[0], k * (float)l[1], k * (float)l[2], k * (float)l[3]),
                                   pk4_bf8(k * (float)l[4], k * (float)l[5], k * (float)l[6], k * (float)l[7])};
      }
    }
  } else {
    // out_kt (8-bit cross-term kernel): planes written K-tile-major for the next GEMM ([N/32][M][32] halves, [N/64][M][64] bytes)
    const size_t o = p.out_kt ? ((size_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (size_t)m * p.ldc + n;
    *(V8*)((T*)p.out + o) = h;
    if (p.out_lo) *(V8*)((T*)p.out_lo + o) = l;
  }
}

template <int DT, int TM, int TN, int STORE, bool PAIR = false>
__device__ __forceinline__ void epilogue_regs(f32x4 (&acc)[TM][TN], const vdn_gemm_desc& p, int mw, int nw, int lane) {
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 bias4[TN], gam4[TN];
  if constexpr (PAIR) {
    static_assert(TN == 4, "paired columns: 64-column wave slabs");
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + (j >> 1) * 32 + fq * 8 + (j & 1) * 4;
      bias4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int u = 0; u < TN / 2; ++u)
        emit8<DT, STORE>(p, mw + i * 16 + fr, nw + u * 32 + fq * 8, acc[i][2 * u], acc[i][2 * u + 1], bias4[2 * u], bias4[2 * u + 1]);
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nw + j * 16 + fq * 4;
    bias4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    gam4[j] = (p.gamma && n < p.N) ? *(const f32x4*)(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
  }
  if constexpr (STORE == VDN_STX_RES) {
    // residual rows are fetched one 16-row slab AHEAD of the stores: a load issued after a store waits for
    // that store's acknowledgement (vmcnt counts both on gfx9), which serialised every group on a round trip
    f32x4 r[2][TN];
    auto fetch = [&](int i, f32x4 (&dst)[TN]) {
      const int m = mw + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 16 + fq * 4;
        dst[j] = (m < p.M && n < p.N) ? *(const f32x4*)((const float*)p.res1 + (size_t)m * p.ldr1 + n) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    fetch(0, r[0]);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i + 1 < TM) fetch(i + 1, r[(i + 1) & 1]);
      const int m = mw + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = nw + j * 16 + fq * 4;
        if (m < p.M && n < p.N) *(f32x4*)((float*)p.out + (size_t)m * p.ldc + n) = (acc[i][j] + bias4[j]) * gam4[j] + r[i & 1][j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      emit4<DT, STORE>(p, mw + i * 16 + fr, nw + j * 16 + fq * 4, acc[i][j], acc[i][j + 1 < TN ? j + 1 : j], bias4[j],
                       bias4[j + 1 < TN ? j + 1 : j], gam4[j]);
}

template <int DT, int TM, int TN>
__device__ __forceinline__ void epilogue_dispatch(f32x4 (&acc)[TM][TN], const vdn_gemm_desc& p, int mw, int nw, int lane) {
  switch (epi_flavour(p)) {
    case VDN_STX_HALF: epilogue_regs<DT, TM, TN, VDN_STX_HALF>(acc, p, mw, nw, lane); break;
    case VDN_STX_RESHALF1: epilogue_regs<DT, TM, TN, VDN_STX_RESHALF1>(acc, p, mw, nw, lane); break;
    case VDN_STX_RESHALF2: epilogue_regs<DT, TM, TN, VDN_STX_RESHALF2>(acc, p, mw, nw, lane); break;
    case VDN_STX_RES: epilogue_regs<DT, TM, TN, VDN_STX_RES>(acc, p, mw, nw, lane); break;
    case VDN_STX_FC1:
    case VDN_ST_PLAIN: epilogue_regs<DT, TM, TN, VDN_ST_PLAIN>(acc, p, mw, nw, lane); break;
    case VDN_ST_CONVT: epilogue_regs<DT, TM, TN, VDN_ST_CONVT>(acc, p, mw, nw, lane); break;
    case VDN_ST_GEGLU: epilogue_regs<DT, TM, TN, VDN_ST_GEGLU>(acc, p, mw, nw, lane); break;
    default: epilogue_regs<DT, TM, TN, VDN_ST_HEADS>(acc, p, mw, nw, lane); break;  // VDN_ST_HEADS, VDN_STX_HEADS
  }
}

template <int DT, int BM, int BN, int WM, int WN, int AMODE /*0 plain,1 conv,2 conv+relu,3 plain+relu*/>
__global__ __launch_bounds__(256) void gemm_kernel(const vdn_gemm_desc p) {
  using H = Half<DT>;
  using V8 = typename H::V8;
  using T = typename H::T;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  constexpr int A_IT = BM / 32, B_IT = BN / 32;  // 1-KiB DMA pieces per wave and operand
  constexpr bool CONV = (AMODE == 1 || AMODE == 2);
  constexpr bool RELU_A = (AMODE == 2 || AMODE == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  // ---- staging geometry (fixed per lane over the whole K loop)
  const int lr = lane >> 3;                                   // row inside an 8-row DMA piece
  const int chunk = (lane & 7) ^ ((((wave & 1) << 2) + (lr >> 1)) & 7);  // source 16-B chunk
  const T* a_row[A_IT];
  int a_iy[A_IT], a_ix[A_IT];
  const T* A = (const T*)p.A;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + (i * 4 + wave) * 8 + lr;
    m = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int hw = p.cOH * p.cOW;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
      a_row[i] = A + (size_t)b * p.cH * p.cW * p.cC;
      a_iy[i] = oy * p.cstride - 1;
      a_ix[i] = ox * p.cstride - 1;
    } else {
      a_row[i] = A + (size_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  const T* b_row[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    int n = n0 + (i * 4 + wave) * 8 + lr;
    n = n < p.N ? n : p.N - 1;
    b_row[i] = (const T*)p.W + (size_t)n * p.ldb;
  }
  const T* zeros = (const T*)p.zeros;
  const float inv_cin = CONV ? 1.0f / (float)(p.cC >> 3) : 0.f;
  ConvTap ctap[A_IT];
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ctap[i] = conv_tap_setup(a_row[i], a_iy[i], a_ix[i], chunk * 8, p);
  }

  // K segments: [A_hi x W_hi] (+ [A_hi x W_lo]) (+ [A_lo x W_hi]) — the split-precision planes are
  // just further stretches of the same accumulation loop, selected by a plane byte offset.
  const int nk1 = p.ldb / BK;
  const int seg_wlo = p.W_lo ? 1 : -1;
  const int nseg = 1 + (p.W_lo ? 1 : 0) + (p.A_lo ? 1 : 0);
  const int seg_alo = p.A_lo ? nseg - 1 : -1;
  const ptrdiff_t a_delta = p.A_lo ? (const char*)p.A_lo - (const char*)p.A : 0;
  const ptrdiff_t w_delta = p.W_lo ? (const char*)p.W_lo - (const char*)p.W : 0;

  auto stage = [&](int buf, int kt_all) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
    const int seg = kt_all / nk1;
    const int kt = kt_all - seg * nk1;
    const ptrdiff_t ad = (seg == seg_alo) ? a_delta : 0;
    const ptrdiff_t wd = (seg == seg_wlo) ? w_delta : 0;
    const int k = kt * BK + chunk * 8;
    if constexpr (CONV) {
      // (tap, ci) of this lane's chunk; Cin % 8 == 0 so a chunk never straddles two taps
      int tap, ci;
      if (p.conv_korder) {  // (ci/64, tap, ci%64): one 64-channel block of one tap per K step
        const int c64 = kt / 9;
        tap = kt - c64 * 9;
        ci = c64 * 64 + chunk * 8;
        if (ci >= p.cC) tap = 9;
      } else {
        const int kc = k >> 3;
        tap = (int)(((float)kc + 0.5f) * inv_cin);
        ci = k - tap * p.cC;
      }
      const int ky = tap / 3, kx = tap - ky * 3;
      const int fast_off = (p.conv_korder && tap < 9) ? conv_tap_offset(tap, p) + (kt / 9) * 64 : 0;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        bool ok;
        const char* src;
        if (p.conv_korder) {
          ok = (tap < 9) & ((ctap[i].ok9 >> (tap < 9 ? tap : 0)) & 1);
          src = (const char*)((const T*)ctap[i].center + fast_off) + ad;
        } else {
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          ok = (tap < 9) & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
          src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci) + ad;
        }
        src = ok ? src : (const char*)zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (i * 4 + wave) * 1024),
                                         16, 0, 0);
      }
    } else {
      const bool ok = k < p.K;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const char* src = ok ? (const char*)(a_row[i] + k) + ad : (const char*)zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (i * 4 + wave) * 1024),
                                         16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const char* src = (const char*)(b_row[i] + k) + wd;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sB + (i * 4 + wave) * 1024),
                                       16, 0, 0);
    }
  };

  // ---- fragment read geometry
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TM][2], b_off[TN][2];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * WTM + t * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) a_off[t][kk] = row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * WTN + t * 16 + fr;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_off[t][kk] = row * 128 + (((kk * 4 + fq) ^ ((row >> 1) & 7)) << 4);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = nk1 * nseg;
  stage(0, 0);
  stage_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      V8 af[TM], bf[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        af[t] = *(const V8*)(sA + a_off[t][kk]);
        if constexpr (RELU_A) af[t] = relu8(af[t]);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) bf[t] = *(const V8*)(sB + b_off[t][kk]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = H::mfma16(bf[j], af[i], acc[i][j]);
    }
    stage_barrier();
  }

  epilogue_dispatch<DT, TM, TN>(acc, p, m0 + wm * WTM, n0 + wn * WTN, lane);
}

// Split-precision main loop with the planes FUSED per K step (BK = 32): one stage holds
// A_hi, A_lo, W_hi, W_lo tiles; every fragment is read from LDS once and feeds
// hi*hi + hi*lo + lo*hi, i.e. 16 ds_read_b128 per 48 MFMAs (vs 16 per 32 in the 1-product loop) and
// 1.5x more MFMA work per barrier. 64-byte LDS rows, swizzle chunk ^ ((-(row>>2)) & 3) keeps the
// ds_read_b128 lane groups {0-3,12-15,20-27}.. on 16 distinct 16-byte slots.
template <int DT, int AMODE /*0 plain,1 conv,2 conv+relu,3 plain+relu*/>
__global__ __launch_bounds__(256) void gemm_x3_kernel(const vdn_gemm_desc p) {
  using H = Half<DT>;
  using V8 = typename H::V8;
  using T = typename H::T;
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2, BK3 = 32;
  constexpr int TILE = 128 * BK3 * 2;           // 8 KiB per operand plane
  constexpr int STAGE = 4 * TILE;               // A_hi | A_lo | W_hi | W_lo
  constexpr int WTM = 64, WTN = 64, TM = 4, TN = 4;
  constexpr bool CONV = (AMODE == 1 || AMODE == 2);
  constexpr bool RELU_A = (AMODE == 2 || AMODE == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

  // staging: a 1-KiB DMA piece = 16 rows x 64 B; each wave moves pieces 2w, 2w+1 of every plane
  const int lr = lane >> 2;                                   // row inside the piece
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);     // source 16-B chunk (row>>2 == lane>>4 mod 4)
  const T* a_row[2];
  int a_iy[2], a_ix[2];
  const T* A = (const T*)p.A;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + (wave * 2 + i) * 16 + lr;
    m = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int hw = p.cOH * p.cOW;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
      a_row[i] = A + (size_t)b * p.cH * p.cW * p.cC;
      a_iy[i] = oy * p.cstride - 1;
      a_ix[i] = ox * p.cstride - 1;
    } else {
      a_row[i] = A + (size_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  const T* b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int n = n0 + (wave * 2 + i) * 16 + lr;
    n = n < p.N ? n : p.N - 1;
    b_row[i] = (const T*)p.W + (size_t)n * p.ldb;
  }
  const char* zeros = (const char*)p.zeros;
  const ptrdiff_t a_delta = (const char*)p.A_lo - (const char*)p.A;
  const ptrdiff_t w_delta = (const char*)p.W_lo - (const char*)p.W;
  const float inv_cin = CONV ? 1.0f / (float)(p.cC >> 3) : 0.f;
  ConvTap ctap[2];
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ctap[i] = conv_tap_setup(a_row[i], a_iy[i], a_ix[i], chunk * 8, p);
  }

#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
  auto stage = [&](int buf, int kt) {
    char* s0 = smem + buf * STAGE;
    const int k = kt * BK3 + chunk * 8;
    int ky = 0, kx = 0, ci = 0;
    bool kok = k < p.K;
    if constexpr (CONV) {
      int tap;
      if (p.conv_korder) {  // (ci/64, tap, half, ci%32): scalar decode, same for every lane of the step
        const int half = kt & 1, t2 = kt >> 1;
        const int c64 = t2 / 9;
        tap = t2 - c64 * 9;
        ci = c64 * 64 + half * 32 + chunk * 8;
        if (ci >= p.cC) tap = 9;
      } else {
        tap = (int)(((float)(k >> 3) + 0.5f) * inv_cin);
        ci = k - tap * p.cC;
      }
      ky = tap / 3;
      kx = tap - ky * 3;
      kok = tap < 9;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave * 2 + i;
      const char* src;
      bool ok = kok;
      if constexpr (CONV) {
        if (p.conv_korder) {
          const int tap = ky * 3 + kx;
          ok = ok & ((ctap[i].ok9 >> tap) & 1);
          src = (const char*)((const T*)ctap[i].center + (kok ? conv_tap_offset(tap, p) : 0) + (ci - chunk * 8));
        } else {
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          ok = ok & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
          src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci);
        }
      } else {
        src = (const char*)(a_row[i] + k);
      }
      VDN_GLDS(ok ? src : zeros, s0 + pc * 1024);
      VDN_GLDS(ok ? src + a_delta : zeros, s0 + TILE + pc * 1024);
      const char* ws = (const char*)(b_row[i] + k);
      VDN_GLDS(ws, s0 + 2 * TILE + pc * 1024);
      VDN_GLDS(ws + w_delta, s0 + 3 * TILE + pc * 1024);
    }
  };
#undef VDN_GLDS

  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TM], b_off[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * WTM + t * 16 + fr;
    a_off[t] = row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int row = wn * WTN + t * 16 + fr;
    b_off[t] = row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.ldb / BK3;
  stage(0, 0);
  stage_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* s0 = smem + cur * STAGE;
    V8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      ah[t] = *(const V8*)(s0 + a_off[t]);
      al[t] = *(const V8*)(s0 + TILE + a_off[t]);
      if constexpr (RELU_A) { ah[t] = relu8(ah[t]); al[t] = relu8(al[t]); }
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      bh[t] = *(const V8*)(s0 + 2 * TILE + b_off[t]);
      bl[t] = *(const V8*)(s0 + 3 * TILE + b_off[t]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = H::mfma16(bh[j], al[i], acc[i][j]);
        acc[i][j] = H::mfma16(bl[j], ah[i], acc[i][j]);
        acc[i][j] = H::mfma16(bh[j], ah[i], acc[i][j]);
      }
    stage_barrier();
  }
  epilogue_dispatch<DT, TM, TN>(acc, p, m0 + wm * WTM, n0 + wn * WTN, lane);
}

// Large-tile split-precision main loop: BM x 256 x 32, 8 waves (2 x 4, wave tile BM/2 x 64), one
// workgroup per CU. Per K step a CU moves (BM + 256) * 128 B into LDS for BM*256*32*3 MACs:
// 150-200 flop per byte, which is what the ~25 B/clk/CU global->LDS path can feed (the 128 x 128
// tile needs 43 B/clk at full MFMA rate and stalls on it — profiles/r01_*).
// BM in {128, 192, 256} is chosen per launch so that the tile count fills the 256 CUs evenly.
template <int DT, int AMODE, int BM, int STORE, bool PIPE>
__global__ __launch_bounds__(512) void gemm_x3_big_kernel(const vdn_gemm_desc p) {
  using H = Half<DT>;
  using V8 = typename H::V8;
  using T = typename H::T;
  constexpr int BN = 256, BK3 = 32;
  constexpr int A_TILE = BM * 64, W_TILE = BN * 64;      // bytes per plane
  constexpr int STAGE = 2 * A_TILE + 2 * W_TILE;         // A_hi | A_lo | W_hi | W_lo
  constexpr int TMW = BM / 32, TNW = 4, HALF = TMW / 2;  // frags per wave, A processed in two halves
  constexpr int AP = BM / 16, WP = BN / 16;              // 1-KiB pieces per plane
  constexpr bool CONV = (AMODE == 1 || AMODE == 2);
  constexpr bool RELU_A = (AMODE == 2 || AMODE == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // split-K (convolutions only): the grid is ksplit copies of the tile grid, slice s reduces K steps [kt0, kt0 + nk)
  const int ntiles = tiles_m * tiles_n;
  const int slice = (STORE == VDN_STX_SPLITK) ? (int)blockIdx.x / ntiles : 0;
  const int bid = (int)blockIdx.x - slice * ntiles;
  // tile order inside an XCD's run: groups of 4 m-tiles walk n first, so that the ~32 tiles an
  // XCD has in flight share A rows 8-fold and W columns 4-fold through its L2
  int tile = xcd_remap(bid, ntiles);
  int tm_i, tn_i;
  {
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gm = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tn_i = r / gm;
    tm_i = g * GM + (r - tn_i * gm);
  }
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int nk_total = (AMODE == 1 || AMODE == 2) ? p.ldb / BK3 : p.K / BK3;
  const int nk_slice = (STORE == VDN_STX_SPLITK) ? (nk_total + p.ksplit - 1) / p.ksplit : nk_total;
  const int kt0 = slice * nk_slice;

  const int lr = lane >> 2;
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const T* A = (const T*)p.A;
  const T* a_row[2];
  int a_iy[2], a_ix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pc = wave + 8 * i;
    int m = m0 + pc * 16 + lr;
    m = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int hw = p.cOH * p.cOW;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
      a_row[i] = A + (size_t)b * p.cH * p.cW * p.cC;
      a_iy[i] = oy * p.cstride - 1;
      a_ix[i] = ox * p.cstride - 1;
    } else {
      a_row[i] = A + (size_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  const T* b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int pcw = wave + 8 * i;
    int n = n0 + (vdn_pair8<STORE> ? (pcw >> 2) * 64 + pair8_col(pcw & 3, lr) : pcw * 16 + lr);
    n = n < p.N ? n : p.N - 1;
    b_row[i] = (const T*)p.W + (size_t)n * p.ldb;
  }
  const char* zeros = (const char*)p.zeros;
  const ptrdiff_t a_delta = (const char*)p.A_lo - (const char*)p.A;
  const ptrdiff_t w_delta = (const char*)p.W_lo - (const char*)p.W;
  const float inv_cin = CONV ? 1.0f / (float)(p.cC >> 3) : 0.f;
  ConvTap ctap[2];
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ctap[i] = conv_tap_setup(a_row[i], a_iy[i], a_ix[i], chunk * 8, p);
  }

#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
  auto stage = [&](int buf, int kt) {
    char* s0 = smem + buf * STAGE;
    kt += kt0;  // absolute K step of this slice
    const int k = kt * BK3 + chunk * 8;
    int ky = 0, kx = 0, ci = 0;
    bool kok = k < p.K;
    if constexpr (CONV) {
      int tap;
      if (p.conv_korder) {  // (ci/64, tap, half, ci%32): scalar decode, same for every lane of the step
        const int half = kt & 1, t2 = kt >> 1;
        const int c64 = t2 / 9;
        tap = t2 - c64 * 9;
        ci = c64 * 64 + half * 32 + chunk * 8;
        if (ci >= p.cC) tap = 9;
      } else {
        tap = (int)(((float)(k >> 3) + 0.5f) * inv_cin);
        ci = k - tap * p.cC;
      }
      ky = tap / 3;
      kx = tap - ky * 3;
      kok = tap < 9;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 8 * i;
      if (pc < AP) {
        const char* src;
        bool ok = kok;
        if constexpr (CONV) {
          if (p.conv_korder) {
            const int tap = ky * 3 + kx;
            ok = ok & ((ctap[i].ok9 >> tap) & 1);
            src = (const char*)((const T*)ctap[i].center + (kok ? conv_tap_offset(tap, p) : 0) + (ci - chunk * 8));
          } else {
            const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
            ok = ok & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
            src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci);
          }
        } else {
          src = (const char*)(a_row[i] + k);
        }
        VDN_GLDS(ok ? src : zeros, s0 + pc * 1024);
        VDN_GLDS(ok ? src + a_delta : zeros, s0 + A_TILE + pc * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 8 * i;
      const char* ws = (const char*)(b_row[i] + k);
      VDN_GLDS(ws, s0 + 2 * A_TILE + pc * 1024);
      VDN_GLDS(ws + w_delta, s0 + 2 * A_TILE + W_TILE + pc * 1024);
    }
  };
  // plain rows: per-lane source pointers advance by 64 B per K step (no per-step address math)
  const char* ap[2][2];
  const char* wp[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ap[i][0] = (const char*)(a_row[i] + chunk * 8) + (size_t)kt0 * 64;  // kt0: first K step of this split-K slice
    ap[i][1] = ap[i][0] + a_delta;
    wp[i][0] = (const char*)(b_row[i] + chunk * 8) + (size_t)kt0 * 64;
    wp[i][1] = wp[i][0] + w_delta;
  }
  auto stage_plain = [&](int buf) {
    char* s0 = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 8 * i;
      if (AP == 16 || pc < AP) {
        VDN_GLDS(ap[i][0], s0 + pc * 1024);
        VDN_GLDS(ap[i][1], s0 + A_TILE + pc * 1024);
        ap[i][0] += 64;
        ap[i][1] += 64;
      }
      VDN_GLDS(wp[i][0], s0 + 2 * A_TILE + pc * 1024);
      VDN_GLDS(wp[i][1], s0 + 2 * A_TILE + W_TILE + pc * 1024);
      wp[i][0] += 64;
      wp[i][1] += 64;
    }
  };
#undef VDN_GLDS

  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[TMW], b_off[TNW];
#pragma unroll
  for (int t = 0; t < TMW; ++t) {
    const int row = wm * (BM / 2) + t * 16 + fr;
    a_off[t] = row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int t = 0; t < TNW; ++t) {
    const int row = wn * 64 + t * 16 + fr;
    b_off[t] = 2 * A_TILE + row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }

  f32x4 acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // plain rows advance by pointer increments with no K-tail select: stop at K (a multiple of 32 on this
  // path), NOT at the padded weight stride — reading A columns K..ldb would run into the next row and,
  // on the last row, past the buffer (0 x NaN = NaN even though the padded weights are zero).
  const int nk = (nk_total - kt0) < nk_slice ? (nk_total - kt0) : nk_slice;
  if constexpr (!PIPE) {
    if constexpr (CONV) stage(0, 0); else stage_plain(0);
    stage_barrier();
  }

  // one K step on stage `cur`; STAGED: the next stage's DMA is issued inside the step
  auto step = [&](int kt, auto staged) {
    constexpr bool STAGED = decltype(staged)::value;
    const int cur = kt & 1;
    if constexpr (STAGED) {
      if constexpr (CONV) stage(cur ^ 1, kt + 1); else stage_plain(cur ^ 1);
    }
    const char* s0 = smem + cur * STAGE;
    V8 bh[TNW], bl[TNW];
    V8 ah[2][HALF], al[2][HALF];
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
      bh[t] = *(const V8*)(s0 + b_off[t]);
      bl[t] = *(const V8*)(s0 + W_TILE + b_off[t]);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < HALF; ++t) {
        ah[hf][t] = *(const V8*)(s0 + a_off[hf * HALF + t]);
        al[hf][t] = *(const V8*)(s0 + A_TILE + a_off[hf * HALF + t]);
        if constexpr (RELU_A) { ah[hf][t] = relu8(ah[hf][t]); al[hf][t] = relu8(al[hf][t]); }
      }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < HALF; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          f32x4 c = acc[hf * HALF + i][j];
          c = H::mfma16(bh[j], al[hf][i], c);
          c = H::mfma16(bl[j], ah[hf][i], c);
          c = H::mfma16(bh[j], ah[hf][i], c);
          acc[hf * HALF + i][j] = c;
        }
    stage_barrier();
  };
  if constexpr (!PIPE) {
    for (int kt = 0; kt + 1 < nk; ++kt) step(kt, std::true_type{});
    step(nk - 1, std::false_type{});
  }


  // ---- phase-shifted pipeline (PIPE): the stage barrier sits in the MIDDLE of a K step.
  //   phase 1: MFMAs of A-half 0 (operands already in registers) || ds_read A-half 1 of this stage
  //   [own LDS reads + own DMA complete; barrier]  -> this stage's buffer is free, next stage has landed
  //   phase 2: issue DMA for stage k+2 || ds_read W and A-half 0 of stage k+1 || MFMAs of A-half 1
  // so the matrix pipe always has register-resident work while DMA issue, LDS latency and the barrier
  // pass (the plain loop idles ~1000 cycles per step on them). Costs a second W fragment set.
  if constexpr (PIPE) {
    auto issue = [&](int buf, int kt) {
      if constexpr (CONV) stage(buf, kt); else stage_plain(buf);
    };
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    stage_barrier();
    V8 b0h[TNW], b0l[TNW], b1h[TNW], b1l[TNW], a0h[HALF], a0l[HALF], a1h[HALF], a1l[HALF];
    auto read_b = [&](const char* s0, V8 (&h)[TNW], V8 (&l)[TNW]) {
#pragma unroll
      for (int t = 0; t < TNW; ++t) {
        h[t] = *(const V8*)(s0 + b_off[t]);
        l[t] = *(const V8*)(s0 + W_TILE + b_off[t]);
      }
    };
    auto read_a = [&](const char* s0, int hf, V8 (&h)[HALF], V8 (&l)[HALF]) {
#pragma unroll
      for (int t = 0; t < HALF; ++t) {
        h[t] = *(const V8*)(s0 + a_off[hf * HALF + t]);
        l[t] = *(const V8*)(s0 + A_TILE + a_off[hf * HALF + t]);
        if constexpr (RELU_A) { h[t] = relu8(h[t]); l[t] = relu8(l[t]); }
      }
    };
    auto mma = [&](int hf, V8 (&ah_)[HALF], V8 (&al_)[HALF], V8 (&bh_)[TNW], V8 (&bl_)[TNW]) {
#pragma unroll
      for (int i = 0; i < HALF; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          f32x4 c = acc[hf * HALF + i][j];
          c = H::mfma16(bh_[j], al_[i], c);
          c = H::mfma16(bl_[j], ah_[i], c);
          c = H::mfma16(bh_[j], ah_[i], c);
          acc[hf * HALF + i][j] = c;
        }
    };
    read_b(smem, b0h, b0l);
    read_a(smem, 0, a0h, a0l);
    auto pstep = [&](int kt, V8 (&bch)[TNW], V8 (&bcl)[TNW], V8 (&bnh)[TNW], V8 (&bnl)[TNW]) {
      const int cur = kt & 1;
      const char* sc = smem + cur * STAGE;
      const char* sn = smem + (cur ^ 1) * STAGE;
      read_a(sc, 1, a1h, a1l);
      mma(0, a0h, a0l, bch, bcl);
      stage_barrier();
      if (kt + 2 < nk) issue(cur, kt + 2);
      if (kt + 1 < nk) {
        read_b(sn, bnh, bnl);
        read_a(sn, 0, a0h, a0l);
      }
      mma(1, a1h, a1l, bch, bcl);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      pstep(kt, b0h, b0l, b1h, b1l);
      pstep(kt + 1, b1h, b1l, b0h, b0l);
    }
    if (kt < nk) pstep(kt, b0h, b0l, b1h, b1l);
    stage_barrier();
  }

  if constexpr (STORE == VDN_STX_SPLITK) {
    vdn_gemm_desc q = p;
    q.out = (float*)p.splitk_ws + (size_t)slice * p.M * p.N;
    q.ldc = p.N;
    epilogue_regs<DT, TMW, TNW, STORE, false>(acc, q, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  } else {
    epilogue_regs<DT, TMW, TNW, STORE, vdn_pair8<STORE>>(acc, p, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  }
}

// ---------------------------------------------------------------------------------------------------
// gemm_x3_p8_kernel: 256 x 256 x 32 tile, 8 waves, split operands, ping-pong phases.
//
// The BM x 256 kernel above has every wave do the same thing at the same time (issue DMA, read fragments,
// 96 MFMAs, barrier): the matrix pipe idles while all waves issue / wait, and the per-CU L2->LDS feed idles
// while all waves multiply (tools/gemm_ablate.sh: MFMA-only 1.27 us + DMA-only 1.16 us -> 1.89 us per K step).
// Here the two wave groups (waves 0-3 = A rows 0..127, waves 4-7 = A rows 128..255; wave w and w+4 share
// a SIMD) run ONE BARRIER APART: between two consecutive barriers one group multiplies a 64 x 32 quadrant
// of its 128 x 64 output (24 MFMAs = 384 pipe cycles) while the other reads its next fragments from LDS
// and issues its share of the next K tile's DMA. Per K tile (4 phases, quadrants (0,0) (0,1) (1,1) (1,0)):
//     phase    LDS fragment reads          DMA unit issued (next K tile, 16 KiB = 2 pieces per wave)
//       1      A sub-half 0, W sub-half 0   A rows {0-63, 128-191}      (read in phase 1 of the next tile)
//       2      W sub-half 1                 W rows {32 j .. 32 j + 15 ...} sub-half 0  (phase 1 and 4)
//       3      A sub-half 1                 W sub-half 1                               (phase 2)
//       4      W sub-half 0                 A rows {64-127, 192-255}                   (phase 3)
// Ordering (MI355X_MICROARCH.md 'Two waves per SIMD' item 7, guide '256^2 8-phase template'):
//   RAW: a unit issued in phase p is read no earlier than phase p+3: every wave waits vmcnt(4) (all but its
//        two newest units) before the first barrier of phase p+2, and because the groups are one barrier
//        apart one more barrier has to pass before the other group's pieces are covered.
//   WAR: a unit's LDS region is last read >= 2 phases before it is re-issued (W sub-half 0: read in phase 4,
//        re-issued in phase 2 of the next tile; the reads are retired by the lgkmcnt(0) after that phase's
//        first barrier, two barriers before the earliest re-issue by either group).
//   The last K tile issues nothing, so its waits count down 2, 0 instead of 4.
template <int DT, int AMODE, int STORE, int BM>
__global__ __launch_bounds__(512) void gemm_x3_p8_kernel(const vdn_gemm_desc p) {
  using H = Half<DT>;
  using V8 = typename H::V8;
  using T = typename H::T;
  static_assert(BM == 256 || BM == 192, "A sub-halves of 64 or 48 rows");
  constexpr int BN = 256, BK3 = 32;
  constexpr int A_TILE = BM * 64, W_TILE = BN * 64;  // bytes per operand plane and stage
  constexpr int STAGE = 2 * A_TILE + 2 * W_TILE;     // A_hi | A_lo | W_hi | W_lo
  constexpr int TQ = BM / 64;                        // 16-row fragments per A sub-half (per wave and phase)
  constexpr bool CONV = (AMODE == 1 || AMODE == 2);
  constexpr bool RELU_A = (AMODE == 2 || AMODE == 3);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // Persistent form (launched with fewer workgroups than tiles, a multiple of 8): workgroup b walks the tiles
  // slot, slot + G/8, ... of ITS XCD's contiguous chunk of the tile order, so the 32 workgroups of an XCD still
  // sweep consecutive tiles together; the stores of a finished tile drain while the next tile's main loop runs.
  const int ntiles = tiles_m * tiles_n;
  constexpr bool SPLITK = STORE == VDN_STX_SPLITK;  // grid = ksplit copies of the tile grid (see gemm_x3_big_kernel)
  const int slice = SPLITK ? (int)blockIdx.x / ntiles : 0;
  const bool persistent = !SPLITK && (int)gridDim.x < ntiles;
  for (int it = 0;; ++it) {
  int tile;
  if constexpr (SPLITK) {
    if (it) break;
    tile = xcd_remap((int)blockIdx.x - slice * ntiles, ntiles);
  } else if (persistent) {
    const int xcd = blockIdx.x & 7, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int idx = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
    if (idx >= q8 + (xcd < r8 ? 1 : 0)) break;
    tile = base + idx;
  } else {
    if (it) break;
    tile = xcd_remap(blockIdx.x, ntiles);
  }
  int tm_i, tn_i;
  {
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gm = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tn_i = r / gm;
    tm_i = g * GM + (r - tn_i * gm);
  }
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int wm = wave >> 2, wn = wave & 3;

  const int kt0s = SPLITK ? slice * ((((AMODE == 1 || AMODE == 2) ? p.ldb / BK3 : p.K / BK3) + p.ksplit - 1) / p.ksplit) : 0;
  // ---- DMA duty of this wave: one 16-row piece (x 2 planes) of each of the four units
  const int lr = lane >> 2;
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  // A unit = sub-half `sh` of both groups = 2 TQ pieces x 2 planes. BM 256: 16 piece-planes, wave w takes both
  // planes of piece-list entry w. BM 192: 12 piece-planes: waves 0-3 both planes of entries 0-3, waves 4-7
  // ONE plane of entries 4, 4, 5, 5 (their counted waits differ, see VDN_PHASE).
  const int a_li = (BM == 256 || wave < 4) ? wave : 4 + ((wave - 4) >> 1);
  const bool a_both = BM == 256 || wave < 4;
  const int a_plane = a_both ? 0 : (wave & 1);
  const int pa[2] = {(a_li / TQ) * 2 * TQ + a_li % TQ, (a_li / TQ) * 2 * TQ + TQ + a_li % TQ};
  const int pw[2] = {4 * (wave >> 1) + (wave & 1), 4 * (wave >> 1) + (wave & 1) + 2};  // W pieces of sub-half 0 / 1
  const T* A = (const T*)p.A;
  const T* a_row[2];
  int a_iy[2], a_ix[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + pa[i] * 16 + lr;
    m = m < p.M ? m : p.M - 1;
    if constexpr (CONV) {
      const int hw = p.cOH * p.cOW;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.cOW, ox = rem - oy * p.cOW;
      a_row[i] = A + (size_t)b * p.cH * p.cW * p.cC;
      a_iy[i] = oy * p.cstride - 1;
      a_ix[i] = ox * p.cstride - 1;
    } else {
      a_row[i] = A + (size_t)m * p.lda;
      a_iy[i] = a_ix[i] = 0;
    }
  }
  // conv fast path: pointer to the centre tap's channel chunk and a 9-bit "tap inside the image" mask per piece
  const T* center[2] = {nullptr, nullptr};
  unsigned tap_ok[2] = {0, 0};
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      center[i] = a_row[i] + ((ptrdiff_t)(a_iy[i] + 1) * p.cW + (a_ix[i] + 1)) * p.cC + chunk * 8;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = a_iy[i] + t / 3, ix = a_ix[i] + t % 3;
        tap_ok[i] |= (unsigned)((iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW)) << t;
      }
    }
  }
  const char* zeros = (const char*)p.zeros;
  const ptrdiff_t a_delta = (const char*)p.A_lo - (const char*)p.A;
  const ptrdiff_t w_delta = (const char*)p.W_lo - (const char*)p.W;
  const char* ap[2];
  const char* wp[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int n = n0 + (vdn_pair8<STORE> ? (pw[i] >> 2) * 64 + pair8_col(pw[i] & 3, lr) : pw[i] * 16 + lr);
    n = n < p.N ? n : p.N - 1;
    wp[i] = (const char*)((const T*)p.W + (size_t)n * p.ldb + chunk * 8);
    ap[i] = (const char*)(a_row[i] + chunk * 8);
  }
#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
  // unit u of K tile kt into stage `buf`: 0 = A sub-half 0, 1 = W sub-half 0, 2 = W sub-half 1, 3 = A sub-half 1
  auto issue = [&](auto uc, int buf, int kt) {
    constexpr int u = decltype(uc)::value;
    char* s0 = smem + buf * STAGE;
    if constexpr (u == 0 || u == 3) {
      constexpr int i = u == 0 ? 0 : 1;
      char* dst = s0 + pa[i] * 1024;
      if constexpr (CONV) {
        bool ok;
        const char* src;
        kt += kt0s;  // absolute K step (split-K slices)
        if (p.conv_korder) {
          // (ci/64, tap, ci%64) K order: the K tile fixes one tap and one 32-channel slice for every lane, so
          // the decode is wave-uniform scalar work; per lane only a precomputed validity bit and one 64-bit
          // add remain (the general path below costs ~25 VALU per piece and made the conv K step 40 %
          // slower than the plain GEMM's)
          const int half = kt & 1, t2 = kt >> 1;
          const int c64 = t2 / 9, tap = t2 - c64 * 9;  // scalar
          const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
          const int off = ((ky - 1) * p.cW + (kx - 1)) * p.cC + c64 * 64 + half * 32;  // scalar, elements
          ok = (tap_ok[i] >> tap) & 1;
          src = (const char*)(center[i] + off);
        } else {
          const int k = kt * BK3 + chunk * 8;
          const int tap = (int)(((float)(k >> 3) + 0.5f) * (1.0f / (float)(p.cC >> 3)));
          const int ci = k - tap * p.cC;
          const int ky = tap / 3, kx = tap - ky * 3;
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          ok = (tap < 9) & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
          src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci);
        }
        if (a_both) {
          VDN_GLDS(ok ? src : zeros, dst);
          VDN_GLDS(ok ? src + a_delta : zeros, dst + A_TILE);
        } else {
          VDN_GLDS(ok ? src + (a_plane ? a_delta : 0) : zeros, dst + a_plane * A_TILE);
        }
      } else {
        if (a_both) {
          VDN_GLDS(ap[i], dst);
          VDN_GLDS(ap[i] + a_delta, dst + A_TILE);
        } else {
          VDN_GLDS(ap[i] + (a_plane ? a_delta : 0), dst + a_plane * A_TILE);
        }
        ap[i] += 64;
      }
    } else {
      constexpr int i = u == 1 ? 0 : 1;
      char* dst = s0 + 2 * A_TILE + pw[i] * 1024;
      VDN_GLDS(wp[i], dst);
      VDN_GLDS(wp[i] + w_delta, dst + W_TILE);
      wp[i] += 64;
    }
  };
#undef VDN_GLDS

  // ---- fragment read offsets (same image and swizzle as the BM x 256 kernel)
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2 * TQ], b_off[4];
#pragma unroll
  for (int t = 0; t < 2 * TQ; ++t) {
    const int row = wm * (BM / 2) + t * 16 + fr;
    a_off[t] = row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = wn * 64 + t * 16 + fr;
    b_off[t] = 2 * A_TILE + row * 64 + ((fq ^ ((0 - (row >> 2)) & 3)) << 4);
  }

  f32x4 acc[2 * TQ][4];
#pragma unroll
  for (int i = 0; i < 2 * TQ; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_total = CONV ? p.ldb / BK3 : p.K / BK3;
  const int nk_slice = SPLITK ? (nk_total + p.ksplit - 1) / p.ksplit : nk_total;
  const int kt0 = slice * nk_slice;
  const int nk = (nk_total - kt0) < nk_slice ? (nk_total - kt0) : nk_slice;
  if constexpr (SPLITK) {  // plain rows start at this slice's first K step (pointer-increment staging)
    if constexpr (!CONV) { ap[0] += (size_t)kt0 * 64; ap[1] += (size_t)kt0 * 64; }
    wp[0] += (size_t)kt0 * 64;
    wp[1] += (size_t)kt0 * 64;
  }
  V8 ah[TQ], al[TQ], bh[2], bl[2];
  auto read_a = [&](const char* s0, int qa) {
#pragma unroll
    for (int t = 0; t < TQ; ++t) {
      ah[t] = *(const V8*)(s0 + a_off[qa * TQ + t]);
      al[t] = *(const V8*)(s0 + A_TILE + a_off[qa * TQ + t]);
      if constexpr (RELU_A) { ah[t] = relu8(ah[t]); al[t] = relu8(al[t]); }
    }
  };
  auto read_b = [&](const char* s0, int qb) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bh[t] = *(const V8*)(s0 + b_off[qb * 2 + t]);
      bl[t] = *(const V8*)(s0 + W_TILE + b_off[qb * 2 + t]);
    }
  };
  auto quad = [&](int qa, int qb) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TQ; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 c = acc[qa * TQ + i][qb * 2 + j];
        c = H::mfma16(bh[j], al[i], c);
        c = H::mfma16(bl[j], ah[i], c);
        c = H::mfma16(bh[j], ah[i], c);
        acc[qa * TQ + i][qb * 2 + j] = c;
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // one phase: [fragment reads, DMA issue, counted wait] barrier [MFMAs] barrier
// counted wait: all but the two newest units (VA for waves that carry 2 pieces of an A unit, VB for BM 192's
// waves 4-7 that carry one)
#define VDN_PHASE(READS, ISSUE, VA, VB, QA, QB)                  \
  do {                                                           \
    READS;                                                       \
    ISSUE;                                                       \
    if (a_both) asm volatile("s_waitcnt vmcnt(" #VA ")" ::: "memory");  \
    else asm volatile("s_waitcnt vmcnt(" #VB ")" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
    __builtin_amdgcn_sched_barrier(0);                           \
    quad(QA, QB);                                                \
    __builtin_amdgcn_sched_barrier(0);                           \
    __builtin_amdgcn_s_barrier();                                \
  } while (0)
  constexpr std::integral_constant<int, 0> U0{};
  constexpr std::integral_constant<int, 1> U1{};
  constexpr std::integral_constant<int, 2> U2{};
  constexpr std::integral_constant<int, 3> U3{};

  // prologue: the whole first K tile
  issue(U0, 0, 0);
  issue(U1, 0, 0);
  issue(U2, 0, 0);
  issue(U3, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // second group runs one barrier behind

  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* sc = smem + (kt & 1) * STAGE;
    const int nb = (kt + 1) & 1;
    VDN_PHASE((read_a(sc, 0), read_b(sc, 0)), issue(U0, nb, kt + 1), 4, 2, 0, 0);
    VDN_PHASE(read_b(sc, 1), issue(U1, nb, kt + 1), 4, 3, 0, 1);
    VDN_PHASE(read_a(sc, 1), issue(U2, nb, kt + 1), 4, 4, 1, 1);
    VDN_PHASE(read_b(sc, 0), issue(U3, nb, kt + 1), 4, 3, 1, 0);
  }
  {
    const char* sc = smem + ((nk - 1) & 1) * STAGE;
    VDN_PHASE((read_a(sc, 0), read_b(sc, 0)), (void)0, 2, 1, 0, 0);
    VDN_PHASE(read_b(sc, 1), (void)0, 0, 0, 0, 1);
    VDN_PHASE(read_a(sc, 1), (void)0, 0, 0, 1, 1);
    VDN_PHASE(read_b(sc, 0), (void)0, 0, 0, 1, 0);
  }
#undef VDN_PHASE
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count of the two groups

  if constexpr (SPLITK) {
    vdn_gemm_desc q = p;
    q.out = (float*)p.splitk_ws + (size_t)slice * p.M * p.N;
    q.ldc = p.N;
    epilogue_regs<DT, 2 * TQ, 4, STORE, false>(acc, q, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  } else {
    epilogue_regs<DT, 2 * TQ, 4, STORE, vdn_pair8<STORE>>(acc, p, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  }
  }  // tile loop
}

template <int DT, int BM, int BN, int WM, int WN>
int launch_tile(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  const size_t lds = 2 * (size_t)(BM + BN) * BK * 2;
  const bool conv = d.a_mode == VDN_A_CONV3X3;
  const int amode = conv ? (d.relu_a ? 2 : 1) : (d.relu_a ? 3 : 0);
  switch (amode) {
    case 0: hipLaunchKernelGGL((gemm_kernel<DT, BM, BN, WM, WN, 0>), dim3(tiles), dim3(256), lds, s, d); break;
    case 1: hipLaunchKernelGGL((gemm_kernel<DT, BM, BN, WM, WN, 1>), dim3(tiles), dim3(256), lds, s, d); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<DT, BM, BN, WM, WN, 2>), dim3(tiles), dim3(256), lds, s, d); break;
    default: hipLaunchKernelGGL((gemm_kernel<DT, BM, BN, WM, WN, 3>), dim3(tiles), dim3(256), lds, s, d); break;
  }
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <int DT>
int launch_x3(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + 127) / 128) * ((d.N + 127) / 128);
  const size_t lds = 65536;  // 2 stages x 4 planes x 8 KiB
  const bool conv = d.a_mode == VDN_A_CONV3X3;
  const int amode = conv ? (d.relu_a ? 2 : 1) : (d.relu_a ? 3 : 0);
  switch (amode) {
    case 0: hipLaunchKernelGGL((gemm_x3_kernel<DT, 0>), dim3(tiles), dim3(256), lds, s, d); break;
    case 1: hipLaunchKernelGGL((gemm_x3_kernel<DT, 1>), dim3(tiles), dim3(256), lds, s, d); break;
    case 2: hipLaunchKernelGGL((gemm_x3_kernel<DT, 2>), dim3(tiles), dim3(256), lds, s, d); break;
    default: hipLaunchKernelGGL((gemm_x3_kernel<DT, 3>), dim3(tiles), dim3(256), lds, s, d); break;
  }
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <int DT, int BM>
int launch_x3_big(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + 255) / 256);
  const size_t lds = 2 * (size_t)(2 * BM * 64 + 2 * 256 * 64);
  const dim3 g(tiles), b(512);
  const bool no_pipe = tuning(d).no_pipe != 0;
  constexpr bool CAN_PIPE = BM <= 192;  // BM = 256 has no registers for the second W fragment set
  const bool pipe = CAN_PIPE && !no_pipe;
#define VDN_LAUNCH_BIG(AM, ST)                                                                          \
  do {                                                                                                  \
    if constexpr (CAN_PIPE && (AM == 0 || BM == 128)) { /* conv + BM=192 would spill */                \
      if (pipe) hipLaunchKernelGGL((gemm_x3_big_kernel<DT, AM, BM, ST, true>), g, b, lds, s, d);        \
      else hipLaunchKernelGGL((gemm_x3_big_kernel<DT, AM, BM, ST, false>), g, b, lds, s, d);            \
    } else {                                                                                            \
      hipLaunchKernelGGL((gemm_x3_big_kernel<DT, AM, BM, ST, false>), g, b, lds, s, d);                 \
    }                                                                                                   \
  } while (0)
  // straight-line epilogue flavours exist for the plain-A kernels at BM 256 / 192 (the encoder / memory linears)
  int fl = epi_flavour(d);
  // plane-output flavours store 8 columns (16 bytes) per lane: rows and column counts must keep that aligned
  const bool a8 = !(d.N & 7) && !(d.ldc & 7) && !((uintptr_t)d.out & 15) && !((uintptr_t)d.out_lo & 15) && !(d.ldr1 & 7) &&
                  !(d.ldr2 & 7) && !(((uintptr_t)d.res1 | (uintptr_t)d.res1_lo | (uintptr_t)d.res2 | (uintptr_t)d.res2_lo) & 15);
  if (!a8 && fl != VDN_STX_RES && fl != VDN_STX_HEADS) fl = d.store;
  if (d.a_mode == VDN_A_CONV3X3) {
    if (fl != VDN_STX_HALF && fl != VDN_STX_RESHALF1 && fl != VDN_STX_RESHALF2) fl = VDN_ST_PLAIN;
  } else if (fl == VDN_STX_RESHALF1 || fl == VDN_STX_RESHALF2) {
    fl = d.store;
  }
  if constexpr (BM == 256 || BM == 192) {  // ping-pong 8-phase kernel (VDN_GEMM_P8=0 falls back to the lock-step one)
    // measured (tools/gemm_bench.py): the ping-pong loop runs at 97 % of the clock-limited MFMA rate at BM 256
    // (24 MFMAs cover a load segment) but not at BM 192 (18 do not), where the lock-step PIPE loop is as fast:
    // default = BM 256 only; VDN_GEMM_P8=2 also BM 192, =0 never.
    if (tuning(d).p8 >= (BM == 256 ? 1 : 2)) {
      const dim3 g8 = g;
#define VDN_LAUNCH_P8(AM, ST) hipLaunchKernelGGL((gemm_x3_p8_kernel<DT, AM, ST, BM>), g8, b, lds, s, d)
      if (d.a_mode == VDN_A_CONV3X3) {
#define VDN_CONV_P8(ST) do { if (d.relu_a) VDN_LAUNCH_P8(2, ST); else VDN_LAUNCH_P8(1, ST); } while (0)
        switch (fl) {
          case VDN_STX_HALF: VDN_CONV_P8(VDN_STX_HALF); break;
          case VDN_STX_RESHALF1: VDN_CONV_P8(VDN_STX_RESHALF1); break;
          case VDN_STX_RESHALF2: VDN_CONV_P8(VDN_STX_RESHALF2); break;
          default: VDN_CONV_P8(VDN_ST_PLAIN); break;
        }
#undef VDN_CONV_P8
      } else {
        switch (fl) {
          case VDN_ST_PLAIN: VDN_LAUNCH_P8(0, VDN_ST_PLAIN); break;
          case VDN_STX_HALF: VDN_LAUNCH_P8(0, VDN_STX_HALF); break;
          case VDN_ST_CONVT: VDN_LAUNCH_P8(0, VDN_ST_CONVT); break;
          case VDN_ST_GEGLU: VDN_LAUNCH_P8(0, VDN_ST_GEGLU); break;
          case VDN_STX_FC1: VDN_LAUNCH_P8(0, VDN_STX_FC1); break;
          case VDN_STX_RES: VDN_LAUNCH_P8(0, VDN_STX_RES); break;
          case VDN_STX_HEADS: VDN_LAUNCH_P8(0, VDN_STX_HEADS); break;
          default: VDN_LAUNCH_P8(0, VDN_ST_HEADS); break;
        }
      }
#undef VDN_LAUNCH_P8
      VDN_CHECK_LAUNCH();
      return VDN_OK;
    }
  }
  if (d.a_mode == VDN_A_CONV3X3) {  // convolutions always store plain NHWC rows
#define VDN_CONV_BIG(ST) do { if (d.relu_a) VDN_LAUNCH_BIG(2, ST); else VDN_LAUNCH_BIG(1, ST); } while (0)
    switch (fl) {
      case VDN_STX_HALF: VDN_CONV_BIG(VDN_STX_HALF); break;
      case VDN_STX_RESHALF1: VDN_CONV_BIG(VDN_STX_RESHALF1); break;
      case VDN_STX_RESHALF2: VDN_CONV_BIG(VDN_STX_RESHALF2); break;
      default: VDN_CONV_BIG(VDN_ST_PLAIN); break;
    }
#undef VDN_CONV_BIG
  } else {
    switch (fl) {
      case VDN_ST_PLAIN: VDN_LAUNCH_BIG(0, VDN_ST_PLAIN); break;
      case VDN_STX_HALF: VDN_LAUNCH_BIG(0, VDN_STX_HALF); break;
      case VDN_ST_CONVT: VDN_LAUNCH_BIG(0, VDN_ST_CONVT); break;
      case VDN_ST_GEGLU: VDN_LAUNCH_BIG(0, VDN_ST_GEGLU); break;
      case VDN_STX_FC1: VDN_LAUNCH_BIG(0, VDN_STX_FC1); break;
      case VDN_STX_RES: VDN_LAUNCH_BIG(0, VDN_STX_RES); break;
      case VDN_STX_HEADS: VDN_LAUNCH_BIG(0, VDN_STX_HEADS); break;
      default: VDN_LAUNCH_BIG(0, VDN_ST_HEADS); break;
    }
  }
#undef VDN_LAUNCH_BIG
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <int DT> int big_entry(const vdn_gemm_desc& d, int bm, hipStream_t s);  // defined in gemm_big_*.hip
int x8_entry(const vdn_gemm_desc& d, hipStream_t s);                             // gemm_x8.hip (fp16 only)

// ---- split-K for convolutions whose tile grid covers a fraction of the chip (include/vdn.h: splitk_ws)
// second pass: partial sums of the K slices added in slice order, then the SAME straight-line epilogue flavour
template <int DT, int FL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const vdn_gemm_desc p) {
  const int n4 = p.N >> 2;
  const size_t total = (size_t)p.M * n4, mn = (size_t)p.M * p.N;
  const float* ws = (const float*)p.splitk_ws;
  const f32x4 one = {1.f, 1.f, 1.f, 1.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
    f32x4 a = *(const f32x4*)(ws + (size_t)m * p.N + n);
    for (int sl = 1; sl < p.ksplit; ++sl) a += *(const f32x4*)(ws + sl * mn + (size_t)m * p.N + n);
    const f32x4 b4 = p.bias ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 g4 = p.gamma ? *(const f32x4*)(p.gamma + n) : one;
    emit4<DT, FL>(p, m, n, a, a, b4, b4, g4);
  }
}

template <int DT>
int launch_splitk(const vdn_gemm_desc& d0, int ksplit, int fl, hipStream_t s) {
  vdn_gemm_desc d = d0;
  const bool p8 = ksplit < 0;  // negative: the 256 x 256 ping-pong kernel (plain A only), |ksplit| slices
  d.ksplit = p8 ? -ksplit : ksplit;
  const int tiles = p8 ? ((d.M + 255) / 256) * ((d.N + 255) / 256) : ((d.M + 127) / 128) * ((d.N + 255) / 256);
  const size_t lds = p8 ? 2 * (size_t)(2 * 256 * 64 + 2 * 256 * 64) : 2 * (size_t)(2 * 128 * 64 + 2 * 256 * 64);
  if (p8) hipLaunchKernelGGL((gemm_x3_p8_kernel<DT, 0, VDN_STX_SPLITK, 256>), dim3(tiles * d.ksplit), dim3(512), lds, s, d);
  else if (d.a_mode != VDN_A_CONV3X3) hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 0, 128, VDN_STX_SPLITK, true>), dim3(tiles * ksplit), dim3(512), lds, s, d);
  else if (d.relu_a) hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 2, 128, VDN_STX_SPLITK, true>), dim3(tiles * ksplit), dim3(512), lds, s, d);
  else hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 1, 128, VDN_STX_SPLITK, true>), dim3(tiles * ksplit), dim3(512), lds, s, d);
  const size_t work = (size_t)d.M * (d.N >> 2);
  const dim3 g((unsigned)((work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096));
  switch (fl) {
    case VDN_STX_HALF: hipLaunchKernelGGL((splitk_reduce_kernel<DT, VDN_STX_HALF>), g, dim3(256), 0, s, d); break;
    case VDN_STX_RESHALF1: hipLaunchKernelGGL((splitk_reduce_kernel<DT, VDN_STX_RESHALF1>), g, dim3(256), 0, s, d); break;
    case VDN_STX_RES: hipLaunchKernelGGL((splitk_reduce_kernel<DT, VDN_STX_RES>), g, dim3(256), 0, s, d); break;
    default: hipLaunchKernelGGL((splitk_reduce_kernel<DT, VDN_STX_RESHALF2>), g, dim3(256), 0, s, d); break;
  }
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
template <int DT> int splitk_entry(const vdn_gemm_desc& d, int ksplit, int fl, hipStream_t s);  // defined in gemm_big_*.hip

// Pick the M tile that wastes the fewest CU-rounds: cost = rounds(256 CUs) * BM, padded work included.
inline int pick_bm(int M, int N, int cu_hint, const vdn_gemm_tuning& tu) {
  const int tn = (N + 255) / 256;
  const int cus = tu.cus > 0 ? tu.cus : (cu_hint > 0 && cu_hint <= 256 ? cu_hint : 256);  // CUs one launch can count on
  int best = 0;
  double best_cost = 1e30;
  for (int bm : {256, 192, 128}) {
    const long tiles = (long)((M + bm - 1) / bm) * tn;
    const long rounds = (tiles + cus - 1) / cus;
    const double cost = (double)rounds * bm * (bm == 128 ? tu.f128 : (bm == 192 ? tu.f192 : 1.0));  // smaller tiles feed worse
    if (cost < best_cost) { best_cost = cost; best = bm; }
  }
  return best;
}

template <int DT>
int launch_dt(const vdn_gemm_desc& d, hipStream_t s) {
  // 8-bit cross terms (gemm_x8.hip): plain A, fp16, K a multiple of 64, enough rows to fill 256 x 256 tiles
  if constexpr (DT == VDN_F16) {
    if (d.A8 && d.W8) {
      if (d.a_mode == VDN_A_PLAIN && !d.relu_a && !(d.K & 63) && d.N >= 192 && (long)d.M * d.N >= 1024L * 1024 && tuning(d).x8 != 0)
        return x8_entry(d, s);
      // only that kernel reads K-tile-major planes and writes out8 / a lo-less half output
      if (d.a_kt || d.w_kt || d.out_kt || d.out8 || !d.A_lo || !d.W_lo) return VDN_EUNSUPPORTED;
    }
  }
  // 8-wave kernels: large problems, and small ones whose deep reduction makes them split-K candidates
  const bool deep = d.splitk_ws && (d.a_mode == VDN_A_CONV3X3 ? d.ldb : d.K) >= 2048 && (long)d.M * d.N >= 32L * 1024;
  if (d.A_lo && d.W_lo && d.N >= 192 && ((long)d.M * d.N >= 256L * 1024 || deep) &&
      (d.a_mode == VDN_A_CONV3X3 ? d.store == VDN_ST_PLAIN : ((d.K & 31) == 0 && !d.relu_a))) {
    const vdn_gemm_tuning& tu = tuning(d);
    const int force = tu.force_bm;
    if (d.splitk_ws && !force && !(d.N & 3) && !tu.no_splitk) {
      const bool conv = d.a_mode == VDN_A_CONV3X3;
      const int fl = epi_flavour(d);
      const int cus = d.cu_hint > 0 && d.cu_hint <= 256 ? d.cu_hint : 256;
      const long tiles128 = (long)((d.M + 127) / 128) * ((d.N + 255) / 256);
      const int nk_total = conv ? d.ldb / 32 : d.K / 32;
      const bool fl_ok = fl == VDN_STX_HALF || (conv ? (fl == VDN_STX_RESHALF1 || fl == VDN_STX_RESHALF2) : fl == VDN_STX_RES);
      const long occ = tu.splitk_occ, ks_max = tu.splitk_max;  // occupancy threshold (percent), slice cap
      if (fl_ok && tiles128 * 100 <= cus * occ && nk_total >= (conv ? 64 : 32)) {
        long ks = (cus + tiles128 - 1) / tiles128;
        ks = ks < ks_max ? ks : ks_max;
        ks = ks < nk_total / 16 ? ks : nk_total / 16;
        const long fit = d.splitk_ws_bytes / ((long)d.M * d.N * 4);
        ks = ks < fit ? ks : fit;
        while (ks >= 2 && (ks - 1) * ((nk_total + ks - 1) / ks) >= nk_total) --ks;  // every slice non-empty
        if (ks >= 2) return splitk_entry<DT>(d, (int)ks, fl, s);
      }
    }
    {  // experiment (measured neutral): deep residual linears on the ping-pong kernel, K split
      const int ks = tu.splitk_p8;
      if (ks >= 2 && d.splitk_ws && !force && d.a_mode == VDN_A_PLAIN && epi_flavour(d) == VDN_STX_RES && d.K >= 2048 &&
          (long)d.M * d.N * 4 * ks <= d.splitk_ws_bytes)
        return splitk_entry<DT>(d, -ks, VDN_STX_RES, s);
    }
    const int bm = force ? force : pick_bm(d.M, d.N, d.cu_hint, tu);
    // small problems (batch 1: M = 1370): a grid of 128 x 256 tiles covers a fraction of the chip; the 4-wave 128 x 128
    // kernel launches twice the workgroups (two per CU) with half the K-loop work each
    // (batch 1: 15.9 -> 14.8 ms per frame, batch 2: 18.4 -> 17.3 ms with the threshold at 96 tiles)
    const long min_tiles = tu.min_tiles;  // 0 disables
    const long tiles128 = (long)((d.M + 127) / 128) * ((d.N + 255) / 256);
    const bool small = !force && ((min_tiles > 0 && tiles128 < min_tiles && d.a_mode != VDN_A_CONV3X3) ||
                                 (long)d.M * d.N < 256L * 1024);  // admitted only as a split-K candidate
    if (!small && (bm == 256 || bm == 192 || bm == 128)) return big_entry<DT>(d, bm, s);
  }
  if (d.A_lo && d.W_lo && (d.store == VDN_ST_HEADS || d.N > 64)) return launch_x3<DT>(d, s);
  if (d.store == VDN_ST_HEADS || d.N > 64) return launch_tile<DT, 128, 128, 2, 2>(d, s);
  if (d.N > 32) return launch_tile<DT, 128, 64, 2, 2>(d, s);
  return launch_tile<DT, 128, 32, 4, 1>(d, s);
}


}  // namespace vdn_gemm_impl
