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
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int BK = 64;

// Fused fp32 epilogue over a ROWS x BN fp32 tile held in LDS (row stride BN + 4 floats):
// bias / row-add / activation / LayerScale / pos-embed table / residuals, then wide stores in the
// layout the consumer reads (plain rows, per-head Q/K/V^T with RoPE, pixel-shuffle, GEGLU).
template <int DT, int ROWS, int BN, int NT>
__device__ __forceinline__ void epilogue_tile(const vdn_gemm_desc& p, const float* tileC, int mbase, int n0, int tid) {
  using H = Half<DT>;
  using T = typename H::T;
  constexpr int LDT = BN + 4;
  constexpr int CG = BN / 4;  // 4-column groups per tile row
  if (p.store == VDN_ST_PLAIN || p.store == VDN_ST_CONVT) {
    for (int idx = tid; idx < ROWS * CG; idx += NT) {
      const int r = idx / CG, c = (idx - r * CG) * 4;
      const int m = mbase + r, n = n0 + c;
      if (m >= p.M || n >= p.N) continue;
      f32x4 a = *(const f32x4*)(tileC + r * LDT + c);
      if (p.bias) a += *(const f32x4*)(p.bias + n);
      if (p.rowadd) a += p.rowadd[m];
      if (p.act == VDN_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = gelu_fast(a[e]);
      } else if (p.act == VDN_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
      }
      if (p.gamma) a *= *(const f32x4*)(p.gamma + n);
      if (p.tab) a += *(const f32x4*)(p.tab + (size_t)(m % p.tab_mod + p.tab_off) * p.N + n);
      if (p.res1) {
        a += load4_as_float(p.res1, p.res1_dt, (size_t)m * p.ldr1 + n);
        if (p.res1_lo) a += load4_as_float(p.res1_lo, p.res1_dt, (size_t)m * p.ldr1 + n);
      }
      if (p.res2) {
        a += load4_as_float(p.res2, p.res2_dt, (size_t)m * p.ldr2 + n);
        if (p.res2_lo) a += load4_as_float(p.res2_lo, p.res2_dt, (size_t)m * p.ldr2 + n);
      }
      const float v[4] = {a[0], a[1], a[2], a[3]};
      size_t o;
      if (p.store == VDN_ST_CONVT) {
        const int hw = p.cH * p.cW;
        const int cb = m / hw, rem = m - cb * hw;
        const int cy = rem / p.cW, cx = rem - cy * p.cW;
        const int kk = n / p.cout, co = n - kk * p.cout;  // cout % 4 == 0: the group stays in one tap
        const int ky = kk / p.ck, kx = kk - ky * p.ck;
        o = (((size_t)cb * (p.cH * p.ck) + cy * p.ck + ky) * (p.cW * p.ck) + cx * p.ck + kx) * p.cout + co;
      } else {
        o = (size_t)(p.row_group > 0 ? m + (m / p.row_group + 1) * p.row_skip : m) * p.ldc + n;
      }
      if (p.out_dt == VDN_F32) {
        *(f32x4*)((float*)p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
      } else if (p.out_lo) {
        typename H::V4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { T a, b; split_rtz(v[e], a, b); h[e] = a; l[e] = b; }
        *(typename H::V4*)((T*)p.out + o) = h;
        *(typename H::V4*)((T*)p.out_lo + o) = l;
      } else {
        typename H::V4 h = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *(typename H::V4*)((T*)p.out + o) = h;
      }
    }
  } else if (p.store == VDN_ST_GEGLU) {
    // packed columns: 16-wide blocks alternate [h | gate]; output column nc <- (h, gate) pair
    for (int idx = tid; idx < ROWS * (CG / 2); idx += NT) {
      const int r = idx / (CG / 2), oc = (idx - r * (CG / 2)) * 4;  // output column inside the tile
      const int ch = (oc >> 4) * 32 + (oc & 15);                     // packed h column inside the tile
      const int m = mbase + r;
      if (m >= p.M || n0 + ch + 16 >= p.N) continue;
      const f32x4 hh = *(const f32x4*)(tileC + r * LDT + ch);
      const f32x4 gg = *(const f32x4*)(tileC + r * LDT + ch + 16);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hv = hh[e] + (p.bias ? p.bias[n0 + ch + e] : 0.f);
        const float gv = gg[e] + (p.bias ? p.bias[n0 + ch + 16 + e] : 0.f);
        v[e] = hv * gelu_fast(gv);
      }
      const size_t o = (size_t)m * p.ldc + (n0 >> 1) + oc;
      if (p.out_dt == VDN_F32) {
        *(f32x4*)((float*)p.out + o) = f32x4{v[0], v[1], v[2], v[3]};
      } else if (p.out_lo) {
        typename H::V4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { T a, b; split_rtz(v[e], a, b); h[e] = a; l[e] = b; }
        *(typename H::V4*)((T*)p.out + o) = h;
        *(typename H::V4*)((T*)p.out_lo + o) = l;
      } else {
        typename H::V4 h = {(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
        *(typename H::V4*)((T*)p.out + o) = h;
      }
    }
  } else {  // VDN_ST_HEADS
    const int hc = p.heads * 64;
    // pass 1: token-major splits, one thread = 4 columns (plain) or 4 rotated pairs (RoPE)
    for (int idx = tid; idx < ROWS * CG; idx += NT) {
      const int r = idx / CG, c = (idx - r * CG) * 4;
      const int m = mbase + r, n = n0 + c;
      if (m >= p.M || n >= p.N) continue;
      const int split = n / hc;
      if (p.transposed[split]) continue;
      const int head = (n - split * hc) >> 6, e0 = n & 63;
      const int bt = m / p.tokens, tl = m - bt * p.tokens;
      const size_t doff = (((size_t)bt * p.heads + head) * p.tpad + tl + p.tok_off) * 64;
      T* dst = (T*)p.dst[split] + doff;
      T* dlo = p.dst_lo[split] ? (T*)p.dst_lo[split] + doff : nullptr;
      const f32x4 a = *(const f32x4*)(tileC + r * LDT + c);
      if (p.rope[split]) {
        if (e0 & 16) continue;  // imaginary tile: consumed by the thread owning the real tile
        const f32x4 b = *(const f32x4*)(tileC + r * LDT + c + 16);
        const int pi = ((e0 >> 5) << 4) + (e0 & 15);  // first of 4 consecutive pair indices
        const float* cs = p.rope_cs + (size_t)(tl % p.rope_mod) * 64 + 2 * pi;
        typename H::V8 o8, l8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float re = a[e] + (p.bias ? p.bias[n + e] : 0.f);
          const float im = b[e] + (p.bias ? p.bias[n + 16 + e] : 0.f);
          const float cc = cs[2 * e], ss = cs[2 * e + 1];
          const float ore = re * cc - im * ss, oim = re * ss + im * cc;
          if (dlo) {
            T x0, x1, y0, y1;
            split_rtz(ore, x0, x1);
            split_rtz(oim, y0, y1);
            o8[2 * e] = x0; l8[2 * e] = x1; o8[2 * e + 1] = y0; l8[2 * e + 1] = y1;
          } else {
            o8[2 * e] = (T)ore;
            o8[2 * e + 1] = (T)oim;
          }
        }
        *(typename H::V8*)(dst + 2 * pi) = o8;
        if (dlo) *(typename H::V8*)(dlo + 2 * pi) = l8;
      } else {
        typename H::V4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = a[e] + (p.bias ? p.bias[n + e] : 0.f);
          if (dlo) { T x0, x1; split_rtz(v, x0, x1); h[e] = x0; l[e] = x1; }
          else h[e] = (T)v;
        }
        *(typename H::V4*)(dst + e0) = h;
        if (dlo) *(typename H::V4*)(dlo + e0) = l;
      }
    }
    // pass 2: dim-major (V^T) splits, one thread = 4 consecutive tokens of one column, so that
    // neighbouring threads write neighbouring tokens of the same [64, tpad] row
    bool any_t = false;
    for (int i = 0; i < p.nsplit; ++i) any_t |= (p.transposed[i] != 0);
    if (any_t) {
      constexpr int RG = ROWS / 4;
      for (int idx = tid; idx < BN * RG; idx += NT) {
        const int c = idx / RG, r = (idx - c * RG) * 4;
        const int n = n0 + c;
        if (n >= p.N) continue;
        const int split = n / hc;
        if (!p.transposed[split]) continue;
        const int head = (n - split * hc) >> 6, e = n & 63;
        const float bv = p.bias ? p.bias[n] : 0.f;
        T* dst = (T*)p.dst[split];
        T* dlo = (T*)p.dst_lo[split];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int m = mbase + r + k;
          if (m >= p.M) break;
          const int bt = m / p.tokens, tl = m - bt * p.tokens;
          store_half(dst, dlo, (((size_t)bt * p.heads + head) * 64 + e) * p.tpad + tl + p.tok_off,
                     tileC[(r + k) * LDT + c] + bv);
        }
      }
    }
  }
}

// 4-wave kernels: accumulators -> LDS, then the shared epilogue. Must be entered after a barrier
// (the staging buffers are reused for the tile).
template <int DT, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], const vdn_gemm_desc& p,
                                              char* smem, int m0, int n0, int tid, int lane, int wave) {
  constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 16, TN = WTN / 16;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fq = lane >> 4;
  constexpr int LDT = BN + 4;  // padded fp32 row (floats)
  float* tileC = (float*)smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        tileC[(wm * WTM + i * 16 + fq * 4 + j) * LDT + wn * WTN + t * 16 + fr] = acc[i][t][j];
  __syncthreads();
  epilogue_tile<DT, BM, BN, 256>(p, tileC, m0, n0, tid);
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
      const int kc = k >> 3;
      const int tap = (int)(((float)kc + 0.5f) * inv_cin);
      const int ci = k - tap * p.cC;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        const bool ok = (tap < 9) & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
        const char* src = ok ? (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci) + ad : (const char*)zeros;
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
  __syncthreads();
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
        for (int j = 0; j < TN; ++j) acc[i][j] = H::mfma16(af[i], bf[j], acc[i][j]);
    }
    __syncthreads();
  }

  gemm_epilogue<DT, BM, BN, WM, WN>(acc, p, smem, m0, n0, tid, lane, wave);
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

#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
  auto stage = [&](int buf, int kt) {
    char* s0 = smem + buf * STAGE;
    const int k = kt * BK3 + chunk * 8;
    int ky = 0, kx = 0, ci = 0;
    bool kok = k < p.K;
    if constexpr (CONV) {
      const int tap = (int)(((float)(k >> 3) + 0.5f) * inv_cin);
      ci = k - tap * p.cC;
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
        const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
        ok = ok & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
        src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci);
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
  __syncthreads();
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
        acc[i][j] = H::mfma16(al[i], bh[j], acc[i][j]);
        acc[i][j] = H::mfma16(ah[i], bl[j], acc[i][j]);
        acc[i][j] = H::mfma16(ah[i], bh[j], acc[i][j]);
      }
    __syncthreads();
  }
  gemm_epilogue<DT, BM, BN, WM, WN>(acc, p, smem, m0, n0, tid, lane, wave);
}

// Large-tile split-precision main loop: BM x 256 x 32, 8 waves (2 x 4, wave tile BM/2 x 64), one
// workgroup per CU. Per K step a CU moves (BM + 256) * 128 B into LDS for BM*256*32*3 MACs:
// 150-200 flop per byte, which is what the ~25 B/clk/CU global->LDS path can feed (the 128 x 128
// tile needs 43 B/clk at full MFMA rate and stalls on it — profiles/r01_*).
// BM in {128, 192, 256} is chosen per launch so that the tile count fills the 256 CUs evenly.
template <int DT, int AMODE, int BM>
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
  // tile order inside an XCD's run: groups of 4 m-tiles walk n first, so that the ~32 tiles an
  // XCD has in flight share A rows 8-fold and W columns 4-fold through its L2
  int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
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
    int n = n0 + (wave + 8 * i) * 16 + lr;
    n = n < p.N ? n : p.N - 1;
    b_row[i] = (const T*)p.W + (size_t)n * p.ldb;
  }
  const char* zeros = (const char*)p.zeros;
  const ptrdiff_t a_delta = (const char*)p.A_lo - (const char*)p.A;
  const ptrdiff_t w_delta = (const char*)p.W_lo - (const char*)p.W;
  const float inv_cin = CONV ? 1.0f / (float)(p.cC >> 3) : 0.f;

#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
  auto stage = [&](int buf, int kt) {
    char* s0 = smem + buf * STAGE;
    const int k = kt * BK3 + chunk * 8;
    int ky = 0, kx = 0, ci = 0;
    bool kok = k < p.K;
    if constexpr (CONV) {
      const int tap = (int)(((float)(k >> 3) + 0.5f) * inv_cin);
      ci = k - tap * p.cC;
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
          const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
          ok = ok & (iy >= 0) & (iy < p.cH) & (ix >= 0) & (ix < p.cW);
          src = (const char*)(a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci);
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
    ap[i][0] = (const char*)(a_row[i] + chunk * 8);
    ap[i][1] = ap[i][0] + a_delta;
    wp[i][0] = (const char*)(b_row[i] + chunk * 8);
    wp[i][1] = wp[i][0] + w_delta;
  }
  auto stage_plain = [&](int buf) {
    char* s0 = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 8 * i;
      if (pc < AP) {
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

  const int nk = p.ldb / BK3;
  if constexpr (CONV) stage(0, 0); else stage_plain(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      if constexpr (CONV) stage(cur ^ 1, kt + 1); else stage_plain(cur ^ 1);
    }
    const char* s0 = smem + cur * STAGE;
    V8 bh[TNW], bl[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
      bh[t] = *(const V8*)(s0 + b_off[t]);
      bl[t] = *(const V8*)(s0 + W_TILE + b_off[t]);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      V8 ah[HALF], al[HALF];
#pragma unroll
      for (int t = 0; t < HALF; ++t) {
        ah[t] = *(const V8*)(s0 + a_off[hf * HALF + t]);
        al[t] = *(const V8*)(s0 + A_TILE + a_off[hf * HALF + t]);
        if constexpr (RELU_A) { ah[t] = relu8(ah[t]); al[t] = relu8(al[t]); }
      }
#pragma unroll
      for (int i = 0; i < HALF; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
          f32x4 c = acc[hf * HALF + i][j];
          c = H::mfma16(al[i], bh[j], c);
          c = H::mfma16(ah[i], bl[j], c);
          c = H::mfma16(ah[i], bh[j], c);
          acc[hf * HALF + i][j] = c;
        }
    }
    __syncthreads();
  }

  // epilogue in 64-row passes through a [64][BN + 4] fp32 LDS tile
  constexpr int LDT = BN + 4;
  float* tileC = (float*)smem;
#pragma unroll
  for (int pass = 0; pass < BM / 64; ++pass) {
#pragma unroll
    for (int t = 0; t < TMW; ++t) {
      const int row0 = wm * (BM / 2) + t * 16;  // wave-uniform
      if (row0 / 64 == pass) {
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            tileC[(row0 - pass * 64 + fq * 4 + e) * LDT + wn * 64 + j * 16 + fr] = acc[t][j][e];
      }
    }
    __syncthreads();
    epilogue_tile<DT, 64, BN, 512>(p, tileC, m0 + pass * 64, n0, tid);
    __syncthreads();
  }
}

template <int DT, int BM, int BN, int WM, int WN>
int launch_tile(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  const size_t stage2 = 2 * (size_t)(BM + BN) * BK * 2, ctile = (size_t)BM * (BN + 4) * 4;
  const size_t lds = stage2 > ctile ? stage2 : ctile;
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
  const size_t lds = 128 * 132 * 4;  // max(2 stages x 32 KiB, fp32 epilogue tile)
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
  const bool conv = d.a_mode == VDN_A_CONV3X3;
  const int amode = conv ? (d.relu_a ? 2 : 1) : (d.relu_a ? 3 : 0);
  switch (amode) {
    case 0: hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 0, BM>), dim3(tiles), dim3(512), lds, s, d); break;
    case 1: hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 1, BM>), dim3(tiles), dim3(512), lds, s, d); break;
    case 2: hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 2, BM>), dim3(tiles), dim3(512), lds, s, d); break;
    default: hipLaunchKernelGGL((gemm_x3_big_kernel<DT, 3, BM>), dim3(tiles), dim3(512), lds, s, d); break;
  }
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

// Pick the M tile that wastes the fewest CU-rounds: cost = rounds(256 CUs) * BM, padded work included.
inline int pick_bm(int M, int N) {
  const int tn = (N + 255) / 256;
  int best = 0;
  double best_cost = 1e30;
  for (int bm : {256, 192, 128}) {
    const long tiles = (long)((M + bm - 1) / bm) * tn;
    const long rounds = (tiles + 255) / 256;
    const double cost = (double)rounds * bm * (bm == 128 ? 1.12 : (bm == 192 ? 1.04 : 1.0));  // smaller tiles feed worse
    if (cost < best_cost) { best_cost = cost; best = bm; }
  }
  return best;
}

template <int DT>
int launch_dt(const vdn_gemm_desc& d, hipStream_t s) {
  if (d.A_lo && d.W_lo && d.N >= 192 && (long)d.M * d.N >= 256L * 1024 &&
      (d.a_mode == VDN_A_CONV3X3 || (d.K & 31) == 0)) {
    const char* force = getenv("VDN_GEMM_BM");
    const int bm = force ? atoi(force) : pick_bm(d.M, d.N);
    if (bm == 256) return launch_x3_big<DT, 256>(d, s);
    if (bm == 192) return launch_x3_big<DT, 192>(d, s);
    if (bm == 128) return launch_x3_big<DT, 128>(d, s);
  }
  if (d.A_lo && d.W_lo && (d.store == VDN_ST_HEADS || d.N > 64)) return launch_x3<DT>(d, s);
  if (d.store == VDN_ST_HEADS || d.N > 64) return launch_tile<DT, 128, 128, 2, 2>(d, s);
  if (d.N > 32) return launch_tile<DT, 128, 64, 2, 2>(d, s);
  return launch_tile<DT, 128, 32, 4, 1>(d, s);
}

}  // namespace

extern "C" int vdn_gemm(const vdn_gemm_desc* dp, vdn_stream stream) {
  if (!dp) return VDN_EINVAL;
  const vdn_gemm_desc& d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.W || !d.zeros) return VDN_EINVAL;
  if (d.dt != VDN_F16 && d.dt != VDN_BF16) return VDN_EUNSUPPORTED;
  if ((d.K & 7) || (d.ldb & 63) || d.ldb < d.K) return VDN_EALIGN;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15) || ((uintptr_t)d.zeros & 15)) return VDN_EALIGN;
  if (d.a_mode == VDN_A_CONV3X3) {
    if ((d.cC & 7) || d.K != 9 * d.cC || d.M != d.cB * d.cOH * d.cOW) return VDN_EINVAL;
    if (d.cstride != 1 && d.cstride != 2) return VDN_EUNSUPPORTED;
    if (d.cOH != (d.cH + 2 - 3) / d.cstride + 1 || d.cOW != (d.cW + 2 - 3) / d.cstride + 1) return VDN_EINVAL;
  } else if (d.a_mode == VDN_A_PLAIN) {
    if ((d.lda & 7) || d.lda < d.K) return VDN_EALIGN;
  } else {
    return VDN_EUNSUPPORTED;
  }
  if (d.N & 3) return VDN_EALIGN;  // the epilogue stores 4 columns per lane
  if (((uintptr_t)d.A_lo | (uintptr_t)d.W_lo | (uintptr_t)d.out_lo) & 15) return VDN_EALIGN;
  if (d.out_lo && (d.out_dt == VDN_F32 || !d.out)) return VDN_EINVAL;
  if ((d.res1 && (d.ldr1 & 3)) || (d.res2 && (d.ldr2 & 3))) return VDN_EALIGN;
  if (((uintptr_t)d.bias | (uintptr_t)d.gamma | (uintptr_t)d.tab | (uintptr_t)d.res1 | (uintptr_t)d.res2) & 7) return VDN_EALIGN;
  switch (d.store) {
    case VDN_ST_PLAIN:
      if (!d.out || d.ldc < d.N) return VDN_EINVAL;
      if ((d.ldc & 3) || ((uintptr_t)d.out & 15)) return VDN_EALIGN;
      break;
    case VDN_ST_GEGLU:
      if (!d.out || (d.N & 31) || d.ldc < d.N / 2 || d.act || d.gamma || d.res1 || d.res2 || d.tab || d.rowadd)
        return VDN_EINVAL;
      if ((d.ldc & 3) || ((uintptr_t)d.out & 15)) return VDN_EALIGN;
      break;
    case VDN_ST_CONVT:
      if (!d.out || d.ck <= 0 || d.cout <= 0 || d.N != d.ck * d.ck * d.cout || d.M != d.cB * d.cH * d.cW || d.res1 ||
          d.res2 || d.tab)
        return VDN_EINVAL;
      if ((d.cout & 3) || ((uintptr_t)d.out & 15)) return VDN_EALIGN;
      break;
    case VDN_ST_HEADS:
      if (d.nsplit < 1 || d.nsplit > 3 || d.heads <= 0 || d.N != d.nsplit * d.heads * 64 || d.tokens <= 0 ||
          d.M % d.tokens || d.tok_off < 0 || d.tok_off + d.tokens > d.tpad || d.act || d.gamma || d.res1 || d.res2 ||
          d.tab || d.rowadd)
        return VDN_EINVAL;
      for (int i = 0; i < d.nsplit; ++i) {
        if (!d.dst[i] || ((uintptr_t)d.dst[i] & 15)) return VDN_EINVAL;
        if (d.rope[i] && (!d.rope_cs || d.rope_mod <= 0)) return VDN_EINVAL;
      }
      break;
    default:
      return VDN_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  return d.dt == VDN_F16 ? launch_dt<VDN_F16>(d, s) : launch_dt<VDN_BF16>(d, s);
}
