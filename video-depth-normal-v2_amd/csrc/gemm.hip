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

namespace {

constexpr int BK = 64;

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

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
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
        const T* src = ok ? a_row[i] + ((size_t)iy * p.cW + ix) * p.cC + ci : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (i * 4 + wave) * 1024),
                                         16, 0, 0);
      }
    } else {
      const bool ok = k < p.K;
#pragma unroll
      for (int i = 0; i < A_IT; ++i) {
        const T* src = ok ? a_row[i] + k : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sA + (i * 4 + wave) * 1024),
                                         16, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_row[i] + k),
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

  const int nk = p.ldb / BK;
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

  // ---- epilogue (fp32)
  const int nb = n0 + wn * WTN;  // first column of this wave's tile
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * WTM + i * 16 + fq * 4 + j;
      if (m >= p.M) continue;
      const float radd = p.rowadd ? p.rowadd[m] : 0.f;
      const size_t tab_row = p.tab ? (size_t)(m % p.tab_mod + p.tab_off) * p.N : 0;
      if (p.store == VDN_ST_GEGLU) {
        if constexpr ((TN & 1) == 0) {
#pragma unroll
        for (int t = 0; t < TN; t += 2) {
          const int nh = nb + t * 16 + fr, ng = nh + 16;
          if (ng >= p.N) continue;
          const float h = acc[i][t][j] + (p.bias ? p.bias[nh] : 0.f);
          const float g = acc[i][t + 1][j] + (p.bias ? p.bias[ng] : 0.f);
          const int nc = ((nb + t * 16) >> 1) + fr;
          store_from_float(p.out, p.out_dt, (size_t)m * p.ldc + nc, h * gelu_erf(g));
        }
        }
        continue;
      }
      if (p.store == VDN_ST_HEADS) {
        const int bt = m / p.tokens;
        const int tl = m - bt * p.tokens;
        const int tk = tl + p.tok_off;
        const int hc = p.heads * 64;
        const int split = nb / hc;
        const int head = (nb - split * hc) >> 6;
        if (nb >= p.N) continue;
        T* dst = (T*)p.dst[split];
        const size_t hb = ((size_t)bt * p.heads + head);
        if constexpr (TN == 4) {
          if (p.rope[split]) {
            const float* cs = p.rope_cs + (size_t)(tl % p.rope_mod) * 64;
#pragma unroll
            for (int t = 0; t < 4; t += 2) {
              const int pi = (t << 3) + fr;  // pair index 0..31
              float re = acc[i][t][j] + (p.bias ? p.bias[nb + t * 16 + fr] : 0.f);
              float im = acc[i][t + 1][j] + (p.bias ? p.bias[nb + t * 16 + 16 + fr] : 0.f);
              const float c = cs[2 * pi], s = cs[2 * pi + 1];
              const float ore = re * c - im * s, oim = re * s + im * c;
              if (p.transposed[split]) {
                dst[(hb * 64 + 2 * pi) * p.tpad + tk] = (T)ore;
                dst[(hb * 64 + 2 * pi + 1) * p.tpad + tk] = (T)oim;
              } else {
                typename H::V2 pr = {(T)ore, (T)oim};
                *(typename H::V2*)(dst + (hb * p.tpad + tk) * 64 + 2 * pi) = pr;
              }
            }
            continue;
          }
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          const int e = t * 16 + fr;
          const int n = nb + e;
          if (n >= p.N) continue;
          const float v = acc[i][t][j] + (p.bias ? p.bias[n] : 0.f);
          if (p.transposed[split]) dst[(hb * 64 + e) * p.tpad + tk] = (T)v;
          else dst[(hb * p.tpad + tk) * 64 + e] = (T)v;
        }
        continue;
      }
      // PLAIN / CONVT share the arithmetic
      size_t orow;
      int cb = 0, cy = 0, cx = 0;
      if (p.store == VDN_ST_CONVT) {
        const int hw = p.cH * p.cW;
        cb = m / hw;
        const int rem = m - cb * hw;
        cy = rem / p.cW;
        cx = rem - cy * p.cW;
        orow = 0;
      } else {
        orow = (size_t)(p.row_group > 0 ? m + (m / p.row_group + 1) * p.row_skip : m) * p.ldc;
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int n = nb + t * 16 + fr;
        if (n >= p.N) continue;
        float v = acc[i][t][j];
        if (p.bias) v += p.bias[n];
        v += radd;
        if (p.act == VDN_ACT_GELU) v = gelu_erf(v);
        else if (p.act == VDN_ACT_RELU) v = fmaxf(v, 0.f);
        if (p.gamma) v *= p.gamma[n];
        if (p.tab) v += p.tab[tab_row + n];
        if (p.res1) v += load_as_float(p.res1, p.res1_dt, (size_t)m * p.ldr1 + n);
        if (p.res2) v += load_as_float(p.res2, p.res2_dt, (size_t)m * p.ldr2 + n);
        size_t o;
        if (p.store == VDN_ST_CONVT) {
          const int kk = n / p.cout, co = n - kk * p.cout;
          const int ky = kk / p.ck, kx = kk - ky * p.ck;
          o = (((size_t)cb * (p.cH * p.ck) + cy * p.ck + ky) * (p.cW * p.ck) + cx * p.ck + kx) * p.cout + co;
        } else {
          o = orow + n;
        }
        store_from_float(p.out, p.out_dt, o, v);
      }
    }
  }
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
int launch_dt(const vdn_gemm_desc& d, hipStream_t s) {
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
  switch (d.store) {
    case VDN_ST_PLAIN:
      if (!d.out || d.ldc < d.N) return VDN_EINVAL;
      break;
    case VDN_ST_GEGLU:
      if (!d.out || (d.N & 31) || d.ldc < d.N / 2 || d.act || d.gamma || d.res1 || d.res2 || d.tab || d.rowadd)
        return VDN_EINVAL;
      break;
    case VDN_ST_CONVT:
      if (!d.out || d.ck <= 0 || d.cout <= 0 || d.N != d.ck * d.ck * d.cout || d.M != d.cB * d.cH * d.cW || d.res1 ||
          d.res2 || d.tab)
        return VDN_EINVAL;
      break;
    case VDN_ST_HEADS:
      if (d.nsplit < 1 || d.nsplit > 3 || d.heads <= 0 || d.N != d.nsplit * d.heads * 64 || d.tokens <= 0 ||
          d.M % d.tokens || d.tok_off < 0 || d.tok_off + d.tokens > d.tpad || d.act || d.gamma || d.res1 || d.res2 ||
          d.tab || d.rowadd)
        return VDN_EINVAL;
      for (int i = 0; i < d.nsplit; ++i) {
        if (!d.dst[i]) return VDN_EINVAL;
        if (d.rope[i] && (!d.rope_cs || d.rope_mod <= 0)) return VDN_EINVAL;
      }
      break;
    default:
      return VDN_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  return d.dt == VDN_F16 ? launch_dt<VDN_F16>(d, s) : launch_dt<VDN_BF16>(d, s);
}
