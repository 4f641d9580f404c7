// Attention kernels for gfx950 (wave64, v_mfma_f32_32x32x16_{f16,bf16}).
//
// flash_attn_kernel — head_dim 64, Nq x Nk scores never leave registers.
//   S^T = K Q^T is computed with the KEY on the accumulator row and the QUERY on the lane
//   (col = lane & 31), so a lane owns one query's scores: row max / sum are in-lane plus one
//   cross-half shuffle, and the exponentiated tile, converted pairwise to half, is directly the
//   B operand of O^T = V^T P^T (guide §3 'An accumulator tile as the next MFMA's operand').
//   The K rows are fed through the bit-2<->bit-3 row permutation so that the accumulator's
//   k-order is the natural key order and V^T fragments are one contiguous 16-byte LDS read.
//   K and V^T tiles (64 keys) arrive by LDS-DMA into a double buffer shared by the 4 waves
//   (4 x 32 = 128 queries per workgroup); chunk ^ ((row>>1)&7) source-side swizzle keeps the
//   ds_read_b128 fragment reads bank-conflict free.
//   SPLIT: every operand comes as (hi, lo) 16-bit planes and each product is accumulated as
//   hi*hi + hi*lo + lo*hi (fp32-faithful "x3" mode, see include/vdn.h).
//
// temporal_attn_kernel — <= 32 frames per (pixel, head): one wave per sequence, fragments loaded
//   straight from global memory (no LDS), same accumulator-as-operand chaining.
#include "common.hpp"

namespace {

__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
  return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
// key offset (0..31) held by accumulator register `reg` of lane-half `h` after the row permutation
__device__ __forceinline__ int acc_key(int reg, int h) {
  return (reg & 3) + 4 * ((reg >> 2) & 1) + 8 * h + 16 * (reg >> 3);
}

#define GLDS16(src, dst)                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int DT, bool SPLIT>
__global__ __launch_bounds__(256) void flash_attn_kernel(const typename Half<DT>::T* __restrict__ Q,
                                                         const typename Half<DT>::T* __restrict__ K,
                                                         const typename Half<DT>::T* __restrict__ Vt,
                                                         typename Half<DT>::T* __restrict__ out,
                                                         const typename Half<DT>::T* __restrict__ Ql,
                                                         const typename Half<DT>::T* __restrict__ Kl,
                                                         const typename Half<DT>::T* __restrict__ Vtl,
                                                         typename Half<DT>::T* __restrict__ outl, int H, int nq,
                                                         int nq_pad, int nk, int nk_pad, float scale_log2) {
  using HT = Half<DT>;
  using T = typename HT::T;
  using V8 = typename HT::V8;
  constexpr int TILE = 8192;         // 64 rows x 128 B
  constexpr int NT = SPLIT ? 4 : 2;  // tiles per stage: K, Vt (, K_lo, Vt_lo)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, XCD-aware: the q-blocks of one (batch, head) are consecutive logical ids and therefore run
  // on ONE XCD, so its K / V^T tiles are fetched from HBM once and re-read by the other q-blocks from
  // that XCD's L2 (rocprofv3 FETCH_SIZE showed 7.6x over-fetch with a round-robin 2-D grid).
  const int nqb = (nq + 127) >> 7;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = logical / nqb;
  const int q0 = (logical - bh * nqb) * 128 + wave * 32;
  const int r = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16 ks + 8 h + j]
  V8 qf[4], ql[4];
  {
    int q = q0 + r;
    q = q < nq ? q : nq - 1;
    const size_t qo = ((size_t)bh * nq_pad + q) * 64 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const V8*)(Q + qo + 16 * ks);
      if constexpr (SPLIT) ql[ks] = *(const V8*)(Ql + qo + 16 * ks);
    }
  }

  // ---- staging: 8 pieces (1 KiB = 8 rows) per tile and operand, 2 per wave
  const int lr = lane >> 3;
  const size_t kbase = (size_t)bh * nk_pad * 64;
  const size_t vbase = (size_t)bh * 64 * nk_pad;
  auto stage = [&](int buf, int t) {
    char* sK = smem + buf * NT * TILE;
    char* sV = sK + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 4 * i;
      const int row = pc * 8 + lr;
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const size_t ko = kbase + (size_t)(t * 64 + row) * 64 + c * 8;
      const size_t vo = vbase + (size_t)row * nk_pad + t * 64 + c * 8;
      GLDS16(K + ko, sK + pc * 1024);
      GLDS16(Vt + vo, sV + pc * 1024);
      if constexpr (SPLIT) {
        GLDS16(Kl + ko, sK + 2 * TILE + pc * 1024);
        GLDS16(Vtl + vo, sV + 2 * TILE + pc * 1024);
      }
    }
  };

  // ---- fragment read offsets
  int k_off[2][4], v_off[2][4];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int row = kb * 32 + perm23(r);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_off[kb][ks] = row * 128 + (((2 * ks + h) ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const int row = db * 32 + r;
#pragma unroll
    for (int c = 0; c < 4; ++c) v_off[db][c] = row * 128 + (((2 * c + h) ^ ((row >> 1) & 7)) << 4);
  }

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (nk + 63) >> 6;
  stage(0, 0);
  stage_barrier();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    const char* sK = smem + cur * NT * TILE;
    const char* sV = sK + TILE;

    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 a = *(const V8*)(sK + k_off[kb][ks]);
        s[kb] = HT::mfma32(a, qf[ks], s[kb]);
        if constexpr (SPLIT) {
          const V8 al = *(const V8*)(sK + 2 * TILE + k_off[kb][ks]);
          s[kb] = HT::mfma32(a, ql[ks], s[kb]);
          s[kb] = HT::mfma32(al, qf[ks], s[kb]);
        }
      }
    }
    // scale into the log2 domain, mask the ragged last tile
    const bool tail = (t + 1) * 64 > nk;
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = s[kb][i] * scale_log2;
        if (tail) v = (t * 64 + kb * 32 + acc_key(i, h) < nk) ? v : -INFINITY;
        s[kb][i] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float ls = 0.f;
    V8 pf[2][2], pl[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[kb][i] - m_new);
        ls += pv;
        if constexpr (SPLIT) {
          T a, b;
          split_rtz(pv, a, b);
          pf[kb][i >> 3][i & 7] = a;
          pl[kb][i >> 3][i & 7] = b;
        } else {
          pf[kb][i >> 3][i & 7] = (T)pv;
        }
      }
    l_run = l_run * alpha + ls;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const V8 a = *(const V8*)(sV + v_off[db][c]);
        o[db] = HT::mfma32(a, pf[c >> 1][c & 1], o[db]);
        if constexpr (SPLIT) {
          const V8 al = *(const V8*)(sV + 2 * TILE + v_off[db][c]);
          o[db] = HT::mfma32(a, pl[c >> 1][c & 1], o[db]);
          o[db] = HT::mfma32(al, pf[c >> 1][c & 1], o[db]);
        }
      }
    stage_barrier();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < nq) {
    const int b = bh / H, hd = bh - b * H;
    const size_t oo = (((size_t)b * nq + q) * H + hd) * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename HT::V4 v, vl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = o[db][4 * g + e] * inv;
          if constexpr (SPLIT) {
            T a, b2;
            split_rtz(x, a, b2);
            v[e] = a;
            vl[e] = b2;
          } else {
            v[e] = (T)x;
          }
        }
        *(typename HT::V4*)(out + oo + db * 32 + 8 * g + 4 * h) = v;
        if constexpr (SPLIT) *(typename HT::V4*)(outl + oo + db * 32 + 8 * g + 4 * h) = vl;
      }
  }
}

template <int DT, bool SPLIT>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const typename Half<DT>::T* __restrict__ qkv,
                                                            typename Half<DT>::T* __restrict__ out,
                                                            const typename Half<DT>::T* __restrict__ qkv_lo,
                                                            typename Half<DT>::T* __restrict__ out_lo, int nseq,
                                                            int Tn, int D, int c, int heads, float scale_log2) {
  using HT = Half<DT>;
  using T = typename HT::T;
  using V8 = typename HT::V8;
  const int lane = threadIdx.x & 63;
  const int seq = blockIdx.x * 4 + (threadIdx.x >> 6);  // (b, d, head)
  if (seq >= nseq) return;
  const int head = seq % heads;
  const int bd = seq / heads;
  const int d = bd % D, b = bd / D;
  const int dh = c / heads;
  const int r = lane & 31, h = lane >> 5;
  const size_t rs = (size_t)D * 3 * c;  // stride between frames
  const size_t boff = ((size_t)b * Tn * D + d) * 3 * c + head * dh;
  const T* base = qkv + boff;
  const T* base_lo = SPLIT ? qkv_lo + boff : nullptr;

  // ---- S^T = K Q^T
  const int fq = r < Tn ? r : Tn - 1;
  const int pk = perm23(r);
  const int fk = pk < Tn ? pk : Tn - 1;
  f32x16 s;
#pragma unroll
  for (int i = 0; i < 16; ++i) s[i] = 0.f;
  const int nks = (dh + 15) >> 4;
  for (int ks = 0; ks < nks; ++ks) {
    const int e0 = 16 * ks + 8 * h;
    V8 a, bq, al, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (T)0.f; bq[j] = (T)0.f; al[j] = (T)0.f; bl[j] = (T)0.f; }
    if (e0 < dh) {
      bq = *(const V8*)(base + fq * rs + e0);
      a = *(const V8*)(base + fk * rs + c + e0);
      if constexpr (SPLIT) {
        bl = *(const V8*)(base_lo + fq * rs + e0);
        al = *(const V8*)(base_lo + fk * rs + c + e0);
      }
    }
    s = HT::mfma32(a, bq, s);
    if constexpr (SPLIT) {
      s = HT::mfma32(a, bl, s);
      s = HT::mfma32(al, bq, s);
    }
  }
  float mx = -1e30f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = s[i] * scale_log2;
    v = (acc_key(i, h) < Tn) ? v : -INFINITY;
    s[i] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float ls = 0.f;
  V8 pf[2], pl[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float pv = __builtin_amdgcn_exp2f(s[i] - mx);
    ls += pv;
    if constexpr (SPLIT) {
      T a2, b2;
      split_rtz(pv, a2, b2);
      pf[i >> 3][i & 7] = a2;
      pl[i >> 3][i & 7] = b2;
    } else {
      pf[i >> 3][i & 7] = (T)pv;
    }
  }
  ls += __shfl_xor(ls, 32);
  const float inv = 1.0f / ls;

  // ---- O^T = V^T P^T, 32 output dims per pass; V^T fragments gathered element-wise (tiny op)
  const T* vb = base + 2 * c;
  const T* vb_lo = SPLIT ? base_lo + 2 * c : nullptr;
  const size_t ooff = ((size_t)b * Tn * D + d) * c + head * dh;
  T* ob = out + ooff;
  T* ob_lo = SPLIT ? out_lo + ooff : nullptr;
  const size_t os = (size_t)D * c;
  const int neb = (dh + 31) >> 5;
  for (int eb = 0; eb < neb; ++eb) {
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    const int e = eb * 32 + r;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      V8 a, al;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 16 * sp + 8 * h + j;
        const bool ok = key < Tn && e < dh;
        a[j] = ok ? vb[key * rs + e] : (T)0.f;
        if constexpr (SPLIT) al[j] = ok ? vb_lo[key * rs + e] : (T)0.f;
      }
      o = HT::mfma32(a, pf[sp], o);
      if constexpr (SPLIT) {
        o = HT::mfma32(a, pl[sp], o);
        o = HT::mfma32(al, pf[sp], o);
      }
    }
    if (r < Tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int e0 = eb * 32 + 8 * g + 4 * h;
        if (e0 < dh) {
          typename HT::V4 v, vl;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x = o[4 * g + q] * inv;
            if constexpr (SPLIT) {
              T a2, b2;
              split_rtz(x, a2, b2);
              v[q] = a2;
              vl[q] = b2;
            } else {
              v[q] = (T)x;
            }
          }
          *(typename HT::V4*)(ob + r * os + e0) = v;
          if constexpr (SPLIT) *(typename HT::V4*)(ob_lo + r * os + e0) = vl;
        }
      }
    }
  }
}

template <int DT>
int flash_launch(const void* Q, const void* K, const void* Vt, void* out, const void* Ql, const void* Kl, const void* Vtl,
                 void* outl, int B, int H, int nq, int nq_pad, int nk, int nk_pad, float sl2, hipStream_t s) {
  using T = typename Half<DT>::T;
  const dim3 grid(((nq + 127) / 128) * B * H);
  if (Ql)
    hipLaunchKernelGGL((flash_attn_kernel<DT, true>), grid, dim3(256), 65536, s, (const T*)Q, (const T*)K, (const T*)Vt,
                       (T*)out, (const T*)Ql, (const T*)Kl, (const T*)Vtl, (T*)outl, H, nq, nq_pad, nk, nk_pad, sl2);
  else
    hipLaunchKernelGGL((flash_attn_kernel<DT, false>), grid, dim3(256), 32768, s, (const T*)Q, (const T*)K, (const T*)Vt,
                       (T*)out, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, H, nq, nq_pad, nk,
                       nk_pad, sl2);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <int DT>
int temporal_launch(const void* qkv, void* out, const void* qkv_lo, void* out_lo, int nseq, int T, int D, int c, int heads,
                    float sl2, hipStream_t s) {
  using TT = typename Half<DT>::T;
  const dim3 grid((nseq + 3) / 4);
  if (qkv_lo)
    hipLaunchKernelGGL((temporal_attn_kernel<DT, true>), grid, dim3(256), 0, s, (const TT*)qkv, (TT*)out,
                       (const TT*)qkv_lo, (TT*)out_lo, nseq, T, D, c, heads, sl2);
  else
    hipLaunchKernelGGL((temporal_attn_kernel<DT, false>), grid, dim3(256), 0, s, (const TT*)qkv, (TT*)out,
                       (const TT*)nullptr, (TT*)nullptr, nseq, T, D, c, heads, sl2);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

}  // namespace

extern "C" int vdn_flash_attn(int dt, const void* Q, const void* K, const void* Vt, void* out, const void* Q_lo,
                              const void* K_lo, const void* Vt_lo, void* out_lo, int B, int H, int nq, int nq_pad, int nk,
                              int nk_pad, float scale, vdn_stream stream) {
  if (!Q || !K || !Vt || !out || B <= 0 || H <= 0 || nq <= 0 || nk <= 0) return VDN_EINVAL;
  if (nq_pad < nq || nk_pad < nk || (nk_pad & 63)) return VDN_EALIGN;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)out) & 15) return VDN_EALIGN;
  const int nlo = (Q_lo != nullptr) + (K_lo != nullptr) + (Vt_lo != nullptr) + (out_lo != nullptr);
  if (nlo != 0 && nlo != 4) return VDN_EINVAL;  // split precision is all-or-nothing here
  if (((uintptr_t)Q_lo | (uintptr_t)K_lo | (uintptr_t)Vt_lo | (uintptr_t)out_lo) & 15) return VDN_EALIGN;
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    return flash_launch<VDN_F16>(Q, K, Vt, out, Q_lo, K_lo, Vt_lo, out_lo, B, H, nq, nq_pad, nk, nk_pad, sl2, s);
  if (dt == VDN_BF16)
    return flash_launch<VDN_BF16>(Q, K, Vt, out, Q_lo, K_lo, Vt_lo, out_lo, B, H, nq, nq_pad, nk, nk_pad, sl2, s);
  return VDN_EUNSUPPORTED;
}

extern "C" int vdn_temporal_attn(int dt, const void* qkv, void* out, const void* qkv_lo, void* out_lo, int Bv, int T,
                                 int D, int c, int heads, float scale, vdn_stream stream) {
  if (!qkv || !out || Bv <= 0 || T <= 0 || T > 32 || D <= 0 || heads <= 0 || c % heads) return VDN_EINVAL;
  if ((qkv_lo == nullptr) != (out_lo == nullptr)) return VDN_EINVAL;
  const int dh = c / heads;
  if ((dh & 7) || dh > 256 || (c & 7)) return VDN_EALIGN;
  if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)qkv_lo | (uintptr_t)out_lo) & 15) return VDN_EALIGN;
  const int nseq = Bv * D * heads;
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16) return temporal_launch<VDN_F16>(qkv, out, qkv_lo, out_lo, nseq, T, D, c, heads, sl2, s);
  if (dt == VDN_BF16) return temporal_launch<VDN_BF16>(qkv, out, qkv_lo, out_lo, nseq, T, D, c, heads, sl2, s);
  return VDN_EUNSUPPORTED;
}
