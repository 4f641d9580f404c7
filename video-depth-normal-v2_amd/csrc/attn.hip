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

template <int DT>
__global__ __launch_bounds__(256) void flash_attn_kernel(const typename Half<DT>::T* __restrict__ Q,
                                                         const typename Half<DT>::T* __restrict__ K,
                                                         const typename Half<DT>::T* __restrict__ Vt,
                                                         typename Half<DT>::T* __restrict__ out, int H, int nq,
                                                         int nq_pad, int nk, int nk_pad, float scale_log2) {
  using HT = Half<DT>;
  using T = typename HT::T;
  using V8 = typename HT::V8;
  constexpr int TILE = 8192;  // 64 rows x 128 B
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K tile | Vt tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int r = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16 ks + 8 h + j]
  V8 qf[4];
  {
    int q = q0 + r;
    q = q < nq ? q : nq - 1;
    const T* qp = Q + ((size_t)bh * nq_pad + q) * 64 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const V8*)(qp + 16 * ks);
  }

  // ---- staging: 8 pieces (1 KiB = 8 rows) per tile and operand, 2 per wave
  const int lr = lane >> 3;
  const T* Kb = K + (size_t)bh * nk_pad * 64;
  const T* Vb = Vt + (size_t)bh * 64 * nk_pad;
  auto stage = [&](int buf, int t) {
    char* sK = smem + buf * 2 * TILE;
    char* sV = sK + TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pc = wave + 4 * i;
      const int row = pc * 8 + lr;
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const T* ks = Kb + (size_t)(t * 64 + row) * 64 + c * 8;
      const T* vs = Vb + (size_t)row * nk_pad + t * 64 + c * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ks,
                                       (__attribute__((address_space(3))) void*)(sK + pc * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)vs,
                                       (__attribute__((address_space(3))) void*)(sV + pc * 1024), 16, 0, 0);
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
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) stage(cur ^ 1, t + 1);
    const char* sK = smem + cur * 2 * TILE;
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
    V8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[kb][i] - m_new);
        ls += pv;
        pf[kb][i >> 3][i & 7] = (T)pv;
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
      }
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < nq) {
    const int b = bh / H, hd = bh - b * H;
    T* op = out + (((size_t)b * nq + q) * H + hd) * 64;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename HT::V4 v = {(T)(o[db][4 * g] * inv), (T)(o[db][4 * g + 1] * inv), (T)(o[db][4 * g + 2] * inv),
                             (T)(o[db][4 * g + 3] * inv)};
        *(typename HT::V4*)(op + db * 32 + 8 * g + 4 * h) = v;
      }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const typename Half<DT>::T* __restrict__ qkv,
                                                            typename Half<DT>::T* __restrict__ out, int nseq, int Tn,
                                                            int D, int c, int heads, float scale_log2) {
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
  const T* base = qkv + ((size_t)b * Tn * D + d) * 3 * c + head * dh;

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
    V8 a, bq;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (T)0.f; bq[j] = (T)0.f; }
    if (e0 < dh) {
      bq = *(const V8*)(base + fq * rs + e0);
      a = *(const V8*)(base + fk * rs + c + e0);
    }
    s = HT::mfma32(a, bq, s);
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
  V8 pf[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float pv = __builtin_amdgcn_exp2f(s[i] - mx);
    ls += pv;
    pf[i >> 3][i & 7] = (T)pv;
  }
  ls += __shfl_xor(ls, 32);
  const float inv = 1.0f / ls;

  // ---- O^T = V^T P^T, 32 output dims per pass; V^T fragments gathered element-wise (tiny op)
  const T* vb = base + 2 * c;
  T* ob = out + ((size_t)b * Tn * D + d) * c + head * dh;
  const size_t os = (size_t)D * c;
  const int neb = (dh + 31) >> 5;
  for (int eb = 0; eb < neb; ++eb) {
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    const int e = eb * 32 + r;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      V8 a;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 16 * sp + 8 * h + j;
        a[j] = (key < Tn && e < dh) ? vb[key * rs + e] : (T)0.f;
      }
      o = HT::mfma32(a, pf[sp], o);
    }
    if (r < Tn) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int e0 = eb * 32 + 8 * g + 4 * h;
        if (e0 < dh) {
          typename HT::V4 v = {(T)(o[4 * g] * inv), (T)(o[4 * g + 1] * inv), (T)(o[4 * g + 2] * inv),
                               (T)(o[4 * g + 3] * inv)};
          *(typename HT::V4*)(ob + r * os + e0) = v;
        }
      }
    }
  }
}

}  // namespace

extern "C" int vdn_flash_attn(int dt, const void* Q, const void* K, const void* Vt, void* out, int B, int H, int nq,
                              int nq_pad, int nk, int nk_pad, float scale, vdn_stream stream) {
  if (!Q || !K || !Vt || !out || B <= 0 || H <= 0 || nq <= 0 || nk <= 0) return VDN_EINVAL;
  if (nq_pad < nq || nk_pad < nk || (nk_pad & 63)) return VDN_EALIGN;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)out) & 15) return VDN_EALIGN;
  const dim3 grid((nq + 127) / 128, B * H);
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(flash_attn_kernel<VDN_F16>, grid, dim3(256), 32768, s, (const _Float16*)Q, (const _Float16*)K,
                       (const _Float16*)Vt, (_Float16*)out, H, nq, nq_pad, nk, nk_pad, sl2);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(flash_attn_kernel<VDN_BF16>, grid, dim3(256), 32768, s, (const __bf16*)Q, (const __bf16*)K,
                       (const __bf16*)Vt, (__bf16*)out, H, nq, nq_pad, nk, nk_pad, sl2);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

extern "C" int vdn_temporal_attn(int dt, const void* qkv, void* out, int Bv, int T, int D, int c, int heads,
                                 float scale, vdn_stream stream) {
  if (!qkv || !out || Bv <= 0 || T <= 0 || T > 32 || D <= 0 || heads <= 0 || c % heads) return VDN_EINVAL;
  const int dh = c / heads;
  if ((dh & 7) || dh > 256 || (c & 7)) return VDN_EALIGN;
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return VDN_EALIGN;
  const int nseq = Bv * D * heads;
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    hipLaunchKernelGGL(temporal_attn_kernel<VDN_F16>, dim3((nseq + 3) / 4), dim3(256), 0, s, (const _Float16*)qkv,
                       (_Float16*)out, nseq, T, D, c, heads, sl2);
  else if (dt == VDN_BF16)
    hipLaunchKernelGGL(temporal_attn_kernel<VDN_BF16>, dim3((nseq + 3) / 4), dim3(256), 0, s, (const __bf16*)qkv,
                       (__bf16*)out, nseq, T, D, c, heads, sl2);
  else
    return VDN_EUNSUPPORTED;
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
