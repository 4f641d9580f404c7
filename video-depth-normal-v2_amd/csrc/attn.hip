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
//   PV2 (SPLIT only, the default): P, which is born in registers, is rounded ONCE to 16 bits (p~) and O accumulates
//   p~ (V_hi + V_lo): 2 products instead of 3 and no lo split of P on the VALU. The row sum l is accumulated from the
//   SAME rounded p~ (one v_dot2 per pair), so O / l is an exact convex combination of the (21-bit) V rows with
//   weights p~ / sum p~: the rounding perturbs each weight by <= 2^-11 relative and any part of it common to a
//   row cancels in the normalisation (DESIGN.md §3 has the measured end-to-end effect).
//   QK8 (with PV2, fp16 only): the two CROSS terms of S^T = K Q^T run on the block-scaled 8-bit MFMA
//   (v_mfma_scale_f32_32x32x64_f8f6f4, e5m2 operands, K = 64 = the whole head dimension in one instruction, 2.3x the
//   fp16 rate measured): S^T = K_hi Q_hi^T [4 fp16 MFMAs] + K8 (Q_lo8)^T 2^-10 + K_lo8 (Q8)^T 2^-10 [2 MFMAs], where
//   X8 = e5m2(X) and X_lo8 = e5m2((X - X_hi) 2^10) are written by the projection GEMM's epilogue (vdn_gemm_desc.dst8)
//   and the 2^-10 is the MFMA's E8M0 scale operand. A cross term is 2^-11 of the product, so the 3-bit e5m2 significand
//   leaves 2^-14 per term: logits good to ~1e-5 (tools/micro/mfma_scale_probe.hip pins layout, scale and rate).
//
// temporal_attn_kernel — <= 32 frames per (pixel, head): one wave per sequence, fragments loaded
//   straight from global memory (no LDS), same accumulator-as-operand chaining.
#include "common.hpp"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

__device__ __forceinline__ bool lane_hi() { return (threadIdx.x & 32) != 0; }
__device__ __forceinline__ int perm23(int i) {  // swap bits 2 and 3
  return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1);
}
// key offset (0..31) held by accumulator register `reg` of lane-half `h` after the row permutation
__device__ __forceinline__ int acc_key(int reg, int h) {
  return (reg & 3) + 4 * ((reg >> 2) & 1) + 8 * h + 16 * (reg >> 3);
}

// value held by the same row's lane in the other 32-lane half (v_permlane32_swap: VALU, no LDS round trip)
__device__ __forceinline__ float xhalf(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (lane_hi() ? r[0] : r[1]));
}

template <class F, int... J>
__device__ __forceinline__ void for_each_slot_impl(F& f, std::integer_sequence<int, J...>) {
  (f(std::integral_constant<int, J>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void for_each_slot(F& f) {
  for_each_slot_impl(f, std::make_integer_sequence<int, N>{});
}

typedef int i32x8 __attribute__((ext_vector_type(8)));

// Timing ablations for tools/attn_ablate.sh variant builds only (results are wrong with any bit set; the shipped
// library is built without the macro): 1 no LDS-DMA in the loop, 2 no workgroup barrier, 4 no softmax VALU, 8 no MFMA, 16 no fragment reads from LDS, 32 one workgroup per CU, 64 s_memtime stamps, 128 softmax without its fma
// (96 KB of LDS requested: one wave per SIMD).
#ifndef VDN_ATTN_ABL
#define VDN_ATTN_ABL 0
#endif

template <int DT, bool SPLIT, bool PV2, bool QK8>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn_kernel(const typename Half<DT>::T* __restrict__ Q,
                                                         const typename Half<DT>::T* __restrict__ K,
                                                         const typename Half<DT>::T* __restrict__ Vt,
                                                         typename Half<DT>::T* __restrict__ out,
                                                         const typename Half<DT>::T* __restrict__ Ql,
                                                         const typename Half<DT>::T* __restrict__ Kl,
                                                         const typename Half<DT>::T* __restrict__ Vtl,
                                                         typename Half<DT>::T* __restrict__ outl,
                                                         const uint8_t* __restrict__ Q8, const uint8_t* __restrict__ K8,
                                                         int H, int nq, int nq_pad, int nk, int nk_pad, float scale_log2) {
  static_assert(!QK8 || (SPLIT && PV2 && DT == VDN_F16), "8-bit cross terms: split fp16 planes, 2-product P V");
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
  i32x8 q8h, q8l;  // QK8: B operands of the 32x32x64 8-bit MFMA: Q8[q][32 h + j], j = 0..31
  {
    int q = q0 + r;
    q = q < nq ? q : nq - 1;
    const size_t qo = ((size_t)bh * nq_pad + q) * 64 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = *(const V8*)(Q + qo + 16 * ks);
      if constexpr (SPLIT && !QK8) ql[ks] = *(const V8*)(Ql + qo + 16 * ks);
    }
    if constexpr (QK8) {
      const uint8_t* q8 = Q8 + ((size_t)bh * nq_pad + q) * 128 + 32 * h;
      const u32x4 a0 = *(const u32x4*)q8, a1 = *(const u32x4*)(q8 + 16), b0 = *(const u32x4*)(q8 + 64), b1 = *(const u32x4*)(q8 + 80);
#pragma unroll
      for (int e = 0; e < 4; ++e) { q8h[e] = (int)a0[e]; q8h[4 + e] = (int)a1[e]; q8l[e] = (int)b0[e]; q8l[4 + e] = (int)b1[e]; }
    }
  }

  // ---- staging: 8 pieces (1 KiB = 8 rows) per tile and operand, 2 per wave, as `buffer_load_dwordx4 ... offen lds`:
  // the (batch, head)'s plane is a buffer resource (4 SGPRs, hardware range check), the lane's place in a piece a
  // loop-invariant 32-bit VGPR offset and the tile a SCALAR offset — no vector address arithmetic per tile (the flat
  // global_load_lds form cost 16 v_lshl_add_u64 per tile and 16 VGPRs of per-lane 64-bit pointers).
  const int lr = lane >> 3;
  const unsigned k_bytes = (unsigned)nk_pad * 64 * sizeof(T);  // one plane of one (batch, head): K rows, V^T rows, K8 rows
  const auto rsrc = [&](const void* base, size_t off) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + off), 0, (int)k_bytes, 0x00020000);
  };
  const size_t plane_off = (size_t)bh * k_bytes;
  const __amdgpu_buffer_rsrc_t rK = rsrc(K, plane_off), rV = rsrc(Vt, plane_off);
  const __amdgpu_buffer_rsrc_t rKl = rsrc(QK8 ? (const void*)K8 : (const void*)(SPLIT ? Kl : K), plane_off);
  const __amdgpu_buffer_rsrc_t rVl = rsrc(SPLIT ? Vtl : Vt, plane_off);
  int koff[2], voff[2];  // byte offsets
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave + 4 * i) * 8 + lr;
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    koff[i] = (row * 64 + c * 8) * (int)sizeof(T);
    voff[i] = (row * nk_pad + c * 8) * (int)sizeof(T);
  }
#define BLDS16(rs, voffset, soffset, dst) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 16, voffset, soffset, 0, 0)
  // piece i (0, 1) of a tile of K (hi plane; second plane = K_lo rows or the 8-bit rows: both 128 B per key) / of V^T
  auto stage_k_piece = [&](int buf, int t, int i, bool second) {
    char* sK = smem + buf * NT * TILE + (second ? 2 * TILE : 0);
    BLDS16(second ? rKl : rK, koff[i], t * (64 * 64 * (int)sizeof(T)), sK + (wave + 4 * i) * 1024);
  };
  auto stage_v_piece = [&](int buf, int t, int i, bool second) {
    char* sV = smem + buf * NT * TILE + TILE + (second ? 2 * TILE : 0);
    BLDS16(second ? rVl : rV, voff[i], t * (64 * (int)sizeof(T)), sV + (wave + 4 * i) * 1024);
  };
  auto stage_k = [&](int buf, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stage_k_piece(buf, t, i, false);
      if constexpr (SPLIT) stage_k_piece(buf, t, i, true);
    }
  };
  auto stage_v = [&](int buf, int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      stage_v_piece(buf, t, i, false);
      if constexpr (SPLIT) stage_v_piece(buf, t, i, true);
    }
  };

  // ---- fragment read offsets. Row r (+32 kb) of a tile, 16-byte chunk (2 ks + h) ^ swz(row): the kb / db
  // term is a compile-time immediate and the ks / c term an XOR with (ks << 5), so ONE register per operand
  // is kept (not 8) and the address of each fragment pair costs one v_xor (the loop is register-bound).
  const int k_base = perm23(r) * 128 + ((h ^ ((perm23(r) >> 1) & 7)) << 4);
  const int v_base = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  auto k_addr = [&](int kb, int ks) { return kb * 4096 + (k_base ^ (ks << 5)); };
  // 8-bit A operand of key row perm23(r): bytes 32 h .. 32 h + 31 of the e5m2(K) half (16-byte chunks 2h, 2h+1 of the
  // row) or of the remainder half (chunks 4+2h, 5+2h), same XOR swizzle: one base, the chunk as an XOR immediate
  const int k8_base = perm23(r) * 128 + (((2 * h) ^ ((perm23(r) >> 1) & 7)) << 4);
  auto k8_read = [&](const char* sK, int kb, int lo) {
    const char* p = sK + 2 * TILE + kb * 4096;
    const u32x4 a0 = *(const u32x4*)(p + (k8_base ^ ((4 * lo) << 4))), a1 = *(const u32x4*)(p + (k8_base ^ ((4 * lo + 1) << 4)));
    i32x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (int)a0[e]; v[4 + e] = (int)a1[e]; }
    return v;
  };
  // cross terms of one 32-key block on the scaled 8-bit MFMA (e5m2 x e5m2; the remainder planes carry 2^10)
  auto cross_hl = [&](const i32x8& k8, f32x16 c) {  // K8 (Q_lo8)^T
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8, q8l, c, 1, 1, 0, 127, 0, VDN_LO8_E8M0);
  };
  auto cross_lh = [&](const i32x8& k8l, f32x16 c) {  // K_lo8 (Q8)^T
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l, q8h, c, 1, 1, 0, VDN_LO8_E8M0, 0, 127);
  };
  auto v_addr = [&](int db, int c) { return db * 4096 + (v_base ^ (c << 5)); };

  f32x16 o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;  // running max in the RAW score domain; the scale is folded into the exp2 argument

  // S^T of one 64-key tile from K buffer `buf`
  auto qk = [&](int buf, f32x16 (&sc)[2]) {
    const char* sK = smem + buf * NT * TILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[kb][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const V8 a = *(const V8*)(sK + k_addr(kb, ks));
        sc[kb] = HT::mfma32(a, qf[ks], sc[kb]);
        if constexpr (SPLIT && !QK8) {
          const V8 al = *(const V8*)(sK + 2 * TILE + k_addr(kb, ks));
          sc[kb] = HT::mfma32(a, ql[ks], sc[kb]);
          sc[kb] = HT::mfma32(al, qf[ks], sc[kb]);
        }
      }
      if constexpr (QK8) {
        sc[kb] = cross_hl(k8_read(sK, kb, 0), sc[kb]);
        sc[kb] = cross_lh(k8_read(sK, kb, 1), sc[kb]);
      }
    }
  };
  // O^T += V^T P^T of one tile from V buffer `buf`
  auto pv = [&](int buf, const V8 (&pf)[2][2], const V8 (&pl)[2][2]) {
    const char* sV = smem + buf * NT * TILE + TILE;
#pragma unroll
    for (int c = 0; c < 4; ++c)  // c outer: a P fragment is dead after its 6 MFMAs (registers for the next tile's P)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const V8 a = *(const V8*)(sV + v_addr(db, c));
        o[db] = HT::mfma32(a, pf[c >> 1][c & 1], o[db]);
        if constexpr (SPLIT) {
          const V8 al = *(const V8*)(sV + 2 * TILE + v_addr(db, c));
          if constexpr (!PV2) o[db] = HT::mfma32(a, pl[c >> 1][c & 1], o[db]);
          o[db] = HT::mfma32(al, pf[c >> 1][c & 1], o[db]);
        }
      }
  };

  // Three-stage software pipeline, all in ONE basic block per iteration so the scheduler can slot the
  // softmax VALU work into the MFMA gaps of the SAME wave (an in-order wave only overlaps the two pipes
  // when they alternate in program order; PMC before: VALU busy 46 % + MFMA busy 53 % = 99 %, i.e. serial):
  //     iteration t:   S(t+1) = K_{t+1} Q^T   ||   P(t) = softmax-numerator(S(t))   ||   O += V_{t-1} P(t-1)
  // LDS ring (2 buffers each): iteration t reads K_{t+1} and V_{t-1}; its DMA fills K_{t+2} (buffer of K_t,
  // consumed in iteration t-1) and V_t (buffer of V_{t-2}, consumed in iteration t-1).
  // Lazy rescale: the reference point m_run moves only when some row's tile maximum exceeds it by more than
  // 2^LAZY (P stays <= 2^LAZY: exact in fp32 / split fp16; the 1/l normalisation makes the result
  // independent of the reference), so in steady state the O accumulators are never touched by the VALU.
  constexpr float LAZY = 6.0f;
  const int nt = (nk + 63) >> 6;
  f32x16 s[2];
  V8 pf[2][2], pl[2][2];
  auto iter = [&](int t, auto has_prev_c, auto has_next_c) {
    constexpr bool HAS_PREV = decltype(has_prev_c)::value, HAS_NEXT = decltype(has_next_c)::value;
    const int cur = t & 1;
    if constexpr (!(VDN_ATTN_ABL & 1)) {
      stage_v(cur, t);
      if constexpr (HAS_NEXT) {
        if (t + 2 < nt) stage_k(cur, t + 2);
      }
    }
    if constexpr (!HAS_NEXT) {  // only the last tile can be ragged
      if ((t + 1) * 64 > nk) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t * 64 + kb * 32 + acc_key(i, h) >= nk) s[kb][i] = -INFINITY;
      }
    }
    const char* sKn = smem + (cur ^ 1) * NT * TILE;         // K_{t+1}
    const char* sVp = smem + (cur ^ 1) * NT * TILE + TILE;  // V_{t-1}

    // ---- hand-placed stream: 48 slots = 16 MFMA triples (8 of PV(t-1), c outer; then 8 of QK(t+1)), each
    // slot = one small piece of softmax(t) + one MFMA, fenced by sched_barrier so the order survives the
    // scheduler (its own choice is all MFMAs first, then all VALU: no overlap inside the in-order wave;
    // sched_group_barrier pipelines were not honoured on this block). A 32x32x16 MFMA keeps the matrix pipe
    // 32 cycles and the issue port 8, so <= 24 cycles of VALU issue per slot are hidden.
    // S and P are updated IN PLACE (the loop is register-bound: O 32 + S 32 + Q 32 + P 32 + fragments):
    //   P fragment f is read by PV(t-1) in slots 6f..6f+5 and rewritten by softmax(t) in slots >= 7+8f;
    //   S[0] is last read in slot 20 and overwritten by QK(t+1) from slot 24, S[1] read in 36 (softmax piece
    //   first in the slot), overwritten from slot 36's MFMA on.
    V8 fa[2], fl[2];
    i32x8 f8[2];  // QK8: the 8-bit K fragment of triples (kb, ks = 0) [e5m2(K)] and (kb, ks = 1) [remainder plane]
    auto frag = [&](auto jc, V8& a, V8& al) {
      constexpr int j = decltype(jc)::value;
      if constexpr (VDN_ATTN_ABL & 16) {
        asm volatile("" : "+v"(a), "+v"(al));  // timing ablation: no fragment reads (operands stay what they were)
      } else if constexpr (j < 8) {
        if constexpr (HAS_PREV) {
          a = *(const V8*)(sVp + v_addr(j & 1, j >> 1));
          if constexpr (SPLIT) al = *(const V8*)(sVp + 2 * TILE + v_addr(j & 1, j >> 1));
        }
      } else if constexpr (HAS_NEXT) {
        a = *(const V8*)(sKn + k_addr((j - 8) >> 2, (j - 8) & 3));
        if constexpr (QK8) {
          constexpr int ks = (j - 8) & 3;
          if constexpr (ks < 2) f8[ks] = k8_read(sKn, (j - 8) >> 2, ks);
        } else if constexpr (SPLIT) {
          al = *(const V8*)(sKn + 2 * TILE + k_addr((j - 8) >> 2, (j - 8) & 3));
        }
      }
    };
    auto mma = [&](auto jc, auto qc, const V8& a, const V8& al) {
      constexpr int j = decltype(jc)::value, q = decltype(qc)::value;
      if constexpr (!SPLIT && q > 0) return;
      if constexpr (j < 8) {
        if constexpr (HAS_PREV) {
          constexpr int c = j >> 1, db = j & 1;
          if constexpr (q == 0) o[db] = HT::mfma32(a, pf[c >> 1][c & 1], o[db]);
          if constexpr (q == 1 && !PV2) o[db] = HT::mfma32(a, pl[c >> 1][c & 1], o[db]);  // PV2: a VALU-only slot
          if constexpr (q == 2) o[db] = HT::mfma32(al, pf[c >> 1][c & 1], o[db]);
        }
      } else if constexpr (HAS_NEXT) {
        constexpr int kb = (j - 8) >> 2, ks = (j - 8) & 3;
        if constexpr (ks == 0 && q == 0) {
          f32x16 z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          s[kb] = HT::mfma32(a, qf[ks], z);
        } else if constexpr (QK8) {
          // 6 MFMAs per 32-key block in its 12 slots: 4 fp16 (q == 0) + the two 8-bit cross terms in the q == 1 slots
          // of ks = 0, 1 (twice as long as a fp16 MFMA: they hide two softmax pieces); the other slots are VALU-only
          if constexpr (q == 0) s[kb] = HT::mfma32(a, qf[ks], s[kb]);
          if constexpr (q == 1 && ks == 0) s[kb] = cross_hl(f8[0], s[kb]);
          if constexpr (q == 1 && ks == 1) s[kb] = cross_lh(f8[1], s[kb]);
        } else {
          if constexpr (q == 0) s[kb] = HT::mfma32(a, qf[ks], s[kb]);
          if constexpr (q == 1) s[kb] = HT::mfma32(a, ql[ks], s[kb]);
          if constexpr (q == 2) s[kb] = HT::mfma32(al, qf[ks], s[kb]);
        }
      }
    };
    float mx = -1e30f, alpha = 1.f, mb = 0.f, ls = 0.f, px0 = 0.f, px1 = 0.f;
    bool bump = false;
    auto vstep = [&](auto kc) {
      constexpr int k = decltype(kc)::value;
      if constexpr (k < 4) {  // tile maximum, 8 scores per slot
        constexpr int kb = k >> 1, i = (k & 1) * 8;
        mx = fmaxf(fmaxf(mx, s[kb][i]), s[kb][i + 1]);
        mx = fmaxf(fmaxf(mx, s[kb][i + 2]), s[kb][i + 3]);
        mx = fmaxf(fmaxf(mx, s[kb][i + 4]), s[kb][i + 5]);
        mx = fmaxf(fmaxf(mx, s[kb][i + 6]), s[kb][i + 7]);
      } else if constexpr (k == 4) {
        mx = fmaxf(mx, xhalf(mx));
      } else if constexpr (k == 5) {
        bump = __builtin_amdgcn_ballot_w64((mx - m_run) * scale_log2 > LAZY) != 0;  // wave-uniform
        const float m_new = bump ? fmaxf(m_run, mx) : m_run;
        alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);  // == 1 when not bumped
        m_run = m_new;
        mb = -m_new * scale_log2;
      } else if constexpr (k < 38) {
        constexpr int pi = (k - 6) >> 1, kb = pi >> 3, i = (pi & 7) * 2;
        if constexpr (((k - 6) & 1) == 0) {
          px0 = __builtin_amdgcn_exp2f(fmaf(s[kb][i], scale_log2, mb));
          px1 = __builtin_amdgcn_exp2f(fmaf(s[kb][i + 1], scale_log2, mb));
        } else if constexpr (SPLIT && PV2) {
          if constexpr (DT == VDN_F16) {  // one packed convert, one dot2 for the row sum of the ROUNDED weights
            const f16x2 pp = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(px0, px1));
            pf[kb][i >> 3][i & 7] = pp[0]; pf[kb][i >> 3][(i & 7) + 1] = pp[1];
            ls = __builtin_amdgcn_fdot2(pp, f16x2{(_Float16)1.f, (_Float16)1.f}, ls, false);
            asm volatile("" : "+v"(ls));  // or LLVM sinks all 16 dot2 behind the last MFMA of the iteration
          } else {
            const T a0 = (T)px0, a1 = (T)px1;
            pf[kb][i >> 3][i & 7] = a0; pf[kb][i >> 3][(i & 7) + 1] = a1;
            ls += (float)a0 + (float)a1;
          }
        } else {
          ls += px0 + px1;
          if constexpr (SPLIT) {
            T a0, a1, b0, b1;
            split2_rtz(px0, px1, a0, a1, b0, b1);
            pf[kb][i >> 3][i & 7] = a0; pf[kb][i >> 3][(i & 7) + 1] = a1;
            pl[kb][i >> 3][i & 7] = b0; pl[kb][i >> 3][(i & 7) + 1] = b1;
          } else {
            pf[kb][i >> 3][i & 7] = (T)px0; pf[kb][i >> 3][(i & 7) + 1] = (T)px1;
          }
        }
      } else if constexpr (k == 38) {
        l_run = l_run * alpha + ls;
      }
    };
    auto slot = [&](auto jc) {
      constexpr int j = decltype(jc)::value, tj = j / 3, q = j % 3;
      if constexpr (q == 0 && tj + 1 < 16) frag(std::integral_constant<int, tj + 1>{}, fa[(tj + 1) & 1], fl[(tj + 1) & 1]);
      if constexpr (!(VDN_ATTN_ABL & 4)) vstep(jc);
      if constexpr (!(VDN_ATTN_ABL & 8)) mma(std::integral_constant<int, tj>{}, std::integral_constant<int, q>{}, fa[tj & 1], fl[tj & 1]);
      __builtin_amdgcn_sched_barrier(0);
    };
    frag(std::integral_constant<int, 0>{}, fa[0], fl[0]);
    for_each_slot<48>(slot);

    // P(t) is only consumed by the next iteration's MFMAs: keep LLVM from sinking its computation there.
    asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]));
    if constexpr (SPLIT && !PV2) asm volatile("" : "+v"(pl[0][0]), "+v"(pl[0][1]), "+v"(pl[1][0]), "+v"(pl[1][1]));
    if (bump) {  // O is in the old reference (it just received tile t-1): move it to the new one
      asm volatile("" ::: "memory");  // keep this a real (rarely taken) branch: if-converted it costs 16 v_pk_mul per tile
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    }
    if constexpr (VDN_ATTN_ABL & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else stage_barrier();
  };

  stage_k(0, 0);
  if (nt > 1) stage_k(1, 1);
  stage_barrier();
  qk(0, s);
  stage_barrier();  // every wave has read K_0 before iteration 0 lets K_2 overwrite it
  constexpr std::true_type Y{};
  constexpr std::false_type N{};
  if (nt == 1) {
    iter(0, N, N);
  } else {
    iter(0, N, Y);
    for (int t = 1; t + 1 < nt; ++t) iter(t, Y, Y);
    iter(nt - 1, Y, N);
  }
  pv((nt - 1) & 1, pf, pl);

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
          if (outl) {  // the output is written as planes whenever the caller passes a lo plane (also in 1-product mode)
            T a, b2;
            split_rtz(x, a, b2);
            v[e] = a;
            vl[e] = b2;
          } else {
            v[e] = (T)x;
          }
        }
        *(typename HT::V4*)(out + oo + db * 32 + 8 * g + 4 * h) = v;
        if (outl) *(typename HT::V4*)(outl + oo + db * 32 + 8 * g + 4 * h) = vl;
      }
  }
}

#include "attn2_kernel.hpp"

template <int DT, bool SPLIT, int NB /*32-frame blocks: sequences of up to 32 NB frames*/>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const typename Half<DT>::T* __restrict__ qkv,
                                                            typename Half<DT>::T* __restrict__ out,
                                                            const typename Half<DT>::T* __restrict__ qkv_lo,
                                                            typename Half<DT>::T* __restrict__ out_lo, int nseq,
                                                            int Tn, int D, int c, int heads, float scale_log2,
                                                            const float* __restrict__ rope_cs) {
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
  const T* vb = base + 2 * c;
  const T* vb_lo = SPLIT ? base_lo + 2 * c : nullptr;
  const size_t ooff = ((size_t)b * Tn * D + d) * c + head * dh;
  T* ob = out + ooff;
  T* ob_lo = SPLIT ? out_lo + ooff : nullptr;
  const size_t os = (size_t)D * c;
  const int nks = (dh + 15) >> 4;
  const int neb = (dh + 31) >> 5;
  const int pk = perm23(r);

  for (int qb = 0; qb < NB; ++qb) {  // 32 query frames at a time (the v5 refiner runs 64-frame clips)
    if (qb * 32 >= Tn) break;
    const int tq = qb * 32 + r;
    const int fq = tq < Tn ? tq : Tn - 1;
    // ---- S^T = K Q^T, one 32 x 32 tile per key block
    f32x16 s[NB];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
      if (kb * 32 >= Tn) continue;
      const int tk = kb * 32 + pk;
      const int fk = tk < Tn ? tk : Tn - 1;
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
          if (rope_cs) {  // pe = 'rope' (motion_module.py:279-282): pairs (2i, 2i+1) of the c channels rotate with the frame index
            const size_t pair0 = (size_t)(head * dh + e0) >> 1;
            auto rot = [&](V8& hi, V8& lo, int frame) {
              const float* cs = rope_cs + ((size_t)frame * (c >> 1) + pair0) * 2;
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                const float x0 = (float)hi[j] + (SPLIT ? (float)lo[j] : 0.f), x1 = (float)hi[j + 1] + (SPLIT ? (float)lo[j + 1] : 0.f);
                const float cc = cs[j], ss = cs[j + 1];
                const float y0 = x0 * cc - x1 * ss, y1 = x0 * ss + x1 * cc;
                if constexpr (SPLIT) {
                  T h0, h1, l0, l1;
                  split2_rtz(y0, y1, h0, h1, l0, l1);
                  hi[j] = h0; hi[j + 1] = h1; lo[j] = l0; lo[j + 1] = l1;
                } else {
                  hi[j] = (T)y0; hi[j + 1] = (T)y1;
                }
              }
            };
            rot(bq, bl, fq);
            rot(a, al, fk);
          }
        }
        s[kb] = HT::mfma32(a, bq, s[kb]);
        if constexpr (SPLIT) {
          s[kb] = HT::mfma32(a, bl, s[kb]);
          s[kb] = HT::mfma32(al, bq, s[kb]);
        }
      }
    }
    float mx = -1e30f;
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = s[kb][i] * scale_log2;
        v = (kb * 32 + acc_key(i, h) < Tn) ? v : -INFINITY;
        s[kb][i] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float ls = 0.f;
    V8 pf[NB][2], pl[NB][2];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[kb][i] - mx);
        ls += pv;
        if constexpr (SPLIT) {
          T a2, b2;
          split_rtz(pv, a2, b2);
          pf[kb][i >> 3][i & 7] = a2;
          pl[kb][i >> 3][i & 7] = b2;
        } else {
          pf[kb][i >> 3][i & 7] = (T)pv;
        }
      }
    ls += __shfl_xor(ls, 32);
    const float inv = 1.0f / ls;

    // ---- O^T = V^T P^T, 32 output dims per pass; V^T fragments gathered element-wise (tiny op)
    for (int eb = 0; eb < neb; ++eb) {
      f32x16 o;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = 0.f;
      const int e = eb * 32 + r;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (kb * 32 >= Tn) continue;
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          V8 a, al;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int key = kb * 32 + 16 * sp + 8 * h + j;
            const bool ok = key < Tn && e < dh;
            a[j] = ok ? vb[key * rs + e] : (T)0.f;
            if constexpr (SPLIT) al[j] = ok ? vb_lo[key * rs + e] : (T)0.f;
          }
          o = HT::mfma32(a, pf[kb][sp], o);
          if constexpr (SPLIT) {
            o = HT::mfma32(a, pl[kb][sp], o);
            o = HT::mfma32(al, pf[kb][sp], o);
          }
        }
      }
      if (tq < Tn) {
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
            *(typename HT::V4*)(ob + tq * os + e0) = v;
            if constexpr (SPLIT) *(typename HT::V4*)(ob_lo + tq * os + e0) = vl;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// temporal_last_kernel — streaming mode (video_depth_stream.py:133-158, motion_module.py:255-277) with a
// PROJECTED key/value cache: every cached frame keeps its q|k|v projection WITHOUT the
// frame-position term (f32 [HW, 3c], the position is not known when the frame is cached: the window slides);
// by linearity  W (x + pe[t]) = W x + W pe[t],  so the position enters as three small tables
// (pe W_q^T, pe W_k^T, pe W_v^T: f32 [T, c] each) added on load. Only the NEWEST frame queries (the reference
// computes all T and keeps the last). One wave per pixel: lane owns c/64 consecutive channels, a head
// (c/8 channels) is 8 consecutive lanes; scores are reduced inside the 8-lane group, softmax over the
// <= 32 frames in registers, output written as operand planes for the out-projection GEMM. HBM-bound:
// T x 2c f32 per pixel are read once.
struct SlotTable { int s[32]; };  // ring slots of the window's frames, oldest first; passed BY VALUE (no table in HBM)

template <int DT, int CPL /*channels per lane: c / 64*/>
__global__ __launch_bounds__(256) void temporal_last_kernel(const float* __restrict__ pool, size_t slot_stride, SlotTable tab,
                                                            int T, int HW, int c,
                                                            const float* __restrict__ peq, const float* __restrict__ pek,
                                                            const float* __restrict__ pev, float scale,
                                                            typename Half<DT>::T* __restrict__ out,
                                                            typename Half<DT>::T* __restrict__ out_lo) {
  using Th = typename Half<DT>::T;
  const int lane = threadIdx.x & 63;
  const int px = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (px >= HW) return;
  const int ch = lane * CPL;
  const size_t row = (size_t)px * 3 * c;
  float q[CPL];
  {
    const float* qn = pool + (size_t)tab.s[T - 1] * slot_stride + row + ch;  // the newest frame is the query, at position T-1
#pragma unroll
    for (int e = 0; e < CPL; ++e) q[e] = (qn[e] + peq[(size_t)(T - 1) * c + ch + e]) * scale;
  }
  float sc[32];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    sc[t] = -INFINITY;
    if (t < T) {
      const float* kt = pool + (size_t)tab.s[t] * slot_stride + row + c + ch;
      const float* pk = pek + (size_t)t * c + ch;
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < CPL; ++e) d = fmaf(q[e], kt[e] + pk[e], d);
      d += __shfl_xor(d, 1);
      d += __shfl_xor(d, 2);
      d += __shfl_xor(d, 4);
      sc[t] = d;
      mx = fmaxf(mx, d);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    sc[t] = t < T ? __expf(sc[t] - mx) : 0.f;
    den += sc[t];
  }
  const float inv = 1.0f / den;
  float o[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) o[e] = 0.f;
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    if (t < T) {
      const float* vt = pool + (size_t)tab.s[t] * slot_stride + row + 2 * c + ch;
      const float* pv = pev + (size_t)t * c + ch;
      const float p = sc[t] * inv;
#pragma unroll
      for (int e = 0; e < CPL; ++e) o[e] = fmaf(p, vt[e] + pv[e], o[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < CPL; ++e) store_half(out, out_lo, (size_t)px * c + ch + e, o[e]);
}

// pv: MFMA products per P V term in split mode (include/vdn.h): 1 (default), 2 or 3 — a per-call argument: the library keeps
// no selection state.
template <int DT>
int flash_launch(const void* Q, const void* K, const void* Vt, void* out, const void* Ql, const void* Kl, const void* Vtl,
                 void* outl, const void* Q8, const void* K8, void* out8, int out_kt, int B, int H, int nq, int nq_pad, int nk, int nk_pad,
                 float sl2, int pv, hipStream_t s) {
  using T = typename Half<DT>::T;
  const dim3 grid(((nq + 127) / 128) * B * H);
  const bool pv3 = pv == 3;  // the 3-product P V with P split into hi / lo planes
  const uint8_t* q8 = (const uint8_t*)Q8;
  const uint8_t* k8 = (const uint8_t*)K8;
  // the 8-bit / K-tile-major output planes and a lo-less output exist in the default kernel only (fp16 split planes + Q8 / K8)
  if ((out8 || out_kt || (Ql && !outl)) && !(Ql && !pv3 && q8 && k8 && DT == VDN_F16)) return VDN_EUNSUPPORTED;
  if (Ql && pv3)
    hipLaunchKernelGGL((flash_attn_kernel<DT, true, false, false>), grid, dim3(256), 65536, s, (const T*)Q, (const T*)K, (const T*)Vt,
                       (T*)out, (const T*)Ql, (const T*)Kl, (const T*)Vtl, (T*)outl, q8, k8, H, nq, nq_pad, nk, nk_pad, sl2);
  else if (Ql && q8 && k8 && DT == VDN_F16) {
    if constexpr (DT == VDN_F16) {
#ifndef VDN_ATTN2_NW
#define VDN_ATTN2_NW 4
#endif
      constexpr int NW = VDN_ATTN2_NW;
      const dim3 grid2(((nq + 32 * NW - 1) / (32 * NW)) * B * H);
      if (pv == 1)
        hipLaunchKernelGGL((flash_attn2_kernel<1, NW>), grid2, dim3(64 * NW), (VDN_ATTN_ABL & 32) ? 98304 : 65536, s, (const T*)Q, (const T*)K, (const T*)Vt, (T*)out,
                           (const T*)Kl, (const T*)Vtl, (T*)outl, q8, k8, H, nq, nq_pad, nk, nk_pad, sl2, (uint8_t*)out8, out_kt, B * nq);
      else
        hipLaunchKernelGGL((flash_attn2_kernel<2, NW>), grid2, dim3(64 * NW), (VDN_ATTN_ABL & 32) ? 98304 : 65536, s, (const T*)Q, (const T*)K, (const T*)Vt, (T*)out,
                           (const T*)Kl, (const T*)Vtl, (T*)outl, q8, k8, H, nq, nq_pad, nk, nk_pad, sl2, (uint8_t*)out8, out_kt, B * nq);
    }
  } else if (Ql)
    hipLaunchKernelGGL((flash_attn_kernel<DT, true, true, false>), grid, dim3(256), 65536, s, (const T*)Q, (const T*)K, (const T*)Vt,
                       (T*)out, (const T*)Ql, (const T*)Kl, (const T*)Vtl, (T*)outl, q8, k8, H, nq, nq_pad, nk, nk_pad, sl2);
  else
    hipLaunchKernelGGL((flash_attn_kernel<DT, false, false, false>), grid, dim3(256), 32768, s, (const T*)Q, (const T*)K, (const T*)Vt,
                       (T*)out, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr, (T*)outl, q8, k8, H, nq, nq_pad, nk,
                       nk_pad, sl2);
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

template <int DT>
int temporal_launch(const void* qkv, void* out, const void* qkv_lo, void* out_lo, int nseq, int T, int D, int c, int heads,
                    float sl2, const float* rope_cs, hipStream_t s) {
  using TT = typename Half<DT>::T;
  const dim3 grid((nseq + 3) / 4);
#define VDN_TA(SP, NB_)                                                                                          \
  hipLaunchKernelGGL((temporal_attn_kernel<DT, SP, NB_>), grid, dim3(256), 0, s, (const TT*)qkv, (TT*)out,       \
                     (const TT*)(SP ? qkv_lo : nullptr), (TT*)(SP ? out_lo : nullptr), nseq, T, D, c, heads, sl2, rope_cs)
  if (qkv_lo) {
    if (T <= 32) VDN_TA(true, 1); else VDN_TA(true, 2);
  } else {
    if (T <= 32) VDN_TA(false, 1); else VDN_TA(false, 2);
  }
#undef VDN_TA
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

}  // namespace

extern "C" int vdn_flash_attn(int dt, const void* Q, const void* K, const void* Vt, void* out, const void* Q_lo,
                              const void* K_lo, const void* Vt_lo, void* out_lo, const void* Q8, const void* K8, void* out8,
                              int out_kt, int B, int H, int nq, int nq_pad, int nk, int nk_pad, float scale, int pv_products,
                              vdn_stream stream) {
  if (!Q || !K || !Vt || !out || B <= 0 || H <= 0 || nq <= 0 || nk <= 0) return VDN_EINVAL;
  if (pv_products < 0 || pv_products > 3) return VDN_EINVAL;
  const int pv = pv_products ? pv_products : 1;
  if (nq_pad < nq || nk_pad < nk || (nk_pad & 63)) return VDN_EALIGN;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)out) & 15) return VDN_EALIGN;
  const int nlo = (Q_lo != nullptr) + (K_lo != nullptr) + (Vt_lo != nullptr);
  if (nlo != 0 && nlo != 3) return VDN_EINVAL;  // operand planes are all-or-nothing; the output may be split either way
  if (nlo == 3 && !out_lo && !out8) return VDN_EINVAL;  // split operands: the output carries its remainder as a 16-bit or an 8-bit plane
  if (((uintptr_t)Q_lo | (uintptr_t)K_lo | (uintptr_t)Vt_lo | (uintptr_t)out_lo | (uintptr_t)Q8 | (uintptr_t)K8 | (uintptr_t)out8) & 15) return VDN_EALIGN;
  if ((Q8 == nullptr) != (K8 == nullptr) || (Q8 && (nlo != 3 || dt != VDN_F16))) return VDN_EINVAL;  // 8-bit planes come in pairs, split fp16 only
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16)
    return flash_launch<VDN_F16>(Q, K, Vt, out, Q_lo, K_lo, Vt_lo, out_lo, Q8, K8, out8, out_kt, B, H, nq, nq_pad, nk, nk_pad, sl2, pv, s);
  if (dt == VDN_BF16)
    return flash_launch<VDN_BF16>(Q, K, Vt, out, Q_lo, K_lo, Vt_lo, out_lo, nullptr, nullptr, out8, out_kt, B, H, nq, nq_pad, nk, nk_pad, sl2, pv, s);
  return VDN_EUNSUPPORTED;
}

extern "C" int vdn_temporal_attn(int dt, const void* qkv, void* out, const void* qkv_lo, void* out_lo, int Bv, int T,
                                 int D, int c, int heads, float scale, const float* rope_cs, vdn_stream stream) {
  if (!qkv || !out || Bv <= 0 || T <= 0 || T > 64 || D <= 0 || heads <= 0 || c % heads) return VDN_EINVAL;
  if ((qkv_lo == nullptr) != (out_lo == nullptr)) return VDN_EINVAL;
  const int dh = c / heads;
  if ((dh & 7) || dh > 256 || (c & 7)) return VDN_EALIGN;
  if (((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)qkv_lo | (uintptr_t)out_lo) & 15) return VDN_EALIGN;
  const int nseq = Bv * D * heads;
  const float sl2 = scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  if (dt == VDN_F16) return temporal_launch<VDN_F16>(qkv, out, qkv_lo, out_lo, nseq, T, D, c, heads, sl2, rope_cs, s);
  if (dt == VDN_BF16) return temporal_launch<VDN_BF16>(qkv, out, qkv_lo, out_lo, nseq, T, D, c, heads, sl2, rope_cs, s);
  return VDN_EUNSUPPORTED;
}

extern "C" int vdn_temporal_attn_last(int dt, const float* pool, size_t slot_stride, const int32_t* slots, int T, int HW,
                                      int c, const float* pe_q, const float* pe_k, const float* pe_v, float scale,
                                      void* out, void* out_lo, vdn_stream stream) {
  if (!pool || !slots || !pe_q || !pe_k || !pe_v || !out || T < 1 || T > 32 || HW <= 0) return VDN_EINVAL;
  if (slot_stride < (size_t)HW * 3 * c) return VDN_EINVAL;
  SlotTable tab;
  for (int t = 0; t < 32; ++t) {
    tab.s[t] = t < T ? slots[t] : 0;
    if (tab.s[t] < 0) return VDN_EINVAL;
  }
  if (c <= 0 || (c & 63) || c > 1024) return VDN_EUNSUPPORTED;  // 8 heads of c/8 = 8 lanes x c/64 channels
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((HW + 3) / 4), block(256);
#define VDN_TL(DT_, CPL_)                                                                                                  \
  hipLaunchKernelGGL((temporal_last_kernel<DT_, CPL_>), grid, block, 0, s, pool, slot_stride, tab, T, HW, c, pe_q, pe_k, \
                     pe_v, scale, (typename Half<DT_>::T*)out, (typename Half<DT_>::T*)out_lo)
#define VDN_TL_C(DT_)                                  \
  switch (c) {                                         \
    case 64: VDN_TL(DT_, 1); break;                    \
    case 128: VDN_TL(DT_, 2); break;                   \
    case 192: VDN_TL(DT_, 3); break;                   \
    case 256: VDN_TL(DT_, 4); break;                   \
    case 384: VDN_TL(DT_, 6); break;                   \
    case 512: VDN_TL(DT_, 8); break;                   \
    case 768: VDN_TL(DT_, 12); break;                  \
    case 1024: VDN_TL(DT_, 16); break;                 \
    default: return VDN_EUNSUPPORTED;                  \
  }
  if (dt == VDN_F16) { VDN_TL_C(VDN_F16) } else if (dt == VDN_BF16) { VDN_TL_C(VDN_BF16) } else return VDN_EUNSUPPORTED;
#undef VDN_TL_C
#undef VDN_TL
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}
