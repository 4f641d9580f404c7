// flash_attn2_kernel — the attention of the default precision mode (fp16 split planes, 2-product P V, score cross terms
// on the block-scaled 8-bit MFMA), head_dim 64. Same data layout, same arithmetic per element and the same MFMA
// products as flash_attn_kernel<F16, SPLIT, PV2, QK8> (attn.hip, whose header describes the S^T = K Q^T / O^T = V^T P^T
// formulation, the row permutation and the LDS image); what differs is the instruction stream of a tile iteration:
//
//   * S and P are double-buffered by tile parity (iterations are instantiated for even and odd tiles), so QK(t+1),
//     softmax(t) and PV(t-1) share no registers and may interleave freely;
//   * the stream is GENERATED (tools/gen_attn_stream.py -> attn2_stream.inc): 28 MFMA gaps, each with the VALU pieces
//     its matrix-pipe time can shadow; the softmax of a score pair is the 3-stage pipeline F (2 fma) -> X (2 exp2) ->
//     C (convert + row sum) with the stages in different gaps, so no VALU instruction waits for its neighbour's result
//     (measured on the 48-slot stream of flash_attn_kernel: VALU alone 8.7 cycles per instruction, MFMA and VALU time
//     simply added up; profiles/r02_attn_ablate*.log);
//   * operand fragments are read LOOKAHEAD MFMAs ahead into a register ring, the 8 LDS-DMA pieces of the iteration are
//     spread over the gaps, LDS addresses are immediates (the buffer parity is a template parameter).
//
// Included by attn.hip (helpers perm23 / acc_key / xhalf / stage_barrier and the Half<> traits come from there).

// PV: MFMA products per P V term: 2 = P~ (V_hi + V_lo), 1 = P~ V_hi with V_hi rounded to nearest by the producer
// (vdn_gemm rounds the hi plane of transposed head splits that way): 8 MFMAs, 8 fragment reads and 2 LDS-DMA pieces fewer per tile.
// NW: waves per workgroup (32 queries each). 4 (shipped): two workgroups per CU, every wave issues 2 LDS-DMA pieces per tile
// and plane. 8 (A/B build -DVDN_ATTN2_NW=8): one workgroup per CU, one piece per wave — the same 2 waves per SIMD and half the
// DMA instructions per wave (an LDS-DMA piece costs its wave 60-180 cycles of issue: s_memtime stamps put ~500 of a tile's
// 3000 cycles there), but one barrier over 8 waves instead of two independent 4-wave groups: measured 134 vs 125 us. Stays 4.
template <int PV, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void flash_attn2_kernel(
    const _Float16* __restrict__ Q, const _Float16* __restrict__ K, const _Float16* __restrict__ Vt, _Float16* __restrict__ out,
    const _Float16* __restrict__ Kl, const _Float16* __restrict__ Vtl, _Float16* __restrict__ outl, const uint8_t* __restrict__ Q8,
    const uint8_t* __restrict__ K8, int H, int nq, int nq_pad, int nk, int nk_pad, float scale_log2, uint8_t* __restrict__ out8,
    int out_kt, int out_rows) {
  using HT = Half<VDN_F16>;
  using T = _Float16;
  using V8 = typename HT::V8;
  constexpr int TILE = 8192;  // 64 rows x 128 B
  constexpr int NT = 4;       // tiles per stage: K, V^T, K8, V^T_lo
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int QB = 32 * NW;      // queries per workgroup
  constexpr int NPC = 8 / NW;      // LDS-DMA pieces (of the 8 of a 64-row tile plane) per wave
  const int nqb = (nq + QB - 1) / QB;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);  // the q-blocks of one (batch, head) share an XCD's L2
  const int bh = logical / nqb;
  const int q0 = (logical - bh * nqb) * QB + wave * 32;
  const int r = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operands): fp16 Q[q][16 ks + 8 h + j] and the two 8-bit planes Q8[q][32 h + j]
  V8 qf[4];
  i32x8 q8h, q8l;
  {
    int q = q0 + r;
    q = q < nq ? q : nq - 1;
    const size_t qo = ((size_t)bh * nq_pad + q) * 64 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const V8*)(Q + qo + 16 * ks);
    const uint8_t* q8 = Q8 + ((size_t)bh * nq_pad + q) * 128 + 32 * h;
    const u32x4 a0 = *(const u32x4*)q8, a1 = *(const u32x4*)(q8 + 16), b0 = *(const u32x4*)(q8 + 64), b1 = *(const u32x4*)(q8 + 80);
#pragma unroll
    for (int e = 0; e < 4; ++e) { q8h[e] = (int)a0[e]; q8h[4 + e] = (int)a1[e]; q8l[e] = (int)b0[e]; q8l[4 + e] = (int)b1[e]; }
  }

  // ---- staging (buffer_load_dwordx4 ... offen lds): plane of this (batch, head) as a buffer resource, loop-invariant
  // 32-bit lane offsets, the tile as a scalar offset; 8 pieces of 1 KiB per tile and plane, 2 per wave
  const int lr = lane >> 3;
  const unsigned k_bytes = (unsigned)nk_pad * 64 * sizeof(T);
  const auto rsrc = [&](const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (size_t)bh * k_bytes), 0, (int)k_bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rK = rsrc(K), rV = rsrc(Vt), rK8 = rsrc(K8), rVl = rsrc(Vtl);
  int koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int row = (wave + 4 * i) * 8 + lr;
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    koff[i] = (row * 64 + c * 8) * (int)sizeof(T);
    voff[i] = (row * nk_pad + c * 8) * (int)sizeof(T);
  }
#define A2_BLDS16(rs, voffset, soffset, dst) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst), 16, voffset, soffset, 0, 0)
  auto stage_k_piece = [&](int buf, int t, int i, bool second) {
    A2_BLDS16(second ? rK8 : rK, koff[i], t * (64 * 64 * (int)sizeof(T)), smem + buf * NT * TILE + (second ? 2 * TILE : 0) + (wave + 4 * i) * 1024);
  };
  auto stage_v_piece = [&](int buf, int t, int i, bool second) {
    A2_BLDS16(second ? rVl : rV, voff[i], t * (64 * (int)sizeof(T)), smem + buf * NT * TILE + TILE + (second ? 2 * TILE : 0) + (wave + 4 * i) * 1024);
  };

  // ---- fragment addresses: row r (+32 kb) of a tile, 16-byte chunk (2 ks + h) ^ swz(row)
  const int k_base = perm23(r) * 128 + ((h ^ ((perm23(r) >> 1) & 7)) << 4);
  const int v_base = r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
  const int k8_base = perm23(r) * 128 + (((2 * h) ^ ((perm23(r) >> 1) & 7)) << 4);
  auto k_addr = [&](int kb, int ks) { return kb * 4096 + (k_base ^ (ks << 5)); };
  auto v_addr = [&](int db, int c) { return db * 4096 + (v_base ^ (c << 5)); };
  auto k8_read = [&](const char* sK, int kb, int lo) {
    const char* p = sK + 2 * TILE + kb * 4096;
    const u32x4 a0 = *(const u32x4*)(p + (k8_base ^ ((4 * lo) << 4))), a1 = *(const u32x4*)(p + (k8_base ^ ((4 * lo + 1) << 4)));
    i32x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (int)a0[e]; v[4 + e] = (int)a1[e]; }
    return v;
  };
  auto cross_hl = [&](const i32x8& k8, f32x16 c) {  // K8 (Q_lo8)^T 2^-10
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8, q8l, c, 1, 1, 0, 127, 0, VDN_LO8_E8M0);
  };
  auto cross_lh = [&](const i32x8& k8l, f32x16 c) {  // K_lo8 (Q8)^T 2^-10
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k8l, q8h, c, 1, 1, 0, VDN_LO8_E8M0, 0, 127);
  };

  f32x16 o[2], s[2][2];  // O^T accumulators; S^T of the current / next tile by tile parity
  V8 pf[2][2][2];        // P~ of the previous / current tile by tile parity: [parity][key block][fragment]
#pragma unroll
  for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
  float m_run = -1e30f, l_run = 0.f;
  constexpr float LAZY = 6.0f;  // the reference maximum moves only when a tile exceeds it by 2^LAZY (flash_attn_kernel)
  const int nt = (nk + 63) >> 6;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#if VDN_ATTN_ABL & 64
  unsigned long long stamp_sum[5] = {0, 0, 0, 0, 0}, stamp_last = 0;
#endif
#define A2_IC(n) std::integral_constant<int, n> {}
  // pin a value to this point of the stream (LLVM otherwise sinks work whose result is only used next iteration)
#define A2_PIN(x) asm volatile("" : "+v"(x))
  // one tile iteration t of parity PAR: S(t+1) = K_{t+1} Q^T -> s[PAR ^ 1] || P(t) = exp2(S(t) - m) -> pf[PAR] || O += V_{t-1} P(t-1)
  auto iter = [&](int t, auto has_prev_c, auto has_next_c, auto par_c) {
    constexpr bool HAS_PREV = decltype(has_prev_c)::value, HAS_NEXT = decltype(has_next_c)::value;
    constexpr int PAR = decltype(par_c)::value;
    const char* sKn = smem + (PAR ^ 1) * NT * TILE;         // K_{t+1}, K8_{t+1}
    const char* sVp = smem + (PAR ^ 1) * NT * TILE + TILE;  // V_{t-1}, V_lo_{t-1}
    f32x16(&sc)[2] = s[PAR];
    f32x16(&sn)[2] = s[PAR ^ 1];
    if constexpr (!HAS_NEXT) {  // only the last tile can be ragged
      if ((t + 1) * 64 > nk) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t * 64 + kb * 32 + acc_key(i, h) >= nk) sc[kb][i] = -INFINITY;
      }
    }
    V8 fr[8];  // fragment ring (the generated stream uses lookahead + 1 of them)
    i32x8 f8[2];
    float mx = -1e30f, alpha = 1.f, mb = 0.f, ls0 = 0.f, ls1 = 0.f;
    float ff[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, fe[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    unsigned pw[16];  // the 16 packed pairs of P~(t)
    bool bump = false;
    const int tk = t + 2 < nt ? t + 2 : nt - 1;  // past the end the last K tile is fetched again, into the buffer nobody reads

    auto ldv = [&](auto cc, auto dbc, auto wc, auto bc) {
      if constexpr (HAS_PREV) fr[decltype(bc)::value] = *(const V8*)(sVp + decltype(wc)::value * 2 * TILE + v_addr(decltype(dbc)::value, decltype(cc)::value));
    };
    auto ldk = [&](auto kbc, auto ksc, auto bc) {
      if constexpr (HAS_NEXT) fr[decltype(bc)::value] = *(const V8*)(sKn + k_addr(decltype(kbc)::value, decltype(ksc)::value));
    };
    auto ldk8 = [&](auto kbc, auto loc, auto bc) {
      if constexpr (HAS_NEXT) f8[decltype(bc)::value] = k8_read(sKn, decltype(kbc)::value, decltype(loc)::value);
    };
    auto mpv = [&](auto cc, auto dbc, auto bc) {
      constexpr int c = decltype(cc)::value, db = decltype(dbc)::value;
      if constexpr (HAS_PREV) o[db] = HT::mfma32(fr[decltype(bc)::value], pf[PAR ^ 1][c >> 1][c & 1], o[db]);
    };
    auto mqk = [&](auto kbc, auto ksc, auto bc, auto firstc) {
      constexpr int kb = decltype(kbc)::value;
      if constexpr (HAS_NEXT) sn[kb] = HT::mfma32(fr[decltype(bc)::value], qf[decltype(ksc)::value], decltype(firstc)::value ? zero16 : sn[kb]);
    };
    auto mqx = [&](auto kbc, auto loc, auto bc, auto firstc) {
      constexpr int kb = decltype(kbc)::value;
      if constexpr (HAS_NEXT) {
        if constexpr (decltype(loc)::value == 0) sn[kb] = cross_hl(f8[decltype(bc)::value], decltype(firstc)::value ? zero16 : sn[kb]);
        else sn[kb] = cross_lh(f8[decltype(bc)::value], decltype(firstc)::value ? zero16 : sn[kb]);
      }
    };
    auto vmax = [&](auto qc) {
      constexpr int kb = decltype(qc)::value >> 1, i = (decltype(qc)::value & 1) * 8;
      mx = fmaxf(fmaxf(mx, sc[kb][i]), sc[kb][i + 1]);
      mx = fmaxf(fmaxf(mx, sc[kb][i + 2]), sc[kb][i + 3]);
      mx = fmaxf(fmaxf(mx, sc[kb][i + 4]), sc[kb][i + 5]);
      mx = fmaxf(fmaxf(mx, sc[kb][i + 6]), sc[kb][i + 7]);
    };
    auto vxh = [&]() { mx = fmaxf(mx, xhalf(mx)); };
    auto vbp = [&]() {
      bump = __builtin_amdgcn_ballot_w64((mx - m_run) * scale_log2 > LAZY) != 0;  // wave-uniform
      const float m_new = bump ? fmaxf(m_run, mx) : m_run;
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2);  // == 1 when not bumped
      m_run = m_new;
      mb = -m_new * scale_log2;
    };
    auto vf = [&](auto pc) {  // pair p: scores sc[p >> 3][2 (p & 7)], +1
      constexpr int p = decltype(pc)::value, kb = p >> 3, i = (p & 7) * 2;
      if constexpr (VDN_ATTN_ABL & 128) {  // timing ablation: what a pre-scaled Q + the maximum as accumulator init would save (no fma)
        ff[p & 1][0] = sc[kb][i];
        ff[p & 1][1] = sc[kb][i + 1];
      } else {
        ff[p & 1][0] = fmaf(sc[kb][i], scale_log2, mb);
        ff[p & 1][1] = fmaf(sc[kb][i + 1], scale_log2, mb);
      }
    };
    auto vx = [&](auto pc) {
      constexpr int p = decltype(pc)::value;
      fe[p & 1][0] = __builtin_amdgcn_exp2f(ff[p & 1][0]);
      fe[p & 1][1] = __builtin_amdgcn_exp2f(ff[p & 1][1]);
    };
    auto vc = [&](auto pc) {  // one packed convert, one dot2 for the row sum of the ROUNDED weights (two partial sums)
      constexpr int p = decltype(pc)::value, f = p >> 2;
      pw[p] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(fe[p & 1][0], fe[p & 1][1]));
      const f16x2 pp = __builtin_bit_cast(f16x2, pw[p]);
      if constexpr (p & 1) { ls1 = __builtin_amdgcn_fdot2(pp, f16x2{(_Float16)1.f, (_Float16)1.f}, ls1, false); A2_PIN(ls1); }
      else { ls0 = __builtin_amdgcn_fdot2(pp, f16x2{(_Float16)1.f, (_Float16)1.f}, ls0, false); A2_PIN(ls0); }
      if constexpr ((p & 3) == 3) {  // fragment f = keys 8f .. 8f+7 of the tile is complete: one 4-register value, pinned here
        pf[PAR][f >> 1][f & 1] = __builtin_bit_cast(V8, u32x4{pw[4 * f], pw[4 * f + 1], pw[4 * f + 2], pw[4 * f + 3]});
        A2_PIN(pf[PAR][f >> 1][f & 1]);
      }
    };
    auto vlr = [&]() { l_run = l_run * alpha + (ls0 + ls1); A2_PIN(l_run); };
    auto dma = [&](auto ic) {  // the iteration's 8 LDS-DMA pieces of this wave: V_t (4), K_{t+2} (4)
      constexpr int i = decltype(ic)::value;
      if constexpr (((i & 3) >> 1) < NPC) {   // piece index (i >> 1) & 1: a wave of an 8-wave workgroup owns one piece per plane
        if constexpr (i < 4) {
          if constexpr (PV == 2 || !(i & 1)) stage_v_piece(PAR, t, i >> 1, i & 1);
        } else if constexpr (HAS_NEXT) {
          stage_k_piece(PAR, tk, (i - 4) >> 1, i & 1);
        }
      }
    };
#if VDN_ATTN_ABL & 16  // timing ablations (attn.hip): no fragment reads / no MFMA / no softmax VALU / no LDS-DMA
#define A2_LDV(c, db, w, b) A2_PIN(fr[b]);
#define A2_LDK(kb, ks, b) A2_PIN(fr[b]);
#define A2_LDK8(kb, lo, b) A2_PIN(f8[b]);
#else
#define A2_LDV(c, db, w, b) ldv(A2_IC(c), A2_IC(db), A2_IC(w), A2_IC(b));
#define A2_LDK(kb, ks, b) ldk(A2_IC(kb), A2_IC(ks), A2_IC(b));
#define A2_LDK8(kb, lo, b) ldk8(A2_IC(kb), A2_IC(lo), A2_IC(b));
#endif
#if VDN_ATTN_ABL & 8
#define A2_PV(c, db, b)
#define A2_QK(kb, ks, b, f)
#define A2_QX(kb, lo, b, f)
#else
#define A2_PV(c, db, b) mpv(A2_IC(c), A2_IC(db), A2_IC(b));
#define A2_QK(kb, ks, b, f) mqk(A2_IC(kb), A2_IC(ks), A2_IC(b), A2_IC(f));
#define A2_QX(kb, lo, b, f) mqx(A2_IC(kb), A2_IC(lo), A2_IC(b), A2_IC(f));
#endif
#if VDN_ATTN_ABL & 4
#define A2_MAX(q)
#define A2_XH()
#define A2_BP()
#define A2_F(p)
#define A2_X(p)
#define A2_C(p)
#define A2_LR()
#else
#define A2_MAX(q) vmax(A2_IC(q));
#define A2_XH() vxh();
#define A2_BP() vbp();
#define A2_F(p) vf(A2_IC(p));
#define A2_X(p) vx(A2_IC(p));
#define A2_C(p) vc(A2_IC(p));
#define A2_LR() vlr();
#endif
#if VDN_ATTN_ABL & 1
#define A2_DMA(i)
#else
#define A2_DMA(i) dma(A2_IC(i));
#endif
#define A2_FENCE() __builtin_amdgcn_sched_barrier(0);
#if VDN_ATTN_ABL & 64  // timing build: cycles of the four quarters of the stream (and of the iteration's tail) summed per wave
#define A2_STAMP(i) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (i > 0) stamp_sum[i - 1] += now - stamp_last; stamp_last = now; }
#ifndef VDN_ATTN2_STAMPS_INC   // the stamped twin of the stream of the PV mode being timed (tools/gen_attn_stream.py --stamps [--pv 1])
#define VDN_ATTN2_STAMPS_INC "attn2_stream_stamps.inc"
#endif
#include VDN_ATTN2_STAMPS_INC
#undef A2_STAMP
#else
#ifndef VDN_ATTN2_INC
#define VDN_ATTN2_INC "attn2_stream.inc"
#endif
    if constexpr (PV == 1) {
#include "attn2_stream_pv1.inc"
    } else {
#include VDN_ATTN2_INC
    }
#endif
#undef A2_LDV
#undef A2_LDK
#undef A2_LDK8
#undef A2_PV
#undef A2_QK
#undef A2_QX
#undef A2_MAX
#undef A2_XH
#undef A2_BP
#undef A2_F
#undef A2_X
#undef A2_C
#undef A2_LR
#undef A2_DMA
#undef A2_FENCE

    // S(t+1) is consumed by the next iteration only: keep its computation in this one (P(t) and the sums are pinned piecewise)
    if constexpr (HAS_NEXT) { A2_PIN(sn[0]); A2_PIN(sn[1]); }
    if (bump) {  // O is in the old reference (it just received tile t-1): move it to the new one
      asm volatile("" ::: "memory");  // a real (rarely taken) branch
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
    }
    if constexpr (VDN_ATTN_ABL & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else stage_barrier();
#if VDN_ATTN_ABL & 64
    { const unsigned long long now = __builtin_amdgcn_s_memtime(); stamp_sum[4] += now - stamp_last; stamp_last = now; }
#endif
  };

  // ---- prologue: K_0 (and K_1) in flight, S(0) from buffer 0
  auto stage_k_all = [&](int buf, int t) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) { stage_k_piece(buf, t, i, false); stage_k_piece(buf, t, i, true); }
  };
  stage_k_all(0, 0);
  if (nt > 1) stage_k_all(1, 1);
  stage_barrier();
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    s[0][kb] = cross_hl(k8_read(smem, kb, 0), zero16);
    s[0][kb] = cross_lh(k8_read(smem, kb, 1), s[0][kb]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s[0][kb] = HT::mfma32(*(const V8*)(smem + k_addr(kb, ks)), qf[ks], s[0][kb]);
  }
  stage_barrier();  // every wave has read K_0 before iteration 0 lets K_2 overwrite it
  constexpr std::true_type Y{};
  constexpr std::false_type N{};
  if (nt == 1) {
    iter(0, N, N, A2_IC(0));
  } else {
    iter(0, N, Y, A2_IC(0));
    int t = 1;
    for (; t + 2 < nt; t += 2) {
      iter(t, Y, Y, A2_IC(1));
      iter(t + 1, Y, Y, A2_IC(0));
    }
    if (t + 1 < nt) { iter(t, Y, Y, A2_IC(1)); ++t; }
    if (t & 1) iter(t, Y, N, A2_IC(1));
    else iter(t, Y, N, A2_IC(0));
  }
  // ---- O += V_{nt-1} P(nt-1)
  {
    const char* sV = smem + ((nt - 1) & 1) * NT * TILE + TILE;
    auto pv_last = [&](const V8(&p)[2][2]) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          o[db] = HT::mfma32(*(const V8*)(sV + v_addr(db, c)), p[c >> 1][c & 1], o[db]);
          if constexpr (PV == 2) o[db] = HT::mfma32(*(const V8*)(sV + 2 * TILE + v_addr(db, c)), p[c >> 1][c & 1], o[db]);
        }
    };
    if ((nt - 1) & 1) pv_last(pf[1]);
    else pv_last(pf[0]);
  }

#if VDN_ATTN_ABL & 64
  if (blockIdx.x == 300 && tid == 0)  // one wave of a mid-grid workgroup reports (the output is garbage in this build anyway)
    for (int i = 0; i < 5; ++i) ((unsigned long long*)out)[i] = stamp_sum[i];
#endif
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < nq) {
    const int b = bh / H, hd = bh - b * H;
    const size_t row = (size_t)b * nq + q;
    const size_t oo = (row * H + hd) * 64;
    [[maybe_unused]] f16x32 hv6, lv6;   // out8: the lane's 32 channels (stream position 16 db + 4 g + e) = one half of the head's slab
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename HT::V4 v, vl;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          T a, b2;
          f[e] = o[db][4 * g + e] * inv;
          split_rtz(f[e], a, b2);
          v[e] = a;
          vl[e] = b2;
        }
        const int c = db * 32 + 8 * g + 4 * h;  // channel inside the head
        // out_kt: K-tile-major planes for the 8-bit cross-term GEMM that consumes the attention output (vdn_gemm_desc.a_kt):
        // halves [C/32][rows][32] (the head's two 32-channel tiles are 2 hd + db), bytes [C/64][rows][64] (tile hd)
        const size_t od = out_kt ? ((size_t)(2 * hd + db) * out_rows + row) * 32 + 8 * g + 4 * h : oo + c;
        *(typename HT::V4*)(out + od) = v;
        if (outl) *(typename HT::V4*)(outl + od) = vl;
        if constexpr (std::is_same_v<T, _Float16>) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { hv6[16 * db + 4 * g + e] = v[e]; lv6[16 * db + 4 * g + e] = vl[e]; }
        }
      }
    if constexpr (std::is_same_v<T, _Float16>) {
      if (out8) {  // 6-bit rows of the projection's A operand (common.hpp x6 rows, order 2): hi plane, then the remainder plane
        const int sb = x6_scale_byte(hv6);
        uint8_t* d8 = out8 + (out_kt ? ((size_t)hd * out_rows + row) * 64 : oo) + 32 * h;
        x6_store_half(d8, hv6, sb);
        x6_store_half(d8 + (size_t)out_rows * H * 64, lv6, sb - 10);
      }
    }
  }
#undef A2_IC
#undef A2_PIN
#undef A2_BLDS16
}
