// gemm_x8_kernel: out = epilogue(A W^T) with the split-precision cross terms on the block-scaled MFMA.
//
//   A W^T = A_hi W_hi^T                  fp16 v_mfma_f32_32x32x16_f16            (4 per 32x32 block and 64-deep slab)
//         + A_hi6 W_lo6^T + A_lo6 W_hi6^T  e3m2 v_mfma_scale_f32_32x32x64_f8f6f4   (2 per block and slab, 4.2x the fp16 rate)
//
// X_hi6 / X_lo6 are planes of "x6 rows" (common.hpp; include/vdn.h A8 / W8): per row and 64 of K two 32-byte halves of 32
// e3m2 codes + the E8M0 scale byte of the half (scale of the remainder plane = scale of the hi plane - 10). A cross term is
// 2^-11 of the product and e3m2 keeps 3 significant bits of it: 2^-14 per term, against 2^-22 for the third fp16 product it
// replaces and 2^-11 for dropping it. Bytes per element stay 2 + 1 + 1 (the two 16-byte LDS reads of a lane
// are its MFMA operand and its scale operand); matrix-pipe cycles per 32 x 32 block and 64 of K drop from 384 (three fp16
// products) to ~190. Round 3 first shipped these planes as e5m2 bytes (2.3x rate, ~240 cycles); the 6-bit rows took the
// block of four encoder linears from 640 to 615 us (not further: in the two cross-term phases the 48 KiB of fragment reads
// per wave group, 384 LDS cycles, now outlast the 240 matrix-pipe cycles; profiles/r03_x6.md).
//
// 256 x 256 tile, 8 waves (2 x 4; a wave owns 128 A rows x 64 W rows = 4 x 2 blocks of 32 x 32, 128 accumulator
// registers), weights as the MFMA's first operand, so a lane holds one activation row (lane & 31) and runs of 4
// consecutive output columns: the fp32 epilogues of gemm_kernels.hpp (emit4) apply unchanged.
//
// Structure = the ping-pong of gemm_x3_p8_kernel on 64-deep slabs. A slab is 4 UNITS of 32 KiB (256 rows x 64 B of A and
// of W each): fp16 k 0..31, fp16 k 32..63, [A_lo6 | W_hi6], [A_hi6 | W_lo6]; unit n = 4 slab + phase is read in global phase n
// and lives in ring slot n mod NSLOT. The two wave groups (waves 0-3 = A rows 0..127, waves 4-7 = rows 128..255; wave w
// and w + 4 share a SIMD) run ONE BARRIER APART: between two barriers one group issues 512 matrix-pipe cycles (16 fp16 or
// 8 scaled MFMAs) on fragments it already holds while the other reads its next 12 fragments and issues its 4 pieces
// (1 KiB each) of the unit L = NSLOT - 1 phases ahead.
//   WAR: a wave retires its fragment reads (lgkmcnt(0)) BEFORE the barrier in the middle of its phase, so after the
//        barrier that ends phase p of the later group every read of unit p has returned and the slot is re-issued in
//        phase p + 1 by either group (MI355X_MICROARCH.md 'Two waves per SIMD' item 7; guide '256^2 8-phase template').
//   RAW: a unit issued in phase p is read in phase p + L; before the middle barrier of phase q every wave waits for all
//        but its 4 (L - 1) newest pieces, i.e. for everything up to unit q + 1; the later group's pieces are covered one
//        barrier later, still before any read of unit q + 1.
// NSLOT = 5 uses all 160 KiB of LDS (4 units = 128 KiB in flight or being read); ring slots are addressed by a scalar
// base, fragment addresses are one VGPR per operand + immediates.
#include "gemm_kernels.hpp"

namespace vdn_gemm_impl {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int X8_BN = 256;
constexpr int X8_WH = 256 * 64;   // bytes of the W half of a unit: 256 rows x 64 B (32 fp16 or 64 e5m2 per row)
// BM = 256 | 192 rows of A per tile (chosen per launch so that the tile count fills whole rounds of the CUs, x8_entry):
// the A half of a unit is BM rows x 64 B, a wave owns BM / 2 rows = NI blocks of 32.
#ifndef VDN_X8_NSLOT
#define VDN_X8_NSLOT 5
#endif
#ifndef VDN_X8_ABL
#define VDN_X8_ABL 0   // timing-only builds (tools/build_variant.sh): 1 = no MFMAs, 2 = no DMA, 4 = no fragment reads, 8 = clock stamps
#endif
constexpr int X8_NSLOT = VDN_X8_NSLOT;
constexpr int x8_unit(int bm) { return bm * 64 + X8_WH; }  // 32 KiB (BM 256) or 28 KiB (BM 192)
constexpr int X8_L = X8_NSLOT - 1;  // issue lead in phases
// DMA instructions one wave issues for the unit of phase ph (`four`: the wave moves 2 A pieces, else 1)
constexpr int x8_pc(int ph, bool four) {
  const int a = four ? 2 : 1, w = 2;
  return a + w;
}
// ... for the k newest units, the newest being the unit of phase `newest`
constexpr int x8_inflight(int newest, int k, bool four) {
  int n = 0;
  for (int t = 0; t < k; ++t) n += x8_pc((newest - t) & 3, four);
  return n;
}
#define X8_GLDS(src, dst)                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// PAIR: the W rows of a tile are loaded in the order that gives every lane 8 consecutive output columns per block row
template <int STORE, bool PAIR, int X8_BM>
__global__ __launch_bounds__(512) void gemm_x8_kernel(const vdn_gemm_desc p) {
  using H = Half<VDN_F16>;
  using V8 = H::V8;
  static_assert(X8_BM == 256 || X8_BM == 192, "two wave groups of 4 or 3 blocks of 32 rows");
  constexpr int X8_H = X8_BM * 64, X8_U = x8_unit(X8_BM), NI = X8_BM / 64, AP = X8_BM / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + X8_BN - 1) / X8_BN, tiles_m = (p.M + X8_BM - 1) / X8_BM;
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  int tm_i, tn_i;
  {  // groups of 4 m-tiles walk n first (an XCD's tiles in flight share A rows and W columns through its L2)
    constexpr int GM = 4;
    const int per_group = GM * tiles_n;
    const int g = tile / per_group, r = tile - g * per_group;
    const int gm = (tiles_m - g * GM) < GM ? (tiles_m - g * GM) : GM;
    tn_i = r / gm;
    tm_i = g * GM + (r - tn_i * gm);
  }
  const int m0 = tm_i * X8_BM, n0 = tn_i * X8_BN;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;

  // ---- DMA geometry: an operand half of a unit is 16 pieces of 16 rows x 64 B, source chunk
  // (lane & 3) ^ ((-(lane >> 4)) & 3) (the image of gemm_x3_p8_kernel); wave w moves pieces w and w + 8 of the A half and
  // of the W half: 4 DMA per thread and phase. Per row the fp16 planes advance 64 B per phase, the byte planes 64 B per slab.
  const int lr = lane >> 2, chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  // Source address of a 16-byte chunk = plane + K-tile index * kts + row * rs + 16 chunk, with (rs, kts) per operand:
  //   row-major [rows, ld]        fp16: rs = 2 ld, kts = 64 (32 columns)      byte planes: rs = ld, kts = 64 (64 columns)
  //   K-tile-major (a_kt / w_kt)  fp16 [K/32][rows][32]: rs = 64, kts = 64 rows  bytes [K/64][rows][64]: rs = 64, kts = 64 rows
  // K-tile-major makes a 16-row piece ONE contiguous KiB (8 full 128-byte lines); a row-major piece touches 16 lines and
  // uses half of each: measured 44 vs 67 GB/s per CU of L2 -> LDS feed (profiles/r03_x8_ablations.log).
  const size_t a_rs = p.a_kt ? 64 : (size_t)p.lda * 2, a_kts = p.a_kt ? (size_t)p.M * 64 : 64;
  const size_t w_rs = p.w_kt ? 64 : (size_t)p.ldb * 2, w_kts = p.w_kt ? (size_t)p.N * 64 : 64;
  const size_t a8_rs = p.a_kt ? 64 : (size_t)p.lda, a8_kts = p.a_kt ? (size_t)p.M * 64 : 64;
  const size_t w8_rs = p.w_kt ? 64 : (size_t)p.ldb, w8_kts = p.w_kt ? (size_t)p.N * 64 : 64;
  // 32-bit byte offsets of this thread's two pieces (rows clamped; < 2^32: vdn_gemm rejects larger planes)
  unsigned ah_o[2], wh_o[2], a8_o[2], w8_o[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    // plane-output flavours (PAIR): LDS row rho of a W piece holds W row rho with bits 2 and 3 swapped, so that a lane's
    // accumulator registers 8 u .. 8 u + 7 of a block are 8 CONSECUTIVE output columns (16-byte stores; see the epilogue)
    const int lrw = PAIR ? ((lr & 3) | ((lr & 4) << 1) | ((lr & 8) >> 1)) : lr;
    int m = m0 + (wave + 8 * i) * 16 + lr, n = n0 + (wave + 8 * i) * 16 + lrw;
    m = m < p.M ? m : p.M - 1;
    n = n < p.N ? n : p.N - 1;
    ah_o[i] = (unsigned)(m * a_rs) + chunk * 16;
    wh_o[i] = (unsigned)(n * w_rs) + chunk * 16;
    a8_o[i] = (unsigned)(m * a8_rs) + chunk * 16;
    w8_o[i] = (unsigned)(n * w8_rs) + chunk * 16;
  }
  const char* Ah = (const char*)p.A;               // fp16 plane
  const char* Wh = (const char*)p.W;
  const char* A8v = (const char*)p.A8;             // x6 rows of A_hi; the remainder plane follows at + M K bytes
  const char* W8v = (const char*)p.W8;             // x6 rows of W_hi; remainder plane at + N ldb bytes
  const size_t a8l = (size_t)p.M * p.K, w8l = (size_t)p.N * p.ldb;
  // issue this wave's pieces of unit (slab, ph) into ring slot `slot` (ph is a compile-time constant)
  auto issue = [&](int slab, auto phc, int slot) {
    constexpr int ph = decltype(phc)::value;
    if constexpr (VDN_X8_ABL & 2) return;
    char* ua = smem + slot * X8_U;
    char* uw = ua + X8_H;
    const char *ba, *bw;  // wave-uniform bases
    if constexpr (ph < 2) {
      ba = Ah + (size_t)(2 * slab + ph) * a_kts;
      bw = Wh + (size_t)(2 * slab + ph) * w_kts;
    } else {  // phase 2: A_lo6 with W_hi6; phase 3: A_hi6 with W_lo6
      ba = A8v + (ph == 2 ? a8l : 0) + (size_t)slab * a8_kts;
      bw = W8v + (ph == 2 ? 0 : w8l) + (size_t)slab * w8_kts;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // the 32-bit lane offset is made opaque HERE so that its zero-extension is not hoisted out of the loop as a 64-bit
      // register pair: the DMA then takes the scalar-base + 32-bit-offset form (8 address registers instead of 16, no
      // 64-bit vector adds in the loop; block of four 620 -> 604 us)
      unsigned oa = ph < 2 ? ah_o[i] : a8_o[i], ow = ph < 2 ? wh_o[i] : w8_o[i];
      asm volatile("" : "+v"(oa), "+v"(ow));
      if (AP == 16 || wave + 8 * i < AP) X8_GLDS(ba + oa, ua + (wave + 8 * i) * 1024);
      X8_GLDS(bw + ow, uw + (wave + 8 * i) * 1024);
    }
  };
  // Counted waits: the K newest units may stay in flight, the newest being the unit of phase NEWPH. A wave's DMA instructions
  // per unit depend on the unit's phase (x8_pc): 2 A + 2 W pieces, 1 + 2 for waves 4-7 of the 192-row tile (12 A pieces over
  // 8 waves).
  const bool four = AP == 16 || wave < 4;
#define X8_WAIT(NEWPH, K)                                                                                     \
  do {                                                                                                        \
    if (four) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(x8_inflight((NEWPH), (K), true)) : "memory");          \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(x8_inflight((NEWPH), (K), false)) : "memory");              \
  } while (0)

  // ---- fragment addresses inside a unit (64-byte rows): 16-byte chunk c of row `row` sits at c ^ ((-(row >> 2)) & 3).
  // fp16 32x32x16 operand, k-step ks (0, 1) of the unit: chunk 2 ks + h. e3m2 32x32x64 operand: half h of the row-slab =
  // chunks 2 h, 2 h + 1.
  // The swizzle term depends on (row >> 2) & 3 = (r >> 2) & 3 only (a wave's blocks start at multiples of 32 rows), so
  // block i / j of an operand is the block-0 address plus an instruction immediate.
  const int swz = (0 - (r >> 2)) & 3;
  const int a_off0 = (wm * (X8_BM / 2) + r) * 64 + ((h ^ swz) << 4);
  const int w_off0 = X8_H + (wn * 64 + r) * 64 + ((h ^ swz) << 4);
  const int a8_off0 = (wm * (X8_BM / 2) + r) * 64 + (((2 * h) ^ swz) << 4);  // byte operand: chunk 2 h; chunk 2 h + 1 = bit 4 flipped
  const int w8_off0 = X8_H + (wn * 64 + r) * 64 + (((2 * h) ^ swz) << 4);
  auto rd8 = [&](const char* u, int off) {
    const u32x4 a0 = *(const u32x4*)(u + off), a1 = *(const u32x4*)(u + (off ^ 16));
    i32x8 v;   // registers 0..5: the 32 codes; register 6: the scale byte; 7: unused
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (int)a0[e]; v[4 + e] = (int)a1[e]; }
    return v;
  };

  f32x16 acc[NI][2];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- phase bodies: the fragment READS of a phase and its MFMAs
  const int m_alo = p.x8_terms == 1 ? 0 : -1, m_wlo = p.x8_terms == 2 ? 0 : -1;  // 0: that cross term is dropped (scale 2^-127)
  V8 hw[2][2] = {}, ha[2][NI] = {};  // fp16 phases: [k-step][block]
  i32x8 cw[2] = {}, ca[NI] = {};     // byte phases
  auto reads = [&](auto phc, int slot) {
    constexpr int ph = decltype(phc)::value;
    if constexpr (VDN_X8_ABL & 4) return;
    const char* u = smem + slot * X8_U;
    if constexpr (ph < 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) hw[ks][j] = *(const V8*)(u + (w_off0 ^ (ks << 5)) + j * 2048);
#pragma unroll
        for (int i = 0; i < NI; ++i) ha[ks][i] = *(const V8*)(u + (a_off0 ^ (ks << 5)) + i * 2048);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) cw[j] = rd8(u + j * 2048, w8_off0);
#pragma unroll
      for (int i = 0; i < NI; ++i) ca[i] = rd8(u + i * 2048, a8_off0);
    }
  };
  // fp16 phases: A_hi W_hi^T over a 32-deep unit (2 k-steps x 8 blocks). 6-bit phases: one cross term over the slab,
  // phase 2 = W_hi6 A_lo6^T, phase 3 = W_lo6 A_hi6^T.
  auto mfmas = [&](auto phc) {
    constexpr int ph = decltype(phc)::value;
    if constexpr (VDN_X8_ABL & 1) {  // keep the fragments live
      if constexpr (ph < 2) { for (int ks = 0; ks < 2; ++ks) { for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(hw[ks][j])); for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(ha[ks][i])); } }
      else { for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(cw[j])); for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(ca[i])); }
      return;
    }
    __builtin_amdgcn_s_setprio(1);
    if constexpr (ph < 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = H::mfma32(hw[ks][j], ha[ks][i], acc[i][j]);
            if (j == 1) __builtin_amdgcn_sched_barrier(0);   // issue order as written: 8 independent accumulators between two MFMAs on the same one
          }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // operands = registers 0..5 (32 e3m2 codes), E8M0 scale = byte 0 of register 6 (this lane's row and K half); a
          // launch that drops a cross term (include/vdn.h x8_terms) zeroes that plane's scale byte: 2^-127, the product
          // vanishes in the fp32 accumulator, no branch in the loop
          const int sw = cw[j][6], sa = ca[i][6];
          if constexpr (ph == 2) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cw[j], ca[i], acc[i][j], 3, 3, 0, sw, 0, sa & m_alo);
          else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cw[j], ca[i], acc[i][j], 3, 3, 0, sw & m_wlo, 0, sa);
          if (j == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  const int nslab = p.K >> 6;
  const int nunits = 4 * nslab;
#if VDN_X8_ABL & 8  // diagnostic build: in-kernel clock of the main loop (MI355X_MICROARCH.md 'DVFS give-back' item 6)
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  constexpr std::integral_constant<int, 0> P0{};
  constexpr std::integral_constant<int, 1> P1{};
  constexpr std::integral_constant<int, 2> P2{};
  constexpr std::integral_constant<int, 3> P3{};
  // ring slots of the unit being read and of the unit being issued (scalars, advanced once per phase)
  int rd_slot = 0, wr_slot = X8_L % X8_NSLOT;
  auto next = [](int s) { return s + 1 == X8_NSLOT ? 0 : s + 1; };

  // prologue: the first L units
  {
    int s = 0;
    auto pro = [&](int n, auto phc) {
      if (n < X8_L && n < nunits) { issue(n >> 2, phc, s); s = next(s); }
    };
    pro(0, P0); pro(1, P1); pro(2, P2); pro(3, P3);
    if constexpr (X8_L > 4) pro(4, P0);
  }
  // unit 0 has landed when all but the L - 1 newest units have (short K: fewer units were issued, wait for all)
  if (nunits >= X8_L) X8_WAIT((X8_L - 1) & 3, X8_L - 1);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // second group runs one barrier behind

  // one phase: [fragment reads of unit n | DMA of unit n + L | counted wait | reads retired] barrier
  // [MFMAs] barrier. At the wait of phase n the newest unit issued is n + L and everything up to n + 1 must have landed:
  // L - 1 units stay in flight.
#define X8_PHASE(PC, PH)                                                                    \
  do {                                                                                      \
    reads(PC, rd_slot);                                                                     \
    X8_ISSUE(PH);                                                                           \
    X8_WAIT(((PH) + X8_L) & 3, X8_L - 1);                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    mfmas(PC);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    rd_slot = next(rd_slot);                                                                \
    wr_slot = next(wr_slot);                                                                \
  } while (0)
  // phase ph of slab s issues unit 4 s + ph + L = phase (ph + L) & 3 of slab s + (ph + L) / 4
#define X8_ISSUE(PH) issue(s + ((PH) + X8_L) / 4, std::integral_constant<int, ((PH) + X8_L) & 3>{}, wr_slot)
  int s = 0;
  // steady state: every phase of slab s issues (the last unit issued is 4 s + 3 + L <= nunits - 1)
  for (; 4 * s + 3 + X8_L <= nunits - 1; ++s) {
    X8_PHASE(P0, 0);
    X8_PHASE(P1, 1);
    X8_PHASE(P2, 2);
    X8_PHASE(P3, 3);
  }
  // last slab (L = 4: exactly one slab is left and none of its phases issues): the units beyond n + 1 stay in flight, the
  // newest being the last unit of the launch (phase 3)
  static_assert(X8_L == 4, "the tail below is written for a lead of one slab");
#define X8_TAIL(PC, LEFT)                                                                   \
  do {                                                                                      \
    reads(PC, rd_slot);                                                                     \
    if constexpr ((LEFT) > 0) X8_WAIT(3, (LEFT));                                           \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    mfmas(PC);                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    __builtin_amdgcn_s_barrier();                                                           \
    rd_slot = next(rd_slot);                                                                \
  } while (0)
  // (written as a loop — it runs once — so that the last slab stays a block of its own: in straight-line code the register
  // allocator places the epilogue's spills between the last MFMAs)
  for (; s < nslab; ++s) {
    X8_TAIL(P0, 2);
    X8_TAIL(P1, 1);
    X8_TAIL(P2, 0);
    X8_TAIL(P3, 0);
  }
#undef X8_TAIL
#undef X8_ISSUE
#undef X8_PHASE
  if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count of the two groups
  __builtin_amdgcn_sched_barrier(0);  // nothing of the epilogue is scheduled into the last slab
#if VDN_X8_ABL & 8
  if (p.splitk_ws && tid == 0 && (blockIdx.x % 37) == 0) {  // a few workgroups report into scratch nothing else reads
    unsigned long long* dbg = (unsigned long long*)p.splitk_ws + (blockIdx.x / 37) * 2;
    dbg[0] = __builtin_amdgcn_s_memtime() - st_c0;
    dbg[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif

  // ---- epilogue: lane = activation row m (lane & 31); registers 4 g .. 4 g + 3 of block (i, j) = output columns
  // 32 j + 8 g + 4 h + {0..3} — or, with PAIR, registers 8 u .. 8 u + 7 = columns 32 j + 16 u + 8 h + {0..7}.
  // Every load (bias, LayerScale, residual) is issued BEFORE the stores it would otherwise queue behind: vmcnt counts loads
  // and stores in one order, so a load behind a store waits for that store's round trip (measured on the first version of
  // this epilogue: 42 us per fc1 tile round against 13 us of store bandwidth).
  const int mw = m0 + wm * (X8_BM / 2), nw = n0 + wn * 64;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, o4 = {1.f, 1.f, 1.f, 1.f};
  if constexpr (PAIR) {
    f32x4 bias8[2][2][2];  // [j][u][half]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int n = nw + 32 * j + 16 * u + 8 * h + 4 * q;
          bias8[j][u][q] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : z4;
        }
    if (STORE == VDN_STX_FC1 && p.out8) {
      // bias + GELU -> the fp16 hi plane and the two planes of 6-bit rows of the consuming GEMM's A operand. The lane's 32
      // values of block row i (stream position 16 j + 8 u + e = column 32 j + 16 u + 8 h + e of the wave's 64-wide slab) are
      // one HALF of the slab: its scale comes from the lane's own values, no exchange (common.hpp x6 rows, order 1).
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = mw + 32 * i + r;
        f16x32 hv, lv;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 g0 = gelu4(f32x4{acc[i][j][8 * u], acc[i][j][8 * u + 1], acc[i][j][8 * u + 2], acc[i][j][8 * u + 3]} + bias8[j][u][0]);
            const f32x4 g1 = gelu4(f32x4{acc[i][j][8 * u + 4], acc[i][j][8 * u + 5], acc[i][j][8 * u + 6], acc[i][j][8 * u + 7]} + bias8[j][u][1]);
            V8 hh, ll;
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              _Float16 h0, h1, l0, l1;
              split2_rtz(g0[e], g0[e + 1], h0, h1, l0, l1);
              hh[e] = h0; hh[e + 1] = h1; ll[e] = l0; ll[e + 1] = l1;
              split2_rtz(g1[e], g1[e + 1], h0, h1, l0, l1);
              hh[4 + e] = h0; hh[5 + e] = h1; ll[4 + e] = l0; ll[5 + e] = l1;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { hv[16 * j + 8 * u + e] = hh[e]; lv[16 * j + 8 * u + e] = ll[e]; }
            const int n = nw + 32 * j + 16 * u + 8 * h;
            if (m < p.M && n < p.N) {
              const size_t o = p.out_kt ? ((size_t)(n >> 5) * p.M + m) * 32 + (n & 31) : (size_t)m * p.ldc + n;
              *(V8*)((_Float16*)p.out + o) = hh;
              if (p.out_lo) *(V8*)((_Float16*)p.out_lo + o) = ll;
            }
          }
        if (m < p.M && nw < p.N) {
          const int sb = x6_scale_byte(hv);
          uint8_t* d8 = (uint8_t*)p.out8 + (p.out_kt ? ((size_t)(nw >> 6) * p.M + m) * 64 : (size_t)m * p.ldc + nw) + 32 * h;
          x6_store_half(d8, hv, sb);
          x6_store_half(d8 + (size_t)p.M * p.ldc, lv, sb - 10);
        }
      }
    } else if ((STORE == VDN_STX_HEADS || STORE == VDN_ST_HEADS) && nw < p.N && !p.transposed[nw / (p.heads * 64)] && p.dst8[nw / (p.heads * 64)]) {
      // Q / K head split with the attention's 8-bit planes (per token 64 B of e5m2(v) | 64 B of e5m2(remainder 2^10)). The wave's
      // 64-column slab is ONE head; a lane holds its 8-byte groups 16 u + 32 j + 8 h of both planes. Written as they come that
      // is sixteen 8-byte stores per token pair into 128-byte rows (15 us of the 165 us QKV launch); the two lanes of a token
      // swap halves instead (v_permlane32_swap: lane h keeps block j = h of each plane) and store 32 contiguous bytes per plane.
      const int hc = p.heads * 64, split = nw / hc, head = (nw - split * hc) >> 6;
      _Float16* dst = (_Float16*)p.dst[split];
      _Float16* dlo = (_Float16*)p.dst_lo[split];
      if (STORE == VDN_ST_HEADS && p.rope[split]) {
        // RoPE'd split (sam2 apply_rotary_enc; weights packed so that a pair's real part sits 16 columns before its imaginary
        // part): under the paired mapping both are in this lane — registers 0..7 and 8..15 of block j are pairs 16 j + 8 h + e —
        // and the 16 rotated values are the CONSECUTIVE output channels 32 j + 16 h + {0..15}: two 16-byte stores per plane,
        // 16 bytes per 8-bit plane. The (cos, sin) rows of block row i + 1 are fetched before the stores of block row i.
        f32x4 cs[2][2][4];   // [buffer][j][4 x (cos, sin, cos, sin)]
        auto fetch = [&](int i, f32x4 (&d)[2][4]) {
          const int m = mw + 32 * i + r, mc = m < p.M ? m : p.M - 1;
          const int tl = mc - (mc / p.tokens) * p.tokens;
          const float* t = p.rope_cs + (size_t)(tl % p.rope_mod) * 64 + 16 * h;
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) d[j][q] = *(const f32x4*)(t + 32 * j + 4 * q);
        };
        fetch(0, cs[0]);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (i + 1 < NI) fetch(i + 1, cs[(i + 1) & 1]);
          const int m = mw + 32 * i + r, mc = m < p.M ? m : p.M - 1;
          const int bt = mc / p.tokens, tk = mc - bt * p.tokens + p.tok_off;
          const size_t row = ((size_t)bt * p.heads + head) * p.tpad + tk;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float o[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float re = acc[i][j][e] + bias8[j][0][e >> 2][e & 3], im = acc[i][j][8 + e] + bias8[j][1][e >> 2][e & 3];
              const float cc = cs[i & 1][j][e >> 1][2 * (e & 1)], ss = cs[i & 1][j][e >> 1][2 * (e & 1) + 1];
              o[2 * e] = re * cc - im * ss;
              o[2 * e + 1] = re * ss + im * cc;
            }
            V8 hh[2], ll[2];
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
              _Float16 h0, h1, l0, l1;
              split2_rtz(o[e], o[e + 1], h0, h1, l0, l1);
              hh[e >> 3][e & 7] = h0; hh[e >> 3][(e & 7) + 1] = h1; ll[e >> 3][e & 7] = l0; ll[e >> 3][(e & 7) + 1] = l1;
            }
            if (m < p.M) {
              const size_t oo = row * 64 + 32 * j + 16 * h;
              *(V8*)(dst + oo) = hh[0];
              *(V8*)(dst + oo + 8) = hh[1];
              if (dlo) { *(V8*)(dlo + oo) = ll[0]; *(V8*)(dlo + oo + 8) = ll[1]; }
              const float k = VDN_LO8_SCALE;
              uint8_t* d8 = (uint8_t*)p.dst8[split] + row * 128 + 32 * j + 16 * h;
              *(u32x4*)d8 = u32x4{pk4_bf8(o[0], o[1], o[2], o[3]), pk4_bf8(o[4], o[5], o[6], o[7]), pk4_bf8(o[8], o[9], o[10], o[11]),
                                  pk4_bf8(o[12], o[13], o[14], o[15])};
              *(u32x4*)(d8 + 64) = u32x4{pk4_bf8(k * (float)ll[0][0], k * (float)ll[0][1], k * (float)ll[0][2], k * (float)ll[0][3]),
                                         pk4_bf8(k * (float)ll[0][4], k * (float)ll[0][5], k * (float)ll[0][6], k * (float)ll[0][7]),
                                         pk4_bf8(k * (float)ll[1][0], k * (float)ll[1][1], k * (float)ll[1][2], k * (float)ll[1][3]),
                                         pk4_bf8(k * (float)ll[1][4], k * (float)ll[1][5], k * (float)ll[1][6], k * (float)ll[1][7])};
            }
          }
        }
      } else
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = mw + 32 * i + r, mc = m < p.M ? m : p.M - 1;
        const int bt = mc / p.tokens, tk = mc - bt * p.tokens + p.tok_off;
        const size_t row = ((size_t)bt * p.heads + head) * p.tpad + tk;
        unsigned hi8[2][2][2], lo8[2][2][2];   // [j][u][dword]
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[i][j][8 * u + e] + bias8[j][u][0][e]; v[4 + e] = acc[i][j][8 * u + 4 + e] + bias8[j][u][1][e]; }
            V8 hh, ll;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
              _Float16 h0, h1, l0, l1;
              split2_rtz(v[e], v[e + 1], h0, h1, l0, l1);
              hh[e] = h0; hh[e + 1] = h1; ll[e] = l0; ll[e + 1] = l1;
            }
            if (m < p.M) {
              const size_t o = row * 64 + 32 * j + 16 * u + 8 * h;
              *(V8*)(dst + o) = hh;
              if (dlo) *(V8*)(dlo + o) = ll;
            }
            const float k = VDN_LO8_SCALE;
            hi8[j][u][0] = pk4_bf8(v[0], v[1], v[2], v[3]);
            hi8[j][u][1] = pk4_bf8(v[4], v[5], v[6], v[7]);
            lo8[j][u][0] = pk4_bf8(k * (float)ll[0], k * (float)ll[1], k * (float)ll[2], k * (float)ll[3]);
            lo8[j][u][1] = pk4_bf8(k * (float)ll[4], k * (float)ll[5], k * (float)ll[6], k * (float)ll[7]);
          }
        // after the swap: lanes h = 0 hold (own, partner's) group of block j = 0, lanes h = 1 (partner's, own) of block j = 1
        u32x4 bh[2], bl[2];   // 32 bytes of each plane: [u0 h0 | u0 h1], [u1 h0 | u1 h1]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            const auto sh = __builtin_amdgcn_permlane32_swap(hi8[0][u][w], hi8[1][u][w], false, false);
            const auto sl = __builtin_amdgcn_permlane32_swap(lo8[0][u][w], lo8[1][u][w], false, false);
            bh[u][w] = sh[0]; bh[u][2 + w] = sh[1];
            bl[u][w] = sl[0]; bl[u][2 + w] = sl[1];
          }
        if (m < p.M) {
          uint8_t* d8 = (uint8_t*)p.dst8[split] + row * 128 + 32 * h;
          *(u32x4*)d8 = bh[0];
          *(u32x4*)(d8 + 16) = bh[1];
          *(u32x4*)(d8 + 64) = bl[0];
          *(u32x4*)(d8 + 80) = bl[1];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const f32x4 a0 = {acc[i][j][8 * u], acc[i][j][8 * u + 1], acc[i][j][8 * u + 2], acc[i][j][8 * u + 3]};
            const f32x4 a1 = {acc[i][j][8 * u + 4], acc[i][j][8 * u + 5], acc[i][j][8 * u + 6], acc[i][j][8 * u + 7]};
            emit8<VDN_F16, STORE == VDN_ST_HEADS ? VDN_STX_HEADS : STORE>(p, mw + 32 * i + r, nw + 32 * j + 16 * u + 8 * h, a0, a1, bias8[j][u][0],
                                                                              bias8[j][u][1]);
          }
    }
  } else if constexpr (STORE == VDN_STX_RES) {  // (acc + bias) * gamma + f32 residual -> f32 rows (in place)
    f32x4 bias4[2][4], gam4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + 32 * j + 8 * g + 4 * h;
        bias4[j][g] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : z4;
        gam4[j][g] = (p.gamma && n < p.N) ? *(const f32x4*)(p.gamma + n) : o4;
      }
    f32x4 res[2][2][4];  // the residual of block row i + 1 is fetched before the stores of block row i
    auto fetch = [&](int i, f32x4 (&dst)[2][4]) {
      const int m = mw + 32 * i + r;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + 32 * j + 8 * g + 4 * h;
          dst[j][g] = (m < p.M && n < p.N) ? *(const f32x4*)((const float*)p.res1 + (size_t)m * p.ldr1 + n) : z4;
        }
    };
    fetch(0, res[0]);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (i + 1 < NI) fetch(i + 1, res[(i + 1) & 1]);
      const int m = mw + 32 * i + r;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nw + 32 * j + 8 * g + 4 * h;
          const f32x4 a = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (m < p.M && n < p.N) *(f32x4*)((float*)p.out + (size_t)m * p.ldc + n) = (a + bias4[j][g]) * gam4[j][g] + res[i & 1][j][g];
        }
    }
  } else {
    f32x4 bias4[2][4], gam4[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + 32 * j + 8 * g + 4 * h;
        bias4[j][g] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : z4;
        gam4[j][g] = (p.gamma && n < p.N) ? *(const f32x4*)(p.gamma + n) : o4;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + 32 * j + 8 * g + 4 * h;
        // `b` (GEGLU gate / RoPE partner) is the group 16 columns to the right = registers of g + 2; only read by flavours
        // that skip the groups where it would wrap
        const int gb = (g + 2) & 3;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const f32x4 a = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          const f32x4 b = {acc[i][j][4 * gb], acc[i][j][4 * gb + 1], acc[i][j][4 * gb + 2], acc[i][j][4 * gb + 3]};
          emit4<VDN_F16, STORE>(p, mw + 32 * i + r, n, a, b, bias4[j][g], bias4[j][gb], gam4[j][g]);
        }
      }
  }
}
#undef X8_WAIT
#undef X8_GLDS

template <int BM>
static int x8_launch(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + BM - 1) / BM) * ((d.N + X8_BN - 1) / X8_BN);
  const dim3 g(tiles), b(512);
  constexpr int LDS = X8_NSLOT * x8_unit(BM);
#define VDN_X8(ST, PAIR)                                                                                               \
  do {                                                                                                                   \
    static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_x8_kernel<ST, PAIR, BM>,                        \
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS); /* once */      \
    if (attr != hipSuccess) return -(1000 + (int)attr);                                                                  \
    hipLaunchKernelGGL((gemm_x8_kernel<ST, PAIR, BM>), g, b, LDS, s, d);                                                 \
  } while (0)
  int fl = epi_flavour(d);
  // plane-output flavours store 8 columns (16 bytes) per lane: column counts and strides must keep that aligned
  const bool a8 = !(d.N & 7) && !(d.ldc & 7) && !((uintptr_t)d.out & 15) && !((uintptr_t)d.out_lo & 15) && !((uintptr_t)d.out8 & 7);
  if (d.out8 && !(a8 && fl == VDN_STX_FC1 && !(d.N & 63))) return VDN_EUNSUPPORTED;  // 6-bit output rows: the paired bias + GELU epilogue only
  if (!a8 && fl != VDN_STX_RES && fl != VDN_STX_HEADS) fl = d.store;
  if (fl == VDN_STX_RESHALF1 || fl == VDN_STX_RESHALF2) fl = d.store;
  switch (fl) {
    case VDN_STX_FC1: VDN_X8(VDN_STX_FC1, true); break;
    case VDN_STX_HALF: VDN_X8(VDN_STX_HALF, true); break;
    case VDN_STX_HEADS: VDN_X8(VDN_STX_HEADS, true); break;
    case VDN_STX_RES: VDN_X8(VDN_STX_RES, false); break;
    case VDN_ST_GEGLU: VDN_X8(VDN_ST_GEGLU, false); break;
    case VDN_ST_HEADS: {
      // RoPE'd head splits: the paired epilogue when every split is a transposed one (V^T) or a Q / K split with its 8-bit planes
      bool pair = a8 && d.nsplit >= 1 && d.rope_cs && !((uintptr_t)d.rope_cs & 15);
      for (int i = 0; i < d.nsplit && pair; ++i)
        pair = d.dst[i] && (d.transposed[i] ? !d.rope[i] : d.dst8[i] != nullptr);
      if (pair) VDN_X8(VDN_ST_HEADS, true);
      else VDN_X8(VDN_ST_HEADS, false);
      break;
    }
    default: VDN_X8(VDN_ST_PLAIN, false); break;
  }
#undef VDN_X8
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

// M tile: the one that needs the least tile-row-rounds of the CUs this launch can count on (192-row tiles run 8 % slower
// per row: 12 instead of 16 MFMAs per phase between the same two barriers, and W is re-read per 192 instead of 256 rows)
int x8_entry(const vdn_gemm_desc& d, hipStream_t s) {
  const vdn_gemm_tuning& tu = tuning(d);
  const int cus = tu.cus > 0 ? tu.cus : (d.cu_hint > 0 && d.cu_hint <= 256 ? d.cu_hint : 256);
  const long tn = (d.N + X8_BN - 1) / X8_BN;
  auto cost = [&](int bm, double f) { const long t = (long)((d.M + bm - 1) / bm) * tn; return (double)((t + cus - 1) / cus) * bm * f; };
  int bm = tu.force_bm == 192 || tu.force_bm == 256 ? tu.force_bm : (cost(192, 1.08) < cost(256, 1.0) ? 192 : 256);
  return bm == 192 ? x8_launch<192>(d, s) : x8_launch<256>(d, s);
}

}  // namespace vdn_gemm_impl
