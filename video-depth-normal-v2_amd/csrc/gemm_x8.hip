// gemm_x8_kernel: out = epilogue(A W^T) with the split-precision cross terms on the block-scaled 8-bit MFMA.
//
//   A W^T = A_hi W_hi^T                      fp16 v_mfma_f32_32x32x16_f16            (4 per 32x32 block and 64-deep slab)
//         + 2^-10 (A8 W_lo8^T + A_lo8 W8^T)  e5m2 v_mfma_scale_f32_32x32x64_f8f6f4   (2 per block and slab, 2.3x the fp16 rate)
//
// X8 = e5m2(X), X_lo8 = e5m2((X - X_hi) 2^10) come as two byte planes per operand, shaped like the fp16 operand
// (u8 [2, rows, K]; include/vdn.h A8 / W8). A cross term is 2^-11 of the product and e5m2 keeps 3 significant bits of it: 2^-14 per term, against 2^-22 for the
// third fp16 product it replaces and 2^-11 for dropping it. Bytes per element are unchanged (2 + 1 + 1).
//
// 256 x 256 tile, 8 waves (2 x 4; a wave owns 128 A rows x 64 W rows = 4 x 2 blocks of 32 x 32, 128 accumulator
// registers), weights as the MFMA's first operand, so a lane holds one activation row (lane & 31) and runs of 4
// consecutive output columns: the fp32 epilogues of gemm_kernels.hpp (emit4) apply unchanged.
//
// The scaled MFMA needs 64 of K at once, so the unit of the main loop is a 64-deep SLAB = 128 KiB of LDS = all the LDS
// the kernel has: 8 units of 16 KiB (256 rows x 64 B each: A_hi / W_hi of k 0..31, of k 32..63, and the byte planes
// W8 + A_lo8, W_lo8 + A8), consumed two at a time by 4 uniform phases and refilled as soon as their phase has been read
// (see the loop). There is no second buffer at slab granularity — which is also why this kernel does NOT beat the
// 3-product ping-pong kernel (measurements at the loop): it is an experiment, reachable only through explicit A8 / W8.
#include "gemm_kernels.hpp"

namespace vdn_gemm_impl {

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int X8_BM = 256, X8_BN = 256;
constexpr int X8_U = 256 * 64;  // bytes of one unit: 256 rows x 64 B (32 fp16 or 64 e5m2 per row), 16 KiB
// LDS map, by phase: [A_hi k0-31 | W_hi k0-31] [A_hi k32-63 | W_hi k32-63] [A_lo8 | W8] [A8 | W_lo8]
constexpr int X8_LDS = 8 * X8_U;  // 131072

#define X8_GLDS(src, dst)                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

template <int STORE>
__global__ __launch_bounds__(512) void gemm_x8_kernel(const vdn_gemm_desc p) {
  using H = Half<VDN_F16>;
  using V8 = H::V8;
  using T = H::T;
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

  // ---- DMA geometry: every unit is 16 pieces of 16 rows x 64 B, source chunk (lane & 3) ^ ((-(lane >> 4)) & 3) (the
  // image of gemm_x3_p8_kernel); wave w moves pieces w and w + 8 of the A unit and of the W unit of a phase: 4 DMA per
  // thread and phase. Per row the fp16 planes advance 64 B per phase pair, the byte planes 64 B per slab.
  const int lr = lane >> 2, chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  // 32-bit byte offsets of this thread's two pieces inside the fp16 planes (rows clamped; < 2^31: M K <= 2^30 checked on
  // the host) and inside the byte planes; the slab / phase / plane part of an address is wave-uniform (scalar)
  unsigned ah_o[2], wh_o[2];  // the byte planes' offsets are (fp16 offset + 16 chunk) / 2: row * K + 16 chunk
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + (wave + 8 * i) * 16 + lr, n = n0 + (wave + 8 * i) * 16 + lr;
    m = m < p.M ? m : p.M - 1;
    n = n < p.N ? n : p.N - 1;
    ah_o[i] = ((unsigned)m * (unsigned)p.lda + chunk * 8) * 2;
    wh_o[i] = ((unsigned)n * (unsigned)p.ldb + chunk * 8) * 2;
  }
  const unsigned c16 = (unsigned)chunk * 16;
  const char* Ah = (const char*)p.A;               // fp16 [M, K]
  const char* Wh = (const char*)p.W;               // fp16 [N, ldb]
  const char* A8v = (const char*)p.A8;             // e5m2(A) [M, K]; the remainder plane follows at + M K bytes
  const char* W8v = (const char*)p.W8;             // e5m2(W) [N, ldb]; remainder plane at + N ldb bytes
  const size_t a8l = (size_t)p.M * p.K, w8l = (size_t)p.N * p.ldb;
  // issue the two units of phase `ph` of slab `slab` into LDS units 2 ph, 2 ph + 1 (ph is a compile-time constant)
  auto issue = [&](int slab, auto phc) {
    constexpr int ph = decltype(phc)::value;
    char* ua = smem + (2 * ph) * X8_U;
    char* uw = ua + X8_U;
    const char *ba, *bw;  // wave-uniform bases
    if constexpr (ph < 2) {
      ba = Ah + (size_t)slab * 128 + ph * 64;
      bw = Wh + (size_t)slab * 128 + ph * 64;
    } else {  // phase 2: A_lo8 with W8; phase 3: A8 with W_lo8
      ba = A8v + (ph == 2 ? a8l : 0) + (size_t)slab * 64;
      bw = W8v + (ph == 2 ? 0 : w8l) + (size_t)slab * 64;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      X8_GLDS(ba + (ph < 2 ? ah_o[i] : (ah_o[i] + c16) >> 1), ua + (wave + 8 * i) * 1024);
      X8_GLDS(bw + (ph < 2 ? wh_o[i] : (wh_o[i] + c16) >> 1), uw + (wave + 8 * i) * 1024);
    }
  };

  // ---- fragment addresses inside a unit (64-byte rows): 16-byte chunk c of row `row` sits at c ^ ((-(row >> 2)) & 3).
  // fp16 32x32x16 operand, k-step ks (0, 1) of the unit: chunk 2 ks + h. e5m2 32x32x64 operand: bytes 32 h .. 32 h + 31 =
  // chunks 2 h, 2 h + 1.
  // The swizzle term depends on (row >> 2) & 3 = (r >> 2) & 3 only (a wave's blocks start at multiples of 32 rows), so
  // block i / j of an operand is the block-0 address plus an instruction immediate: 4 address registers in all.
  const int swz = (0 - (r >> 2)) & 3;
  const int a_off0 = (wm * 128 + r) * 64 + ((h ^ swz) << 4);
  const int w_off0 = X8_U + (wn * 64 + r) * 64 + ((h ^ swz) << 4);
  const int a8_off0 = (wm * 128 + r) * 64 + (((2 * h) ^ swz) << 4);  // byte operand: chunk 2 h; chunk 2 h + 1 = bit 4 flipped
  const int w8_off0 = X8_U + (wn * 64 + r) * 64 + (((2 * h) ^ swz) << 4);
  auto rd8 = [&](const char* u, int off) {
    const u32x4 a0 = *(const u32x4*)(u + off), a1 = *(const u32x4*)(u + (off ^ 16));
    i32x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (int)a0[e]; v[4 + e] = (int)a1[e]; }
    return v;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- phase bodies, split into the fragment READS of a phase and its MFMAs (48 fragment registers either way)
  V8 hw[2][2], ha[2][4];  // fp16 phases: [k-step][block]
  i32x8 cw[2], ca[4];     // byte phases
  auto reads = [&](auto phc) {
    constexpr int ph = decltype(phc)::value;
    const char* u = smem + (2 * ph) * X8_U;
    if constexpr (ph < 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) hw[ks][j] = *(const V8*)(u + (w_off0 ^ (ks << 5)) + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) ha[ks][i] = *(const V8*)(u + (a_off0 ^ (ks << 5)) + i * 2048);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) cw[j] = rd8(u + j * 2048, w8_off0);
#pragma unroll
      for (int i = 0; i < 4; ++i) ca[i] = rd8(u + i * 2048, a8_off0);
    }
  };
  // fp16 phases: A_hi W_hi^T over a 32-deep unit pair (2 k-steps x 8 blocks). Byte phases: one cross term over the slab,
  // phase 2 = W8 A_lo8^T, phase 3 = W_lo8 A8^T (the remainder planes carry 2^10; the E8M0 scale of that operand removes it).
  auto mfmas = [&](auto phc) {
    constexpr int ph = decltype(phc)::value;
    __builtin_amdgcn_s_setprio(1);
    if constexpr (ph < 2) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = H::mfma32(hw[ks][j], ha[ks][i], acc[i][j]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (ph == 2) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cw[j], ca[i], acc[i][j], 1, 1, 0, 127, 0, VDN_LO8_E8M0);
          else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cw[j], ca[i], acc[i][j], 1, 1, 0, VDN_LO8_E8M0, 0, 127);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- main loop, lock-step: global phase p (= 4 slab + ph) reads unit pair U_{p mod 4}, which is refilled with the data
  // of phase p + 4 as soon as every wave has read it: three phases of lead and, at any time, three unit pairs (96 KiB) in
  // flight while one is being read. DMAs complete in issue order and a thread issues 4 per phase, so "all but the three
  // newest phases" = vmcnt(12); the last slab issues nothing and counts down 12, 8, 4, 0.
  //
  // Measured (fc2 shape, M = 10960, N = 1024, K = 4096, tools/x8_bench.py; gemm_x3_p8_kernel = 3.5 us per 64 of K):
  //   this loop 4.05 us per slab; a ping-pong variant (two wave groups one barrier apart, refill two phases later,
  //   vmcnt(8) = 64 KiB in flight) 5.3 us, of which DMA-only 3.3 us and MFMA + fragment reads only 2.7 us (ablated builds).
  //   The feed is LDS-capacity bound: bytes in flight / DMA latency (~1.65 us under load) = 39 GB/s per CU with 64 KiB,
  //   ~58 GB/s with 96 KiB, i.e. >= 2.2 us per 128-KiB slab — a slab-sized LDS leaves no room to prefetch deeper.
  //   A 3-bytes-per-element variant was also built and timed (e5m2(X_hi) derived from the fp16 fragments in registers,
  //   remainder planes stored in the MFMA's slot order, 96 KiB per slab in a 5-slot ring with 128 KiB in flight): 3.86 us
  //   per slab — with the feed out of the way the loop is bound by its own structure (6 barriers and 36 fragment reads per
  //   slab around 48 MFMAs per wave), not by bytes. Both lose to gemm_x3_p8_kernel, whose 8-phase ping-pong hides exactly
  //   that; an 8-bit kernel would need the same treatment on a 3-phase slab. Kept as a tested experiment (the engines do
  //   not pass A8 / W8).
  const int nslab = p.K >> 6;
  constexpr std::integral_constant<int, 0> P0{};
  constexpr std::integral_constant<int, 1> P1{};
  constexpr std::integral_constant<int, 2> P2{};
  constexpr std::integral_constant<int, 3> P3{};
  issue(0, P0);
  issue(0, P1);
  issue(0, P2);
  issue(0, P3);
#define X8_PHASE(PC, WAIT, NEXT)                                                                       \
  do {                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #WAIT ")" ::: "memory");                                           \
    __builtin_amdgcn_s_barrier(); /* every thread's share of this phase's units has landed */          \
    reads(PC);                                                                                         \
    mfmas(PC);                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_s_barrier(); /* every wave has read them */                                       \
    NEXT;                                                                                              \
  } while (0)
  for (int s = 0; s + 1 < nslab; ++s) {
    X8_PHASE(P0, 12, issue(s + 1, P0));
    X8_PHASE(P1, 12, issue(s + 1, P1));
    X8_PHASE(P2, 12, issue(s + 1, P2));
    X8_PHASE(P3, 12, issue(s + 1, P3));
  }
  X8_PHASE(P0, 12, (void)0);
  X8_PHASE(P1, 8, (void)0);
  X8_PHASE(P2, 4, (void)0);
  X8_PHASE(P3, 0, (void)0);
#undef X8_PHASE

  // ---- epilogue: lane = activation row m (lane & 31); registers 4 g .. 4 g + 3 of block (i, j) = output columns
  // 32 j + 8 g + 4 h + {0..3}; `b` (GEGLU gate / RoPE partner) is the group 16 columns to the right = registers of g + 2
  const int mw = m0 + wm * 128, nw = n0 + wn * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nw + 32 * j + 8 * g + 4 * h;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, o4 = {1.f, 1.f, 1.f, 1.f};
      const f32x4 bias_a = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : z4;
      const f32x4 bias_b = (p.bias && n + 16 < p.N) ? *(const f32x4*)(p.bias + n + 16) : z4;
      const f32x4 gam = (p.gamma && n < p.N) ? *(const f32x4*)(p.gamma + n) : o4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4 a = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        const int gb = (g + 2) & 3;  // only read by flavours that skip the groups where it would wrap
        const f32x4 b = {acc[i][j][4 * gb], acc[i][j][4 * gb + 1], acc[i][j][4 * gb + 2], acc[i][j][4 * gb + 3]};
        emit4<VDN_F16, STORE>(p, mw + 32 * i + r, n, a, b, bias_a, bias_b, gam);
      }
    }
}
#undef X8_GLDS

int x8_entry(const vdn_gemm_desc& d, hipStream_t s) {
  const int tiles = ((d.M + X8_BM - 1) / X8_BM) * ((d.N + X8_BN - 1) / X8_BN);
  const dim3 g(tiles), b(512);
#define VDN_X8(ST) hipLaunchKernelGGL((gemm_x8_kernel<ST>), g, b, X8_LDS, s, d)
  switch (epi_flavour(d)) {
    case VDN_STX_FC1: VDN_X8(VDN_STX_FC1); break;
    case VDN_STX_RES: VDN_X8(VDN_STX_RES); break;
    case VDN_STX_HEADS: VDN_X8(VDN_STX_HEADS); break;
    case VDN_STX_HALF: VDN_X8(VDN_STX_HALF); break;
    case VDN_ST_GEGLU: VDN_X8(VDN_ST_GEGLU); break;
    case VDN_ST_HEADS: VDN_X8(VDN_ST_HEADS); break;
    default: VDN_X8(VDN_ST_PLAIN); break;
  }
#undef VDN_X8
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

}  // namespace vdn_gemm_impl
