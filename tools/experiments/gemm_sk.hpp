// gemm_sk_kernel: the 256 x 256 ping-pong GEMM as a PERSISTENT, K-balanced ("stream-K") launch.
//
// The tile grids of the encoder linears do not divide the chip: at M = 10 960 the 256 x 256 tiles of qkv / fc1 / fc2
// make 516 / 688 / 172 workgroups = 2.02 / 2.69 / 0.67 rounds of 256 CUs, and in the round structure every CU runs its
// epilogue (and the HBM write burst that comes with it) at the same time while the matrix pipes idle. Here the launch
// is G <= 256 workgroups, one per CU, and the work is the flat sequence of (tile, K unit) pairs in tile order: workgroup
// g owns units [g U / G, (g + 1) U / G), i.e. the tail of one tile, some whole tiles and the head of another. A tile
// whose K range is shared by several workgroups is finished by the LAST of them to arrive:
//   * every part stores its fp32 accumulators to its slab (register order: 1 KiB per wave-instruction), drains the
//     stores, and one lane publishes with an agent-scope release + a ticket from the tile's counter;
//   * the part that draws the last ticket acquires, adds the other parts (two parts: own registers + the other slab —
//     fp32 addition commutes, so the sum does not depend on who came last; more parts: all slabs in part order, its
//     own included), runs the ordinary fused epilogue and resets the counter for the next launch.
// No workgroup ever waits for another one (nothing spins), so the launch cannot deadlock whatever else shares the GPU
// (a second lane's kernels, another process), and the result is bitwise reproducible for a given (shape, G).
// Main loop, LDS image, DMA units and the counted waits are those of gemm_x3_p8_kernel (gemm_kernels.hpp).
#pragma once
#include "gemm_kernels.hpp"

namespace vdn_gemm_impl {

struct SkPlan {
  int tiles_m, tiles_n;  // 256 x 256 tiles
  int upt;               // K units per tile (unit = one 32-deep K tile of the x3 loop)
  int G;                 // workgroups (<= total units)
  int total;             // tiles * upt (total * G < 2^31: checked by sk_plan)
  float* slabs;          // [2 G][8 waves][32 regs][64 lanes] f32x4: slot 2 g = g's first segment, 2 g + 1 = its last
  int* flags;            // [tiles] arrival counters, zero between launches
};

constexpr int SK_SLAB_FLOATS = 256 * 256;
constexpr int SK_LDS = 131072 + 16;  // the p8 ring + the "am I last" word

// workgroup g owns units [sk_start(g), sk_start(g + 1)); sk_owner(u) = the workgroup whose range holds unit u
__device__ __forceinline__ int sk_owner(int u, int G, int U) { return (int)(((unsigned)(u + 1) * (unsigned)G - 1u) / (unsigned)U); }
__device__ __forceinline__ int sk_start(int g, int G, int U) { return (int)((unsigned)g * (unsigned)U / (unsigned)G); }

template <int DT, int STORE>
__global__ __launch_bounds__(512) void gemm_x3_sk_kernel(const vdn_gemm_desc p, const SkPlan sk) {
  using H = Half<DT>;
  using V8 = typename H::V8;
  using T = typename H::T;
  constexpr int BM = 256, BN = 256, BK3 = 32;
  constexpr int A_TILE = BM * 64, W_TILE = BN * 64;  // bytes per operand plane and stage
  constexpr int STAGE = 2 * A_TILE + 2 * W_TILE;     // A_hi | A_lo | W_hi | W_lo
  constexpr int TQ = BM / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int g = xcd_remap(blockIdx.x, sk.G);  // an XCD's workgroups own one contiguous stretch of the tile order
  const int U = sk.total;
  const int u_end = sk_start(g + 1, sk.G, U);
  const int tiles_n = sk.tiles_n, tiles_m = sk.tiles_m, upt = sk.upt;

  // ---- per-lane constants of the DMA and fragment geometry (gemm_x3_p8_kernel, BM 256: every wave carries both planes
  // of one 16-row piece of each of the four units)
  const int lr = lane >> 2;
  const int chunk = (lane & 3) ^ ((0 - (lane >> 4)) & 3);
  const int pa[2] = {(wave / TQ) * 2 * TQ + wave % TQ, (wave / TQ) * 2 * TQ + TQ + wave % TQ};
  const int pw[2] = {4 * (wave >> 1) + (wave & 1), 4 * (wave >> 1) + (wave & 1) + 2};
  const ptrdiff_t a_delta = (const char*)p.A_lo - (const char*)p.A;
  const ptrdiff_t w_delta = (const char*)p.W_lo - (const char*)p.W;
  int* last_word = (int*)(smem + 2 * STAGE);

  for (int u = sk_start(g, sk.G, U); u < u_end;) {
    const int tile = u / upt;
    const int k0 = u - tile * upt;
    const int k1 = (upt - k0) < (u_end - u) ? upt : k0 + (u_end - u);
    u += k1 - k0;
    int tm_i, tn_i;
    {  // groups of 4 m-tiles walk n first (gemm_x3_p8_kernel)
      constexpr int GM = 4;
      const int per_group = GM * tiles_n;
      const int gi = tile / per_group, r = tile - gi * per_group;
      const int gm = (tiles_m - gi * GM) < GM ? (tiles_m - gi * GM) : GM;
      tn_i = r / gm;
      tm_i = gi * GM + (r - tn_i * gm);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int nk = k1 - k0;
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

    const char* ap[2];
    const char* wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + pa[i] * 16 + lr;
      m = m < p.M ? m : p.M - 1;
      int n = n0 + (vdn_pair8<STORE> ? (pw[i] >> 2) * 64 + pair8_col(pw[i] & 3, lr) : pw[i] * 16 + lr);
      n = n < p.N ? n : p.N - 1;
      ap[i] = (const char*)((const T*)p.A + (size_t)m * p.lda + chunk * 8) + (size_t)k0 * 64;
      wp[i] = (const char*)((const T*)p.W + (size_t)n * p.ldb + chunk * 8) + (size_t)k0 * 64;
    }
#define VDN_GLDS(src, dst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
    // unit u of the next K tile into stage `buf`: 0 = A sub-half 0, 1 = W sub-half 0, 2 = W sub-half 1, 3 = A sub-half 1
    auto issue = [&](auto uc, int buf) {
      constexpr int un = decltype(uc)::value;
      char* s0 = smem + buf * STAGE;
      if constexpr (un == 0 || un == 3) {
        constexpr int i = un == 0 ? 0 : 1;
        char* dst = s0 + pa[i] * 1024;
        VDN_GLDS(ap[i], dst);
        VDN_GLDS(ap[i] + a_delta, dst + A_TILE);
        ap[i] += 64;
      } else {
        constexpr int i = un == 1 ? 0 : 1;
        char* dst = s0 + 2 * A_TILE + pw[i] * 1024;
        VDN_GLDS(wp[i], dst);
        VDN_GLDS(wp[i] + w_delta, dst + W_TILE);
        wp[i] += 64;
      }
    };
#undef VDN_GLDS

    f32x4 acc[2 * TQ][4];
#pragma unroll
    for (int i = 0; i < 2 * TQ; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    V8 ah[TQ], al[TQ], bh[2], bl[2];
    auto read_a = [&](const char* s0, int qa) {
#pragma unroll
      for (int t = 0; t < TQ; ++t) {
        ah[t] = *(const V8*)(s0 + a_off[qa * TQ + t]);
        al[t] = *(const V8*)(s0 + A_TILE + a_off[qa * TQ + t]);
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
#define VDN_PHASE(READS, ISSUE, VA, QA, QB)                      \
  do {                                                           \
    READS;                                                       \
    ISSUE;                                                       \
    asm volatile("s_waitcnt vmcnt(" #VA ")" ::: "memory");       \
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

    issue(U0, 0);
    issue(U1, 0);
    issue(U2, 0);
    issue(U3, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();  // second group runs one barrier behind

    for (int kt = 0; kt + 1 < nk; ++kt) {
      const char* sc = smem + (kt & 1) * STAGE;
      const int nb = (kt + 1) & 1;
      VDN_PHASE((read_a(sc, 0), read_b(sc, 0)), issue(U0, nb), 4, 0, 0);
      VDN_PHASE(read_b(sc, 1), issue(U1, nb), 4, 0, 1);
      VDN_PHASE(read_a(sc, 1), issue(U2, nb), 4, 1, 1);
      VDN_PHASE(read_b(sc, 0), issue(U3, nb), 4, 1, 0);
    }
    {
      const char* sc = smem + ((nk - 1) & 1) * STAGE;
      VDN_PHASE((read_a(sc, 0), read_b(sc, 0)), (void)0, 2, 0, 0);
      VDN_PHASE(read_b(sc, 1), (void)0, 0, 0, 1);
      VDN_PHASE(read_a(sc, 1), (void)0, 0, 1, 1);
      VDN_PHASE(read_b(sc, 0), (void)0, 0, 1, 0);
    }
#undef VDN_PHASE
    if (wm == 0) __builtin_amdgcn_s_barrier();  // balance the barrier count of the two groups

    if (nk != upt) {
      // ---- a part of a shared tile: publish the partial sums; the last part to arrive finishes the tile
      const int t0u = tile * upt;
      const int g_first = sk_owner(t0u, sk.G, U), g_last = sk_owner(t0u + upt - 1, sk.G, U);
      const int parts = g_last - g_first + 1;
      auto slab_of = [&](int gp) {
        const int second = sk_start(gp, sk.G, U) / upt != tile;  // the tile is not the one gp's range starts in
        return sk.slabs + (size_t)(2 * gp + second) * SK_SLAB_FLOATS + (size_t)wave * (32 * 64 * 4) + lane * 4;
      };
      float* mine = slab_of(g);
#pragma unroll
      for (int i = 0; i < 2 * TQ; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(mine + (i * 4 + j) * 256) = acc[i][j];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(sk.flags + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = t == parts - 1;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(sk.flags + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // clean for the next launch
        }
        *last_word = last;
      }
      __syncthreads();
      const int last = *(volatile int*)last_word;
      __syncthreads();  // the word is rewritten by the next shared tile
      if (!last) continue;
      if (parts == 2) {
        const float* other = slab_of(g == g_first ? g_last : g_first);
#pragma unroll
        for (int i = 0; i < 2 * TQ; ++i) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += *(const f32x4*)(other + (i * 4 + j) * 256);
          __builtin_amdgcn_sched_barrier(0);  // 16 registers of loads in flight at a time: the accumulators fill the file
        }
      } else {
        for (int gp = g_first; gp <= g_last; ++gp) {
          const float* sp = slab_of(gp);
#pragma unroll
          for (int i = 0; i < 2 * TQ; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const f32x4 v = *(const f32x4*)(sp + (i * 4 + j) * 256);
              acc[i][j] = gp == g_first ? v : acc[i][j] + v;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    epilogue_regs<DT, 2 * TQ, 4, STORE, vdn_pair8<STORE>>(acc, p, m0 + wm * (BM / 2), n0 + wn * 64, lane);
  }
}

// Which launches take the stream-K kernel, and its plan. Returns false when the shape / workspace does not qualify.
inline bool sk_plan(const vdn_gemm_desc& d, SkPlan& sk) {
  if (!d.sk_flags || !d.splitk_ws || d.a_mode != VDN_A_PLAIN || d.relu_a || !d.A_lo || !d.W_lo || (d.K & 31) || d.N < 192) return false;
  sk.tiles_m = (d.M + 255) / 256;
  sk.tiles_n = (d.N + 255) / 256;
  sk.upt = d.K / 32;
  const long tiles = (long)sk.tiles_m * sk.tiles_n;
  if (tiles * sk.upt >= (1L << 22)) return false;  // (total + 1) * G stays below 2^31 in the kernel's index math
  sk.total = (int)(tiles * sk.upt);
  const int cus = tuning().cus > 0 ? tuning().cus : (d.cu_hint > 0 && d.cu_hint <= 256 ? d.cu_hint : 256);
  sk.G = sk.total < cus ? (int)sk.total : cus;
  sk.G &= ~7;  // whole XCD rounds (xcd_remap keeps an XCD's workgroups on neighbouring tiles either way)
  if (sk.G < 8) return false;
  // every workgroup gets at least 8 K tiles of work, and no tile is cut into more than ~8 parts
  if (sk.total / sk.G < 8 || tiles * 8 < sk.G) return false;
  if ((long)sk.G * 2 * SK_SLAB_FLOATS * 4 > d.splitk_ws_bytes || tiles * 4 > d.sk_flags_bytes) return false;
  sk.slabs = (float*)d.splitk_ws;
  sk.flags = (int*)d.sk_flags;
  return true;
}

template <int DT>
int launch_sk(const vdn_gemm_desc& d, const SkPlan& sk, hipStream_t s) {
  int fl = epi_flavour(d);
  const bool a8 = !(d.N & 7) && !(d.ldc & 7) && !((uintptr_t)d.out & 15) && !((uintptr_t)d.out_lo & 15);
  if (!a8 && fl != VDN_STX_RES && fl != VDN_STX_HEADS) fl = d.store;
  if (fl == VDN_STX_RESHALF1 || fl == VDN_STX_RESHALF2) fl = d.store;
  const dim3 g(sk.G), b(512);
#define VDN_SK(ST) hipLaunchKernelGGL((gemm_x3_sk_kernel<DT, ST>), g, b, SK_LDS, s, d, sk)
  switch (fl) {
    case VDN_ST_PLAIN: VDN_SK(VDN_ST_PLAIN); break;
    case VDN_STX_HALF: VDN_SK(VDN_STX_HALF); break;
    case VDN_ST_CONVT: VDN_SK(VDN_ST_CONVT); break;
    case VDN_ST_GEGLU: VDN_SK(VDN_ST_GEGLU); break;
    case VDN_STX_FC1: VDN_SK(VDN_STX_FC1); break;
    case VDN_STX_RES: VDN_SK(VDN_STX_RES); break;
    case VDN_STX_HEADS: VDN_SK(VDN_STX_HEADS); break;
    default: VDN_SK(VDN_ST_HEADS); break;
  }
#undef VDN_SK
  VDN_CHECK_LAUNCH();
  return VDN_OK;
}

}  // namespace vdn_gemm_impl
