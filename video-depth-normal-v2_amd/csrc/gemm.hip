// vdn_gemm: argument validation + dispatch. Kernels live in gemm_kernels.hpp and are instantiated
// per operand dtype in gemm_small_{f16,bf16}.hip / gemm_big_{f16,bf16}.hip (parallel compilation).
#include "common.hpp"
#include <stdlib.h>

namespace vdn_gemm_impl {
int launch_f16(const vdn_gemm_desc& d, hipStream_t s);
int launch_bf16(const vdn_gemm_desc& d, hipStream_t s);

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static float env_float(const char* name, float dflt) {
  const char* e = getenv(name);
  return e ? (float)atof(e) : dflt;
}
static const vdn_gemm_tuning& tuning_defaults() {
  static const vdn_gemm_tuning t = [] {  // the environment is read here, once per process; immutable afterwards
    vdn_gemm_tuning v;
    v.force_bm = env_int("VDN_GEMM_BM", 0);
    v.p8 = env_int("VDN_GEMM_P8", 1);
    v.no_splitk = getenv("VDN_GEMM_NOSPLITK") != nullptr;
    v.no_pipe = getenv("VDN_GEMM_NOPIPE") != nullptr;
    v.splitk_p8 = env_int("VDN_SPLITK_P8", 0);
    v.cus = env_int("VDN_GEMM_CUS", 0);
    v.splitk_occ = env_int("VDN_SPLITK_OCC", 50);
    v.splitk_max = env_int("VDN_SPLITK_MAX", 8);
    v.min_tiles = env_int("VDN_GEMM_MIN_TILES", 96);
    v.f128 = env_float("VDN_GEMM_F128", 1.3f);   // lock-step 128-row kernel (same loop, fewer rows per W slab): keeps its order behind the 192-row one
    v.f192 = env_float("VDN_GEMM_F192", 1.2f);  // lock-step 192-row kernel: 1.65 us per K step against 1.76 for 256 rows = 1.25x per row
    v.x8 = env_int("VDN_GEMM_X8", 1);
    return v;
  }();
  return t;
}
const vdn_gemm_tuning& tuning(const vdn_gemm_desc& d) { return d.tuning ? *d.tuning : tuning_defaults(); }
}

extern "C" int vdn_gemm_get_tuning(vdn_gemm_tuning* out) {
  if (!out) return VDN_EINVAL;
  vdn_gemm_desc none = {};
  *out = vdn_gemm_impl::tuning(none);
  return VDN_OK;
}

extern "C" int vdn_gemm(const vdn_gemm_desc* dp, vdn_stream stream) {
  if (!dp) return VDN_EINVAL;
  const vdn_gemm_desc& d = *dp;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.W || !d.zeros) return VDN_EINVAL;
  if (d.tuning) {
    const vdn_gemm_tuning& t = *d.tuning;
    if ((t.force_bm != 0 && t.force_bm != 128 && t.force_bm != 192 && t.force_bm != 256) || t.splitk_occ < 0 || t.splitk_max < 0 ||
        t.min_tiles < 0 || t.cus < 0 || t.cus > 256)
      return VDN_EINVAL;
  }
  if (d.dt != VDN_F16 && d.dt != VDN_BF16) return VDN_EUNSUPPORTED;
  if ((d.K & 7) || (d.ldb & 63) || d.ldb < d.K) return VDN_EALIGN;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15) || ((uintptr_t)d.zeros & 15)) return VDN_EALIGN;
  if (d.a_mode == VDN_A_CONV3X3) {
    if ((d.cC & 7) || d.K != 9 * d.cC || d.M != d.cB * d.cOH * d.cOW) return VDN_EINVAL;
    if (d.conv_korder && ((d.cC & 63) || d.conv_korder != 1)) return VDN_EINVAL;
    if (d.cstride != 1 && d.cstride != 2) return VDN_EUNSUPPORTED;
    if (d.cOH != (d.cH + 2 - 3) / d.cstride + 1 || d.cOW != (d.cW + 2 - 3) / d.cstride + 1) return VDN_EINVAL;
  } else if (d.a_mode == VDN_A_PLAIN) {
    if ((d.lda & 7) || d.lda < d.K) return VDN_EALIGN;
  } else {
    return VDN_EUNSUPPORTED;
  }
  if (d.N & 3) return VDN_EALIGN;  // the epilogue stores 4 columns per lane
  if (d.ksplit != 0 || d.splitk_ws_bytes < 0 || (d.splitk_ws == nullptr) != (d.splitk_ws_bytes == 0)) return VDN_EINVAL;
  if ((uintptr_t)d.splitk_ws & 15) return VDN_EALIGN;
  if (((uintptr_t)d.A_lo | (uintptr_t)d.W_lo | (uintptr_t)d.out_lo) & 15) return VDN_EALIGN;
  if (d.out_lo && (d.out_dt == VDN_F32 || !d.out)) return VDN_EINVAL;
  if ((d.res1 && (d.ldr1 & 3)) || (d.res2 && (d.ldr2 & 3))) return VDN_EALIGN;
  if (((uintptr_t)d.A8 | (uintptr_t)d.W8 | (uintptr_t)d.out8) & 15) return VDN_EALIGN;
  if ((d.A8 || d.W8) && (d.dt != VDN_F16 || d.a_mode != VDN_A_PLAIN || (d.K & 63) || (d.lda != d.K && !d.a_kt))) return VDN_EINVAL;
  if (d.out8 && (d.dt != VDN_F16 || d.store != VDN_ST_PLAIN || d.out_dt != VDN_F16 || d.act != VDN_ACT_GELU || d.N != d.ldc || (d.N & 63) || !d.A8 || !d.W8))
    return VDN_EINVAL;
  if ((d.a_kt || d.w_kt || d.out_kt || d.x8_terms) && (!d.A8 || !d.W8)) return VDN_EINVAL;
  // the 8-bit kernel addresses a lane's rows with 32-bit byte offsets inside a plane (row-major planes: rows * ld elements)
  if (d.A8 && d.W8 && ((!d.a_kt && (long)d.M * d.lda >= (1L << 31)) || (!d.w_kt && (long)d.N * d.ldb >= (1L << 31)) || d.M >= (1 << 25) || d.N >= (1 << 25)))
    return VDN_EUNSUPPORTED;
  if (d.x8_terms < 0 || d.x8_terms > 2) return VDN_EINVAL;  // K-tile-major planes: the 8-bit cross-term kernel only
  if (d.out_kt && (d.store != VDN_ST_PLAIN || d.out_dt != VDN_F16 || d.ldc != d.N || (d.N & 63) || d.res1 || d.res2 || d.tab ||
                   d.rowadd || d.gamma || d.row_group > 0 || d.act == VDN_ACT_RELU)) return VDN_EINVAL;
  if (((uintptr_t)d.bias | (uintptr_t)d.gamma | (uintptr_t)d.tab | (uintptr_t)d.res1 | (uintptr_t)d.res2) & 7) return VDN_EALIGN;
  switch (d.store) {
    case VDN_ST_PLAIN:
      if (!d.out || d.ldc < d.N) return VDN_EINVAL;
      if ((d.ldc & 3) || ((uintptr_t)d.out & 15)) return VDN_EALIGN;
      break;
    case VDN_ST_GEGLU:
      if (!d.out || (d.N & 31) || d.ldc < d.N / 2 || (d.act && d.act != VDN_ACT_SILU) || d.gamma || d.res1 || d.res2 || d.tab || d.rowadd)
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
        if (d.dst8[i] && (d.transposed[i] || (!d.W_lo && !d.W8) || d.dt != VDN_F16 || ((uintptr_t)d.dst8[i] & 15))) return VDN_EINVAL;
      }
      break;
    default:
      return VDN_EUNSUPPORTED;
  }
  hipStream_t s = (hipStream_t)stream;
  return d.dt == VDN_F16 ? vdn_gemm_impl::launch_f16(d, s) : vdn_gemm_impl::launch_bf16(d, s);
}
