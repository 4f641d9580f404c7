// 8-wave BM x 256 split-precision GEMM kernels for f16 operands.
#include "gemm_kernels.hpp"
namespace vdn_gemm_impl {
template <> int big_entry<VDN_F16>(const vdn_gemm_desc& d, int bm, hipStream_t s) {
  if (bm == 256) return launch_x3_big<VDN_F16, 256>(d, s);
  if (bm == 192) return launch_x3_big<VDN_F16, 192>(d, s);
  return launch_x3_big<VDN_F16, 128>(d, s);
}
template <> int splitk_entry<VDN_F16>(const vdn_gemm_desc& d, int ksplit, int fl, hipStream_t s) {
  return launch_splitk<VDN_F16>(d, ksplit, fl, s);
}
}
