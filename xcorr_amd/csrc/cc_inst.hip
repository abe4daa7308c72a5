// cc_inst.hip -- one instantiation of the column or row kernels.
// Compile with -DXCX_INST_COL=1|0 -DXCX_INST_L=<len> -DXCX_INST_R=<r0,r1,..>
#include "cc_launch.h"

namespace xcx {

using RX = Radices<XCX_INST_R>;
static_assert(RX::L == XCX_INST_L, "radices do not multiply to the length");

#define XCX_CAT2(a, b) a##b
#define XCX_CAT(a, b) XCX_CAT2(a, b)

template <class K> static hipError_t ensure_lds(K kernel, size_t bytes) {
  // static LDS limit is 64 KiB; larger tiles need the explicit opt-in
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

#if XCX_INST_COL
hipError_t XCX_CAT(launch_fwd_cols_, XCX_INST_L)(const F2Args& a, dim3 grid,
                                                 hipStream_t s) {
  constexpr size_t lds = (RX::S > 1) ? sizeof(cf) * RX::L * kCB : 0;
  static hipError_t once = ensure_lds(k_fwd_cols<RX>, lds);
  if (once != hipSuccess) return once;
  hipLaunchKernelGGL(k_fwd_cols<RX>, grid, dim3(kNT), lds, s, a);
  return hipGetLastError();
}
hipError_t XCX_CAT(launch_pair_cols_, XCX_INST_L)(const I1Args& a, dim3 grid,
                                                  hipStream_t s) {
  constexpr size_t lds = (RX::S > 1) ? sizeof(cf) * RX::L * kCB : 0;
  static hipError_t once = ensure_lds(k_pair_cols<RX>, lds);
  if (once != hipSuccess) return once;
  hipLaunchKernelGGL(k_pair_cols<RX>, grid, dim3(kNT), lds, s, a);
  return hipGetLastError();
}
#else
hipError_t XCX_CAT(launch_fwd_rows_, XCX_INST_L)(const F1Args& a, dim3 grid,
                                                 hipStream_t s) {
  constexpr size_t lds = sizeof(cf) * RX::L * kRBP;
  static hipError_t once = ensure_lds(k_fwd_rows<RX>, lds);
  if (once != hipSuccess) return once;
  hipLaunchKernelGGL(k_fwd_rows<RX>, grid, dim3(kNT), lds, s, a);
  return hipGetLastError();
}
hipError_t XCX_CAT(launch_pair_rows_, XCX_INST_L)(const I2Args& a, dim3 grid,
                                                  hipStream_t s) {
  constexpr size_t lds = sizeof(cf) * RX::L * kRBP;
  static hipError_t once = ensure_lds(k_pair_rows<RX>, lds);
  if (once != hipSuccess) return once;
  hipLaunchKernelGGL(k_pair_rows<RX>, grid, dim3(kNT), lds, s, a);
  return hipGetLastError();
}
#endif

}  // namespace xcx
