// wave_inst.hip -- one instantiation of the wave column or row kernels.
// Compile with -DXCX_INST_COL=1|0 -DXCX_INST_L=<len> -DXCX_INST_R=<r0,r1,..>
#include "wave_launch.h"

namespace xcx {

using RX = Radices<XCX_INST_R>;
static_assert(RX::L == XCX_INST_L, "radices do not multiply to the length");

#define XCX_CAT2(a, b) a##b
#define XCX_CAT(a, b) XCX_CAT2(a, b)

template <class K> static hipError_t ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 32768) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#if XCX_INST_COL
constexpr size_t kLds = sizeof(float) * lds_col_floats(RX::L) * kWPB;
hipError_t XCX_CAT(launch_wfwd_cols_, XCX_INST_L)(const WF2Args& a, dim3 grid, hipStream_t s) {
  grid.x = (grid.x + kWPB - 1) / kWPB;
  if (a.nt) hipLaunchKernelGGL((k_wfwd_cols<RX, true>), grid, dim3(kWNT * kWPB), kLds, s, a);
  else hipLaunchKernelGGL((k_wfwd_cols<RX, false>), grid, dim3(kWNT * kWPB), kLds, s, a);
  return hipGetLastError();
}
hipError_t XCX_CAT(launch_wpair_cols_, XCX_INST_L)(const WI1Args& a, dim3 grid, hipStream_t s) {
  grid.x = 8u * ((grid.x / 8u + kWPB - 1) / kWPB);     // waves per XCD -> workgroups per XCD
  if (a.nstack < 0)
    hipLaunchKernelGGL((k_wpair_cols<RX, 2>), grid, dim3(kWNT * kWPB), kLds, s, a);
  else if (a.nstack > 1)
    hipLaunchKernelGGL((k_wpair_cols<RX, 1>), grid, dim3(kWNT * kWPB), kLds, s, a);
  else
    hipLaunchKernelGGL((k_wpair_cols<RX, 0>), grid, dim3(kWNT * kWPB), kLds, s, a);
  return hipGetLastError();
}
#else
constexpr size_t kLdsF = sizeof(float) * lds_frow_floats(RX::L) * kF1WPB;   // forward rows keep full complex LDS
constexpr size_t kLds = sizeof(float) * lds_row_floats(RX::L) * kWPB;
hipError_t XCX_CAT(launch_wfwd_rows_, XCX_INST_L)(const WF1Args& a, dim3 grid, hipStream_t s) {
  grid.x = (grid.x + kF1WPB - 1) / kF1WPB;
  static hipError_t once = ensure_lds((k_wfwd_rows<RX, false, false>), kLdsF);
  static hipError_t once1 = ensure_lds((k_wfwd_rows<RX, true, false>), kLdsF);
  static hipError_t once2 = ensure_lds((k_wfwd_rows<RX, false, true>), kLdsF);
  static hipError_t once3 = ensure_lds((k_wfwd_rows<RX, true, true>), kLdsF);
  for (hipError_t e : {once, once1, once2, once3})
    if (e != hipSuccess) return e;
  const bool ob = a.g.flags & XCX_F_ONEBIT;
  if (ob && a.nt) hipLaunchKernelGGL((k_wfwd_rows<RX, true, true>), grid, dim3(kWNT * kF1WPB), kLdsF, s, a);
  else if (ob) hipLaunchKernelGGL((k_wfwd_rows<RX, true, false>), grid, dim3(kWNT * kF1WPB), kLdsF, s, a);
  else if (a.nt) hipLaunchKernelGGL((k_wfwd_rows<RX, false, true>), grid, dim3(kWNT * kF1WPB), kLdsF, s, a);
  else hipLaunchKernelGGL((k_wfwd_rows<RX, false, false>), grid, dim3(kWNT * kF1WPB), kLdsF, s, a);
  return hipGetLastError();
}
hipError_t XCX_CAT(launch_wpair_rows_, XCX_INST_L)(const WI2Args& a, dim3 grid, hipStream_t s) {
  grid.x = (grid.x + kWPB - 1) / kWPB;
  static hipError_t once = ensure_lds(k_wpair_rows<RX>, kLds);
  if (once != hipSuccess) return once;
  hipLaunchKernelGGL(k_wpair_rows<RX>, grid, dim3(kWNT * kWPB), kLds, s, a);
  return hipGetLastError();
}
#endif

}  // namespace xcx
