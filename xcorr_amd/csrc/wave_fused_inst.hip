// wave_fused_inst.hip -- one (N1, N2h) instantiation of the dual and flow pair kernels.
// Compile with -DXCX_FC_L=<N1> -DXCX_FC_R=<radices> -DXCX_FR_L=<N2h> -DXCX_FR_R=<radices>
#include <algorithm>
#include <cstdlib>
#include "wave_launch.h"

namespace xcx {

using RXC = Radices<XCX_FC_R>;
using RXR = Radices<XCX_FR_R>;
static_assert(RXC::L == XCX_FC_L && RXR::L == XCX_FR_L, "radices do not multiply to the length");

#define XCX_CAT3_(a, b, c, d) a##b##c##d
#define XCX_CAT3(a, b, c, d) XCX_CAT3_(a, b, c, d)

hipError_t XCX_CAT3(launch_wdual_, XCX_FC_L, _, XCX_FR_L)(const WDualArgs& a, hipStream_t s) {
  using PG = PairGeom<RXC, RXR>;
  constexpr size_t lds = sizeof(float) * PG::kLdsFloats;
  const int Bm = a.Bc > a.Br ? a.Bc : a.Br;
  const dim3 grid(8u * (unsigned)((PG::CWX + PG::RWX) * Bm));
  // occupancy knob: pad the LDS request so that at most `wgs` workgroups share a CU
  static const int wgs = getenv("XCX_DUAL_WGS") ? atoi(getenv("XCX_DUAL_WGS")) : 0;
  size_t lds_req = lds;
  if (wgs > 0) {
    lds_req = std::max(lds, (size_t)(160 * 1024 / wgs) & ~(size_t)255);
    static hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wpair_dual<RXC, RXR, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    static hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(k_wpair_dual<RXC, RXR, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess) return e1;
    if (e2 != hipSuccess) return e2;
  }
  if (a.nstack < 0)
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 2>), grid, dim3(kWNT * kWPB), lds_req, s, a);
  else if (a.nstack > 1)
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 1>), grid, dim3(kWNT * kWPB), lds_req, s, a);
  else
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 0>), grid, dim3(kWNT * kWPB), lds_req, s, a);
  return hipGetLastError();
}

hipError_t XCX_CAT3(launch_wflow_, XCX_FC_L, _, XCX_FR_L)(const WFlowArgs& a, hipStream_t s) {
  using PG = PairGeom<RXC, RXR>;
  constexpr size_t lds = sizeof(float) * PG::kLdsFloats;
  const int nbatch = (a.count + a.B - 1) / a.B;
  const long wpg = (long)(PG::CWX + PG::RWX) * a.B;
  const long nwg = 8L * wpg * (nbatch + a.lag);               // trailing groups: rows of the last batches
  if (nwg > 0x7fffffffL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)nwg);
  if (a.nstack < 0)
    hipLaunchKernelGGL((k_wpair_flow<RXC, RXR, 2>), grid, dim3(kWNT * kWPB), lds, s, a);
  else if (a.nstack > 1)
    hipLaunchKernelGGL((k_wpair_flow<RXC, RXR, 1>), grid, dim3(kWNT * kWPB), lds, s, a);
  else
    hipLaunchKernelGGL((k_wpair_flow<RXC, RXR, 0>), grid, dim3(kWNT * kWPB), lds, s, a);
  return hipGetLastError();
}

}  // namespace xcx
