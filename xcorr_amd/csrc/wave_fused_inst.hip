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

// q24 launch with the zero-stretch skip a.zb0 (one of ZSkipList<N1>::v)
template <int I> static bool launch_zskip(const WDualArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  using ZL = ZSkipList<RXC::L>;
  if constexpr (I < ZL::n) {
    if (a.zb0 == ZL::v[I]) {
      hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 0, 1, ZL::v[I]>), grid, dim3(kWNT * kWPB), lds, s, a);
      return true;
    }
    return launch_zskip<I + 1>(a, grid, lds, s);
  } else {
    return false;
  }
}

hipError_t XCX_CAT3(launch_wdual_, XCX_FC_L, _, XCX_FR_L)(const WDualArgs& a, hipStream_t s) {
  using PG = PairGeom<RXC, RXR>;
  constexpr size_t lds = sizeof(float) * PG::kLdsFloats;
  const int Bm = a.Bc > a.Br ? a.Bc : a.Br;
  const dim3 grid(8u * (unsigned)((PG::CWX + PG::RWX) * Bm));
  if (a.ts == 1) {
    // q24 storage of T: plain pair products of whitened spectra only (the bound behind the fixed-point scale)
    if constexpr (DualQ24<RXC, RXR>::ok) {
      if (a.nstack != 1) return hipErrorInvalidValue;
      if (a.zb0 == 0) hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 0, 1>), grid, dim3(kWNT * kWPB), lds, s, a);
      else if (!launch_zskip<0>(a, grid, lds, s)) return hipErrorInvalidValue;
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (a.nstack < 0)
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 2>), grid, dim3(kWNT * kWPB), lds, s, a);
  else if (a.nstack > 1)
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 1>), grid, dim3(kWNT * kWPB), lds, s, a);
  else
    hipLaunchKernelGGL((k_wpair_dual<RXC, RXR, 0>), grid, dim3(kWNT * kWPB), lds, s, a);
  return hipGetLastError();
}

int XCX_CAT3(wdual_q24_, XCX_FC_L, _, XCX_FR_L)() { return DualQ24<RXC, RXR>::ok ? 1 : 0; }
int XCX_CAT3(wdual_zskip_, XCX_FC_L, _, XCX_FR_L)(const Geom& g) { return DualQ24<RXC, RXR>::ok ? zskip_pick<RXC::L>(g) : 0; }

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
