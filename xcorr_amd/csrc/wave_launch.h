// wave_launch.h -- launch table of the one-wave-per-workgroup kernels
#pragma once
#include <hip/hip_runtime.h>
#include "wave_kernels.h"

namespace xcx {

#define XCX_WCOL_LIST(X) X(64) X(128) X(256) X(512) X(768) X(1024) X(1536) X(2048) X(3072)
#define XCX_WROW_LIST(X) X(16) X(32) X(64) X(120) X(128)

#define XCX_DECL_WCOL(L)                                                       \
  hipError_t launch_wfwd_cols_##L(const WF2Args& a, dim3 grid, hipStream_t s); \
  hipError_t launch_wpair_cols_##L(const WI1Args& a, dim3 grid, hipStream_t s);
#define XCX_DECL_WROW(L)                                                       \
  hipError_t launch_wfwd_rows_##L(const WF1Args& a, dim3 grid, hipStream_t s); \
  hipError_t launch_wpair_rows_##L(const WI2Args& a, dim3 grid, hipStream_t s);
XCX_WCOL_LIST(XCX_DECL_WCOL)
XCX_WROW_LIST(XCX_DECL_WROW)

// fused XCD-local pair kernel: instantiated (N1, N2h) combinations
#define XCX_WFUSED_LIST(X) X(64, 16) X(256, 16) X(512, 16) X(1536, 16) X(1536, 120) X(2048, 120) X(1024, 120) X(1536, 64) X(768, 120) X(512, 120) X(1536, 128) X(2048, 64) X(2048, 128) X(3072, 120)
#define XCX_DECL_WFUSED(C, R) \
  hipError_t launch_wdual_##C##_##R(const WDualArgs& a, hipStream_t s);              \
  int wdual_q24_##C##_##R();                                                         \
  int wdual_zskip_##C##_##R(const Geom& g);                                          \
  hipError_t launch_wflow_##C##_##R(const WFlowArgs& a, hipStream_t s);
XCX_WFUSED_LIST(XCX_DECL_WFUSED)
struct WFusedEntry {
  int N1, N2h;
  hipError_t (*dual)(const WDualArgs&, hipStream_t);
  hipError_t (*flow)(const WFlowArgs&, hipStream_t);
  int (*q24)();      // 1: the dual kernel has a q24 (24-bit fixed-point T) instantiation
  int (*zskip)(const Geom&);   // instantiated zero-stretch skip (ZB0 of the column body) for a plan's band, 0: none
};

struct WColEntry {
  int L;
  hipError_t (*fwd)(const WF2Args&, dim3, hipStream_t);
  hipError_t (*pair)(const WI1Args&, dim3, hipStream_t);
};
struct WRowEntry {
  int L;
  hipError_t (*fwd)(const WF1Args&, dim3, hipStream_t);
  hipError_t (*pair)(const WI2Args&, dim3, hipStream_t);
};

}  // namespace xcx
