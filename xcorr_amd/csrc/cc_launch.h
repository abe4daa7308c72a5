// cc_launch.h -- host-side launch table of the templated FFT kernels.
// Each supported sub-FFT length is compiled in its own translation unit
// (cc_inst.hip with -DXCX_INST_*), see Makefile.
#pragma once
#include <hip/hip_runtime.h>
#include "cc_kernels.h"

namespace xcx {

// supported column lengths N1 and row lengths N2h (radices in the Makefile)
#define XCX_COL_LIST(X) X(16) X(32) X(64) X(128) X(256) X(384) X(512)
#define XCX_ROW_LIST(X) X(16) X(32) X(64) X(128) X(240) X(256) X(480) X(512)

#define XCX_DECL_COL(L)                                                     \
  hipError_t launch_fwd_cols_##L(const F2Args& a, dim3 grid, hipStream_t s); \
  hipError_t launch_pair_cols_##L(const I1Args& a, dim3 grid, hipStream_t s);
#define XCX_DECL_ROW(L)                                                     \
  hipError_t launch_fwd_rows_##L(const F1Args& a, dim3 grid, hipStream_t s); \
  hipError_t launch_pair_rows_##L(const I2Args& a, dim3 grid, hipStream_t s);
XCX_COL_LIST(XCX_DECL_COL)
XCX_ROW_LIST(XCX_DECL_ROW)

struct ColEntry {
  int L;
  hipError_t (*fwd)(const F2Args&, dim3, hipStream_t);
  hipError_t (*pair)(const I1Args&, dim3, hipStream_t);
};
struct RowEntry {
  int L;
  hipError_t (*fwd)(const F1Args&, dim3, hipStream_t);
  hipError_t (*pair)(const I2Args&, dim3, hipStream_t);
};

}  // namespace xcx
