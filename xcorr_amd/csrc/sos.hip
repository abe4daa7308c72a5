// sos.hip -- Butterworth second-order-section filters (placeholder body is
// replaced below in this round).
#include "../../include/xcx.h"
#include "xcx_internal.h"
using namespace xcx;
extern "C" int xcx_sosfiltfilt(xcx_ctx*, float*, long, long, const double*, int, int) {
  return set_err(XCX_EUNSUPPORTED, "xcx_sosfiltfilt: not built yet");
}
extern "C" int xcx_sosfilt(xcx_ctx*, float*, long, long, const double*, int, int) {
  return set_err(XCX_EUNSUPPORTED, "xcx_sosfilt: not built yet");
}
