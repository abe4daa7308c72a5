// xcx_api.hip -- C ABI (include/xcx.h): context, planner, and the
// pair x window cross-correlation driver.  No exceptions cross the boundary.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/xcx.h"
#include "cc_kernels.h"
#include "wave_launch.h"
#include "pair_order.h"
#include "xcx_internal.h"
#include "xcx_plan.h"

using namespace xcx;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
namespace xcx {
static thread_local std::string g_err;
int set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace xcx


// ---------------------------------------------------------------------------
// launch tables
// ---------------------------------------------------------------------------
#define XCX_WCOL_ROW(L) {L, launch_wfwd_cols_##L, launch_wpair_cols_##L},
#define XCX_WROW_ROW(L) {L, launch_wfwd_rows_##L, launch_wpair_rows_##L},
static const WColEntry kWCols[] = {XCX_WCOL_LIST(XCX_WCOL_ROW)};
static const WRowEntry kWRows[] = {XCX_WROW_LIST(XCX_WROW_ROW)};
static const int kNWCols = sizeof(kWCols) / sizeof(kWCols[0]);
static const int kNWRows = sizeof(kWRows) / sizeof(kWRows[0]);
#define XCX_WFUSED_ROW(C, R) {C, R, launch_wdual_##C##_##R, launch_wflow_##C##_##R, wdual_q24_##C##_##R, wdual_zskip_##C##_##R},
static const WFusedEntry kWFused[] = {XCX_WFUSED_LIST(XCX_WFUSED_ROW)};
static const int kNWFused = sizeof(kWFused) / sizeof(kWFused[0]);
// default T storage of whitened plans: q24 after the error table of profiles/r05_t24_error_table.txt
#ifndef XCX_Q24_DEFAULT
#define XCX_Q24_DEFAULT 1
#endif
static const bool kQ24Default = XCX_Q24_DEFAULT != 0;

// ---------------------------------------------------------------------------
// k_wstats: partial sums of a station-window, grid = (kStatParts, cnt).
// PHASE 0: sum(x) -> psum (only with DEMEAN); PHASE 1: sum(x'^2) -> psq with
// x' the demeaned / one-bit sample.  Consumers add the kStatParts partials in
// fixed order (wave_kernels.h sw_mean / sw_scale): deterministic.
// ---------------------------------------------------------------------------
template <int PHASE>
__global__ __launch_bounds__(kNT) void k_wstats(
    const float* __restrict__ traces, long stride_s, long stride_w, int W,
    int w0, int sw0, int n, int flags, const double* __restrict__ psum,
    double* __restrict__ dst) {
  __shared__ double red[kNT];
  const int swl = blockIdx.y;
  const int sw = sw0 + swl;
  const int s = sw / W, wl = sw % W;
  const float* x = traces + (long)s * stride_s + (long)(w0 + wl) * stride_w;
  const int tid = threadIdx.x;
  const int part = blockIdx.x;
  const int per = (n + kStatParts - 1) / kStatParts;
  const int lo = part * per, hi = min(n, lo + per);
  double acc = 0.0;
  if (PHASE == 0) {
    for (int i = lo + tid; i < hi; i += kNT) acc += (double)x[i];
  } else {
    const double mean_d = sw_mean_d(psum, swl, n, flags & XCX_F_DEMEAN);
    const float mean = (float)mean_d;
    for (int i = lo + tid; i < hi; i += kNT) {
      const float v = (flags & XCX_F_ONEBIT) ? onebit_of(x[i], mean_d) : x[i] - mean;
      acc += (double)v * (double)v;
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = kNT / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) dst[swl * kStatParts + part] = red[0];
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
static int flow_check(xcx_ctx* ctx);

extern "C" int xcx_device_count(int* count) {
  if (!count) return set_err(XCX_EINVAL, "count is null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *count = 0; return set_err(XCX_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *count = c;
  return XCX_OK;
}

extern "C" int xcx_init(int device, xcx_ctx** out) {
  if (!out) return set_err(XCX_EINVAL, "ctx out pointer is null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c == 0)
    return set_err(XCX_ENODEV, "no HIP device available (%s)",
                   e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (device < 0 || device >= c)
    return set_err(XCX_EINVAL, "device %d out of range [0,%d)", device, c);
  HIP_TRY(hipSetDevice(device));
  xcx_ctx* ctx = new (std::nothrow) xcx_ctx();
  if (!ctx) return set_err(XCX_ENOMEM, "host allocation failed");
  ctx->device = device;
  // streams and events: a failure part-way releases what exists (xcx_destroy copes with null members)
  hipError_t he = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  for (int i = 0; i < 2 && he == hipSuccess; ++i)
    he = hipStreamCreateWithFlags(&ctx->pipe[i], hipStreamNonBlocking);   // see xcx_internal.h
  if (he == hipSuccess) he = hipEventCreate(&ctx->ev0);
  if (he == hipSuccess) he = hipEventCreate(&ctx->ev1);
  if (he != hipSuccess) {
    (void)hipGetLastError();
    xcx_destroy(ctx);
    return set_err(XCX_EHIP, "xcx_init: %s", hipGetErrorString(he));
  }
  *out = ctx;
  return XCX_OK;
}

extern "C" int xcx_destroy(xcx_ctx* ctx) {
  if (!ctx) return XCX_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  xcx_comm_destroy(ctx);
  if (ctx->ts_plan) xcx_plan_destroy(ctx->ts_plan);
  for (auto* bp : ctx->bs_plan) if (bp) xcx_plan_destroy(bp);
  for (auto* bp : ctx->bs_seg_plan) if (bp) xcx_plan_destroy(bp);
  ctx->sg_tab.release();
  for (auto& t : ctx->sos_tab) t.release();
  ctx->bs_tab.release();
  for (auto& b : ctx->scratch) b.release();
  if (ctx->flow_T) (void)hipFree(ctx->flow_T);
  if (ctx->flow_err_dev) (void)hipFree(ctx->flow_err_dev);
  if (ctx->flow_err_host) (void)hipHostFree(ctx->flow_err_host);
  for (auto& ps : ctx->pipe)
    if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); }
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return XCX_OK;
}

// Give back what the context keeps between calls: the grow-only workspaces of the row ops / filters / transforms, the
// cached tables, and the cached plans of xcx_timeshift and of the any-length transforms (a day-long rfft, time shift or
// segmented correlation leaves tens of GB in them).  Plans the CALLER created are the caller's to destroy.
extern "C" int xcx_trim(xcx_ctx* ctx) {
  if (!ctx) return set_err(XCX_EINVAL, "null ctx");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (auto& ps : ctx->pipe) if (ps) HIP_TRY(hipStreamSynchronize(ps));
  if (ctx->ts_plan) { xcx_plan_destroy(ctx->ts_plan); ctx->ts_plan = nullptr; ctx->ts_n = 0; }
  for (int i = 0; i < 2; ++i) {
    if (ctx->bs_plan[i]) { xcx_plan_destroy(ctx->bs_plan[i]); ctx->bs_plan[i] = nullptr; }
    ctx->bs_key[i][0] = ctx->bs_key[i][1] = ctx->bs_key[i][2] = 0;
    if (ctx->bs_seg_plan[i]) { xcx_plan_destroy(ctx->bs_seg_plan[i]); ctx->bs_seg_plan[i] = nullptr; }
    ctx->bs_seg_n[i] = 0;
  }
  ctx->sg_tab.release(); ctx->sg_key = 0;
  for (int i = 0; i < xcx_ctx::kSosTabs; ++i) { ctx->sos_tab[i].release(); ctx->sos_key[i] = 0; }
  ctx->bs_tab.release(); ctx->bs_tab_n = 0;
  for (auto& b : ctx->scratch) b.release();
  return XCX_OK;
}

extern "C" int xcx_last_error(char* buf, size_t len) {
  if (!buf || len == 0) return XCX_EINVAL;
  snprintf(buf, len, "%s", g_err.c_str());
  return XCX_OK;
}

extern "C" int xcx_device_name(xcx_ctx* ctx, char* buf, size_t len) {
  if (!ctx || !buf) return set_err(XCX_EINVAL, "null argument");
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, ctx->device));
  snprintf(buf, len, "%s %s cu=%d", p.name, p.gcnArchName, p.multiProcessorCount);
  return XCX_OK;
}

extern "C" int xcx_malloc(xcx_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(ctx_malloc(ctx, dptr, bytes ? bytes : 1));
  return XCX_OK;
}
// every numeric limit include/xcx.h quotes, as this build enforces it
static_assert(kMaxTransformN == XCX_MAX_SINGLE_TRANSFORM, "include/xcx.h: XCX_MAX_SINGLE_TRANSFORM");
extern "C" int xcx_limits(long* out, int count) {
  if (!out || count < 1) return set_err(XCX_EINVAL, "null argument");
  const long v[XCX_LIM_COUNT] = {
      XCX_MAX_SINGLE_TRANSFORM, XCX_MAX_CHIRP_POINTS, XCX_MAX_ROW_SAMPLES, XCX_MAX_SPLIT_PARTS, XCX_CC2D_MAX_F, XCX_CC2D_MAX_L,
      XCX_RAM_MAX_HALF_WIDTH, XCX_SOS_MAX_SECTIONS, XCX_SPECTROGRAM_MAX_WIN, XCX_SPECTROGRAM_MAX_NFFT,
      XCX_SPECTROGRAM_SEG_MAX_NFFT, XCX_ROWS_MAX_BATCH, XCX_SEG_WORKSPACE_MAX_GB};
  for (int i = 0; i < count && i < XCX_LIM_COUNT; ++i) out[i] = v[i];
  return XCX_OK;
}
extern "C" int xcx_set_reclaim(xcx_ctx* ctx, size_t (*cb)(void*), void* user) {
  if (!ctx) return set_err(XCX_EINVAL, "null ctx");
  ctx->reclaim = cb;
  ctx->reclaim_user = user;
  return XCX_OK;
}
extern "C" int xcx_free(xcx_ctx* ctx, void* dptr) {
  if (!ctx) return set_err(XCX_EINVAL, "null ctx");
  if (!dptr) return XCX_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipFree(dptr));
  return XCX_OK;
}
// page-locked host memory for result arrays the library's callers hand back to Python: a D2H into
// it runs at PCIe speed (pageable destinations go through the runtime's staging at ~9 GB/s)
extern "C" int xcx_host_alloc(xcx_ctx* ctx, size_t bytes, void** hptr) {
  if (!ctx || !hptr) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return XCX_OK;
}
extern "C" int xcx_host_free(xcx_ctx* ctx, void* hptr) {
  (void)ctx;                       // may be null: a result array can outlive its context, hipHostFree needs none
  if (!hptr) return XCX_OK;
  HIP_TRY(hipHostFree(hptr));
  return XCX_OK;
}
extern "C" int xcx_memset(xcx_ctx* ctx, void* dptr, int value, size_t bytes) {
  if (!ctx || !dptr) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return XCX_OK;
}
extern "C" int xcx_memcpy_h2d(xcx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return XCX_OK;
}
extern "C" int xcx_memcpy_d2h(xcx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return flow_check(ctx);
}
extern "C" int xcx_memcpy_d2d(xcx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return XCX_OK;
}
extern "C" int xcx_memcpy2d_d2h(xcx_ctx* ctx, void* dst, size_t dpitch, const void* src,
                                size_t spitch, size_t width_bytes, size_t rows) {
  if (!ctx || !dst || !src) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows,
                           hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return flow_check(ctx);
}
// after a synchronisation of the context's stream: did a flow launch give up on a dependency?
static int flow_check(xcx_ctx* ctx) {
  if (!ctx->flow_ran || !ctx->flow_err_host) return XCX_OK;
  if (hipStreamQuery(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return XCX_OK; }   // still running: checked later
  ctx->flow_ran = false;
  if (*ctx->flow_err_host) {
    *ctx->flow_err_host = 0;
    (void)hipMemsetAsync(ctx->flow_err_dev, 0, sizeof(int), ctx->stream);
    return set_err(XCX_EHIP, "flow pair kernel: a batch dependency wait timed out; the CCFs of the last call are invalid "
                             "(set XCX_FLOW=0 to use the per-batch launches)");
  }
  return XCX_OK;
}

extern "C" int xcx_sync(xcx_ctx* ctx) {
  if (!ctx) return set_err(XCX_EINVAL, "null ctx");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return flow_check(ctx);
}
extern "C" int xcx_timer_start(xcx_ctx* ctx) {
  if (!ctx) return set_err(XCX_EINVAL, "null ctx");
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  return XCX_OK;
}
extern "C" int xcx_timer_stop(xcx_ctx* ctx, float* ms) {
  if (!ctx || !ms) return set_err(XCX_EINVAL, "null argument");
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return flow_check(ctx);
}

// ---------------------------------------------------------------------------
// workspace helper
// ---------------------------------------------------------------------------
namespace xcx {
size_t ctx_reclaim(xcx_ctx* ctx) {
  if (!ctx || !ctx->reclaim) return 0;
  auto cb = ctx->reclaim;
  ctx->reclaim = nullptr;                       // not re-entered from the xcx_free calls it makes
  const size_t got = cb(ctx->reclaim_user);
  ctx->reclaim = cb;
  return got;
}
hipError_t ctx_malloc(xcx_ctx* ctx, void** ptr, size_t nbytes) {
  hipError_t e = hipMalloc(ptr, nbytes);
  if (e == hipErrorOutOfMemory && ctx_reclaim(ctx) > 0) {
    (void)hipGetLastError();
    e = hipMalloc(ptr, nbytes);
  }
  return e;
}
int DevBuf::ensure(xcx_ctx* ctx, size_t need) {
  if (need <= bytes) return XCX_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ptr) HIP_TRY(hipFree(ptr));
  ptr = nullptr;
  bytes = 0;
  HIP_TRY(ctx_malloc(ctx, &ptr, need));
  bytes = need;
  // process-wide allocation number: (ptr, gen) identifies THIS memory even when the allocator hands the same
  // address to another buffer later (a per-buffer count could repeat there and revive a stale "zeroed" record)
  static std::atomic<unsigned long long> next_gen{0};
  gen = ++next_gen;
  return XCX_OK;
}
void DevBuf::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}
}  // namespace xcx

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
static const char* kKernelNames[KID_COUNT] = {"k_stats", "k_fwd_rows", "k_fwd_cols",
                                              "k_pair_cols", "k_pair_rows", "k_pair_dual", "k_gram",
                                              "pair_phase"};

static void fill_table(std::vector<float2>& v, int count, int L, int mul = 1) {
  v.resize(count);
  for (int q = 0; q < count; ++q) {
    // exact reduction of (q*mul) mod L before the trig call
    long long a = ((long long)q * mul) % L;
    double th = -2.0 * M_PI * (double)a / (double)L;
    v[q] = make_float2((float)cos(th), (float)sin(th));
  }
}

static int upload(xcx_ctx* ctx, DevBuf& b, const void* src, size_t bytes) {
  int rc = b.ensure(ctx, bytes);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(b.ptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return XCX_OK;
}

// The planner's choice of N = 2 N1 N2h: the smallest instantiated product >= need (or == nfft); among equal N the
// longest column up to 2048 (one column per wave; 3072 runs at two waves per SIMD).  Returns N or -1.
static long long plan_choose(long long need, int nfft, int& bi, int& bj) {
  long long best = -1;
  bi = bj = -1;
  for (int i = 0; i < kNWCols; ++i)
    for (int j = 0; j < kNWRows; ++j) {
      const int L1 = kWCols[i].L;
      const int L2 = kWRows[j].L;
      long long N = 2LL * L1 * L2;
      if (N < need) continue;
      if (nfft > 0 && N != nfft) continue;
      bool better = best < 0 || N < best;
      if (!better && N == best) {
        const int b1 = kWCols[bi].L;
        auto pref = [](int l) { return l <= 2048 ? l : 1; };
        better = pref(L1) > pref(b1);
      }
      if (better) { best = N; bi = i; bj = j; }
    }
  return best;
}

// What a plan for `need` points (n + max|lag|; nfft != 0: exactly that transform) would run on -- no GPU, no plan:
// out5 = {N, N1, N2h, has a dual / flow pair kernel, that kernel has the q24 T storage}; N = -1: no single transform
// (a segmented plan).  tests/test_abi.py walks it: every instantiated pair kernel is reachable and vice versa.
extern "C" int xcx_plan_geometry(long long need, int nfft, int* out5) {
  if (!out5 || need < 1) return set_err(XCX_EINVAL, "bad argument");
  int bi, bj;
  const long long N = plan_choose(need, nfft, bi, bj);
  out5[0] = (int)N; out5[1] = out5[2] = out5[3] = out5[4] = 0;
  if (N < 0) return XCX_OK;
  out5[1] = kWCols[bi].L; out5[2] = kWRows[bj].L;
  for (int i = 0; i < kNWFused; ++i)
    if (kWFused[i].N1 == out5[1] && kWFused[i].N2h == out5[2]) { out5[3] = 1; out5[4] = kWFused[i].q24(); }
  return XCX_OK;
}
// every instantiated dual / flow kernel, for the same test: out[2 i], out[2 i + 1] = (N1, N2h); returns the count
extern "C" int xcx_plan_fused_list(int* out, int cap, int* count) {
  if (!count) return set_err(XCX_EINVAL, "null argument");
  *count = kNWFused;
  for (int i = 0; i < kNWFused && out && 2 * i + 1 < cap; ++i) { out[2 * i] = kWFused[i].N1; out[2 * i + 1] = kWFused[i].N2h; }
  return XCX_OK;
}

#define seg_plan_create xcx_internal_seg_plan_create      // seg_plan.hip
#define seg_run xcx_internal_seg_run

extern "C" int xcx_plan_cc1d(xcx_ctx* ctx, int n, int nfft, int idx_min,
                             int idx_max, int flags, xcx_plan** out) {
  if (!ctx || !out) return set_err(XCX_EINVAL, "null argument");
  if (n < 1) return set_err(XCX_EINVAL, "n must be >= 1 (got %d)", n);
  const long long full = 2LL * n - 1;
  if (idx_max == 0) idx_max = (int)full;
  if (idx_min < 0 || idx_max > full || idx_min >= idx_max)
    return set_err(XCX_EINVAL, "lag slice [%d,%d) outside [0,%lld)", idx_min,
                   idx_max, full);
  // A circular correlation of length N aliases lag m with m +- N, and the linear correlation
  // vanishes beyond |m| = n-1: the retained lags [idx_min, idx_max) - (n-1) are exact as soon
  // as N >= n + max|lag|.  Full-lag plans need 2n-1 as ever; clipped plans (the reference's
  // clip_lag) get a proportionally shorter transform.
  const long long maxlag = std::max<long long>((long long)n - 1 - idx_min, (long long)idx_max - n);
  // kPlanShortOperand (library-internal, fftn.hip): the caller guarantees that the first operand is zero
  // beyond its first idx_min + 1 samples and that only lags <= 0 are retained -- the aliases of the retained
  // lags then fall outside the linear correlation's support as soon as N >= n (chirp convolutions, the blocks of the
  // segmented plans).  Mirror image (idx_min >= n - 1, segmented plans): the SECOND operand is the short one and only
  // lags >= 0 are retained; Geom::wrap tells the row pass that no output position belongs to a negative lag.
  const long long need = (flags & kPlanShortOperand) ? (long long)n : (long long)n + maxlag;
  // choose N = 2*N1*N2h: smallest supported >= need (or == nfft)
  int bi = -1, bj = -1;
  long long best = plan_choose(need, nfft, bi, bj);
  // XCX_SEG_FORCE=1 (tests): segmented plan even where one transform would do
  const bool seg_force = getenv("XCX_SEG_FORCE") && atoi(getenv("XCX_SEG_FORCE")) != 0;      // read per call
  if (seg_force && nfft == 0 && !(flags & (XCX_F_WHITEN | XCX_F_STACK))) best = -1;
  if (best < 0) {
    const bool seg_ok = nfft == 0 && !(flags & (XCX_F_WHITEN | XCX_F_STACK));
    if (!seg_ok)
      return set_err(XCX_EUNSUPPORTED,
                     "no supported transform length for n=%d (nfft=%d); "
                     "supported N = 2*N1*N2h up to %d (longer windows: segmented plans, without "
                     "XCX_WHITEN / XCX_STACK / nfft)", n, nfft,
                     2 * kWCols[kNWCols - 1].L * kWRows[kNWRows - 1].L);
    return seg_plan_create(ctx, n, idx_min, idx_max, flags, out);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  xcx_plan* p = new (std::nothrow) xcx_plan();
  if (!p) return set_err(XCX_ENOMEM, "host allocation failed");
  p->ctx = ctx;
  p->wcol = &kWCols[bi]; p->wrow = &kWRows[bj];
  Geom& g = p->g;
  g.n = n;
  g.N = (int)best;
  g.N1 = kWCols[bi].L;
  g.N2h = kWRows[bj].L;
  g.N2 = 2 * g.N2h;
  {
    const char* de = getenv("XCX_DUAL");
    if (!(de && strcmp(de, "0") == 0))
      for (int i = 0; i < kNWFused; ++i)
        if (kWFused[i].N1 == g.N1 && kWFused[i].N2h == g.N2h) p->dual = &kWFused[i];
    if (const char* ce = getenv("XCX_CHAINS")) p->nchains = std::max(1, std::min(xcx_plan::kMaxChains, atoi(ce)));
    {
      // opt-in for now (XCX_FLOW=1): parity-green, within +-3 % of the per-batch launches
      const char* fl = getenv("XCX_FLOW");
      p->flow = p->dual && fl && strcmp(fl, "1") == 0;
      if (const char* fr = getenv("XCX_FLOW_RING")) p->flow_R = std::max(kFlowLag + 1, std::min(64, atoi(fr)));
    }
    if (!p->dual) p->nchains = 1;
    if (p->dual) {
      // batch so that the whole T ring (chains x 2 buffers) is ~160 MB: it then stays
      // in the 256 MiB Infinity Cache next to the spectra, and T traffic stops at the cache.
      // (160 MB = 28 pair-windows at N = 368 640: the 28 pairs among 8 stations, pair_order.h)
      const size_t tm_bytes = sizeof(cf) * (size_t)g.N1 * (g.N2h + 1);
      static const size_t ring = (size_t)(getenv("XCX_RING_MB") ? std::max(16, atoi(getenv("XCX_RING_MB"))) : 160) << 20;
      p->batch = (int)std::max<size_t>(4, std::min<size_t>(512, ring / (2 * (size_t)p->nchains * tm_bytes)));
    }
  }
  g.idx_min = idx_min;
  g.idx_max = idx_max;
  g.shift = (int)(((long long)n - 1 - idx_min) % g.N1);
  if (g.shift < 0) g.shift += g.N1;
  g.wrap = ((flags & kPlanShortOperand) && idx_min >= n - 1) ? g.N : g.N - idx_min;
  g.flags = flags & 0xffff;
  g.band_hi = g.N;
  p->L = idx_max - idx_min;
  size_t sp = (size_t)g.N1 * g.N2h + g.N1;
  p->spec_pitch = (sp + 15) / 16 * 16;

  std::vector<float2> t;
  int rc;
  fill_table(t, g.N1, g.N1);
  if ((rc = upload(ctx, p->tw_N1, t.data(), t.size() * sizeof(float2)))) { delete p; return rc; }
  fill_table(t, g.N2h, g.N2h);
  if ((rc = upload(ctx, p->tw_N2h, t.data(), t.size() * sizeof(float2)))) { delete p; return rc; }
  fill_table(t, g.N2h, g.N2);
  if ((rc = upload(ctx, p->tw_unt, t.data(), t.size() * sizeof(float2)))) { delete p; return rc; }
  {
    const int nhi = (g.N / 2) / kTwQL + 2;
    std::vector<float2> lo, hi;
    fill_table(lo, kTwQL, g.N);
    fill_table(hi, nhi, g.N, kTwQL);
    lo.insert(lo.end(), hi.begin(), hi.end());
    if ((rc = upload(ctx, p->tw_N, lo.data(), lo.size() * sizeof(float2)))) { delete p; return rc; }
  }
  *out = p;
  return XCX_OK;
}

extern "C" int xcx_plan_destroy(xcx_plan* p) {
  if (!p) return XCX_OK;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->stream);
  for (auto& e : p->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto& ch : p->chains) {
    if (ch.stream && ch.stream != p->ctx->stream) (void)hipStreamSynchronize(ch.stream);   // the context owns it
    if (ch.done) (void)hipEventDestroy(ch.done);
  }
  if (p->ev_fwd) (void)hipEventDestroy(p->ev_fwd);
  if (p->fwd_last) (void)hipEventDestroy(p->fwd_last);
  if (p->prep_stream) { (void)hipStreamSynchronize(p->prep_stream); (void)hipStreamDestroy(p->prep_stream); }
  if (p->prep_done) (void)hipEventDestroy(p->prep_done);
  for (auto& e : p->pairs_done) if (e) (void)hipEventDestroy(e);
  p->spectra_next.release();
  p->flow_sync.release();
  if (p->seg) {
    if (p->seg->sub) xcx_plan_destroy(p->seg->sub);
    if (p->seg->subp) xcx_plan_destroy(p->seg->subp);
    DevBuf* sb[] = {&p->seg->segbuf, &p->seg->subout, &p->seg->stats, &p->seg->ptrs, &p->seg->pairs};
    for (DevBuf* b : sb) b->release();
    delete p->seg;
    p->seg = nullptr;
  }
  DevBuf* bufs[] = {&p->tw_N1, &p->tw_N2h, &p->tw_unt, &p->tw_N,
                    &p->Tm_f, &p->spectra, &p->T, &p->descs,
                    &p->dtraces, &p->dout, &p->dpairs, &p->psum, &p->psq, &p->gramC, &p->gramTiles, &p->gramPidx,
                    &p->gramPack, &p->gramRecv, &p->gramOwn, &p->bandTab};
  for (DevBuf* b : bufs) b->release();
  delete p;
  return XCX_OK;
}

extern "C" int xcx_plan_set_whiten(xcx_plan* p, double fs, double fmin,
                                   double fmax, double ftaper) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  if (p->seg) return set_err(XCX_EUNSUPPORTED, "segmented (long-window) plans do not whiten");
  if (!(fs > 0) || !(fmin >= 0) || !(fmax > fmin) || ftaper < 0)      // before anything is touched: a refused
    return set_err(XCX_EINVAL, "bad whitening band");                  // call leaves the plan as it was
  if (p->acc_count > 0 || p->acc_pending > 0 || p->acc_broken)   // days already transformed with the old band wait for their Gram
    return set_err(XCX_EINVAL, "xcx_plan_set_whiten during a stack accumulation (%d adds, %d spectra sets "
                   "pending): finish or reset it first", p->acc_count, p->acc_pending);   // update (also after a failed flush)
  if (p->prep_valid) {            // spectra prepared ahead (xcx_cc1d_prepare) carry the old band: let them finish, drop them
    HIP_TRY(hipStreamWaitEvent(p->ctx->stream, p->prep_done, 0));
    p->prep_valid = false;
  }
  const double bin = (double)p->g.N / fs;   // bins per Hz
  // band edges in bins, float for the kernels; rounding to float must not carry an edge across an integer bin
  // (fmax * bin = 21377.9994 rounds to 21378.0f and would admit bin 21378): lower edges compare k >= edge,
  // upper edges k <= edge, so step the float back to the double's side of the integer it crossed
  auto lower = [](double e) { float f = (float)e; return ((double)f < e && ceil((double)f) < ceil(e)) ? nextafterf(f, INFINITY) : f; };
  auto upper = [](double e) { float f = (float)e; return ((double)f > e && floor((double)f) > floor(e)) ? nextafterf(f, -INFINITY) : f; };
  p->wb[0] = lower((fmin - ftaper) * bin);
  p->wb[1] = lower(fmin * bin);
  p->wb[2] = upper(fmax * bin);
  p->wb[3] = upper((fmax + ftaper) * bin);
  // Parseval norm of the ideal whitened spectrum |X[k]| = g[k]
  const int N = p->g.N;
  double acc = 0.0;
  for (int k = 0; k <= N / 2; ++k) {
    float f = (float)k;
    double gk;
    if (f >= p->wb[1] && f <= p->wb[2]) gk = 1.0;
    else if (f >= p->wb[0] && f < p->wb[1])
      gk = 0.5 * (1.0 - cos(M_PI * (f - p->wb[0]) / (p->wb[1] - p->wb[0])));
    else if (f > p->wb[2] && f <= p->wb[3])
      gk = 0.5 * (1.0 + cos(M_PI * (f - p->wb[2]) / (p->wb[3] - p->wb[2])));
    else gk = 0.0;
    acc += ((k == 0 || k == N / 2) ? 1.0 : 2.0) * gk * gk;
  }
  p->fixed_norm = (float)sqrt(acc / N);
  p->whiten_set = true;
  // q24 storage of T: |X[k]| = gain(k) * scale exactly, so |T[m1][k2]| <= scale^2 * sum_k1 gain(k1 N2 + k2)^2 for
  // every input; the largest column sum gives the fixed-point scale (2 % headroom for the float32 rounding of the
  // transform).  XCX_T_STORAGE=f32 keeps float2.
  {
    const Geom& g = p->g;
    double bmax = 0.0;
    for (int k2 = 0; k2 <= g.N2h; ++k2) {
      double sum = 0.0;
      for (int k1 = 0; k1 < g.N1; ++k1) {
        long long k = (long long)k1 * g.N2 + k2;
        if (k > N - k) k = N - k;
        const float f = (float)k;
        double gk;
        if (f >= p->wb[1] && f <= p->wb[2]) gk = 1.0;
        else if (f >= p->wb[0] && f < p->wb[1]) gk = 0.5 * (1.0 - cos(M_PI * (f - p->wb[0]) / (p->wb[1] - p->wb[0])));
        else if (f > p->wb[2] && f <= p->wb[3]) gk = 0.5 * (1.0 + cos(M_PI * (f - p->wb[2]) / (p->wb[3] - p->wb[2])));
        else gk = 0.0;
        sum += gk * gk;
      }
      bmax = std::max(bmax, sum);
    }
    const double nrm = (g.flags & XCX_F_NORMALIZE) ? (double)p->fixed_norm : 1.0;
    const double sc = (double)(float)(1.0 / (nrm * sqrt((double)N)));      // sw_scale()
    const double B = bmax * sc * sc;
    const char* ts_env = getenv("XCX_T_STORAGE");      // read per call: a process may compare the two
    const bool want = ts_env ? !strcmp(ts_env, "q24") : kQ24Default;
    p->t_q24 = false;
    if (want && (g.flags & XCX_F_WHITEN) && B > 0.0 && p->dual && p->dual->q24 && p->dual->q24()) {
      const double tq = 0.98 * 8388608.0 / B;
      if (std::isfinite(tq) && tq < 1e30 && tq > 1e-30) {
        p->tq = (float)tq;
        p->tqi = (float)(1.0 / (double)p->tq);
        p->t_q24 = true;
      }
    }
  }
  // whiten_gain() is 0 for every folded bin k > wb[3] (float compare, k exact below 2^24): those spectrum
  // elements are stored as exact zeros by the forward column pass and the pair pass skips their loads
  {
    static const bool prune = !(getenv("XCX_BAND_PRUNE") && atoi(getenv("XCX_BAND_PRUNE")) == 0);
    const double hi = floor((double)p->wb[3]);
    const int band_hi = (prune && (p->g.flags & XCX_F_WHITEN) && hi >= 0 && hi < (double)N) ? (int)hi : N;
    if (band_hi != p->g.band_hi) {
      // another band: whatever the buffers hold above the new edge is no longer "stored zeros" (a wider band of an
      // earlier call wrote there) -- forget every buffer that was zeroed for the old one
      for (auto& z : p->zeroed) z = {};
      p->gram_zeroed = {};
    }
    p->g.band_hi = band_hi;
  }
  return XCX_OK;
}

extern "C" int xcx_pair_groups(const int* pairs, int S, int P, int batch, int* perm, int* bounds, int* ngroups) {
  if (!pairs || !perm || !bounds || !ngroups) return set_err(XCX_EINVAL, "null argument");
  if (S < 1 || P < 1 || batch < 1) return set_err(XCX_EINVAL, "S, P, batch must be >= 1");
  for (int i = 0; i < P; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= S || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= S)
      return set_err(XCX_EINVAL, "pair %d = (%d,%d) outside [0,%d)", i, pairs[2 * i], pairs[2 * i + 1], S);
  std::vector<int> pm, bd;
  pair_groups(pairs, S, P, batch, pm, bd);
  std::copy(pm.begin(), pm.end(), perm);
  std::copy(bd.begin(), bd.end(), bounds);
  *ngroups = (int)bd.size() - 1;
  return XCX_OK;
}

extern "C" int xcx_plan_set_batch(xcx_plan* p, int b) {
  if (!p || b < 1 || b > 65535) return set_err(XCX_EINVAL, "batch must be in [1,65535]");
  p->batch = b;
  p->batch_user = true;
  p->desc_key = 0;
  return XCX_OK;
}

extern "C" int xcx_plan_info(xcx_plan* p, int* out6) {
  if (!p || !out6) return set_err(XCX_EINVAL, "null argument");
  const xcx_plan* q = (p->seg && p->seg->sub) ? p->seg->sub : p;       // segmented: the block transform
  out6[0] = q->g.N; out6[1] = q->g.N1; out6[2] = q->g.N2; out6[3] = p->L;
  out6[4] = q->batch; out6[5] = (int)(2 * q->spec_pitch);
  return XCX_OK;
}

extern "C" int xcx_plan_t_storage(xcx_plan* p, int* ts) {
  if (!p || !ts) return set_err(XCX_EINVAL, "null argument");
  const xcx_plan* q = (p->seg && p->seg->sub) ? p->seg->sub : p;
  *ts = (q->t_q24 && q->dual && !q->flow && (q->g.flags & XCX_F_WHITEN) && !(q->g.flags & XCX_F_STACK) && q->whiten_set) ? 1 : 0;
  return XCX_OK;
}

extern "C" int xcx_plan_zero_skip(xcx_plan* p, int* out2) {
  if (!p || !out2) return set_err(XCX_EINVAL, "null argument");
  const xcx_plan* q = (p->seg && p->seg->sub) ? p->seg->sub : p;
  int ts = 0;
  (void)xcx_plan_t_storage(p, &ts);
  static const bool zs = !(getenv("XCX_ZSKIP") && atoi(getenv("XCX_ZSKIP")) == 0);
  out2[0] = (ts && zs && q->dual && q->dual->zskip) ? q->dual->zskip(q->g) : 0;
  out2[1] = q->g.N1 / kWNT;
  return XCX_OK;
}

extern "C" int xcx_plan_counters(xcx_plan* p, long long* launches, long long* pw) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  if (launches) *launches = p->launches;
  if (pw) *pw = p->pair_windows;
  return XCX_OK;
}

extern "C" int xcx_plan_profile(xcx_plan* p, int enable) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  for (auto& e : p->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  p->evs.clear();
  p->profiling = enable < 0 ? 0 : enable;
  return XCX_OK;
}

extern "C" int xcx_plan_profile_read(xcx_plan* p, int kernel_id, char* name, size_t name_len,
                                     double* total_us, long long* launches, long long* units) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  if (kernel_id < 0 || kernel_id >= KID_COUNT) return set_err(XCX_EINVAL, "kernel id out of range");
  HIP_TRY(hipStreamSynchronize(p->ctx->stream));
  double tot = 0.0; long long nl = 0, nu = 0;
  for (auto& e : p->evs) {
    if (e.kid != kernel_id) continue;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
    tot += 1e3 * ms; nl += 1; nu += e.units;
  }
  if (name && name_len) snprintf(name, name_len, "%s", kKernelNames[kernel_id]);
  if (total_us) *total_us = tot;
  if (launches) *launches = nl;
  if (units) *units = nu;
  return XCX_OK;
}

// ---------------------------------------------------------------------------
// forward: station-windows sw in [sw_begin, sw_end) of an S x W grid
// (sw = s*W + wl, window index w0 + wl) -> spectrum slots slot0 + sw
// ---------------------------------------------------------------------------
int xcx_internal_forward(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                         int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid);
// n_valid > 0: the windows hold n_valid < n samples (zero beyond; plans without demean / normalise only)
static int forward(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                   int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid = 0) {
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  // station-windows per forward launch: large launches fill the chip
  static const int chunk_cap = getenv("XCX_FWD_CHUNK") ? std::max(1, atoi(getenv("XCX_FWD_CHUNK"))) : 512;
  const int kChunk = std::max(std::min(64, chunk_cap), std::min(chunk_cap, sw_end - sw_begin));
  int rc;
  if ((g.flags & XCX_F_WHITEN) && !p->whiten_set)
    return set_err(XCX_EINVAL, "XCX_WHITEN set but xcx_plan_set_whiten() not called");
  // Tm_f / psum / psq are shared by every forward of the plan, whichever stream it runs on (xcx_cc1d_prepare runs it
  // on the preparation stream while the main stream may still be inside the forward of an unprepared call, and the
  // other way round): wait for the previous one when it ran on another stream
  if (!p->fwd_last) HIP_TRY(hipEventCreateWithFlags(&p->fwd_last, hipEventDisableTiming));
  else if (p->fwd_last_stream && p->fwd_last_stream != ctx->stream) HIP_TRY(hipStreamWaitEvent(ctx->stream, p->fwd_last, 0));
  if (g.band_hi < g.N) {
    // band-limited plan: the forward column pass skips the stores of the out-of-band stretch of every
    // column (60 % of the spectrum bytes for 0.1-10 Hz at 50 Hz), so the buffer is zeroed once per
    // allocation and band; the pair pass reads its "stored zeros" from there (wave_kernels.h)
    bool ok = false;
    for (auto& z : p->zeroed) ok = ok || (z.ptr == p->spectra.ptr && z.gen == p->spectra.gen && z.band_hi == g.band_hi);
    if (!ok) {
      HIP_TRY(hipMemsetAsync(p->spectra.ptr, 0, p->spectra.bytes, ctx->stream));
      p->zeroed[p->zeroed_next] = {p->spectra.ptr, p->spectra.gen, g.band_hi};
      p->zeroed_next ^= 1;
    }
  }
  const size_t tf_per = (size_t)(g.N2h + 1) * g.N1;
  if ((rc = p->psum.ensure(ctx, sizeof(double) * kStatParts * kChunk))) return rc;
  if ((rc = p->psq.ensure(ctx, sizeof(double) * kStatParts * kChunk))) return rc;
  if ((rc = p->Tm_f.ensure(ctx, sizeof(cf) * tf_per * kChunk))) return rc;
  for (int b = sw_begin; b < sw_end; b += kChunk) {
    const int cnt = std::min(kChunk, sw_end - b);
    {
      ProfScope ps(p, KID_STATS, cnt);
      if (g.flags & XCX_F_DEMEAN)
        hipLaunchKernelGGL(k_wstats<0>, dim3(kStatParts, cnt), dim3(kNT), 0, ctx->stream, traces, stride_s,
                           stride_w, W, w0, b, g.n, g.flags, (const double*)nullptr, (double*)p->psum.ptr);
      if ((g.flags & XCX_F_NORMALIZE) && !(g.flags & XCX_F_WHITEN))
        hipLaunchKernelGGL(k_wstats<1>, dim3(kStatParts, cnt), dim3(kNT), 0, ctx->stream, traces, stride_s,
                           stride_w, W, w0, b, g.n, g.flags, (const double*)p->psum.ptr, (double*)p->psq.ptr);
    }
    HIP_TRY(hipGetLastError());
    WF1Args f1{};
    f1.traces = traces; f1.stride_s = stride_s; f1.stride_w = stride_w;
    f1.W = W; f1.w0 = w0; f1.sw0 = b;
    f1.psum = (const double*)p->psum.ptr;
    f1.Tf = (cf*)p->Tm_f.ptr;
    f1.tw_sub = (const cf*)p->tw_N2h.ptr;
    f1.tw_unt = (const cf*)p->tw_unt.ptr;
    f1.tw_N = (const cf*)p->tw_N.ptr;
    f1.g = g;
    if (n_valid > 0) f1.g.n = n_valid;
    // a launch whose Tf (written here, read back by the column pass) is beyond what stays cache-resident streams
    const int nt_env = getenv("XCX_FWD_NT") ? atoi(getenv("XCX_FWD_NT")) : -1;    // A/B and tests (read per call): 0 never, 1 always
    const int nt = nt_env >= 0 ? (nt_env != 0) : (sizeof(cf) * tf_per * (size_t)cnt >= kFwdNtBytes);
    f1.nt = nt;
    {
      ProfScope ps(p, KID_FWD_ROWS, cnt);
      HIP_TRY(p->wrow->fwd(f1, dim3(g.N1 / kWRB, cnt), ctx->stream));
    }
    WF2Args f2{};
    f2.Tf = (const cf*)p->Tm_f.ptr;
    f2.psq = (const double*)p->psq.ptr;
    f2.X = (cf*)p->spectra.ptr;
    f2.spec_pitch = p->spec_pitch;
    f2.slot0 = p->fwd_slot_ld ? slot0 : slot0 + b;
    f2.sw_first = b; f2.slot_w = p->fwd_slot_w; f2.slot_ld = p->fwd_slot_ld;
    f2.tw_sub = (const cf*)p->tw_N1.ptr;
    f2.wb0 = p->wb[0]; f2.wb1 = p->wb[1]; f2.wb2 = p->wb[2]; f2.wb3 = p->wb[3];
    f2.fixed_norm = p->fixed_norm;
    f2.g = g;
    f2.nt = nt;
    {
      ProfScope ps(p, KID_FWD_COLS, cnt);
      HIP_TRY(p->wcol->fwd(f2, dim3(g.N2h + 1, cnt), ctx->stream));
    }
    p->launches += 4;
  }
  HIP_TRY(hipEventRecord(p->fwd_last, ctx->stream));
  p->fwd_last_stream = ctx->stream;
  return XCX_OK;
}

// pair-window batch: descriptors [d0, d0+B) of the device descriptor array
static int pair_batch(xcx_plan* p, size_t d0, int B, float* out, int nstack, const cf* Xsrc = nullptr) {
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  if (nstack > 2)      // MODE 1 of the column pass sums exactly two slots; longer stacks go through the Gram kernels
    return set_err(XCX_EINVAL, "pair_batch: a stack of %d windows in the pair column pass (at most 2)", nstack);
  WI1Args a{};
  a.X = Xsrc ? Xsrc : (const cf*)p->spectra.ptr;
  a.spec_pitch = p->spec_pitch;
  a.pw = (const PwDesc*)p->descs.ptr + d0;
  a.B = B;
  a.nstack = nstack;
  a.T = (cf*)p->T.ptr;
  a.tw_sub = (const cf*)p->tw_N1.ptr;
  a.tw_N = (const cf*)p->tw_N.ptr;
  a.g = g;
  const int per_xcd = (g.N2h + 1 + 7) / 8;
  {
    ProfScope ps(p, KID_PAIR_COLS, B);
    HIP_TRY(p->wcol->pair(a, dim3(8u * per_xcd * B), ctx->stream));
  }
  WI2Args r{};
  r.T = (const cf*)p->T.ptr;
  r.pw = a.pw;
  r.out = out;
  r.tw_sub = (const cf*)p->tw_N2h.ptr;
  r.tw_unt = (const cf*)p->tw_unt.ptr;
  r.g = g;
  {
    ProfScope ps(p, KID_PAIR_ROWS, B);
    HIP_TRY(p->wrow->pair(r, dim3(g.N1 / kWRB, B), ctx->stream));
  }
  p->launches += 2;
  p->pair_windows += B;
  return XCX_OK;
}

// Dual pipeline: dual_push(batch) launches [rows of the pending batch +
// columns of this batch]; dual_flush() launches the rows of the last batch.
// T holds two batch-sized buffers used alternately.
static int dual_launch(xcx_plan* p, int chain, size_t d0, int Bc, int nstack, float* out, const cf* Xsrc = nullptr) {
  const Geom& g = p->g;
  xcx_plan::Chain& ch = p->chains[chain];
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  cf* Tbase = (cf*)p->T.ptr + (size_t)chain * 2 * tm_per * p->batch;
  if (nstack > 2)      // MODE 1 of the column pass sums exactly two slots; longer stacks go through the Gram kernels
    return set_err(XCX_EINVAL, "dual_launch: a stack of %d windows in the pair column pass (at most 2)", nstack);
  WDualArgs a{};
  a.X = Xsrc ? Xsrc : (const cf*)p->spectra.ptr;
  a.spec_pitch = p->spec_pitch;
  const int cs = ch.slot ^ 1;                       // slot written by this launch
  a.pw_c = (const PwDesc*)p->descs.ptr + d0;
  a.Bc = Bc;
  a.nstack = nstack;
  a.Tc = Tbase + (size_t)cs * tm_per * p->batch;
  a.pw_r = (const PwDesc*)p->descs.ptr + ch.d0;
  a.Br = ch.B;
  a.Tr = Tbase + (size_t)ch.slot * tm_per * p->batch;
  a.out = out;
  a.tw_N1 = (const cf*)p->tw_N1.ptr;
  a.tw_N2h = (const cf*)p->tw_N2h.ptr;
  a.tw_unt = (const cf*)p->tw_unt.ptr;
  a.tw_N = (const cf*)p->tw_N.ptr;
  a.g = g;
  // plain pair products of whitened spectra (nstack == 1, no precomputed cross-spectrum): T may go through the ring as
  // 2 x 24-bit fixed point; the row launch that reads a slot follows the column launch that wrote it within one run
  // -- decided ONCE per run (dual_run): the launch that drains a pipeline carries nstack = 1 whatever the run was
  a.ts = p->run_ts;
  a.tq = p->tq; a.tqi = p->tqi;
  a.zb0 = p->run_ts ? p->run_zb0 : 0;
  if (Bc == 0 && a.Br == 0) return XCX_OK;
#ifdef XCX_STAMPS
  {
    // diagnostic build: the XCX_STAMP_LAUNCH-th dual launch (a steady-state one) records
    // per-wave phase stamps; read back with xcx_debug_stamps()
    static const long which = getenv("XCX_STAMP_LAUNCH") ? atol(getenv("XCX_STAMP_LAUNCH")) : 400;
    a.stamps = nullptr;
    if (p->launches == which) {
      const size_t nw = (size_t)8 * 4 * 64 * (size_t)std::max(Bc, a.Br);     // generous upper bound of waves
      if (p->stampbuf.ensure(p->ctx, nw * kNStamp * sizeof(unsigned long long))) return XCX_ENOMEM;
      HIP_TRY(hipMemsetAsync(p->stampbuf.ptr, 0, nw * kNStamp * sizeof(unsigned long long), ch.stream));
      a.stamps = (unsigned long long*)p->stampbuf.ptr;
      p->stamp_waves = nw;
    }
  }
#endif
  {
    ProfScope ps(p, KID_PAIR_FUSED, Bc > 0 ? Bc : a.Br, ch.stream);
    HIP_TRY(p->dual->dual(a, ch.stream));
  }
  p->launches += 1;
  p->pair_windows += Bc;
  ch.d0 = d0; ch.B = Bc; ch.slot = cs;
  return XCX_OK;
}

// run descriptors [first, first+count) through the dual pipelines: batches are
// dealt round-robin to the chains; on return every chain has been drained and
// the context's main stream waits for all of them
static int dual_run(xcx_plan* p, size_t first, size_t count, int nstack, float* out, const cf* Xsrc = nullptr,
                    const std::vector<int>* bounds = nullptr) {
  xcx_ctx* ctx = p->ctx;
  int rc;
  const int B = p->batch;
  // station-grouped batches: the descriptors are whole windows of P pairs, each cut at the same group starts
  const size_t Pw = (bounds && bounds->size() >= 2) ? (size_t)bounds->back() : 0;
  if (Pw == 0 || count % Pw != 0) bounds = nullptr;
  const int nch = p->nchains;
  for (int c = 0; c < nch; ++c) {
    xcx_plan::Chain& ch = p->chains[c];
    if (!ch.stream) {
      if (nch == 1) ch.stream = ctx->stream;
      else {
        if (!ctx->pipe[c]) HIP_TRY(hipStreamCreateWithFlags(&ctx->pipe[c], hipStreamNonBlocking));   // XCX_CHAINS > 2
        ch.stream = ctx->pipe[c];
      }
      HIP_TRY(hipEventCreateWithFlags(&ch.done, hipEventDisableTiming));
    }
    ch.B = 0;
  }
  p->run_ts = (p->t_q24 && nstack == 1 && !Xsrc && (p->g.flags & XCX_F_WHITEN) && !(p->g.flags & XCX_F_STACK) && p->whiten_set) ? 1 : 0;
  {
    // q24 runs read the plan's own band-limited spectra: the column waves do not issue the loads of the blocks that
    // are zero in every column (XCX_ZSKIP=0: the generic kernel, for A/B)
    static const bool zs = !(getenv("XCX_ZSKIP") && atoi(getenv("XCX_ZSKIP")) == 0);
    p->run_zb0 = (p->run_ts && zs && p->dual->zskip) ? p->dual->zskip(p->g) : 0;
  }
  if (!p->ev_fwd) HIP_TRY(hipEventCreateWithFlags(&p->ev_fwd, hipEventDisableTiming));
  // wall time of the whole pair phase on the main stream (the pipelines overlap,
  // so per-launch durations do not add up)
  ProfScope phase(p, KID_PAIR_PHASE, (int)count);
  if (nch > 1) {
    HIP_TRY(hipEventRecord(p->ev_fwd, ctx->stream));          // spectra / descriptors are ready
    for (int c = 0; c < nch; ++c) HIP_TRY(hipStreamWaitEvent(p->chains[c].stream, p->ev_fwd, 0));
  }
  size_t k = 0;
  if (bounds) {
    for (size_t w0 = first; w0 < first + count; w0 += Pw)
      for (size_t gi = 0; gi + 1 < bounds->size(); ++gi) {
        const int b0 = (*bounds)[gi], b1 = (*bounds)[gi + 1];
        for (int q0 = b0; q0 < b1; q0 += B, ++k)       // a group never exceeds B; the loop is a guard
          if ((rc = dual_launch(p, (int)(k % nch), w0 + q0, std::min(B, b1 - q0), nstack, out, Xsrc))) return rc;
      }
  } else {
    for (size_t d0 = first; d0 < first + count; d0 += B, ++k)
      if ((rc = dual_launch(p, (int)(k % nch), d0, (int)std::min((size_t)B, first + count - d0), nstack, out, Xsrc)))
        return rc;
  }
  for (int c = 0; c < nch; ++c) {
    if (p->chains[c].B > 0 && (rc = dual_launch(p, c, 0, 0, 1, out, Xsrc))) return rc;   // rows of the last batch
    p->chains[c].B = 0;
    if (nch > 1) {
      HIP_TRY(hipEventRecord(p->chains[c].done, p->chains[c].stream));
      HIP_TRY(hipStreamWaitEvent(ctx->stream, p->chains[c].done, 0));
    }
  }
  return XCX_OK;
}

// the whole run [first, first+count) as ONE launch of the flow kernel (see wave_kernels.h)
static int flow_run(xcx_plan* p, size_t first, size_t count, int nstack, float* out, const cf* Xsrc = nullptr) {
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  int rc;
  if ((rc = flow_check(ctx))) return rc;                     // a previous launch timed out?
  static const int lag = getenv("XCX_FLOW_LAG") ? std::max(1, atoi(getenv("XCX_FLOW_LAG"))) : kFlowLag;
  const int B = p->batch, R = std::max(p->flow_R, lag + 2);
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  const size_t tbytes = sizeof(cf) * tm_per * (size_t)B * R;
  if (ctx->flow_T_bytes < tbytes) {
    if (ctx->flow_T) { HIP_TRY(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->flow_T); ctx->flow_T = nullptr; ctx->flow_T_bytes = 0; }
    // uncached: producers and consumers of T sit on different XCDs inside one launch
    hipError_t e = hipExtMallocWithFlags(&ctx->flow_T, tbytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); return set_err(XCX_ENOMEM, "uncached T ring of %zu bytes: %s", tbytes, hipGetErrorString(e)); }
    ctx->flow_T_bytes = tbytes;
  }
  if (!ctx->flow_err_dev) {
    HIP_TRY(hipMalloc((void**)&ctx->flow_err_dev, sizeof(int)));
    HIP_TRY(hipHostMalloc((void**)&ctx->flow_err_host, sizeof(int), hipHostMallocDefault));
    *ctx->flow_err_host = 0;
    HIP_TRY(hipMemsetAsync(ctx->flow_err_dev, 0, sizeof(int), ctx->stream));
  }
  const int nbatch = (int)((count + B - 1) / B);
  const size_t nsync = (size_t)(nbatch + 1) * 8 * kFlowShardInts;     // ints per counter array
  if ((rc = p->flow_sync.ensure(ctx, sizeof(int) * 2 * nsync))) return rc;
  HIP_TRY(hipMemsetAsync(p->flow_sync.ptr, 0, sizeof(int) * 2 * nsync, ctx->stream));
  if (nstack > 2)      // MODE 1 of the column pass sums exactly two slots; longer stacks go through the Gram kernels
    return set_err(XCX_EINVAL, "flow_run: a stack of %d windows in the pair column pass (at most 2)", nstack);
  WFlowArgs a{};
  a.X = Xsrc ? Xsrc : (const cf*)p->spectra.ptr;
  a.spec_pitch = p->spec_pitch;
  a.pw = (const PwDesc*)p->descs.ptr + first;
  a.count = (int)count;
  a.B = B; a.R = R;
  a.nstack = nstack;
  a.T = (cf*)ctx->flow_T;
  a.out = out;
  a.cdone = (int*)p->flow_sync.ptr;
  a.rdone = a.cdone + nsync;
  a.err = ctx->flow_err_dev;
  a.tw_N1 = (const cf*)p->tw_N1.ptr;
  a.tw_N2h = (const cf*)p->tw_N2h.ptr;
  a.tw_unt = (const cf*)p->tw_unt.ptr;
  a.tw_N = (const cf*)p->tw_N.ptr;
  a.g = g;
  a.lag = lag;
  {
    ProfScope phase(p, KID_PAIR_PHASE, (int)count);
    ProfScope ps(p, KID_PAIR_FUSED, (int)count);
    HIP_TRY(p->dual->flow(a, ctx->stream));
  }
  HIP_TRY(hipMemcpyAsync(ctx->flow_err_host, ctx->flow_err_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  static const bool dbg_sync = getenv("XCX_FLOW_SYNC") != nullptr;     // debugging aid
  if (dbg_sync) HIP_TRY(hipDeviceSynchronize());
  ctx->flow_ran = true;
  p->launches += 1;
  p->pair_windows += (long long)count;
  return XCX_OK;
}

static unsigned long long fnv(const void* data, size_t bytes, unsigned long long h) {
  const unsigned char* c = (const unsigned char*)data;
  for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ULL; }
  return h;
}

// ---------------------------------------------------------------------------
// Time-stack through the cross-spectrum ("Gram") buffer:
//   stack_tables : pair list -> row of gramC per pair, 8x8 station tiles, inverse descriptors
//   stack_gram   : gramC (+)= sum_w X_{i,w} conj X_{j,w} for the spectra currently in p->spectra
//   stack_inverse: one inverse transform per pair, gramC -> out
// xcx_cc1d_pairs(XCX_STACK) = tables + forward + gram + inverse of ONE call;
// xcx_cc1d_stack_add / _finish split it so that several calls (days) accumulate in the
// frequency domain and pay for the P inverse transforms once.
// ---------------------------------------------------------------------------
static int stack_tables(xcx_plan* p, const int* pairs, int S, int P, long ld_out) {
  xcx_ctx* ctx = p->ctx;
  int rc;
  unsigned long long key = fnv(pairs, sizeof(int) * 2 * (size_t)P, 1469598103934665603ULL);
  long hdr[3] = {S, P, ld_out};
  key = fnv(hdr, sizeof hdr, key) | 1ULL;
  if (key == p->gram_key && key == p->desc_key && p->hdescs.size() == (size_t)P) return XCX_OK;
  std::vector<int> perm(P);
  for (int i = 0; i < P; ++i) perm[i] = i;
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
    int ta = (pairs[2 * a] >> 3) * 8192 + (pairs[2 * a + 1] >> 3);
    int tb = (pairs[2 * b] >> 3) * 8192 + (pairs[2 * b + 1] >> 3);
    return ta < tb;
  });
  const int nb = (S + kGT - 1) / kGT;
  std::vector<int> pidx((size_t)S * S, -1);
  std::vector<char> used((size_t)nb * nb, 0);
  p->hdescs.resize(P);
  for (int q = 0; q < P; ++q) {
    // processing order q; duplicate pairs share the row of their first occurrence
    const int pi = perm[q];
    const int i = pairs[2 * pi], j = pairs[2 * pi + 1];
    if (pidx[(size_t)i * S + j] < 0) pidx[(size_t)i * S + j] = q;
    used[(size_t)(i / kGT) * nb + j / kGT] = 1;
    const int row = pidx[(size_t)i * S + j];
    p->hdescs[q] = PwDesc{row, row, (long long)pi * ld_out};
  }
  std::vector<int2> tiles;
  for (int bi = 0; bi < nb; ++bi)
    for (int bj = 0; bj < nb; ++bj)
      if (used[(size_t)bi * nb + bj]) tiles.push_back(make_int2(bi * kGT, bj * kGT));
  p->desc_key = 0; p->gram_key = 0;
  if ((rc = upload(ctx, p->gramPidx, pidx.data(), pidx.size() * sizeof(int)))) return rc;
  if ((rc = upload(ctx, p->gramTiles, tiles.data(), tiles.size() * sizeof(int2)))) return rc;
  if ((rc = upload(ctx, p->descs, p->hdescs.data(), (size_t)P * sizeof(PwDesc)))) return rc;
  p->gram_ntiles = (int)tiles.size();
  p->gram_key = key;
  p->desc_key = key;
  return XCX_OK;
}

static int stack_gram(xcx_plan* p, int S, int W, int P, int accumulate, int ldw = 0) {
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  int rc;
  if ((rc = p->gramC.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)P))) return rc;
  if (g.band_hi < g.N && (p->gram_zeroed.ptr != p->gramC.ptr || p->gram_zeroed.gen != p->gramC.gen ||
                          p->gram_zeroed.band_hi != g.band_hi)) {
    // band-limited plan: the Gram kernel skips the out-of-band frequency quads, which therefore hold the
    // zeros of this one fill for the lifetime of the buffer (the pair pass reads its stored zeros from them)
    HIP_TRY(hipMemsetAsync(p->gramC.ptr, 0, p->gramC.bytes, ctx->stream));
    p->gram_zeroed = {p->gramC.ptr, p->gramC.gen, g.band_hi};
  }
  GramArgs ga{};
  ga.g = g;
  ga.X = (const cf*)p->spectra.ptr;
  ga.spec_pitch = p->spec_pitch;
  ga.S = S; ga.W = W;
  ga.nfreq = g.N1 * (g.N2h + 1);
  ga.tiles = (const int2*)p->gramTiles.ptr;
  ga.ntiles = p->gram_ntiles;
  ga.pidx = (const int*)p->gramPidx.ptr;
  ga.C = (cf*)p->gramC.ptr;
  ga.c_pitch = p->spec_pitch;
  ga.accumulate = accumulate;
  ga.ldw = ldw;
  {
    ProfScope ps(p, KID_GRAM, P);
    HIP_TRY(launch_gram(ga, ctx->stream));
  }
  p->launches += 1;
  return XCX_OK;
}

// Gram update over the pending adds: K = acc_pending * acc_W windows per station
static int stack_flush(xcx_plan* p) {
  if (p->acc_pending == 0) return XCX_OK;
  const int rc = stack_gram(p, p->acc_S, p->acc_pending * p->acc_W, p->acc_P, p->acc_grams > 0 ? 1 : 0, p->acc_W * p->acc_cap);
  if (rc) return rc;
  p->acc_grams += 1;
  p->acc_pending = 0;
  return XCX_OK;
}

static int stack_inverse(xcx_plan* p, int P, float* dout, int first = 0, const cf* Csrc = nullptr) {
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  int rc;
  const int B = p->batch;
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  if ((rc = p->T.ensure(ctx, sizeof(cf) * tm_per * B * 2 * (p->dual ? p->nchains : 1)))) return rc;
  const cf* src = Csrc ? Csrc : (const cf*)p->gramC.ptr;
  if (p->flow) return flow_run(p, (size_t)first, (size_t)P, -1, dout, src);
  if (p->dual) return dual_run(p, (size_t)first, (size_t)P, -1, dout, src);
  for (size_t d0 = 0; d0 < (size_t)P; d0 += B)
    if ((rc = pair_batch(p, first + d0, (int)std::min((size_t)B, P - d0), dout, -1, src))) return rc;
  return XCX_OK;
}

static int check_pairs(const int* pairs, int S, int P) {
  for (int i = 0; i < P; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= S || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= S)
      return set_err(XCX_EINVAL, "pair %d = (%d,%d) outside [0,%d)", i, pairs[2 * i], pairs[2 * i + 1], S);
  return XCX_OK;
}

extern "C" int xcx_cc1d_stack_add(xcx_plan* p, const float* traces, int S, int W, long stride_s,
                                  long stride_w, const int* pairs, int P, int on_device) {
  if (!p || !traces || !pairs) return set_err(XCX_EINVAL, "null argument");
  if (S < 1 || W < 1 || P < 1) return set_err(XCX_EINVAL, "S, W, P must be >= 1");
  if (stride_s < 0 || stride_w < 0) return set_err(XCX_EINVAL, "negative stride");
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  if (p->seg) return set_err(XCX_EUNSUPPORTED, "segmented (long-window) plans do not stack");
  if (!(g.flags & XCX_F_STACK)) return set_err(XCX_EINVAL, "xcx_cc1d_stack_add needs an XCX_STACK plan");
  if ((long long)S * W > 0x7fffffffLL / 2) return set_err(XCX_EINVAL, "S*W too large");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  if ((rc = check_pairs(pairs, S, P))) return rc;
  unsigned long long akey = fnv(pairs, sizeof(int) * 2 * (size_t)P, 1469598103934665603ULL);
  akey = fnv(&S, sizeof S, akey) | 1ULL;
  if (p->acc_broken)
    return set_err(XCX_EINVAL, "xcx_cc1d_stack_add: an earlier add of this accumulation failed; call xcx_cc1d_stack_reset first");
  if (p->acc_count > 0 && akey != p->acc_key)
    return set_err(XCX_EINVAL, "xcx_cc1d_stack_add: pair list / station count differ from the accumulation in progress");
  const float* dtr = traces;
  if (!on_device) {
    const size_t span = (size_t)(S - 1) * stride_s + (size_t)(W - 1) * stride_w + g.n;
    if ((rc = p->dtraces.ensure(ctx, span * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(p->dtraces.ptr, traces, span * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    dtr = (const float*)p->dtraces.ptr;
  }
  // tables keyed without ld_out: the finish call supplies it (descriptors are rebuilt there if it differs)
  if ((rc = stack_tables(p, pairs, S, P, p->acc_ld))) return rc;
  // adds of equal W are collected (station-major slots, cap so that the spectra stay below 24 GB) and go
  // through one Gram update; a different W flushes what is pending first
  // (a failure from here on leaves spectra slots and the accumulator in an unknown state: the plan then refuses
  // everything but xcx_cc1d_stack_reset -- a later add would otherwise index slots past the spectra buffer)
  if (p->acc_pending > 0 && W != p->acc_W) { if ((rc = stack_flush(p))) { p->acc_broken = true; return rc; } }
  if (p->acc_pending == 0) {
    static const int want = getenv("XCX_STACK_DAYS") ? std::max(1, atoi(getenv("XCX_STACK_DAYS"))) : 4;
    const size_t per_add = sizeof(cf) * p->spec_pitch * (size_t)S * W;
    p->acc_cap = (int)std::max<size_t>(1, std::min<size_t>((size_t)want, ((size_t)24 << 30) / std::max<size_t>(per_add, 1)));
    p->acc_W = W;
  }
  if ((rc = p->spectra.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)S * W * p->acc_cap))) return rc;
  p->fwd_slot_w = W; p->fwd_slot_ld = W * p->acc_cap;
  rc = forward(p, dtr, stride_s, stride_w, W, 0, 0, S * W, p->acc_pending * W);
  p->fwd_slot_w = p->fwd_slot_ld = 0;
  if (rc) { p->acc_broken = p->acc_count > 0 || p->acc_pending > 0; return rc; }
  p->acc_key = akey; p->acc_S = S; p->acc_P = P;
  p->acc_pending += 1;
  if (p->acc_pending >= p->acc_cap && (rc = stack_flush(p))) { p->acc_broken = true; return rc; }
  if (!on_device) HIP_TRY(hipStreamSynchronize(ctx->stream));     // the staging buffer may be reused
  p->acc_count += 1; p->acc_windows += W;
  return XCX_OK;
}

extern "C" int xcx_cc1d_stack_reset(xcx_plan* p) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  // abandon an accumulation (a failed add, an exception in the driver loop): the plan takes
  // xcx_cc1d_pairs / a new accumulation again; the accumulator is overwritten by the next first flush
  if (p->ctx) (void)hipStreamSynchronize(p->ctx->stream);
  p->acc_count = 0; p->acc_pending = 0; p->acc_grams = 0; p->acc_windows = 0; p->acc_key = 0;
  p->acc_broken = false;
  return XCX_OK;
}

extern "C" int xcx_cc1d_stack_finish(xcx_plan* p, const int* pairs, int P, float* out, long ld_out,
                                     int on_device, long long* windows) {
  if (!p || !pairs || !out) return set_err(XCX_EINVAL, "null argument");
  if (p->acc_broken)
    return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish: an add of this accumulation failed; call xcx_cc1d_stack_reset");
  if (p->acc_count < 1) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish: nothing accumulated");
  xcx_ctx* ctx = p->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  const long L = p->L;
  if (ld_out == 0) ld_out = L;
  if (ld_out < L) return set_err(XCX_EINVAL, "ld_out (%ld) < L (%ld)", ld_out, L);
  int rc;
  const int S = p->acc_S;
  if (P != p->acc_P) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish: P differs from the accumulated pair list");
  if ((rc = check_pairs(pairs, S, P))) return rc;
  unsigned long long akey = fnv(pairs, sizeof(int) * 2 * (size_t)P, 1469598103934665603ULL);
  akey = fnv(&S, sizeof S, akey) | 1ULL;
  if (akey != p->acc_key) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish: pair list differs from the accumulated one");
  p->acc_ld = ld_out;
  if ((rc = stack_tables(p, pairs, S, P, ld_out))) return rc;     // same rows of gramC, out offsets for this pitch
  float* dout = out;
  HostPin pin_out(on_device ? nullptr : out, ((size_t)(P - 1) * (size_t)ld_out + (size_t)L) * sizeof(float));
  if (!on_device) {
    if ((rc = p->dout.ensure(ctx, (size_t)P * (size_t)ld_out * sizeof(float)))) return rc;
    dout = (float*)p->dout.ptr;
  }
  if ((rc = stack_flush(p))) { p->acc_broken = true; return rc; }
  if ((rc = stack_inverse(p, P, dout))) return rc;
  if (windows) *windows = p->acc_windows;
  p->acc_count = 0; p->acc_windows = 0; p->acc_grams = 0;
  if (!on_device) {
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(float), dout, (size_t)ld_out * sizeof(float),
                             (size_t)L * sizeof(float), (size_t)P, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return flow_check(ctx);
  }
  return XCX_OK;
}

// ---------------------------------------------------------------------------
// Time-stack across ranks in the FREQUENCY domain (SURVEY 8e, configs[3]): every rank has accumulated the
// cross-spectra of its station-days (xcx_cc1d_stack_add); instead of P inverse transforms per rank followed by a
// reduce-scatter of P x L float32 stacks (2.90 GB per rank at P = 2016), the ranks reduce-scatter the IN-BAND
// elements of the accumulators (a whitened plan: 40 % of the half spectrum for 0.1-10 Hz at 50 Hz, 1.19 GB) and each
// rank inverse-transforms only the P / nranks rows it receives.  k_band_pack / k_band_unpack move the in-band
// stretches [0, za) and (zb, N1) of every column k2 between the [k2][N1] accumulator rows and dense rows.
// ---------------------------------------------------------------------------
struct BandCol { int za, zb, off; };
template <bool PACK>
__global__ __launch_bounds__(256) void k_band_move(cf* __restrict__ rows, size_t pitch, cf* __restrict__ dense, size_t nib,
                                                   const BandCol* __restrict__ tab, int N1, int nrows_valid) {
  const int k2 = blockIdx.x, r = blockIdx.y;
  const BandCol b = tab[k2];
  const int cnt = (b.zb >= b.za) ? b.za + (N1 - 1 - b.zb) : N1;
  cf* __restrict__ row = rows + (size_t)r * pitch + (size_t)k2 * N1;
  cf* __restrict__ d = dense + (size_t)r * nib + b.off;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int k1 = (b.zb >= b.za && i >= b.za) ? b.zb + 1 + (i - b.za) : i;
    if (PACK) d[i] = (r < nrows_valid) ? row[k1] : make_float2(0.f, 0.f);       // rows beyond P pad the last share
    else row[k1] = d[i];
  }
}

extern "C" int xcx_cc1d_stack_finish_reduce(xcx_plan* p, const int* pairs, int P, float* out, long ld_out,
                                            unsigned char* owned, long long* windows) {
  if (!p || !pairs || !out || !owned) return set_err(XCX_EINVAL, "null argument");
  if (p->acc_broken)
    return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish_reduce: an add of this accumulation failed; call xcx_cc1d_stack_reset");
  if (p->acc_count < 1) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish_reduce: nothing accumulated");
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  HIP_TRY(hipSetDevice(ctx->device));
  // XCX_STACK_REDUCE_FAKE="nranks,rank" (tests on one GPU): the partition of rank `rank` of `nranks` without RCCL --
  // the other ranks contribute nothing, so the rows this rank owns equal those of xcx_cc1d_stack_finish
  int nr = ctx->nranks, rk = ctx->rank;
  bool fake = false;
  if (const char* fk = getenv("XCX_STACK_REDUCE_FAKE")) {
    if (sscanf(fk, "%d,%d", &nr, &rk) != 2 || nr < 1 || rk < 0 || rk >= nr) return set_err(XCX_EINVAL, "XCX_STACK_REDUCE_FAKE=nranks,rank");
    fake = true;
  } else if (!ctx->comm) {
    return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish_reduce: xcx_comm_init() not called on this context");
  }
  const long L = p->L;
  if (ld_out == 0) ld_out = L;
  if (ld_out < L) return set_err(XCX_EINVAL, "ld_out (%ld) < L (%ld)", ld_out, L);
  int rc;
  const int S = p->acc_S;
  if (P != p->acc_P) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish_reduce: P differs from the accumulated pair list");
  if ((rc = check_pairs(pairs, S, P))) return rc;
  unsigned long long akey = fnv(pairs, sizeof(int) * 2 * (size_t)P, 1469598103934665603ULL);
  akey = fnv(&S, sizeof S, akey) | 1ULL;
  if (akey != p->acc_key) return set_err(XCX_EINVAL, "xcx_cc1d_stack_finish_reduce: pair list differs from the accumulated one");
  p->acc_ld = ld_out;
  if ((rc = stack_tables(p, pairs, S, P, ld_out))) return rc;
  if ((rc = stack_flush(p))) { p->acc_broken = true; return rc; }
  // band table of the accumulator columns
  std::vector<BandCol> tab(g.N2h + 1);
  size_t nib = 0;
  for (int k2 = 0; k2 <= g.N2h; ++k2) {
    int za = (g.band_hi >= k2) ? (g.band_hi - k2) / g.N2 + 1 : 0;
    int zb = (g.N - g.band_hi - k2 + g.N2 - 1) / g.N2 - 1;                      // band_zero_stretch (wave_kernels.h)
    if (g.band_hi >= g.N) { za = g.N1; zb = g.N1 - 1; }                         // no band limit: everything
    if (za > g.N1) za = g.N1;
    tab[k2] = BandCol{za, zb, (int)nib};
    nib += (zb >= za) ? (size_t)za + (size_t)(g.N1 - 1 - zb) : (size_t)g.N1;
  }
  const int Pr = (P + nr - 1) / nr, Ppad = Pr * nr;                             // rows per rank, rows with padding
  if ((rc = upload(ctx, p->bandTab, tab.data(), tab.size() * sizeof(BandCol)))) return rc;
  if ((rc = p->gramPack.ensure(ctx, sizeof(cf) * nib * (size_t)Ppad))) return rc;
  if ((rc = p->gramRecv.ensure(ctx, sizeof(cf) * nib * (size_t)Pr))) return rc;
  if ((rc = p->gramOwn.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)Pr))) return rc;
  if (p->gram_own_zeroed != p->gramOwn.ptr || p->gram_own_gen != p->gramOwn.gen || p->gram_own_band != g.band_hi) {
    HIP_TRY(hipMemsetAsync(p->gramOwn.ptr, 0, p->gramOwn.bytes, ctx->stream));   // out-of-band stretches stay zero
    p->gram_own_zeroed = p->gramOwn.ptr; p->gram_own_gen = p->gramOwn.gen; p->gram_own_band = g.band_hi;
  }
  hipLaunchKernelGGL(k_band_move<true>, dim3(g.N2h + 1, Ppad), dim3(256), 0, ctx->stream, (cf*)p->gramC.ptr, p->spec_pitch,
                     (cf*)p->gramPack.ptr, nib, (const BandCol*)p->bandTab.ptr, g.N1, P);
  HIP_TRY(hipGetLastError());
  if (fake) {
    HIP_TRY(hipMemcpyAsync(p->gramRecv.ptr, (const cf*)p->gramPack.ptr + (size_t)rk * Pr * nib, sizeof(cf) * nib * (size_t)Pr,
                           hipMemcpyDeviceToDevice, ctx->stream));
  } else if ((rc = comm_reduce_scatter(ctx, (const float*)p->gramPack.ptr, (float*)p->gramRecv.ptr, 2 * nib * (size_t)Pr))) {
    return rc;
  }
  hipLaunchKernelGGL(k_band_move<false>, dim3(g.N2h + 1, Pr), dim3(256), 0, ctx->stream, (cf*)p->gramOwn.ptr, p->spec_pitch,
                     (cf*)p->gramRecv.ptr, nib, (const BandCol*)p->bandTab.ptr, g.N1, Pr);
  HIP_TRY(hipGetLastError());
  // this rank's rows: accumulator rows [rk Pr, rk Pr + Pr) in processing order; descriptors against gramOwn
  std::vector<PwDesc> own;
  memset(owned, 0, (size_t)P);
  for (int q = 0; q < P; ++q) {
    const PwDesc& d = p->hdescs[q];
    if (d.slot_a < rk * Pr || d.slot_a >= (rk + 1) * Pr) continue;
    own.push_back(PwDesc{d.slot_a - rk * Pr, d.slot_b - rk * Pr, d.out_off});
    owned[d.out_off / ld_out] = 1;
  }
  p->desc_key = 0;                                                               // p->descs no longer holds the full table
  if (!own.empty()) {
    if ((rc = upload(ctx, p->descs, own.data(), own.size() * sizeof(PwDesc)))) return rc;
    if ((rc = stack_inverse(p, (int)own.size(), out, 0, (const cf*)p->gramOwn.ptr))) return rc;
  }
  if (windows) *windows = p->acc_windows;
  p->acc_count = 0; p->acc_windows = 0; p->acc_grams = 0;
  return XCX_OK;
}

extern "C" int xcx_cc1d_pairs(xcx_plan* p, const float* traces, int S, int W,
                              long stride_s, long stride_w, const int* pairs,
                              int P, float* out, long ld_out, int on_device) {
  if (!p || !traces || !pairs || !out) return set_err(XCX_EINVAL, "null argument");
  if (p->acc_count > 0 || p->acc_pending > 0 || p->acc_broken)   // before anything touches the spectra that wait for their Gram update
    return set_err(XCX_EINVAL, "xcx_cc1d_pairs: a xcx_cc1d_stack_add accumulation is in progress on this plan");
  if (S < 1 || W < 1 || P < 1) return set_err(XCX_EINVAL, "S, W, P must be >= 1");
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  HIP_TRY(hipSetDevice(ctx->device));
  const bool stack = g.flags & XCX_F_STACK;
  const long L = p->L;
  if (ld_out == 0) ld_out = L;
  if (ld_out < L) return set_err(XCX_EINVAL, "ld_out (%ld) < L (%ld)", ld_out, L);
  if (stride_s < 0 || stride_w < 0) return set_err(XCX_EINVAL, "negative stride");
  int rc;
  if ((rc = check_pairs(pairs, S, P))) return rc;
  if ((long long)S * W > 0x7fffffffLL / 2) return set_err(XCX_EINVAL, "S*W too large");
  const float* dtr = traces;
  float* dout = out;
  const size_t nrows = stack ? (size_t)P : (size_t)P * W;
  const size_t span_h = (size_t)(S - 1) * stride_s + (size_t)(W - 1) * stride_w + g.n;
  HostPin pin_in(on_device ? nullptr : traces, span_h * sizeof(float));
  HostPin pin_out(on_device ? nullptr : out, ((nrows - 1) * (size_t)ld_out + (size_t)L) * sizeof(float));
  if (!on_device) {
    const size_t span = (size_t)(S - 1) * stride_s + (size_t)(W - 1) * stride_w + g.n;
    if ((rc = p->dtraces.ensure(ctx, span * sizeof(float)))) return rc;
    HIP_TRY(hipMemcpyAsync(p->dtraces.ptr, traces, span * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = p->dout.ensure(ctx, nrows * (size_t)ld_out * sizeof(float)))) return rc;
    dtr = (const float*)p->dtraces.ptr;
    dout = (float*)p->dout.ptr;
    static const bool poison = getenv("XCX_POISON") != nullptr;       // debugging aid: unwritten lags read as NaN
    if (poison) HIP_TRY(hipMemsetAsync(dout, 0xff, nrows * (size_t)ld_out * sizeof(float), ctx->stream));
  }
  if (p->seg) {                                     // long windows: every (pair, window) block-wise
    std::vector<const float*> xs((size_t)P * W), ys((size_t)P * W);
    for (int q = 0; q < P; ++q)
      for (int w = 0; w < W; ++w) {
        xs[(size_t)q * W + w] = dtr + (size_t)pairs[2 * q] * stride_s + (size_t)w * stride_w;
        ys[(size_t)q * W + w] = dtr + (size_t)pairs[2 * q + 1] * stride_s + (size_t)w * stride_w;
      }
    if ((rc = seg_run(p, xs.data(), ys.data(), P * W, dout, ld_out))) return rc;
    if (!on_device) {
      HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(float), dout, (size_t)ld_out * sizeof(float),
                               (size_t)L * sizeof(float), nrows, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return XCX_OK;
  }
  // processing order: pairs grouped by 8x8 station tiles so that one launch
  // batch touches few distinct spectra (L2 / Infinity-Cache reuse)
  unsigned long long key = fnv(pairs, sizeof(int) * 2 * (size_t)P, 1469598103934665603ULL);
  // spectra of all windows resident at once when they fit in 24 GB (1.5 MB each)
  const bool allw = !stack && W > 1 && sizeof(cf) * p->spec_pitch * (size_t)S * W <= ((size_t)24 << 30);
  int hdr[5] = {S, W, P, (int)stack, (int)allw};
  key = fnv(hdr, sizeof hdr, key);
  key = fnv(&ld_out, sizeof ld_out, key) & ~1ULL;      // even: never equal to a stack_tables() key (odd)
  const size_t ndesc = stack ? (size_t)P : (size_t)P * W;
  // Stacks of TWO windows (the chirp convolutions of fftn.hip) sum the two products in the pair column pass
  // itself (MODE 1: four column loads per pair, no loop): building, writing and re-reading a P x N/2
  // cross-spectrum buffer only pays once several windows share it (k_gram_mfma was 60 % of xcx_rfft).
  const bool via_gram = stack && W >= 3;             // descriptors come from stack_tables()
  if (!via_gram && (key != p->desc_key || p->hdescs.size() != ndesc)) {
    std::vector<int> perm(P);
    // processing order.  Dual pipelines: launch batches = station groups (pair_order.h: few distinct
    // stations per launch, so few first-touch spectrum fetches); otherwise 8 x 8 station tiles.
    static const bool grouped = !(getenv("XCX_GROUPS") && atoi(getenv("XCX_GROUPS")) == 0);
    p->bounds.clear();
    if (grouped && p->dual) {          // the flow kernel takes the order (its batches are fixed-size cuts of it)
      pair_groups(pairs, S, P, p->batch, perm, p->bounds);
    } else {
      for (int i = 0; i < P; ++i) perm[i] = i;
      static const int tile = getenv("XCX_TILE") ? std::max(1, atoi(getenv("XCX_TILE"))) : 8;
      std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) {
        int ta = (pairs[2 * a] / tile) * 8192 + (pairs[2 * a + 1] / tile);
        int tb = (pairs[2 * b] / tile) * 8192 + (pairs[2 * b + 1] / tile);
        return ta < tb;
      });
    }
    p->hdescs.resize(ndesc);
    if (stack) {
      for (int q = 0; q < P; ++q) {
        int pi = perm[q];
        p->hdescs[q] = PwDesc{pairs[2 * pi] * W, pairs[2 * pi + 1] * W, (long long)pi * ld_out};
      }
    } else {
      for (int w = 0; w < W; ++w)
        for (int q = 0; q < P; ++q) {
          int pi = perm[q];
          p->hdescs[(size_t)w * P + q] =
              allw ? PwDesc{pairs[2 * pi] * W + w, pairs[2 * pi + 1] * W + w, ((long long)pi * W + w) * ld_out}
                   : PwDesc{pairs[2 * pi], pairs[2 * pi + 1], ((long long)pi * W + w) * ld_out};
        }
    }
    if ((rc = upload(ctx, p->descs, p->hdescs.data(), ndesc * sizeof(PwDesc)))) return rc;
    p->desc_key = key;
    // the Gram path rewrites slot_a/slot_b of these descriptors (rows of gramC): a rebuilt
    // descriptor array must be remapped again even if the Gram tables themselves are current
    p->gram_key = 0;
  }
  const int B = p->batch;
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  if ((rc = p->T.ensure(ctx, sizeof(cf) * tm_per * B * 2 * (p->dual ? p->nchains : 1)))) return rc;
  // spectra prepared ahead by xcx_cc1d_prepare for exactly this call?
  const bool prepared = p->prep_valid && allw && on_device && p->prep_traces == traces && p->prep_S == S &&
                        p->prep_W == W && p->prep_ss == stride_s && p->prep_sw == stride_w;
  if (p->prep_valid && !prepared) {                   // stale preparation: let it finish, drop it
    HIP_TRY(hipStreamWaitEvent(ctx->stream, p->prep_done, 0));
    p->prep_valid = false;
  }
  if (stack) {
    if ((rc = p->spectra.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)S * W))) return rc;
    if ((rc = forward(p, dtr, stride_s, stride_w, W, 0, 0, S * W, 0))) return rc;
    const cf* src = nullptr;
    int ns = W;
    if (via_gram) {
      // cgemm-style stack: G_k = A_k A_k^H per frequency, then one inverse per pair
      if ((rc = stack_tables(p, pairs, S, P, ld_out))) return rc;
      if ((rc = stack_gram(p, S, W, P, 0))) return rc;
      src = (const cf*)p->gramC.ptr;
      ns = -1;
    }
    if (p->flow) {
      if ((rc = flow_run(p, 0, (size_t)P, ns, dout, src))) return rc;
    } else if (p->dual) {
      if ((rc = dual_run(p, 0, (size_t)P, ns, dout, src, via_gram ? nullptr : &p->bounds))) return rc;
    } else {
      for (size_t d0 = 0; d0 < (size_t)P; d0 += B)
        if ((rc = pair_batch(p, d0, (int)std::min((size_t)B, P - d0), dout, ns, src))) return rc;
    }
  } else if (allw) {
    // every window's station spectra first (few large launches), then the pairs
    if (prepared) {
      std::swap(p->spectra, p->spectra_next);
      HIP_TRY(hipStreamWaitEvent(ctx->stream, p->prep_done, 0));
      p->prep_valid = false;
    } else {
      if ((rc = p->spectra.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)S * W))) return rc;
      if ((rc = forward(p, dtr, stride_s, stride_w, W, 0, 0, S * W, 0))) return rc;
    }
    if (p->flow) {
      if ((rc = flow_run(p, 0, ndesc, 1, dout))) return rc;
    } else if (p->dual) {
      if ((rc = dual_run(p, 0, ndesc, 1, dout, nullptr, &p->bounds))) return rc;
    } else {
      for (size_t d0 = 0; d0 < ndesc; d0 += B)
        if ((rc = pair_batch(p, d0, (int)std::min((size_t)B, ndesc - d0), dout, 1))) return rc;
    }
  } else {
    if ((rc = p->spectra.ensure(ctx, sizeof(cf) * p->spec_pitch * (size_t)S))) return rc;
    for (int w = 0; w < W; ++w) {
      // one window of every station: sw = s*1 + 0 with window index w
      if ((rc = forward(p, dtr, stride_s, stride_w, 1, w, 0, S, 0))) return rc;
      if (p->flow) {
        if ((rc = flow_run(p, (size_t)w * P, (size_t)P, 1, dout))) return rc;
      } else if (p->dual) {
        if ((rc = dual_run(p, (size_t)w * P, (size_t)P, 1, dout, nullptr, &p->bounds))) return rc;
      } else {
        for (size_t q0 = 0; q0 < (size_t)P; q0 += B)
          if ((rc = pair_batch(p, (size_t)w * P + q0, (int)std::min((size_t)B, P - q0), dout, 1))) return rc;
      }
    }
  }
  if (p->pairs_done[0]) {                             // a later xcx_cc1d_prepare must not overwrite these spectra early
    HIP_TRY(hipEventRecord(p->pairs_done[p->pairs_count & 1], ctx->stream));
    p->pairs_count += 1;
  }
  if (!on_device) {
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld_out * sizeof(float), dout, (size_t)ld_out * sizeof(float),
                             (size_t)L * sizeof(float), nrows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return flow_check(ctx);
  }
  return XCX_OK;
}

extern "C" int xcx_cc1d_prepare(xcx_plan* p, const float* traces, int S, int W, long stride_s, long stride_w) {
  if (!p || !traces) return set_err(XCX_EINVAL, "null argument");
  if (S < 1 || W < 1 || stride_s < 0 || stride_w < 0) return set_err(XCX_EINVAL, "bad shape");
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  if (p->seg) return set_err(XCX_EUNSUPPORTED, "xcx_cc1d_prepare does not apply to segmented (long-window) plans");
  if (g.flags & XCX_F_STACK) return set_err(XCX_EUNSUPPORTED, "xcx_cc1d_prepare does not apply to stacking plans");
  if ((long long)S * W > 0x7fffffffLL / 2) return set_err(XCX_EINVAL, "S*W too large");
  const size_t bytes = sizeof(cf) * p->spec_pitch * (size_t)S * W;
  if (W < 2 || bytes > ((size_t)24 << 30))
    return set_err(XCX_EUNSUPPORTED, "xcx_cc1d_prepare needs 2 <= W and all spectra resident (<= 24 GB)");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  if (!p->prep_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&p->prep_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->prep_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->pairs_done[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->pairs_done[1], hipEventDisableTiming));
  }
  if ((rc = p->spectra_next.ensure(ctx, bytes))) return rc;
  // the buffer being filled was last read by the pair phase BEFORE the most recent one (the
  // most recent one reads p->spectra and is what this preparation overlaps with): wait for
  // that one on the GPU, not on the host
  if (p->pairs_count >= 2) HIP_TRY(hipStreamWaitEvent(p->prep_stream, p->pairs_done[p->pairs_count & 1], 0));
  // forward() launches on the context's main stream and writes p->spectra: redirect both
  hipStream_t main_stream = ctx->stream;
  ctx->stream = p->prep_stream;
  std::swap(p->spectra, p->spectra_next);
  rc = forward(p, traces, stride_s, stride_w, W, 0, 0, S * W, 0);
  std::swap(p->spectra, p->spectra_next);
  ctx->stream = main_stream;
  if (rc) return rc;
  HIP_TRY(hipEventRecord(p->prep_done, p->prep_stream));
  p->prep_valid = true;
  p->prep_traces = traces; p->prep_S = S; p->prep_W = W; p->prep_ss = stride_s; p->prep_sw = stride_w;
  return XCX_OK;
}

extern "C" int xcx_cc1d_prepare_sync(xcx_plan* p) {
  if (!p) return set_err(XCX_EINVAL, "null plan");
  if (p->prep_done) HIP_TRY(hipEventSynchronize(p->prep_done));
  return XCX_OK;
}

extern "C" int xcx_cc1d(xcx_plan* p, const float* x, const float* y, int batch,
                        float* out, int on_device) {
  if (!p || !x || !y || !out) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1) return set_err(XCX_EINVAL, "batch must be >= 1");
  xcx_ctx* ctx = p->ctx;
  const Geom& g = p->g;
  if (g.flags & XCX_F_STACK) return set_err(XCX_EINVAL, "xcx_cc1d does not stack");
  HIP_TRY(hipSetDevice(ctx->device));
  const long L = p->L;
  const int n = g.n;
  int rc;
  if (p->seg) {                                     // long windows: block-wise (see seg_run)
    HostPin pin_x(on_device ? nullptr : x, (size_t)batch * n * sizeof(float));
    HostPin pin_y(on_device ? nullptr : y, (size_t)batch * n * sizeof(float));
    HostPin pin_o(on_device ? nullptr : out, (size_t)batch * L * sizeof(float));
    const float *dx = x, *dy = y;
    float* dout = out;
    if (!on_device) {
      const size_t inb = (size_t)batch * n * sizeof(float);
      if ((rc = p->dtraces.ensure(ctx, 2 * inb))) return rc;
      if ((rc = p->dout.ensure(ctx, (size_t)batch * L * sizeof(float)))) return rc;
      HIP_TRY(hipMemcpyAsync(p->dtraces.ptr, x, inb, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipMemcpyAsync((char*)p->dtraces.ptr + inb, y, inb, hipMemcpyHostToDevice, ctx->stream));
      dx = (const float*)p->dtraces.ptr; dy = dx + (size_t)batch * n; dout = (float*)p->dout.ptr;
    }
    std::vector<const float*> xs(batch), ys(batch);
    for (int i = 0; i < batch; ++i) { xs[i] = dx + (size_t)i * n; ys[i] = dy + (size_t)i * n; }
    if ((rc = seg_run(p, xs.data(), ys.data(), batch, dout, L))) return rc;
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(out, dout, (size_t)batch * L * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return XCX_OK;
  }
  const int kRows = 256;   // rows per chunk
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  if ((rc = p->spectra.ensure(ctx, sizeof(cf) * p->spec_pitch * 2 * (size_t)std::min(batch, kRows)))) return rc;
  if ((rc = p->T.ensure(ctx, sizeof(cf) * tm_per * p->batch * 2 * (p->dual ? p->nchains : 1)))) return rc;
  p->desc_key = 0;
  const float *dx = x, *dy = y;
  float* dout = out;
  HostPin pin_x(on_device ? nullptr : x, (size_t)batch * n * sizeof(float));
  HostPin pin_y(on_device ? nullptr : y, (size_t)batch * n * sizeof(float));
  HostPin pin_o(on_device ? nullptr : out, (size_t)batch * L * sizeof(float));
  if (!on_device) {
    const size_t inb = (size_t)batch * n * sizeof(float);
    if ((rc = p->dtraces.ensure(ctx, 2 * inb))) return rc;
    HIP_TRY(hipMemcpyAsync(p->dtraces.ptr, x, inb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync((char*)p->dtraces.ptr + inb, y, inb, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = p->dout.ensure(ctx, (size_t)batch * L * sizeof(float)))) return rc;
    dx = (const float*)p->dtraces.ptr;
    dy = dx + (size_t)batch * n;
    dout = (float*)p->dout.ptr;
  }
  // descriptors of every chunk in one upload (row i of a chunk: spectrum slots i and cnt + i), so that the chunks
  // queue back to back without a host synchronisation between them
  {
    std::vector<PwDesc> d((size_t)batch);
    for (int r0 = 0; r0 < batch; r0 += kRows) {
      const int cnt = std::min(kRows, batch - r0);
      for (int i = 0; i < cnt; ++i) d[(size_t)r0 + i] = PwDesc{i, cnt + i, (long long)(r0 + i) * L};
    }
    if ((rc = upload(ctx, p->descs, d.data(), sizeof(PwDesc) * (size_t)batch))) return rc;
  }
  for (int r0 = 0; r0 < batch; r0 += kRows) {
    const int cnt = std::min(kRows, batch - r0);
    if ((rc = forward(p, dx + (size_t)r0 * n, n, 0, 1, 0, 0, cnt, 0))) return rc;
    if ((rc = forward(p, dy + (size_t)r0 * n, n, 0, 1, 0, 0, cnt, cnt))) return rc;
    // the rows of a chunk through the dual pipelines (row pass of one batch overlapped with the column pass of the
    // next, two launch chains) where the size has them; batch by batch, columns then rows, otherwise
    if (p->dual) {
      if ((rc = dual_run(p, (size_t)r0, (size_t)cnt, 1, dout))) return rc;
    } else {
      for (int q0 = 0; q0 < cnt; q0 += p->batch)
        if ((rc = pair_batch(p, (size_t)r0 + q0, std::min(p->batch, cnt - q0), dout, 1))) return rc;
    }
  }
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)batch * L * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return XCX_OK;
}

// ---------------------------------------------------------------------------
// internals shared with timeshift.hip
// ---------------------------------------------------------------------------
int xcx_internal_forward(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                         int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid) {
  if (n_valid > 0 && (n_valid > p->g.n || (p->g.flags & (XCX_F_DEMEAN | XCX_F_NORMALIZE | XCX_F_ONEBIT))))
    return set_err(XCX_EINVAL, "short windows need a plan without demean / normalise / one-bit");
  return forward(p, traces, stride_s, stride_w, W, w0, sw_begin, sw_end, slot0, n_valid);
}
// ... for seg_plan.hip: the same without the argument check, the inverse of a cross-spectrum buffer, table uploads
int xcx_internal_forward_raw(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                             int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid) {
  return forward(p, traces, stride_s, stride_w, W, w0, sw_begin, sw_end, slot0, n_valid);
}
int xcx_internal_stack_inverse(xcx_plan* p, int P, float* dout, int first, const cf* Csrc) {
  return stack_inverse(p, P, dout, first, Csrc);
}
int xcx_internal_upload(xcx_ctx* ctx, DevBuf& b, const void* src, size_t bytes) { return upload(ctx, b, src, bytes); }
int xcx_internal_pairs(xcx_plan* p, const PwDesc* hdesc, int count, float* out) {
  xcx_ctx* ctx = p->ctx;
  int rc;
  const Geom& g = p->g;
  const size_t tm_per = (size_t)g.N1 * (g.N2h + 1);
  if ((rc = p->T.ensure(ctx, sizeof(cf) * tm_per * p->batch * 2 * (p->dual ? p->nchains : 1)))) return rc;
  if ((rc = p->descs.ensure(ctx, sizeof(PwDesc) * (size_t)count))) return rc;
  p->desc_key = 0;
  HIP_TRY(hipMemcpyAsync(p->descs.ptr, hdesc, sizeof(PwDesc) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (p->dual) return dual_run(p, 0, (size_t)count, 1, out);
  for (int q0 = 0; q0 < count; q0 += p->batch)
    if ((rc = pair_batch(p, q0, std::min(p->batch, count - q0), out, 1))) return rc;
  return XCX_OK;
}
int xcx_internal_spectra(xcx_plan* p, size_t slots) {
  return p->spectra.ensure(p->ctx, sizeof(cf) * p->spec_pitch * slots);
}
xcx_ctx* xcx_internal_ctx(xcx_plan* p) { return p->ctx; }
// long windows (block decomposition): device row pointers x[i], y[i] -> out + i*ld
bool xcx_internal_segmented(xcx_plan* p) { return p->seg != nullptr; }
// circular convolution of rows xs (n samples) with two-period kernels ks (2n - 1 samples): see seg_run(conv_len)
int xcx_internal_seg_conv(xcx_plan* p, const float* const* xs, const float* const* ks, int count, float* out, long ld) {
  return seg_run(p, xs, ks, count, out, ld, 2L * p->g.n - 1);
}
// ... with the rows in twos summed in the frequency domain: out row t = conv(row 2t) + signs[t] conv(row 2t + 1), count / 2 rows
int xcx_internal_seg_conv_sum(xcx_plan* p, const float* const* xs, const float* const* ks, int count, const int* signs,
                              float* out, long ld) {
  if (count < 2 || (count & 1) || !signs) return set_err(XCX_EINVAL, "xcx_internal_seg_conv_sum: rows come in twos");
  return seg_run(p, xs, ks, count, out, ld, 2L * p->g.n - 1, signs);
}

#ifdef XCX_STAMPS
// diagnostic build only (not part of include/xcx.h): copy the recorded phase stamps to the host
extern "C" long xcx_debug_stamps(xcx_plan* p, unsigned long long* host, long max_waves) {
  if (!p || !p->stampbuf.ptr) return 0;
  (void)hipDeviceSynchronize();
  const long n = (long)std::min<size_t>(p->stamp_waves, (size_t)max_waves);
  if (hipMemcpy(host, p->stampbuf.ptr, (size_t)n * kNStamp * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}
#endif

