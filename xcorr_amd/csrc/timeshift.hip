// timeshift.hip -- fractional time shift of [batch][size] float32 rows with the
// reference's semantics (xcorr/signal/timeshift.py:90-96 with
// xcorr/signal/fft.py:357-358, 475-479):
//     n = size - (size & 1);  X = rfft(x[:n]);  X *= exp(-2 pi i delay f);
//     y = irfft(X, n)        with f = linspace(0, fN, n/2+1), fN = round(n/2 / ((n-1) delta))
// i.e. a circular convolution of period n with the Dirichlet kernel
//     h[m] = (1/n) [ sin((n-1) pi u)/sin(pi u) + cos(pi n alpha) (-1)^m ],  u = m/n - alpha,
//     alpha = delay * fN / (n/2)   (cycles per frequency bin),
// which needs no length-n FFT (n = 359 998 = 2 * 179 999 has a prime factor): the
// kernel is generated analytically in float64 on the device and the circular
// convolution runs on the linear cross-correlation engine at a fast length N >= 2n-1:
// the row is correlated against TWO periods of the kernel (2n-1 samples), so the n retained
// lags ARE the circular convolution and the pair pass writes y directly (round 3; before: one
// period, all 2n-1 lags written, then a fold pass y[m] = R[m] + R[m-n] -- kept for rows longer
// than one transform, which go through the block decomposition).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/xcx.h"
#include "cc_kernels.h"
#include "xcx_internal.h"

using namespace xcx;

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess)                                                 \
      return set_err(e_ == hipErrorOutOfMemory ? XCX_ENOMEM : XCX_EHIP,   \
                     "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                     __FILE__, __LINE__);                                 \
  } while (0)

int xcx_internal_forward(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                         int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid);
int xcx_internal_pairs(xcx_plan* p, const PwDesc* hdesc, int count, float* out);
int xcx_internal_spectra(xcx_plan* p, size_t slots);
bool xcx_internal_segmented(xcx_plan* p);
int xcx_internal_seg_conv(xcx_plan* p, const float* const* xs, const float* const* ks, int count, float* out, long ld);

namespace {

// hr[r][t] = h_r[(-t) mod n]: the correlation engine computes
// R[lag] = sum_t x[t+lag] hr[t], so R folded over the period is x (*) h
// With n u = m - n alpha:  sin((n-1) pi u) = sin(n pi u) cos(pi u) - cos(n pi u) sin(pi u) and
// sin(n pi u) = -(-1)^m sin(pi n alpha), cos(n pi u) = (-1)^m cos(pi n alpha), so the two
// cosine terms cancel and   h[m] = -(-1)^m sin(pi n alpha) cot(pi u) / n   -- one sincospi
// per sample (float64: u is a small difference near the peak) instead of three circular
// functions; the removable singularity u -> 0 takes the limit (n-1 + cos(pi n alpha)(-1)^m)/n.
// Odd n (only reachable through the fft/ifft path, which keeps every sample): all n frequencies
// |k| <= (n-1)/2 exist and  h[m] = sin(n pi u) / (n sin(pi u)) = -(-1)^m sin(pi n alpha) / (n sin(pi u)),
// limit 1 at u -> 0.
// Each thread produces kDK samples t, 64 apart: the first from sincospi(u) as written above, the others by
// rotating (sin pi u, cos pi u) by the constant step 64 pi/n (float64, <= kDK - 1 roundings) -- one sincospi per
// kDK samples instead of one per sample (the kernel was 1.7 us of a full-lag row's 9.7).  A rotated sine that
// comes out below 1e-6 is re-evaluated exactly: next to the kernel's peak the quotient would amplify the
// rounding of the recurrence.
constexpr int kDK = 8;
__global__ void k_dirichlet(float* __restrict__ hr, long n, long rows, const double* __restrict__ alpha) {
  // grid (ceil(n / (256 kDK)), rows): no 64-bit division per sample
  // a wave owns 64 kDK consecutive samples; lane l takes t = first + l + 64 k, so every store instruction of the
  // wave writes 64 consecutive floats and a lane's samples are 64 apart: rotate by -64 pi/n per step
  const long r = blockIdx.y;
  const long t0 = ((long)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * kDK + (threadIdx.x & 63);
  if (t0 >= n || r >= rows) return;
  const double al = alpha[r];
  const double inv_n = 1.0 / (double)n;
  double sa, ca;
  sincospi(fmod((double)n * al, 2.0), &sa, &ca);    // row constants
  double sd, cd;
  sincospi(64.0 * inv_n, &sd, &cd);                 // 64 samples further in t = 64 samples back in m
  // state: (sin pi u, cos pi u) of the UNREDUCED u = m/n - alpha.  sincospi is evaluated on the reduced argument
  // u - rint(u) (a small difference next to the peak) and both values get the sign (-1)^rint(u) back; with the true
  // sine the odd-n kernel needs no separate reduction sign: sin(n pi u) / sin(pi u) is even in the reduction.
  auto exact = [&](long m, double& su, double& cu) {
    double u = (double)m * inv_n - al;
    const double uw = rint(u);
    u -= uw;
    sincospi(u, &su, &cu);
    if ((long)uw & 1) { su = -su; cu = -cu; }
  };
  double su = 0.0, cu = 1.0;
#pragma unroll
  for (int k = 0; k < kDK; ++k) {
    const long t = t0 + 64 * k;
    if (t >= n) break;
    const long m = (t == 0) ? 0 : n - t;
    if (k == 0 || t == 64) exact(m, su, cu);        // t = 0 -> m = 0, t = 64 -> m = n - 64: not one step apart
    else {
      const double s2 = su * cd - cu * sd, c2 = cu * cd + su * sd;
      su = s2; cu = c2;
      if (fabs(su) < 1e-6) exact(m, su, cu);
    }
    const double sgn = (m & 1) ? -1.0 : 1.0;
    double h;
    if (n & 1) {
      if (fabs(su) < 1e-12) h = 1.0;
      else h = -sgn * sa / su * inv_n;
    } else {
      if (fabs(su) < 1e-12) h = ((double)(n - 1) + ca * sgn) * inv_n;
      else h = -sgn * sa * (cu / su) * inv_n;
    }
    // two periods, g[j] = h[(n-1-j) mod n], j < 2n-1: with R[lag] = sum_j x[j+lag] g[j] (x zero beyond n samples)
    // the lags -(n-1) .. 0 are y[m], m = n-1+lag.  h[m] = this sample's value (m = (-t) mod n).
    float* g = hr + r * (2 * n - 1);
    g[n - 1 - m] = (float)h;
    if (m >= 1) g[2 * n - 1 - m] = (float)h;
  }
}

// y[r][m] *= cos(2 pi m / n)   (odd-length fft / ifft path of the reference, see xcx_timeshift_fft)
__global__ void k_modulate(float* __restrict__ y, long n, long rows) {
  const long r = blockIdx.y;
  const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n || r >= rows) return;
  y[r * n + m] = (float)((double)y[r * n + m] * cospi(2.0 * (double)m / (double)n));
}

}  // namespace

// rows [batch][size] -> [batch][n] (n <= size), alpha[r] = shift of row r in cycles per frequency bin
static int timeshift_rows(xcx_ctx* ctx, const float* x, float* y, long batch, long size, long n,
                          const std::vector<double>& alpha, int modulate, int on_device) {
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  if (!ctx->ts_plan || ctx->ts_n != n) {
    if (ctx->ts_plan) { xcx_plan_destroy(ctx->ts_plan); ctx->ts_plan = nullptr; }
    if (n > XCX_MAX_ROW_SAMPLES) return set_err(XCX_EUNSUPPORTED, "row too long");
    // direct form: operands of 2n-1 samples (the row is zero beyond its first n), lags -(n-1) .. 0 retained;
    // such a plan is alias-free at N >= 2n-1 (kPlanShortOperand, xcx_api.hip)
    const int np = (int)(2 * n - 1);
    ctx->ts_direct = false;
    if (np <= kMaxTransformN) {
      if ((rc = xcx_plan_cc1d(ctx, np, 0, (int)(n - 1), np, kPlanShortOperand, &ctx->ts_plan))) return rc;
      ctx->ts_direct = !xcx_internal_segmented(ctx->ts_plan);
      if (!ctx->ts_direct) { xcx_plan_destroy(ctx->ts_plan); ctx->ts_plan = nullptr; }
    }
    // longer than one transform: the block decomposition (xcx_api.hip: seg_run) in its convolution mode
    if (!ctx->ts_direct) {
      if ((rc = xcx_plan_cc1d(ctx, (int)n, 0, 0, 0, 0, &ctx->ts_plan))) return rc;
      if (!xcx_internal_segmented(ctx->ts_plan)) {       // cannot happen: 2n - 1 exceeds every single transform
        xcx_plan_destroy(ctx->ts_plan); ctx->ts_plan = nullptr;
        return set_err(XCX_EUNSUPPORTED, "time shift of %ld-sample rows: no plan", n);
      }
    }
    ctx->ts_n = n;
  }
  xcx_plan* plan = ctx->ts_plan;
  const bool direct = ctx->ts_direct;
  const long L = 2 * n - 1;
  const int kRows = n > 400000 ? 64 : 256;            // rows per launch group: bounded workspace for long rows; 128 -> 256 measured -5 % (launch tails)
  DevBuf &hr = ctx->scratch[1], &dal = ctx->scratch[3], &dx = ctx->scratch[4],
         &dy = ctx->scratch[5];
  const int chunk = (int)std::min<long>(kRows, batch);
  const bool seg = !direct;                          // rows longer than one transform: block decomposition
  // kernels of two periods (2n - 1 samples): the n retained lags are the circular convolution itself
  if ((rc = hr.ensure(ctx, sizeof(float) * L * chunk))) return rc;
  if ((rc = dal.ensure(ctx, sizeof(double) * batch))) return rc;
  HIP_TRY(hipMemcpyAsync(dal.ptr, alpha.data(), sizeof(double) * batch, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const float* xd = x;
  float* yd = y;
  HostPin pin_x(on_device ? nullptr : x, sizeof(float) * (size_t)size * batch);      // host rows: DMA at link speed
  HostPin pin_y(on_device ? nullptr : y, sizeof(float) * (size_t)n * batch);
  if (!on_device) {
    if ((rc = dx.ensure(ctx, sizeof(float) * size * batch))) return rc;
    if ((rc = dy.ensure(ctx, sizeof(float) * n * batch))) return rc;
    HIP_TRY(hipMemcpyAsync(dx.ptr, x, sizeof(float) * size * batch, hipMemcpyHostToDevice, ctx->stream));
    xd = (const float*)dx.ptr;
    yd = (float*)dy.ptr;
  }
  if (!seg && (rc = xcx_internal_spectra(plan, 2 * (size_t)chunk))) return rc;
  std::vector<PwDesc> d(chunk);
  std::vector<const float*> xs(chunk), ys(chunk);
  // Rows of one launch group that share a shift share ONE Dirichlet kernel: it is generated and transformed once
  // (the reference's delay is -(pair_offset + time_offset), core/postprocess.py:118 -- constant over long stretches
  // of a pair's windows, and a scalar `delay` is one value for every row).  ukern[i] = kernel slot of row i.
  std::vector<int> ukern(chunk);
  std::vector<double> ualpha;
  for (long r0 = 0; r0 < batch; r0 += chunk) {
    const int cnt = (int)std::min<long>(chunk, batch - r0);
    ualpha.clear();
    for (int i = 0; i < cnt; ++i) {
      const double al = alpha[r0 + i];
      int u = -1;
      // groups hold at most 256 rows and real delay sets are tiny: a linear scan of the distinct values is enough
      for (int k = (int)ualpha.size() - 1; k >= 0; --k) if (ualpha[k] == al) { u = k; break; }
      if (u < 0) { u = (int)ualpha.size(); ualpha.push_back(al); }
      ukern[i] = u;
    }
    const int nk = (int)ualpha.size();
    if (nk < cnt) {       // fewer kernels than rows: the distinct shifts replace this group's slice of the device array
      HIP_TRY(hipMemcpyAsync((double*)dal.ptr + r0, ualpha.data(), sizeof(double) * nk, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));      // `ualpha` is reused by the next group
    }
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)cnt);
    const dim3 kgrid((unsigned)((n + 256 * kDK - 1) / (256 * kDK)), (unsigned)nk);
    hipLaunchKernelGGL(k_dirichlet, kgrid, dim3(256), 0, ctx->stream, (float*)hr.ptr, n, (long)nk,
                       (const double*)dal.ptr + r0);
    HIP_TRY(hipGetLastError());
    if (direct) {
      if ((rc = xcx_internal_forward(plan, xd + r0 * size, size, 0, 1, 0, 0, cnt, 0, (int)n))) return rc;
      if ((rc = xcx_internal_forward(plan, (const float*)hr.ptr, L, 0, 1, 0, 0, nk, cnt, 0))) return rc;
      for (int i = 0; i < cnt; ++i) d[i] = PwDesc{i, cnt + ukern[i], (long long)i * n};
      if ((rc = xcx_internal_pairs(plan, d.data(), cnt, yd + r0 * n))) return rc;
      if (modulate) {
        hipLaunchKernelGGL(k_modulate, grid, dim3(256), 0, ctx->stream, yd + r0 * n, n, (long)cnt);
        HIP_TRY(hipGetLastError());
      }
      continue;
    }
    // block-decomposed convolution with the two-period kernel in the segment role: one pass, the n values written
    // in place (rows that share a delay share the kernel's segment transforms; before round 4: both lag signs
    // against one period, 2n - 1 lags, then a fold)
    for (int i = 0; i < cnt; ++i) { xs[i] = xd + (r0 + i) * size; ys[i] = (const float*)hr.ptr + (long)ukern[i] * L; }
    if ((rc = xcx_internal_seg_conv(plan, xs.data(), ys.data(), cnt, yd + r0 * n, n))) return rc;
    if (modulate) {
      hipLaunchKernelGGL(k_modulate, grid, dim3(256), 0, ctx->stream, yd + r0 * n, n, (long)cnt);
      HIP_TRY(hipGetLastError());
    }
  }
  if (!on_device) {
    HIP_TRY(hipMemcpyAsync(y, yd, sizeof(float) * n * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return XCX_OK;
}

extern "C" int xcx_timeshift(xcx_ctx* ctx, const float* x, float* y, long batch, long size,
                             const double* delay, double sampling_rate, int on_device) {
  if (!ctx || !x || !y || !delay) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || size < 2) return set_err(XCX_EINVAL, "batch >= 1 and size >= 2 required");
  if (!(sampling_rate > 0)) return set_err(XCX_EINVAL, "sampling_rate must be positive");
  const long n = size - (size & 1);
  // alpha = delay * fN / (n/2), fN = round(n/2 / ((n-1) delta))   (fft.py:358)
  const double delta = 1.0 / sampling_rate;
  const double fN = std::round((double)(n / 2) / ((double)(n - 1) * delta));
  std::vector<double> alpha(batch);
  for (long r = 0; r < batch; ++r) alpha[r] = delay[r] * fN / (double)(n / 2);
  return timeshift_rows(ctx, x, y, batch, size, n, alpha, 0, on_device);
}

// fast=False of the reference (signal/timeshift.py:90-96): X = fftshift(fft(x)) on the axis
// f = fftshift(fftfreq(n, round(delta, 10)))  (fft.py:104-113), phase ramp, then
// real(ifft(fftshift(X)))  (fft.py:239-241).  Every sample is kept.  For even n fftshift undoes
// itself and the result is the period-n shift; for odd n the second fftshift is NOT the inverse
// of the first (the spectrum ends up rolled by one bin), so the reference returns the shifted row
// times cos(2 pi m / n) -- reproduced as it is.
extern "C" int xcx_timeshift_fft(xcx_ctx* ctx, const float* x, float* y, long batch, long size,
                                 const double* delay, double delta, int on_device) {
  if (!ctx || !x || !y || !delay) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || size < 2) return set_err(XCX_EINVAL, "batch >= 1 and size >= 2 required");
  if (!(delta > 0)) return set_err(XCX_EINVAL, "delta must be positive");
  const double dr = std::nearbyint(delta * 1e10) / 1e10;          // numpy.round(delta, decimals=10)
  if (!(dr > 0)) return set_err(XCX_EINVAL, "delta rounds to zero at 1e-10");
  std::vector<double> alpha(batch);
  for (long r = 0; r < batch; ++r) alpha[r] = delay[r] / ((double)size * dr);
  return timeshift_rows(ctx, x, y, batch, size, size, alpha, (int)(size & 1), on_device);
}
