// resample.hip -- sample-rate changing preprocess ops on [batch][n] float32 rows
// (the decimate / interpolate / running_rms entries of xcorr/stream/process.py:29-94,
// typical chain scripts/cc.py:46-113).  decimate and interpolate are ObsPy methods in
// the reference (not vendored): restated from their documented behaviour, parity
// unverified against obspy itself.  running_rms is the reference's own
// xcorr/stream/running_rms.py:73-113.
//
//   xcx_decimate  : optional causal SOS low-pass (the caller passes ObsPy's
//                   lowpass_cheby_2 design) on a work copy, then y[i] = x[i * factor]
//   xcx_lanczos   : y[i] = sum_m x[m] K(p_i - m), p_i = offset + i * step (in input
//                   samples), K(u) = sinc(u) sinc(u/a) for |u| < a, samples outside the
//                   row skipped -- obspy.signal.interpolation.lanczos_interpolation
//   xcx_block_rms : y[i] = sqrt(mean(x[i*step : i*step + win]^2))
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>

#include "../../include/xcx.h"
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

int xcx_internal_sosfilt_decimate(xcx_ctx* ctx, const float* x, long batch, long n, const double* sos, int K, float* dst,
                                  long nout, long factor);

namespace {

constexpr int NT = 256;
inline unsigned grid_for(long total) { return (unsigned)std::min<long>((total + NT - 1) / NT, 256L * 32); }

__global__ __launch_bounds__(NT) void k_stride_gather(const float* __restrict__ x, float* __restrict__ y, long batch,
                                                      long n, long nout, long factor) {
  const long total = batch * nout;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long b = i / nout, k = i % nout;
    y[i] = x[b * n + k * factor];
  }
}

// Lanczos kernel sinc(u) sinc(u / a), |u| < a, at u = f + d with f = frac(p) and integer d: sin(pi u) = (-1)^d sin(pi f)
// and sin(pi u / a) follows from (sin, cos)(pi f / a) by rotating with the constant step pi / a -- two sincospi per
// OUTPUT and four multiply-adds per tap instead of two sinpi per tap (float64 throughout; the rotation accumulates
// at most 2a roundings; the two taps with |u| < 1 take sinpi(u / a) directly).  The kernel was 87 % of the 40 -> 50 Hz preprocess list of scripts/cc.py.
__global__ __launch_bounds__(NT) void k_lanczos(const float* __restrict__ x, float* __restrict__ y, long batch, long n,
                                                long nout, double offset, double step, int a) {
  const long total = batch * nout;
  const double inv_a = 1.0 / (double)a;
  double sd, cd;
  sincospi(inv_a, &sd, &cd);                       // rotation by pi / a
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long b = i / nout, k = i % nout;
    const double p = offset + (double)k * step;
    const long c = (long)floor(p);
    const double f = p - (double)c;                 // [0, 1)
    const float* __restrict__ row = x + b * n;
    // taps m = c - a + 1 .. c + a, u = f + (c - m) = f + d, d = a - 1 .. -a
    double sf, cf_unused;
    sincospi(f, &sf, &cf_unused);
    (void)cf_unused;
    double s, co;                                   // (sin, cos)(pi (f + d) / a), starting at d = a - 1
    sincospi((f + (double)(a - 1)) * inv_a, &s, &co);
    double sgn = ((a - 1) & 1) ? -1.0 : 1.0;        // (-1)^d
    double acc = 0.0;
    for (long m = c - a + 1; m <= c + a; ++m) {
      const double u = p - (double)m;
      if (m >= 0 && m < n && u > -(double)a && u < (double)a) {
        double w;
        if (u == 0.0) {
          w = 1.0;
        } else {
          const double pu = M_PI * u;
          // the two taps next to p (|u| < 1): sin(pi u / a) is small there and the rotated value carries the
          // absolute rounding of up to a steps -- evaluate it directly (a fraction p - floor(p) of 1e-12 made the
          // centre weight wrong by 1e-3)
          const double su = (u > -1.0 && u < 1.0) ? sinpi(u * inv_a) : s;
          w = (sgn * sf) * su / (pu * pu * inv_a);
        }
        acc += (double)row[m] * w;
      }
      // next tap: d -> d - 1, rotate by -pi / a
      const double s2 = s * cd - co * sd, c2 = co * cd + s * sd;
      s = s2; co = c2; sgn = -sgn;
    }
    y[i] = (float)acc;
  }
}

// Rational steps (40 -> 50 Hz: step 4/5; 50 -> 20 Hz: 5/2 -- every resampling between round rates): the positions
// p_k = offset + k step repeat their fractional part with period Q (step = P / Q), so the 2a weights of an output
// depend on k mod Q only.  k_lanczos_table evaluates them once per call, directly (sinpi of the tap distance, float64:
// the reference's own expression), and k_lanczos_poly (below) is a polyphase FIR: 2a multiply-adds per output instead of
// 2a rotations, divisions and range tests (the general kernel above ran at 40 % of the float64 issue rate; it remains for
// irrational steps).  W[q][j]: weight of tap m = c - a + 1 + j for phase q; cq[q] = floor(offset + q step).
__global__ void k_lanczos_table(double* __restrict__ W, long* __restrict__ cq, int Q, int a, double offset, double step) {
  const int q = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= Q || j >= 2 * a) return;
  const double p = offset + (double)q * step;
  const long c = (long)floor(p);
  if (j == 0) cq[q] = c;
  const double u = p - (double)(c - a + 1 + j);
  double w = 0.0;
  if (u > -(double)a && u < (double)a) {
    if (u == 0.0) w = 1.0;
    else {
      const double pu = M_PI * u;
      w = sinpi(u) * sinpi(u / (double)a) / (pu * pu / (double)a);
    }
  }
  W[(size_t)q * 2 * a + j] = w;
}

// One phase per workgroup row (blockIdx.y = q): the outputs k = q + Q r of consecutive lanes use the SAME weights -- a
// uniform address, so they arrive through the scalar cache and sit in SGPRs -- and their first taps are P samples apart,
// so the four-sample loads of a wave are (nearly) contiguous.  2a multiply-adds, a/2 vector loads per output.
struct __attribute__((packed, aligned(4))) F4u { float v[4]; };       // four samples at a 4-byte aligned address
__global__ __launch_bounds__(NT) void k_lanczos_poly(const float* __restrict__ x, float* __restrict__ y, long n, long nout,
                                                     const double* __restrict__ W, const long* __restrict__ cq, int Q, long P,
                                                     int a) {
  const int q = blockIdx.y;
  const long b = blockIdx.z;
  const long r = (long)blockIdx.x * NT + threadIdx.x;
  const long k = q + (long)Q * r;
  if (k >= nout) return;
  const long m0 = cq[q] + r * P - a + 1;            // first tap
  const float* __restrict__ row = x + b * n;
  const double* __restrict__ w = W + (size_t)q * 2 * a;
  double acc = 0.0;
  const int ntap = 2 * a;
  if (m0 >= 0 && m0 + ntap <= n) {
    int j = 0;
    for (; j + 8 <= ntap; j += 8) {                  // taps in order, eight samples (two loads) in flight
      const F4u v0 = *reinterpret_cast<const F4u*>(row + m0 + j), v1 = *reinterpret_cast<const F4u*>(row + m0 + j + 4);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = fma((double)v0.v[t], w[j + t], acc);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = fma((double)v1.v[t], w[j + 4 + t], acc);
    }
    for (; j < ntap; ++j) acc = fma((double)row[m0 + j], w[j], acc);
  } else {
    for (int j = 0; j < ntap; ++j) {
      const long m = m0 + j;
      if (m >= 0 && m < n) acc = fma((double)row[m], w[j], acc);
    }
  }
  y[b * nout + k] = (float)acc;
}

// one wave per output block
__global__ __launch_bounds__(NT) void k_block_rms(const float* __restrict__ x, float* __restrict__ y, long batch, long n,
                                                  long nwin, long win, long step) {
  const long total = batch * nwin;
  const int lane = threadIdx.x & 63;
  for (long w = (long)blockIdx.x * (NT / 64) + (threadIdx.x >> 6); w < total; w += (long)gridDim.x * (NT / 64)) {
    const long b = w / nwin, k = w % nwin;
    const float* __restrict__ src = x + b * n + k * step;
    double acc = 0.0;
    for (long t = lane; t < win; t += 64) { const double v = src[t]; acc += v * v; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if (lane == 0) y[w] = (float)sqrt(acc / (double)win);
  }
}

// host <-> device staging for out-of-place ops (device pointers pass through)
struct IO {
  xcx_ctx* ctx; const float* in_d = nullptr; float* out_d = nullptr; float* out_h = nullptr;
  void *a = nullptr, *b = nullptr; size_t out_bytes = 0; int on_device;
  void *pin_in = nullptr, *pin_out = nullptr;      // host buffers page-registered for the call (link-speed DMA)
  int begin(xcx_ctx* c, const float* in, size_t in_bytes, float* out, size_t ob, int dev) {
    ctx = c; on_device = dev; out_bytes = ob;
    if (dev) { in_d = in; out_d = out; return XCX_OK; }
    if (in_bytes >= ((size_t)1 << 20) && hipHostRegister(const_cast<float*>(in), in_bytes, hipHostRegisterDefault) == hipSuccess) pin_in = const_cast<float*>(in);
    else (void)hipGetLastError();
    if (ob >= ((size_t)1 << 20) && hipHostRegister(out, ob, hipHostRegisterDefault) == hipSuccess) pin_out = out;
    else (void)hipGetLastError();
    HIP_TRY(ctx_malloc(ctx, &a, in_bytes));
    HIP_TRY(ctx_malloc(ctx, &b, ob));
    HIP_TRY(hipMemcpyAsync(a, in, in_bytes, hipMemcpyHostToDevice, c->stream));
    in_d = (const float*)a; out_d = (float*)b; out_h = out;
    return XCX_OK;
  }
  int end(int rc) {
    if (!on_device) {
      if (rc == XCX_OK && hipMemcpyAsync(out_h, out_d, out_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc = set_err(XCX_EHIP, "copy back failed");
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == XCX_OK) rc = set_err(XCX_EHIP, "stream sync failed");
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    a = b = nullptr;
    if (pin_in) (void)hipHostUnregister(pin_in);        // after the synchronisation above
    if (pin_out) (void)hipHostUnregister(pin_out);
    pin_in = pin_out = nullptr;
    return rc;
  }
};

}  // namespace

extern "C" int xcx_decimate(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long factor,
                            const double* sos, int n_sections, int on_device) {
  if (!ctx || !x || !y) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || factor < 1) return set_err(XCX_EINVAL, "batch, n and factor must be >= 1");
  if (n_sections < 0 || (n_sections > 0 && !sos)) return set_err(XCX_EINVAL, "bad sos argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const long nout = (n + factor - 1) / factor;            // len(x[::factor])
  const size_t in_bytes = sizeof(float) * (size_t)batch * n;
  IO io;
  int rc = io.begin(ctx, x, in_bytes, y, sizeof(float) * (size_t)batch * nout, on_device);
  if (rc) return io.end(rc);
  const float* src = io.in_d;
  if (n_sections > 0 && factor <= 64) {
    // the filter's last pass stores only the samples that are kept (sos.hip: Decim)
    return io.end(xcx_internal_sosfilt_decimate(ctx, io.in_d, batch, n, sos, n_sections, io.out_d, nout, factor));
  }
  if (n_sections > 0) {
    DevBuf& work = ctx->scratch[7];                        // the filter runs in place on a copy
    if ((rc = work.ensure(ctx, in_bytes))) return io.end(rc);
    if (hipMemcpyAsync(work.ptr, io.in_d, in_bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
      return io.end(set_err(XCX_EHIP, "work copy failed"));
    if ((rc = xcx_sosfilt(ctx, (float*)work.ptr, batch, n, sos, n_sections, 1))) return io.end(rc);
    src = (const float*)work.ptr;
  }
  hipLaunchKernelGGL(k_stride_gather, dim3(grid_for(batch * nout)), dim3(NT), 0, ctx->stream, src, io.out_d, batch, n,
                     nout, factor);
  if (hipGetLastError() != hipSuccess) rc = set_err(XCX_EHIP, "k_stride_gather launch failed");
  return io.end(rc);
}

extern "C" int xcx_lanczos(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long n_out, double offset,
                           double step, int a, int on_device) {
  if (!ctx || !x || !y) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || n_out < 1) return set_err(XCX_EINVAL, "batch, n and n_out must be >= 1");
  if (a < 1 || a > 1024 || !(step > 0.0)) return set_err(XCX_EINVAL, "a must be in [1,1024] and step > 0");
  HIP_TRY(hipSetDevice(ctx->device));
  IO io;
  int rc = io.begin(ctx, x, sizeof(float) * (size_t)batch * n, y, sizeof(float) * (size_t)batch * n_out, on_device);
  if (rc) return io.end(rc);
  // step = P / Q with a small Q?  (exact in float64 for the ratios of round sampling rates)
  int Q = 0;
  long P = 0;
  for (int q = 1; q <= 64 && !Q; ++q) {
    const double pq = step * (double)q, pr = std::round(pq);
    if (pr >= 1.0 && std::fabs(pq - pr) <= 4e-16 * pr) { Q = q; P = (long)pr; }
  }
  if (Q && n_out >= 4 * Q) {
    DevBuf& tab = ctx->scratch[6];
    const size_t wbytes = sizeof(double) * (size_t)Q * 2 * a;
    if ((rc = tab.ensure(ctx, wbytes + sizeof(long) * Q))) return io.end(rc);
    double* W = (double*)tab.ptr;
    long* cq = (long*)((char*)tab.ptr + wbytes);
    hipLaunchKernelGGL(k_lanczos_table, dim3((unsigned)((2 * a + 63) / 64), (unsigned)Q), dim3(64), 0, ctx->stream, W, cq, Q, a,
                       offset, step);
    const long per_phase = (n_out + Q - 1) / Q;
    for (long b0 = 0; b0 < batch; b0 += 65535) {       // grid z limit
      const long nb = std::min<long>(65535, batch - b0);
      hipLaunchKernelGGL(k_lanczos_poly, dim3((unsigned)((per_phase + NT - 1) / NT), (unsigned)Q, (unsigned)nb), dim3(NT), 0,
                         ctx->stream, io.in_d + b0 * n, io.out_d + b0 * n_out, n, n_out, (const double*)W, (const long*)cq, Q, P, a);
    }
  } else {
    hipLaunchKernelGGL(k_lanczos, dim3(grid_for(batch * n_out)), dim3(NT), 0, ctx->stream, io.in_d, io.out_d, batch, n,
                       n_out, offset, step, a);
  }
  if (hipGetLastError() != hipSuccess) rc = set_err(XCX_EHIP, "k_lanczos launch failed");
  return io.end(rc);
}

extern "C" int xcx_block_rms(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long win, long step,
                             long nwin, int on_device) {
  if (!ctx || !x || !y) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || win < 1 || step < 1 || nwin < 1) return set_err(XCX_EINVAL, "sizes must be >= 1");
  if ((nwin - 1) * step + win > n) return set_err(XCX_EINVAL, "windows exceed the row: (%ld-1)*%ld+%ld > %ld", nwin, step, win, n);
  HIP_TRY(hipSetDevice(ctx->device));
  IO io;
  int rc = io.begin(ctx, x, sizeof(float) * (size_t)batch * n, y, sizeof(float) * (size_t)batch * nwin, on_device);
  if (rc) return io.end(rc);
  hipLaunchKernelGGL(k_block_rms, dim3((unsigned)std::min<long>((batch * nwin + 3) / 4, 256L * 32)), dim3(NT), 0,
                     ctx->stream, io.in_d, io.out_d, batch, n, nwin, win, step);
  if (hipGetLastError() != hipSuccess) rc = set_err(XCX_EHIP, "k_block_rms launch failed");
  return io.end(rc);
}
