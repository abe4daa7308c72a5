// ops.hip -- row-wise signal operations on [batch][n] float32, in place.
// Reference semantics are cited per entry point in include/xcx.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <vector>

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

namespace {

constexpr int NT = 256;

// block-wide sum in double, fixed order; result broadcast to all threads
__device__ inline double block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// One-workgroup-per-row loops.  As `for (i = tid; i < n; i += NT) use(row[i])` the compiler emits ONE load per
// iteration and waits for it before the next (a rolled loop: n / NT serialised memory round trips -- 1400 of them
// for a full-lag CCF row, ~1.3 us each); here kUL loads are issued back to back and then consumed IN THE SAME
// ORDER, so every float64 sum keeps its value bit for bit.
constexpr int kUL = 8;
template <class F> __device__ inline void row_visit(const float* __restrict__ row, long lo, long hi, F f) {
  long i = lo + threadIdx.x;
  for (; i + (kUL - 1) * NT < hi; i += kUL * NT) {
    float v[kUL];
#pragma unroll
    for (int k = 0; k < kUL; ++k) v[k] = row[i + k * NT];
#pragma unroll
    for (int k = 0; k < kUL; ++k) f(i + k * NT, v[k]);
  }
  for (; i < hi; i += NT) f(i, row[i]);
}
// dst[i] = f(i, src[i]) over [lo, hi); dst may be src (each element is read before it is written, by its own thread)
template <class F> __device__ inline void row_update(const float* src, float* dst, long lo, long hi, F f) {
  long i = lo + threadIdx.x;
  for (; i + (kUL - 1) * NT < hi; i += kUL * NT) {
    float v[kUL];
#pragma unroll
    for (int k = 0; k < kUL; ++k) v[k] = src[i + k * NT];
#pragma unroll
    for (int k = 0; k < kUL; ++k) dst[i + k * NT] = f(i + k * NT, v[k]);
  }
  for (; i < hi; i += NT) dst[i] = f(i, src[i]);
}

enum { OP_NORM = 0, OP_DEMEAN = 1, OP_DETREND = 2 };

// one workgroup per row; the row is read twice (second time from L2)
template <int OP>
__global__ __launch_bounds__(NT) void k_row_reduce_apply(float* __restrict__ x, long n) {
  __shared__ double red[NT];
  float* __restrict__ row = x + (size_t)blockIdx.x * n;
  if (OP == OP_NORM) {
    double acc = 0.0;
    row_visit(row, 0, n, [&](long, float f) { const double v = f; acc += v * v; });
    const double nrm = sqrt(block_sum(acc, red));
    // x / ||x|| in the input precision (signal/normalize.py:80 `y = x/n`)
    const float fn = (float)nrm;
    row_update(row, row, 0, n, [=](long, float v) { return v / fn; });
  } else if (OP == OP_DEMEAN) {
    double acc = 0.0;
    row_visit(row, 0, n, [&](long, float v) { acc += (double)v; });
    const float mean = (float)(block_sum(acc, red) / (double)n);
    row_update(row, row, 0, n, [=](long, float v) { return v - mean; });
  } else {
    // least-squares line over t = 0..n-1 (scipy.stats.linregress)
    const double tbar = 0.5 * (double)(n - 1);
    double sy = 0.0, sty = 0.0;
    row_visit(row, 0, n, [&](long i, float f) {
      const double v = f;
      sy += v;
      sty += ((double)i - tbar) * v;
    });
    sy = block_sum(sy, red);
    sty = block_sum(sty, red);
    const double stt = (double)n * ((double)n * (double)n - 1.0) / 12.0;
    const double m = (n > 1) ? sty / stt : 0.0;
    const double b = sy / (double)n - m * tbar;
    row_update(row, row, 0, n, [=](long i, float v) { return (float)((double)v - (m * (double)i + b)); });
  }
}

// few long rows (day-long station traces): split every row over kParts
// workgroups -- partial sums first, then each workgroup re-derives the row
// totals from the partials in fixed order and applies them to its slice.
constexpr int kParts = 64;

template <int OP>
__global__ __launch_bounds__(NT) void k_row_partial(const float* __restrict__ x, long n, double* __restrict__ part) {
  __shared__ double red[NT];
  const float* __restrict__ row = x + (size_t)blockIdx.y * n;
  const int tid = threadIdx.x;
  // slices of a multiple of 4 samples: 16-byte loads whenever the row itself is 16-byte aligned (one read
  // of 64 day-long rows: 0.32 ms with 4-byte loads, the replicated part of a window-sharded step)
  const long per = ((n + kParts - 1) / kParts + 3) & ~3L;
  const long lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  const double tbar = 0.5 * (double)(n - 1);
  double a0 = 0.0, a1 = 0.0;
  auto acc = [&](long i, double v) {
    if (OP == OP_NORM) a0 += v * v;
    else { a0 += v; if (OP == OP_DETREND) a1 += ((double)i - tbar) * v; }
  };
  long i0 = lo;
  if ((reinterpret_cast<unsigned long long>(row) & 15ULL) == 0) {
    const long nv = (hi > lo) ? (hi - lo) / 4 : 0;
    for (long q = tid; q < nv; q += NT) {
      const long i = lo + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      acc(i, v.x); acc(i + 1, v.y); acc(i + 2, v.z); acc(i + 3, v.w);
    }
    i0 = lo + 4 * nv;
  }
  for (long i = i0 + tid; i < hi; i += NT) acc(i, row[i]);
  a0 = block_sum(a0, red);
  if (OP == OP_DETREND) a1 = block_sum(a1, red);
  if (tid == 0) {
    part[((size_t)blockIdx.y * kParts + blockIdx.x) * 2] = a0;
    part[((size_t)blockIdx.y * kParts + blockIdx.x) * 2 + 1] = a1;
  }
}

// src -> dst (dst may be src): out of place the separate copy of an untouched original is saved
// [c0, c1): the columns written (the statistics are always those of the whole row)
template <int OP>
__global__ __launch_bounds__(NT) void k_row_apply(const float* src, float* dst, long n, const double* __restrict__ part,
                                                  long c0, long c1) {
  const float* rin = src + (size_t)blockIdx.y * n;
  float* row = dst + (size_t)blockIdx.y * n;
  const long per = (c1 - c0 + kParts - 1) / kParts;
  const long lo = c0 + blockIdx.x * per, hi = (lo + per < c1) ? lo + per : c1;
  double s0 = 0.0, s1 = 0.0;
  for (int p = 0; p < kParts; ++p) {
    s0 += part[((size_t)blockIdx.y * kParts + p) * 2];
    s1 += part[((size_t)blockIdx.y * kParts + p) * 2 + 1];
  }
  if (OP == OP_NORM) {
    const float fn = (float)sqrt(s0);
    row_update(rin, row, lo, hi, [=](long, float v) { return v / fn; });
  } else if (OP == OP_DEMEAN) {
    const float mean = (float)(s0 / (double)n);
    row_update(rin, row, lo, hi, [=](long, float v) { return v - mean; });
  } else {
    const double tbar = 0.5 * (double)(n - 1);
    const double stt = (double)n * ((double)n * (double)n - 1.0) / 12.0;
    const double m = (n > 1) ? s1 / stt : 0.0;
    const double b = s0 / (double)n - m * tbar;
    row_update(rin, row, lo, hi, [=](long i, float v) { return (float)((double)v - (m * (double)i + b)); });
  }
}

// taper: both flanks of nmax samples.  wtype 0: periodic Hann of length
// 2*nmax (scipy.signal.windows.hann(2*nmax, False)); 1: ObsPy cosine flank
__global__ void k_taper(float* __restrict__ x, long batch, long n, long nmax, int wtype) {
  const long per = 2 * nmax;
  const long total = batch * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / per, q = idx % per;
    long i;      // sample index
    double w;
    if (q < nmax) {                  // left flank, unless overwritten by the right one
      i = q;
      if (i >= n - nmax) continue;   // right flank wins on overlap (window.py:105-106)
      w = (wtype == 0) ? 0.5 * (1.0 - cos(M_PI * (double)q / (double)nmax))
                       : ((nmax > 1) ? 0.5 * (1.0 - cos(M_PI * (double)q / (double)(nmax - 1))) : 0.0);
    } else {
      const long e = q - nmax;       // 0..nmax-1 from the start of the right flank
      i = n - nmax + e;
      if (i < 0) continue;
      w = (wtype == 0) ? 0.5 * (1.0 + cos(M_PI * (double)e / (double)nmax))
                       : ((nmax > 1) ? 0.5 * (1.0 - cos(M_PI * (double)(nmax - 1 - e) / (double)(nmax - 1))) : 0.0);
    }
    float* p = x + (size_t)r * n + i;
    *p = *p * (float)w;
  }
}

__global__ void k_unbias(float* __restrict__ x, long batch, long L, long npts, long idx_min) {
  const long total = batch * L;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long o = idx % L;
    long lag = idx_min + o - (npts - 1);
    if (lag < 0) lag = -lag;
    const float w = (float)((double)npts / (double)(npts - lag));
    x[idx] = x[idx] * w;
  }
}

// detrend with the reference's NaN handling (signal/detrend.py:87-110, skipna=True): the mean /
// least-squares line is fitted to the finite samples only (scipy.stats.linregress over t[~nan],
// y[~nan]) and subtracted everywhere; NaN samples stay NaN.  One workgroup per row, float64 sums.
__global__ __launch_bounds__(NT) void k_detrend_nan(float* __restrict__ x, long n, int linear) {
  __shared__ double red[NT];
  float* __restrict__ row = x + (size_t)blockIdx.x * n;
  double c = 0.0, sy = 0.0, st = 0.0;
  row_visit(row, 0, n, [&](long i, float v) {
    if (v == v) { c += 1.0; sy += (double)v; st += (double)i; }
  });
  c = block_sum(c, red); sy = block_sum(sy, red); st = block_sum(st, red);
  if (c == 0.0) return;                                   // all-NaN row stays as it is
  const double ybar = sy / c, tbar = st / c;
  double m = 0.0;
  if (linear) {
    double stt = 0.0, sty = 0.0;
    row_visit(row, 0, n, [&](long i, float v) {
      if (v == v) { const double dt = (double)i - tbar; stt += dt * dt; sty += dt * ((double)v - ybar); }
    });
    stt = block_sum(stt, red); sty = block_sum(sty, red);
    m = (stt > 0.0) ? sty / stt : 0.0;
  }
  const double b = ybar - m * tbar;
  if (linear) {
    row_update(row, row, 0, n, [=](long i, float v) { return (float)((double)v - (m * (double)i + b)); });
  } else {
    const float mean = (float)ybar;
    row_update(row, row, 0, n, [=](long, float v) { return v - mean; });
  }
}

// taper with an arbitrary flank table (any scipy.signal.windows type, signal/window.py:92-106):
// w[0:nmax] = tab[0:nmax], w[-nmax:] = tab[nmax:2 nmax], ones between; the right flank wins on overlap
__global__ void k_taper_table(float* __restrict__ x, long batch, long n, long nmax, const float* __restrict__ tab) {
  const long per = 2 * nmax;
  const long total = batch * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long r = idx / per, q = idx % per;
    long i;
    if (q < nmax) { i = q; if (i >= n - nmax) continue; }
    else { i = n - nmax + (q - nmax); if (i < 0) continue; }
    float* p = x + (size_t)r * n + i;
    *p = *p * tab[q];
  }
}

// fused head of the post-correlation chain (core/postprocess.py:133-136): unbias -> demean ->
// taper(hann flanks), one workgroup per row, the row read twice (second time from L2) and written
// once instead of five passes over it.  Every intermediate is rounded to float32 exactly where the
// three separate ops round (x * (float)w, mean in float64 rounded to float32, * (float)window), so
// the result is bit-identical to xcx_unbias + xcx_demean + xcx_taper.
__device__ inline float taper_weight(long i, long n, long nmax, int wtype) {
  // right flank wins on overlap (window.py:105-106)
  if (i >= n - nmax) {
    const long e = i - (n - nmax);
    return (float)((wtype == 0) ? 0.5 * (1.0 + cos(M_PI * (double)e / (double)nmax))
                                : ((nmax > 1) ? 0.5 * (1.0 - cos(M_PI * (double)(nmax - 1 - e) / (double)(nmax - 1))) : 0.0));
  }
  if (i < nmax)
    return (float)((wtype == 0) ? 0.5 * (1.0 - cos(M_PI * (double)i / (double)nmax))
                                : ((nmax > 1) ? 0.5 * (1.0 - cos(M_PI * (double)i / (double)(nmax - 1))) : 0.0));
  return 1.f;
}

__global__ __launch_bounds__(NT) void k_post_prep(float* __restrict__ x, long L, long npts, long idx_min, long nmax,
                                                   int wtype, int do_unbias) {
  __shared__ double red[NT];
  float* __restrict__ row = x + (size_t)blockIdx.x * L;
  auto unb = [=](long o, float v) -> float {
    if (!do_unbias) return v;
    long lag = idx_min + o - (npts - 1);
    if (lag < 0) lag = -lag;
    return v * (float)((double)npts / (double)(npts - lag));
  };
  double acc = 0.0;
  row_visit(row, 0, L, [&](long i, float v) { acc += (double)unb(i, v); });
  const float mean = (float)(block_sum(acc, red) / (double)L);
  row_update(row, row, 0, L, [=](long i, float x0) {
    float v = unb(i, x0) - mean;
    if (nmax > 0 && (i < nmax || i >= L - nmax)) v = v * taper_weight(i, L, nmax, wtype);
    return v;
  });
}

// The same for few long rows (the production lag axis: 1.62 M lags per row): the row split over kParts workgroups
// exactly like k_row_partial / k_row_apply of xcx_demean (same slices, same order of the float64 sums), so that the
// result stays bit-identical to xcx_unbias + xcx_demean + xcx_taper there too.  One workgroup per row read such a
// row at 0.7 TB/s.
__global__ __launch_bounds__(NT) void k_post_partial(const float* __restrict__ x, long L, long npts, long idx_min, int do_unbias,
                                                      double* __restrict__ part) {
  __shared__ double red[NT];
  const float* __restrict__ row = x + (size_t)blockIdx.y * L;
  const int tid = threadIdx.x;
  const long per = ((L + kParts - 1) / kParts + 3) & ~3L;
  const long lo = blockIdx.x * per, hi = (lo + per < L) ? lo + per : L;
  auto unb = [=](long o, float v) -> float {
    if (!do_unbias) return v;
    long lag = idx_min + o - (npts - 1);
    if (lag < 0) lag = -lag;
    return v * (float)((double)npts / (double)(npts - lag));
  };
  double a0 = 0.0;
  long i0 = lo;
  if ((reinterpret_cast<unsigned long long>(row) & 15ULL) == 0) {
    const long nv = (hi > lo) ? (hi - lo) / 4 : 0;
    for (long q = tid; q < nv; q += NT) {
      const long i = lo + 4 * q;
      const float4 v = *reinterpret_cast<const float4*>(row + i);
      a0 += (double)unb(i, v.x); a0 += (double)unb(i + 1, v.y); a0 += (double)unb(i + 2, v.z); a0 += (double)unb(i + 3, v.w);
    }
    i0 = lo + 4 * nv;
  }
  for (long i = i0 + tid; i < hi; i += NT) a0 += (double)unb(i, row[i]);
  a0 = block_sum(a0, red);
  if (tid == 0) {
    part[((size_t)blockIdx.y * kParts + blockIdx.x) * 2] = a0;
    part[((size_t)blockIdx.y * kParts + blockIdx.x) * 2 + 1] = 0.0;
  }
}
__global__ __launch_bounds__(NT) void k_post_apply(float* __restrict__ x, long L, long npts, long idx_min, long nmax, int wtype,
                                                    int do_unbias, const double* __restrict__ part) {
  float* __restrict__ row = x + (size_t)blockIdx.y * L;
  const long per = (L + kParts - 1) / kParts;
  const long lo = blockIdx.x * per, hi = (lo + per < L) ? lo + per : L;
  double s0 = 0.0;
  for (int p = 0; p < kParts; ++p) s0 += part[((size_t)blockIdx.y * kParts + p) * 2];
  const float mean = (float)(s0 / (double)L);
  row_update(row, row, lo, hi, [=](long i, float x0) {
    float v = x0;
    if (do_unbias) {
      long lag = idx_min + i - (npts - 1);
      if (lag < 0) lag = -lag;
      v = v * (float)((double)npts / (double)(npts - lag));
    }
    v = v - mean;
    if (nmax > 0 && (i < nmax || i >= L - nmax)) v = v * taper_weight(i, L, nmax, wtype);
    return v;
  });
}

__global__ void k_onebit(float* __restrict__ x, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const float v = x[idx];
    x[idx] = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f);
  }
}

// snr pieces (xcorr/signal/snr.py:107-116): per row the rms over the noise
// mask and the maximum absolute value (first occurrence) over the signal mask
__global__ __launch_bounds__(NT) void k_snr(const float* __restrict__ x, long n,
                                            const unsigned char* __restrict__ sig,
                                            const unsigned char* __restrict__ noise,
                                            float* __restrict__ s_out, float* __restrict__ n_out,
                                            int* __restrict__ arg_out) {
  __shared__ double red[NT];
  __shared__ float smax[NT];
  __shared__ long sidx[NT];
  const float* __restrict__ row = x + (size_t)blockIdx.x * n;
  const int tid = threadIdx.x;
  double acc = 0.0, cnt = 0.0;
  float best = -1.f;
  long bi = 0x7fffffffffffffffL;
  auto take = [&](long i, float v, unsigned char nz, unsigned char sg) {
    if (nz) { acc += (double)v * (double)v; cnt += 1.0; }
    if (sg) {
      const float a = fabsf(v);
      if (a > best) { best = a; bi = i; }        // strided scan keeps the smallest index per thread
    }
  };
  {
    long i = tid;
    for (; i + (kUL - 1) * NT < n; i += kUL * NT) {     // row and masks of kUL trips in flight, consumed in order
      float v[kUL]; unsigned char nz[kUL], sg[kUL];
#pragma unroll
      for (int k = 0; k < kUL; ++k) { v[k] = row[i + k * NT]; nz[k] = noise[i + k * NT]; sg[k] = sig[i + k * NT]; }
#pragma unroll
      for (int k = 0; k < kUL; ++k) take(i + k * NT, v[k], nz[k], sg[k]);
    }
    for (; i < n; i += NT) take(i, row[i], noise[i], sig[i]);
  }
  const double ss = block_sum(acc, red);
  const double cc = block_sum(cnt, red);
  smax[tid] = best; sidx[tid] = bi;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if (tid < o) {
      const float b2 = smax[tid + o]; const long i2 = sidx[tid + o];
      if (b2 > smax[tid] || (b2 == smax[tid] && i2 < sidx[tid])) { smax[tid] = b2; sidx[tid] = i2; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    n_out[blockIdx.x] = (float)sqrt(ss / cc);
    s_out[blockIdx.x] = smax[0];
    arg_out[blockIdx.x] = (int)sidx[0];
  }
}

// stage host data on the device when on_device == 0
struct Staged {
  // host rows of a call are staged in a device copy for its duration; the destructor releases the copy on every
  // path that leaves without end() / staged_discard() (an error return between begin() and end())
  xcx_ctx* ctx = nullptr; float* host = nullptr; float* dev = nullptr; size_t bytes = 0; bool staged = false;
  Staged() = default;
  Staged(const Staged&) = delete;
  Staged& operator=(const Staged&) = delete;
  ~Staged() { drop(); }
  void drop() {
    if (staged && dev) {
      (void)hipStreamSynchronize(ctx->stream);        // a kernel may still be using the copy
      (void)hipFree(dev);
    }
    staged = false;
  }
  int begin(xcx_ctx* c, float* x, size_t nbytes, int on_device) {
    ctx = c; host = x; bytes = nbytes; staged = false; dev = x;
    if (!on_device) {
      dev = nullptr;
      HIP_TRY(ctx_malloc(ctx, (void**)&dev, bytes ? bytes : 1));
      staged = true;
      HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    return XCX_OK;
  }
  int end() {
    if (staged) {
      HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      staged = false;
      HIP_TRY(hipFree(dev));
    }
    return XCX_OK;
  }
  void staged_discard() { drop(); }         // read-only op: nothing to copy back
};

inline int grid_for(long total) {
  long g = (total + NT - 1) / NT;
  return (int)std::max(1L, std::min(g, 256L * 8));
}

template <int OP> int row_op(xcx_ctx* ctx, float* x, long batch, long n, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) return rc;
  if (batch < 1024 && n >= 65536) {
    if ((rc = ctx->scratch[0].ensure(ctx, sizeof(double) * 2 * kParts * (size_t)batch))) return rc;
    double* part = (double*)ctx->scratch[0].ptr;
    hipLaunchKernelGGL(k_row_partial<OP>, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, st.dev, n, part);
    hipLaunchKernelGGL(k_row_apply<OP>, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, (const float*)st.dev, st.dev, n, part, 0L, n);
    HIP_TRY(hipGetLastError());
    return st.end();
  }
  hipLaunchKernelGGL(k_row_reduce_apply<OP>, dim3((unsigned)batch), dim3(NT), 0, ctx->stream, st.dev, n);
  HIP_TRY(hipGetLastError());
  return st.end();
}

// device rows src -> dst; the source stays untouched
template <int OP> int row_op_to(xcx_ctx* ctx, const float* src, float* dst, long batch, long n, long c0 = 0, long c1 = -1) {
  if (!ctx || !src || !dst) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (c1 < 0) c1 = n;
  if (c0 < 0 || c1 > n || c0 >= c1) return set_err(XCX_EINVAL, "bad column range [%ld, %ld) of %ld", c0, c1, n);
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  if ((batch < 1024 && n >= 65536) || c0 != 0 || c1 != n) {
    if ((rc = ctx->scratch[0].ensure(ctx, sizeof(double) * 2 * kParts * (size_t)batch))) return rc;
    double* part = (double*)ctx->scratch[0].ptr;
    hipLaunchKernelGGL(k_row_partial<OP>, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, src, n, part);
    hipLaunchKernelGGL(k_row_apply<OP>, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, src, dst, n, part, c0, c1);
    HIP_TRY(hipGetLastError());
    return XCX_OK;
  }
  if (dst != src) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)batch * n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  hipLaunchKernelGGL(k_row_reduce_apply<OP>, dim3((unsigned)batch), dim3(NT), 0, ctx->stream, dst, n);
  HIP_TRY(hipGetLastError());
  return XCX_OK;
}

}  // namespace

extern "C" int xcx_norm1d(xcx_ctx* ctx, float* x, long batch, long n, int on_device) {
  return row_op<OP_NORM>(ctx, x, batch, n, on_device);
}
extern "C" int xcx_demean(xcx_ctx* ctx, float* x, long batch, long n, int on_device) {
  return row_op<OP_DEMEAN>(ctx, x, batch, n, on_device);
}
extern "C" int xcx_detrend_linear(xcx_ctx* ctx, float* x, long batch, long n, int on_device) {
  return row_op<OP_DETREND>(ctx, x, batch, n, on_device);
}
extern "C" int xcx_detrend_to(xcx_ctx* ctx, const float* src, float* dst, long batch, long n, int linear) {
  return linear ? row_op_to<OP_DETREND>(ctx, src, dst, batch, n) : row_op_to<OP_DEMEAN>(ctx, src, dst, batch, n);
}
extern "C" int xcx_detrend_to_cols(xcx_ctx* ctx, const float* src, float* dst, long batch, long n, int linear,
                                   long col0, long col1) {
  return linear ? row_op_to<OP_DETREND>(ctx, src, dst, batch, n, col0, col1)
                : row_op_to<OP_DEMEAN>(ctx, src, dst, batch, n, col0, col1);
}

extern "C" int xcx_taper(xcx_ctx* ctx, float* x, long batch, long n, long nmax, int wtype, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (nmax < 1 || nmax > n)
    return set_err(XCX_EINVAL, "taper length nmax=%ld must be in [1, n=%ld]", nmax, n);
  if (wtype != 0 && wtype != 1) return set_err(XCX_EINVAL, "wtype must be 0 (hann) or 1 (obspy cosine)");
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_taper, dim3(grid_for(batch * 2 * nmax)), dim3(NT), 0, ctx->stream, st.dev, batch, n, nmax, wtype);
  HIP_TRY(hipGetLastError());
  return st.end();
}

extern "C" int xcx_unbias(xcx_ctx* ctx, float* x, long batch, long npts, long idx_min, long idx_max, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (idx_max == 0) idx_max = 2 * npts - 1;
  if (batch < 1 || npts < 1 || idx_min < 0 || idx_max > 2 * npts - 1 || idx_min >= idx_max)
    return set_err(XCX_EINVAL, "bad unbias arguments");
  const long L = idx_max - idx_min;
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * L * sizeof(float), on_device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_unbias, dim3(grid_for(batch * L)), dim3(NT), 0, ctx->stream, st.dev, batch, L, npts, idx_min);
  HIP_TRY(hipGetLastError());
  return st.end();
}

extern "C" int xcx_detrend_nan(xcx_ctx* ctx, float* x, long batch, long n, int linear, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_detrend_nan, dim3((unsigned)batch), dim3(NT), 0, ctx->stream, st.dev, n, linear ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return st.end();
}

extern "C" int xcx_taper_table(xcx_ctx* ctx, float* x, long batch, long n, const float* flanks, long nmax, int on_device) {
  if (!ctx || !x || !flanks) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (nmax < 1 || nmax > n) return set_err(XCX_EINVAL, "taper length nmax=%ld must be in [1, n=%ld]", nmax, n);
  HIP_TRY(hipSetDevice(ctx->device));
  float* dtab = nullptr;                                   // the table is host memory in both modes
  HIP_TRY(ctx_malloc(ctx, (void**)&dtab, sizeof(float) * 2 * (size_t)nmax));
  if (hipMemcpyAsync(dtab, flanks, sizeof(float) * 2 * (size_t)nmax, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    (void)hipFree(dtab);
    return set_err(XCX_EHIP, "window table upload failed");
  }
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) { (void)hipFree(dtab); return rc; }
  hipLaunchKernelGGL(k_taper_table, dim3(grid_for(batch * 2 * nmax)), dim3(NT), 0, ctx->stream, st.dev, batch, n, nmax, dtab);
  rc = st.end();
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(dtab);
  return rc;
}

extern "C" int xcx_post_prep(xcx_ctx* ctx, float* x, long batch, long npts, long idx_min, long idx_max,
                             int unbias, long taper_nmax, int wtype, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (idx_max == 0) idx_max = 2 * npts - 1;
  if (batch < 1 || npts < 1 || idx_min < 0 || idx_max > 2 * npts - 1 || idx_min >= idx_max)
    return set_err(XCX_EINVAL, "bad lag slice");
  const long L = idx_max - idx_min;
  if (taper_nmax < 0 || taper_nmax > L) return set_err(XCX_EINVAL, "taper length nmax=%ld must be in [0, L=%ld]", taper_nmax, L);
  if (wtype != 0 && wtype != 1) return set_err(XCX_EINVAL, "wtype must be 0 (hann) or 1 (obspy cosine)");
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * L * sizeof(float), on_device);
  if (rc) return rc;
  if (batch < 1024 && L >= 65536) {          // few long rows: split like xcx_demean does (same sums, bit for bit)
    if ((rc = ctx->scratch[0].ensure(ctx, sizeof(double) * 2 * kParts * (size_t)batch))) return rc;
    double* part = (double*)ctx->scratch[0].ptr;
    hipLaunchKernelGGL(k_post_partial, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, (const float*)st.dev, L, npts,
                       idx_min, unbias, part);
    hipLaunchKernelGGL(k_post_apply, dim3(kParts, (unsigned)batch), dim3(NT), 0, ctx->stream, st.dev, L, npts, idx_min,
                       taper_nmax, wtype, unbias, (const double*)part);
    HIP_TRY(hipGetLastError());
    return st.end();
  }
  hipLaunchKernelGGL(k_post_prep, dim3((unsigned)batch), dim3(NT), 0, ctx->stream, st.dev, L, npts, idx_min, taper_nmax,
                     wtype, unbias);
  HIP_TRY(hipGetLastError());
  return st.end();
}

// x[r][:] *= factor[r] (float64 factor, product rounded once to float32): Trace.remove_sensitivity / Trace.normalize
// of the preprocess lists (stream/process.py:29-94) on device-resident rows
__global__ void k_scale_rows(float* __restrict__ x, long n, const double* __restrict__ f) {
  const double fr = f[blockIdx.y];
  float* __restrict__ row = x + (size_t)blockIdx.y * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    row[i] = (float)((double)row[i] * fr);
}
// out[r][part] = max |x[r][i]| over the part's slice, NaN if the slice holds one (numpy's np.abs(x).max() and the
// reference propagate it: a gappy row normalises to NaN on the host path and must do so here); the host combines
__global__ __launch_bounds__(NT) void k_absmax_part(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[NT];
  const float* __restrict__ row = x + (size_t)blockIdx.y * n;
  const long per = (n + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  float m = 0.f;
  row_visit(row, lo, hi, [&](long, float v) { const float a = fabsf(v); m = (a > m || a != a) ? a : m; });   // NaN sticks: NaN > x is false
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float a = red[threadIdx.x], b = red[threadIdx.x + o];
      red[threadIdx.x] = (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b);     // fmaxf drops a NaN operand
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

extern "C" int xcx_scale_rows(xcx_ctx* ctx, float* x, long batch, long n, const double* factor, int on_device) {
  if (!ctx || !x || !factor) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || batch > XCX_ROWS_MAX_BATCH) return set_err(XCX_EINVAL, "batch in [1, %ld] and n >= 1 required", XCX_ROWS_MAX_BATCH);
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf& fb = ctx->scratch[0];
  int rc;
  if ((rc = fb.ensure(ctx, sizeof(double) * (size_t)batch))) return rc;
  HIP_TRY(hipMemcpyAsync(fb.ptr, factor, sizeof(double) * (size_t)batch, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));          // `factor` is the caller's
  Staged st;
  if ((rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device))) return rc;
  hipLaunchKernelGGL(k_scale_rows, dim3((unsigned)std::min<long>((n + NT - 1) / NT, 4096), (unsigned)batch), dim3(NT), 0,
                     ctx->stream, st.dev, n, (const double*)fb.ptr);
  HIP_TRY(hipGetLastError());
  return st.end();
}

// (a, b) <- (c00 a + c01 b, c10 a + c11 b), float64 products rounded once
__global__ __launch_bounds__(NT) void k_rotate2(float* __restrict__ a, float* __restrict__ b, long total,
                                                double c00, double c01, double c10, double c11) {
  const long stride = (long)gridDim.x * NT;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
    const double x = (double)a[i], y = (double)b[i];
    a[i] = (float)(c00 * x + c01 * y);
    b[i] = (float)(c10 * x + c11 * y);
  }
}

extern "C" int xcx_rotate2(xcx_ctx* ctx, float* a, float* b, long batch, long n, const double* c4, int on_device) {
  if (!ctx || !a || !b || !c4) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (a == b) return set_err(XCX_EINVAL, "the two components must be different arrays");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)batch * n * sizeof(float);
  Staged sa, sb;
  int rc;
  if ((rc = sa.begin(ctx, a, bytes, on_device))) return rc;
  if ((rc = sb.begin(ctx, b, bytes, on_device))) return rc;
  hipLaunchKernelGGL(k_rotate2, dim3(grid_for(batch * n)), dim3(NT), 0, ctx->stream, sa.dev, sb.dev, batch * n,
                     c4[0], c4[1], c4[2], c4[3]);
  HIP_TRY(hipGetLastError());
  if ((rc = sa.end())) return rc;
  return sb.end();
}

extern "C" int xcx_absmax_rows(xcx_ctx* ctx, const float* x, long batch, long n, float* out, int on_device) {
  if (!ctx || !x || !out) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || batch > XCX_ROWS_MAX_BATCH) return set_err(XCX_EINVAL, "batch in [1, %ld] and n >= 1 required", XCX_ROWS_MAX_BATCH);
  HIP_TRY(hipSetDevice(ctx->device));
  const int parts = (int)std::max<long>(1, std::min<long>(64, n / 4096));
  DevBuf& pb = ctx->scratch[0];
  int rc;
  if ((rc = pb.ensure(ctx, sizeof(float) * (size_t)batch * parts))) return rc;
  Staged st;
  if ((rc = st.begin(ctx, const_cast<float*>(x), (size_t)batch * n * sizeof(float), on_device))) return rc;
  hipLaunchKernelGGL(k_absmax_part, dim3((unsigned)parts, (unsigned)batch), dim3(NT), 0, ctx->stream, (const float*)st.dev, n,
                     (float*)pb.ptr);
  HIP_TRY(hipGetLastError());
  std::vector<float> h((size_t)batch * parts);
  HIP_TRY(hipMemcpyAsync(h.data(), pb.ptr, sizeof(float) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (long r = 0; r < batch; ++r) {
    float m = 0.f;
    for (int q = 0; q < parts; ++q) {
      const float v = h[(size_t)r * parts + q];
      m = (v > m || v != v) ? v : m;                    // a NaN part makes the row's maximum NaN
    }
    out[r] = m;
  }
  st.staged_discard();
  return XCX_OK;
}

extern "C" int xcx_onebit(xcx_ctx* ctx, float* x, long batch, long n, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) return rc;
  hipLaunchKernelGGL(k_onebit, dim3(grid_for(batch * n)), dim3(NT), 0, ctx->stream, st.dev, batch * n);
  HIP_TRY(hipGetLastError());
  return st.end();
}

// ---------------------------------------------------------------------------
// Running-absolute-mean temporal normalisation (Bensen et al. 2007, eq. 1; named by the
// north star, not present in the reference -> self-defined):
//     w[t] = mean_{|tau| <= h, 0 <= t+tau < n} |x[t+tau]|,   y[t] = x[t] / w[t]  (0 where w = 0)
// One workgroup per chunk of kRamChunk samples: |x| of the chunk and its 2h halo goes to
// LDS in float64, an inclusive prefix sum is built with wavefront shuffles (thread-local
// run -> __shfl_up scan across the 64 lanes -> 4 wave totals through LDS), and each
// output takes the difference of two prefix values.
// ---------------------------------------------------------------------------
constexpr int kRamChunk = 2048;
constexpr int kRamMaxHalf = (int)XCX_RAM_MAX_HALF_WIDTH;

__global__ __launch_bounds__(NT) void k_ram(const float* __restrict__ src, float* __restrict__ dst, long n, int h) {
  extern __shared__ double pre[];                 // kRamChunk + 2h prefix sums
  __shared__ double wtot[NT / 64];
  const long b = blockIdx.y;
  const long c0 = (long)blockIdx.x * kRamChunk;
  const long g0 = c0 - h;
  const int span = kRamChunk + 2 * h;
  const float* __restrict__ x = src + b * n;
  const int E = (span + NT - 1) / NT;             // contiguous elements per thread
  const int i0 = threadIdx.x * E;
  double run = 0.0;
  for (int e = 0; e < E; ++e) {
    const int i = i0 + e;
    if (i < span) {
      const long g = g0 + i;
      run += (g >= 0 && g < n) ? (double)fabsf(x[g]) : 0.0;
      pre[i] = run;                               // thread-local inclusive prefix
    }
  }
  // exclusive scan of the per-thread totals: across lanes, then across waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double inc = run;
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  double off = inc - run;
  for (int w = 0; w < wave; ++w) off += wtot[w];
  for (int e = 0; e < E; ++e) {
    const int i = i0 + e;
    if (i < span) pre[i] += off;
  }
  __syncthreads();
  float* __restrict__ y = dst + b * n;
  for (int k = threadIdx.x; k < kRamChunk; k += NT) {
    const long t = c0 + k;
    if (t >= n) break;
    const long lo = t - h < 0 ? 0 : t - h, hi = t + h > n - 1 ? n - 1 : t + h;
    const int il = (int)(lo - g0) - 1, ih = (int)(hi - g0);
    const double sum = pre[ih] - (il >= 0 ? pre[il] : 0.0);
    const double w = sum / (double)(hi - lo + 1);
    y[t] = w > 0.0 ? (float)((double)x[t] / w) : 0.f;
  }
}

extern "C" int xcx_ram_normalize(xcx_ctx* ctx, float* x, long batch, long n, long half_width, int on_device) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (half_width < 0 || half_width > kRamMaxHalf)
    return set_err(XCX_EUNSUPPORTED, "half_width %ld outside [0, %d]", half_width, kRamMaxHalf);
  HIP_TRY(hipSetDevice(ctx->device));
  Staged st;
  int rc = st.begin(ctx, x, (size_t)batch * n * sizeof(float), on_device);
  if (rc) return rc;
  // the window reaches into neighbouring chunks: read from a copy, write in place
  const size_t bytes = (size_t)batch * n * sizeof(float);
  if ((rc = ctx->scratch[1].ensure(ctx, bytes))) { (void)st.end(); return rc; }
  HIP_TRY(hipMemcpyAsync(ctx->scratch[1].ptr, st.dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  const size_t lds = sizeof(double) * (size_t)(kRamChunk + 2 * half_width);
  static hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ram),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sizeof(double) * (kRamChunk + 2 * kRamMaxHalf)));
  HIP_TRY(attr);
  hipLaunchKernelGGL(k_ram, dim3((unsigned)((n + kRamChunk - 1) / kRamChunk), (unsigned)batch), dim3(NT), lds, ctx->stream,
                     (const float*)ctx->scratch[1].ptr, st.dev, n, (int)half_width);
  HIP_TRY(hipGetLastError());
  return st.end();
}

extern "C" int xcx_snr(xcx_ctx* ctx, const float* x, long batch, long n, const unsigned char* signal_mask,
                       const unsigned char* noise_mask, float* s, float* noise_rms, int* argmax, int on_device) {
  if (!ctx || !x || !signal_mask || !noise_mask || !s || !noise_rms || !argmax)
    return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  const float* dx = x;
  if ((rc = ctx->scratch[0].ensure(ctx, 2 * (size_t)n + (size_t)batch * 12 + 64))) return rc;
  unsigned char* dm = (unsigned char*)ctx->scratch[0].ptr;
  float* ds = (float*)(dm + ((2 * (size_t)n + 15) & ~(size_t)15));
  float* dn = ds + batch;
  int* da = (int*)(dn + batch);
  HIP_TRY(hipMemcpyAsync(dm, signal_mask, n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dm + n, noise_mask, n, hipMemcpyHostToDevice, ctx->stream));
  if (!on_device) {
    if ((rc = ctx->scratch[1].ensure(ctx, sizeof(float) * (size_t)batch * n))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->scratch[1].ptr, x, sizeof(float) * (size_t)batch * n, hipMemcpyHostToDevice, ctx->stream));
    dx = (const float*)ctx->scratch[1].ptr;
  }
  hipLaunchKernelGGL(k_snr, dim3((unsigned)batch), dim3(NT), 0, ctx->stream, dx, n, dm, dm + n, ds, dn, da);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(s, ds, sizeof(float) * batch, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(noise_rms, dn, sizeof(float) * batch, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(argmax, da, sizeof(int) * batch, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return XCX_OK;
}

// ---- measurement aid: what the chip writes ---------------------------------------------------
// The pair pass is output-dominated (one 1.44 MB full-lag CCF per pair-window, written once with
// streaming stores).  xcx_stream_write_rate fills `bytes` of device memory with 16-byte-per-lane
// non-temporal stores -- the access shape of the CCF store -- and returns the elapsed time of `reps`
// passes (HIP events on the context's stream): the ceiling an output-bound pass can reach on this
// chip, reported by bench.py beside the nominal 8 TB/s.
namespace {
// Every wave streams through a contiguous region of its own, 16 KB (sixteen 1 KB instructions) at a time: 5.7-5.8 TB/s.
// (Until round 5 one grid-wide front of 16 bytes per thread swept the buffer: 4.4 TB/s -- the chip writes many separate
// streams faster than one wide front; tools/micro/hol.hip mode 3.  The runtime's own memset, plain stores, reaches
// 6.5 TB/s on a 4.5 GB buffer, profiles/r05c_summary.txt.)
constexpr int kFillWaves = 2048;                              // 8 per CU
__global__ __launch_bounds__(256) void k_stream_fill(float4* __restrict__ dst, size_t n16, float v) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 w = {v, v, v, v};
  f32x4* __restrict__ d = reinterpret_cast<f32x4*>(dst);
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t lane = threadIdx.x & 63;
  const size_t per = ((n16 + kFillWaves - 1) / kFillWaves + 63) & ~(size_t)63;     // 16-byte elements per wave, whole 1 KB runs
  const size_t b = wave * per, e = b + per < n16 ? b + per : n16;
  size_t i = b + lane;
  for (; i + 15 * 64 < e; i += 16 * 64) {
#pragma unroll
    for (int j = 0; j < 16; ++j) __builtin_nontemporal_store(w, d + i + j * 64);
  }
  for (; i < e; i += 64) __builtin_nontemporal_store(w, d + i);
}
}  // namespace

extern "C" int xcx_stream_write_rate(xcx_ctx* ctx, void* dptr, size_t bytes, int reps, float* ms) {
  if (!ctx || !dptr || !ms) return set_err(XCX_EINVAL, "null argument");
  if (bytes < 16 || reps < 1) return set_err(XCX_EINVAL, "bytes >= 16 and reps >= 1 required");
  HIP_TRY(hipSetDevice(ctx->device));
  const size_t n16 = bytes / 16;
  const unsigned grid = kFillWaves / 4;
  hipLaunchKernelGGL(k_stream_fill, dim3(grid), dim3(256), 0, ctx->stream, (float4*)dptr, n16, 0.f);   // warm-up
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(k_stream_fill, dim3(grid), dim3(256), 0, ctx->stream, (float4*)dptr, n16, 0.f);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return XCX_OK;
}
