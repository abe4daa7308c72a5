// fftn.hip -- labelled-FFT kernels of xcorr/signal/fft.py (fft :111-113, ifft :239-241,
// rfft :367-374, irfft :487-494) for ARBITRARY length n (the reference relies on
// pocketfft's Bluestein path; lag axes such as 359 998 = 2 * 179 999 are the rule).
//
// Bluestein on the cross-correlation engine.  With the chirp c[m] = exp(s i pi m^2 / n),
// s = -1 forward / +1 inverse, and mk = (m^2 + k^2 - (k-m)^2) / 2:
//     out[k] = sum_{m < nin} in[m] exp(s 2 pi i m k / n)
//            = c[k] * sum_m (in[m] c[m]) conj(c)[k - m],          k < nout
// The inner sum is a linear convolution of a = in*c (complex, nin points) with conj(c)
// over indices -(nout-1) .. nin-1, i.e. np = nin + nout - 1 points.  It is evaluated by the
// time-stacking pair engine (xcx_cc1d_pairs with XCX_STACK, two "windows" per station):
//     station A_b  = (Re a_b, Im a_b)           one per batch row
//     station C1   = (Re conj c, -Im conj c),  C2 = (Im conj c, Re conj c)
//     stack(A_b, C1) = Re(a (*) conj c),        stack(A_b, C2) = Im(a (*) conj c)
// at lags nin-1 .. nin-1+nout (exactly the nout wanted outputs, in natural order), so one
// inverse transform per real output row; the chirps are evaluated in float64 from the
// exact integer m^2 mod 2n.
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

int xcx_internal_seg_conv_sum(xcx_plan* p, const float* const* xs, const float* const* ks, int count, const int* signs,
                              float* out, long ld);
bool xcx_internal_segmented(xcx_plan* p);

namespace {

struct BsGeom {
  long nin, nout, n, np;   // np = nin + nout - 1
  int sign;                // -1 forward, +1 inverse
  int in_complex;          // 1: input rows are interleaved complex; 2: real rows taken as (even, odd) pairs; 0: real
  int half_weights;        // irfft: double every bin except DC and Nyquist (bin n/2)
  long in_pitch;           // elements (floats or complex) between input rows
  // partial transform (long axes are split): inputs m0 .. m0+nin, outputs k0 .. k0+nout of the
  // length-n transform; rows of `out` are out_pitch elements apart (0: nout); accumulate: out +=
  long m0 = 0, k0 = 0, out_pitch = 0;
  int accumulate = 0;
};

// c[m] = exp(sign i pi m^2 / n), exact argument reduction in integers
__device__ __forceinline__ double2 chirp(long m, long n, int sign) {
  const unsigned long long mm = (unsigned long long)(m < 0 ? -m : m);
  const unsigned long long r = (mm * mm) % (unsigned long long)(2 * n);
  double s, c;
  sincospi((double)r / (double)n, &s, &c);
  return make_double2(c, sign < 0 ? -s : s);
}

// tab[j] = exp(+ i pi j^2 / n), j = 0 .. n: the chirp of a transform length, evaluated once per context and n
// (per element and row it cost a 64-bit modulo and a float64 sincospi: 40 % of a lag-axis rfft)
__global__ void k_chirp_table(double2* __restrict__ tab, long n) {
  const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= n) tab[j] = chirp(j, n, +1);
}

// c[m] from the table: c[-m] = c[m], c[m + n] = (-1)^n c[m]
__device__ __forceinline__ double2 chirp_tab(const double2* __restrict__ tab, long m, long n, int sign) {
  long mm = m < 0 ? -m : m;
  bool flip = false;
  while (mm > n) { mm -= n; flip ^= (bool)(n & 1); }
  double2 c = tab[mm];
  if (flip) { c.x = -c.x; c.y = -c.y; }
  if (sign < 0) c.y = -c.y;
  return c;
}

// traces [B + 2][2][np]; grid (ceil(np / 256), B + 1): row B writes the two chirp stations
__global__ void k_bs_pre(const float* __restrict__ in, BsGeom g, long B, float* __restrict__ tr,
                         const double2* __restrict__ tab) {
  const long np = g.np;
  const long b = blockIdx.y;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= np) return;
  if (b < B) {
    float ar = 0.f, ai = 0.f;
    if (t < g.nin) {
      const long m = g.m0 + t;
      double vr, vi = 0.0;
      if (g.in_complex == 2) {                 // real rows read as (even, odd) sample pairs; pitch in floats
        vr = in[b * g.in_pitch + 2 * m];
        vi = in[b * g.in_pitch + 2 * m + 1];
      } else if (g.in_complex) {
        const float2 v = reinterpret_cast<const float2*>(in)[b * g.in_pitch + m];
        vr = v.x; vi = v.y;
      } else {
        vr = in[b * g.in_pitch + m];
      }
      if (g.half_weights && m != 0 && 2 * m != g.n) { vr *= 2.0; vi *= 2.0; }
      const double2 c = chirp_tab(tab, m, g.n, g.sign);
      ar = (float)(vr * c.x - vi * c.y);
      ai = (float)(vr * c.y + vi * c.x);
    }
    tr[(b * 2 + 0) * np + t] = ar;
    tr[(b * 2 + 1) * np + t] = ai;
  } else {                                    // the two chirp stations
    // conj(c)[m - k] at m - k = (m0 + s) - (k0 + u), s - u = t - (nout-1)
    const double2 c = chirp_tab(tab, t - (g.nout - 1) - (g.k0 - g.m0), g.n, -g.sign);
    tr[((B + 0) * 2 + 0) * np + t] = (float)c.x;
    tr[((B + 0) * 2 + 1) * np + t] = (float)-c.y;
    tr[((B + 1) * 2 + 0) * np + t] = (float)c.y;
    tr[((B + 1) * 2 + 1) * np + t] = (float)c.x;
  }
}

// out[b][k] = c[k] (R[2b][k] + i R[2b+1][k]) scale ; real_only keeps the real part
// out_mode 0: complex, 1: real part, 2: modulus.   grid (ceil(nout / 256), B)
__global__ void k_bs_post(const float* __restrict__ R, long ld, BsGeom g, long B, double scale, int real_only,
                          float* __restrict__ out, const double2* __restrict__ tab) {
  const long b = blockIdx.y;
  const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.nout || b >= B) return;
  const double re = R[(2 * b) * ld + k], im = R[(2 * b + 1) * ld + k];
  const double2 c = chirp_tab(tab, g.k0 + k, g.n, g.sign);
  double orr = (re * c.x - im * c.y) * scale, oi = (re * c.y + im * c.x) * scale;
  const long o = b * (g.out_pitch ? g.out_pitch : g.nout) + g.k0 + k;
  if (real_only == 1) out[o] = (float)(g.accumulate ? (double)out[o] + orr : orr);
  else if (real_only == 2) out[o] = (float)sqrt(orr * orr + oi * oi);
  else {
    float2* oc = reinterpret_cast<float2*>(out) + o;
    if (g.accumulate) { orr += (double)oc->x; oi += (double)oc->y; }
    *oc = make_float2((float)orr, (float)oi);
  }
}

constexpr long kBsMaxNp = XCX_MAX_CHIRP_POINTS;                    // np <= N of the longest tuned correlation plan (kPlanShortOperand)

int bluestein_core(xcx_ctx* ctx, const float* in, float* out, long batch, BsGeom g, double scale, int real_only,
                   int in_dev, int out_dev) {
  if (!ctx || !in || !out) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || g.n < 2) return set_err(XCX_EINVAL, "batch >= 1 and n >= 2 required");
  HIP_TRY(hipSetDevice(ctx->device));
  g.np = g.nin + g.nout - 1;
  if (g.np > kBsMaxNp) return set_err(XCX_EUNSUPPORTED, "transform length %ld too long for the Bluestein path", g.n);
  int rc;
  const long imin = g.nin - 1, imax = g.nin - 1 + g.nout;
  int slot = -1;
  for (int i = 0; i < 2; ++i)
    if (ctx->bs_plan[i] && ctx->bs_key[i][0] == g.np && ctx->bs_key[i][1] == imin && ctx->bs_key[i][2] == imax) slot = i;
  if (slot < 0) {                                   // two cached plans, replaced in turn
    slot = ctx->bs_next;
    ctx->bs_next ^= 1;
    if (ctx->bs_plan[slot]) { xcx_plan_destroy(ctx->bs_plan[slot]); ctx->bs_plan[slot] = nullptr; }
    // station A is zero beyond its first nin samples and the retained lags are -(nout-1) .. 0: a transform of
    // N >= np points is alias-free for them (kPlanShortOperand), not np + max|lag|
    if ((rc = xcx_plan_cc1d(ctx, (int)g.np, 0, (int)imin, (int)imax, XCX_STACK | kPlanShortOperand, &ctx->bs_plan[slot]))) return rc;
    ctx->bs_key[slot][0] = g.np; ctx->bs_key[slot][1] = imin; ctx->bs_key[slot][2] = imax;
  }
  xcx_plan* plan = ctx->bs_plan[slot];
  if (ctx->bs_tab_n != g.n || !ctx->bs_tab.ptr) {
    ctx->bs_tab_n = 0;
    if ((rc = ctx->bs_tab.ensure(ctx, sizeof(double2) * (size_t)(g.n + 1)))) return rc;
    hipLaunchKernelGGL(k_chirp_table, dim3((unsigned)((g.n + 256) / 256)), dim3(256), 0, ctx->stream, (double2*)ctx->bs_tab.ptr, g.n);
    HIP_TRY(hipGetLastError());
    ctx->bs_tab_n = g.n;
  }
  const double2* tab = (const double2*)ctx->bs_tab.ptr;
  const long ld = (g.nout + 31) / 32 * 32;
  // rows per pass: bounded workspace (traces + spectra scale with np)
  const long chunk = std::max<long>(1, std::min<long>(std::min<long>(batch, 256), ((long)1024 << 20) / (16 * g.np)));
  DevBuf &tr = ctx->scratch[1], &lin = ctx->scratch[2], &din = ctx->scratch[4], &dout = ctx->scratch[5];
  if ((rc = tr.ensure(ctx, sizeof(float) * 2 * g.np * (chunk + 2)))) return rc;
  if ((rc = lin.ensure(ctx, sizeof(float) * ld * 2 * chunk))) return rc;
  const size_t in_row = (size_t)g.in_pitch * (g.in_complex == 1 ? 8 : 4);
  const size_t out_row = (size_t)(g.out_pitch ? g.out_pitch : g.nout) * (real_only ? 4 : 8);
  const bool stage_in = !in_dev, stage_out = !out_dev;
  HostPin pin_in(stage_in ? in : nullptr, in_row * (size_t)batch);          // host rows: DMA at link speed
  HostPin pin_out(stage_out ? out : nullptr, out_row * (size_t)batch);
  const float* ind = in;
  float* outd = out;
  if (stage_in) {
    if ((rc = din.ensure(ctx, in_row * batch))) return rc;
    HIP_TRY(hipMemcpyAsync(din.ptr, in, in_row * batch, hipMemcpyHostToDevice, ctx->stream));
    ind = (const float*)din.ptr;
  }
  if (stage_out) {
    if ((rc = dout.ensure(ctx, out_row * batch))) return rc;
    outd = (float*)dout.ptr;
  }
  std::vector<int> pairs;
  for (long r0 = 0; r0 < batch; r0 += chunk) {
    const long cnt = std::min(chunk, batch - r0);
    hipLaunchKernelGGL(k_bs_pre, dim3((unsigned)((g.np + 255) / 256), (unsigned)(cnt + 1)), dim3(256), 0, ctx->stream,
                       (const float*)((const char*)ind + in_row * r0), g, cnt, (float*)tr.ptr, tab);
    HIP_TRY(hipGetLastError());
    pairs.resize(4 * cnt);
    for (long b = 0; b < cnt; ++b) {
      pairs[4 * b + 0] = (int)b; pairs[4 * b + 1] = (int)cnt;
      pairs[4 * b + 2] = (int)b; pairs[4 * b + 3] = (int)cnt + 1;
    }
    if ((rc = xcx_cc1d_pairs(plan, (const float*)tr.ptr, (int)cnt + 2, 2, 2 * g.np, g.np, pairs.data(), (int)(2 * cnt),
                             (float*)lin.ptr, ld, 1)))
      return rc;
    hipLaunchKernelGGL(k_bs_post, dim3((unsigned)((g.nout + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream,
                       (const float*)lin.ptr, ld, g, cnt, scale, real_only, (float*)((char*)outd + out_row * r0), tab);
    HIP_TRY(hipGetLastError());
  }
  if (stage_out) {
    HIP_TRY(hipMemcpyAsync(out, outd, out_row * batch, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return XCX_OK;
}

// Axes whose chirp convolution exceeds every single transform (day-long rows: rfft of 4.32 M samples = a complex
// transform of 2.16 M points) as ONE block-decomposed convolution per real product (xcx_api.hip: seg_run's convolution
// mode, the machinery of the long time shifts): rows a = in * c as (Re, Im) of nn = max(nin, nout) samples, the chirp
// kernel over TWO spans, g[j] = conj(c)[nn - 1 - j + k0 - m0] for j < 2 nn - 1, as (Re, Im) -- shared by every row and
// transformed once per call; the four real convolutions (Re a, Im a) x (Re g, Im g) of a row share its block
// transforms and are summed in twos in the frequency domain (Re = ar*gr - ai*gi, Im = ar*gi + ai*gr: k_seg_gram_sum,
// one inverse transform per chunk and part).  Before (round 1-3): si x so partial transforms of <= 737 280 points, each two full correlations --
// 36 of them for a day-long rfft, 848 us per row.
// rows: tr[b][2][nn];  kernels: kr[2][2 nn - 1].   grid (ceil((2 nn - 1) / 256), B + 1): row B writes the kernels
__global__ void k_bs_pre_seg(const float* __restrict__ in, BsGeom g, long nn, long B, float* __restrict__ tr,
                             float* __restrict__ kr, const double2* __restrict__ tab) {
  const long b = blockIdx.y;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) {
    if (t >= nn) return;
    float ar = 0.f, ai = 0.f;
    if (t < g.nin) {
      const long m = g.m0 + t;
      double vr, vi = 0.0;
      if (g.in_complex == 2) {
        vr = in[b * g.in_pitch + 2 * m];
        vi = in[b * g.in_pitch + 2 * m + 1];
      } else if (g.in_complex) {
        const float2 v = reinterpret_cast<const float2*>(in)[b * g.in_pitch + m];
        vr = v.x; vi = v.y;
      } else {
        vr = in[b * g.in_pitch + m];
      }
      if (g.half_weights && m != 0 && 2 * m != g.n) { vr *= 2.0; vi *= 2.0; }
      const double2 c = chirp_tab(tab, m, g.n, g.sign);
      ar = (float)(vr * c.x - vi * c.y);
      ai = (float)(vr * c.y + vi * c.x);
    }
    tr[(b * 2 + 0) * nn + t] = ar;
    tr[(b * 2 + 1) * nn + t] = ai;
  } else {
    const long L = 2 * nn - 1;
    if (t >= L) return;
    const double2 c = chirp_tab(tab, nn - 1 - t + (g.k0 - g.m0), g.n, -g.sign);
    kr[t] = (float)c.x;
    kr[L + t] = (float)c.y;
  }
}

int bluestein_seg(xcx_ctx* ctx, const float* in, float* out, long batch, BsGeom g, double scale, int real_only,
                  int in_dev, int out_dev) {
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  const long nn = std::max(g.nin, g.nout), L = 2 * nn - 1;
  if (nn > XCX_MAX_ROW_SAMPLES) return set_err(XCX_EUNSUPPORTED, "transform length %ld too long", g.n);
  int slot = -1;
  for (int i = 0; i < 2; ++i) if (ctx->bs_seg_plan[i] && ctx->bs_seg_n[i] == nn) slot = i;
  if (slot < 0) {                                   // two cached plans, replaced in turn
    slot = ctx->bs_seg_next;
    ctx->bs_seg_next ^= 1;
    if (ctx->bs_seg_plan[slot]) { xcx_plan_destroy(ctx->bs_seg_plan[slot]); ctx->bs_seg_plan[slot] = nullptr; }
    ctx->bs_seg_n[slot] = 0;
    if ((rc = xcx_plan_cc1d(ctx, (int)nn, 0, 0, 0, 0, &ctx->bs_seg_plan[slot]))) return rc;
    ctx->bs_seg_n[slot] = nn;
  }
  xcx_plan* plan = ctx->bs_seg_plan[slot];
  if (!xcx_internal_segmented(plan)) return set_err(XCX_EUNSUPPORTED, "bluestein_seg: %ld points fit one transform", nn);
  if (ctx->bs_tab_n != g.n || !ctx->bs_tab.ptr) {
    ctx->bs_tab_n = 0;
    if ((rc = ctx->bs_tab.ensure(ctx, sizeof(double2) * (size_t)(g.n + 1)))) return rc;
    hipLaunchKernelGGL(k_chirp_table, dim3((unsigned)((g.n + 256) / 256)), dim3(256), 0, ctx->stream, (double2*)ctx->bs_tab.ptr, g.n);
    HIP_TRY(hipGetLastError());
    ctx->bs_tab_n = g.n;
  }
  const double2* tab = (const double2*)ctx->bs_tab.ptr;
  const long chunk = std::max<long>(1, std::min<long>(batch, ((long)2048 << 20) / (16 * nn)));     // 4 nn floats per row
  DevBuf &tr = ctx->scratch[1], &lin = ctx->scratch[2], &din = ctx->scratch[4], &dout = ctx->scratch[5];
  if ((rc = tr.ensure(ctx, sizeof(float) * (2 * nn * chunk + 2 * L)))) return rc;          // rows, then the two kernel rows
  if ((rc = lin.ensure(ctx, sizeof(float) * 2 * nn * chunk))) return rc;
  float* krp = (float*)tr.ptr + 2 * nn * chunk;
  const size_t in_row = (size_t)g.in_pitch * (g.in_complex == 1 ? 8 : 4);
  const size_t out_row = (size_t)(g.out_pitch ? g.out_pitch : g.nout) * (real_only ? 4 : 8);
  HostPin pin_in(in_dev ? nullptr : in, in_row * (size_t)batch);
  HostPin pin_out(out_dev ? nullptr : out, out_row * (size_t)batch);
  const float* ind = in;
  float* outd = out;
  if (!in_dev) {
    if ((rc = din.ensure(ctx, in_row * batch))) return rc;
    HIP_TRY(hipMemcpyAsync(din.ptr, in, in_row * batch, hipMemcpyHostToDevice, ctx->stream));
    ind = (const float*)din.ptr;
  }
  if (!out_dev) {
    if ((rc = dout.ensure(ctx, out_row * batch))) return rc;
    outd = (float*)dout.ptr;
  }
  std::vector<const float*> xs, ks;
  std::vector<int> signs;
  const float* gr = krp;
  const float* gi = gr + L;
  for (long r0 = 0; r0 < batch; r0 += chunk) {
    const long cnt = std::min(chunk, batch - r0);
    hipLaunchKernelGGL(k_bs_pre_seg, dim3((unsigned)((L + 255) / 256), (unsigned)(cnt + 1)), dim3(256), 0, ctx->stream,
                       (const float*)((const char*)ind + in_row * r0), g, nn, cnt, (float*)tr.ptr, krp, tab);
    HIP_TRY(hipGetLastError());
    xs.resize(4 * cnt); ks.resize(4 * cnt);
    for (long b = 0; b < cnt; ++b) {
      const float* ar = (const float*)tr.ptr + (2 * b) * nn;
      const float* ai = ar + nn;
      xs[4 * b + 0] = ar; ks[4 * b + 0] = gr;           // Re = ar*gr - ai*gi
      xs[4 * b + 1] = ai; ks[4 * b + 1] = gi;
      xs[4 * b + 2] = ar; ks[4 * b + 2] = gi;           // Im = ar*gi + ai*gr
      xs[4 * b + 3] = ai; ks[4 * b + 3] = gr;
    }
    // the two products of a real (imaginary) part are summed in the frequency domain: one inverse per chunk and part
    signs.resize(2 * cnt);
    for (long b = 0; b < cnt; ++b) { signs[2 * b] = -1; signs[2 * b + 1] = +1; }
    if ((rc = xcx_internal_seg_conv_sum(plan, xs.data(), ks.data(), (int)(4 * cnt), signs.data(), (float*)lin.ptr, nn))) return rc;
    hipLaunchKernelGGL(k_bs_post, dim3((unsigned)((g.nout + 255) / 256), (unsigned)cnt), dim3(256), 0, ctx->stream,
                       (const float*)lin.ptr, nn, g, cnt, scale, real_only, (float*)((char*)outd + out_row * r0), tab);
    HIP_TRY(hipGetLastError());
  }
  if (!out_dev) HIP_TRY(hipMemcpyAsync(out, outd, out_row * batch, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return XCX_OK;
}

// Any length: an axis whose chirp convolution (nin + nout - 1 points) exceeds the longest
// correlation plan is split into si input blocks x so output blocks,
//     out[k0 + u] = sum over input blocks of  c[k] sum_t (in[m0+t] c[m0+t]) conj(c)[(k0+u) - (m0+t)],
// each a partial transform of ceil(nin/si) + ceil(nout/so) - 1 points accumulated into `out`
// (the full-lag axis 359 998: rfft = 2 x 1, hilbert = 2 x 1 forward and 1 x 2 back, fft = 2 x 2).
int bluestein(xcx_ctx* ctx, const float* in, float* out, long batch, BsGeom g, double scale, int real_only,
              int in_dev, int out_dev) {
  if (!ctx || !in || !out) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || g.n < 2) return set_err(XCX_EINVAL, "batch >= 1 and n >= 2 required");
  if (g.nin + g.nout - 1 <= kBsMaxNp) return bluestein_core(ctx, in, out, batch, g, scale, real_only, in_dev, out_dev);
  if (2 * std::max(g.nin, g.nout) - 1 > kMaxTransformN)      // beyond every single transform: block convolution
    return bluestein_seg(ctx, in, out, batch, g, scale, real_only, in_dev, out_dev);
  if (real_only == 2) return set_err(XCX_EINVAL, "modulus output of a split transform goes through a complex buffer");
  long si = 1, so = 1;
  auto np_of = [&](long a, long b) { return (g.nin + a - 1) / a + (g.nout + b - 1) / b - 1; };
  while (np_of(si, so) > kBsMaxNp) {
    if ((g.nin + si - 1) / si >= (g.nout + so - 1) / so) ++si; else ++so;
    if (si * so > XCX_MAX_SPLIT_PARTS) return set_err(XCX_EUNSUPPORTED, "transform length %ld needs more than %ld partial transforms", g.n, XCX_MAX_SPLIT_PARTS);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  int rc;
  const size_t in_row = (size_t)g.in_pitch * (g.in_complex == 1 ? 8 : 4);
  const size_t out_row = (size_t)(g.out_pitch ? g.out_pitch : g.nout) * (real_only ? 4 : 8);
  void *din = nullptr, *dout = nullptr;                     // own buffers: the core uses the context scratch
  auto done = [&](int r) { (void)hipFree(din); (void)hipFree(dout); return r; };
  const float* ind = in; float* outd = out;
  if (!in_dev) {
    if (ctx_malloc(ctx, &din, in_row * batch) != hipSuccess) return done(set_err(XCX_ENOMEM, "split transform input"));
    if (hipMemcpyAsync(din, in, in_row * batch, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return done(set_err(XCX_EHIP, "H2D"));
    ind = (const float*)din;
  }
  if (!out_dev) {
    if (ctx_malloc(ctx, &dout, out_row * batch) != hipSuccess) return done(set_err(XCX_ENOMEM, "split transform output"));
    outd = (float*)dout;
  }
  const long ci = (g.nin + si - 1) / si, co = (g.nout + so - 1) / so;
  for (long k0 = 0; k0 < g.nout; k0 += co)
    for (long m0 = 0; m0 < g.nin; m0 += ci) {
      BsGeom p = g;
      p.nin = std::min(ci, g.nin - m0); p.nout = std::min(co, g.nout - k0);
      p.m0 = m0; p.k0 = k0; p.out_pitch = g.out_pitch ? g.out_pitch : g.nout; p.accumulate = m0 > 0;
      if ((rc = bluestein_core(ctx, ind, outd, batch, p, scale, real_only, 1, 1))) return done(rc);
    }
  if (!out_dev) {
    if (hipMemcpyAsync(out, outd, out_row * batch, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return done(set_err(XCX_EHIP, "D2H"));
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(set_err(XCX_EHIP, "split transform sync"));
  return done(XCX_OK);
}

__global__ void k_modulus(const float2* __restrict__ z, float* __restrict__ out, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const double a = z[i].x, b = z[i].y;
    out[i] = (float)sqrt(a * a + b * b);
  }
}

}  // namespace

extern "C" int xcx_fft(xcx_ctx* ctx, const float* in, float* out, long batch, long n, int on_device) {
  BsGeom g{n, n, n, 0, -1, 1, 0, n};
  return bluestein(ctx, in, out, batch, g, 1.0, 0, on_device, on_device);
}

extern "C" int xcx_ifft(xcx_ctx* ctx, const float* in, float* out, long batch, long n, int on_device) {
  BsGeom g{n, n, n, 0, +1, 1, 0, n};
  return bluestein(ctx, in, out, batch, g, n > 0 ? 1.0 / (double)n : 0.0, 0, on_device, on_device);
}

// ---------------------------------------------------------------------------
// Real transforms of EVEN length n = 2m through ONE complex transform of length m (the lag axis
// 359 998 = 2 x 179 999: a chirp convolution of 2m - 1 = 359 997 points fits the tuned N = 368 640
// plan, where the unpacked real transform needed 540 k points, i.e. two partial transforms):
//   z[j] = x[2j] + i x[2j+1],  Z = DFT_m(z),  E = DFT_m(x_even) = (Z[k] + conj Z[m-k]) / 2,
//   O = DFT_m(x_odd) = (Z[k] - conj Z[m-k]) / 2i,  X[k] = E[k] + w^k O[k],  w = exp(-2 pi i / n),
// and back:  E = (X[k] + conj X[m-k]) / 2,  O = w^-k (X[k] - conj X[m-k]) / 2,  Z = E + i O,
// z = IDFT_m(Z): the n real samples are the interleaved (re, im) of z.
// ---------------------------------------------------------------------------
namespace {
__global__ void k_rfft_unpack(const float2* __restrict__ Z, long m, float2* __restrict__ X, long batch) {
  const long nh = m + 1, total = batch * nh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / nh, k = i % nh;
    const float2 a = Z[b * m + (k == m ? 0 : k)];
    const float2 c = Z[b * m + ((k == 0 || k == m) ? 0 : m - k)];
    const double er = 0.5 * ((double)a.x + c.x), ei = 0.5 * ((double)a.y - c.y);        // E = (a + conj c) / 2
    const double orr = 0.5 * ((double)a.y + c.y), oi = -0.5 * ((double)a.x - c.x);     // O = (a - conj c) / 2i
    double s, co;
    sincospi(-(double)k / (double)m, &s, &co);                                         // w^k = exp(-i pi k / m)
    X[i] = make_float2((float)(er + co * orr - s * oi), (float)(ei + co * oi + s * orr));
  }
}

// X rows [batch][in_pitch] (bins 0..m used; imaginary parts of bins 0 and m ignored, as numpy.fft.irfft does)
__global__ void k_irfft_pack(const float2* __restrict__ X, long in_pitch, long m, float2* __restrict__ Z, long batch) {
  const long total = batch * m;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / m, k = i % m;
    float2 a = X[b * in_pitch + k], c = X[b * in_pitch + m - k];
    if (k == 0) { a.y = 0.f; c.y = 0.f; }
    const double er = 0.5 * ((double)a.x + c.x), ei = 0.5 * ((double)a.y - c.y);        // E = (a + conj c) / 2
    const double dr = 0.5 * ((double)a.x - c.x), di = 0.5 * ((double)a.y + c.y);        // (a - conj c) / 2
    double s, co;
    sincospi((double)k / (double)m, &s, &co);                                          // w^-k
    const double orr = co * dr - s * di, oi = co * di + s * dr;                        // O
    Z[i] = make_float2((float)(er - oi), (float)(ei + orr));                           // E + i O
  }
}

int launch_rows(xcx_ctx* ctx, long total) { (void)ctx; return (int)std::min<long>((total + 255) / 256, 256L * 16); }

// rfft of even n: x rows [batch][pitch floats] (device or host) -> X rows [batch][n/2+1] complex
int rfft_even(xcx_ctx* ctx, const float* x, long pitch, float* out, long batch, long n, int in_dev, int out_dev) {
  const long m = n / 2;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf &zb = ctx->scratch[8], &xb = ctx->scratch[9];
  int rc;
  if ((rc = zb.ensure(ctx, sizeof(float2) * m * batch))) return rc;
  BsGeom g{m, m, m, 0, -1, 2, 0, pitch};
  if ((rc = bluestein(ctx, x, (float*)zb.ptr, batch, g, 1.0, 0, in_dev, 1))) return rc;
  float* dst = out;
  if (!out_dev) {
    if ((rc = xb.ensure(ctx, sizeof(float2) * (m + 1) * batch))) return rc;
    dst = (float*)xb.ptr;
  }
  hipLaunchKernelGGL(k_rfft_unpack, dim3(launch_rows(ctx, batch * (m + 1))), dim3(256), 0, ctx->stream,
                     (const float2*)zb.ptr, m, (float2*)dst, batch);
  HIP_TRY(hipGetLastError());
  if (!out_dev) {
    HIP_TRY(hipMemcpyAsync(out, dst, sizeof(float2) * (m + 1) * batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return XCX_OK;
}

// irfft to even n = 2m: X rows [batch][in_pitch complex] -> n real samples per row, scaled by `scale` * n
// (scale = 1/n gives numpy.fft.irfft)
int irfft_even(xcx_ctx* ctx, const float* in, long in_pitch, float* x, long batch, long n, double scale, int in_dev, int out_dev) {
  const long m = n / 2;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf &zb = ctx->scratch[8], &xb = ctx->scratch[9];
  int rc;
  if ((rc = zb.ensure(ctx, sizeof(float2) * m * batch))) return rc;
  const float* src = in;
  if (!in_dev) {
    if ((rc = xb.ensure(ctx, sizeof(float2) * in_pitch * batch))) return rc;
    HIP_TRY(hipMemcpyAsync(xb.ptr, in, sizeof(float2) * in_pitch * batch, hipMemcpyHostToDevice, ctx->stream));
    src = (const float*)xb.ptr;
  }
  hipLaunchKernelGGL(k_irfft_pack, dim3(launch_rows(ctx, batch * m)), dim3(256), 0, ctx->stream,
                     (const float2*)src, in_pitch, m, (float2*)zb.ptr, batch);
  HIP_TRY(hipGetLastError());
  // z = IDFT_m(Z) * 2 scale: x[2j] + i x[2j+1] = (1/m) sum Z e^{+}, and numpy's 1/n = 1/(2m)
  BsGeom g{m, m, m, 0, +1, 1, 0, m};
  return bluestein(ctx, (const float*)zb.ptr, x, batch, g, 2.0 * scale, 0, 1, out_dev);
}
}  // namespace

extern "C" int xcx_rfft(xcx_ctx* ctx, const float* x, float* out, long batch, long size, int on_device) {
  if (!ctx || !x || !out) return set_err(XCX_EINVAL, "null argument");
  const long n = size - (size & 1);              // fft.py:357 drops the last sample of an odd axis
  if (batch < 1 || n < 2) return set_err(XCX_EINVAL, "batch >= 1 and n >= 2 required");
  if (n >= 4) return rfft_even(ctx, x, size, out, batch, n, on_device, on_device);
  BsGeom g{n, n / 2 + 1, n, 0, -1, 0, 0, size};
  return bluestein(ctx, x, out, batch, g, 1.0, 0, on_device, on_device);
}

extern "C" int xcx_irfft(xcx_ctx* ctx, const float* in, float* x, long batch, long m, int on_device) {
  if (!ctx || !in || !x) return set_err(XCX_EINVAL, "null argument");
  const long n = 2 * m - 2;                       // fft.py:475
  if (batch < 1 || n < 2) return set_err(XCX_EINVAL, "batch >= 1 and n >= 2 required");
  if (n >= 4) return irfft_even(ctx, in, m, x, batch, n, 1.0 / (double)n, on_device, on_device);
  BsGeom g{m, n, n, 0, +1, 1, 1, m};
  return bluestein(ctx, in, x, batch, g, n > 0 ? 1.0 / (double)n : 0.0, 1, on_device, on_device);
}

// ---------------------------------------------------------------------------
// Fourier resampling, ObsPy Trace.resample(sampling_rate, window, no_filter=True) semantics
// (registry entry 'resample' of xcorr/stream/process.py:29-94; documented behaviour restated,
// unverified vs obspy):  X = rfft(x) (all npts samples) ; X *= window[0 .. npts/2] ;
// Y[j] = linear interpolation of X at the frequency j * new_rate / num (np.interp, clamped) ;
// y = irfft(Y, num) * num / npts,   num = int(npts / factor).
// ---------------------------------------------------------------------------
namespace {
__global__ void k_resample_spec(const float2* __restrict__ X, long nh, const float* __restrict__ win,
                                float2* __restrict__ Y, long mh, double ratio, long batch) {
  const long total = batch * mh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / mh, j = i % mh;
    const double p = (double)j * ratio;
    long k0 = (long)floor(p);
    const float2* __restrict__ row = X + b * nh;
    double re, im;
    if (k0 >= nh - 1) {
      const double w = win ? (double)win[nh - 1] : 1.0;
      re = (double)row[nh - 1].x * w; im = (double)row[nh - 1].y * w;
    } else {
      const double fr = p - (double)k0;
      const double w0 = win ? (double)win[k0] : 1.0, w1 = win ? (double)win[k0 + 1] : 1.0;
      const double r0 = (double)row[k0].x * w0, i0 = (double)row[k0].y * w0;
      const double r1 = (double)row[k0 + 1].x * w1, i1 = (double)row[k0 + 1].y * w1;
      re = r0 + (r1 - r0) * fr; im = i0 + (i1 - i0) * fr;
    }
    Y[i] = make_float2((float)re, (float)im);
  }
}
}  // namespace

extern "C" int xcx_resample(xcx_ctx* ctx, const float* x, float* y, long batch, long npts, long num,
                            double ratio, const float* window_half, int on_device) {
  if (!ctx || !x || !y) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || npts < 2 || num < 2) return set_err(XCX_EINVAL, "batch >= 1, npts >= 2 and num >= 2 required");
  if (!(ratio > 0)) return set_err(XCX_EINVAL, "ratio must be positive");
  HIP_TRY(hipSetDevice(ctx->device));
  const long nh = npts / 2 + 1, mh = num / 2 + 1;
  DevBuf &sx = ctx->scratch[6], &sy = ctx->scratch[7], &dwin = ctx->scratch[3];
  int rc;
  if ((rc = sx.ensure(ctx, sizeof(float2) * nh * batch))) return rc;
  if ((rc = sy.ensure(ctx, sizeof(float2) * mh * batch))) return rc;
  const float* dw = nullptr;
  if (window_half) {                                           // host table in both modes
    if ((rc = dwin.ensure(ctx, sizeof(float) * nh))) return rc;
    HIP_TRY(hipMemcpyAsync(dwin.ptr, window_half, sizeof(float) * nh, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    dw = (const float*)dwin.ptr;
  }
  // rfft of all npts samples: even lengths as ONE complex transform of half the length (rfft_even)
  if (npts % 2 == 0 && npts >= 4) {
    if ((rc = rfft_even(ctx, x, npts, (float*)sx.ptr, batch, npts, on_device, 1))) return rc;
  } else {
    BsGeom gf{npts, nh, npts, 0, -1, 0, 0, npts};
    if ((rc = bluestein(ctx, x, (float*)sx.ptr, batch, gf, 1.0, 0, on_device, 1))) return rc;
  }
  const long total = batch * mh;
  hipLaunchKernelGGL(k_resample_spec, dim3((unsigned)std::min<long>((total + 255) / 256, 256L * 16)), dim3(256), 0,
                     ctx->stream, (const float2*)sx.ptr, nh, dw, (float2*)sy.ptr, mh, ratio, batch);
  HIP_TRY(hipGetLastError());
  // irfft to num samples, times num / npts: even num through the packed form as well
  if (num % 2 == 0 && num >= 4) return irfft_even(ctx, (const float*)sy.ptr, mh, y, batch, num, 1.0 / (double)npts, 1, on_device);
  BsGeom gi{mh, num, num, 0, +1, 1, 1, mh};
  return bluestein(ctx, (const float*)sy.ptr, y, batch, gi, 1.0 / (double)npts, 1, 1, on_device);
}

// scipy.signal.hilbert along the last axis (xcorr/signal/hilbert.py:69, envelope :120):
// one-sided spectrum (bins 0..n/2, doubled except DC and, for even n, Nyquist) transformed back
namespace {
// S[k] <- -i S[k] for 0 < k < m, S[0] = S[m] = 0: irfft of it is the Hilbert transform of the row
__global__ void k_hilbert_spec(float2* __restrict__ S, long m, long batch) {
  const long nh = m + 1, total = batch * nh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long k = i % nh;
    const float2 v = S[i];
    S[i] = (k == 0 || k == m) ? make_float2(0.f, 0.f) : make_float2(v.y, -v.x);
  }
}
// out = x + i H (complex) or |x + i H| (envelope)
__global__ void k_hilbert_out(const float* __restrict__ x, const float* __restrict__ H, long total, int envelope,
                              float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float a = x[i], b = H[i];
    if (envelope) out[i] = (float)sqrt((double)a * a + (double)b * b);
    else reinterpret_cast<float2*>(out)[i] = make_float2(a, b);
  }
}

// even n: two packed real transforms (rfft_even, irfft_even) instead of an unpacked forward transform and a
// complex inverse of length n; the real part of the analytic signal is the input itself
int hilbert_even(xcx_ctx* ctx, const float* x, float* out, long batch, long n, int envelope, int on_device) {
  const long m = n / 2, nh = m + 1;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf &spec = ctx->scratch[6], &hb = ctx->scratch[7], &xin = ctx->scratch[3];
  int rc;
  const long rows = std::max<long>(1, std::min<long>(batch, ((long)512 << 20) / (8 * nh)));
  if ((rc = spec.ensure(ctx, sizeof(float2) * nh * rows))) return rc;
  if ((rc = hb.ensure(ctx, sizeof(float) * n * rows))) return rc;
  if (!on_device && (rc = xin.ensure(ctx, sizeof(float) * n * rows * (envelope ? 2 : 3)))) return rc;
  const size_t orow = (size_t)n * (envelope ? 4 : 8);
  for (long r0 = 0; r0 < batch; r0 += rows) {
    const long cnt = std::min(rows, batch - r0);
    const float* xd = x + r0 * n;
    float* od = (float*)((char*)out + orow * r0);
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync(xin.ptr, x + r0 * n, sizeof(float) * n * cnt, hipMemcpyHostToDevice, ctx->stream));
      xd = (const float*)xin.ptr;
      od = (float*)xin.ptr + n * rows;
    }
    if ((rc = rfft_even(ctx, xd, n, (float*)spec.ptr, cnt, n, 1, 1))) return rc;
    hipLaunchKernelGGL(k_hilbert_spec, dim3(launch_rows(ctx, cnt * nh)), dim3(256), 0, ctx->stream, (float2*)spec.ptr, m, cnt);
    HIP_TRY(hipGetLastError());
    if ((rc = irfft_even(ctx, (const float*)spec.ptr, nh, (float*)hb.ptr, cnt, n, 1.0 / (double)n, 1, 1))) return rc;
    hipLaunchKernelGGL(k_hilbert_out, dim3(launch_rows(ctx, cnt * n)), dim3(256), 0, ctx->stream, xd, (const float*)hb.ptr,
                       cnt * n, envelope, od);
    HIP_TRY(hipGetLastError());
    if (!on_device) {
      HIP_TRY(hipMemcpyAsync((char*)out + orow * r0, od, orow * cnt, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
  }
  return XCX_OK;
}
}  // namespace

extern "C" int xcx_hilbert(xcx_ctx* ctx, const float* x, float* out, long batch, long n, int envelope,
                           int on_device) {
  if (!ctx || !x || !out) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 2) return set_err(XCX_EINVAL, "batch >= 1 and n >= 2 required");
  if (n % 2 == 0 && n >= 4) return hilbert_even(ctx, x, out, batch, n, envelope, on_device);
  const long nh = n / 2 + 1;
  DevBuf& spec = ctx->scratch[6];
  int rc;
  const long rows = std::max<long>(1, std::min<long>(batch, ((long)512 << 20) / (8 * nh)));
  if ((rc = spec.ensure(ctx, sizeof(float) * 2 * nh * rows))) return rc;
  const size_t orow = (size_t)n * (envelope ? 4 : 8);
  for (long r0 = 0; r0 < batch; r0 += rows) {
    const long cnt = std::min(rows, batch - r0);
    BsGeom f{n, nh, n, 0, -1, 0, 0, n};
    if ((rc = bluestein(ctx, x + r0 * n, (float*)spec.ptr, cnt, f, 1.0, 0, on_device, 1))) return rc;
    BsGeom b{nh, n, n, 0, +1, 1, 1, nh};
    if (envelope && nh + n - 1 > kBsMaxNp) {
      // split transform: the partial sums are complex, the modulus comes last
      void *dz = nullptr, *dm = nullptr;
      auto done = [&](int r) { (void)hipFree(dz); (void)hipFree(dm); return r; };
      if (ctx_malloc(ctx, &dz, sizeof(float2) * n * cnt) != hipSuccess) return done(set_err(XCX_ENOMEM, "analytic signal buffer"));
      if ((rc = bluestein(ctx, (const float*)spec.ptr, (float*)dz, cnt, b, 1.0 / (double)n, 0, 1, 1))) return done(rc);
      float* dst = (float*)((char*)out + orow * r0);
      if (!on_device) {
        if (ctx_malloc(ctx, &dm, sizeof(float) * n * cnt) != hipSuccess) return done(set_err(XCX_ENOMEM, "envelope buffer"));
        dst = (float*)dm;
      }
      const long tot = n * cnt;
      hipLaunchKernelGGL(k_modulus, dim3((unsigned)std::min<long>((tot + 255) / 256, 256L * 16)), dim3(256), 0, ctx->stream,
                         (const float2*)dz, dst, tot);
      if (hipGetLastError() != hipSuccess) return done(set_err(XCX_EHIP, "k_modulus launch"));
      if (!on_device && hipMemcpyAsync((char*)out + orow * r0, dm, sizeof(float) * tot, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        return done(set_err(XCX_EHIP, "D2H"));
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) return done(set_err(XCX_EHIP, "envelope sync"));
      (void)done(0);
      continue;
    }
    if ((rc = bluestein(ctx, (const float*)spec.ptr, (float*)((char*)out + orow * r0), cnt, b, 1.0 / (double)n,
                        envelope ? 2 : 0, 1, on_device)))
      return rc;
  }
  return XCX_OK;
}
