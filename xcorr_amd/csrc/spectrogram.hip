// spectrogram.hip -- hop-1 spectrogram of [batch][n] float32 rows, the kernel of
// xcorr/signal/spectrogram.py:124-141 (scipy.signal.spectrogram with nperseg = win_len,
// noverlap = win_len-1, nfft = win_len * padding_factor, mode 'psd', one-sided; zero
// columns at the outer edges) -- the producer of the tiles that correlate2d consumes
// (scripts/timelapse.py:207-208: duration 2.5 s -> win_len 125, nfft 500, 251 bins).
//
// With one segment per sample every output column is a short DFT of a sliding window:
//     Y[f][t] = sum_{m < win_len} D[f][m] x[t + m],  D[f][m] = w[m] exp(-2 pi i f m / nfft)
// i.e. the product of the constant (2 nfreq x win_len) matrix D with the Hankel matrix of
// the row: dense GEMM work -> fp32 MFMA (v_mfma_f32_32x32x2_f32, exact f32 accumulate).
//   * M: 32-row tiles = 16 frequencies x (re, im), rows ordered so that re and im of one
//     frequency land in the SAME lane of the 32x32 accumulator (rows r and r+8)
//   * K: win_len padded to 2*KS; the A tile (32 x 2KS of D) lives in KS registers per lane
//     for the whole workgroup lifetime
//   * N: 32 consecutive window starts t; the B operand is read straight from an LDS copy of
//     the row chunk (Hankel structure: B[k][j] = x[t0 + j + k]), no matrix is materialised
// 'constant' detrend (scipy default) is applied algebraically: Y - mean_t * sum_m D[f][m],
// after the chunk mean has been removed from the LDS copy so nothing large cancels.
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

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TC = 256;          // window starts (output columns) per workgroup
constexpr int NT = 256;          // 4 waves, one M tile each

struct SgArgs {
  const float* x; float* S;
  const float* D;                // [tiles][KS][64] A-operand image of the DFT matrix
  const float2* Wsum;            // [tiles*16] sum_m D[f][m]
  const float* sc;               // [tiles*16] per-frequency scale (one-sided doubling included)
  long n, nseg; int nfreq, tiles, win_len, edge, detrend;
  int hop;                       // samples between segment starts (1: the hop-1 spectrogram; > 1: spectrogram_mtc)
  int tcs;                       // segments per workgroup: a multiple of 32, <= TC, (tcs - 1) hop + 2 KS floats of LDS
  long ldS;                      // floats per output row [b][f]: n (hop 1, zero edge columns) or nseg (scipy layout)
};

template <int KS>
__global__ __launch_bounds__(NT) void k_spectrogram(SgArgs a) {
  extern __shared__ float xs[];                       // (tcs - 1) hop + 2 KS samples of the row
  __shared__ float mu[TC];
  __shared__ float red[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long b = blockIdx.z;
  const long t0 = (long)blockIdx.x * a.tcs;            // first segment of the workgroup
  const long g0 = t0 * a.hop;                          // its first sample
  const int hop = a.hop, span = (a.tcs - 1) * hop + 2 * KS;
  const float* __restrict__ xr = a.x + b * a.n;
  // ---- stage the row chunk, remove its mean, running means of every window ----
  float part = 0.f;
  {
    int i = tid;
    for (; i + 3 * NT < span; i += 4 * NT) {            // four loads in flight per trip (a rolled loop waits for each)
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const long g = g0 + i + k * NT; v[k] = xr[g < a.n ? g : a.n - 1]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float w = (g0 + i + k * NT < a.n) ? v[k] : 0.f; xs[i + k * NT] = w; part += w; }
    }
    for (; i < span; i += NT) {
      const long g = g0 + i;
      const float v = g < a.n ? xr[g] : 0.f;
      xs[i] = v;
      part += v;
    }
  }
  if (a.detrend) {
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const long valid = std::min<long>(span, a.n - g0);
    const float cm = (red[0] + red[1] + red[2] + red[3]) / (float)valid;
    for (int i = tid; i < span; i += NT)
      if (g0 + i < a.n) xs[i] -= cm;
    __syncthreads();
    float s = 0.f;
    if (tid < a.tcs)                                    // TC == NT: one segment mean per thread
      for (int k = 0; k < a.win_len; ++k) s += xs[tid * hop + k];
    mu[tid] = s / (float)a.win_len;
  } else {
    mu[tid] = 0.f;
  }
  __syncthreads();
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= a.tiles) return;
  // ---- A tile of the DFT matrix: KS registers -----------------------------------
  float A[KS];
  const float* __restrict__ Dt = a.D + ((size_t)tile * KS) * 64 + lane;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) A[ks] = Dt[(size_t)ks * 64];
  const int g = lane >> 5, col = lane & 31;
  float2 W[8]; float scl[8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int f = tile * 16 + 8 * h + s + 4 * g;
      W[4 * h + s] = a.Wsum[f];
      scl[4 * h + s] = a.sc[f];
    }
  float* __restrict__ Sb = a.S + (size_t)b * a.nfreq * a.ldS;
  for (int nt = 0; nt < a.tcs / 32; ++nt) {
    const long t = t0 + nt * 32 + col;
    if (t0 + nt * 32 >= a.nseg) break;                              // wave-uniform
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* __restrict__ xb = xs + (nt * 32 + col) * hop + g;   // Hankel with stride hop: B[k][j] = x[(t0 + j) hop + k]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[ks], xb[2 * ks], acc, 0, 0, 0);
    if (t < a.nseg) {
      const float m = mu[nt * 32 + col];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int f = tile * 16 + 8 * h + s + 4 * g;
          if (f < a.nfreq) {
            const float re = acc[8 * h + s] - m * W[4 * h + s].x;
            const float im = acc[8 * h + 4 + s] - m * W[4 * h + s].y;
            Sb[(size_t)f * a.ldS + a.edge + t] = (re * re + im * im) * scl[4 * h + s];
          }
        }
    }
  }
}

// zero columns outside [edge, edge + nseg)
__global__ void k_sg_edges(float* __restrict__ S, long rows, long n, long edge, long nseg) {
  const long nz = n - nseg;
  const long total = rows * nz;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / nz, c = i % nz;
    S[r * n + (c < edge ? c : c + nseg)] = 0.f;
  }
}

}  // namespace

// tables + launch shared by the hop-1 spectrogram and the segmented one (device pointers)
static int sg_launch(xcx_ctx* ctx, const float* xd, float* Sd, long batch, long n, int win_len, int nfft,
                     const double* window, double scale, int detrend, long hop, long nseg, long ldS, long edge) {
  const int nfreq = nfft / 2 + 1;
  const int tiles = (nfreq + 15) / 16;
  const int KS = win_len <= 64 ? 32 : win_len <= 128 ? 64 : win_len <= 192 ? 96 : 128;
  // ---- tables (float64 on the host), cached per (win_len, nfft, window, scale) ----
  const size_t nD = (size_t)tiles * KS * 64, nW = (size_t)tiles * 16;
  const size_t bD = sizeof(float) * nD, bW = sizeof(float2) * nW, bS = sizeof(float) * nW;
  unsigned long long key = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t bytes) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < bytes; ++i) { key ^= c[i]; key *= 1099511628211ULL; }
  };
  mix(&win_len, sizeof win_len); mix(&nfft, sizeof nfft); mix(&scale, sizeof scale);
  mix(window, sizeof(double) * win_len);
  int rc;
  DevBuf& tab = ctx->sg_tab;
  if (key != ctx->sg_key || !tab.ptr) {
    std::vector<float> D(nD, 0.f), sc(nW, 0.f);
    std::vector<float2> Ws(nW, make_float2(0.f, 0.f));
    for (int tile = 0; tile < tiles; ++tile) {
      for (int fi = 0; fi < 16; ++fi) {
        const int f = tile * 16 + fi;
        if (f >= nfreq) continue;
        double sr = 0.0, si = 0.0;
        for (int m = 0; m < win_len; ++m) {
          const double th = -2.0 * M_PI * (double)(((long)f * m) % nfft) / (double)nfft;
          sr += window[m] * cos(th); si += window[m] * sin(th);
        }
        Ws[(size_t)tile * 16 + fi] = make_float2((float)sr, (float)si);
        const bool single = f == 0 || (nfft % 2 == 0 && f == nfft / 2);
        sc[(size_t)tile * 16 + fi] = (float)(scale * (single ? 1.0 : 2.0));
      }
      for (int ks = 0; ks < KS; ++ks)
        for (int l = 0; l < 64; ++l) {
          const int i = l & 31, k = 2 * ks + (l >> 5);
          const int fi = (i & 7) + 8 * (i >> 4), im = (i >> 3) & 1;
          const int f = tile * 16 + fi;
          if (f >= nfreq || k >= win_len) continue;
          const double th = -2.0 * M_PI * (double)(((long)f * k) % nfft) / (double)nfft;
          D[((size_t)tile * KS + ks) * 64 + l] = (float)(window[k] * (im ? sin(th) : cos(th)));
        }
    }
    ctx->sg_key = 0;
    if ((rc = tab.ensure(ctx, bD + bW + bS))) return rc;
    char* up = (char*)tab.ptr;
    HIP_TRY(hipMemcpyAsync(up, D.data(), bD, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(up + bD, Ws.data(), bW, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(up + bD + bW, sc.data(), bS, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));                     // host vectors go out of scope
    ctx->sg_key = key;
  }
  char* tp = (char*)tab.ptr;
  // segments per workgroup: as many as 32 KB of LDS hold at this hop (256 at hop 1 ... 32 at the largest hops)
  int tcs = (int)std::min<long>(TC, ((long)(8192 - 2 * KS) / hop + 1) / 32 * 32);
  if (tcs < 32) tcs = 32;
  const size_t lds = sizeof(float) * ((size_t)(tcs - 1) * hop + 2 * KS);
  SgArgs a{xd, Sd, (const float*)tp, (const float2*)(tp + bD), (const float*)(tp + bD + bW),
           n, nseg, nfreq, tiles, win_len, (int)edge, detrend ? 1 : 0, (int)hop, tcs, ldS};
  const dim3 grid((unsigned)((nseg + tcs - 1) / tcs), (unsigned)((tiles + 3) / 4), (unsigned)batch);
  switch (KS) {
    case 32: hipLaunchKernelGGL(k_spectrogram<32>, grid, dim3(NT), lds, ctx->stream, a); break;
    case 64: hipLaunchKernelGGL(k_spectrogram<64>, grid, dim3(NT), lds, ctx->stream, a); break;
    case 96: hipLaunchKernelGGL(k_spectrogram<96>, grid, dim3(NT), lds, ctx->stream, a); break;
    default: hipLaunchKernelGGL(k_spectrogram<128>, grid, dim3(NT), lds, ctx->stream, a); break;
  }
  if (hipGetLastError() != hipSuccess) return set_err(XCX_EHIP, "k_spectrogram launch failed");
  return XCX_OK;
}

extern "C" int xcx_spectrogram(xcx_ctx* ctx, const float* x, float* S, long batch, long n, int win_len, int nfft,
                               const double* window, double scale, int detrend, int on_device) {
  if (!ctx || !x || !S || !window) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (win_len < 2 || win_len > XCX_SPECTROGRAM_MAX_WIN) return set_err(XCX_EUNSUPPORTED, "win_len %d outside [2, %ld]", win_len, XCX_SPECTROGRAM_MAX_WIN);
  if (nfft < win_len || nfft > XCX_SPECTROGRAM_MAX_NFFT) return set_err(XCX_EINVAL, "nfft %d outside [win_len, %ld]", nfft, XCX_SPECTROGRAM_MAX_NFFT);
  if (n < win_len) return set_err(XCX_EINVAL, "row shorter than one segment (%ld < %d)", n, win_len);
  HIP_TRY(hipSetDevice(ctx->device));
  const int nfreq = nfft / 2 + 1;
  const long nseg = n - win_len + 1;
  const long edge = (long)std::rint(win_len / 2.0);                 // spectrogram.py:121 (np.rint: half to even)
  const size_t in_b = sizeof(float) * (size_t)batch * n, out_b = in_b * nfreq;
  const float* xd = x; float* Sd = S;
  void *ha = nullptr, *hb = nullptr;
  HostPin pin_x(on_device ? nullptr : x, in_b), pin_S(on_device ? nullptr : S, out_b);
  if (!on_device) {
    HIP_TRY(ctx_malloc(ctx, &ha, in_b));
    if (ctx_malloc(ctx, &hb, out_b) != hipSuccess) { (void)hipFree(ha); return set_err(XCX_ENOMEM, "spectrogram output (%zu bytes)", out_b); }
    (void)hipMemcpyAsync(ha, x, in_b, hipMemcpyHostToDevice, ctx->stream);
    xd = (const float*)ha; Sd = (float*)hb;
  }
  int rc = sg_launch(ctx, xd, Sd, batch, n, win_len, nfft, window, scale, detrend, 1, nseg, n, edge);
  if (rc == XCX_OK && n > nseg) {
    const long rows = batch * nfreq;
    hipLaunchKernelGGL(k_sg_edges, dim3((unsigned)std::min<long>((rows * (n - nseg) + 255) / 256, 8192)), dim3(256), 0,
                       ctx->stream, Sd, rows, n, edge, nseg);
  }
  if (!on_device) {
    if (rc == XCX_OK && hipMemcpyAsync(S, Sd, out_b, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = set_err(XCX_EHIP, "copy back failed");
  }
  if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == XCX_OK) rc = set_err(XCX_EHIP, "stream sync failed");
  if (ha) (void)hipFree(ha);
  if (hb) (void)hipFree(hb);
  return rc;
}

// ---- segmented spectrogram (any hop): signal/spectrogram.py:199-335 (spectrogram_mtc) -------------
// scipy.signal.spectrogram(x, fs, nperseg = win_len, noverlap = win_len - hop, nfft, window,
// detrend, scaling, mode='psd'): segment s starts at s*hop; per segment: remove the mean
// ('constant'), multiply by the window, zero-pad to nfft, one-sided |rfft|^2 * scale (x2 except DC
// and, for even nfft, Nyquist).  Segments are framed on the device and transformed by the batched
// any-length DFT of fftn.hip; output [batch][nfreq][nseg] like scipy's Sxx.
namespace {

// frames [rows = batch*nseg][nfft] real (cplx = 0) or interleaved complex with zero imaginary part
__global__ void k_frames(const float* __restrict__ x, float* __restrict__ fr, const float* __restrict__ w,
                         long n, long nseg, int win_len, long hop, int nfft, int detrend, int cplx, long row0) {
  const long row = row0 + blockIdx.x;               // b * nseg + s
  const long b = row / nseg, s = row - b * nseg;
  const float* __restrict__ xs = x + b * n + s * hop;
  __shared__ float red[4];
  float mean = 0.f;
  if (detrend) {
    double part = 0.0;                              // short segments, but offsets can dwarf the signal
    for (int m = threadIdx.x; m < win_len; m += blockDim.x) part += (double)xs[m];
    float pf = (float)part;
    for (int o = 32; o > 0; o >>= 1) pf += __shfl_down(pf, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pf;
    __syncthreads();
    mean = (red[0] + red[1] + red[2] + red[3]) / (float)win_len;
  }
  float* __restrict__ o = fr + (long)blockIdx.x * nfft * (cplx ? 2 : 1);
  for (int m = threadIdx.x; m < nfft; m += blockDim.x) {
    const float v = m < win_len ? (xs[m] - mean) * w[m] : 0.f;
    if (cplx) { o[2 * m] = v; o[2 * m + 1] = 0.f; }
    else o[m] = v;
  }
}

// S[b][f][s] = |X[b*nseg + s][f]|^2 * sc(f); spec rows have `pitch` complex entries
__global__ void k_psd_t(const float2* __restrict__ spec, float* __restrict__ S, long nseg, int nfreq, long pitch,
                        int nfft, float scale, long rows0, long nrows, long nseg_total) {
  // tile transpose: 32 segments x 32 frequencies
  __shared__ float tile[32][33];
  const long r0 = (long)blockIdx.x * 32;            // local frame rows
  const int f0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    const long r = r0 + j; const int f = f0 + tx;
    float v = 0.f;
    if (r < nrows && f < nfreq) {
      const float2 c = spec[r * pitch + f];
      const bool single = f == 0 || (nfft % 2 == 0 && f == nfft / 2);
      v = (c.x * c.x + c.y * c.y) * scale * (single ? 1.f : 2.f);
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int f = f0 + j; const long r = r0 + tx;
    if (r < nrows && f < nfreq) {
      const long g = rows0 + r;                     // global frame row
      const long b = g / nseg_total, s = g - b * nseg_total;
      S[(b * nfreq + f) * nseg_total + s] = tile[tx][j];
    }
  }
}

}  // namespace

extern "C" int xcx_spectrogram_seg(xcx_ctx* ctx, const float* x, float* S, long batch, long n, int win_len,
                                   long hop, int nfft, const double* window, double scale, int detrend,
                                   int on_device) {
  if (!ctx || !x || !S || !window) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  if (win_len < 2) return set_err(XCX_EINVAL, "win_len must be >= 2");
  if (hop < 1 || hop > win_len) return set_err(XCX_EINVAL, "hop %ld outside [1, win_len]", hop);
  if (nfft < win_len) return set_err(XCX_EINVAL, "nfft %d < win_len %d", nfft, win_len);
  if (nfft > XCX_SPECTROGRAM_SEG_MAX_NFFT) return set_err(XCX_EUNSUPPORTED, "nfft %d beyond %ld", nfft, XCX_SPECTROGRAM_SEG_MAX_NFFT);
  if (n < win_len) return set_err(XCX_EINVAL, "row shorter than one segment (%ld < %d)", n, win_len);
  HIP_TRY(hipSetDevice(ctx->device));
  const long nseg = (n - win_len) / hop + 1;
  const int nfreq = nfft / 2 + 1;
  if (win_len <= XCX_SPECTROGRAM_MAX_WIN && nfft <= XCX_SPECTROGRAM_MAX_NFFT) {
    // short segments: the same Hankel GEMM on the matrix cores as the hop-1 spectrogram, with column stride `hop`
    // (the DFT-engine route below spent 205 us on a row the hop-1 kernel, with 25 x the segments, does in 14)
    const size_t in_b = sizeof(float) * (size_t)batch * n, out_b = sizeof(float) * (size_t)batch * nfreq * nseg;
    const float* xd = x; float* Sd = S;
    void *ha = nullptr, *hb = nullptr;
    HostPin pin_x(on_device ? nullptr : x, in_b), pin_S(on_device ? nullptr : S, out_b);
    if (!on_device) {
      HIP_TRY(ctx_malloc(ctx, &ha, in_b));
      if (ctx_malloc(ctx, &hb, out_b) != hipSuccess) { (void)hipFree(ha); return set_err(XCX_ENOMEM, "spectrogram output (%zu bytes)", out_b); }
      xd = (const float*)ha; Sd = (float*)hb;
    }
    int rc = XCX_OK;
    if (!on_device && hipMemcpyAsync(ha, x, in_b, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      rc = set_err(XCX_EHIP, "copy of the input rows to the device failed");      // never run the kernel on uninitialised memory
    if (rc == XCX_OK) rc = sg_launch(ctx, xd, Sd, batch, n, win_len, nfft, window, scale, detrend, hop, nseg, nseg, 0);
    if (!on_device && rc == XCX_OK && hipMemcpyAsync(S, Sd, out_b, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
      rc = set_err(XCX_EHIP, "copy back failed");
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == XCX_OK) rc = set_err(XCX_EHIP, "stream sync failed");
    if (ha) (void)hipFree(ha);
    if (hb) (void)hipFree(hb);
    return rc;
  }
  const int cplx = nfft & 1;                        // xcx_rfft handles even lengths; odd ones go through the complex DFT
  const long total = batch * nseg;
  // frames per chunk: bound the temporaries to ~256 MB
  const size_t per_row = sizeof(float) * (size_t)nfft * (cplx ? 2 : 1) + sizeof(float2) * (size_t)(cplx ? nfft : nfreq);
  const long chunk = std::max<long>(1, std::min<long>(total, (long)(((size_t)256 << 20) / per_row)));
  std::vector<float> wf(window, window + win_len);
  void *dw = nullptr, *dfr = nullptr, *dsp = nullptr, *ha = nullptr, *hb = nullptr;
  auto cleanup = [&]() { (void)hipFree(dw); (void)hipFree(dfr); (void)hipFree(dsp); (void)hipFree(ha); (void)hipFree(hb); };
  auto fail = [&](int rc) { cleanup(); return rc; };
  if (ctx_malloc(ctx, &dw, sizeof(float) * win_len) != hipSuccess ||
      ctx_malloc(ctx, &dfr, sizeof(float) * (size_t)nfft * (cplx ? 2 : 1) * chunk) != hipSuccess ||
      ctx_malloc(ctx, &dsp, sizeof(float2) * (size_t)(cplx ? nfft : nfreq) * chunk) != hipSuccess)
    return fail(set_err(XCX_ENOMEM, "spectrogram workspaces"));
  const size_t in_b = sizeof(float) * (size_t)batch * n, out_b = sizeof(float) * (size_t)batch * nfreq * nseg;
  const float* xd = x; float* Sd = S;
  if (!on_device) {
    if (ctx_malloc(ctx, &ha, in_b) != hipSuccess || ctx_malloc(ctx, &hb, out_b) != hipSuccess)
      return fail(set_err(XCX_ENOMEM, "spectrogram buffers"));
    (void)hipMemcpyAsync(ha, x, in_b, hipMemcpyHostToDevice, ctx->stream);
    xd = (const float*)ha; Sd = (float*)hb;
  }
  if (hipMemcpyAsync(dw, wf.data(), sizeof(float) * win_len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(set_err(XCX_EHIP, "window upload"));
  const long pitch = cplx ? nfft : nfreq;
  for (long r0 = 0; r0 < total; r0 += chunk) {
    const long cnt = std::min(chunk, total - r0);
    hipLaunchKernelGGL(k_frames, dim3((unsigned)cnt), dim3(256), 0, ctx->stream,
                       xd, (float*)dfr, (const float*)dw, n, nseg, win_len, hop, nfft, detrend, cplx, r0);
    if (hipGetLastError() != hipSuccess) return fail(set_err(XCX_EHIP, "k_frames launch"));
    int rc = cplx ? xcx_fft(ctx, (const float*)dfr, (float*)dsp, cnt, nfft, 1)
                  : xcx_rfft(ctx, (const float*)dfr, (float*)dsp, cnt, nfft, 1);
    if (rc) return fail(rc);
    hipLaunchKernelGGL(k_psd_t, dim3((unsigned)((cnt + 31) / 32), (unsigned)((nfreq + 31) / 32)), dim3(256), 0,
                       ctx->stream, (const float2*)dsp, Sd, nseg, nfreq, pitch, nfft, (float)scale, r0, cnt, nseg);
    if (hipGetLastError() != hipSuccess) return fail(set_err(XCX_EHIP, "k_psd_t launch"));
  }
  if (!on_device) (void)hipMemcpyAsync(S, Sd, out_b, hipMemcpyDeviceToHost, ctx->stream);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(set_err(XCX_EHIP, "spectrogram_seg sync"));
  cleanup();
  return XCX_OK;
}
