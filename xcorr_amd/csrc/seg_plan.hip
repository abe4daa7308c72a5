// seg_plan.hip -- segmented plans: correlations (and the convolutions of the time shift and of the long chirp
// transforms) of windows longer than any single transform, block-decomposed onto the time-stacking engine of a
// sub-plan at the tuned transform length.  Split out of xcx_api.hip in round 5.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "xcx_plan.h"

using namespace xcx;

// forward transforms of windows that may be shorter than the plan's n (zero beyond n_valid): xcx_api.hip
static int seg_forward(xcx_plan* p, const float* traces, long stride_s, long stride_w, int W, int w0, int sw_begin,
                       int sw_end, int slot0, int n_valid = 0) {
  return xcx_internal_forward_raw(p, traces, stride_s, stride_w, W, w0, sw_begin, sw_end, slot0, n_valid);
}

// ---------------------------------------------------------------------------
// Segmented correlation: windows whose n + max|lag| exceeds the largest transform.
//
// For non-negative lags k,  C[k] = sum_t a[t+k] b[t]  is cut twice: b into blocks of Bk samples
// (b_beta = b[beta Bk .. beta Bk + Bk)) and the lags into chunks of Kb.  For lag chunk c,
//     C[k0 + c Kb + k'] = sum_beta  cc(A_{beta,c}, B_beta)[k'],   0 <= k' < Kb,
//     A_{beta,c}[i] = a[beta Bk + k0 + c Kb + i]   (i < Bk + Kb, zero outside the window),
//     B_beta[i]     = b[beta Bk + i] for i < Bk, zero up to Bk + Kb,
// i.e. ONE time-stacked pair (station A_c, station B) over the windows beta = 0..nb-1 of a
// sub-plan with window length n' = Bk + Kb that retains the Kb non-negative lags -- exactly what
// the stacking engine (forward transforms, MFMA cross-spectrum accumulation, one inverse per pair)
// does at its tuned transform length N' = 368 640 (Bk = N'/2, Kb = N'/4).  Negative lags are the
// non-negative lags with the roles of x and y exchanged.  Cost ~ 2 n K / (Bk Kb) block transforms
// instead of one of length >= n + K: a 4 320 000-sample window with 1 620 000 retained lags
// (the reference's production setting) takes 24 x 19 forward transforms + 18 inverses.
// ---------------------------------------------------------------------------
namespace {

// Window statistics, once per DISTINCT window of the call (`ptrs`: the windows that occur as x or y of some row):
// su[2u] = mean (0 without demean), su[2u + 1] = energy of what is transformed (after the mean / one-bit), both float64.
// A day-long window is split over kSegParts workgroups (one workgroup per row read its 4.32 M samples in a rolled
// loop, one load and one wait per trip: 7.4 ms per call, 59 % of the production window, until round 4); the loads go
// out in batches of 8 and are consumed in order, the partial sums are combined in fixed order by k_seg_stats_fin.
constexpr int kSegParts = 128;
__global__ __launch_bounds__(256) void k_seg_stats(const float* const* __restrict__ ptrs, long n, int flags,
                                                    const double* __restrict__ su, double* __restrict__ part /*[nu][kSegParts]*/,
                                                    int phase) {
  __shared__ double red[256];
  const int u = blockIdx.y, tid = threadIdx.x;
  const float* __restrict__ x = ptrs[u];
  const long per = (n + kSegParts - 1) / kSegParts;
  const long lo = (long)blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  double a = 0.0;
  const double mx = phase ? su[2 * u] : 0.0;
  const bool ob = flags & XCX_F_ONEBIT;
  auto take = [&](float xf) {
    if (phase == 0) { a += (double)xf; return; }
    double v = (double)xf - mx;
    if (ob) v = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
    else v = (double)(float)v;                                 // the windows are transformed as float32
    a += v * v;
  };
  if (phase == 1 || (flags & XCX_F_DEMEAN)) {
    constexpr int U = 8;
    long i = lo + tid;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
      float xv[U];
#pragma unroll
      for (int k = 0; k < U; ++k) xv[k] = x[i + k * 256];
#pragma unroll
      for (int k = 0; k < U; ++k) take(xv[k]);
    }
    for (; i < hi; i += 256) take(x[i]);
  }
  red[tid] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) part[(size_t)u * kSegParts + blockIdx.x] = red[0];
}
__global__ void k_seg_stats_fin(const double* __restrict__ part, long n, double* __restrict__ su, int nu, int phase) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= nu) return;
  double a = 0.0;
  for (int p = 0; p < kSegParts; ++p) a += part[(size_t)u * kSegParts + p];
  if (phase == 0) su[2 * u] = a / (double)n;
  else su[2 * u + 1] = a;
}

// Distinct-segment form (round 4).  With Bk = Kb the window of chunk c and block beta starts at sample
// k0 + (beta + c) Kb: it depends on s = beta + c only, so the nb x nc windows are NS = nb - 1 + nc DISTINCT segments
// (32 instead of 216 for the production window).  And they depend on the window they are cut from, not on the row:
// rows (pairs) that share a window share its spectra.  The block is the SHORT operand of the sub-plan (Bk samples of
// N = Bk + Kb, lags -(Kb-1) .. 0 of (block, segment) retained: alias-free at N, kPlanShortOperand) -- before: the
// generic rule N >= npr + Kb with Bk = 2 Kb = N/2, i.e. twice the inverse transforms per lag.  Per chunk of rows:
//   k_seg_gather : for every distinct a-window NS segments a[k0 + s Kb + i], i < npr; for every distinct b-window
//                  nb blocks of Bk samples (zero beyond) -- each transformed ONCE
//   k_seg_gram   : G_c[k] = sum_beta B[ib(r)][beta][k] conj(A[ia(r)][beta + c][k])  for row r
// windows: (ids, nwin) -> dst[(v nwin + w)][npr], sample base0 + w * wstride + i for i < len (zero beyond, and beyond n)
constexpr int kSegU = 8;
__global__ __launch_bounds__(256) void k_seg_gather(const float* const* __restrict__ ptrs, const double* __restrict__ su,
                                                     const int* __restrict__ ids, long n, int flags, int npr, int nwin,
                                                     long base0, long wstride, int len, float* __restrict__ dst) {
  const int w = blockIdx.y, v = blockIdx.z;
  const int win = ids[v];
  const float* __restrict__ src = ptrs[win];
  const double m = su[2 * win];
  const bool ob = flags & XCX_F_ONEBIT;
  float* __restrict__ d = dst + ((size_t)v * nwin + w) * (size_t)npr;
  const long base = base0 + (long)w * wstride;
  // kSegU elements per thread, their loads issued together (a grid-stride loop here is one load and one wait per trip)
  const int i0 = blockIdx.x * (256 * kSegU) + threadIdx.x;
  float x[kSegU];
  bool in[kSegU];
#pragma unroll
  for (int u = 0; u < kSegU; ++u) {
    const int i = i0 + u * 256;
    const long j = base + i;
    in[u] = i < len && j < n;
    x[u] = src[in[u] ? j : 0];                       // unconditional load, then a select (no branch around the load)
  }
#pragma unroll
  for (int u = 0; u < kSegU; ++u) {
    const int i = i0 + u * 256;
    const double uu = (double)x[u] - m;
    const float val = ob ? ((uu > 0.0) ? 1.f : ((uu < 0.0) ? -1.f : 0.f)) : (float)uu;
    if (i < npr) d[i] = in[u] ? val : 0.f;
  }
}

// C[(r nc + c)][k] = sum_beta X[slotB + ib[r] nb + beta][k] conj(X[ia[r] NS + beta + c][k]); one frequency per lane
// (512-byte runs per wave load), SG chunks per workgroup.  The SG segments a block meets are a WINDOW of the segment
// axis that slides by one per block: it lives in registers and ONE load per block refills it (2 loads per block and
// SG products; the first form loaded every operand of every product: 7 loads per 6 products).
// R = 2: two rows that share the a-window (pairs of one station against two others, rows of one delay) ride the same
// register window: 3 loads per block and 2 SG products.  grp[z R + i] = row i of group z.
constexpr int kSegUB = 4;
typedef float f2 __attribute__((ext_vector_type(2)));
template <int SG, int R>
__global__ __launch_bounds__(256) void k_seg_gram(const cf* __restrict__ X, size_t pitch, int nfreq, int NS, int nb, int nc,
                                                   size_t slotB, const int* __restrict__ ia, const int* __restrict__ ib,
                                                   const int* __restrict__ grp, int G, int gy, cf* __restrict__ C,
                                                   const double* __restrict__ su, const int* __restrict__ rx,
                                                   const int* __restrict__ ry, int scaled, int seg_first) {
  // One wave = 64 frequencies of one group of rows; the waves of a workgroup take four groups at the SAME frequencies,
  // and the frequency tiles are dealt to the XCDs (tile f on XCD f mod 8, workgroup ids go round the XCDs): all groups
  // of a tile run back to back on one XCD, so every window's 512 bytes of the tile are fetched into that L2 once and
  // the other rows that meet the window hit there (with frequency tiles as the fast grid axis a window's lines were
  // long gone when the next row asked for them: 60 MB per row from beyond the L2).
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int per_tile = ((G + 3) >> 2) * gy;
  const int t = q / per_tile, rem = q - t * per_tile;
  const int f = x + 8 * t;
  // everything but the lane's frequency is uniform over the wave: said so (readfirstlane), the row numbers, window
  // indices and segment addresses are scalar work and the loads take a scalar base plus the lane's 32-bit offset
  const int grp_i = (rem / gy) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned k = (unsigned)f * 64u + (threadIdx.x & 63);
  if (k >= (unsigned)nfreq || grp_i >= G) return;
  const int c0 = (rem % gy) * SG;
  int r[R];
  const cf* __restrict__ Xb[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    r[i] = grp[grp_i * R + i];
    Xb[i] = X + (slotB + (size_t)ib[r[i]] * nb) * pitch;
  }
  const cf* __restrict__ Xa = X + ((size_t)ia[r[0]] * NS + c0) * pitch;
  const int smax = NS - 1 - c0;                      // segments past the last one belong to chunks >= nc (not stored)
  const unsigned koff = k * (unsigned)sizeof(cf);    // scalar row address + 32-bit lane offset: the load's own address mode
  auto ld = [koff](const cf* row) { return *reinterpret_cast<const cf*>(reinterpret_cast<const char*>(row) + koff); };
  f2 acc[R][SG];
  // the window: w[0 .. SG-2] = segments beta .. beta+SG-2 at the start of a trip, the trip's loads append one segment
  // per block, block u of the trip meets w[u .. u+SG-1] (static registers), one shift per trip
  cf w[SG - 1 + kSegUB];
#pragma unroll
  for (int j = 0; j < SG; ++j)
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i][j] = f2{0.f, 0.f};
#pragma unroll
  for (int j = 0; j + 1 < SG; ++j) w[j] = ld(Xa + (size_t)min(j, smax) * pitch);
  // kSegUB blocks per trip, their loads issued together: with one block per trip (3 loads, then a full wait) the kernel
  // ran on latency x occupancy, not on bandwidth
  for (int beta = 0; beta < nb; beta += kSegUB) {
    cf b[kSegUB][R];
#pragma unroll
    for (int u = 0; u < kSegUB; ++u) {
      const int bb = min(beta + u, nb - 1);
#pragma unroll
      for (int i = 0; i < R; ++i) b[u][i] = ld(Xb[i] + (size_t)bb * pitch);
      w[SG - 1 + u] = ld(Xa + (size_t)min(beta + u + SG - 1, smax) * pitch);
    }
#pragma unroll
    for (int u = 0; u < kSegUB; ++u) {
      if (beta + u < nb) {
        f2 bv[R], bs[R];
#pragma unroll
        for (int i = 0; i < R; ++i) { bv[i] = f2{b[u][i].x, b[u][i].y}; bs[i] = f2{b[u][i].y, -b[u][i].x}; }
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
          for (int j = 0; j < SG; ++j)      // b conj(a) as two packed FMAs: (bx, by) ax + (by, -bx) ay
            acc[i][j] = __builtin_elementwise_fma(bs[i], f2{w[u + j].y, w[u + j].y},
                                                  __builtin_elementwise_fma(bv[i], f2{w[u + j].x, w[u + j].x}, acc[i][j]));
      }
    }
#pragma unroll
    for (int j = 0; j + 1 < SG; ++j) w[j] = w[j + kSegUB];
  }
  // acc = B conj(A): the cross-spectrum of (block, segment), lags <= 0 of the sub-plan `sub`; its conjugate is the one of
  // (segment, block), lags >= 0 of `subp` (seg_first).  scaled: 1 / (|x| |y|) of the row goes in here, and the inverse
  // transforms write the row's lags in place.
#pragma unroll
  for (int i = 0; i < R; ++i) {
    float sx = 1.f;
    if (scaled) sx = (float)(1.0 / (sqrt(su[2 * rx[r[i]] + 1]) * sqrt(su[2 * ry[r[i]] + 1])));
    const float sy = seg_first ? -sx : sx;
#pragma unroll
    for (int j = 0; j < SG; ++j)
      if (c0 + j < nc) {                             // streamed: nothing reads C before the inverse's column pass
        float* dst = reinterpret_cast<float*>(C + ((size_t)r[i] * nc + c0 + j) * pitch + k);
        __builtin_nontemporal_store(acc[i][j].x * sx, dst);
        __builtin_nontemporal_store(acc[i][j].y * sy, dst + 1);
      }
  }
}
// Sum of two products per output row (convolution mode, complex chirp convolutions of fftn.hip): row z of C is
// B[r0] conj(A[r0]) + sign[z] B[r1] conj(A[r1]) for the rows (r0, r1) = grp[2z], grp[2z+1] -- the real (or imaginary) part
// of a complex convolution needs ONE inverse transform per chunk instead of two.  Two register windows, same walk.
template <int SG>
__global__ __launch_bounds__(256) void k_seg_gram_sum(const cf* __restrict__ X, size_t pitch, int nfreq, int NS, int nb, int nc,
                                                       size_t slotB, const int* __restrict__ ia, const int* __restrict__ ib,
                                                       const int* __restrict__ grp, const int* __restrict__ sign, int G, int gy,
                                                       cf* __restrict__ C) {
  const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int per_tile = ((G + 3) >> 2) * gy;
  const int t = q / per_tile, rem = q - t * per_tile;
  const int f = x + 8 * t;
  const int grp_i = (rem / gy) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned k = (unsigned)f * 64u + (threadIdx.x & 63);
  if (k >= (unsigned)nfreq || grp_i >= G) return;
  const int c0 = (rem % gy) * SG;
  const cf* __restrict__ Xb[2];
  const cf* __restrict__ Xa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = grp[grp_i * 2 + i];
    Xb[i] = X + (slotB + (size_t)ib[r] * nb) * pitch;
    Xa[i] = X + ((size_t)ia[r] * NS + c0) * pitch;
  }
  const float sg1 = sign[grp_i] < 0 ? -1.f : 1.f;
  const int smax = NS - 1 - c0;
  const unsigned koff = k * (unsigned)sizeof(cf);
  auto ld = [koff](const cf* row) { return *reinterpret_cast<const cf*>(reinterpret_cast<const char*>(row) + koff); };
  f2 acc[SG];
  cf w[2][SG - 1 + kSegUB];
#pragma unroll
  for (int j = 0; j < SG; ++j) acc[j] = f2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j + 1 < SG; ++j) w[i][j] = ld(Xa[i] + (size_t)min(j, smax) * pitch);
  for (int beta = 0; beta < nb; beta += kSegUB) {
    cf b[kSegUB][2];
#pragma unroll
    for (int u = 0; u < kSegUB; ++u) {
      const int bb = min(beta + u, nb - 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        b[u][i] = ld(Xb[i] + (size_t)bb * pitch);
        w[i][SG - 1 + u] = ld(Xa[i] + (size_t)min(beta + u + SG - 1, smax) * pitch);
      }
    }
#pragma unroll
    for (int u = 0; u < kSegUB; ++u) {
      if (beta + u < nb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float sgi = i ? sg1 : 1.f;
          const f2 bv = f2{b[u][i].x * sgi, b[u][i].y * sgi}, bs = f2{b[u][i].y * sgi, -b[u][i].x * sgi};
#pragma unroll
          for (int j = 0; j < SG; ++j)
            acc[j] = __builtin_elementwise_fma(bs, f2{w[i][u + j].y, w[i][u + j].y},
                                               __builtin_elementwise_fma(bv, f2{w[i][u + j].x, w[i][u + j].x}, acc[j]));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j + 1 < SG; ++j) w[i][j] = w[i][j + kSegUB];
  }
#pragma unroll
  for (int j = 0; j < SG; ++j)
    if (c0 + j < nc) {
      float* dst = reinterpret_cast<float*>(C + ((size_t)grp_i * nc + c0 + j) * pitch + k);
      __builtin_nontemporal_store(acc[j].x, dst);
      __builtin_nontemporal_store(acc[j].y, dst + 1);
    }
}
using SegGramSumFn = void (*)(const cf*, size_t, int, int, int, int, size_t, const int*, const int*, const int*, const int*, int,
                              int, cf*);
using SegGramFn = void (*)(const cf*, size_t, int, int, int, int, size_t, const int*, const int*, const int*, int, int, cf*,
                           const double*, const int*, const int*, int, int);
struct SegGramVar { int sg; SegGramFn fn1, fn2; SegGramSumFn fsum; };
static const SegGramVar kSegGram[] = {{3, k_seg_gram<3, 1>, k_seg_gram<3, 2>, k_seg_gram_sum<3>},
                                      {6, k_seg_gram<6, 1>, k_seg_gram<6, 2>, k_seg_gram_sum<6>},
                                      {9, k_seg_gram<9, 1>, k_seg_gram<9, 2>, k_seg_gram_sum<9>},
                                      {12, k_seg_gram<12, 1>, k_seg_gram<12, 2>, k_seg_gram_sum<12>}};
// fewest loads per row: groups x (SG + 2 nb); ties go to the wider window
static const SegGramVar& seg_gram_pick(int nc, int nb) {
  const SegGramVar* best = &kSegGram[0];
  long bc = -1;
  for (const SegGramVar& v : kSegGram) {
    const long cost = (long)((nc + v.sg - 1) / v.sg) * (v.sg + 2L * nb);
    if (bc < 0 || cost <= bc) { bc = cost; best = &v; }
  }
  return *best;
}

// sub[(r*nc + c)][rev ? Kb-1-k' : k'] -> out[r][...]: lag k = k0 + (c0+c) Kb + k' (rev: the lags of `sub` run -(Kb-1) .. 0;
// sign -1 for the exchanged roles)
__global__ __launch_bounds__(256) void k_seg_scatter(const float* __restrict__ sub, const double* __restrict__ su,
                                                      const int* __restrict__ rx, const int* __restrict__ ry, int flags,
                                                      long n, int Kb, int nc, int c0, long k0, long kend /*exclusive*/,
                                                      int sign, int rev, long idx_min, long L, float* __restrict__ out,
                                                      long ld_out) {
  const int r = blockIdx.z, c = blockIdx.y;
  double scale = 1.0;
  if (flags & XCX_F_NORMALIZE) scale = 1.0 / (sqrt(su[2 * rx[r] + 1]) * sqrt(su[2 * ry[r] + 1]));
  const float* __restrict__ srow = sub + ((size_t)r * nc + c) * (size_t)Kb;
  const int kk0 = blockIdx.x * (256 * kSegU) + threadIdx.x;
  float sv[kSegU];
#pragma unroll
  for (int u = 0; u < kSegU; ++u) { const int kq = min(kk0 + u * 256, Kb - 1); sv[u] = srow[rev ? Kb - 1 - kq : kq]; }
#pragma unroll
  for (int u = 0; u < kSegU; ++u) {
    const int kk = kk0 + u * 256;
    const long k = k0 + (long)(c0 + c) * Kb + kk;
    if (kk >= Kb || k >= kend) continue;
    const long j = (n - 1) + sign * k - idx_min;
    if (j < 0 || j >= L) continue;
    double v = (double)sv[u] * scale;
    if (flags & XCX_F_UNBIAS) v *= (double)n / (double)(n - k);
    out[(size_t)r * ld_out + j] = (float)v;
  }
}

}  // namespace

int xcx_internal_seg_plan_create(xcx_ctx* ctx, int n, int idx_min, int idx_max, int flags, xcx_plan** out) {
  HIP_TRY(hipSetDevice(ctx->device));
  xcx_plan* p = new (std::nothrow) xcx_plan();
  if (!p) return set_err(XCX_ENOMEM, "host allocation failed");
  p->ctx = ctx;
  p->g.n = n;
  p->g.idx_min = idx_min; p->g.idx_max = idx_max;
  p->g.flags = flags & 0xffff;
  p->L = idx_max - idx_min;
  p->seg = new (std::nothrow) xcx_plan::Seg();
  if (!p->seg) { delete p; return set_err(XCX_ENOMEM, "host allocation failed"); }
  // block geometry on the tuned transform N' = 368 640 = Bk + Kb, Bk = Kb (XCX_SEG_N overrides)
  long Np = 368640;
  if (const char* e = getenv("XCX_SEG_N")) {
    Np = atol(e);
    if (Np < 4096 || Np % 4 != 0) {
      delete p;
      return set_err(XCX_EINVAL, "XCX_SEG_N=%s: the block transform length must be a multiple of 4 and >= 4096", e);
    }
  }
  p->seg->Kb = (int)(Np / 2);
  p->seg->Bk = (int)(Np / 2);
  p->seg->npr = p->seg->Bk + p->seg->Kb;
  const int npr = p->seg->npr;
  // first operand = block (zero beyond Bk = idx_min samples), lags -(Kb-1) .. 0: no alias at N = npr
  int rc = xcx_plan_cc1d(ctx, npr, 0, npr - p->seg->Kb, npr, XCX_F_STACK | kPlanShortOperand, &p->seg->sub);
  if (rc) { xcx_plan_destroy(p); return rc; }
  // the mirror image for the non-negative lags: (segment, block), the SECOND operand is the short one, lags 0 .. Kb-1 --
  // alias-free at N = npr by the same argument; its inverse writes a chunk of lags in rising order
  rc = xcx_plan_cc1d(ctx, npr, 0, npr - 1, npr - 1 + p->seg->Kb, XCX_F_STACK | kPlanShortOperand, &p->seg->subp);
  if (rc) { xcx_plan_destroy(p); return rc; }
  *out = p;
  return XCX_OK;
}

// all pointers are device pointers; out row r at out + r*ld_out, L floats
// conv_len > 0 (time shift of rows longer than one transform, timeshift.hip): ys are kernels of conv_len = 2n - 1 samples
// (two periods), xs the rows of n samples; out[r][n - 1 - k] = sum_t ys[r][t + k] xs[r][t] for k in [0, n) -- ONE pass
// with the kernel in the segment role, no statistics, no scaling: the n retained values are the circular convolution.
// psign (convolution mode): the rows come in twos and output row t = conv(row 2t) + psign[t] conv(row 2t + 1), summed in
// the frequency domain (k_seg_gram_sum): rows / 2 output rows.
int xcx_internal_seg_run(xcx_plan* p, const float* const* xs_h, const float* const* ys_h, int rows, float* out, long ld_out,
                   long conv_len, const int* psign) {
  xcx_ctx* ctx = p->ctx;
  xcx_plan::Seg& sg = *p->seg;
  const Geom& g = p->g;
  const long n = g.n, L = p->L;
  const int Bk = sg.Bk, Kb = sg.Kb, npr = sg.npr;
  const int nb = (int)((n + Bk - 1) / Bk);
  const bool conv = conv_len > 0;
  const int flags = conv ? 0 : g.flags;
  const long lmin = conv ? -(n - 1) : (long)g.idx_min - (n - 1);                        // retained lags, inclusive
  const long lmax = conv ? 0 : (long)g.idx_max - 1 - (n - 1);
  const long idx_min_eff = conv ? 0 : (long)g.idx_min, L_eff = conv ? n : L;
  int rc;
  // the distinct windows of the call: statistics and transforms belong to a window, not to a row
  std::vector<const float*> uptr;
  std::vector<int> ix(rows), iy(rows);
  {
    std::unordered_map<const float*, int> id;
    auto get = [&](const float* q) {
      auto it = id.find(q);
      if (it != id.end()) return it->second;
      const int k = (int)uptr.size();
      id.emplace(q, k); uptr.push_back(q);
      return k;
    };
    for (int r = 0; r < rows; ++r) { ix[r] = get(xs_h[r]); iy[r] = get(ys_h[r]); }
  }
  const int nu = (int)uptr.size();
  if ((rc = xcx_internal_upload(ctx, sg.ptrs, uptr.data(), sizeof(float*) * (size_t)nu))) return rc;
  if ((rc = sg.stats.ensure(ctx, sizeof(double) * (2 + kSegParts) * (size_t)nu))) return rc;   // totals, then partial sums
  const float* const* dptr = (const float* const*)sg.ptrs.ptr;
  double* su = (double*)sg.stats.ptr;
  HIP_TRY(hipMemsetAsync(su, 0, sizeof(double) * 2 * (size_t)nu, ctx->stream));
  double* part = su + 2 * (size_t)nu;
  for (int phase = 0; phase < (conv ? 0 : (flags & XCX_F_NORMALIZE) ? 2 : 1); ++phase) {
    hipLaunchKernelGGL(k_seg_stats, dim3(kSegParts, nu), dim3(256), 0, ctx->stream, dptr, n, flags, (const double*)su, part, phase);
    hipLaunchKernelGGL(k_seg_stats_fin, dim3((nu + 63) / 64), dim3(64), 0, ctx->stream, (const double*)part, n, su, nu, phase);
  }
  HIP_TRY(hipGetLastError());
  xcx_plan* sub = sg.sub;
  const size_t pitch = sub->spec_pitch;
  const int nfreq = sub->g.N1 * (sub->g.N2h + 1);
  // two passes: non-negative lags (a = x, b = y), then negative lags (a = y, b = x, lags 1..-lmin)
  for (int pass = 0; pass < 2; ++pass) {
    long k0, kend;            // lags [k0, kend) of C[k] = sum a[t+k] b[t]
    if (pass == 0) { k0 = std::max<long>(lmin, 0); kend = lmax + 1; }
    else { k0 = std::max<long>(1, -lmax); kend = -lmin + 1; }
    if (conv) {                                   // lags 0 .. n-1 of (kernel, row), written at n - 1 - k
      if (pass == 0) continue;
      k0 = 0; kend = n;
    }
    if (kend <= k0) continue;
    const int nchunks = (int)((kend - k0 + Kb - 1) / Kb);
    const std::vector<int>& ida = pass == 0 ? ix : iy;
    const std::vector<int>& idb = pass == 0 ? iy : ix;
    const int NS = nb - 1 + nchunks;
    // bytes per distinct a-window (NS segments: samples + spectra), per distinct b-window (nb blocks), per row
    // (nchunks cross-spectra + Kb lags each)
    const size_t per_win = (size_t)npr * sizeof(float) + pitch * sizeof(cf);
    const size_t per_a = (size_t)NS * per_win, per_b = (size_t)nb * per_win;
    const size_t per_row = (size_t)nchunks * (pitch * sizeof(cf) + (size_t)Kb * sizeof(float));
    // workspace of one chunk of rows: half of what the device has free, between 4 and 64 GB (XCX_SEG_BUDGET_MB overrides);
    // every chunk transforms the windows its rows meet, so three chunks of a 64-station day transformed every
    // station's blocks three times
    size_t budget = (size_t)16 << 30;
    {
      const char* eb = getenv("XCX_SEG_BUDGET_MB");           // read per call (tests)
      const long env_mb = eb ? atol(eb) : 0;
      size_t free_b = 0, total_b = 0;
      if (env_mb > 0) budget = (size_t)env_mb << 20;
      else {
        (void)ctx_reclaim(ctx);                  // buffers idling in the caller's pool count as free memory
        const char* ef = getenv("XCX_SEG_FREE_MB");             // tests: pretend this much is free
        if (ef && atol(ef) > 0) { free_b = (size_t)atol(ef) << 20; }
        else if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = (size_t)32 << 30; }
        // half of what is free, at most 64 GB; no floor above the free memory itself (a 4 GB floor with 3 GB free
        // failed where smaller chunks run); 64 MB is what one row of the largest block transform needs
        budget = std::min<size_t>((size_t)XCX_SEG_WORKSPACE_MAX_GB << 30, std::max<size_t>((size_t)64 << 20, free_b / 2));
      }
    }
    std::vector<int> ua, ub, ia, ib, ints;
    std::unordered_map<int, int> posa, posb;
    for (int r0 = 0; r0 < rows;) {
      // rows of the chunk: as many as fit the budget with their distinct windows (at most 4096: grid z of the scatter)
      ua.clear(); ub.clear(); ia.clear(); ib.clear(); posa.clear(); posb.clear();
      int rr = 0;
      size_t bytes = 0;
      while (r0 + rr < rows && rr < 4096) {
        const int a = ida[r0 + rr], b = idb[r0 + rr];
        const size_t add = per_row + (posa.count(a) ? 0 : per_a) + (posb.count(b) ? 0 : per_b);
        if (rr > 0 && !(psign && (rr & 1)) && bytes + add > budget) break;
        bytes += add;
        if (!posa.count(a)) { posa.emplace(a, (int)ua.size()); ua.push_back(a); }
        if (!posb.count(b)) { posb.emplace(b, (int)ub.size()); ub.push_back(b); }
        ia.push_back(posa[a]); ib.push_back(posb[b]);
        ++rr;
      }
      const int na = (int)ua.size(), nbw = (int)ub.size();
      const int orr = psign ? rr / 2 : rr, or0 = psign ? r0 / 2 : r0;      // output rows of the chunk
      const int P = orr * nchunks;
      const size_t slotB = (size_t)na * NS, slots = slotB + (size_t)nbw * nb;
      if ((rc = sg.segbuf.ensure(ctx, sizeof(float) * slots * npr))) return rc;
      if ((rc = sub->spectra.ensure(ctx, sizeof(cf) * pitch * slots))) return rc;
      // int tables of the chunk in one upload: [ua | ub | ia | ib | rx | ry]
      ints.clear();
      ints.insert(ints.end(), ua.begin(), ua.end());
      ints.insert(ints.end(), ub.begin(), ub.end());
      ints.insert(ints.end(), ia.begin(), ia.end());
      ints.insert(ints.end(), ib.begin(), ib.end());
      ints.insert(ints.end(), ix.begin() + r0, ix.begin() + r0 + rr);
      ints.insert(ints.end(), iy.begin() + r0, iy.begin() + r0 + rr);
      // rows of one a-window in twos (k_seg_gram<SG, 2>), the odd ones out alone.  Twos are cut at even b-window
      // numbers so that the twos of neighbouring a-windows meet the same b-windows, and the groups run in the order
      // (block of kSegAB a-windows, b-window, a-window): while the b-windows sweep past, the block's a-windows stay in
      // the Infinity Cache and each b-window is fetched from memory once per block, not once per a-window (the spectra
      // of 64 stations are 5 GB: in row order every row fetched its 35 MB of b-blocks from HBM).
      const size_t g2_at = ints.size();
      int n2 = 0, n1 = 0;
      if (psign) {                                   // pair sums: group t = rows (2t, 2t + 1), then the signs
        n2 = orr;
        for (int t = 0; t < orr; ++t) { ints.push_back(2 * t); ints.push_back(2 * t + 1); }
        for (int t = 0; t < orr; ++t) ints.push_back(psign[or0 + t]);
      } else {
        static const int kSegAB = getenv("XCX_SEG_ABLOCK") ? std::max(1, atoi(getenv("XCX_SEG_ABLOCK"))) : 2;
        std::vector<int> ord(rr);
        for (int i = 0; i < rr; ++i) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](int p, int q) {
          return ia[p] != ia[q] ? ia[p] < ia[q] : (ib[p] != ib[q] ? ib[p] < ib[q] : p < q); });
        struct Grp { int r0, r1; };
        std::vector<Grp> twos, ones;
        std::vector<int> left;
        for (int i = 0; i < rr;) {
          int e = i;
          while (e < rr && ia[ord[e]] == ia[ord[i]]) ++e;
          left.clear();
          for (int q = i; q < e;) {
            if (q + 1 < e && (ib[ord[q]] >> 1) == (ib[ord[q + 1]] >> 1)) { twos.push_back({ord[q], ord[q + 1]}); q += 2; }
            else { left.push_back(ord[q]); ++q; }
          }
          for (size_t q = 0; q + 1 < left.size(); q += 2) twos.push_back({left[q], left[q + 1]});
          if (left.size() & 1) ones.push_back({left.back(), -1});
          i = e;
        }
        auto before = [&](const Grp& p, const Grp& q) {
          const int ap = ia[p.r0], aq = ia[q.r0];
          if (ap / kSegAB != aq / kSegAB) return ap / kSegAB < aq / kSegAB;
          if ((ib[p.r0] >> 1) != (ib[q.r0] >> 1)) return (ib[p.r0] >> 1) < (ib[q.r0] >> 1);
          return ap != aq ? ap < aq : p.r0 < q.r0;
        };
        std::sort(twos.begin(), twos.end(), before);
        std::sort(ones.begin(), ones.end(), before);
        n2 = (int)twos.size(); n1 = (int)ones.size();
        for (const Grp& t : twos) { ints.push_back(t.r0); ints.push_back(t.r1); }
        for (const Grp& t : ones) ints.push_back(t.r0);
      }
      if ((rc = xcx_internal_upload(ctx, sg.pairs, ints.data(), sizeof(int) * ints.size()))) return rc;
      const int* d_ua = (const int*)sg.pairs.ptr;
      const int* d_ub = d_ua + na;
      const int* d_ia = d_ub + nbw;
      const int* d_ib = d_ia + rr;
      const int* d_rx = d_ib + rr;
      const int* d_ry = d_rx + rr;
      const int* d_g2 = (const int*)sg.pairs.ptr + g2_at;
      const int* d_g1 = d_g2 + 2 * n2;                              // pair sums: the signs
      float* segA = (float*)sg.segbuf.ptr;
      float* segB = segA + slotB * npr;
      const unsigned gx = (unsigned)((npr + 256 * kSegU - 1) / (256 * kSegU));
      hipLaunchKernelGGL(k_seg_gather, dim3(gx, NS, na), dim3(256), 0, ctx->stream, dptr, (const double*)su, d_ua,
                         conv ? conv_len : n, flags, npr, NS, k0, (long)Kb, npr, segA);
      hipLaunchKernelGGL(k_seg_gather, dim3(gx, nb, nbw), dim3(256), 0, ctx->stream, dptr, (const double*)su, d_ub, n, flags,
                         npr, nb, 0L, (long)Bk, Bk, segB);
      HIP_TRY(hipGetLastError());
      if ((rc = seg_forward(sub, segA, (long)NS * npr, npr, NS, 0, 0, na * NS, 0))) return rc;
      if ((rc = seg_forward(sub, segB, (long)nb * npr, npr, nb, 0, 0, nbw * nb, (int)slotB))) return rc;
      // Orientation of the pass.  Non-negative lags of (a, b) (pass 0): the cross-spectra of (segment, block) through
      // `subp`, whose inverse writes the Kb lags of a chunk in rising order; exchanged roles / convolution: (block,
      // segment) through `sub`, whose lags -(Kb-1) .. 0 land in rising order of the output index as well.  Either way a
      // FULL chunk is a run of Kb consecutive output samples of the row, and the inverse writes it in place (the row's
      // 1 / (|x| |y|) went into the cross-spectrum); only the ragged last chunk -- its tail would run into the next
      // row -- and rows that want the unbias weights go through the scatter kernel.
      const bool seg_first = pass == 0;
      xcx_plan* inv = seg_first ? sg.subp : sub;
      const bool direct = !(flags & XCX_F_UNBIAS);
      const int nfull = direct ? (int)((kend - k0) / Kb) : 0;          // chunks written in place
      const int ne = nchunks - nfull;                                   // chunks through the staging rows
      if ((rc = sub->gramC.ensure(ctx, sizeof(cf) * pitch * (size_t)P))) return rc;      // one buffer for both orientations
      const SegGramVar& gv = seg_gram_pick(nchunks, nb);
      const int gy = (nchunks + gv.sg - 1) / gv.sg, tiles8 = ((nfreq + 63) / 64 + 7) / 8;      // tiles per XCD
      auto gram_grid = [&](int G) { return dim3((unsigned)(8 * (size_t)tiles8 * ((G + 3) / 4) * gy)); };
      const int scaled = direct && (flags & XCX_F_NORMALIZE) ? 1 : 0;
      if (psign)
        hipLaunchKernelGGL(gv.fsum, gram_grid(n2), dim3(256), 0, ctx->stream, (const cf*)sub->spectra.ptr, pitch, nfreq, NS, nb,
                           nchunks, slotB, d_ia, d_ib, d_g2, d_g1, n2, gy, (cf*)sub->gramC.ptr);
      else if (n2 > 0)
        hipLaunchKernelGGL(gv.fn2, gram_grid(n2), dim3(256), 0, ctx->stream, (const cf*)sub->spectra.ptr, pitch, nfreq,
                           NS, nb, nchunks, slotB, d_ia, d_ib, d_g2, n2, gy, (cf*)sub->gramC.ptr, (const double*)su, d_rx, d_ry,
                           scaled, seg_first ? 1 : 0);
      if (!psign && n1 > 0)
        hipLaunchKernelGGL(gv.fn1, gram_grid(n1), dim3(256), 0, ctx->stream, (const cf*)sub->spectra.ptr, pitch, nfreq,
                           NS, nb, nchunks, slotB, d_ia, d_ib, d_g1, n1, gy, (cf*)sub->gramC.ptr, (const double*)su, d_rx, d_ry,
                           scaled, seg_first ? 1 : 0);
      HIP_TRY(hipGetLastError());
      // one inverse per (row, chunk): descriptor (slot q = row of the cross-spectrum buffer, output offset).  First the
      // chunks written in place (offsets into `out`), then the staged ones ([row][edge chunk][Kb])
      inv->hdescs.resize(P);
      {
        int d = 0;
        for (int r = 0; r < orr; ++r)
          for (int c = 0; c < nfull; ++c) {
            const long kc = k0 + (long)c * Kb;                          // first lag of the chunk
            const long j0 = seg_first ? (n - 1) + kc - idx_min_eff : (n - 1) - (kc + Kb - 1) - idx_min_eff;
            inv->hdescs[d++] = PwDesc{r * nchunks + c, r * nchunks + c, (long long)(or0 + r) * ld_out + j0};
          }
        for (int r = 0; r < orr; ++r)
          for (int c = nfull; c < nchunks; ++c)
            inv->hdescs[d++] = PwDesc{r * nchunks + c, r * nchunks + c, ((long long)r * ne + (c - nfull)) * Kb};
      }
      inv->desc_key = 0; inv->gram_key = 0;
      if ((rc = xcx_internal_upload(ctx, inv->descs, inv->hdescs.data(), (size_t)P * sizeof(PwDesc)))) return rc;
      const int Pd = orr * nfull;
      if (Pd > 0 && (rc = xcx_internal_stack_inverse(inv, Pd, out, 0, (const cf*)sub->gramC.ptr))) return rc;
      if (ne > 0) {
        if ((rc = sg.subout.ensure(ctx, sizeof(float) * (size_t)orr * ne * Kb))) return rc;
        if ((rc = xcx_internal_stack_inverse(inv, orr * ne, (float*)sg.subout.ptr, Pd, (const cf*)sub->gramC.ptr))) return rc;
        hipLaunchKernelGGL(k_seg_scatter, dim3((Kb + 256 * kSegU - 1) / (256 * kSegU), ne, orr), dim3(256), 0, ctx->stream,
                           (const float*)sg.subout.ptr, (const double*)su, d_rx, d_ry, scaled ? (flags & ~XCX_F_NORMALIZE) : flags,
                           n, Kb, ne, nfull, k0, kend, pass == 0 ? 1 : -1, seg_first ? 0 : 1, idx_min_eff, L_eff,
                           out + (size_t)or0 * ld_out, ld_out);
        HIP_TRY(hipGetLastError());
      }
      r0 += rr;
    }
  }
  return XCX_OK;
}
