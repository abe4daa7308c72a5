// sos.hip -- Butterworth second-order-section IIR filters on [batch][n]
// float32 rows, in place, float64 state (scipy semantics):
//   xcx_sosfiltfilt : scipy.signal.sosfiltfilt(sos, x, axis=-1)
//                     (xcorr/signal/filter.py:86-96) -- odd extension by
//                     3*ntaps samples, sosfilt_zi initial conditions, forward
//                     and backward pass, edges stripped
//   xcx_sosfilt     : scipy.signal.sosfilt from rest (ObsPy Trace.filter with
//                     zerophase=False, xcorr/stream/process.py:279-282)
//
// The recursion is sequential in time.  Parallelism comes from (a) one lane
// per row when there are many rows (post-correlation CCFs: pairs x windows
// rows), and (b) for few long rows (day-long station traces) an exact
// chunked scan: every row is cut into chunks of Lc samples, each chunk's
// zero-state response end state is computed in parallel, a tiny sequential
// pass propagates true start states with the D x D transition matrix A^Lc
// (D = 2 * sections), and the chunks are re-run from their true states.
// A wave handles 64 sequences; time is walked in tiles of 64 samples that are
// loaded/stored coalesced (256 B per sequence) and transposed through LDS.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
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

constexpr int kMaxK = (int)XCX_SOS_MAX_SECTIONS;          // sections
constexpr int kTile = 64;

struct SosCoef { double b0, b1, b2, a1, a2; };
// decimating output of a forward pass (xcx_decimate): the filtered samples t = 0, factor, 2 factor, ... of row r go to
// dst[r * stride + t / factor] and x is left as it is (factor <= 64)
struct Decim { float* dst = nullptr; long stride = 0; int factor = 1; };
struct SosParams { int K; SosCoef c[kMaxK]; };

template <int K>
__device__ __forceinline__ double sos_step(const SosParams& P, double (&z)[K][2], double v) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    // five float64 operations per section, explicitly fused: the compiler's six-operation form of the
    // same expressions costs the tiles kernel 6 % (SQ counters: a wave spends 25 % of its life issuing
    // VALU, ~2.25 waves per SIMD; profiles/r02_ablation.txt)
    const SosCoef c = P.c[k];
    const double y = fma(c.b0, v, z[k][0]);
    z[k][0] = fma(c.b1, v, fma(-c.a1, y, z[k][1]));
    z[k][1] = fma(c.b2, v, -c.a2 * y);
    v = y;
  }
  return v;
}

// value of lane `src` (wave-uniform index) as a scalar: v_readlane_b32 into SGPRs.  The tile loads and
// stores walk the 64 sequences of the workgroup one after the other and need sequence i's base address and
// length in every lane: as wave shuffles (ds_bpermute) those broadcasts were half of the kernel's LDS instructions.
__device__ __forceinline__ long lane_bcast(long v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xffffffffULL), src);
  const int hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), src);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}

// sequence q -> (row, chunk); chunk covers [chunk*Lc, min(n, (chunk+1)*Lc))
// REVERSE walks each sequence from its last sample to its first.
// s0   : [nseq][2K] start states (null = rest)
// send : [nseq][2K] end states (null = not stored)
template <int K, bool REVERSE, bool WRITE>
__global__ __launch_bounds__(64) void k_sos_tiles(
    float* __restrict__ x, long row_stride, long n, long Lc, int nchunks,
    long nseq, SosParams P, const double* __restrict__ s0,
    double* __restrict__ send, Decim dc = Decim{}) {
  __shared__ float tile[kTile][kTile + 1];
  const int lane = threadIdx.x;
  const long q0 = (long)blockIdx.x * 64;
  const long q = q0 + lane;
  const bool live = q < nseq;
  long mylen = 0;
  double z[K][2];
#pragma unroll
  for (int k = 0; k < K; ++k) { z[k][0] = 0.0; z[k][1] = 0.0; }
  if (live) {
    const long chunk = q % nchunks;
    mylen = std::min(Lc, n - chunk * Lc);
    if (s0) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        z[k][0] = s0[q * 2 * K + 2 * k];
        z[k][1] = s0[q * 2 * K + 2 * k + 1];
      }
    }
  }
  // every lane knows where ITS sequence starts; the cooperative tile loads below fetch
  // sequence i's address and length from lane i (no 64-bit divisions in the loop)
  long mybase = 0;
  if (live) mybase = (q / nchunks) * row_stride + (q % nchunks) * Lc;
  // coalesced load: 64 consecutive samples of sequence i per instruction, all 64 sequences of a tile issued together
  // into registers (lanes beyond nact hold mylen = 0 and load nothing).  The NEXT tile is fetched while this one is
  // filtered and written back: with ~2.3 waves per SIMD (the LDS tile) the kernel otherwise sat out four memory round
  // trips per tile with nothing to do (skewing the sections alone, K independent chains, changed nothing).
  float nx[kTile];
  auto fetch = [&](long t0) {
#pragma unroll
    for (int k = 0; k < kTile; ++k) {
      const long base = lane_bcast(mybase, k);
      const long len = lane_bcast(mylen, k);
      const long u = t0 + lane;               // step index within the sequence
      const long t = REVERSE ? (len - 1 - u) : u;
      nx[k] = (u < len) ? x[base + t] : 0.f;
    }
  };
  fetch(0);
  for (long t0 = 0; t0 < Lc; t0 += kTile) {
#pragma unroll
    for (int k = 0; k < kTile; ++k) tile[k][lane] = nx[k];
    __syncthreads();
    if (t0 + kTile < Lc) fetch(t0 + kTile);
    const int steps = live ? (int)std::max(0L, std::min((long)kTile, mylen - t0)) : 0;
    if (K > 1 && __builtin_amdgcn_ballot_w64(live && steps != kTile) == 0) {
      // Whole tiles (all but the ragged end of a short last chunk), the sections SKEWED in time: in step s section k
      // works on sample s - k, taking what section k-1 produced one step earlier (pipe[k]).  Every section still sees
      // its inputs in order and performs the same operations -- the results are bit for bit those of the sample-by-
      // sample cascade -- but the K section updates of a step do not depend on each other: K chains of three dependent
      // float64 operations in flight instead of one chain of 3 K (the kernel runs at ~2.3 waves per SIMD on its LDS
      // tile; it was bound by that latency chain, not by the float64 issue rate).  Lanes without a sequence run on zeros.
      double pipe[K];
#pragma unroll
      for (int k = 0; k < K; ++k) pipe[k] = 0.0;
      auto section = [&](int k, double v) {
        const SosCoef c = P.c[k];
        const double y = fma(c.b0, v, z[k][0]);
        z[k][0] = fma(c.b1, v, fma(-c.a1, y, z[k][1]));
        z[k][1] = fma(c.b2, v, -c.a2 * y);
        return y;
      };
      // prologue: steps 0 .. K-2, sections 0 .. s
#pragma unroll
      for (int st = 0; st < K - 1; ++st) {
#pragma unroll
        for (int k = K - 2; k >= 0; --k)
          if (k <= st) pipe[k + 1] = section(k, k == 0 ? (double)tile[lane][st] : pipe[k]);
      }
      // steady state: every section busy
      for (int st = K - 1; st < kTile; ++st) {
        const double yl = section(K - 1, pipe[K - 1]);
#pragma unroll
        for (int k = K - 2; k >= 0; --k) pipe[k + 1] = section(k, k == 0 ? (double)tile[lane][st] : pipe[k]);
        if (WRITE) tile[lane][st - (K - 1)] = (float)yl;
      }
      // epilogue: steps kTile .. kTile + K - 2, sections s - kTile + 1 .. K - 1
#pragma unroll
      for (int e = 0; e < K - 1; ++e) {
        const double yl = section(K - 1, pipe[K - 1]);
#pragma unroll
        for (int k = K - 2; k >= 1; --k)
          if (k > e) pipe[k + 1] = section(k, pipe[k]);
        if (WRITE) tile[lane][kTile - (K - 1) + e] = (float)yl;
      }
    } else if (live) {
      for (int u = 0; u < steps; ++u) {
        const double y = sos_step<K>(P, z, (double)tile[lane][u]);
        if (WRITE) tile[lane][u] = (float)y;
      }
    }
    __syncthreads();
    if (WRITE && !REVERSE && dc.dst) {
      // every factor-th sample of the ROW: the tile of sequence i starts at row sample g0 = chunk Lc + t0 = factor qd + rd
      // (one division per lane and tile, for its own sequence); lane l holds sample g0 + l, kept iff (rd + l) % factor == 0.
      // (rd + l) / factor by a 16-bit reciprocal: exact for (rd + l) factor < 65536
      long dbase = 0;
      int rd = 0;
      if (live) {
        const long g0 = (q % nchunks) * Lc + t0;
        const long qd = g0 / dc.factor;
        rd = (int)(g0 - qd * dc.factor);
        dbase = (q / nchunks) * dc.stride + qd;
      }
      const unsigned M = (65536u + (unsigned)dc.factor - 1u) / (unsigned)dc.factor;
      for (int i0 = 0; i0 < 64; i0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tile[i0 + k][lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const long base = lane_bcast(dbase, i0 + k);
          const long len = lane_bcast(mylen, i0 + k);
          const unsigned e = (unsigned)__builtin_amdgcn_readlane(rd, i0 + k) + (unsigned)lane;
          const unsigned ed = (e * M) >> 16;
          if (t0 + lane < len && ed * (unsigned)dc.factor == e) dc.dst[base + ed] = v[k];
        }
      }
      __syncthreads();
    } else if (WRITE) {
      // 16 sequences per trip: the LDS reads of a trip are issued together (lanes beyond nact have len = 0)
      for (int i0 = 0; i0 < 64; i0 += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tile[i0 + k][lane];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const long base = lane_bcast(mybase, i0 + k);
          const long len = lane_bcast(mylen, i0 + k);
          const long u = t0 + lane;
          if (u < len) {
            const long t = REVERSE ? (len - 1 - u) : u;
            x[base + t] = v[k];
          }
        }
      }
      __syncthreads();
    }
  }
  if (live && send) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      send[q * 2 * K + 2 * k] = z[k][0];
      send[q * 2 * K + 2 * k + 1] = z[k][1];
    }
  }
}

// sequential propagation of chunk start states.  One row per 16-lane group:
// lane d holds state component s[d] and row d of the transition matrix; the
// matrix-vector product gathers s[e] with wave shuffles.
//   forward : S0[c] = s; s = Z[c] + M_c s,   c = 0..nchunks-1
//   reverse : c = nchunks-1..0
// M_full = A^Lc, M_last = A^(len of the last chunk); init: [rows][D] or null.
// Z (zero-state end states) is overwritten by the start states S0.
__global__ __launch_bounds__(64) void k_sos_scan(double* __restrict__ Z, const double* __restrict__ init,
                           const double* __restrict__ Mfull, const double* __restrict__ Mlast,
                           int D, int nchunks, long rows, int reverse) {
  const int lane = threadIdx.x;
  const int d = lane & 15;
  const long row = (long)blockIdx.x * 4 + (lane >> 4);
  const bool live = row < rows && d < D;
  const long rrow = row < rows ? row : rows - 1;       // keep every lane in the shuffles
  double mf[2 * kMaxK], ml[2 * kMaxK];
#pragma unroll
  for (int e = 0; e < 2 * kMaxK; ++e) {
    mf[e] = (d < D && e < D) ? Mfull[d * D + e] : 0.0;
    ml[e] = (d < D && e < D) ? Mlast[d * D + e] : 0.0;
  }
  double s = (live && init) ? init[rrow * D + d] : 0.0;
  double* zr = Z + (size_t)rrow * nchunks * D;
  int c = reverse ? nchunks - 1 : 0;
  double znext = live ? zr[(size_t)c * D + d] : 0.0;
  for (int i = 0; i < nchunks; ++i) {
    const int cn = reverse ? c - 1 : c + 1;
    const double zc = znext;
    if (i + 1 < nchunks) znext = live ? zr[(size_t)cn * D + d] : 0.0;
    double acc = zc;
    const bool last = c == nchunks - 1;
#pragma unroll
    for (int e = 0; e < 2 * kMaxK; ++e) {
      const double se = __shfl(s, (lane & 48) | e, 64);
      acc += (last ? ml[e] : mf[e]) * se;
    }
    if (live) zr[(size_t)c * D + d] = s;
    s = acc;
    c = cn;
  }
}

// ---- two-level scan over the chunks of long rows ------------------------------
// The start-state recurrence S0[i+1] = Z[i] + M_i S0[i] (i = processing order) is affine
// and all M_i are powers of one matrix, so it splits into groups of kScanG chunks:
//   A  k_sos_scan_local : per (row, group) the recurrence from a zero start; Z[i] <- local
//                         prefix T[i], E[row][group] <- group end value
//   B  k_sos_scan       : the same sequential kernel over the G group values (66 steps
//                         instead of 4219 for a day at Lc = 1024)
//   C  k_sos_scan_apply : S0[i] = T[i] + A^(samples before i inside its group) * group start
// Processing order i <-> chunk c = i (forward) or C-1-i (reverse); groups are formed in
// processing order and stored in chunk order (group index gc = reverse ? G-1-grp : grp), so
// the short last chunk sits in the last stored group in both directions.
constexpr int kScanG = 64;

__global__ __launch_bounds__(64) void k_sos_scan_local(double* __restrict__ Z, double* __restrict__ E,
                                                       const double* __restrict__ Mfull,
                                                       const double* __restrict__ Mlast, int D, int nchunks,
                                                       int G, long rows, int reverse) {
  const int lane = threadIdx.x;
  const int d = lane & 15;
  const long vr = (long)blockIdx.x * 4 + (lane >> 4);          // virtual row = (row, group in processing order)
  const long nvr = rows * G;
  const long vrc = vr < nvr ? vr : nvr - 1;                      // keep every lane in the shuffles
  const long row = vrc / G;
  const int grp = (int)(vrc % G);
  const bool live = vr < nvr && d < D;
  double mf[2 * kMaxK], ml[2 * kMaxK];
#pragma unroll
  for (int e = 0; e < 2 * kMaxK; ++e) {
    mf[e] = (d < D && e < D) ? Mfull[d * D + e] : 0.0;
    ml[e] = (d < D && e < D) ? Mlast[d * D + e] : 0.0;
  }
  double* zr = Z + (size_t)row * nchunks * D;
  const int i0 = grp * kScanG;
  const int cnt = min(kScanG, nchunks - i0);
  double t = 0.0;
  for (int j = 0; j < cnt; ++j) {
    const int i = i0 + j;
    const int c = reverse ? nchunks - 1 - i : i;
    const double zc = live ? zr[(size_t)c * D + d] : 0.0;
    double acc = zc;
    const bool last = c == nchunks - 1;
#pragma unroll
    for (int e = 0; e < 2 * kMaxK; ++e) {
      const double te = __shfl(t, (lane & 48) | e, 64);
      acc += (last ? ml[e] : mf[e]) * te;
    }
    if (live) zr[(size_t)c * D + d] = t;
    t = acc;
  }
  const int gc = reverse ? G - 1 - grp : grp;
  if (live) E[((size_t)row * G + gc) * D + d] = t;
}

// PW: [kScanG+1][D][D] powers A^(j*Lc); PWs: the same for the group that starts with the
// short chunk (reverse passes): PWs[0] = I, PWs[j] = A^(last_len + (j-1) Lc)
__global__ void k_sos_scan_apply(double* __restrict__ Z, const double* __restrict__ E,
                                 const double* __restrict__ PW, const double* __restrict__ PWs, int D,
                                 int nchunks, int G, long rows, int reverse) {
  const long total = rows * nchunks * D;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const int d = (int)(q % D);
    const long rc = q / D;
    const int c = (int)(rc % nchunks);
    const long row = rc / nchunks;
    const int i = reverse ? nchunks - 1 - c : c;
    const int grp = i / kScanG, j = i % kScanG;
    const int gc = reverse ? G - 1 - grp : grp;
    const double* __restrict__ sg = E + ((size_t)row * G + gc) * D;
    const double* __restrict__ Pj = ((reverse && grp == 0) ? PWs : PW) + (size_t)j * D * D + (size_t)d * D;
    double acc = Z[q];
    for (int e = 0; e < D; ++e) acc += Pj[e] * sg[e];
    Z[q] = acc;
  }
}

// sosfiltfilt edges.  One thread per row.
// phase 0 (before the forward pass):
//    save tail[row][0..edge] = x[n-1-edge .. n-1];
//    run the left odd extension 2x[0]-x[edge-i] from zi*(2x[0]-x[edge]);
//    init[row] = state after the extension.
// phase 1 (after the forward pass; x holds y_f):
//    state = fend[row]; run the right extension 2x[n-1]-x[n-2-i] -> yr[i];
//    backward start state = zi * yr[edge-1]; run yr reversed;
//    init[row] = state before the backward main pass.
template <int K>
__global__ void k_sos_edges(const float* __restrict__ x, long row_stride, long n,
                            long rows, int edge, SosParams P,
                            const double* __restrict__ zi, float* __restrict__ tail,
                            const double* __restrict__ fend, double* __restrict__ init,
                            int phase) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float* xr = x + row * row_stride;
  float* tl = tail + row * (edge + 1);
  double z[K][2];
  if (phase == 0) {
    for (int j = 0; j <= edge; ++j) tl[j] = xr[n - 1 - edge + j];
    const double x0 = (double)xr[0];
    const double e0 = 2.0 * x0 - (double)xr[edge];
#pragma unroll
    for (int k = 0; k < K; ++k) { z[k][0] = zi[2 * k] * e0; z[k][1] = zi[2 * k + 1] * e0; }
    for (int i = 0; i < edge; ++i) sos_step<K>(P, z, 2.0 * x0 - (double)xr[edge - i]);
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) { z[k][0] = fend[row * 2 * K + 2 * k]; z[k][1] = fend[row * 2 * K + 2 * k + 1]; }
    const double xl = (double)tl[edge];
    double yr[3 * (2 * kMaxK + 1)];
    for (int i = 0; i < edge; ++i) yr[i] = sos_step<K>(P, z, 2.0 * xl - (double)tl[edge - 1 - i]);
    const double y0 = yr[edge - 1];
#pragma unroll
    for (int k = 0; k < K; ++k) { z[k][0] = zi[2 * k] * y0; z[k][1] = zi[2 * k + 1] * y0; }
    for (int i = edge - 1; i >= 0; --i) sos_step<K>(P, z, yr[i]);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { init[row * 2 * K + 2 * k] = z[k][0]; init[row * 2 * K + 2 * k + 1] = z[k][1]; }
}

// ---- host helpers -----------------------------------------------------
typedef long double ld;

void step_zero_input(const SosParams& P, std::vector<ld>& s) {
  ld v = 0;
  for (int k = 0; k < P.K; ++k) {
    const SosCoef& c = P.c[k];
    ld y = (ld)c.b0 * v + s[2 * k];
    s[2 * k] = (ld)c.b1 * v - (ld)c.a1 * y + s[2 * k + 1];
    s[2 * k + 1] = (ld)c.b2 * v - (ld)c.a2 * y;
    v = y;
  }
}

std::vector<ld> mat_mul(const std::vector<ld>& a, const std::vector<ld>& b, int D) {
  std::vector<ld> c((size_t)D * D, 0);
  for (int i = 0; i < D; ++i)
    for (int k = 0; k < D; ++k) {
      ld aik = a[i * D + k];
      for (int j = 0; j < D; ++j) c[i * D + j] += aik * b[k * D + j];
    }
  return c;
}

// A^e for the one-step zero-input transition A of the cascade
std::vector<ld> transition_power_ld(const SosParams& P, long e);
std::vector<double> transition_power(const SosParams& P, long e) {
  const std::vector<ld> R = transition_power_ld(P, e);
  std::vector<double> out(R.size());
  for (size_t i = 0; i < out.size(); ++i) out[i] = (double)R[i];
  return out;
}
std::vector<ld> transition_power_ld(const SosParams& P, long e) {
  const int D = 2 * P.K;
  std::vector<ld> A((size_t)D * D, 0), R((size_t)D * D, 0);
  for (int j = 0; j < D; ++j) {
    std::vector<ld> s(D, 0);
    s[j] = 1;
    step_zero_input(P, s);
    for (int i = 0; i < D; ++i) A[i * D + j] = s[i];
    R[j * D + j] = 1;
  }
  while (e > 0) {
    if (e & 1) R = mat_mul(R, A, D);
    A = mat_mul(A, A, D);
    e >>= 1;
  }
  return R;
}

int parse_sos(const double* sos, int K, SosParams& P) {
  if (!sos) return set_err(XCX_EINVAL, "sos is null");
  if (K < 1 || K > kMaxK) return set_err(XCX_EUNSUPPORTED, "n_sections must be in [1,%d]", kMaxK);
  P.K = K;
  for (int k = 0; k < K; ++k) {
    const double* r = sos + 6 * k;
    if (r[3] == 0.0) return set_err(XCX_EINVAL, "sos[%d][3] (a0) is zero", k);
    P.c[k] = SosCoef{r[0] / r[3], r[1] / r[3], r[2] / r[3], r[4] / r[3], r[5] / r[3]};
  }
  return XCX_OK;
}

// scipy.signal.sosfilt_zi: per section solve (I - A^T) zi = B, scaled by the
// cumulative DC gain of the preceding sections
void sosfilt_zi(const SosParams& P, std::vector<double>& zi) {
  zi.assign(2 * P.K, 0.0);
  double scale = 1.0;
  for (int k = 0; k < P.K; ++k) {
    const SosCoef& c = P.c[k];
    // [[1+a1, -1], [a2, 1]] [z0, z1]^T = [b1 - a1 b0, b2 - a2 b0]^T
    const double m00 = 1.0 + c.a1, m01 = -1.0, m10 = c.a2, m11 = 1.0;
    const double r0 = c.b1 - c.a1 * c.b0, r1 = c.b2 - c.a2 * c.b0;
    const double det = m00 * m11 - m01 * m10;
    zi[2 * k] = scale * (r0 * m11 - m01 * r1) / det;
    zi[2 * k + 1] = scale * (m00 * r1 - m10 * r0) / det;
    scale *= (c.b0 + c.b1 + c.b2) / (1.0 + c.a1 + c.a2);
  }
}

template <int K, bool REV, bool WR>
hipError_t launch_tiles(xcx_ctx* ctx, float* x, long row_stride, long n, long Lc, int nchunks,
                        long nseq, const SosParams& P, const double* s0, double* send, Decim dc = Decim{}) {
  const unsigned grid = (unsigned)((nseq + 63) / 64);
  hipLaunchKernelGGL((k_sos_tiles<K, REV, WR>), dim3(grid), dim3(64), 0, ctx->stream, x, row_stride,
                     n, Lc, nchunks, nseq, P, s0, send, dc);
  return hipGetLastError();
}

// one directional pass over every row from per-row start states `init`
// ([rows][2K] or null = rest); stores the per-row end states in `fend` if
// non-null.
template <int K, bool REV>
int run_pass(xcx_ctx* ctx, float* x, long rows, long n, const SosParams& P,
             const double* init, double* fend, DevBuf& zbuf, DevBuf& /*unused: matrices live in ctx->sos_tab*/,
             long row_stride = 0, Decim dc = Decim{}) {
  if (row_stride == 0) row_stride = n;                 // dense rows; > n: a column range of wider rows
  const int D = 2 * K;
  long Lc = n;
  int nchunks = 1;
  if (rows < 8192 && n > 4096) {          // few long rows: chunked scan
    Lc = 1024;
    // a wave walks 64 chunks; a handful of day-long rows at 1024 samples per chunk is ~1 000 waves, one per SIMD, and the
    // kernel then waits out every memory round trip and dependent operation alone (SQ counters: 34 % of a wave's cycles
    // issue VALU, the CUs hold 1-2 waves of the 9 their LDS allows): shorter chunks until there are ~8 waves per CU
    while (Lc > 128 && rows * ((n + Lc - 1) / Lc) < 131072) Lc /= 2;
    while ((n + Lc - 1) / Lc > 65536) Lc *= 2;
    nchunks = (int)((n + Lc - 1) / Lc);
  }
  const long nseq = rows * nchunks;
  if (nchunks == 1) {
    HIP_TRY((launch_tiles<K, REV, true>(ctx, x, row_stride, n, Lc, 1, nseq, P, init, fend, dc)));
    return XCX_OK;
  }
  int rc;
  const int G = (nchunks + kScanG - 1) / kScanG;
  if ((rc = zbuf.ensure(ctx, sizeof(double) * D * ((size_t)nseq + (size_t)rows * G)))) return rc;
  const long last_len = n - (long)(nchunks - 1) * Lc;
  // transition-matrix powers (float64 from long double on the host), cached per filter/geometry
  const size_t nmat = 2 * (size_t)(kScanG + 1);
  unsigned long long key = 1469598103934665603ULL;
  auto mix = [&](const void* p, size_t bytes) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < bytes; ++i) { key ^= c[i]; key *= 1099511628211ULL; }
  };
  mix(&P.K, sizeof P.K); mix(P.c, sizeof(SosCoef) * P.K); mix(&Lc, sizeof Lc); mix(&last_len, sizeof last_len);
  int slot = -1;
  for (int i = 0; i < xcx_ctx::kSosTabs; ++i)
    if (ctx->sos_key[i] == key && ctx->sos_tab[i].ptr) slot = i;
  if (slot < 0) {
    // A^(j Lc) and A^(last_len + (j-1) Lc), j = 0 .. kScanG, by ONE long-double product per table entry from A^Lc and
    // A^last_len (computing every power from scratch by repeated squaring cost 12 ms for a 6-section filter)
    std::vector<double> tab(nmat * D * D);
    const std::vector<ld> M = transition_power_ld(P, Lc), Ml = transition_power_ld(P, last_len);
    std::vector<ld> a((size_t)D * D, 0), b((size_t)D * D, 0);
    for (int i = 0; i < D; ++i) a[(size_t)i * D + i] = b[(size_t)i * D + i] = 1;      // j = 0: identity for both
    for (int j = 0; j <= kScanG; ++j) {
      if (j == 1) b = Ml;
      else if (j > 1) b = mat_mul(b, M, D);
      if (j >= 1) a = mat_mul(a, M, D);
      for (size_t i = 0; i < (size_t)D * D; ++i) {
        tab[(size_t)j * D * D + i] = (double)a[i];
        tab[((size_t)(kScanG + 1) + j) * D * D + i] = (double)b[i];
      }
    }
    slot = ctx->sos_next;
    ctx->sos_next = (ctx->sos_next + 1) % xcx_ctx::kSosTabs;
    ctx->sos_key[slot] = 0;
    if ((rc = ctx->sos_tab[slot].ensure(ctx, sizeof(double) * tab.size()))) return rc;
    // the slot may still be read by queued kernels of an earlier call: the copy is stream-ordered behind them
    HIP_TRY(hipMemcpyAsync(ctx->sos_tab[slot].ptr, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->sos_key[slot] = key;
  }
  const double* PW = (const double*)ctx->sos_tab[slot].ptr;
  const double* PWs = PW + (size_t)(kScanG + 1) * D * D;
  const double* Mfull = PW + (size_t)D * D;                  // A^Lc
  const double* Mlast = PWs + (size_t)D * D;                 // A^last_len
  double* Z = (double*)zbuf.ptr;
  double* E = Z + (size_t)D * nseq;
  HIP_TRY((launch_tiles<K, REV, false>(ctx, x, row_stride, n, Lc, nchunks, nseq, P, nullptr, Z)));
  hipLaunchKernelGGL(k_sos_scan_local, dim3((unsigned)((rows * G + 3) / 4)), dim3(64), 0, ctx->stream, Z, E, Mfull, Mlast,
                     D, nchunks, G, rows, (int)REV);
  // group start states: the sequential kernel over G values; a full group advances by
  // A^(kScanG Lc), the stored-last group (which holds the short chunk) by PWs[kScanG]
  hipLaunchKernelGGL(k_sos_scan, dim3((unsigned)((rows + 3) / 4)), dim3(64), 0, ctx->stream, E, init,
                     PW + (size_t)kScanG * D * D, PWs + (size_t)kScanG * D * D, D, G, rows, (int)REV);
  hipLaunchKernelGGL(k_sos_scan_apply, dim3((unsigned)std::min<long>((rows * nchunks * D + 255) / 256, 8192)), dim3(256), 0,
                     ctx->stream, Z, (const double*)E, PW, PWs, D, nchunks, G, rows, (int)REV);
  HIP_TRY(hipGetLastError());
  // re-run from the true start states; end states land back in Z
  HIP_TRY((launch_tiles<K, REV, true>(ctx, x, row_stride, n, Lc, nchunks, nseq, P, Z, fend ? Z : nullptr, dc)));
  if (fend) {
    // the row's end state is that of its last-processed chunk
    const int endc = REV ? 0 : nchunks - 1;
    HIP_TRY(hipMemcpy2DAsync(fend, sizeof(double) * D, Z + (size_t)endc * D, sizeof(double) * D * nchunks,
                             sizeof(double) * D, rows, hipMemcpyDeviceToDevice, ctx->stream));
  }
  return XCX_OK;
}

enum { kModeForward = 0, kModeFiltFilt = 1, kModeZeroPhase = 2 };

template <int K>
int filt_impl(xcx_ctx* ctx, float* dx, long rows, long n, const SosParams& P, int mode) {
  const bool filtfilt = mode == kModeFiltFilt;
  // grow-only per-context workspaces (no allocation in the steady state)
  DevBuf &zbuf = ctx->scratch[1], &mbuf = ctx->scratch[2], &zi_d = ctx->scratch[3],
         &tail = ctx->scratch[4], &init = ctx->scratch[5], &fend = ctx->scratch[6];
  int rc = XCX_OK;
  auto cleanup = [&]() {};
  if (!filtfilt) {
    rc = run_pass<K, false>(ctx, dx, rows, n, P, nullptr, nullptr, zbuf, mbuf);
    // ObsPy zerophase=True: the same filter once more over the reversed signal, again from rest
    // (no padding, no steady-state start values -- unlike scipy.signal.sosfiltfilt)
    if (rc == XCX_OK && mode == kModeZeroPhase)
      rc = run_pass<K, true>(ctx, dx, rows, n, P, nullptr, nullptr, zbuf, mbuf);
    cleanup();
    return rc;
  }
  // scipy: ntaps = 2*n_sections + 1 - min(#(b2 == 0), #(a2 == 0)); edge = 3*ntaps
  int zb = 0, za = 0;
  for (int k = 0; k < K; ++k) { zb += (P.c[k].b2 == 0.0); za += (P.c[k].a2 == 0.0); }
  const int ntaps = 2 * K + 1 - std::min(zb, za);
  const int edge = 3 * ntaps;
  if (n <= edge) {
    return set_err(XCX_EINVAL, "The length of the input vector x must be greater than padlen, which is %d.", edge);
  }
  std::vector<double> zi;
  sosfilt_zi(P, zi);
  const int D = 2 * K;
  if ((rc = zi_d.ensure(ctx, sizeof(double) * D)) ||
      (rc = tail.ensure(ctx, sizeof(float) * (edge + 1) * (size_t)rows)) ||
      (rc = init.ensure(ctx, sizeof(double) * D * (size_t)rows)) ||
      (rc = fend.ensure(ctx, sizeof(double) * D * (size_t)rows))) { cleanup(); return rc; }
  if (hipMemcpyAsync(zi_d.ptr, zi.data(), sizeof(double) * D, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); return set_err(XCX_EHIP, "zi upload failed"); }
  const unsigned g = (unsigned)((rows + 63) / 64);
  hipLaunchKernelGGL((k_sos_edges<K>), dim3(g), dim3(64), 0, ctx->stream, dx, n, n, rows, edge, P,
                     (const double*)zi_d.ptr, (float*)tail.ptr, (const double*)nullptr, (double*)init.ptr, 0);
  rc = run_pass<K, false>(ctx, dx, rows, n, P, (const double*)init.ptr, (double*)fend.ptr, zbuf, mbuf);
  if (rc) { cleanup(); return rc; }
  hipLaunchKernelGGL((k_sos_edges<K>), dim3(g), dim3(64), 0, ctx->stream, dx, n, n, rows, edge, P,
                     (const double*)zi_d.ptr, (float*)tail.ptr, (const double*)fend.ptr, (double*)init.ptr, 1);
  rc = run_pass<K, true>(ctx, dx, rows, n, P, (const double*)init.ptr, nullptr, zbuf, mbuf);
  hipError_t e = hipGetLastError();
  cleanup();
  if (rc) return rc;
  if (e != hipSuccess) return set_err(XCX_EHIP, "sosfiltfilt launch: %s", hipGetErrorString(e));
  return XCX_OK;
}

int filt_entry(xcx_ctx* ctx, float* x, long batch, long n, const double* sos, int K, int on_device,
               int filtfilt) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1) return set_err(XCX_EINVAL, "batch and n must be >= 1");
  SosParams P;
  int rc = parse_sos(sos, K, P);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  float* dx = x;
  const size_t bytes = (size_t)batch * n * sizeof(float);
  HostPin pin_x(on_device ? nullptr : x, bytes);
  if (!on_device) {
    HIP_TRY(ctx_malloc(ctx, (void**)&dx, bytes));
    HIP_TRY(hipMemcpyAsync(dx, x, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  switch (K) {
    case 1: rc = filt_impl<1>(ctx, dx, batch, n, P, filtfilt); break;
    case 2: rc = filt_impl<2>(ctx, dx, batch, n, P, filtfilt); break;
    case 3: rc = filt_impl<3>(ctx, dx, batch, n, P, filtfilt); break;
    case 4: rc = filt_impl<4>(ctx, dx, batch, n, P, filtfilt); break;
    case 5: rc = filt_impl<5>(ctx, dx, batch, n, P, filtfilt); break;
    case 6: rc = filt_impl<6>(ctx, dx, batch, n, P, filtfilt); break;
    case 7: rc = filt_impl<7>(ctx, dx, batch, n, P, filtfilt); break;
    default: rc = filt_impl<8>(ctx, dx, batch, n, P, filtfilt); break;
  }
  if (!on_device) {
    if (rc == XCX_OK) {
      HIP_TRY(hipMemcpyAsync(x, dx, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(dx));
  }
  return rc;
}

}  // namespace

// xcx_decimate (resample.hip): the causal filter from rest, every factor-th output sample written to dst rows of `nout`
// samples, x untouched (device pointers): the zero-state pass reads x, the pass from the true chunk states reads it again
// and stores only what is kept -- no working copy, no in-place rewrite, no gather pass
int xcx_internal_sosfilt_decimate(xcx_ctx* ctx, const float* x, long batch, long n, const double* sos, int K, float* dst,
                                  long nout, long factor) {
  if (factor < 1 || factor > 64) return set_err(XCX_EINVAL, "decimating filter pass: factor %ld outside [1, 64]", factor);
  SosParams P;
  int rc = parse_sos(sos, K, P);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  DevBuf &zbuf = ctx->scratch[1], &mbuf = ctx->scratch[2];
  float* xx = const_cast<float*>(x);                  // never written: every writing pass has a destination of its own
  const Decim dc{dst, nout, (int)factor};
  switch (K) {
    case 1: return run_pass<1, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 2: return run_pass<2, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 3: return run_pass<3, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 4: return run_pass<4, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 5: return run_pass<5, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 6: return run_pass<6, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    case 7: return run_pass<7, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
    default: return run_pass<8, false>(ctx, xx, batch, n, P, nullptr, nullptr, zbuf, mbuf, 0, dc);
  }
}

extern "C" int xcx_sosfiltfilt(xcx_ctx* ctx, float* x, long batch, long n, const double* sos,
                               int n_sections, int on_device) {
  return filt_entry(ctx, x, batch, n, sos, n_sections, on_device, kModeFiltFilt);
}

extern "C" int xcx_sosfilt_zerophase(xcx_ctx* ctx, float* x, long batch, long n, const double* sos,
                                     int n_sections, int on_device) {
  return filt_entry(ctx, x, batch, n, sos, n_sections, on_device, kModeZeroPhase);
}

extern "C" int xcx_sosfilt(xcx_ctx* ctx, float* x, long batch, long n, const double* sos,
                           int n_sections, int on_device) {
  return filt_entry(ctx, x, batch, n, sos, n_sections, on_device, kModeForward);
}

// ---- column ranges of long rows (window-sharded front end) ----------------------
// A stable IIR forgets: after e samples the state started from rest differs from the true one by
// A^e s(t - e), A = zero-input transition of the cascade.  The warm-up is the smallest multiple of
// 1024 samples with ||A^e||_inf <= 2^-64 -- below the float64 rounding of the state itself for
// any state within 2^10 of the input amplitude, 9 decimal orders below the float32 output.
static long sos_warmup(const SosParams& P) {
  const int D = 2 * P.K;
  for (long e = 1024; e <= (1L << 40); e *= 2) {
    const std::vector<double> M = transition_power(P, e);
    double nrm = 0.0;
    bool finite = true;
    for (int i = 0; i < D; ++i) {
      double r = 0.0;
      for (int j = 0; j < D; ++j) { r += std::fabs(M[(size_t)i * D + j]); finite = finite && std::isfinite(M[(size_t)i * D + j]); }
      nrm = std::max(nrm, r);
    }
    if (!finite) break;
    if (nrm <= 0x1p-64) {
      // refine between e/2 and e in steps of 1024
      long lo = e / 2, hi = e;
      while (hi - lo > 1024) {
        const long mid = ((lo + hi) / 2 + 1023) / 1024 * 1024;
        if (mid >= hi) break;
        const std::vector<double> Q = transition_power(P, mid);
        double q = 0.0;
        for (int i = 0; i < D; ++i) { double r = 0.0; for (int j = 0; j < D; ++j) r += std::fabs(Q[(size_t)i * D + j]); q = std::max(q, r); }
        if (q <= 0x1p-64) hi = mid; else lo = mid;
      }
      return hi;
    }
  }
  return -1;                                           // not (numerically) stable: no finite warm-up
}

extern "C" int xcx_sosfilt_warmup(const double* sos, int n_sections, long* warm) {
  if (!warm) return set_err(XCX_EINVAL, "null argument");
  SosParams P;
  int rc = parse_sos(sos, n_sections, P);
  if (rc) return rc;
  *warm = sos_warmup(P);
  return XCX_OK;
}

extern "C" int xcx_sosfilt_cols(xcx_ctx* ctx, float* x, long batch, long n, const double* sos, int n_sections,
                                long col0, long col1, long* first_col) {
  if (!ctx || !x) return set_err(XCX_EINVAL, "null argument");
  if (batch < 1 || n < 1 || col0 < 0 || col1 > n || col0 >= col1) return set_err(XCX_EINVAL, "bad column range");
  SosParams P;
  int rc = parse_sos(sos, n_sections, P);
  if (rc) return rc;
  HIP_TRY(hipSetDevice(ctx->device));
  const long warm = sos_warmup(P);
  const long c0 = (warm < 0 || warm >= col0) ? 0 : col0 - warm;      // unstable or short prefix: from the row start
  if (first_col) *first_col = c0;
  DevBuf &zbuf = ctx->scratch[1], &mbuf = ctx->scratch[2];
  float* xs = x + c0;
  const long len = col1 - c0;
  switch (n_sections) {
    case 1: return run_pass<1, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 2: return run_pass<2, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 3: return run_pass<3, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 4: return run_pass<4, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 5: return run_pass<5, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 6: return run_pass<6, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    case 7: return run_pass<7, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
    default: return run_pass<8, false>(ctx, xs, batch, len, P, nullptr, nullptr, zbuf, mbuf, n);
  }
}
