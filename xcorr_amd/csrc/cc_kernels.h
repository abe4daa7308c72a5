// cc_kernels.h -- the four FFT kernels of the pair x window cross-correlation
// engine (templated on the sub-FFT radices), gfx950.
//
// Transform length N = N1 * N2 (N2 even, N2h = N2/2), N >= 2n-1.
// Index split (prototype: tools/proto_fft.py):
//     time      m = m1 + N1*m2      m1 in [0,N1), m2 in [0,N2)
//     frequency k = k1*N2 + k2      k1 in [0,N1), k2 in [0,N2)
// A station-window spectrum is stored in "column layout":
//     Xm[k1][k2]  k2 in [0,N2h)      (pitch N2h float2)
//     Xn[k1]      k2 = N2h            (the Nyquist column)
// holding X[k] for ALL k1 (so k runs past N/2; the Hermitian mirror is
// produced by the forward column FFT itself and the pair kernel needs no
// index flips: C[k] = X_i[k] conj(X_j[k]) everywhere).
//
//   forward   F1 rows   : R2C of length N2 over m2 for RB rows m1 (stride-N1
//                         gather of the zero-padded window), x w_N^{-m1 k2}
//             F2 columns: complex FFT of length N1 over m1 for CB columns,
//                         x scale (1/(||x|| sqrt N)), optional whitening
//   inverse   I1 columns: load X_i, X_j, multiply, complex IFFT length N1 over
//                         k1, x w_N^{+m1 k2}  -> T[m1][k2] (Hermitian in k2,
//                         Nyquist real packed into T[m1][0].y)
//             I2 rows   : C2R of length N2 over k2 for RB rows, lag rotation,
//                         clip to [idx_min, idx_max), coalesced store
//
// Reference semantics: xcorr/util/cc.py:60-79 and
// xcorr/signal/correlate.py:97-113 (see include/xcx.h).
#pragma once
#include "fft_device.h"

namespace xcx {

constexpr int kNT = 256;   // threads per workgroup
constexpr int kCB = 16;    // columns per workgroup (128-byte segments)
constexpr int kRB = 16;    // rows per workgroup
constexpr int kRBP = 17;   // LDS pitch of the row kernels (transposing loads)

enum : int {
  XCX_F_NORMALIZE = 1,
  XCX_F_UNBIAS = 2,
  XCX_F_STACK = 4,
  XCX_F_ONEBIT = 8,
  XCX_F_WHITEN = 16,
  XCX_F_DEMEAN = 32,
};

struct Geom {
  int n;        // window length (samples)
  int N;        // transform length
  int N1, N2, N2h;
  int idx_min, idx_max;   // retained lag slice of the 2n-1 lags
  int shift;    // (n-1-idx_min) mod N1 : output residue of row m1 = 0
  int flags;
  int band_hi;  // whitened plans: spectra are EXACTLY zero at folded bins min(k, N-k) > band_hi (the whitening
                // gain is 0 there), so the pair column pass never loads them; N = no band limit
};

// per station-window statistics produced by k_stats
struct SwStat { float mean; float scale; };

// ---------------------------------------------------------------------------
// F1: forward rows.  grid = (N1/RB, station-windows of the batch)
// ---------------------------------------------------------------------------
struct F1Args {
  const float* traces; long stride_s, stride_w; int W, w0, sw0;
  const SwStat* stats;
  cf* Tm; cf* Tn;            // [sw][N1][N2h], [sw][N1]
  const cf* tw_sub;          // W_{N2h}
  const cf* tw_unt;          // W_{N2}[k], k < N2h
  const cf* tw_N;            // two-level W_N
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kNT) void k_fwd_rows(F1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N2h = RX::L;
  const Geom g = a.g;
  const int sw = blockIdx.y;                 // index inside this forward chunk
  const int swg = a.sw0 + sw;                // index in the S x W grid
  const int s = swg / a.W, wl = swg % a.W;
  const int row0 = blockIdx.x * kRB;
  const float* __restrict__ x =
      a.traces + (long)s * a.stride_s + (long)(a.w0 + wl) * a.stride_w;
  const SwStat st = a.stats[sw];
  const int n = g.n, N1 = g.N1;
  const bool onebit = g.flags & XCX_F_ONEBIT;
  const float mean = st.mean;

  auto in = [=](int m, int r) -> cf {
    const int i0 = row0 + r + N1 * (2 * m);
    const int i1 = i0 + N1;
    float v0 = 0.f, v1 = 0.f;
    if (i0 < n) {
      v0 = x[i0] - mean;
      if (onebit) v0 = (v0 > 0.f) ? 1.f : ((v0 < 0.f) ? -1.f : 0.f);
    }
    if (i1 < n) {
      v1 = x[i1] - mean;
      if (onebit) v1 = (v1 > 0.f) ? 1.f : ((v1 < 0.f) ? -1.f : 0.f);
    }
    return make_float2(v0, v1);
  };
  auto to_lds = [=](int p, int r, cf v) { lds[p * kRBP + r] = v; };
  if constexpr (RX::S == 1) {
    fft_stage<N2h, RX::R0, 1, kRB, kNT, false, false>(a.tw_sub, in, to_lds);
  } else {
    // all stages through LDS; the last one lands in LDS as well
    auto lds_ld = [=](int p, int r) -> cf { return lds[p * kRBP + r]; };
    fft_stage<N2h, RX::R0, 1, kRB, kNT, false, false>(a.tw_sub, in, to_lds);
    __syncthreads();
    fft_stage<N2h, RX::R1, RX::R0, kRB, kNT, false, true>(a.tw_sub, lds_ld, to_lds);
    if constexpr (RX::S >= 3) {
      __syncthreads();
      fft_stage<N2h, RX::R2, RX::R0 * RX::R1, kRB, kNT, false, true>(a.tw_sub, lds_ld, to_lds);
    }
    if constexpr (RX::S >= 4) {
      __syncthreads();
      fft_stage<N2h, RX::R3, RX::R0 * RX::R1 * RX::R2, kRB, kNT, false, true>(a.tw_sub, lds_ld, to_lds);
    }
  }
  __syncthreads();
  // untangle + four-step twiddle + coalesced store (lanes along k)
  cf* __restrict__ Tm = a.Tm + (size_t)sw * N1 * N2h;
  for (int idx = threadIdx.x; idx < kRB * N2h; idx += kNT) {
    const int k = idx % N2h, r = idx / N2h;
    const int m1 = row0 + r;
    const cf u = lds[k * kRBP + r];
    const cf v = cconj(lds[((k == 0) ? 0 : (N2h - k)) * kRBP + r]);
    const cf sm = cadd(u, v), df = csub(u, v);
    // H = 0.5*sm - 0.5 i e^{-2 pi i k/N2} df
    const cf e = cmul(a.tw_unt[k], df);
    cf H = make_float2(0.5f * (sm.x + e.y), 0.5f * (sm.y - e.x));
    H = cmul(H, twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)k));
    Tm[(size_t)m1 * N2h + k] = H;
  }
  if (threadIdx.x < kRB) {
    const int r = threadIdx.x, m1 = row0 + r;
    const cf u = lds[r];
    const float hn = u.x - u.y;       // Nyquist bin of the row spectrum (real)
    const cf w = twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)N2h);
    a.Tn[(size_t)sw * N1 + m1] = make_float2(hn * w.x, hn * w.y);
  }
}

// ---------------------------------------------------------------------------
// F2: forward columns.  grid = (N2h/CB + 1, station-windows); the last
// block.x handles the Nyquist column alone.
// ---------------------------------------------------------------------------
struct F2Args {
  const cf* Tm; const cf* Tn;
  const SwStat* stats;
  cf* Xm; cf* Xn;            // spectra of this batch: slot = slot0 + sw
  size_t spec_pitch;         // float2 elements between consecutive slots
  int slot0;
  const cf* tw_sub;          // W_{N1}
  // whitening band (bins of the N-grid, float): 0 below b0, ramp b0..b1,
  // 1 in b1..b2, ramp b2..b3, 0 above
  float wb0, wb1, wb2, wb3;
  Geom g;
};

XCX_DEV float whiten_gain(float f, float b0, float b1, float b2, float b3) {
  if (f >= b1 && f <= b2) return 1.f;
  if (f >= b0 && f < b1)
    return 0.5f * (1.f - cosf(3.14159265358979f * (f - b0) / (b1 - b0)));
  if (f > b2 && f <= b3)
    return 0.5f * (1.f + cosf(3.14159265358979f * (f - b2) / (b3 - b2)));
  return 0.f;
}

template <class RX>
__global__ __launch_bounds__(kNT) void k_fwd_cols(F2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N1 = RX::L;
  const Geom g = a.g;
  const int N2h = g.N2h, N2 = g.N2, N = g.N;
  const int sw = blockIdx.y;
  const int ncb = N2h / kCB;
  const bool nyq = (int)blockIdx.x == ncb;
  const int col0 = blockIdx.x * kCB;
  const float scale = a.stats[sw].scale;
  const bool whiten = g.flags & XCX_F_WHITEN;
  const cf* __restrict__ Tm = a.Tm + (size_t)sw * N1 * N2h;
  const cf* __restrict__ Tn = a.Tn + (size_t)sw * N1;
  cf* __restrict__ Xm = a.Xm + (size_t)(a.slot0 + sw) * a.spec_pitch;
  cf* __restrict__ Xn = a.Xn + (size_t)(a.slot0 + sw) * a.spec_pitch;
  const float wb0 = a.wb0, wb1 = a.wb1, wb2 = a.wb2, wb3 = a.wb3;

  auto in = [=](int m1, int c) -> cf {
    if (nyq) return (c == 0) ? Tn[m1] : make_float2(0.f, 0.f);
    return Tm[(size_t)m1 * N2h + col0 + c];
  };
  auto out = [=](int k1, int c, cf v) {
    if (nyq && c != 0) return;
    const int k2 = nyq ? N2h : (col0 + c);
    if (whiten) {
      int k = k1 * N2 + k2;
      if (k > N - k) k = N - k;
      const float gain = whiten_gain((float)k, wb0, wb1, wb2, wb3);
      const float mag = sqrtf(v.x * v.x + v.y * v.y);
      const float f = (mag > 0.f) ? gain / mag : 0.f;
      v = cscale(v, f);
    }
    v = cscale(v, scale);
    if (nyq) Xn[k1] = v; else Xm[(size_t)k1 * N2h + k2] = v;
  };
  subfft<RX, kCB, kCB, kNT, false, false>(lds, a.tw_sub, in, out);
}

// ---------------------------------------------------------------------------
// I1: pair columns.  One workgroup = (pair-window, block of CB columns).
// Workgroups are dealt round-robin over the 8 XCDs, so blockIdx%8 labels the
// XCD; XCD x takes the column blocks cb = x, x+8, ... and walks through all
// pair-windows of the batch for one column block before the next: the
// station slices of that block (N1*CB*8 B each) stay in that XCD's L2 and are
// re-read from there by every pair that shares the station.
// ---------------------------------------------------------------------------
struct PwDesc { int slot_a, slot_b; long long out_off; };   // per pair-window

struct I1Args {
  const cf* Xm; const cf* Xn; size_t spec_pitch;
  const PwDesc* pw;          // batch descriptors
  int B;                     // pair-windows in this batch
  int nstack;                // >1: sum products over nstack consecutive slots
  int stack_stride;          // slot stride between stacked windows
  cf* T;                     // [B][N1][N2h]
  const cf* tw_sub;          // W_{N1}
  const cf* tw_N;
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kNT) void k_pair_cols(I1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N1 = RX::L;
  const Geom g = a.g;
  const int N2h = g.N2h;
  const int ncbt = N2h / kCB + 1;            // + Nyquist block
  const int x = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int cb = (q / a.B) * 8 + x;
  const int pwl = q % a.B;
  if (cb >= ncbt) return;
  const bool nyq = cb == ncbt - 1;
  const int col0 = cb * kCB;
  const PwDesc d = a.pw[pwl];
  const cf* __restrict__ A = (nyq ? a.Xn : a.Xm) + (size_t)d.slot_a * a.spec_pitch;
  const cf* __restrict__ Bp = (nyq ? a.Xn : a.Xm) + (size_t)d.slot_b * a.spec_pitch;
  const int nstack = a.nstack;
  const size_t sstride = (size_t)a.stack_stride * a.spec_pitch;
  cf* __restrict__ T = a.T + (size_t)pwl * N1 * N2h;
  const cf* __restrict__ twN = a.tw_N;

  auto in = [=](int k1, int c) -> cf {
    if (nyq) {
      if (c != 0) return make_float2(0.f, 0.f);
      cf acc = cmulc(A[k1], Bp[k1]);
      for (int w = 1; w < nstack; ++w)
        acc = cadd(acc, cmulc(A[k1 + w * sstride], Bp[k1 + w * sstride]));
      return acc;
    }
    const size_t o = (size_t)k1 * N2h + col0 + c;
    cf acc = cmulc(A[o], Bp[o]);
    for (int w = 1; w < nstack; ++w)
      acc = cadd(acc, cmulc(A[o + w * sstride], Bp[o + w * sstride]));
    return acc;
  };
  auto out = [=](int m1, int c, cf v) {
    if (nyq) {
      if (c != 0) return;
      const cf w = twN_lookup(twN, (unsigned)m1 * (unsigned)N2h);
      const cf r = cmulc(v, w);
      reinterpret_cast<float*>(T + (size_t)m1 * N2h)[1] = r.x;
      return;
    }
    const int k2 = col0 + c;
    const cf w = twN_lookup(twN, (unsigned)m1 * (unsigned)k2);
    const cf r = cmulc(v, w);
    if (k2 == 0) reinterpret_cast<float*>(T + (size_t)m1 * N2h)[0] = r.x;
    else T[(size_t)m1 * N2h + k2] = r;
  };
  subfft<RX, kCB, kCB, kNT, true, false>(lds, a.tw_sub, in, out);
}

// ---------------------------------------------------------------------------
// I2: pair rows.  grid = (N1/RB, B).  Block b owns the output residues
// r in [RB*b, RB*b+RB) (mod N1), i.e. rows m1 = (r - shift) mod N1, so that
// every store instruction writes RB consecutive lags starting on a 64-byte
// boundary of the output row.
// ---------------------------------------------------------------------------
struct I2Args {
  const cf* T;
  const PwDesc* pw;
  float* out;
  const cf* tw_sub;          // W_{N2h}
  const cf* tw_unt;          // W_{N2}[k]
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kNT) void k_pair_rows(I2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N2h = RX::L;
  const Geom g = a.g;
  const int N1 = g.N1, N = g.N, n = g.n;
  const int pwl = blockIdx.y;
  const int res0 = blockIdx.x * kRB;
  const cf* __restrict__ T = a.T + (size_t)pwl * N1 * N2h;
  float* __restrict__ out = a.out + a.pw[pwl].out_off;
  const int shift = g.shift;

  // transposing load: lanes along k (contiguous), LDS [k][row] pitch 17
  for (int idx = threadIdx.x; idx < kRB * N2h; idx += kNT) {
    const int k = idx % N2h, r = idx / N2h;
    int m1 = res0 + r - shift;
    if (m1 < 0) m1 += N1;
    lds[k * kRBP + r] = T[(size_t)m1 * N2h + k];
  }
  __syncthreads();
  const cf* __restrict__ twu = a.tw_unt;
  auto in = [=](int k, int r) -> cf {
    const cf h = lds[k * kRBP + r];
    if (k == 0) return make_float2(h.x + h.y, h.x - h.y);
    const cf hb = cconj(lds[(N2h - k) * kRBP + r]);
    const cf sm = cadd(h, hb), df = csub(h, hb);
    const cf e = cmulc(df, twu[k]);          // e^{+2 pi i k/N2} * df
    return make_float2(sm.x - e.y, sm.y + e.x);   // sm + i e
  };
  const int idx_min = g.idx_min, idx_max = g.idx_max;
  const bool unbias = g.flags & XCX_F_UNBIAS;
  auto out_f = [=](int m, int r, cf u) {
    int m1 = res0 + r - shift;
    if (m1 < 0) m1 += N1;
    int c0 = m1 + N1 * (2 * m);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int j = c0 + h * N1 + (n - 1);
      if (j >= N) j -= N;
      if (j >= idx_min && j < idx_max) {
        float v = h ? u.y : u.x;
        if (unbias) {
          const int lag = j - (n - 1);
          v *= (float)n / (float)(n - (lag < 0 ? -lag : lag));
        }
        out[j - idx_min] = v;
      }
    }
  };
  subfft<RX, kRB, kRBP, kNT, true, true>(lds, a.tw_sub, in, out_f);
}

}  // namespace xcx
