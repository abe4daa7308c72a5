// wave_kernels.h -- barrier-free, one-wave-per-workgroup FFT kernels of the
// pair x window cross-correlation engine (gfx950, wave64).
//
// Same mathematics as cc_kernels.h / tools/proto_fft.py (N = N1*N2,
// m = m1 + N1*m2, k = k1*N2 + k2, Hermitian column layout), but every
// workgroup is ONE 64-lane wave that owns whole sub-FFTs, so no workgroup
// barrier is ever waited on by another wave and the CU hides latency with
// 8-11 independent waves:
//
//   column kernels : one wave = ONE column k2, complex FFT of length N1
//                    (24 points per lane at N1 = 1536), LDS [pos] padded
//   row kernels    : one wave = 16 consecutive rows, real FFT of length N2
//                    via N2h = N2/2 complex points (30 per lane at N2h = 120),
//                    LDS [pos][16]
//
// Layouts (all chosen so that every global access of a wave instruction is a
// run of full 128-byte lines):
//   Tf / X : [k2 in 0..N2h][N1]        column-major, Nyquist = column N2h
//   T      : [res >> 4][k2 in 0..N2h-1][res & 15]   16-row tiles, where
//            res = (m1 + shift) mod N1 is the OUTPUT residue of row m1; the
//            real Nyquist value is packed into the imaginary slot of k2 = 0
//
// The C2R row kernel reads each T element exactly once: butterflies j and
// S-j of its first stage are computed by the same lane, which therefore holds
// both H[k] and H[N2h-k] for the Hermitian untangle.
#pragma once
#include "cc_kernels.h"

namespace xcx {

constexpr int kWRB = 16;     // rows per wave
constexpr int kWNT = 64;     // one wave

// LDS index of column kernels: one 8-byte pad slot every 8 positions makes the
// stride-R stores of the first Stockham stage bank-conflict free
XCX_DEV int cpad(int p) { return p + (p >> 3); }
constexpr int cpad_size(int L) { return L + (L >> 3) + 1; }

// generic NB-sequence sub-FFT through LDS with caller-provided index maps
template <class RX, int NB, bool INV, class Idx, class In, class Out>
XCX_DEV void wave_subfft(cf* __restrict__ lds, const cf* __restrict__ tw, Idx idx, In in, Out out) {
  constexpr int L = RX::L;
  constexpr int R0 = RX::R0, R1 = RX::R1, R2 = RX::R2, R3 = RX::R3;
  auto lds_ld = [=](int p, int c) -> cf { return lds[idx(p, c)]; };
  auto lds_st = [=](int p, int c, cf v) { lds[idx(p, c)] = v; };
  static_assert(RX::S >= 2, "wave kernels need at least two stages");
  fft_stage<L, R0, 1, NB, kWNT, INV, false>(tw, in, lds_st);
  __syncthreads();
  if constexpr (RX::S == 2) {
    fft_stage<L, R1, R0, NB, kWNT, INV, false>(tw, lds_ld, out);
  } else {
    fft_stage<L, R1, R0, NB, kWNT, INV, true>(tw, lds_ld, lds_st);
    __syncthreads();
    if constexpr (RX::S == 3) {
      fft_stage<L, R2, R0 * R1, NB, kWNT, INV, false>(tw, lds_ld, out);
    } else {
      fft_stage<L, R2, R0 * R1, NB, kWNT, INV, true>(tw, lds_ld, lds_st);
      __syncthreads();
      fft_stage<L, R3, R0 * R1 * R2, NB, kWNT, INV, false>(tw, lds_ld, out);
    }
  }
}

// per station-window statistics as partial sums (deterministic order):
// part[sw][p] (p < kStatParts): phase 0 sum(x), phase 1 sum(x'^2)
constexpr int kStatParts = 16;

XCX_DEV float sw_mean(const double* __restrict__ psum, int sw, int n, bool demean) {
  if (!demean) return 0.f;
  double s = 0.0;
#pragma unroll
  for (int p = 0; p < kStatParts; ++p) s += psum[sw * kStatParts + p];
  return (float)(s / (double)n);
}
XCX_DEV float sw_scale(const double* __restrict__ psq, int sw, int N, int flags, float fixed_norm) {
  double nrm = 1.0;
  if (flags & XCX_F_NORMALIZE) {
    if (flags & XCX_F_WHITEN) nrm = (double)fixed_norm;
    else {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < kStatParts; ++p) s += psq[sw * kStatParts + p];
      nrm = sqrt(s);
    }
  }
  return (float)(1.0 / (nrm * sqrt((double)N)));
}

// ---------------------------------------------------------------------------
// W-F1: forward rows.  grid = (N1/16, station-windows), block = 64
// ---------------------------------------------------------------------------
struct WF1Args {
  const float* traces; long stride_s, stride_w; int W, w0, sw0;
  const double* psum;        // [cnt][kStatParts]
  cf* Tf;                    // [cnt][N2h+1][N1]
  const cf* tw_sub;          // W_{N2h}
  const cf* tw_unt;          // W_{N2}[k], k < N2h
  const cf* tw_N;
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kWNT) void k_wfwd_rows(WF1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N2h = RX::L;
  const Geom g = a.g;
  const int sw = blockIdx.y;
  const int swg = a.sw0 + sw;
  const int s = swg / a.W, wl = swg % a.W;
  const int row0 = blockIdx.x * kWRB;
  const float* __restrict__ x = a.traces + (long)s * a.stride_s + (long)(a.w0 + wl) * a.stride_w;
  const int n = g.n, N1 = g.N1;
  const bool onebit = g.flags & XCX_F_ONEBIT;
  const float mean = sw_mean(a.psum, sw, n, g.flags & XCX_F_DEMEAN);
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
  auto idx = [](int p, int r) { return p * kWRB + r; };
  auto to_lds = [=](int p, int r, cf v) { lds[p * kWRB + r] = v; };
  auto lds_ld = [=](int p, int r) -> cf { return lds[p * kWRB + r]; };
  (void)idx;
  fft_stage<N2h, RX::R0, 1, kWRB, kWNT, false, false>(a.tw_sub, in, to_lds);
  __syncthreads();
  fft_stage<N2h, RX::R1, RX::R0, kWRB, kWNT, false, true>(a.tw_sub, lds_ld, to_lds);
  if constexpr (RX::S >= 3) {
    __syncthreads();
    fft_stage<N2h, RX::R2, RX::R0 * RX::R1, kWRB, kWNT, false, true>(a.tw_sub, lds_ld, to_lds);
  }
  __syncthreads();
  cf* __restrict__ Tf = a.Tf + (size_t)sw * (size_t)(N2h + 1) * N1;
  for (int i = threadIdx.x; i < kWRB * N2h; i += kWNT) {
    const int r = i % kWRB, k = i / kWRB;
    const int m1 = row0 + r;
    const cf u = lds[k * kWRB + r];
    const cf v = cconj(lds[((k == 0) ? 0 : (N2h - k)) * kWRB + r]);
    const cf sm = cadd(u, v), df = csub(u, v);
    const cf e = cmul(a.tw_unt[k], df);
    cf H = make_float2(0.5f * (sm.x + e.y), 0.5f * (sm.y - e.x));
    H = cmul(H, twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)k));
    Tf[k * N1 + m1] = H;
  }
  if (threadIdx.x < kWRB) {
    const int r = threadIdx.x, m1 = row0 + r;
    const cf u = lds[r];
    const float hn = u.x - u.y;
    const cf w = twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)N2h);
    Tf[N2h * N1 + m1] = make_float2(hn * w.x, hn * w.y);
  }
}

// ---------------------------------------------------------------------------
// W-F2: forward columns.  grid = (N2h+1, station-windows), block = 64
// ---------------------------------------------------------------------------
struct WF2Args {
  const cf* Tf;
  const double* psq;         // [cnt][kStatParts]
  cf* X; size_t spec_pitch; int slot0;
  const cf* tw_sub;          // W_{N1}
  float wb0, wb1, wb2, wb3, fixed_norm;
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kWNT) void k_wfwd_cols(WF2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N1 = RX::L;
  const Geom g = a.g;
  const int N2h = g.N2h, N2 = g.N2, N = g.N;
  const int sw = blockIdx.y;
  const int k2 = blockIdx.x;
  const float scale = sw_scale(a.psq, sw, N, g.flags, a.fixed_norm);
  const bool whiten = g.flags & XCX_F_WHITEN;
  const cf* __restrict__ src = a.Tf + ((size_t)sw * (N2h + 1) + k2) * N1;
  cf* __restrict__ dst = a.X + (size_t)(a.slot0 + sw) * a.spec_pitch + (size_t)k2 * N1;
  const float wb0 = a.wb0, wb1 = a.wb1, wb2 = a.wb2, wb3 = a.wb3;
  auto in = [=](int m1, int) -> cf { return src[m1]; };
  auto out = [=](int k1, int, cf v) {
    if (whiten) {
      int k = k1 * N2 + k2;
      if (k > N - k) k = N - k;
      const float gain = whiten_gain((float)k, wb0, wb1, wb2, wb3);
      const float mag = sqrtf(v.x * v.x + v.y * v.y);
      v = cscale(v, (mag > 0.f) ? gain / mag : 0.f);
    }
    dst[k1] = cscale(v, scale);
  };
  auto idx = [](int p, int) { return cpad(p); };
  wave_subfft<RX, 1, false>(lds, a.tw_sub, idx, in, out);
}

// ---------------------------------------------------------------------------
// W-I1: pair columns.  One wave = (pair-window, column k2).  XCD x (= block
// index mod 8) takes the columns k2 = x, x+8, ... and walks all pair-windows
// of the batch for one column before the next, so a station's 12 KB column is
// fetched into that XCD's L2 once and re-read there by every pair sharing it.
// ---------------------------------------------------------------------------
struct WI1Args {
  const cf* X; size_t spec_pitch;
  const PwDesc* pw; int B; int nstack;
  cf* T;                     // [B][N1/16][N2h][16]
  const cf* tw_sub; const cf* tw_N;
  Geom g;
};

template <class RX, bool STACK>
__global__ __launch_bounds__(kWNT) void k_wpair_cols(WI1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N1 = RX::L;
  const Geom g = a.g;
  const int N2h = g.N2h;
  const int x = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int k2 = (q / a.B) * 8 + x;
  const int pwl = q % a.B;
  if (k2 > N2h) return;
  const PwDesc d = a.pw[pwl];
  const cf* __restrict__ A = a.X + (size_t)d.slot_a * a.spec_pitch + (size_t)k2 * N1;
  const cf* __restrict__ Bp = a.X + (size_t)d.slot_b * a.spec_pitch + (size_t)k2 * N1;
  const int nstack = a.nstack;
  const unsigned sstride = (unsigned)a.spec_pitch;   // consecutive slots = consecutive windows
  cf* __restrict__ T = a.T + (size_t)pwl * N1 * N2h;
  const cf* __restrict__ twN = a.tw_N;
  const bool nyq = k2 == N2h;
  const unsigned kk = (unsigned)k2;
  const int shift = g.shift;
  auto in = [=](int k1, int) -> cf {
    cf acc = cmulc(A[k1], Bp[k1]);
    if constexpr (STACK) {
      for (int w = 1; w < nstack; ++w)
        acc = cadd(acc, cmulc(A[k1 + w * sstride], Bp[k1 + w * sstride]));
    }
    return acc;
  };
  auto out = [=](int m1, int, cf v) {
    const cf w = twN_lookup(twN, (unsigned)m1 * kk);
    const cf r = cmulc(v, w);
    int res = m1 + shift;
    if (res >= N1) res -= N1;
    float* t = reinterpret_cast<float*>(T + ((res >> 4) * N2h + (nyq ? 0 : k2)) * kWRB + (res & 15));
    if (nyq) t[1] = r.x;
    else if (k2 == 0) t[0] = r.x;
    else *reinterpret_cast<cf*>(t) = r;
  };
  auto idx = [](int p, int) { return cpad(p); };
  wave_subfft<RX, 1, true>(lds, a.tw_sub, idx, in, out);
}

// ---------------------------------------------------------------------------
// W-I2: pair rows.  grid = (N1/16, B), block = 64.  Block b owns the output
// residues [16b, 16b+16): the T tile b, read exactly once.
// ---------------------------------------------------------------------------
struct WI2Args {
  const cf* T; const PwDesc* pw; float* out;
  const cf* tw_sub;          // W_{N2h}
  const cf* tw_unt;          // W_{N2}[k]
  Geom g;
};

template <class RX>
__global__ __launch_bounds__(kWNT) void k_wpair_rows(WI2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int N2h = RX::L;
  constexpr int R = RX::R0;
  constexpr int S = N2h / R;
  constexpr int NITEM = S / 2 + 1;
  constexpr int TOTAL = NITEM * kWRB;
  constexpr int IT = (TOTAL + kWNT - 1) / kWNT;
  const Geom g = a.g;
  const int N1 = g.N1, N = g.N, n = g.n;
  const int pwl = blockIdx.y;
  const int blk = blockIdx.x;
  const cf* __restrict__ Tt = a.T + ((size_t)pwl * (N1 / kWRB) + blk) * (size_t)N2h * kWRB;
  float* __restrict__ out = a.out + a.pw[pwl].out_off;
  const cf* __restrict__ twu = a.tw_unt;
  const int lane = threadIdx.x;

  // ---- stage 0: paired butterflies (j, S-j) with the Hermitian untangle ----
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int w = lane + it * kWNT;
    if ((TOTAL % kWNT) != 0 && w >= TOTAL) continue;
    const int r = w % kWRB;
    const int j = w / kWRB;                       // 0..S/2
    const int jp = (j == 0) ? 0 : (S - j);
    const bool single = jp == j;
    cf av[R], bv[R];
#pragma unroll
    for (int t = 0; t < R; ++t) av[t] = Tt[(j + S * t) * kWRB + r];
    if (!single) {
#pragma unroll
      for (int t = 0; t < R; ++t) bv[t] = Tt[(jp + S * t) * kWRB + r];
    } else if (j == 0) {
#pragma unroll
      for (int t = 0; t < R; ++t) bv[t] = av[(t + 1) % R];   // mirror of 15t' is 15(R-t')
    } else {
#pragma unroll
      for (int t = 0; t < R; ++t) bv[t] = av[t];             // j = S/2 mirrors inside itself
    }
    // e^{+2 pi i (j + S t)/N2} = conj(twu[j]) * e^{+2 pi i t/(2R)}
    const cf ej = cconj(twu[j]);
    const cf ejp = cconj(twu[jp]);
    cf uj[R], ujp[R];
    static_for<R>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      {
        const cf h = av[t], hb = cconj(bv[R - 1 - t]);
        const cf sm = cadd(h, hb), df = csub(h, hb);
        const cf e = cmul(mul_wconst<t, 2 * R, true>(ej), df);
        uj[t] = make_float2(sm.x - e.y, sm.y + e.x);
      }
      {
        const cf h = bv[t], hb = cconj(av[R - 1 - t]);
        const cf sm = cadd(h, hb), df = csub(h, hb);
        const cf e = cmul(mul_wconst<t, 2 * R, true>(ejp), df);
        ujp[t] = make_float2(sm.x - e.y, sm.y + e.x);
      }
    });
    if (j == 0) uj[0] = make_float2(av[0].x + av[0].y, av[0].x - av[0].y);   // DC / Nyquist
    Bfly<R, true>::run(uj);
#pragma unroll
    for (int t = 0; t < R; ++t) lds[(j * R + t) * kWRB + r] = uj[t];
    if (!single) {
      Bfly<R, true>::run(ujp);
#pragma unroll
      for (int t = 0; t < R; ++t) lds[(jp * R + t) * kWRB + r] = ujp[t];
    }
  }
  __syncthreads();
  // ---- remaining stages + output ---------------------------------------
  const int idx_min = g.idx_min, idx_max = g.idx_max;
  const bool unbias = g.flags & XCX_F_UNBIAS;
  const int shift = g.shift;
  auto out_f = [=](int m, int r, cf u) {
    int m1 = blk * kWRB + r - shift;
    if (m1 < 0) m1 += N1;
    const int c0 = m1 + N1 * (2 * m);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int jj = c0 + h * N1 + (n - 1);
      if (jj >= N) jj -= N;
      if (jj >= idx_min && jj < idx_max) {
        float v = h ? u.y : u.x;
        if (unbias) {
          const int lag = jj - (n - 1);
          v *= (float)n / (float)(n - (lag < 0 ? -lag : lag));
        }
        out[jj - idx_min] = v;
      }
    }
  };
  auto lds_ld = [=](int p, int r) -> cf { return lds[p * kWRB + r]; };
  auto lds_st = [=](int p, int r, cf v) { lds[p * kWRB + r] = v; };
  if constexpr (RX::S == 2) {
    fft_stage<N2h, RX::R1, RX::R0, kWRB, kWNT, true, false>(a.tw_sub, lds_ld, out_f);
  } else {
    fft_stage<N2h, RX::R1, RX::R0, kWRB, kWNT, true, true>(a.tw_sub, lds_ld, lds_st);
    __syncthreads();
    fft_stage<N2h, RX::R2, RX::R0 * RX::R1, kWRB, kWNT, true, false>(a.tw_sub, lds_ld, out_f);
  }
}

}  // namespace xcx
