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
//   T      : [res / 128][k2 in 0..N2h][res % 128]   where res = (m1 + shift) mod N1 is
//            the OUTPUT residue of row m1 (slots k2 = 0 and N2h are real: only
//            .x is consumed); a column wave stores 1 KB runs, a row wave (16 rows)
//            reads 128-byte runs
//
// The C2R row kernel reads each T element exactly once: butterflies j and
// S-j of its first stage are computed by the same lane, which therefore holds
// both H[k] and H[N2h-k] for the Hermitian untangle.
#pragma once
#include <type_traits>
#include "cc_kernels.h"

namespace xcx {

// ---------------------------------------------------------------------------
// Diagnostic build only (-DXCX_STAMPS, tools/stamps.sh): s_memtime stamps at the phase
// boundaries of the pair bodies, to see where a wave's lifetime goes.  In the product build
// every macro below expands to nothing.
// ---------------------------------------------------------------------------
#ifdef XCX_STAMPS
constexpr int kNStamp = 12;
typedef unsigned long long StampArr[kNStamp];
// the VGPR operand pins the stamp behind the instruction that produced `pin`
XCX_DEV unsigned long long stamp_wait(float pin) {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(pin) : "memory");
  return t;
}
XCX_DEV unsigned long long stamp_nowait(float pin) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(pin) : "memory");
  return t;
}
#define XCX_ST_PARAM , StampArr& xst
#define XCX_ST_ARG , xst
#define XCX_STAMP(i, pin) xst[i] = stamp_wait(pin)
#define XCX_STAMP_NW(i, pin) xst[i] = stamp_nowait(pin)
#else
#define XCX_ST_PARAM
#define XCX_ST_ARG
#define XCX_STAMP(i, pin)
#define XCX_STAMP_NW(i, pin)
#endif

// minimum waves per SIMD the register allocator must leave room for
#ifndef XCX_COL_WAVES
#define XCX_COL_WAVES 1
#endif
#ifndef XCX_ROW_WAVES
#define XCX_ROW_WAVES 1
#endif
constexpr int kWRB = 16;     // rows per wave
constexpr int kTR = 64;      // rows per row workgroup (cooperative CCF store): four waves x 16 rows
// T layout of one pair-window: [res / TC][k2 in 0..N2h][res % TC], res = output residue of row m1,
// TC = 128 rows (64 when N1 is not a multiple of 128).  A column wave (one k2) stores 16 bytes per lane
// = two adjacent rows: lanes 0..31 cover one 64-row half of a chunk, lanes 32..63 the next -- one
// contiguous 1 KB run per store instruction when they share a chunk, two 512-byte runs otherwise.
// A row wave reads, per k2, the 128 bytes of its 16 rows (1 KB apart from one k2 to the next).
constexpr int t_chunk(int N1) { return (N1 % 128 == 0) ? 128 : 64; }
constexpr int kWNT = 64;     // one wave
constexpr int kWPB = 4;      // independent waves packed into one 256-thread workgroup

// per-wave LDS footprints (floats), 16-byte aligned
constexpr int lds_col_floats(int L) { return (L + (L >> 3) + 1 + 3) & ~3; }
constexpr int lds_row_exch_floats(int L) { return ((L + (L >> 3) + 1) * 16 + 3) & ~3; }
constexpr int lds_row_floats(int L) { return lds_row_exch_floats(L); }
constexpr int lds_frow_floats(int L) { return L * 16 * 2; }
// wave-uniform BY CONSTRUCTION: readfirstlane tells the compiler so, and everything derived from it
// (work item, descriptor, column / tile base pointers) lives in SGPRs -- descriptor loads become
// s_load (scalar cache) and the data loads take the saddr + 32-bit voffset form instead of
// per-lane 64-bit addresses
XCX_DEV int wave_in_group() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// Pair-window descriptor `i` (wave-uniform index) through the CONSTANT address space: scalar loads.  The descriptors
// are written by the host before the launch, but the kernels store through other pointers, so the compiler cannot prove
// them unclobbered and loads them with a VECTOR instruction (all lanes one address) followed by v_readfirstlane -- a
// round trip through the CU's congested vector-memory pipeline that the wave sat out before it could issue its first
// spectrum load (generated code until round 5: global_load_dwordx2, s_waitcnt vmcnt, v_readfirstlane, then the loads).
XCX_DEV PwDesc pw_desc(const PwDesc* __restrict__ base, int i) {
  typedef const __attribute__((address_space(4))) int cint;
  cint* p = (cint*)(const int*)(base + __builtin_amdgcn_readfirstlane(i));
  PwDesc d;
  d.slot_a = p[0]; d.slot_b = p[1];
  d.out_off = (long long)(((unsigned long long)(unsigned)p[3] << 32) | (unsigned)p[2]);
  return d;
}

// element offset of residue `res` inside a pair-window's T for k2 = 0 (add k2 * TC)
XCX_DEV unsigned t_off(int res, int TC, int N2h) {
  return (unsigned)((res / TC) * ((N2h + 1) * TC) + (res % TC));
}
// first element (k2 = 0, row 0) of the 16 rows `blk` of one pair-window; k2 advances by TC elements
XCX_DEV const cf* t_rows16(const cf* __restrict__ Tpw, int blk, int N2h, int TC) {
  return Tpw + t_off(blk * kWRB, TC, N2h);
}
// ---------------------------------------------------------------------------
// T storage of the dual pair kernel (template parameter TS)
//   TS = 0 : float2, 8 bytes per element
//   TS = 1 : q24 -- (re, im) as two 24-bit two's-complement FIXED-POINT numbers, 6 bytes per element, for plans whose
//            spectra have a known modulus (whitened plans: |X[k]| = gain(k) * scale exactly, so |T| <= scale^2 *
//            sum_k1 gain^2 =: Bmax is a bound, not a statistic).  A column wave rounds T * tq to the nearest integer
//            (tq = 0.98 * 2^23 / Bmax; v_cvt_rpi_i32_f32), keeps the low three bytes of each number and stores two
//            adjacent rows as 12 bytes per lane; a row wave rebuilds i / tq.  Quantisation step Bmax / (0.98 * 2^23):
//            ~1e-6 of rms(T) for the 0.1-10 Hz band at 50 Hz (Bmax / rms(T) = sqrt(in-band bins of a column));
//            measured against float2 storage in profiles/r05_t24_error_table.txt.  The arithmetic on both sides
//            stays float32.
//   Same element indexing in both: [res / TC][k2][res % TC]; a pair-window's slot keeps its 8-byte pitch.
// ---------------------------------------------------------------------------
constexpr float kQ24Range = 8388608.f;           // 2^23: |T * tq| stays below 0.98 of it
// round to nearest (floor(x + 0.5)); v_cvt_i32_f32 would truncate towards zero, a bias of half a step
XCX_DEV float q24_round(float x) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return __int_as_float(r);
}
template <int TS> XCX_DEV const char* t_rows16_b(const cf* __restrict__ Tpw, int blk, int N2h, int TC) {
  return reinterpret_cast<const char*>(Tpw) + (size_t)t_off(blk * kWRB, TC, N2h) * (TS == 1 ? 6u : 8u);
}
// 4 numbers whose low three bytes are kept -> 3 dwords
XCX_DEV void q24_pack(unsigned d0, unsigned d1, unsigned d2, unsigned d3, unsigned (&o)[3]) {
  o[0] = __builtin_amdgcn_perm(d1, d0, 0x04020100u);
  o[1] = __builtin_amdgcn_perm(d2, d1, 0x05040201u);
  o[2] = __builtin_amdgcn_perm(d3, d2, 0x06050402u);
}
// staging buffer of the workgroup-cooperative CCF store: (N2h + 1) lag blocks x 64 rows of float
constexpr int lds_stage_floats(int N2h) { return (N2h + 1) * kTR; }

// LDS index of column kernels: one 8-byte pad slot every 8 positions makes the
// stride-R stores of the first Stockham stage bank-conflict free
XCX_DEV int cpad(int p) { return p + (p >> 3); }
constexpr int cpad_size(int L) { return L + (L >> 3) + 1; }   // elements (float or float2)
// row kernels: [pos][16] with one 16-element pad row every 8 positions
XCX_DEV int rpad(int p, int r) { return (p + (p >> 3)) * kWRB + r; }
constexpr int rpad_size(int L) { return (L + (L >> 3) + 1) * kWRB; }

// One-column complex FFT of length RX::L by one wave.  All twiddle base
// factors are loaded before the first butterfly; after the input loads the
// wave touches only registers and LDS.  `epi(it, j0, v)` receives each last
// stage butterfly (outputs at positions j0 + t*NSlast, t = 0..Rlast-1).
template <class RX> struct ColGeom {
  static constexpr int L = RX::L;
  static constexpr int S = RX::S;
  static constexpr int RL = (S == 2) ? RX::R1 : ((S == 3) ? RX::R2 : RX::R3);   // last radix
  static constexpr int NSL = L / RL;                                            // its NS
  using Last = StageGeom<L, RL, NSL, 1, kWNT>;
};

// `ltid`: lane -> butterfly slot of the LAST stage (default: the lane itself).  Any permutation
// within 0..63 is valid when that stage has a multiple of 64 butterflies: its inputs come out of
// LDS by position, so only the stage's own indexing has to follow.
template <class RX> struct ColRot {
  using CG = ColGeom<RX>;
  static constexpr bool ok = (CG::Last::NBF % kWNT == 0) && (CG::NSL % kTR == 0);
};
// `in(k, kb)`: input k; kb = the smallest k of this wave instruction (lane 0's), a compile-time constant per call.
// `in2(k, a, b, kb)` (may be a null functor, In2 = int): two adjacent inputs k, k + 1 (k even) at once -- 16-byte loads.
// Used for the first 2 x 64 butterflies of stage 0 when the column has at least that many
// (lane l takes butterflies 2l and 2l+1); see pair_src_index.
struct NoIn2 {};
template <class RX, bool INV, class In, class Epi, class In2 = NoIn2>
XCX_DEV void wave_col_fft(cf* __restrict__ lds, const cf* __restrict__ tw, In in, Epi epi, int ltid XCX_ST_PARAM, In2 in2 = In2()) {
  constexpr int L = RX::L;
  constexpr int R0 = RX::R0, R1 = RX::R1, R2 = RX::R2;
  constexpr bool WIDE0 = !std::is_same<In2, NoIn2>::value && (StageGeom<L, R0, 1, 1, kWNT>::NBF % kWNT == 0) &&
                         (StageGeom<L, R0, 1, 1, kWNT>::IT >= 2);
  static_assert(RX::S == 2 || RX::S == 3, "column configs have two or three stages");
  float* __restrict__ ldsf = reinterpret_cast<float*>(lds);
  using G0 = StageGeom<L, R0, 1, 1, kWNT>;
  using G1 = StageGeom<L, R1, R0, 1, kWNT>;
  cf b1[G1::IT];
  stage_tw_base<L, R1, R0, 1, kWNT>(tw, b1, RX::S == 2 ? ltid : tid_of<kWNT>());
  cf v0[G0::IT][R0];
#ifdef XCX_STAMPS
#pragma unroll
  for (int it = 0; it < G0::IT; ++it) {
    const int j = tid_of<kWNT>() + it * kWNT;
    if (!G0::GUARD || j < G0::NBF) {
#pragma unroll
      for (int t = 0; t < R0; ++t) v0[it][t] = in(j + t * (L / R0), it * kWNT + t * (L / R0));
    }
  }
  XCX_STAMP(1, v0[G0::IT - 1][R0 - 1].x);          // inputs have arrived (product formed)
#pragma unroll
  for (int it = 0; it < G0::IT; ++it) {
    const int j = tid_of<kWNT>() + it * kWNT;
    if (!G0::GUARD || j < G0::NBF) Bfly<R0, INV>::run(v0[it]);
  }
  XCX_STAMP(2, v0[G0::IT - 1][R0 - 1].x);          // stage 0 computed
#else
  if constexpr (WIDE0) {
#pragma unroll
    for (int t = 0; t < R0; ++t) in2(2 * tid_of<kWNT>() + t * (L / R0), v0[0][t], v0[1][t], t * (L / R0));
#pragma unroll
    for (int it = 2; it < G0::IT; ++it) {
      const int j = tid_of<kWNT>() + it * kWNT;
#pragma unroll
      for (int t = 0; t < R0; ++t) v0[it][t] = in(j + t * (L / R0), it * kWNT + t * (L / R0));
    }
#pragma unroll
    for (int it = 0; it < G0::IT; ++it) Bfly<R0, INV>::run(v0[it]);
  } else {
#pragma unroll
    for (int it = 0; it < G0::IT; ++it) {
      const int j = tid_of<kWNT>() + it * kWNT;
      if (!G0::GUARD || j < G0::NBF) {
#pragma unroll
        for (int t = 0; t < R0; ++t) v0[it][t] = in(j + t * (L / R0), it * kWNT + t * (L / R0));
        Bfly<R0, INV>::run(v0[it]);
      }
    }
  }
#endif
  cf v1[G1::IT][R1];
  if constexpr (RX::S == 2) {
    stage_exchange_pad8<L, R0, 1, R1, kWNT, WIDE0>(ldsf, v0, v1, ltid);
    stage_compute<L, R1, R0, 1, kWNT, INV>(v1, b1);
#pragma unroll
    for (int it = 0; it < G1::IT; ++it) {
      const int j = ltid + it * kWNT;
      if (!G1::GUARD || j < G1::NBF) {
        const int k = j % R0;
        epi(it, (j - k) * R1 + k, 0, v1[it]);
      }
    }
  } else {
    using G2 = StageGeom<L, R2, R0 * R1, 1, kWNT>;
    cf b2[G2::IT];
    stage_tw_base<L, R2, R0 * R1, 1, kWNT>(tw, b2, ltid);
    stage_exchange_pad8<L, R0, 1, R1, kWNT, WIDE0>(ldsf, v0, v1);
    XCX_STAMP(3, v1[G1::IT - 1][R1 - 1].y);
    stage_compute<L, R1, R0, 1, kWNT, INV>(v1, b1);
    XCX_STAMP(4, v1[G1::IT - 1][R1 - 1].y);
    cf v2[G2::IT][R2];
    stage_exchange_pad8<L, R1, R0, R2, kWNT>(ldsf, v1, v2, ltid);
    XCX_STAMP(5, v2[G2::IT - 1][R2 - 1].y);
    stage_compute<L, R2, R0 * R1, 1, kWNT, INV>(v2, b2);
    XCX_STAMP(6, v2[G2::IT - 1][R2 - 1].y);
#pragma unroll
    for (int it = 0; it < G2::IT; ++it) {
      const int j = ltid + it * kWNT;
      if (!G2::GUARD || j < G2::NBF) {
        const int k = j % (R0 * R1);
        epi(it, (j - k) * R2 + k, 0, v2[it]);
      }
    }
  }
}

// Band-limited spectra (whitened plans, Geom::band_hi): column k2 is exactly zero for k1 in [za, zb], where both
// k = k1 N2 + k2 and N - k lie above the band.  zb < za: no such stretch (band_hi = N).
XCX_DEV void band_zero_stretch(const Geom& g, int k2, int& za, int& zb) {
  za = (g.band_hi >= k2) ? (g.band_hi - k2) / g.N2 + 1 : 0;
  zb = (g.N - g.band_hi - k2 + g.N2 - 1) / g.N2 - 1;
}

// per station-window statistics as partial sums (deterministic order):
// part[sw][p] (p < kStatParts): phase 0 sum(x), phase 1 sum(x'^2)
constexpr int kStatParts = 16;

XCX_DEV double sw_mean_d(const double* __restrict__ psum, int sw, int n, bool demean) {
  if (!demean) return 0.0;
  double s = 0.0;
#pragma unroll
  for (int p = 0; p < kStatParts; ++p) s += psum[sw * kStatParts + p];
  return s / (double)n;
}
XCX_DEV float sw_mean(const double* __restrict__ psum, int sw, int n, bool demean) {
  return (float)sw_mean_d(psum, sw, n, demean);
}
// one-bit after demeaning: the sign is decided in float64 (x - mean rounded to float32 can
// flip the sign of a sample that sits within an ulp of the window mean)
XCX_DEV float onebit_of(float x, double mean_d) {
  const double d = (double)x - mean_d;
  return (d > 0.0) ? 1.f : ((d < 0.0) ? -1.f : 0.f);
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

// Cache policy of the forward transforms' global accesses (template parameter NT of k_wfwd_rows / k_wfwd_cols).  A large
// launch streams everything it touches exactly once (traces in, Tf out and back in, spectra out: 0.76 GB of Tf per
// 512-window launch against 256 MiB of Infinity Cache), so its accesses are non-temporal and leave the cache to the
// pair pipelines' T ring (headline +0.8 %, profiles/r05_session2_notes.txt); small launches (a few rows of a time
// shift or an any-length transform), whose Tf does fit, keep the default policy.  forward() decides per launch.
constexpr size_t kFwdNtBytes = (size_t)128 << 20;           // Tf bytes of a launch from which it streams
template <bool NT> XCX_DEV float fwd_ld(const float* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT> XCX_DEV cf fwd_ld(const cf* p) {
  if constexpr (NT) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
    return make_float2(v.x, v.y);
  } else {
    return *p;
  }
}
// ... addressed as (uniform base in scalar registers) + (32-bit per-lane byte offset): written as pointer + index the
// forward row kernel built a 64-bit vector address for each of its 64 sample loads and 31 Tf stores (v_lshl_add_u64)
typedef const __attribute__((address_space(1))) char* fwd_gchar;
typedef __attribute__((address_space(1))) char* fwd_gchar_w;
template <bool NT> XCX_DEV float fwd_ld_off(fwd_gchar base, unsigned byte_off) {
  typedef const __attribute__((address_space(1))) float* gf;
  if constexpr (NT) return __builtin_nontemporal_load((gf)(base + (size_t)byte_off));
  else return *(gf)(base + (size_t)byte_off);
}
template <bool NT> XCX_DEV void fwd_st_off(fwd_gchar_w base, unsigned byte_off, cf v) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) f32x2* gf2;
  if constexpr (NT) __builtin_nontemporal_store(f32x2{v.x, v.y}, (gf2)(base + (size_t)byte_off));
  else *(gf2)(base + (size_t)byte_off) = f32x2{v.x, v.y};
}
template <bool NT> XCX_DEV void fwd_st(cf* p, cf v) {
  if constexpr (NT) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(f32x2{v.x, v.y}, reinterpret_cast<f32x2*>(p));
  } else {
    *p = v;
  }
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
  int nt;                    // 1: non-temporal accesses (large launches, see fwd_ld)
};

// waves per workgroup of the forward row kernel: its full-complex LDS slice (15 KB per wave at
// N2h = 120) limits residency, and smaller workgroups pack the 160 KB better
#ifndef XCX_F1_WPB
#define XCX_F1_WPB 2
#endif
constexpr int kF1WPB = XCX_F1_WPB;
// ONEBIT is a template parameter: as a run-time flag the sign() branch sat between every
// pair of loads and its s_waitcnt vmcnt(0) serialised the whole gather (30 memory round
// trips per lane; the kernel ran at 1 TB/s)
template <class RX, bool ONEBIT, bool NT = false>
__global__ __launch_bounds__(kWNT * kF1WPB) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_wfwd_rows(WF1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  constexpr int N2h = RX::L;
  const Geom g = a.g;
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave_in_group() * lds_frow_floats(N2h));
  const int sw = blockIdx.y;
  const int swg = a.sw0 + sw;
  const int s = swg / a.W, wl = swg % a.W;
  const int row0 = (blockIdx.x * kF1WPB + wave_in_group()) * kWRB;
  const float* __restrict__ x = a.traces + (long)s * a.stride_s + (long)(a.w0 + wl) * a.stride_w;
  const int n = g.n, N1 = g.N1;
  const double mean_d = sw_mean_d(a.psum, sw, n, g.flags & XCX_F_DEMEAN);
  const float mean = (float)mean_d;
  // branch-free gather of the zero-padded window: indices beyond the window
  // are clamped (the load hits the cached last sample) and the value zeroed,
  // so all loads of a lane are in flight together
  fwd_gchar xb = (fwd_gchar)x;
  asm volatile("" : "+s"(xb));                                 // the window's first sample: wave-uniform, kept in scalar registers
  auto in = [=](int m, int r) -> cf {
    const int i0 = row0 + r + N1 * (2 * m);
    const int i1 = i0 + N1;
    const float x0 = fwd_ld_off<NT>(xb, 4u * (unsigned)(i0 < n ? i0 : n - 1)), x1 = fwd_ld_off<NT>(xb, 4u * (unsigned)(i1 < n ? i1 : n - 1));
    float v0, v1;
    if constexpr (ONEBIT) {
      v0 = onebit_of(x0, mean_d);
      v1 = onebit_of(x1, mean_d);
    } else {
      v0 = x0 - mean;
      v1 = x1 - mean;
    }
    return make_float2(i0 < n ? v0 : 0.f, i1 < n ? v1 : 0.f);
  };
  auto to_lds = [=](int p, int r, cf v) { lds[p * kWRB + r] = v; };
  auto lds_ld = [=](int p, int r) -> cf { return lds[p * kWRB + r]; };
  static_assert(RX::S == 2, "row configs have two stages");
  cf b0[StageGeom<N2h, RX::R0, 1, kWRB, kWNT>::IT];
  cf b1[StageGeom<N2h, RX::R1, RX::R0, kWRB, kWNT>::IT];
  stage_tw_base<N2h, RX::R1, RX::R0, kWRB, kWNT>(a.tw_sub, b1);
  fft_stage_c<N2h, RX::R0, 1, kWRB, kWNT, false, false, false>(b0, in, to_lds);
  group_sync<kWNT>();
  fft_stage_c<N2h, RX::R1, RX::R0, kWRB, kWNT, false, true, false>(b1, lds_ld, to_lds);
  group_sync<kWNT>();
  cf* __restrict__ Tf = a.Tf + (size_t)sw * (size_t)(N2h + 1) * N1;
  // Hermitian untangle + four-step factor, fully unrolled: lane (r, kq) = (tid % 16, tid / 16) takes the bins
  // k = kq + 4 it of its row.  Both twiddles advance by a constant factor from one bin to the next
  // (w_N2^4 and w_N^(4 m1)), so they are formed by multiplication and re-seeded from the tables every 8 bins;
  // all table loads are issued up front.  (As a rolled loop with three table lookups per bin the epilogue
  // was 30 serialised load round trips per wave.)
  {
    constexpr int EIT = kWRB * N2h / kWNT;                 // bins per lane
    constexpr int NSEED = (EIT + 7) / 8;
    static_assert((kWRB * N2h) % kWNT == 0, "row length must be a multiple of 4");
    const int r = tid_of<kWNT>() % kWRB, kq = tid_of<kWNT>() / kWRB;
    const int m1 = row0 + r;
    cf su[NSEED], sn[NSEED];
#pragma unroll
    for (int j = 0; j < NSEED; ++j) {
      const int ks = kq + 32 * j < N2h ? kq + 32 * j : N2h - 1;      // clamped: unused past the last bin
      su[j] = *uoff(a.tw_unt, (unsigned)ks);
      sn[j] = twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)ks);
    }
    const cf du = *uoff(a.tw_unt, 4u);                     // w_N2^4
    const cf dn = twN_lookup(a.tw_N, 4u * (unsigned)m1);   // w_N^(4 m1)
    cf tu = su[0], tn = sn[0];
    fwd_gchar_w Tfb = (fwd_gchar_w)Tf;
    asm volatile("" : "+s"(Tfb));
    static_for<EIT>([&](auto ic) {
      constexpr int it = decltype(ic)::value;
      const int k = kq + 4 * it;
      if constexpr (it % 8 == 0) { tu = su[it / 8]; tn = sn[it / 8]; }
      else { tu = cmul(tu, du); tn = cmul(tn, dn); }
      const cf u = lds[k * kWRB + r];
      const cf v = cconj(lds[((k == 0) ? 0 : (N2h - k)) * kWRB + r]);
      const cf sm = cadd(u, v), df = csub(u, v);
      const cf e = cmul(tu, df);
      cf H = make_float2(0.5f * (sm.x + e.y), 0.5f * (sm.y - e.x));
      H = cmul(H, tn);
      fwd_st_off<NT>(Tfb, 8u * (unsigned)(k * N1 + m1), H);
    });
  }
  if (tid_of<kWNT>() < kWRB) {
    const int r = tid_of<kWNT>(), m1 = row0 + r;
    const cf u = lds[r];
    const float hn = u.x - u.y;
    const cf w = twN_lookup(a.tw_N, (unsigned)m1 * (unsigned)N2h);
    fwd_st<NT>(Tf + (N2h * N1 + m1), make_float2(hn * w.x, hn * w.y));
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
  // station-major slots with room for more windows per station (multi-day stacks): station-window
  // q = sw_first + blockIdx.y goes to slot slot0 + (q / slot_w) * slot_ld + q % slot_w; slot_ld = 0:
  // dense, slot0 + blockIdx.y
  int sw_first, slot_w, slot_ld;
  int nt;                    // 1: non-temporal accesses (large launches, see fwd_ld)
};

#ifndef XCX_FWDC_WAVES
#define XCX_FWDC_WAVES 3      // the kernel needs 121 VGPRs: at 4 its launches are 8 % shorter (218 -> 201 us), the step is not (r05_session2_notes.txt)
#endif
template <class RX, bool NT = false>
__global__ __launch_bounds__(kWNT * kWPB) __attribute__((amdgpu_waves_per_eu(RX::L > 2048 ? 1 : XCX_FWDC_WAVES, RX::L > 2048 ? 2 : XCX_FWDC_WAVES)))
void k_wfwd_cols(WF2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  constexpr int N1 = RX::L;
  const Geom g = a.g;
  const int N2h = g.N2h, N2 = g.N2, N = g.N;
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave_in_group() * lds_col_floats(N1));
  const int sw = blockIdx.y;
  const int k2 = blockIdx.x * kWPB + wave_in_group();
  if (k2 > N2h) return;
  const float scale = sw_scale(a.psq, sw, N, g.flags, a.fixed_norm);
  const bool whiten = g.flags & XCX_F_WHITEN;
  const cf* __restrict__ src = a.Tf + ((size_t)sw * (N2h + 1) + k2) * N1;
  int slot = a.slot0 + sw;
  if (a.slot_ld) { const int q = a.sw_first + sw; slot = a.slot0 + (q / a.slot_w) * a.slot_ld + q % a.slot_w; }
  cf* __restrict__ dst = a.X + (size_t)slot * a.spec_pitch + (size_t)k2 * N1;
  const float wb0 = a.wb0, wb1 = a.wb1, wb2 = a.wb2, wb3 = a.wb3;
  auto in = [=](int m1, int) -> cf { return fwd_ld<NT>(src + m1); };
  using CG = ColGeom<RX>;
  // the out-of-band stretch of a band-limited plan is never stored: the buffer holds zeros there
  // (zeroed once per allocation by forward()), which is what the whitening gain would produce
  int za, zb;
  band_zero_stretch(g, k2, za, zb);
  const unsigned zlen = (whiten && zb >= za) ? (unsigned)(zb - za + 1) : 0u;
  fwd_gchar_w dstb = (fwd_gchar_w)dst;
  asm volatile("" : "+s"(dstb));                               // uniform base of the spectrum column, 32-bit per-lane offsets
  auto epi = [=](int, int j0, int, cf (&v)[CG::RL]) {
#pragma unroll
    for (int t = 0; t < CG::RL; ++t) {
      const int k1 = j0 + t * CG::NSL;
      if ((unsigned)(k1 - za) < zlen) continue;
      cf o = v[t];
      if (whiten) {
        int k = k1 * N2 + k2;
        if (k > N - k) k = N - k;
        const float gain = whiten_gain((float)k, wb0, wb1, wb2, wb3);
        const float mag = sqrtf(o.x * o.x + o.y * o.y);
        o = cscale(o, (mag > 0.f) ? gain / mag : 0.f);
      }
      fwd_st_off<NT>(dstb, 8u * (unsigned)k1, cscale(o, scale));
    }
  };
#ifdef XCX_STAMPS
  StampArr xst;
#endif
  wave_col_fft<RX, false>(lds, a.tw_sub, in, epi, tid_of<kWNT>() XCX_ST_ARG);
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
  cf* T;                     // [B][N1/16][N2h+1][16]
  const cf* tw_sub; const cf* tw_N;
  Geom g;
};

// column body shared by the pair kernels (separate, dual, flow): pair-window
// descriptor d, column k2, T slot base `T`
// MODE 0: C = X_a conj X_b;  1: sum of that over TWO consecutive slots (nstack == 2);
// 2: C is read from a precomputed cross-spectrum (slot_a of the Gram buffer; nstack < 0)
template <class RX> struct ColPairStore {
  using CG = ColGeom<RX>;
  static constexpr bool ok = ColRot<RX>::ok && (CG::RL % 2 == 0) && (CG::Last::IT == 1) && (CG::NSL == kTR);
};
// ZB0 > 0 (band-limited plans, MODE 0): the 64-point blocks ZB0 .. N1/64 - 1 - ZB0 of EVERY column lie inside the
// zero stretch (zskip_blocks() below; chosen by the plan from its band), so a wave instruction whose points all fall
// into those blocks is not issued at all -- its lane offset is a compile-time constant after unrolling, the test
// folds away and the butterflies that follow see literal zeros.  With ZB0 = 0 such loads go to the zero chunk (L1 hits
// that still cost a register pair and a slot in the memory pipeline each): 14 of the 24 blocks of a 0.1-10 Hz column
// at 50 Hz, +3.7 % on the headline (profiles/r05_session2_notes.txt).
template <class RX, int MODE, int TS = 0, int ZB0 = 0, int N2HC = 0>      // N2HC: the row length when the caller knows it at compile time
XCX_DEV void pair_col_body(cf* __restrict__ lds, const cf* __restrict__ X, size_t spec_pitch,
                           const PwDesc d, int nstack, int k2, cf* __restrict__ T,
                           const cf* __restrict__ tw_sub, const cf* __restrict__ twN, const Geom& g XCX_ST_PARAM,
                           float tq = 0.f) {
  constexpr int N1 = RX::L;
  const int N2h = N2HC ? N2HC : g.N2h;            // compile time in the dual kernels: the T tile offsets fold into shifts and adds
  const cf* __restrict__ A = X + (size_t)d.slot_a * spec_pitch + (size_t)k2 * N1;
  const cf* __restrict__ Bp = X + (size_t)d.slot_b * spec_pitch + (size_t)k2 * N1;
  const unsigned sstride = (unsigned)spec_pitch;   // consecutive slots = consecutive windows
  const unsigned kk = (unsigned)k2;
  const int shift = g.shift;
  using CG = ColGeom<RX>;
  // four-step factors w_N^{-(j0 + t*NSL) k2}: base per (lane, iteration) and
  // the wave-uniform step g = w_N^{-NSL k2}; powers by multiplication
  // last-stage slots rotated so that lane l holds the residues res = l (mod 64): every T store
  // instruction then writes ONE aligned 512-byte run of a 64-row tile (four whole lines) instead
  // of the tail of one tile and the head of the next
  // With NSL = 64 the stores go out as lane PAIRS (below): lanes 0..31 are given the even residues
  // 0, 2, .., 62 and lanes 32..63 the odd ones, so that one v_permlane32_swap per register pair hands
  // lane L rows (2L, 2L+1) of an even tile and lane 32+L the same rows of the next tile.
  constexpr bool kPairStore = ColPairStore<RX>::ok;
  static_assert(TS == 0 || kPairStore, "q24 storage is written by the lane-pair stores only");
  const int rlane = kPairStore ? (2 * (tid_of<kWNT>() & 31) + (tid_of<kWNT>() >> 5)) : tid_of<kWNT>();   // residue mod 64
  const int ltid = ColRot<RX>::ok ? ((rlane - shift) & (kWNT - 1)) : tid_of<kWNT>();
  cf f0[CG::Last::IT];
#pragma unroll
  for (int it = 0; it < CG::Last::IT; ++it) {
    const int b = ltid + it * kWNT;
    const int j = (CG::Last::GUARD && b >= CG::Last::NBF) ? 0 : b;
    const int k = j % CG::NSL;
    f0[it] = twN_lookup(twN, (unsigned)((j - k) * CG::RL + k) * kk);
    if constexpr (TS == 1) f0[it] = cscale(f0[it], tq);         // the four-step factor carries the fixed-point scale
  }
  const cf gstep = twN_lookup_uniform(twN, (unsigned)CG::NSL * kk);      // k2 is wave-uniform: two scalar loads
  // Band-limited spectra (whitened plans, Geom::band_hi): column k2 is exactly zero for k1 in [za, zb],
  // i.e. where both k = k1 N2 + k2 and N - k lie above the band (60 % of a 0.1-10 Hz band at 50 Hz).
  // A wave instruction whose points ALL fall inside that range (kb .. kb + span - 1, kb = lane 0's point,
  // known at compile time per call) does not fetch them: its base pointer -- an SGPR pair, chosen by a
  // scalar select -- is moved so that the same lane offsets land on ONE fixed chunk of stored zeros at
  // the start of the range (512 B / 1 KB per column: L1 hits after the first touch).  No branch, no VALU
  // instruction and no VGPR: a skip branch around each load pair made the compiler wait for every pair
  // before issuing the next (24 serial round trips per wave), and per-lane selects of the offset or of
  // the loaded value cost one VGPR per load and split the gather in two at 128 VGPRs.  Instructions
  // that straddle a band edge load what is stored (zeros included).
  int za, zb;
  band_zero_stretch(g, k2, za, zb);
  const int zev = (za + 1) & ~1;                                  // first even index inside the range
  const bool zok = zb - zev + 1 >= 2 * kWNT;                      // room for the zero chunk of a 16-byte wave load
  auto zbase = [=](const cf* base, int kb, int span) -> const cf* {
    const bool uz = zok && kb >= za && kb + span - 1 <= zb;       // wave-uniform
    return uz ? base + (zev - kb) : base;
  };
  static_assert(ZB0 == 0 || MODE == 0, "the zero stretch is skipped for plain pair products only");
  constexpr int kZLo = ZB0 * kWNT, kZHi = N1 - ZB0 * kWNT - 1;           // elements [kZLo, kZHi] are zero in every column
  auto in = [=](int k1, int kb) -> cf {
    if constexpr (ZB0 > 0) {
      if (kb >= kZLo && kb + kWNT - 1 <= kZHi) return make_float2(0.f, 0.f);      // folds at compile time
    }
    if constexpr (MODE == 1) {
      // exactly two windows per station (the chirp convolutions of fftn.hip), four loads and no loop: as a
      // loop over nstack windows every call waited for its own loads -- 24 serial round trips per wave
      return cadd(cmulc(*uoff(A, (unsigned)k1), *uoff(Bp, (unsigned)k1)),
                  cmulc(*uoff(A, (unsigned)k1 + sstride), *uoff(Bp, (unsigned)k1 + sstride)));
    } else if constexpr (MODE == 2) {
      return *uoff(zbase(A, kb, kWNT), (unsigned)k1);
    } else {
      return cmulc(*uoff(zbase(A, kb, kWNT), (unsigned)k1), *uoff(zbase(Bp, kb, kWNT), (unsigned)k1));
    }
  };
  constexpr int TC = t_chunk(N1);
  cf* __restrict__ Tc = T + k2 * TC;
  const int crow = (N2h + 1) * TC;            // float2 per chunk of TC rows
  // Pair stores: every lane ends up with two adjacent rows and stores 16 bytes; lanes 0..31 cover one
  // 64-row half chunk, lanes 32..63 the next one, in lane order.
  auto epi = [=](int it, int j0, int, cf (&v)[CG::RL]) {
    cf g8 = gstep, f = f0[it];
    if constexpr (CG::RL > 8) g8 = csq(csq(csq(gstep)));
    static_for<CG::RL>([&](auto tc) {                                 // four-step factors, in place
      constexpr int t = decltype(tc)::value;
      if constexpr (t > 0) {
        if constexpr (t % 8 == 0) { f = f0[it]; static_for<t / 8>([&](auto) { f = cmul(f, g8); }); }
        else f = cmul(f, gstep);
      }
      v[t] = cmulc(v[t], f);
      if constexpr (TS == 1) v[t] = make_float2(q24_round(v[t].x), q24_round(v[t].y));     // integer bits from here on
    });
    if constexpr (kPairStore) {
      // NSL = 64: lane l holds residue l of the RL = N1/64 tiles a_l + t, a_l = a0 + (l < shift % 64)
      // (the slot rotation wraps at lane shift % 64).  First bring every lane to the same tile
      // numbering (w[tau] = tile a0 + tau), then swap within lane pairs.
      constexpr int NT = CG::RL;
      const int lane = tid_of<kWNT>();
      const int a0 = shift >> 6;
      // lane's residue is rlane (mod 64) in the tiles a0 + (rlane < shift % 64) + t: bring every lane
      // to the same tile numbering first (w[tau] = tile a0 + tau)
      const bool ahead = rlane < (shift & 63);
      const int up = lane >> 5;
      cf w[NT];
      static_for<NT>([&](auto tc) {
        constexpr int tau = decltype(tc)::value;
        w[tau] = ahead ? v[(tau + NT - 1) % NT] : v[tau];
      });
      static_for<NT / 2>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        // lanes [0,32) of w[2u] / w[2u+1]: rows 2L of tiles 2u / 2u+1; lanes [32,64): rows 2L+1.
        // swap(upper half of the first, lower half of the second):
        //   first  -> lanes [0,32): tile 2u row 2L    | lanes [32,64): tile 2u+1 row 2L
        //   second -> lanes [0,32): tile 2u row 2L+1  | lanes [32,64): tile 2u+1 row 2L+1
        const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[2 * u].x), __float_as_uint(w[2 * u + 1].x), false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(w[2 * u].y), __float_as_uint(w[2 * u + 1].y), false, false);
        int tile = a0 + 2 * u + up;
        if (tile >= NT) tile -= NT;
        // 24-bit multiplies (full rate; v_mul_lo_u32 is a quarter-rate instruction): tile < 64, crow and o < 2^24
        const unsigned o = (TC == 128) ? __umul24((unsigned)(tile >> 1), (unsigned)crow) + (unsigned)((tile & 1) * 64 + 2 * (lane & 31))
                                       : __umul24((unsigned)tile, (unsigned)crow) + (unsigned)(2 * (lane & 31));
        // plain global stores: sc1 / nt policies for T measured slower
        if constexpr (TS == 1) {
          typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
          unsigned q[3];
          q24_pack(sx[0], sy[0], sx[1], sy[1], q);
          char* tb = reinterpret_cast<char*>(T) + (size_t)((unsigned)(k2 * TC) * 6u);
          *reinterpret_cast<u32x3*>(tb + (size_t)__umul24(o, 6u)) = u32x3{q[0], q[1], q[2]};
        } else {
          *reinterpret_cast<float4*>(uoff(Tc, o)) =
              make_float4(__uint_as_float(sx[0]), __uint_as_float(sy[0]), __uint_as_float(sx[1]), __uint_as_float(sy[1]));
        }
      });
    } else {
      static_for<CG::RL>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        int res = j0 + t * CG::NSL + shift;
        if (res >= N1) res -= N1;
        *uoff(Tc, t_off(res, TC, N2h)) = v[t];
      });
    }
  };
#if !defined(XCX_STAMPS)
  // 16-byte spectrum loads for the first 2 x 64 butterflies (+0.9 %, profiles/r02_ablation.txt)
  if constexpr (MODE != 1) {
    auto in2 = [=](int k1, cf& ra, cf& rb, int kb) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      if constexpr (ZB0 > 0) {
        if (kb >= kZLo && kb + 2 * kWNT - 1 <= kZHi) { ra = rb = make_float2(0.f, 0.f); return; }
      }
      const f32x4 qa = *reinterpret_cast<const f32x4*>(uoff(zbase(A, kb, 2 * kWNT), (unsigned)k1));
      if constexpr (MODE == 2) {
        ra = make_float2(qa.x, qa.y); rb = make_float2(qa.z, qa.w);
      } else {
        const f32x4 qb = *reinterpret_cast<const f32x4*>(uoff(zbase(Bp, kb, 2 * kWNT), (unsigned)k1));
        ra = cmulc(make_float2(qa.x, qa.y), make_float2(qb.x, qb.y));
        rb = cmulc(make_float2(qa.z, qa.w), make_float2(qb.z, qb.w));
      }
    };
    wave_col_fft<RX, true>(lds, tw_sub, in, epi, ltid XCX_ST_ARG, in2);
  } else
#endif
  wave_col_fft<RX, true>(lds, tw_sub, in, epi, ltid XCX_ST_ARG);
  XCX_STAMP_NW(7, gstep.x);                          // T stores issued
  XCX_STAMP(8, gstep.x);                             // ... and acknowledged
}

template <class RX, int MODE>
__global__ __launch_bounds__(kWNT * kWPB) __attribute__((amdgpu_waves_per_eu(RX::L > 2048 ? 1 : 3, RX::L > 2048 ? 2 : 3)))
void k_wpair_cols(WI1Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const Geom g = a.g;
  const int N2h = g.N2h;
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave_in_group() * lds_col_floats(RX::L));
  const int x = blockIdx.x & 7;
  const int q = (blockIdx.x >> 3) * kWPB + wave_in_group();
  const int k2 = (q / a.B) * 8 + x;
  const int pwl = q % a.B;
  if (k2 > N2h) return;
#ifdef XCX_STAMPS
  StampArr xst;
#endif
  pair_col_body<RX, MODE>(lds, a.X, a.spec_pitch, pw_desc(a.pw, pwl), a.nstack, k2,
                           a.T + (size_t)pwl * RX::L * (N2h + 1), a.tw_sub, a.tw_N, g XCX_ST_ARG);
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

// T loads of the row body: plain, or sc1 (L1 bypass, served by the XCD's L2)
// when the producer ran in the same launch (flow kernel)
template <bool SC1> XCX_DEV cf load_t(const cf* p) {
  if constexpr (SC1) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
  } else {
    return *p;
  }
}

// row body shared by the pair kernels: tile `Tt` (16 rows,
// N2h+1 slots), output row `out`, block index blk
#ifndef XCX_CCF_AUX
#define XCX_CCF_AUX 2         // cache policy of the CCF stores: 2 = nt (streaming)
#endif
// COOP: the four waves of the workgroup own the four quarters of ONE 64-row tile (blk = 4 wgblk +
// wave) of one pair-window and store the CCF together: every wave parks its 16 x 2 N2h values in
// the workgroup's LDS as [lag block][64 rows], and the workgroup then writes 16 bytes per lane,
// 256 contiguous bytes per lag block (four blocks per wave instruction).  A wave alone can only
// ever write 64-byte runs (16 rows x 4 bytes, the next run N1 floats away): measured on the same
// traffic mix (tools/micro/fabric2.hip) 4-byte segmented stores cap the pass at 1.12 M
// pair-windows/s, 16-byte 256-byte runs at 1.57 M.  `stg` = workgroup-wide staging buffer
// (lds_stage_floats(N2h) floats, may alias the waves' exchange buffers); contains __syncthreads().
// TCT: T chunk rows (t_chunk(N1)) when N1 is known at compile time, 0 = from g.N1
// TS = 1 (q24 storage of T): every lane fetches the aligned 8 bytes that contain the 6 of its element (neighbouring
// lanes overlap by 2 bytes) -- the same lane mapping, load count and schedule as float2 storage.  A mapping with two
// adjacent rows per lane and 12- / 16-byte loads (half the load instructions) measured the same with q24 and 1.6 %
// SLOWER with float2 storage (profiles/r05_t24_notes.txt) and was dropped.
template <class RX, bool SC1, bool UNBIAS, int TCT = 0, int TS = 0>
XCX_DEV void pair_row_body(cf* __restrict__ lds, const cf* __restrict__ Tt, float* __restrict__ out, int blk,
                           const cf* __restrict__ tw_sub, const cf* __restrict__ twu,
                           float* __restrict__ stg, const Geom& g XCX_ST_PARAM, float tqi = 0.f) {
  constexpr int N2h = RX::L;
  constexpr int R = RX::R0;
  constexpr int S = N2h / R;
  constexpr int NITEM = S / 2 + 1;
  constexpr int TOTAL = NITEM * kWRB;
  constexpr int IT = (TOTAL + kWNT - 1) / kWNT;
  static_assert(TS == 0 || !SC1, "q24 storage: dual kernel only");
  const int N1 = g.N1, N = g.N, n = g.n;
  const int TC = TCT ? TCT : t_chunk(N1);          // elements from one k2 to the next
  const int lane = tid_of<kWNT>();
  static_assert(RX::S == 2, "row configs have two stages");
  cf b1[StageGeom<N2h, RX::R1, RX::R0, kWRB, kWNT>::IT];
  stage_tw_base<N2h, RX::R1, RX::R0, kWRB, kWNT>(tw_sub, b1);

  // ---- stage 0: paired butterflies (j, S-j) with the Hermitian untangle ----
  cf s0a[IT][R], s0b[IT][R];
  // Hermitian untangle, shared by the pair: with h = H[k] (k = j + S t),
  // g = H[N2h-k] = bv[R-1-t], e = e^{+2 pi i k/N2} = conj(twu[j]) e^{+2 pi i t/(2R)}
  //   sm = h + conj g,  ed = e (h - conj g)
  //   U_j [t]     = sm + i ed
  //   U_j'[R-1-t] = conj(sm) + i conj(ed)      (because e_{N2h-k} = -conj e_k)
  auto untangle = [&](int it, int j, bool single, cf (&av)[R], cf (&bv)[R], float ny, cf ej) {
    {
      // selects, not branches: the optimiser sinks the mirror loads into the arm of a branch that needs them, and a
      // load inside a branch waits alone
      // (j = S/2 exists for even S only: for odd S, N2h = 120 among them, the second select is not compiled in)
      const bool m0 = single && j == 0, m1 = (S % 2 == 0) && single && j != 0;
      cf sel[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const cf a0 = av[(t + 1) % R];                         // j = 0: mirror of S*t' is S*(R-t')
        const cf a1 = av[t];                                   // j = S/2 mirrors inside itself
        if constexpr (S % 2 == 0) {
          sel[t].x = m0 ? a0.x : (m1 ? a1.x : bv[t].x);
          sel[t].y = m0 ? a0.y : (m1 ? a1.y : bv[t].y);
        } else {
          sel[t].x = m0 ? a0.x : bv[t].x;
          sel[t].y = m0 ? a0.y : bv[t].y;
        }
      }
#pragma unroll
      for (int t = 0; t < R; ++t) bv[t] = sel[t];
    }
    cf uj[R], ujp[R];
    static_for<R>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      const cf h = av[t], hb = cconj(bv[R - 1 - t]);
      const cf sm = cadd(h, hb), df = csub(h, hb);
      const cf ed = cmul(mul_wconst<t, 2 * R, true>(ej), df);
      uj[t] = make_float2(sm.x - ed.y, sm.y + ed.x);
      ujp[R - 1 - t] = make_float2(sm.x + ed.y, ed.x - sm.y);
    });
    if (j == 0) {                                  // DC and Nyquist of the row are real
      const float dc = av[0].x;
      uj[0] = make_float2(dc + ny, dc - ny);
    }
    Bfly<R, true>::run(uj);
    Bfly<R, true>::run(ujp);        // also for `single` lanes (result unused): a branch here is where the optimiser sinks the mirror loads
    // Stockham positions of stage 0 (NS = 1): butterfly j -> j*R + t; real and
    // imaginary parts go through the half-size LDS buffer in two rounds below
#pragma unroll
    for (int t = 0; t < R; ++t) { s0a[it][t] = uj[t]; s0b[it][t] = ujp[t]; }
  };
  {
    // Every load of an item is issued without a branch in between, the untangle twiddles first (round 5).  Until then
    // the mirror loads sat under `if (!single)`, the Nyquist load under `if (j == 0)` and the twiddle load inside the
    // untangle: the generated code waited for the first loads before the branch, for the Nyquist load inside it and
    // for the twiddle after it -- three to four dependent round trips through a congested memory pipeline per item.
    // Lanes of a `single` butterfly (j = 0: its mirror is itself) now fetch the same rows twice inside the same
    // instructions, every lane fetches the Nyquist bin of its row; the values are discarded where they do not apply.
    cf ejs[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) ejs[it] = cconj(*uoff(twu, (unsigned)((lane + it * kWNT) / kWRB)));
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int w = lane + it * kWNT;
      if ((TOTAL % kWNT) != 0 && w >= TOTAL) continue;
      const int r = w % kWRB;
      const int j = w / kWRB;                       // 0..S/2
      const int jp = (j == 0) ? 0 : (S - j);
      const bool single = jp == j;
      cf av[R], bv[R];
      float ny = 0.f;
      if constexpr (TS == 1) {
        // element e = k2 TC + r sits at byte 6 e; the aligned 8 bytes from 6 e - 2 (r odd) or 6 e (r even) hold it
        const char* __restrict__ Tb = reinterpret_cast<const char*>(Tt);
        const unsigned odd = (unsigned)(r & 1);
        const unsigned boff = 6u * (unsigned)r - 2u * odd;
        const unsigned sel_re = odd ? 0x0403020cu : 0x0201000cu;       // three bytes to the top of a dword, low byte 0
        const unsigned sel_im = odd ? 0x0706050cu : 0x0504030cu;
        // addresses as (uniform base per slot, laundered through scalar registers) + (one per-lane 32-bit offset for the
        // direct rows, one for the mirror rows): written with a per-lane element index, the optimiser reassociates
        // base + constant + offset into (base + offset) + constant and builds a 64-bit vector address per load
        typedef const __attribute__((address_space(1))) char* gchar;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) u32x2* gu2;
        auto ld6 = [&](int slot_k2, unsigned voff) -> cf {
          gchar ub = (gchar)(Tb + (size_t)slot_k2 * (size_t)TC * 6u);
          asm volatile("" : "+s"(ub));
          const u32x2 q = *(gu2)(ub + (size_t)voff);
          const int re = (int)__builtin_amdgcn_perm(q.y, q.x, sel_re);
          const int im = (int)__builtin_amdgcn_perm(q.y, q.x, sel_im);
          return make_float2((float)re, (float)im);          // x 256 tq: the scale rides on the second stage's twiddle chain
        };
        unsigned va = (unsigned)(j * TC) * 6u + boff, vb = (unsigned)(jp * TC) * 6u + boff;
        asm volatile("" : "+v"(va), "+v"(vb));       // opaque 32-bit offsets: merged with other index arithmetic they come back as 64-bit pairs
#pragma unroll
        for (int t = 0; t < R; ++t) av[t] = ld6(S * t, va);
#pragma unroll
        for (int t = 0; t < R; ++t) bv[t] = ld6(S * t, vb);
        ny = ld6(N2h, boff).x;
      } else {
        if constexpr (SC1) {
#pragma unroll
          for (int t = 0; t < R; ++t) av[t] = load_t<SC1>(uoff(Tt, (unsigned)((j + S * t) * TC + r)));
#pragma unroll
          for (int t = 0; t < R; ++t) bv[t] = load_t<SC1>(uoff(Tt, (unsigned)((jp + S * t) * TC + r)));
          ny = load_t<SC1>(uoff(Tt, (unsigned)(N2h * TC + r))).x;
        } else {
          // the same addressing as the q24 loads above: scalar base per slot, one per-lane offset per row set
          typedef const __attribute__((address_space(1))) char* gchar;
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          typedef const __attribute__((address_space(1))) f32x2* gf2;
          auto ld8 = [&](int slot_k2, unsigned voff) -> cf {
            gchar ub = (gchar)(Tt + (size_t)slot_k2 * (size_t)TC);
            asm volatile("" : "+s"(ub));
            const f32x2 v = *(gf2)(ub + (size_t)voff);
            return make_float2(v.x, v.y);
          };
          unsigned va = (unsigned)(j * TC + r) * 8u, vb = (unsigned)(jp * TC + r) * 8u;
          asm volatile("" : "+v"(va), "+v"(vb));
#pragma unroll
          for (int t = 0; t < R; ++t) av[t] = ld8(S * t, va);
#pragma unroll
          for (int t = 0; t < R; ++t) bv[t] = ld8(S * t, vb);
          ny = ld8(N2h, (unsigned)r * 8u).x;
        }
      }
      untangle(it, j, single, av, bv, ny, ejs[it]);
    }
    XCX_STAMP(2, s0a[0][0].x);
  }
  float* __restrict__ ldsf = reinterpret_cast<float*>(lds);
  using G1 = StageGeom<N2h, RX::R1, RX::R0, kWRB, kWNT>;
  cf v1[G1::IT][RX::R1];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int w = lane + it * kWNT;
      if ((TOTAL % kWNT) != 0 && w >= TOTAL) continue;
      const int r = w % kWRB;
      const int j = w / kWRB;
      const int jp = (j == 0) ? 0 : (S - j);
      // rpad(p0 + d, r) = rpad(p0, r) + (d + d/8) * 16 when the group of 8 is not left (R % 8 == 0,
      // p0 = j R) or d is a multiple of 8: one base per butterfly, immediates for the rest
      constexpr bool WF = (R % 8 == 0);
      const int wa = rpad(j * R, r), wb = rpad(jp * R, r);
#pragma unroll
      for (int t = 0; t < R; ++t)
        ldsf[WF ? wa + (t + (t >> 3)) * kWRB : rpad(j * R + t, r)] = half ? s0a[it][t].y : s0a[it][t].x;
      if (jp != j) {
#pragma unroll
        for (int t = 0; t < R; ++t)
          ldsf[WF ? wb + (t + (t >> 3)) * kWRB : rpad(jp * R + t, r)] = half ? s0b[it][t].y : s0b[it][t].x;
      }
    }
    group_sync<kWNT>();
#pragma unroll
    for (int it = 0; it < G1::IT; ++it) {
      const int b = lane + it * kWNT;
      if (!G1::GUARD || b < G1::NBF) {
        const int c = b % kWRB, j = b / kWRB;
        constexpr int RD = N2h / RX::R1;
        constexpr bool RF = (RD % 8 == 0);
        const int rb = rpad(j, c);
#pragma unroll
        for (int t = 0; t < RX::R1; ++t) {
          const float f = ldsf[RF ? rb + (t * RD + (t * RD >> 3)) * kWRB : rpad(j + t * RD, c)];
          if (half) v1[it][t].y = f; else v1[it][t].x = f;
        }
      }
    }
    group_sync<kWNT>();
  }
  XCX_STAMP(3, v1[G1::IT - 1][RX::R1 - 1].y);
  static_assert(TS != 1 || RX::R1 <= 16, "q24: the scale rides on a twiddle chain with seeds at 1 and 8");
  if constexpr (TS == 1) stage_compute_scaled<N2h, RX::R1, RX::R0, kWRB, kWNT, true>(v1, b1, tqi * (1.f / 256.f));
  else stage_compute<N2h, RX::R1, RX::R0, kWRB, kWNT, true>(v1, b1);
  XCX_STAMP(4, v1[G1::IT - 1][RX::R1 - 1].y);
  // ---- second stage + output ---------------------------------------------
  const int idx_min = g.idx_min;
  const unsigned Lout = (unsigned)(g.idx_max - g.idx_min);
  const int shift = g.shift;
  const int wrap_at = g.wrap;
  // Buffer stores through a resource that spans exactly the retained lags of this output
  // row: the hardware range check drops every out-of-range lag (negative indices wrap to
  // huge unsigned offsets), so a store needs a 32-bit offset and no compare / select /
  // 64-bit address arithmetic.
  const unsigned long long ob = reinterpret_cast<unsigned long long>(out);
  const unsigned ob_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32));
  const unsigned ob_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);   // unsigned: no sign extension
  float* out_u = reinterpret_cast<float*>(((unsigned long long)ob_hi << 32) | (unsigned long long)ob_lo);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out_u, 0, (int)(Lout * 4u), 0x00020000);
  {
    // Row res = 16 blk + c of the tile holds, for kk = 2 m + h, the lag o = res + N1 (kk + wr + q0)
    // (before the circular wrap), wr = 1 for the rows whose m1 + shift passed N1 (res < shift),
    // q0 = floor((n - 1 - idx_min) / N1).  Lag block s = kk + wr is therefore common to all 64 rows.
    // Two passes (kk < N2h, then the rest) keep the staging buffer at (N2h + 1) x 64 floats.
    constexpr int KH = N2h;
    const int c = lane & 15;
    const int res_l = blk * kWRB + c;
    const int wr = (res_l < shift) ? 1 : 0;
    const int q0 = ((n - 1 - idx_min) - shift) / N1;          // exact: shift = (n - 1 - idx_min) mod N1
    const int col = (blk & 3) * kWRB + c;                      // row inside the tile
    const int res0 = (blk >> 2) * kTR;                         // first residue of the tile
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();                                         // exchange buffers / previous pass drained
#pragma unroll
      for (int it = 0; it < G1::IT; ++it) {
        const int b = lane + it * kWNT;
        if (!G1::GUARD || b < G1::NBF) {
          const int j = b / kWRB;
          const int k = j % RX::R0;
          const int j0 = (j - k) * RX::R1 + k;
#pragma unroll
          for (int t = 0; t < RX::R1; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int kk = 2 * (j0 + t * RX::R0) + h;
              if ((kk >= KH) == (half == 1)) {
                float v = h ? v1[it][t].y : v1[it][t].x;
                if constexpr (UNBIAS) {
                  int o = res_l + N1 * (kk + wr + q0);
                  if (o >= wrap_at) o -= N;
                  int lag = o + idx_min - (n - 1);
                  lag = lag < 0 ? -lag : lag;
                  v *= (float)n / (float)(n - lag);
                }
                stg[(kk + wr - half * KH) * kTR + col] = v;
              }
            }
          }
        }
      }
      __syncthreads();
      // lag blocks s = half KH .. half KH + KH of this pass; 16 float4 groups per block.  The staging reads of a
      // pass are all issued before the first store (round 5): as a rolled loop every iteration waited for its own
      // ds_read_b128 -- eight dependent LDS round trips per pass and thread.
      constexpr int kGroups = (KH + 1) * (kTR / 4);
      constexpr int kStIt = (kGroups + kWNT * kWPB - 1) / (kWNT * kWPB);
#ifndef XCX_STG_BATCH
#define XCX_STG_BATCH 2          // staging reads in flight per thread (measured: 2 +0.8 %, 4 +0.7 %, 8 = all of a pass: 22 spilled registers, +0.0 %)
#endif
      float4 vq[kStIt];
#pragma unroll
      for (int q = 0; q < kStIt; ++q) {
        if (q % XCX_STG_BATCH == 0) {
#pragma unroll
          for (int qq = q; qq < q + XCX_STG_BATCH && qq < kStIt; ++qq) {
            const int i2 = tid + qq * (kWNT * kWPB);
            const int ic = i2 < kGroups ? i2 : kGroups - 1;      // the ragged tail re-reads the last group (no branch)
            vq[qq] = *reinterpret_cast<const float4*>(stg + (ic >> 4) * kTR + 4 * (ic & 15));
          }
        }
        // Per thread and pass only the lag block moves (sl = sl0 + 16 q: the residues r0 .. r0 + 3 of a thread's groups are
        // the same for every q), so everything but one literal add, the circular wrap and one unsigned range compare is
        // computed once per thread (round 5: as written per group it cost a 64-bit multiply-add and ~20 instructions for
        // each of the 16 groups, a tenth of a row wave's instructions -- and the pass runs at the power cap)
        constexpr int kPerIt = kWNT * kWPB;
        static_assert(kPerIt % (kTR / 4) == 0, "a thread keeps its residues from one group to the next");
        const int i = tid + q * kPerIt;
        if ((q + 1) * kPerIt > kGroups && i >= kGroups) continue;      // only the last trip is ragged
        const int gq = tid & 15, sl0 = tid >> 4;
        const int sl = sl0 + q * (kPerIt >> 4);
        const int s = sl + half * KH;
        const float4 v = vq[q];
        const int r0 = res0 + 4 * gq;                          // residues r0 .. r0 + 3
        const int ob = r0 + N1 * (sl0 + q0);                   // q = 0, first pass, before the wrap
        int o = ob + N1 * (q * (kPerIt >> 4) + half * KH);     // a literal away from it
        if (o >= wrap_at) o -= N;                          // N, N1, r0 are multiples of 4: never splits a group
        // element e is real data iff its row has kk = s - wr_e inside this pass, and retained iff
        // 0 <= o + e < Lout
        const int wr0 = (r0 < shift) ? 1 : 0, wr3 = (r0 + 3 < shift) ? 1 : 0;
        // kk = sl - wr in [0, KH): only the first and the last lag blocks of a pass can miss it
        const bool edge = (q == 0) || ((q + 1) * (kPerIt >> 4) > KH);
        const bool in_pass = !edge || ((sl - wr0) >= 0 && (sl - wr0) < KH);
        const bool uniform = wr0 == wr3 && in_pass;
        const unsigned lim = Lout >= 3u ? Lout - 3u : 0u;       // o >= 0 && o + 3 < Lout as one unsigned compare
        if (uniform && (unsigned)o < lim) {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 w4 = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
          __builtin_amdgcn_raw_buffer_store_b128(w4, rsrc, o * 4, 0, XCX_CCF_AUX);
        } else {
          const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = s - ((r0 + e < shift) ? 1 : 0);
            if (kk >= half * KH && kk < (half + 1) * KH)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ve[e]), rsrc, (o + e) * 4, 0, XCX_CCF_AUX);   // range-checked
          }
        }
      }
    }
  }
  XCX_STAMP_NW(7, v1[0][0].x);                       // CCF stores issued
  XCX_STAMP(8, v1[0][0].x);                          // ... and acknowledged
}

template <class RX>
__global__ __launch_bounds__(kWNT * kWPB) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_wpair_rows(WI2Args a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const Geom g = a.g;
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave_in_group() * lds_row_floats(RX::L));
  // the four waves of a workgroup = the four quarters of one 64-row tile (cooperative CCF store)
  const int pwl = blockIdx.y, blk = blockIdx.x * kWPB + wave_in_group();
  const cf* __restrict__ Tt = t_rows16(a.T + (size_t)pwl * g.N1 * (RX::L + 1), blk, RX::L, t_chunk(g.N1));
  float* stg = reinterpret_cast<float*>(lds_all);
#ifdef XCX_STAMPS
  StampArr xst;
#endif
  if (g.flags & XCX_F_UNBIAS)
    pair_row_body<RX, false, true>(lds, Tt, a.out + pw_desc(a.pw, pwl).out_off, blk, a.tw_sub, a.tw_unt, stg, g XCX_ST_ARG);
  else
    pair_row_body<RX, false, false>(lds, Tt, a.out + pw_desc(a.pw, pwl).out_off, blk, a.tw_sub, a.tw_unt, stg, g XCX_ST_ARG);
}

// ---------------------------------------------------------------------------
// Dual pair kernel: ONE launch runs the row pass (I2) of batch b-1 and the
// column pass (I1) of batch b, interleaved per XCD in a 16:12 pattern, so the
// VALU-bound column waves and the HBM-bound row waves share every CU.  No
// dependency exists inside the launch (the rows read the T buffer the previous
// launch wrote; the columns write the other one): no inter-wave wait anywhere.
// ---------------------------------------------------------------------------
struct WDualArgs {
  const cf* X; size_t spec_pitch;
  const PwDesc* pw_c; int Bc; int nstack; cf* Tc;       // column pass: batch b
  const PwDesc* pw_r; int Br; const cf* Tr; float* out;               // row pass: batch b-1
  const cf* tw_N1; const cf* tw_N2h; const cf* tw_unt; const cf* tw_N;
  Geom g;
  int ts;                          // T storage: 0 float2, 1 q24 (whitened plans; tq = fixed-point scale, tqi = 1 / tq)
  float tq, tqi;
  int zb0;                         // q24 launches: instantiated ZB0 of the column body (0: none), from zskip_pick()
#ifdef XCX_STAMPS
  unsigned long long* stamps;      // [waves of the launch][kNStamp] or null
#endif
};
// Zero-stretch skipping (template parameter ZB0 of the column body): the instantiated values per column length, in
// 64-point blocks, and the choice for a plan.  A band [0, band_hi] of an N-point spectrum leaves the blocks
// first .. last of every column k2 = 0 .. N2h inside the zero stretch; the largest instantiated symmetric range
// ZB0 .. N1/64 - 1 - ZB0 inside first .. last is taken (0: none fits, the generic kernel runs).
template <int N1> struct ZSkipList { static constexpr int n = 0; static constexpr int v[1] = {0}; };
template <> struct ZSkipList<1536> { static constexpr int n = 4; static constexpr int v[4] = {3, 5, 7, 9}; };
template <> struct ZSkipList<1024> { static constexpr int n = 3; static constexpr int v[3] = {2, 4, 6}; };
template <> struct ZSkipList<768> { static constexpr int n = 3; static constexpr int v[3] = {2, 3, 4}; };
template <> struct ZSkipList<512> { static constexpr int n = 2; static constexpr int v[2] = {1, 2}; };
inline int zskip_need(const Geom& g) {
  // smallest ZB0 whose range lies inside the zero stretch of every column; 0: there is none
  if (g.band_hi >= g.N) return 0;
  int za_max = 0, zb_min = g.N1;
  for (int k2 = 0; k2 <= g.N2h; ++k2) {
    const int za = (g.band_hi >= k2) ? (g.band_hi - k2) / g.N2 + 1 : 0;
    const int zb = (g.N - g.band_hi - k2 + g.N2 - 1) / g.N2 - 1;
    za_max = za > za_max ? za : za_max;
    zb_min = zb < zb_min ? zb : zb_min;
  }
  const int nb = g.N1 / kWNT;
  const int first = (za_max + kWNT - 1) / kWNT, last = (zb_min + 1) / kWNT - 1;
  const int need = first > nb - 1 - last ? first : nb - 1 - last;
  return (need >= 1 && 2 * need < nb) ? need : 0;
}
template <int N1> inline int zskip_pick(const Geom& g) {
  const int need = zskip_need(g);
  if (need == 0) return 0;
  for (int i = 0; i < ZSkipList<N1>::n; ++i)
    if (ZSkipList<N1>::v[i] >= need && 2 * ZSkipList<N1>::v[i] < N1 / kWNT) return ZSkipList<N1>::v[i];
  return 0;
}
// q24 storage exists for the N1 whose column wave stores lane pairs (any row length)
template <class RXC, class RXR> struct DualQ24 {
  static constexpr bool ok = ColPairStore<RXC>::ok;
};

// amdgpu_waves_per_eu(4,4): the pair pass is bound by memory latency x occupancy (the arithmetic
// is hidden: a build without butterflies runs only 8 % faster), so the fourth wave per SIMD counts
// for more than a spill-free allocation: 128 VGPRs with 12 bytes of scratch measured +1.6 % over
// 137 VGPRs / 3 waves, and 3 waves forced by LDS padding (XCX_DUAL_WGS=3) -10 %.
#ifndef XCX_DUAL_WAVES
#define XCX_DUAL_WAVES 4
#endif
// work decomposition shared by the dual and flow kernels: per XCD and pair-window
//   CWX column workgroups (four columns k2 = x mod 8 each; consecutive items = the same column
//       of consecutive pair-windows, so a station column stays in that XCD's L2), and
//   RWX row workgroups (one 64-row tile each: cooperative CCF store),
// interleaved CWX : RWX (4 : 3 at N1 = 1536, N2h = 120) so that VALU-heavy column waves and
// store-heavy row waves share every CU.
template <class RXC, class RXR> struct PairGeom {
  static constexpr int N1 = RXC::L, N2h = RXR::L;
  static constexpr int NR = N1 / kWRB;                 // 16-row quarters per pair-window
  static constexpr int NRW = N1 / kTR;                 // row workgroups (tiles) per pair-window
  static constexpr int CPX = (N2h + 1 + 7) / 8;        // columns per XCD
  static constexpr int CWX = (CPX + kWPB - 1) / kWPB;  // column workgroups per XCD and pair-window
  static constexpr int RWX = (NRW + 7) / 8;            // row workgroups per XCD and pair-window
  static constexpr int kPerWave = lds_col_floats(N1) > lds_row_floats(N2h) ? lds_col_floats(N1) : lds_row_floats(N2h);
  static constexpr int kLdsFloats = kPerWave * kWPB > lds_stage_floats(N2h) ? kPerWave * kWPB : lds_stage_floats(N2h);
  static_assert(N1 % kTR == 0, "column lengths are multiples of 64");
};

template <class RXC, class RXR, int MODE, int TS = 0, int ZB0 = 0>
__global__ __launch_bounds__(kWNT * kWPB) __attribute__((amdgpu_waves_per_eu(RXC::L > 2048 ? 2 : XCX_DUAL_WAVES, RXC::L > 2048 ? 2 : XCX_DUAL_WAVES)))
void k_wpair_dual(WDualArgs a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const Geom g = a.g;
  using PG = PairGeom<RXC, RXR>;
  constexpr int N1 = PG::N1, N2h = PG::N2h;
  const int wave = wave_in_group();
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave * PG::kPerWave);
  const int x = blockIdx.x & 7;
  const int wgq = (int)(blockIdx.x >> 3);
  const int grp = wgq / (PG::CWX + PG::RWX), pos = wgq % (PG::CWX + PG::RWX);
#ifdef XCX_STAMPS
  const int q = (int)blockIdx.x * kWPB + wave;
  StampArr xst;
  for (int i = 0; i < kNStamp; ++i) xst[i] = 0;
  xst[0] = stamp_nowait(0.f);
  struct StampOut {
    StampArr& t; unsigned long long* dst; int q, kind;
    __device__ ~StampOut() {
      if (!dst || (threadIdx.x & 63)) return;
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      t[9] = stamp_nowait(0.f);
      t[10] = ((unsigned long long)kind << 32) | hwid;
      t[11] = __builtin_amdgcn_s_memrealtime();
      for (int i = 0; i < kNStamp; ++i) dst[(size_t)q * kNStamp + i] = t[i];
    }
  } stamp_out{xst, a.stamps, q, pos < PG::CWX ? 1 : 2};
#endif
  if (pos < PG::CWX) {
    // the (N2h + 1) Bc column items (k2-major: consecutive items = one column of consecutive
    // pair-windows, so a station column is re-read from the XCD's L2) are cut into 8 EQUAL shares,
    // one per XCD label: with k2 = x mod 8, XCD 0 had 16 columns of 121 and the others 15
    const int ci = (grp * PG::CWX + pos) * kWPB + wave;
    const int per = ((N2h + 1) * a.Bc + 7) >> 3;
    const int item = x * per + ci;
    if (ci >= per || item >= (N2h + 1) * a.Bc) return;
    const int k2 = item / a.Bc;
    const int pwl = item - k2 * a.Bc;
    pair_col_body<RXC, MODE, TS, ZB0, PG::N2h>(lds, a.X, a.spec_pitch, pw_desc(a.pw_c, pwl), a.nstack, k2,
                                  a.Tc + (size_t)pwl * N1 * (N2h + 1), a.tw_N1, a.tw_N, g XCX_ST_ARG, a.tq);
  } else {
    const int ri = grp * PG::RWX + (pos - PG::CWX);
    const int pwl = ri / PG::RWX;
    const int wgblk = (ri % PG::RWX) * 8 + x;
    if (wgblk >= PG::NRW || pwl >= a.Br) return;             // workgroup-uniform
    const int blk = wgblk * kWPB + wave;
    const cf* __restrict__ Tt = reinterpret_cast<const cf*>(
        t_rows16_b<TS>(a.Tr + (size_t)pwl * N1 * (N2h + 1), blk, N2h, t_chunk(N1)));
    float* stg = reinterpret_cast<float*>(lds_all);
    if (g.flags & XCX_F_UNBIAS)
      pair_row_body<RXR, false, true, t_chunk(N1), TS>(lds, Tt, a.out + pw_desc(a.pw_r, pwl).out_off, blk, a.tw_N2h, a.tw_unt, stg, g XCX_ST_ARG, a.tqi);
    else
      pair_row_body<RXR, false, false, t_chunk(N1), TS>(lds, Tt, a.out + pw_desc(a.pw_r, pwl).out_off, blk, a.tw_N2h, a.tw_unt, stg, g XCX_ST_ARG, a.tqi);
  }
}

// ---------------------------------------------------------------------------
// Flow pair kernel: the WHOLE pair phase (every batch) in ONE launch.
//
// The dual kernel above is launched once per batch of 25 pair-windows so that the T ring stays
// inside the Infinity Cache; a launch of ~1.3 machine rounds starts all its waves together,
// which then load, compute and store in lock-step (bursts on the memory pipeline, then on the
// VALU: SQ_WAIT_INST_ANY 41 % at 25 % VALU use) and drains through a tail before the stream's
// next launch may begin.  Here the same work items are laid out for all batches back to back:
//   group g  =  column items of batch g  (interleaved CPX:RPX per XCD as in k_wpair_dual)
//             + row items of batch g-2,
// workgroups are dispatched in index order, every wave slot is refilled the moment a wave
// retires, and the phases of neighbouring waves drift apart: a steady, desynchronised flow.
// The batch-level dependencies become counters in memory:
//   rows of batch b  wait until cdone[b] == (N2h+1) * B_b   (all its columns stored T),
//   cols of batch b  wait until rdone[b-R] == NR * B_{b-R}  (ring slot b mod R consumed).
// The rows of batch b ride in group b+2 and R >= 4, so both refer to items at least one WHOLE
// group (~1.3 machine rounds of other work) earlier in dispatch order and a wait is normally
// satisfied on its first poll; every wait is bounded and a
// timeout raises *err (the host then reports XCX_EHIP instead of returning wrong numbers).
// Progress: workgroups of one XCD are dispatched in index order and a wave only ever waits
// for lower-indexed items, so the lowest unfinished item is always resident and runnable.
// T lives in UNCACHED device memory (hipDeviceMallocUncached): its producers and consumers
// sit on different XCDs whose L2s are not coherent, and inside one launch there is no kernel
// boundary to write back / invalidate them; uncached accesses go straight to the fabric
// (Infinity Cache), which is where T traffic went anyway.  A column wave signals after
// s_waitcnt vmcnt(0) (its T stores are acknowledged by the fabric).
// ---------------------------------------------------------------------------
struct WFlowArgs {
  const cf* X; size_t spec_pitch;
  const PwDesc* pw; int count;      // pair-windows of the whole run
  int B;                            // pair-windows per batch (T ring granule)
  int R;                            // ring size in batches
  int nstack;
  cf* T;                            // [R][B][N1/16][N2h+1][16], uncached
  float* out;
  int* cdone; int* rdone;           // [batches][8 shards][32 ints], zeroed before the launch
  int* err;
  const cf* tw_N1; const cf* tw_N2h; const cf* tw_unt; const cf* tw_N;
  Geom g;
  int lag;                          // rows of batch g-lag ride in group g (kFlowLag by default)
};

// completion counters are sharded by the XCD label of the signalling wave (8 words, one 128-byte
// line each): a single word takes ~12 ns per atomic, i.e. 36 us for the 3025 column waves of a
// batch -- longer than the batch itself.  A waiter reads the 8 shards with one wave instruction.
constexpr int kFlowShardInts = 32;                           // ints per shard (one cache line)
constexpr int kFlowLag = 2;                                  // rows of batch g-2 ride in group g

XCX_DEV bool flow_wait8(const int* shards, int target_lane, int* err) {
  const int lane = tid_of<kWNT>();
  bool ok = false;
  for (int i = 0; i < (1 << 21); ++i) {                      // ~2 s at 1 us per poll
    const int v = (lane < 8) ? __hip_atomic_load(shards + lane * kFlowShardInts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : 0x7fffffff;
    if (__builtin_amdgcn_ballot_w64(lane < 8 && v < target_lane) == 0) { ok = true; break; }
    // somebody already timed out: the launch is lost, drain it quickly
    if ((i & 15) == 15 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    __builtin_amdgcn_s_sleep(32);
  }
  if (!ok && lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  return ok;
}

template <class RXC, class RXR, int MODE>
__global__ __launch_bounds__(kWNT * kWPB) __attribute__((amdgpu_waves_per_eu(RXC::L > 2048 ? 2 : XCX_DUAL_WAVES, RXC::L > 2048 ? 2 : XCX_DUAL_WAVES)))
void k_wpair_flow(WFlowArgs a) {
  extern __shared__ __attribute__((aligned(16))) cf lds_all[];
  const Geom g = a.g;
  using PG = PairGeom<RXC, RXR>;
  constexpr int N1 = PG::N1, N2h = PG::N2h;
  const int wave = wave_in_group();
  cf* __restrict__ lds = reinterpret_cast<cf*>(reinterpret_cast<float*>(lds_all) + wave * PG::kPerWave);
  const int B = a.B;
  const int x = blockIdx.x & 7;
  const int wpg = (PG::CWX + PG::RWX) * B;                   // workgroups per (group, XCD)
  const int wg = (int)(blockIdx.x >> 3);
  const int grp_b = wg / wpg;                                 // group = batch index of the column items
  const int wgq = wg - grp_b * wpg;
  const int nbatch = (a.count + B - 1) / B;
  const int grp = wgq / (PG::CWX + PG::RWX), pos = wgq % (PG::CWX + PG::RWX);
  const int lane = tid_of<kWNT>();
  // per-shard item counts of one pair-window (lane l < 8 holds shard l's): column waves with
  // k2 = l mod 8, row waves of the tiles l mod 8
  const int ncx = (lane <= N2h) ? (N2h - lane) / 8 + 1 : 0;
  const int nrx = (lane < PG::NRW) ? ((PG::NRW - 1 - lane) / 8 + 1) * kWPB : 0;
#ifdef XCX_STAMPS
  StampArr xst;
#endif
  if (pos < PG::CWX) {
    const int b = grp_b;
    if (b >= nbatch) return;
    const int Bc = min(B, a.count - b * B);
    const int ci = (grp * PG::CWX + pos) * kWPB + wave;
    const int k2 = (ci / B) * 8 + x;
    const int pwl = ci % B;
    if (k2 > N2h || pwl >= Bc) return;
    if (b >= a.R) {                                           // ring slot consumed?
      const int bp = b - a.R;
      const int Bp = min(B, a.count - bp * B);
      const int tgt = nrx * Bp;
      if (!flow_wait8(a.rdone + (size_t)bp * 8 * kFlowShardInts, tgt, a.err)) return;
    }
    cf* Tb = a.T + (size_t)(b % a.R) * B * N1 * (N2h + 1);
    pair_col_body<RXC, MODE>(lds, a.X, a.spec_pitch, pw_desc(a.pw, b * B + pwl), a.nstack, k2,
                              Tb + (size_t)pwl * N1 * (N2h + 1), a.tw_N1, a.tw_N, g XCX_ST_ARG);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // T stores acknowledged
    if (lane == 0)
      __hip_atomic_fetch_add(a.cdone + ((size_t)b * 8 + x) * kFlowShardInts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    const int b = grp_b - a.lag;
    if (b < 0 || b >= nbatch) return;
    const int Br = min(B, a.count - b * B);
    const int ri = grp * PG::RWX + (pos - PG::CWX);
    const int pwl = ri / PG::RWX;
    const int wgblk = (ri % PG::RWX) * 8 + x;
    if (wgblk >= PG::NRW || pwl >= Br) return;               // workgroup-uniform
    const int blk = wgblk * kWPB + wave;
    // a timed-out wait must not leave the workgroup's barriers one wave short: all four waves
    // poll the same counters and agree
    const int ctgt = ncx * Br;
    if (!flow_wait8(a.cdone + (size_t)b * 8 * kFlowShardInts, ctgt, a.err)) return;
    const cf* __restrict__ Tt = t_rows16(a.T + ((size_t)(b % a.R) * B + pwl) * (size_t)N1 * (N2h + 1), blk, N2h, t_chunk(N1));
    float* o = a.out + pw_desc(a.pw, b * B + pwl).out_off;
    float* stg = reinterpret_cast<float*>(lds_all);
    if (g.flags & XCX_F_UNBIAS)
      pair_row_body<RXR, true, true, t_chunk(N1)>(lds, Tt, o, blk, a.tw_N2h, a.tw_unt, stg, g XCX_ST_ARG);
    else
      pair_row_body<RXR, true, false, t_chunk(N1)>(lds, Tt, o, blk, a.tw_N2h, a.tw_unt, stg, g XCX_ST_ARG);
    // every T load of this wave has returned (the values were consumed by the stores above)
    if (lane == 0)
      __hip_atomic_fetch_add(a.rdone + ((size_t)b * 8 + x) * kFlowShardInts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------
// Time-stack as a batched Hermitian rank-W update (the cgemm reformulation of
// SURVEY 8d): for every frequency k, G_k[i][j] = sum_w X_{i,w}[k] conj X_{j,w}[k].
// One wave = 64 consecutive frequencies (one per lane, so every load is a
// contiguous 512-byte run) x one 8x8 station tile held as 64 complex
// accumulators in registers; per window it loads 16 values and does 64 complex
// multiply-adds.  XCD-aware order: all tiles of a frequency chunk run back to
// back on one XCD, so the chunk of the S*W spectra is fetched from HBM once
// and re-read from that L2 by the other tiles.
// ---------------------------------------------------------------------------
constexpr int kGT = 8;           // station tile edge

struct GramArgs {
  const cf* X; size_t spec_pitch;   // X[(s*W + w)*spec_pitch + k]
  int S, W;
  int nfreq;                        // spectrum elements (N1*(N2h+1))
  const int2* tiles; int ntiles;    // (first station of the i block, of the j block)
  const int* pidx;                  // [S*S] -> row of C, or -1
  cf* C; size_t c_pitch;            // C[p*c_pitch + k]
  int accumulate;                   // 0: C = G;  1: C += G (multi-day stack, xcx_cc1d_stack_add)
  int ldw;                          // slots per station (>= W; 0: W): X[(s*ldw + w)*spec_pitch + k]
  Geom g;                           // band-limited plans (g.band_hi < g.N): frequency quads inside the zero stretch of
                                    // their column are skipped -- C is zeroed once per allocation and never written there
};

hipError_t launch_gram(const GramArgs& a, hipStream_t s);

}  // namespace xcx
