// cc2d.hip -- two-dimensional cross-correlation of tiles (xcorr/signal/correlate.py:
// 162-291, norm2d of signal/normalize.py:95-155; the kernel of the timelapse script,
// scripts/timelapse.py:243-268, which keeps only the peak value and its position).
//
//   cc[i][j] = sum_{a,b} f[a + i-(F-1)][b + j-(L-1)] g[a][b] / (||f||_F ||g||_F),
//   i in [0, 2F-1), j in [0, 2L-1)
//
// computed as a zero-padded 2-D FFT product at (P1 x P2), P1 >= 2F-1, P2 = 2R >= 2L-1
// (the reference pads to exactly (2F-1) x (2L-1) = 511 x 8191, 8191 being prime):
//   forward, per tile : rows  R2C of length P2 (one 256-thread workgroup per row,
//                              R = P2/2 complex points through LDS)
//                       cols  complex FFT of length P1 for 16 adjacent columns,
//                              scaled by 1/||tile||_F
//   per tile pair     : cols  A conj(B), inverse FFT of length P1
//                       rows  C2R of length P2, fftshift index map, optional full
//                              output, block arg-max (first maximum in row-major
//                              order, like numpy.argmax)
// Built from the 256-thread Stockham stages of fft_device.h (table twiddles).
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

namespace {

constexpr int NT = 256;
#ifndef XCX_C2_WAVES
#define XCX_C2_WAVES 3      // waves per SIMD of the pair column pass (register budget 168; 4 -> 128 spills 60 registers)
#endif
// tile block (a / kGA, b / kGB) of a tile group: 8 x 8 tiles = up to 64 pairs on 16 slabs, 0.25 slab reads per pair cross the
// fabric (4 x 4 until round 5: 0.5; 8 x 8 with passes of two groups +2.7 %, 4 x 8 / 8 x 4 +1 %, notes section 10)
constexpr int kGA = 8, kGB = 8;
constexpr int kGS = kGA * kGB;  // pair slots per tile group of the pair pass
constexpr int CB = 16;   // columns per workgroup: 32 (1 per CU) -28 %; 8 (64-byte segments; round 5: 90 registers instead of 144,
                         // both halves of a 128-byte line on one XCD back to back) -6 % at 4 waves per SIMD, -10 % at 5, -13 % at 3

// LDS index padding of the row kernels: one extra slot per 16 keeps the stride-16 /
// stride-256 stage writes off a single bank group
__device__ __forceinline__ int rp(int p) { return p + (p >> 4); }
constexpr int rp_c(int p) { return p + (p >> 4); }

struct G2 {
  int F, L;        // tile shape
  int P1, R;       // column transform length, complex row length (P2 = 2R)
  int pitch;       // complex elements per spectrum row (>= R+1)
};

// Layout of the pair pass's intermediate T2 (one [P1][pitch] block per pair): rows in groups of kT2RB, inside a group the
// 128-byte pieces (16 columns) of the group's rows lie together -- element (m1, k2) at
//   ((m1 / RB) (pitch / 16) + k2 / 16) (16 RB) + (m1 % RB) 16 + k2 % 16.
// RB = 1 is the plain row-major block.  A column workgroup owns 16 columns of every row: it writes RB x 128 contiguous
// bytes per row group instead of 128-byte pieces one row pitch (32.9 KB) apart; a row workgroup reads its row as
// 128-byte pieces RB x 128 bytes apart, the RB workgroups of a group (consecutive blockIdx.x) sharing each stretch.
#ifndef XCX_T2_RB
#define XCX_T2_RB 512          // >= every P1: one group = the whole column block, [k2 / 16][m1][k2 % 16]
#endif
constexpr int kT2RB = XCX_T2_RB;
static_assert((kT2RB & (kT2RB - 1)) == 0, "row groups are powers of two");
// rows per group for a column length P1 (a power of two >= 16): the groups tile the P1 rows exactly
__host__ __device__ constexpr int t2_rb(int P1) { return kT2RB < P1 ? kT2RB : P1; }
// the spectra S2 are read by column workgroups only: one group = the whole column block, [k2 / 16][k1][k2 % 16]
__host__ __device__ constexpr int s2_rb(int P1) { return P1; }
__device__ __forceinline__ unsigned t2_off(int m1, int k2, int pitch, int rb) {
  if (rb == 1) return (unsigned)m1 * (unsigned)pitch + (unsigned)k2;
  return __umul24((unsigned)(m1 / rb) * (unsigned)(pitch >> 4) + (unsigned)(k2 >> 4), 16u * (unsigned)rb) +      // both < 2^24
         (unsigned)(m1 % rb) * 16u + (unsigned)(k2 & 15);
}
// (additive over the butterfly strides: t2_off(j + t S, k2) = t2_off(j, k2) + t2_off(t S, 0) for j < S, S and rb powers of two)

// forward rows: grid (F, T)
template <class RX>
__global__ __launch_bounds__(NT) void k2d_rows_fwd(const float* __restrict__ tiles, G2 g, cf* __restrict__ S1,
                                                   double* __restrict__ rowss, const cf* __restrict__ tw_sub,
                                                   const cf* __restrict__ tw_unt) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  __shared__ double red[NT / 64];
  constexpr int R = RX::L;
  const int f = blockIdx.x, t = blockIdx.y;
  const float* __restrict__ x = tiles + ((size_t)t * g.F + f) * g.L;
  const int L = g.L;
  float ss = 0.f;                              // sum of squares of this thread's samples (norm2d)
  auto in = [&ss, x, L](int m, int) -> cf {
    const int i0 = 2 * m, i1 = i0 + 1;
    const cf v = make_float2(i0 < L ? x[i0] : 0.f, i1 < L ? x[i1] : 0.f);
    ss = fmaf(v.x, v.x, fmaf(v.y, v.y, ss));
    return v;
  };
  auto to_lds = [=](int p, int, cf v) { lds[rp(p)] = v; };
  auto lds_ld = [=](int p, int) -> cf { return lds[rp(p)]; };
  // one base twiddle per butterfly, loaded up front; its powers are formed by multiplication
  typename StageGeom<R, RX::R1, RX::R0, 1, NT>::Base b1;
  typename StageGeom<R, RX::R2, RX::R0 * RX::R1, 1, NT>::Base b2;
  stage_tw_base<R, RX::R1, RX::R0, 1, NT>(tw_sub, b1);
  if constexpr (RX::S >= 3) stage_tw_base<R, RX::R2, RX::R0 * RX::R1, 1, NT>(tw_sub, b2);
  typename StageGeom<R, RX::R0, 1, 1, NT>::Base b0;
  fft_stage_c<R, RX::R0, 1, 1, NT, false, false, false>(b0, in, to_lds);
  {   // row total in double; the column pass adds the rows of a tile
    double acc = ss;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  }
  __syncthreads();
  fft_stage_c<R, RX::R1, RX::R0, 1, NT, false, true, false>(b1, lds_ld, to_lds);
  if constexpr (RX::S >= 3) {
    __syncthreads();
    fft_stage_c<R, RX::R2, RX::R0 * RX::R1, 1, NT, false, true, false>(b2, lds_ld, to_lds);
  }
  __syncthreads();
  cf* __restrict__ dst = S1 + ((size_t)t * g.F + f) * g.pitch;
  for (int k = threadIdx.x; k < R; k += NT) {
    const cf u = lds[rp(k)];
    const cf v = cconj(lds[rp(k == 0 ? 0 : R - k)]);
    const cf sm = cadd(u, v), df = csub(u, v);
    const cf e = cmul(tw_unt[k], df);
    dst[k] = make_float2(0.5f * (sm.x + e.y), 0.5f * (sm.y - e.x));
  }
  if (threadIdx.x == 0) {
    dst[R] = make_float2(lds[0].x - lds[0].y, 0.f);     // Nyquist bin (real)
    double tot = 0.0;
    for (int w = 0; w < NT / 64; ++w) tot += red[w];
    rowss[(size_t)t * g.F + f] = tot;
  }
}

// three-stage column configurations whose stages all fill whole 256-thread iterations take the register path of
// the pair pass (k2d_cols<INV>); LDS floats of its half-size exchange buffer
template <class RX, int NTC = NT>
constexpr bool kColsRegs = RX::S == 3 && (RX::L / RX::R0) * CB % NTC == 0 && (RX::L / RX::R1) * CB % NTC == 0 &&
                           (RX::L / RX::R2) * CB % NTC == 0;
constexpr size_t cols_regs_lds_bytes(int P1) { return sizeof(float) * (size_t)(P1 + (P1 >> 3) + 1) * CB; }

// columns: grid (ceil((R+1)/16), T or pairs).  INV = false: forward transform of
// the F rows of S1 (zero rows beyond), scaled; INV = true: A conj(B), inverse.
template <class RX, bool INV, int NTC = NT>
__global__ __launch_bounds__(NTC) __attribute__((amdgpu_waves_per_eu(INV ? XCX_C2_WAVES : 1, INV ? XCX_C2_WAVES : 8))) void k2d_cols(const cf* __restrict__ A, const cf* __restrict__ B2, G2 g,
                                               const double* __restrict__ rowss, int normalize,
                                               const int2* __restrict__ pairs,
                                               cf* __restrict__ out, const cf* __restrict__ tw_sub,
                                               const int* __restrict__ slots = nullptr, int slot_ncb = 0) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  constexpr int P1 = RX::L;
  int col0 = blockIdx.x * CB;
  int q = blockIdx.y;
  const int ncol = g.R + 1;
  const int pitch = g.pitch;
  if constexpr (INV) {
    // XCD-aware order of the pair pass (1-D grid; blockIdx % 8 labels the XCD under round-robin dispatch, used
    // for locality only): XCD x takes the column blocks cb = x (mod 8); inside an XCD the kGS pair slots of one
    // tile group (a kGA x kGB biclique of tiles, pair_groups2d below) run back to back for one column block, so the
    // group's 16 tile slabs (64 KB each at P1 = 512) are fetched into that L2 once and re-read there by the other
    // pairs of the group: 0.25 instead of 2 slab reads per pair across the fabric.
    const int x = blockIdx.x & 7, sq = (int)(blockIdx.x >> 3);
    const int ncx = (slot_ncb + 7) >> 3;
    const int pg = sq % kGS, cbi = (sq / kGS) % ncx, grp = sq / (kGS * ncx);
    const int cb = cbi * 8 + x;
    if (cb >= slot_ncb) return;
    q = slots[grp * kGS + pg];
    if (q < 0) return;
    col0 = cb * CB;
  }
  if constexpr (!INV) {
    const cf* __restrict__ src = A + (size_t)q * g.F * pitch;
    cf* __restrict__ dst = out + (size_t)q * P1 * pitch;
    const int F = g.F;
    float sc = 1.f;
    if (normalize) {                       // 1/||tile||_F from the row sums (same order in every workgroup)
      __shared__ double red[NTC];
      red[threadIdx.x] = (int)threadIdx.x < F ? rowss[(size_t)q * F + threadIdx.x] : 0.0;   // F <= NTC
      __syncthreads();
      for (int o = NTC / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
      }
      sc = (float)(1.0 / sqrt(red[0]));
    }
    auto in = [=](int f, int c) -> cf {
      const int col = col0 + c;
      // unconditional load (clamped row; padding columns lie inside the row), then a select: a load inside the
      // conditional becomes a branch with its own s_waitcnt and serialises the slab's loads
      const cf v = src[(size_t)(f < F ? f : F - 1) * pitch + col];
      return (f < F && col < ncol) ? v : make_float2(0.f, 0.f);
    };
    auto o = [=](int k1, int c, cf v) {            // spectra S2 in the row-group layout too (read by the pair pass)
      const int col = col0 + c;
      if (col < ncol) dst[t2_off(k1, col, pitch, s2_rb(P1))] = cscale(v, sc);
    };
    subfft<RX, CB, CB, NTC, false, false>(lds, tw_sub, in, o);
  } else {
    // the pair through scalar loads (constant address space: q is workgroup-uniform, but the compiler will not prove
    // the list unclobbered and would fetch it with a vector load + v_readfirstlane in front of every spectrum load)
    typedef const __attribute__((address_space(4))) int cint;
    cint* prp = (cint*)(const int*)(pairs + __builtin_amdgcn_readfirstlane(q));
    const int2 pr = make_int2(prp[0], prp[1]);
    const cf* __restrict__ a = A + (size_t)pr.x * P1 * pitch;
    const cf* __restrict__ b = B2 + (size_t)pr.y * P1 * pitch;
    cf* __restrict__ dst = out + (size_t)q * P1 * pitch;
    auto o = [=](int m1, int c, cf v) {
      const int col = col0 + c;
      if (col < ncol) dst[t2_off(m1, col, pitch, t2_rb(P1))] = v;
    };
    if constexpr (kColsRegs<RX, NTC>) {
      // every load of the slab first (the pitch is a multiple of CB: columns beyond ncol are padding inside the
      // row, their products are never stored), then the products, then the transform with register-resident
      // stages that exchange through a HALF-SIZE LDS buffer (real parts, then imaginary parts: 37 KB instead of
      // 64 KB per workgroup at P1 = 512, four workgroups per CU instead of two)
      constexpr int R0 = RX::R0, R1 = RX::R1, R2 = RX::R2;
      using G0 = StageGeom<P1, R0, 1, CB, NTC>;
      using G1 = StageGeom<P1, R1, R0, CB, NTC>;
      using G2s = StageGeom<P1, R2, R0 * R1, CB, NTC>;
      // loads in batches of two butterflies per thread (2 x 2 x R0 loads in flight): a and b of ALL iterations at
      // once would hold 2 x IT x R0 complex registers (194 VGPRs at P1 = 512: two waves per SIMD)
      typename G0::Regs va;
      constexpr int HB = G0::IT >= 2 ? 2 : 1;
      // addresses = (uniform base per butterfly input t, in scalar registers) + (one 32-bit per-lane byte offset per
      // iteration): written as a[o2 + t * stride] every one of the 64 loads and 32 stores of a thread got a 64-bit
      // vector address of its own (97 v_lshl_add_u64, a register pair each, in a kernel that spills at 168 VGPRs)
      typedef const __attribute__((address_space(1))) char* gchar;
      typedef __attribute__((address_space(1))) char* gchar_w;
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      gchar ua[R0], ub2[R0];
#pragma unroll
      for (int t = 0; t < R0; ++t) {
        ua[t] = (gchar)(a + (size_t)t2_off(t * (P1 / R0), 0, pitch, s2_rb(P1)));
        ub2[t] = (gchar)(b + (size_t)t2_off(t * (P1 / R0), 0, pitch, s2_rb(P1)));
        asm volatile("" : "+s"(ua[t]), "+s"(ub2[t]));
      }
      auto ldg = [](gchar base, unsigned off) -> cf {
        const f32x2 v = *(const __attribute__((address_space(1))) f32x2*)(base + (size_t)off);
        return make_float2(v.x, v.y);
      };
#pragma unroll
      for (int h0 = 0; h0 < G0::IT; h0 += HB) {
        cf vb[HB][R0];
#pragma unroll
        for (int hh = 0; hh < HB; ++hh) {
          const int it = h0 + hh;
          const int bq = threadIdx.x + it * NTC;
          unsigned o2 = 8u * t2_off(bq / CB, col0 + bq % CB, pitch, s2_rb(P1));
          asm volatile("" : "+v"(o2));
#pragma unroll
          for (int t = 0; t < R0; ++t) {
            va[it][t] = ldg(ua[t], o2);
            vb[hh][t] = ldg(ub2[t], o2);
          }
        }
#pragma unroll
        for (int hh = 0; hh < HB; ++hh) {
          const int it = h0 + hh;
#pragma unroll
          for (int t = 0; t < R0; ++t) va[it][t] = cmulc(va[it][t], vb[hh][t]);
          Bfly<R0, true>::run(va[it]);
        }
      }
      float* __restrict__ ldsf = reinterpret_cast<float*>(lds);
      auto idx = [](int p, int c) { return (p + (p >> 3)) * CB + c; };       // one pad row every 8 positions
      typename G1::Base b1;                       // twiddle bases: loaded where their registers are free
      stage_tw_base<P1, R1, R0, CB, NTC>(tw_sub, b1);
      typename G1::Regs v1;
      stage_exchange<P1, R0, 1, R1, CB, NTC>(ldsf, idx, va, v1);
      stage_compute<P1, R1, R0, CB, NTC, true>(v1, b1);
      typename G2s::Base b2;
      stage_tw_base<P1, R2, R0 * R1, CB, NTC>(tw_sub, b2);
      typename G2s::Regs v2;
      stage_exchange<P1, R1, R0, R2, CB, NTC>(ldsf, idx, v1, v2);
      stage_compute<P1, R2, R0 * R1, CB, NTC, true>(v2, b2);
      {
        gchar_w ud[R2];
#pragma unroll
        for (int t = 0; t < R2; ++t) {
          ud[t] = (gchar_w)(dst + (size_t)t2_off(t * (R0 * R1), 0, pitch, t2_rb(P1)));
          asm volatile("" : "+s"(ud[t]));
        }
#pragma unroll
        for (int it = 0; it < G2s::IT; ++it) {
          const int bq = threadIdx.x + it * NTC;
          const int c = bq % CB, j = bq / CB;
          const int k = j % (R0 * R1);
          const int j0 = (j - k) * R2 + k;
          const int col = col0 + c;
          unsigned od = 8u * t2_off(j0, col, pitch, t2_rb(P1));
          asm volatile("" : "+v"(od));
          if (col < ncol) {
#pragma unroll
            for (int t = 0; t < R2; ++t)
              *(__attribute__((address_space(1))) f32x2*)(ud[t] + (size_t)od) = f32x2{v2[it][t].x, v2[it][t].y};
          }
        }
      }
    } else {
      auto in = [=](int k1, int c) -> cf {
        const size_t o2 = t2_off(k1, col0 + c, pitch, s2_rb(P1));          // inside the padded row
        return cmulc(a[o2], b[o2]);
      };
      subfft<RX, CB, CB, NTC, true, false>(lds, tw_sub, in, o);
    }
  }
}

// inverse rows + fftshift map + arg-max: grid (P1, pairs).  FULL = false (the timelapse case: only the peak is kept):
// the 32 conditional map stores per thread, their exec-mask bookkeeping and 64-bit addresses are not compiled in
// (the kernel is VALU-issue-bound: 1 285 vector and 200 scalar instructions per thread and row at R = 4096 -- 1 646 and
// 653 before the arg-max went value-first, see below).
template <class RX, bool FULL>
__global__ __launch_bounds__(NT) void k2d_rows_inv(const cf* __restrict__ T2, G2 g, float inv_n,
                                                   float* __restrict__ full, const int* __restrict__ oidx,
                                                   float* __restrict__ rowbest,
                                                   long long* __restrict__ rowidx, const cf* __restrict__ tw_sub,
                                                   const cf* __restrict__ tw_unt) {
  extern __shared__ __attribute__((aligned(16))) cf lds[];
  __shared__ float bv[NT / 64];
  __shared__ unsigned long long bi;
  constexpr int R = RX::L;
  const int m1 = blockIdx.x, q = blockIdx.y;
  const int P1 = g.P1, P2 = 2 * R;
  // the pair's block of T2 in row groups (t2_off): element (m1, k) = row_base + ((k >> 4) << (4 + rbl)) + (k & 15); the
  // loads of a thread (k = tid + it NT) are one per-lane offset plus a uniform stride
  const cf* __restrict__ src = T2 + (size_t)q * P1 * g.pitch;
  const int rbl = 31 - __builtin_clz((unsigned)t2_rb(P1));              // log2 of the rows per group (uniform)
  const unsigned row_base = (((unsigned)(m1 >> rbl) * (unsigned)(g.pitch >> 4)) << (4 + rbl)) +
                            (((unsigned)m1 & ((1u << rbl) - 1u)) << 4);
  const unsigned lane_off = row_base + ((threadIdx.x >> 4) << (4 + rbl)) + (threadIdx.x & 15u);
  const unsigned it_stride = (unsigned)(NT >> 4) << (4 + rbl);
  typename StageGeom<R, RX::R1, RX::R0, 1, NT>::Base b1;
  typename StageGeom<R, RX::R2, RX::R0 * RX::R1, 1, NT>::Base b2;
  stage_tw_base<R, RX::R1, RX::R0, 1, NT>(tw_sub, b1);
  if constexpr (RX::S >= 3) stage_tw_base<R, RX::R2, RX::R0 * RX::R1, 1, NT>(tw_sub, b2);
  {   // all loads of the row in flight at once, then the LDS writes
    constexpr int NL = (R + NT - 1) / NT;
    cf stage[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int k = threadIdx.x + it * NT;
      stage[it] = (R % NT == 0 || k < R) ? src[lane_off + (unsigned)it * it_stride] : make_float2(0.f, 0.f);
    }
    const cf ny = threadIdx.x == 0 ? src[row_base + ((unsigned)(R >> 4) << (4 + rbl))] : make_float2(0.f, 0.f);
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int k = threadIdx.x + it * NT;
      if (R % NT == 0 || k < R) lds[rp(k)] = stage[it];
    }
    if (threadIdx.x == 0) lds[rp(R)] = ny;
  }
  __syncthreads();
  const int F = g.F, L = g.L;
  const int OW = 2 * L - 1, OH = 2 * F - 1;
  int i = m1 + F - 1;                       // shifted row
  if (i >= P1) i -= P1;
  const bool row_ok = i < OH;
  auto in = [=](int k, int) -> cf {
    const cf h = lds[rp(k)];
    if (k == 0) { const float ny = lds[rp(R)].x; return make_float2(h.x + ny, h.x - ny); }
    const cf hb = cconj(lds[rp(R - k)]);
    const cf sm = cadd(h, hb), df = csub(h, hb);
    const cf e = cmulc(df, tw_unt[k]);
    return make_float2(sm.x - e.y, sm.y + e.x);
  };
  // The arg-max of the row in two steps (the per-output compare / tie-break / index bookkeeping was 280 compare-select
  // and 520 scalar mask instructions per thread, most of them behind branches): every thread keeps its outputs in
  // registers and a running maximum of the valid ones (one range test, one select, one v_max per output); the row
  // maximum goes through the workgroup by value alone; only the threads that hold it then look up the smallest
  // shifted column among their matching outputs.  Output u = 2m + h lands in column (u + L - 1) mod P2, which is one
  // of the 2L - 1 lags unless L <= u <= P2 - L.
  constexpr int RL = (RX::S == 2) ? RX::R1 : RX::R2;
  constexpr int NSL = (RX::S == 2) ? RX::R0 : RX::R0 * RX::R1;
  using GL = StageGeom<R, RL, NSL, 1, NT>;
  float keep[GL::IT][2 * RL];
  int m0[GL::IT];
#pragma unroll
  for (int it = 0; it < GL::IT; ++it) {
    m0[it] = 0;
#pragma unroll
    for (int t = 0; t < 2 * RL; ++t) keep[it][t] = __builtin_nanf("");      // never equal to the maximum
  }
  float best = -INFINITY;
  const unsigned span = row_ok ? (unsigned)(P2 - 2 * L) : 0xffffffffu;       // u - L <= span: outside the lags
  // the pass runs the pairs in tile-group order; the full map goes to the caller's pair index
  float* __restrict__ frow = (FULL && full && row_ok) ? full + ((size_t)oidx[q] * OH + i) * OW : nullptr;
  auto out_v = [&](int it, int j0, int, const cf (&v)[RL]) {
    m0[it] = j0;
#pragma unroll
    for (int t = 0; t < RL; ++t) {
      const unsigned u0 = (unsigned)(2 * (j0 + t * NSL) - L);
      keep[it][2 * t] = v[t].x; keep[it][2 * t + 1] = v[t].y;
      best = fmaxf(best, u0 > span ? v[t].x : -INFINITY);
      best = fmaxf(best, u0 + 1u > span ? v[t].y : -INFINITY);
      if constexpr (FULL) {
        if (frow) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int j = 2 * (j0 + t * NSL) + h + L - 1;             // shifted column
            if (j >= P2) j -= P2;
            if (j < OW) frow[j] = (h ? v[t].y : v[t].x) * inv_n;
          }
        }
      }
    }
  };
  // first stage reads the untangled spectrum from LDS (barrier before it is overwritten)
  auto lds_st = [=](int p, int, cf v) { lds[rp(p)] = v; };
  auto lds_ld = [=](int p, int) -> cf { return lds[rp(p)]; };
  typename StageGeom<R, RX::R0, 1, 1, NT>::Base b0;
  if constexpr (RX::S == 2) {
    fft_stage_c<R, RX::R0, 1, 1, NT, true, true, false>(b0, in, lds_st);
    __syncthreads();
    fft_stage_c<R, RX::R1, RX::R0, 1, NT, true, false, true>(b1, lds_ld, out_v);
  } else {
    fft_stage_c<R, RX::R0, 1, 1, NT, true, true, false>(b0, in, lds_st);
    __syncthreads();
    fft_stage_c<R, RX::R1, RX::R0, 1, NT, true, true, false>(b1, lds_ld, lds_st);
    __syncthreads();
    fft_stage_c<R, RX::R2, RX::R0 * RX::R1, 1, NT, true, false, true>(b2, lds_ld, out_v);
  }
  // row maximum: lanes by shuffles, the four waves through LDS
  float wmax = best;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
  if ((threadIdx.x & 63) == 0) bv[threadIdx.x >> 6] = wmax;
  if (threadIdx.x == 0) bi = 0x7fffffffffffffffULL;
  __syncthreads();
  float rowmax = bv[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) rowmax = fmaxf(rowmax, bv[w]);
  if (row_ok && best == rowmax) {              // this thread holds a maximum of the row: its first column
    int bj = 0x7fffffff;
#pragma unroll
    for (int it = 0; it < GL::IT; ++it)
#pragma unroll
      for (int t = 0; t < 2 * RL; ++t) {
        const int u = 2 * (m0[it] + (t >> 1) * NSL) + (t & 1);
        int j = u + L - 1;
        if (j >= P2) j -= P2;
        const bool hit = ((unsigned)(u - L) > span) & (keep[it][t] == rowmax);
        bj = min(bj, hit ? j : 0x7fffffff);
      }
    if (bj != 0x7fffffff) atomicMin(&bi, (unsigned long long)((long long)i * OW + bj));
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    rowbest[(size_t)q * P1 + m1] = rowmax * inv_n;       // inv_n > 0: the maximum of the scaled values
    rowidx[(size_t)q * P1 + m1] = (long long)bi;
  }
}

// final reduction over the P1 rows of a pair
__global__ __launch_bounds__(NT) void k2d_peak(const float* __restrict__ rowbest, const long long* __restrict__ rowidx,
                                               int P1, int OW, float* __restrict__ peak_val, int* __restrict__ peak_idx) {
  __shared__ float bv[NT];
  __shared__ long long bi[NT];
  const int q = blockIdx.x;
  float best = -INFINITY;
  long long besti = 0x7fffffffffffffffLL;
  for (int r = threadIdx.x; r < P1; r += NT) {
    const float v = rowbest[(size_t)q * P1 + r];
    const long long ix = rowidx[(size_t)q * P1 + r];
    if (v > best || (v == best && ix < besti)) { best = v; besti = ix; }
  }
  bv[threadIdx.x] = best; bi[threadIdx.x] = besti;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const float v2 = bv[threadIdx.x + o]; const long long i2 = bi[threadIdx.x + o];
      if (v2 > bv[threadIdx.x] || (v2 == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v2; bi[threadIdx.x] = i2; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    peak_val[q] = bv[0];
    peak_idx[2 * q] = (int)(bi[0] / OW);
    peak_idx[2 * q + 1] = (int)(bi[0] % OW);
  }
}

void fill_tw(std::vector<float2>& v, int count, int L) {
  v.resize(count);
  for (int q = 0; q < count; ++q) {
    const double th = -2.0 * M_PI * (double)(q % L) / (double)L;
    v[q] = make_float2((float)cos(th), (float)sin(th));
  }
}

template <class K> hipError_t big_lds(K kernel, size_t bytes) {
  if (bytes <= 32768) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// Pair order of the pair pass: pairs are grouped by the kGA x kGB tile block (a / kGA, b / kGB) they fall in -- at most kGS
// pairs on at most kGA + kGB tiles per group -- and run in that order (k2d_cols<INV>); `orig` maps back to the caller's index.
struct PairOrder {
  std::vector<int2> sorted;        // pairs in run order
  std::vector<int> orig;           // run order -> caller's pair index
  std::vector<int> gfirst;         // first run index of every group, plus the end
  const int2* dsorted = nullptr;   // device copies
  const int* dorig = nullptr;
  int* dslots = nullptr;           // [groups][kGS]: pass-local pair index or -1
};

void pair_groups2d(const int* pairs, int P, int T, PairOrder& o) {
  const long nb = (T + kGB - 1) / kGB;
  static const bool grouped = !(getenv("XCX_CC2D_GROUPS") && atoi(getenv("XCX_CC2D_GROUPS")) == 0);   // 0: caller's order (A/B)
  std::vector<std::pair<long, int>> key(P);
  for (int i = 0; i < P; ++i) key[i] = {grouped ? (long)(pairs[2 * i] / kGA) * nb + pairs[2 * i + 1] / kGB : (long)(i / kGS), i};
  std::stable_sort(key.begin(), key.end(), [](const std::pair<long, int>& a, const std::pair<long, int>& b) { return a.first < b.first; });
  o.sorted.resize(P); o.orig.resize(P); o.gfirst.clear();
  for (int r = 0; r < P; ++r) {
    const int i = key[r].second;
    o.sorted[r] = make_int2(pairs[2 * i], pairs[2 * i + 1]);
    o.orig[r] = i;
    if (r == 0 || key[r].first != key[r - 1].first || r - o.gfirst.back() == kGS) o.gfirst.push_back(r);
  }
  o.gfirst.push_back(P);
}

// everything that depends on the row / column radices
template <class RXR, class RXC>
int run2d(xcx_ctx* ctx, const float* dtiles, int T, const G2& g, const PairOrder& ord, int P, int normalize,
          float* dpeak_val, int* dpeak_idx, float* dfull, const cf* tw_r, const cf* tw_u, const cf* tw_c) {
  constexpr int R = RXR::L, P1 = RXC::L;
  DevBuf &S1 = ctx->scratch[1], &S2 = ctx->scratch[2], &T2 = ctx->scratch[4], &misc = ctx->scratch[5];
  int rc;
  const size_t rowb = sizeof(cf) * (size_t)g.pitch;
  if ((rc = S1.ensure(ctx, rowb * g.F * T))) return rc;
  if ((rc = S2.ensure(ctx, rowb * P1 * T))) return rc;
  // pairs per pass: 2.2 GB of intermediate T2 (two tile groups of 64 pairs at 512 x 8192; 1 GiB = three groups of 16 until
  // round 5, 512 MiB until round 4) (larger passes amortise the launch tails; keeping T2 inside the 256 MiB Infinity
  // Cache measured 5-10 % slower, not faster, in rounds 3, 4 and 5)
  // (whole tile groups per pass: at least one group of kGS pairs)
  static const size_t pass_bytes = getenv("XCX_CC2D_PASS_MB") ? (size_t)std::max(1, atoi(getenv("XCX_CC2D_PASS_MB"))) << 20
                                                                : (size_t)2200 << 20;      // (tests force several passes)
  const int PB = (int)std::max<size_t>(kGS, std::min<size_t>((size_t)P, pass_bytes / (rowb * P1)));
  // Two passes in flight (round 5): the column kernel of pass i + 1 runs on one pipeline stream beside the row
  // kernel of pass i on the other, each pass in its own half of T2 (2 x pass_bytes).  The column kernel is latency-
  // bound (a wave lives 45 k cycles and issues vector instructions for 5.5 k of them, three waves per SIMD: 36 % of
  // the vector issue slots, 3.2 TB/s of HBM traffic), the row kernel issue-bound (four waves per SIMD keep the
  // vector unit busy, 3.6 TB/s); side by side: +3 ... +8 % (76.5 -> 79.0 k and 74.7 -> 81.1 k tile-pairs/s on two
  // boxes, profiles/r05_session3_notes.txt) -- less than the sum of the idle slots, because three column waves of
  // 144 registers leave a SIMD no room for a row wave of 96, so the kernels mix only where a CU happens to drain.
  // XCX_CC2D_OVERLAP=0: one pass at a time on the context's stream (A/B, and what a single pass does anyway).
  static const bool overlap = !(getenv("XCX_CC2D_OVERLAP") && atoi(getenv("XCX_CC2D_OVERLAP")) == 0);
  const bool two = overlap && P > PB && ctx->pipe[0] && ctx->pipe[1];
  const size_t t2_half = rowb * P1 * PB;
  if ((rc = T2.ensure(ctx, t2_half * (two ? 2 : 1)))) return rc;
  if ((rc = misc.ensure(ctx, sizeof(double) * (size_t)T * g.F + (sizeof(float) + sizeof(long long)) * (size_t)P1 * PB + 64)))
    return rc;
  double* rowss = (double*)misc.ptr;
  long long* rowidx = (long long*)(rowss + (size_t)T * g.F);
  float* rowbest = (float*)(rowidx + (size_t)P1 * PB);
  hipStream_t s = ctx->stream;
  const size_t lds_r = sizeof(cf) * (rp_c(R) + 1);
  HIP_TRY(big_lds(k2d_rows_fwd<RXR>, lds_r));
  HIP_TRY(big_lds((k2d_rows_inv<RXR, true>), lds_r));
  HIP_TRY(big_lds((k2d_rows_inv<RXR, false>), lds_r));
  hipLaunchKernelGGL(k2d_rows_fwd<RXR>, dim3(g.F, T), dim3(NT), lds_r, s, dtiles, g, (cf*)S1.ptr, rowss, tw_r, tw_u);
  const size_t lds_c = (RXC::S > 1) ? sizeof(cf) * P1 * CB : 0;
  // threads of a pair-pass column workgroup (template parameter NTC of k2d_cols): 512 (two butterflies per thread, 79-90
  // registers, one load batch, 16-24 waves per CU) measured equal at 6 waves per SIMD and -3.5 % at 4 (round 5)
  constexpr int NTP = NT;
  const size_t lds_ci = kColsRegs<RXC, NTP> ? cols_regs_lds_bytes(P1) : lds_c;      // pair pass: half-size exchange buffer
  HIP_TRY(big_lds((k2d_cols<RXC, false>), lds_c));
  HIP_TRY(big_lds((k2d_cols<RXC, true, NTP>), lds_ci));
  const int ncb = (g.R + 1 + CB - 1) / CB;
  hipLaunchKernelGGL((k2d_cols<RXC, false>), dim3(ncb, T), dim3(NT), lds_c, s, (const cf*)S1.ptr, (const cf*)nullptr, g,
                     (const double*)rowss, normalize, (const int2*)nullptr, (cf*)S2.ptr, tw_c);
  HIP_TRY(hipGetLastError());
  const float inv_n = 1.0f / ((float)P1 * (float)(2 * R));
  const size_t OW = 2 * (size_t)g.L - 1;
  const int ngroups = (int)ord.gfirst.size() - 1;
  const int ncx = (ncb + 7) / 8;
  // slot tables of ALL passes, uploaded once (no host synchronisation inside the loop: the passes queue back to back)
  std::vector<int> slots((size_t)ngroups * kGS, -1);
  std::vector<int> pass_g0;
  for (int g0 = 0; g0 < ngroups;) {
    // a pass = consecutive tile groups while their pairs fit the intermediate
    int g1 = g0 + 1;
    while (g1 < ngroups && ord.gfirst[g1 + 1] - ord.gfirst[g0] <= PB) ++g1;
    pass_g0.push_back(g0);
    const int p0 = ord.gfirst[g0];
    for (int gi = g0; gi < g1; ++gi)
      for (int r = ord.gfirst[gi]; r < ord.gfirst[gi + 1]; ++r) slots[(size_t)gi * kGS + (r - ord.gfirst[gi])] = r - p0;
    g0 = g1;
  }
  pass_g0.push_back(ngroups);
  HIP_TRY(hipMemcpyAsync(ord.dslots, slots.data(), sizeof(int) * slots.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));                         // `slots` goes out of scope with this function
  struct Ev {
    hipEvent_t e = nullptr;
    ~Ev() { if (e) (void)hipEventDestroy(e); }
    hipError_t make() { return hipEventCreateWithFlags(&e, hipEventDisableTiming); }
  };
  Ev start, done, coldone[2], rowdone[2];
  hipStream_t sc = s, sr = s;
  if (two) {
    sc = ctx->pipe[0]; sr = ctx->pipe[1];
    HIP_TRY(start.make()); HIP_TRY(done.make());
    for (int h = 0; h < 2; ++h) { HIP_TRY(coldone[h].make()); HIP_TRY(rowdone[h].make()); }
    HIP_TRY(hipEventRecord(start.e, s));                    // spectra S2, slot tables: ready on the context's stream
    HIP_TRY(hipStreamWaitEvent(sc, start.e, 0));
    HIP_TRY(hipStreamWaitEvent(sr, start.e, 0));
  }
  for (size_t ip = 0; ip + 1 < pass_g0.size(); ++ip) {
    const int g0 = pass_g0[ip], g1 = pass_g0[ip + 1];
    const int p0 = ord.gfirst[g0], cnt = ord.gfirst[g1] - p0;
    const int h = two ? (int)(ip & 1) : 0;
    cf* t2 = (cf*)((char*)T2.ptr + (size_t)h * t2_half);
    if (two && ip >= 2) HIP_TRY(hipStreamWaitEvent(sc, rowdone[h].e, 0));       // this half has been read
    hipLaunchKernelGGL((k2d_cols<RXC, true, NTP>), dim3(8u * ncx * (g1 - g0) * kGS), dim3(NTP), lds_ci, sc, (const cf*)S2.ptr,
                       (const cf*)S2.ptr, g, (const double*)nullptr, 0, ord.dsorted + p0, t2, tw_c,
                       (const int*)ord.dslots + (size_t)g0 * kGS, ncb);
    if (two) {
      HIP_TRY(hipEventRecord(coldone[h].e, sc));
      HIP_TRY(hipStreamWaitEvent(sr, coldone[h].e, 0));
    }
    // (rowbest / rowidx are shared by the passes: row kernel and peak kernel of every pass queue on ONE stream)
    if (dfull)
      hipLaunchKernelGGL((k2d_rows_inv<RXR, true>), dim3(P1, cnt), dim3(NT), lds_r, sr, (const cf*)t2, g, inv_n,
                         dfull, ord.dorig + p0, rowbest, rowidx, tw_r, tw_u);
    else
      hipLaunchKernelGGL((k2d_rows_inv<RXR, false>), dim3(P1, cnt), dim3(NT), lds_r, sr, (const cf*)t2, g, inv_n,
                         dfull, ord.dorig + p0, rowbest, rowidx, tw_r, tw_u);
    hipLaunchKernelGGL(k2d_peak, dim3(cnt), dim3(NT), 0, sr, rowbest, rowidx, P1, (int)OW, dpeak_val + p0,
                       dpeak_idx + 2 * (size_t)p0);
    HIP_TRY(hipGetLastError());
    if (two) HIP_TRY(hipEventRecord(rowdone[h].e, sr));
  }
  if (two) {
    // the row stream's last kernels wait for the column stream's last: its end is the end of both
    HIP_TRY(hipEventRecord(done.e, sr));
    HIP_TRY(hipStreamWaitEvent(s, done.e, 0));
  }
  return XCX_OK;
}

template <class RXR>
int run2d_cols(int P1, xcx_ctx* ctx, const float* dt, int T, const G2& g, const PairOrder& dp, int P, int nrm, float* pv,
               int* pi, float* full, const cf* tr, const cf* tu, const cf* tc) {
  switch (P1) {
    case 16: return run2d<RXR, Radices<4, 4>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
    case 32: return run2d<RXR, Radices<8, 4>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
    case 64: return run2d<RXR, Radices<8, 8>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
    case 128: return run2d<RXR, Radices<8, 4, 4>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
    case 256: return run2d<RXR, Radices<8, 8, 4>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
    default: return run2d<RXR, Radices<8, 8, 8>>(ctx, dt, T, g, dp, P, nrm, pv, pi, full, tr, tu, tc);
  }
}

}  // namespace

extern "C" int xcx_cc2d(xcx_ctx* ctx, const float* tiles, int T, int F, int L, const int* pairs, int P,
                        int normalize, float* peak_val, int* peak_idx, float* full, int on_device) {
  if (!ctx || !tiles || !pairs || !peak_val || !peak_idx) return set_err(XCX_EINVAL, "null argument");
  if (T < 1 || F < 1 || L < 1 || P < 1) return set_err(XCX_EINVAL, "T, F, L, P must be >= 1");
  for (int i = 0; i < P; ++i)
    if (pairs[2 * i] < 0 || pairs[2 * i] >= T || pairs[2 * i + 1] < 0 || pairs[2 * i + 1] >= T)
      return set_err(XCX_EINVAL, "pair %d outside [0,%d)", i, T);
  static const int kP1[] = {16, 32, 64, 128, 256, 512};
  static const int kR[] = {32, 64, 128, 256, 512, 1024, 2048, 4096};
  int P1 = 0, R = 0;
  for (int v : kP1) if (v >= 2 * F - 1) { P1 = v; break; }
  for (int v : kR) if (2 * v >= 2 * L - 1) { R = v; break; }
  static_assert(2 * XCX_CC2D_MAX_F == 512 && XCX_CC2D_MAX_L == 4096, "include/xcx.h: the padded sizes instantiated above");
  if (!P1 || !R || F > XCX_CC2D_MAX_F || L > XCX_CC2D_MAX_L)
    return set_err(XCX_EUNSUPPORTED, "tile %d x %d too large (max %ld x %ld)", F, L, XCX_CC2D_MAX_F, XCX_CC2D_MAX_L);
  HIP_TRY(hipSetDevice(ctx->device));
  G2 g{F, L, P1, R, (R + 1 + 15) / 16 * 16};
  std::vector<float2> tr, tu, tc;
  fill_tw(tr, R, R);
  fill_tw(tu, R, 2 * R);
  fill_tw(tc, P1, P1);
  int rc;
  DevBuf& tw = ctx->scratch[3];
  const size_t ntw = (size_t)2 * R + P1;
  const size_t OH = 2 * (size_t)F - 1, OW = 2 * (size_t)L - 1;
  const size_t tile_b = sizeof(float) * (size_t)T * F * L;
  PairOrder ord;
  pair_groups2d(pairs, P, T, ord);
  const size_t nslots = (size_t)(ord.gfirst.size() - 1) * kGS;          // upper bound of one pass
  size_t extra = sizeof(int2) * (size_t)P + sizeof(float) * P + sizeof(int) * 2 * (size_t)P + sizeof(int) * (size_t)P +
                 sizeof(int) * nslots + 5 * 64;
  if (!on_device) extra += tile_b + (full ? sizeof(float) * (size_t)P * OH * OW : 0);
  if ((rc = tw.ensure(ctx, sizeof(cf) * ntw + extra + 256))) return rc;
  cf* dtw = (cf*)tw.ptr;
  char* cur = (char*)(dtw + ntw);
  auto carve = [&](size_t bytes) { char* r = cur; cur += (bytes + 63) & ~(size_t)63; return r; };
  int2* dpairs = (int2*)carve(sizeof(int2) * P);
  float* dpv = (float*)carve(sizeof(float) * P);
  int* dpi = (int*)carve(sizeof(int) * 2 * P);
  int* dorig = (int*)carve(sizeof(int) * P);
  ord.dslots = (int*)carve(sizeof(int) * nslots);
  ord.dsorted = dpairs; ord.dorig = dorig;
  const float* dtiles = tiles;
  float* dfull = full;
  if (!on_device) {
    float* dt = (float*)carve(tile_b);
    HIP_TRY(hipMemcpyAsync(dt, tiles, tile_b, hipMemcpyHostToDevice, ctx->stream));
    dtiles = dt;
    if (full) dfull = (float*)carve(sizeof(float) * (size_t)P * OH * OW);
  }
  HIP_TRY(hipMemcpyAsync(dtw, tr.data(), sizeof(cf) * R, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dtw + R, tu.data(), sizeof(cf) * R, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dtw + 2 * R, tc.data(), sizeof(cf) * P1, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dpairs, ord.sorted.data(), sizeof(int2) * P, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(dorig, ord.orig.data(), sizeof(int) * P, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const cf *ptr_r = dtw, *ptr_u = dtw + R, *ptr_c = dtw + 2 * R;
  switch (R) {
    case 32: rc = run2d_cols<Radices<8, 4>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 64: rc = run2d_cols<Radices<8, 8>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 128: rc = run2d_cols<Radices<8, 4, 4>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 256: rc = run2d_cols<Radices<16, 16>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 512: rc = run2d_cols<Radices<8, 8, 8>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 1024: rc = run2d_cols<Radices<16, 8, 8>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    case 2048: rc = run2d_cols<Radices<16, 16, 8>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
    default: rc = run2d_cols<Radices<16, 16, 16>>(P1, ctx, dtiles, T, g, ord, P, normalize, dpv, dpi, dfull, ptr_r, ptr_u, ptr_c); break;
  }
  if (rc) return rc;
  // peaks come back in run order: scatter to the caller's pair index
  std::vector<float> hv(P);
  std::vector<int> hi(2 * (size_t)P);
  HIP_TRY(hipMemcpyAsync(hv.data(), dpv, sizeof(float) * P, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(hi.data(), dpi, sizeof(int) * 2 * P, hipMemcpyDeviceToHost, ctx->stream));
  if (!on_device && full)
    HIP_TRY(hipMemcpyAsync(full, dfull, sizeof(float) * (size_t)P * OH * OW, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int r = 0; r < P; ++r) {
    const int i = ord.orig[r];
    peak_val[i] = hv[r];
    peak_idx[2 * i] = hi[2 * (size_t)r];
    peak_idx[2 * i + 1] = hi[2 * (size_t)r + 1];
  }
  return XCX_OK;
}
