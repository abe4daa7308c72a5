// fft_device.h -- register/LDS FFT building blocks for gfx950 (wave64).
//
// A "sub-FFT" of compile-time length L = R0*R1*...  is computed for NB
// independent sequences at once by one 256-thread workgroup, Stockham
// autosort (natural order in, natural order out).  Data sits in LDS as
// [position][sequence] (sequence index fastest, pitch NBP float2) so that the
// 16 lanes that share a butterfly index touch 128 contiguous bytes: every
// ds_read_b64 / ds_write_b64 of every stage is bank-conflict free, and the
// global loads/stores of the first/last stage (fused into the stage through
// functors) are full 128-byte lines.
//
// Radix butterflies are fully unrolled in registers; composite radices
// (6, 8, 10, 12, 15, 16, ...) are built by compile-time Cooley-Tukey from the
// hand-written 2/3/4/5 kernels with constexpr twiddles.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

namespace xcx {

typedef float2 cf;

#define XCX_DEV __device__ __forceinline__

XCX_DEV cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
XCX_DEV cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
XCX_DEV cf cmul(cf a, cf b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
XCX_DEV cf cmulc(cf a, cf b) {
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
XCX_DEV cf cconj(cf a) { return make_float2(a.x, -a.y); }
XCX_DEV cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }
// multiply by +i / -i
XCX_DEV cf cmul_pi(cf a) { return make_float2(-a.y, a.x); }
XCX_DEV cf cmul_mi(cf a) { return make_float2(a.y, -a.x); }
// tables hold the forward factor e^{-i theta}; the inverse uses its conjugate
template <bool INV> XCX_DEV cf ctw(cf a, cf w) {
  return INV ? cmulc(a, w) : cmul(a, w);
}

// ---------------------------------------------------------------------------
// constexpr trigonometry for compile-time twiddles: cos/sin(2*pi*m/N)
// ---------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_small(double x) {  // |x| <= pi/4
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
constexpr double cx_cos_small(double x) {
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i < 14; ++i) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
struct cx_cs { double c, s; };
constexpr cx_cs cx_sincos2pi(long m, long N) {
  m %= N;
  if (m < 0) m += N;
  long oct = (8 * m) / N;        // 0..7
  long rem = 8 * m - oct * N;    // angle = (pi/4) * (oct + rem/N)
  long quarter = 0;
  double off = 0.0;
  if (oct & 1) {
    quarter = (oct + 1) / 2;
    off = -(kPi / 4) * double(N - rem) / double(N);
  } else {
    quarter = oct / 2;
    off = (kPi / 4) * double(rem) / double(N);
  }
  double c0 = cx_cos_small(off), s0 = cx_sin_small(off);
  switch (quarter & 3) {
    case 0: return {c0, s0};
    case 1: return {-s0, c0};
    case 2: return {-c0, -s0};
    default: return {s0, -c0};
  }
}

// multiply by the constant e^{-/+ 2 pi i M/N}  (- forward, + inverse)
template <int M, int N, bool INV> XCX_DEV cf mul_wconst(cf a) {
  constexpr int m = ((M % N) + N) % N;
  if constexpr (m == 0) {
    return a;
  } else if constexpr (4 * m == N) {
    return INV ? cmul_pi(a) : cmul_mi(a);
  } else if constexpr (2 * m == N) {
    return make_float2(-a.x, -a.y);
  } else if constexpr (4 * m == 3 * N) {
    return INV ? cmul_mi(a) : cmul_pi(a);
  } else {
    constexpr cx_cs cs = cx_sincos2pi(m, N);
    constexpr float wr = float(cs.c);
    constexpr float wi = INV ? float(cs.s) : float(-cs.s);
    return make_float2(a.x * wr - a.y * wi, a.x * wi + a.y * wr);
  }
}

// base + unsigned 32-bit element offset.  With a wave-uniform base the address is
// SGPR pair + zero-extended VGPR, which selects the saddr form of global_load/store
// (no per-lane 64-bit address arithmetic); a signed int index does not.
template <class T> XCX_DEV T* uoff(T* base, unsigned elem) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
  return reinterpret_cast<T*>(reinterpret_cast<B*>(base) + (size_t)(elem * (unsigned)sizeof(T)));
}

// compile-time loop
template <class F, int... Is>
XCX_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> XCX_DEV void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// thread index inside the cooperating group: the whole workgroup for the
// 256-thread engine, the lane for the one-wave engine (whose workgroups may
// pack several independent waves)
template <int NT> XCX_DEV int tid_of() {
  if constexpr (NT == 64) return (int)(threadIdx.x & 63u);
  else return (int)threadIdx.x;
}
// group-wide sync: a real barrier for multi-wave groups; for a single wave
// LDS instructions execute in program order, so only the compiler must be
// kept from reordering them
template <int NT> XCX_DEV void group_sync() {
  if constexpr (NT == 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// radix butterflies, in place on a register array v[R]
// ---------------------------------------------------------------------------
template <int R, bool INV> struct Bfly;

template <bool INV> struct Bfly<1, INV> {
  static XCX_DEV void run(cf (&v)[1]) {}
};

template <bool INV> struct Bfly<2, INV> {
  static XCX_DEV void run(cf (&v)[2]) {
    cf a = cadd(v[0], v[1]), b = csub(v[0], v[1]);
    v[0] = a; v[1] = b;
  }
};

template <bool INV> struct Bfly<3, INV> {
  static XCX_DEV void run(cf (&v)[3]) {
    constexpr float s = 0.86602540378443864676f;
    cf t = cadd(v[1], v[2]);
    cf d = csub(v[1], v[2]);
    cf m = make_float2(v[0].x - 0.5f * t.x, v[0].y - 0.5f * t.y);
    // forward: X1 = m - i s d, X2 = m + i s d
    cf r = INV ? make_float2(-s * d.y, s * d.x) : make_float2(s * d.y, -s * d.x);
    v[0] = cadd(v[0], t);
    v[1] = cadd(m, r);
    v[2] = csub(m, r);
  }
};

template <bool INV> struct Bfly<4, INV> {
  static XCX_DEV void run(cf (&v)[4]) {
    cf a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
    cf a2 = cadd(v[1], v[3]), a3 = csub(v[1], v[3]);
    cf r = INV ? cmul_pi(a3) : cmul_mi(a3);
    v[0] = cadd(a0, a2);
    v[2] = csub(a0, a2);
    v[1] = cadd(a1, r);
    v[3] = csub(a1, r);
  }
};

template <bool INV> struct Bfly<5, INV> {
  static XCX_DEV void run(cf (&v)[5]) {
    constexpr float c1 = 0.30901699437494742410f;   // cos(2pi/5)
    constexpr float c2 = -0.80901699437494742410f;  // cos(4pi/5)
    constexpr float s1 = 0.95105651629515357212f;   // sin(2pi/5)
    constexpr float s2 = 0.58778525229247312917f;   // sin(4pi/5)
    cf t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    cf t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    cf m1 = make_float2(v[0].x + c1 * t1.x + c2 * t2.x,
                        v[0].y + c1 * t1.y + c2 * t2.y);
    cf m2 = make_float2(v[0].x + c2 * t1.x + c1 * t2.x,
                        v[0].y + c2 * t1.y + c1 * t2.y);
    cf n1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    cf n2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    // forward: X1 = m1 - i n1, X4 = m1 + i n1, X2 = m2 - i n2, X3 = m2 + i n2
    cf r1 = INV ? cmul_pi(n1) : cmul_mi(n1);
    cf r2 = INV ? cmul_pi(n2) : cmul_mi(n2);
    v[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
    v[1] = cadd(m1, r1);
    v[4] = csub(m1, r1);
    v[2] = cadd(m2, r2);
    v[3] = csub(m2, r2);
  }
};

// compile-time Cooley-Tukey: R = R1*R2,
//   X[k1 + R1*k2] = sum_n2 w_R^{n2 k1} (sum_n1 x[n1*R2+n2] w_R1^{n1 k1}) w_R2^{n2 k2}
template <int R1, int R2, bool INV> struct BflyCT {
  static XCX_DEV void run(cf (&v)[R1 * R2]) {
    cf a[R2][R1];
    static_for<R2>([&](auto n2c) {
      constexpr int n2 = decltype(n2c)::value;
      cf t[R1];
      static_for<R1>([&](auto n1c) {
        constexpr int n1 = decltype(n1c)::value;
        t[n1] = v[n1 * R2 + n2];
      });
      Bfly<R1, INV>::run(t);
      static_for<R1>([&](auto k1c) {
        constexpr int k1 = decltype(k1c)::value;
        a[n2][k1] = mul_wconst<n2 * k1, R1 * R2, INV>(t[k1]);
      });
    });
    static_for<R1>([&](auto k1c) {
      constexpr int k1 = decltype(k1c)::value;
      cf t[R2];
      static_for<R2>([&](auto n2c) {
        constexpr int n2 = decltype(n2c)::value;
        t[n2] = a[n2][k1];
      });
      Bfly<R2, INV>::run(t);
      static_for<R2>([&](auto k2c) {
        constexpr int k2 = decltype(k2c)::value;
        v[k1 + R1 * k2] = t[k2];
      });
    });
  }
};

template <bool INV> struct Bfly<6, INV> : BflyCT<2, 3, INV> {};
template <bool INV> struct Bfly<8, INV> : BflyCT<2, 4, INV> {};
template <bool INV> struct Bfly<9, INV> : BflyCT<3, 3, INV> {};
template <bool INV> struct Bfly<10, INV> : BflyCT<2, 5, INV> {};
template <bool INV> struct Bfly<12, INV> : BflyCT<3, 4, INV> {};
template <bool INV> struct Bfly<15, INV> : BflyCT<3, 5, INV> {};
template <bool INV> struct Bfly<16, INV> : BflyCT<4, 4, INV> {};
template <bool INV> struct Bfly<24, INV> : BflyCT<3, 8, INV> {};

// ---------------------------------------------------------------------------
// sub-FFT configuration: length and up to four radices (1 = unused)
// ---------------------------------------------------------------------------
template <int R0_, int R1_ = 1, int R2_ = 1, int R3_ = 1> struct Radices {
  static constexpr int R0 = R0_, R1 = R1_, R2 = R2_, R3 = R3_;
  static constexpr int L = R0_ * R1_ * R2_ * R3_;
  static constexpr int S = 1 + (R1_ > 1) + (R2_ > 1) + (R3_ > 1);
  static_assert(R1_ > 1 || (R2_ == 1 && R3_ == 1), "radices must be packed");
  static_assert(R2_ > 1 || R3_ == 1, "radices must be packed");
};

// ---------------------------------------------------------------------------
// one Stockham stage over NB sequences.
//   butterfly b = tid + it*NT  ->  sequence c = b % NB, index j = b / NB
//   inputs  j + t*L/R (t = 0..R-1), twiddle w_{NS*R}^{k t}, k = j % NS
//   outputs (j-k)*R + k + t*NS
// `ld(pos, c)` / `st(pos, c, value)` are functors (global memory, LDS, or
// fused pre/post-processing).  SYNC inserts the barrier that separates the
// read phase from the write phase when both go through the same LDS buffer.
// `tw` is the length-L forward table e^{-2 pi i q / L}.
// ---------------------------------------------------------------------------
template <int L, int R, int NS, int NB, int NT, bool INV, bool SYNC,
          class LoadF, class StoreF>
XCX_DEV void fft_stage(const cf* __restrict__ tw, LoadF ld, StoreF st) {
  constexpr int NBF = NB * (L / R);
  constexpr int IT = (NBF + NT - 1) / NT;
  constexpr bool GUARD = (NBF % NT) != 0;
  const int tid = threadIdx.x;
  cf v[IT][R];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int b = tid + it * NT;
    if (!GUARD || b < NBF) {
      const int c = b % NB;
      const int j = b / NB;
#pragma unroll
      for (int t = 0; t < R; ++t) v[it][t] = ld(j + t * (L / R), c);
      if constexpr (NS > 1) {
        const int k = j % NS;
        constexpr int STEP = L / (NS * R);
#pragma unroll
        for (int t = 1; t < R; ++t) {
          const cf w = tw[k * t * STEP];
          v[it][t] = ctw<INV>(v[it][t], w);
        }
      }
      Bfly<R, INV>::run(v[it]);
    }
  }
  if constexpr (SYNC) __syncthreads();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int b = tid + it * NT;
    if (!GUARD || b < NBF) {
      const int c = b % NB;
      const int j = b / NB;
      const int k = (NS > 1) ? (j % NS) : 0;
      const int j0 = (j - k) * R + k;
#pragma unroll
      for (int t = 0; t < R; ++t) st(j0 + t * NS, c, v[it][t]);
    }
  }
}

// ---------------------------------------------------------------------------
// Chain-twiddle variant used by the one-wave kernels: instead of R-1 table
// loads per butterfly, ONE base factor w = W_{NS*R}^{k} per (lane, iteration)
// is loaded up front (stage_tw_base, issued at kernel start in the shadow of
// the data loads) and its powers are formed by multiplication, re-seeded at
// w^8 and w^16 from repeated squaring so that no factor is more than ~10
// roundings away from the table value.
// VEC_ST: `st(j0, c, v)` receives the whole butterfly output (positions
// j0 + t*NS) so that the caller can run its own recurrences over t.
// ---------------------------------------------------------------------------
template <int L, int R, int NS, int NB, int NT> struct StageGeom {
  static constexpr int NBF = NB * (L / R);
  static constexpr int IT = (NBF + NT - 1) / NT;
  static constexpr bool GUARD = (NBF % NT) != 0;
  static constexpr int STEP = (NS > 1) ? L / (NS * R) : 0;
  typedef cf Regs[IT][R];
  typedef cf Base[IT];
};

// `tid`: the lane's butterfly slot of this stage (the one-wave column kernels rotate the slots of
// their last stage so that the T stores start on tile boundaries)
template <int L, int R, int NS, int NB, int NT>
XCX_DEV void stage_tw_base(const cf* __restrict__ tw,
                           cf (&base)[StageGeom<L, R, NS, NB, NT>::IT], int tid = tid_of<NT>()) {
  using G = StageGeom<L, R, NS, NB, NT>;
#pragma unroll
  for (int it = 0; it < G::IT; ++it) {
    const int b = tid + it * NT;
    const int j = (G::GUARD && b >= G::NBF) ? 0 : b / NB;
    base[it] = *uoff(tw, (unsigned)((j % NS) * G::STEP));
  }
}

XCX_DEV cf csq(cf a) { return make_float2(a.x * a.x - a.y * a.y, 2.f * a.x * a.y); }

// v[t] *= w^t  (conjugated when INV), t = 1..R-1
template <int R, bool INV> XCX_DEV void apply_tw_chain(cf (&v)[R], cf w1) {
  cf w8 = w1, wt = w1;
  if constexpr (R > 8) w8 = csq(csq(csq(w1)));
  static_for<R - 1>([&](auto tc) {
    constexpr int t = decltype(tc)::value + 1;
    if constexpr (t == 1) wt = w1;
    else if constexpr (t == 8) wt = w8;
    else if constexpr (t == 16) wt = csq(w8);
    else if constexpr (t == 24) wt = cmul(csq(w8), w8);
    else wt = cmul(wt, w1);
    v[t] = ctw<INV>(v[t], wt);
  });
}

// ... and the same with every output scaled by `sc` at the price of six multiplies per butterfly: the chain's seeds
// (t = 1, 8) carry the factor, v[0] takes it directly (the q24 row waves fold their fixed-point scale in here instead of
// multiplying each of their 70 unpacked numbers)
template <int R, bool INV> XCX_DEV void apply_tw_chain_scaled(cf (&v)[R], cf w1, float sc) {
  static_assert(R <= 16, "seeds at t = 1 and t = 8 only");
  cf w8 = w1, wt = w1;
  if constexpr (R > 8) w8 = cscale(csq(csq(csq(w1))), sc);
  v[0] = cscale(v[0], sc);
  static_for<R - 1>([&](auto tc) {
    constexpr int t = decltype(tc)::value + 1;
    if constexpr (t == 1) wt = cscale(w1, sc);
    else if constexpr (t == 8) wt = w8;
    else wt = cmul(wt, w1);
    v[t] = ctw<INV>(v[t], wt);
  });
}

template <int L, int R, int NS, int NB, int NT, bool INV, bool SYNC, bool VEC_ST,
          class LoadF, class StoreF>
XCX_DEV void fft_stage_c(const cf (&base)[StageGeom<L, R, NS, NB, NT>::IT], LoadF ld, StoreF st) {
  using G = StageGeom<L, R, NS, NB, NT>;
  const int tid = tid_of<NT>();
  cf v[G::IT][R];
#pragma unroll
  for (int it = 0; it < G::IT; ++it) {
    const int b = tid + it * NT;
    if (!G::GUARD || b < G::NBF) {
      const int c = b % NB;
      const int j = b / NB;
#pragma unroll
      for (int t = 0; t < R; ++t) v[it][t] = ld(j + t * (L / R), c);
      if constexpr (NS > 1) apply_tw_chain<R, INV>(v[it], base[it]);
      Bfly<R, INV>::run(v[it]);
    }
  }
  if constexpr (SYNC) group_sync<NT>();
#pragma unroll
  for (int it = 0; it < G::IT; ++it) {
    const int b = tid + it * NT;
    if (!G::GUARD || b < G::NBF) {
      const int c = b % NB;
      const int j = b / NB;
      const int k = (NS > 1) ? (j % NS) : 0;
      const int j0 = (j - k) * R + k;
      if constexpr (VEC_ST) {
        st(it, j0, c, v[it]);
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) st(j0 + t * NS, c, v[it][t]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Register-to-register stage exchange through a HALF-SIZE LDS buffer: the real
// parts of all outputs of stage A are written (Stockham positions), read back
// as the inputs of stage B, then the same for the imaginary parts.  The LDS
// footprint of a wave is L*NB*4 bytes instead of L*NB*8, which doubles the
// number of resident single-wave workgroups per CU (the kernels are LDS-
// capacity limited, not LDS-bandwidth limited).  `idx(p, c)` maps (position,
// sequence) to a float index of the LDS buffer.
// ---------------------------------------------------------------------------
template <int L, int RA, int NSA, int RB, int NB, int NT, class Idx>
XCX_DEV void stage_exchange(float* __restrict__ ldsf, Idx idx,
                            const typename StageGeom<L, RA, NSA, NB, NT>::Regs& src,
                            typename StageGeom<L, RB, NSA * RA, NB, NT>::Regs& dst) {
  using GA = StageGeom<L, RA, NSA, NB, NT>;
  using GB = StageGeom<L, RB, NSA * RA, NB, NT>;
  const int tid = tid_of<NT>();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int it = 0; it < GA::IT; ++it) {
      const int b = tid + it * NT;
      if (!GA::GUARD || b < GA::NBF) {
        const int c = b % NB, j = b / NB;
        const int k = (NSA > 1) ? (j % NSA) : 0;
        const int j0 = (j - k) * RA + k;
#pragma unroll
        for (int t = 0; t < RA; ++t)
          ldsf[idx(j0 + t * NSA, c)] = half ? src[it][t].y : src[it][t].x;
      }
    }
    group_sync<NT>();
#pragma unroll
    for (int it = 0; it < GB::IT; ++it) {
      const int b = tid + it * NT;
      if (!GB::GUARD || b < GB::NBF) {
        const int c = b % NB, j = b / NB;
#pragma unroll
        for (int t = 0; t < RB; ++t) {
          const float f = ldsf[idx(j + t * (L / RB), c)];
          if (half) dst[it][t].y = f; else dst[it][t].x = f;
        }
      }
    }
    group_sync<NT>();
  }
}

// The same exchange for ONE sequence (NB = 1) in the column kernels' padded layout
// idx(p) = p + (p >> 3).  The pad makes idx non-linear, and the compiler then recomputes a
// shift + add per LDS access (~200 integer VALU instructions per 1536-point column).  Both
// access patterns step by multiples of 8 (or stay inside one aligned group of 8), where
// idx(p0 + d) = idx(p0) + d + d/8 holds exactly: one base per butterfly, every other address
// is a compile-time immediate of the ds instruction.
// SRC_PAIRS (first exchange only, NSA = 1): the source registers hold the butterflies
// j = 2 tid, 2 tid + 1 (iterations 0, 1) and tid + it NT (later iterations) -- the assignment of
// the 16-byte first-stage loads -- instead of tid + it NT.
template <int NT>
XCX_DEV int pair_src_index(int tid, int it) { return it < 2 ? 2 * tid + it : tid + it * NT; }

template <int L, int RA, int NSA, int RB, int NT, bool SRC_PAIRS = false>
XCX_DEV void stage_exchange_pad8(float* __restrict__ ldsf,
                                 const typename StageGeom<L, RA, NSA, 1, NT>::Regs& src,
                                 typename StageGeom<L, RB, NSA * RA, 1, NT>::Regs& dst,
                                 int tid_read = tid_of<NT>()) {
  using GA = StageGeom<L, RA, NSA, 1, NT>;
  using GB = StageGeom<L, RB, NSA * RA, 1, NT>;
  constexpr int RD = L / RB;
  constexpr bool WFAST = (NSA == 1) ? (RA % 8 == 0) : (NSA % 8 == 0);
  constexpr bool RFAST = (RD % 8 == 0);
  const int tid = tid_of<NT>();
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int it = 0; it < GA::IT; ++it) {
      const int j = SRC_PAIRS ? pair_src_index<NT>(tid, it) : tid + it * NT;
      if (!GA::GUARD || j < GA::NBF) {
        const int k = (NSA > 1) ? (j % NSA) : 0;
        const int j0 = (j - k) * RA + k;
        const int base = j0 + (j0 >> 3);
        static_for<RA>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int d = t * NSA;
          const float f = half ? src[it][t].y : src[it][t].x;
          if constexpr (WFAST) ldsf[base + d + (d >> 3)] = f;
          else ldsf[(j0 + d) + ((j0 + d) >> 3)] = f;
        });
      }
    }
    group_sync<NT>();
#pragma unroll
    for (int it = 0; it < GB::IT; ++it) {
      const int j = tid_read + it * NT;
      if (!GB::GUARD || j < GB::NBF) {
        const int base = j + (j >> 3);
        static_for<RB>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          constexpr int d = t * RD;
          float f;
          if constexpr (RFAST) f = ldsf[base + d + (d >> 3)];
          else f = ldsf[(j + d) + ((j + d) >> 3)];
          if (half) dst[it][t].y = f; else dst[it][t].x = f;
        });
      }
    }
    group_sync<NT>();
  }
}

// twiddle (chain from the per-iteration base factor) + butterfly, in registers
template <int L, int R, int NS, int NB, int NT, bool INV>
XCX_DEV void stage_compute(typename StageGeom<L, R, NS, NB, NT>::Regs& v,
                           const typename StageGeom<L, R, NS, NB, NT>::Base& base) {
  using G = StageGeom<L, R, NS, NB, NT>;
#pragma unroll
  for (int it = 0; it < G::IT; ++it) {
    if (!G::GUARD || (tid_of<NT>() + it * NT) < G::NBF) {
      if constexpr (NS > 1) apply_tw_chain<R, INV>(v[it], base[it]);
      Bfly<R, INV>::run(v[it]);
    }
  }
}
template <int L, int R, int NS, int NB, int NT, bool INV>
XCX_DEV void stage_compute_scaled(typename StageGeom<L, R, NS, NB, NT>::Regs& v,
                                  const typename StageGeom<L, R, NS, NB, NT>::Base& base, float sc) {
  using G = StageGeom<L, R, NS, NB, NT>;
  static_assert(NS > 1, "the scale rides on the twiddle chain");
#pragma unroll
  for (int it = 0; it < G::IT; ++it) {
    if (!G::GUARD || (tid_of<NT>() + it * NT) < G::NBF) {
      apply_tw_chain_scaled<R, INV>(v[it], base[it], sc);
      Bfly<R, INV>::run(v[it]);
    }
  }
}

// ---------------------------------------------------------------------------
// whole sub-FFT: first stage reads through `in`, last stage writes through
// `out`, intermediate stages exchange through `lds` ([L][NBP] float2).
// IN_LDS: `in` itself reads `lds` (barrier needed before stage 0 overwrites).
// On return the LDS buffer may still be being read by other threads only if
// S > 1: callers that reuse `lds` afterwards must __syncthreads() first.
// ---------------------------------------------------------------------------
template <class RX, int NB, int NBP, int NT, bool INV, bool IN_LDS, class In,
          class Out>
XCX_DEV void subfft(cf* __restrict__ lds, const cf* __restrict__ tw, In in,
                    Out out) {
  constexpr int L = RX::L;
  constexpr int R0 = RX::R0, R1 = RX::R1, R2 = RX::R2, R3 = RX::R3;
  auto lds_ld = [=](int p, int c) -> cf { return lds[p * NBP + c]; };
  auto lds_st = [=](int p, int c, cf v) { lds[p * NBP + c] = v; };
  if constexpr (RX::S == 1) {
    fft_stage<L, R0, 1, NB, NT, INV, false>(tw, in, out);
  } else {
    fft_stage<L, R0, 1, NB, NT, INV, IN_LDS>(tw, in, lds_st);
    __syncthreads();
    if constexpr (RX::S == 2) {
      fft_stage<L, R1, R0, NB, NT, INV, false>(tw, lds_ld, out);
    } else {
      fft_stage<L, R1, R0, NB, NT, INV, true>(tw, lds_ld, lds_st);
      __syncthreads();
      if constexpr (RX::S == 3) {
        fft_stage<L, R2, R0 * R1, NB, NT, INV, false>(tw, lds_ld, out);
      } else {
        fft_stage<L, R2, R0 * R1, NB, NT, INV, true>(tw, lds_ld, lds_st);
        __syncthreads();
        fft_stage<L, R3, R0 * R1 * R2, NB, NT, INV, false>(tw, lds_ld, out);
      }
    }
  }
}

// two-level table lookup of w_N^q (forward e^{-2 pi i q/N}), q in [0, N/2]:
// tab[0..QL) = w_N^{ql},  tab[QL + qh] = w_N^{QL*qh}
constexpr int kTwQL = 512;
XCX_DEV cf twN_lookup(const cf* __restrict__ tab, unsigned q) {
  const cf lo = *uoff(tab, q & (kTwQL - 1));
  const cf hi = *uoff(tab, kTwQL + (q >> 9));
  return cmul(lo, hi);
}
// the same for a WAVE-UNIFORM q (the caller guarantees it): two scalar loads through the constant address space
// instead of two vector loads that would fetch one value sixty-four times
XCX_DEV cf twN_lookup_uniform(const cf* __restrict__ tab, unsigned q) {
  typedef const __attribute__((address_space(4))) float cfloat;
  const unsigned qs = (unsigned)__builtin_amdgcn_readfirstlane((int)q);
  cfloat* lo = (cfloat*)(const float*)uoff(tab, qs & (kTwQL - 1));
  cfloat* hi = (cfloat*)(const float*)uoff(tab, kTwQL + (qs >> 9));
  return cmul(make_float2(lo[0], lo[1]), make_float2(hi[0], hi[1]));
}

}  // namespace xcx
