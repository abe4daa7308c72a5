// cc_kernels.h -- definitions shared by the translation units of the cross-correlation engine (gfx950):
// plan flags, transform geometry, pair-window descriptors, the whitening gain.
//
// Transform length N = N1 * N2 (N2 even, N2h = N2/2), N >= 2n-1.
// Index split (prototype: tools/proto_fft.py):
//     time      m = m1 + N1*m2      m1 in [0,N1), m2 in [0,N2)
//     frequency k = k1*N2 + k2      k1 in [0,N1), k2 in [0,N2)
// The kernels live in wave_kernels.h (one 64-lane wave per sub-FFT).
//
// Reference semantics: xcorr/util/cc.py:60-79 and
// xcorr/signal/correlate.py:97-113 (see include/xcx.h).
#pragma once
#include "fft_device.h"

namespace xcx {

constexpr int kNT = 256;   // threads per workgroup of the statistics kernels

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
  int wrap;     // output positions o >= wrap belong to the negative lags (o - N): N - idx_min; N for a plan that keeps only
                // lags >= 0 at N = n (the mirrored short-operand case of the segmented plans)
  int flags;
  int band_hi;  // whitened plans: spectra are EXACTLY zero at folded bins min(k, N-k) > band_hi (the whitening
                // gain is 0 there), so the pair column pass never loads them; N = no band limit
};

XCX_DEV float whiten_gain(float f, float b0, float b1, float b2, float b3) {
  if (f >= b1 && f <= b2) return 1.f;
  if (f >= b0 && f < b1)
    return 0.5f * (1.f - cosf(3.14159265358979f * (f - b0) / (b1 - b0)));
  if (f > b2 && f <= b3)
    return 0.5f * (1.f + cosf(3.14159265358979f * (f - b2) / (b3 - b2)));
  return 0.f;
}

struct PwDesc { int slot_a, slot_b; long long out_off; };   // per pair-window: spectrum slots, output row offset

}  // namespace xcx
