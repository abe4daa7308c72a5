// gram.hip -- time-stack as a batched Hermitian rank-W update (see wave_kernels.h)
#include "wave_kernels.h"

namespace xcx {

__global__ __launch_bounds__(kWNT * kWPB) void k_gram(GramArgs a) {
  const int lane = tid_of<kWNT>();
  const int x = blockIdx.x & 7;
  const int q = (blockIdx.x >> 3) * kWPB + wave_in_group();
  const int chunk = (q / a.ntiles) * 8 + x;
  const int tile = q % a.ntiles;
  const int k = chunk * 64 + lane;
  if (chunk * 64 >= a.nfreq) return;
  const bool live = k < a.nfreq;
  const int kk = live ? k : a.nfreq - 1;
  const int2 t0 = a.tiles[tile];
  cf acc[kGT][kGT];
#pragma unroll
  for (int i = 0; i < kGT; ++i)
#pragma unroll
    for (int j = 0; j < kGT; ++j) acc[i][j] = make_float2(0.f, 0.f);
  const int S = a.S, W = a.W;
  for (int w = 0; w < W; ++w) {
    cf ai[kGT], bj[kGT];
#pragma unroll
    for (int i = 0; i < kGT; ++i) {
      const int si = min(t0.x + i, S - 1), sj = min(t0.y + i, S - 1);
      ai[i] = a.X[((size_t)si * W + w) * a.spec_pitch + kk];
      bj[i] = a.X[((size_t)sj * W + w) * a.spec_pitch + kk];
    }
#pragma unroll
    for (int i = 0; i < kGT; ++i)
#pragma unroll
      for (int j = 0; j < kGT; ++j) {
        // acc += ai * conj(bj)
        acc[i][j].x += ai[i].x * bj[j].x + ai[i].y * bj[j].y;
        acc[i][j].y += ai[i].y * bj[j].x - ai[i].x * bj[j].y;
      }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < kGT; ++i)
#pragma unroll
    for (int j = 0; j < kGT; ++j) {
      const int si = t0.x + i, sj = t0.y + j;
      if (si < S && sj < S) {
        const int p = a.pidx[si * S + sj];
        if (p >= 0) a.C[(size_t)p * a.c_pitch + k] = acc[i][j];
      }
    }
}


hipError_t launch_gram(const GramArgs& a, hipStream_t s) {
  const int nchunks = (a.nfreq + 63) / 64;
  const int per_xcd = (nchunks + 7) / 8;
  const unsigned waves_per_xcd = (unsigned)per_xcd * (unsigned)a.ntiles;
  const dim3 grid(8u * ((waves_per_xcd + kWPB - 1) / kWPB));
  hipLaunchKernelGGL(k_gram, grid, dim3(kWNT * kWPB), 0, s, a);
  return hipGetLastError();
}

}  // namespace xcx
