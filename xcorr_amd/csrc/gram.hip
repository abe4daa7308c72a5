// gram.hip -- time-stack as a batched Hermitian rank-W update (see wave_kernels.h)
#include "wave_kernels.h"
#include <cstdlib>

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
  const int S = a.S, W = a.W, ldw = a.ldw ? a.ldw : a.W;
  for (int w = 0; w < W; ++w) {
    cf ai[kGT], bj[kGT];
#pragma unroll
    for (int i = 0; i < kGT; ++i) {
      const int si = min(t0.x + i, S - 1), sj = min(t0.y + i, S - 1);
      ai[i] = a.X[((size_t)si * ldw + w) * a.spec_pitch + kk];
      bj[i] = a.X[((size_t)sj * ldw + w) * a.spec_pitch + kk];
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
        if (p >= 0) {
          cf* dst = a.C + (size_t)p * a.c_pitch + k;
          *dst = a.accumulate ? cadd(*dst, acc[i][j]) : acc[i][j];
        }
      }
    }
}


// ---------------------------------------------------------------------------
// The same rank-W update as a dense complex GEMM per frequency on the matrix cores
// (v_mfma_f32_32x32x2_f32, exact f32): for S <= 64 stations the whole Gram matrix of one
// frequency, G = A A^H with A = [S x W] complex, is held in the accumulators of ONE wave:
//     Re G = [Ar Ai] [Ar Ai]^T,   Im G = [Ai -Ar] [Ar Ai]^T      (K = 2 per window)
// With the f32 MFMA operand maps (A: lane -> (row l&31, k l>>5), B: (k l>>5, col l&31)) the
// A and B operands of Re G are THE SAME register: lanes 0-31 hold the real parts of 32
// stations, lanes 32-63 the imaginary parts.  Blocks (0,0), (0,1), (1,1) of 32 x 32
// stations x (re, im) = 96 accumulator registers; the lower-left block is never formed.
// A workgroup = 4 waves = 4 adjacent frequencies: the spectra of all stations for a chunk of
// kGW windows are staged in LDS as 32-byte runs (each spectrum element is fetched from
// HBM/L2 exactly once -- the VALU kernel above re-reads every spectrum once per 8x8 tile),
// and the results leave through LDS as 32-byte runs per pair row.
// ---------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kGW = 16;                 // windows per LDS chunk
constexpr int kGPitch = 9;              // floats per (window, station): 4 freq x (re, im) + 1 pad (bank spread)

__global__ __launch_bounds__(256, 3) void k_gram_mfma(GramArgs a) {
  __shared__ __attribute__((aligned(16))) float ls[kGW * 64 * kGPitch];      // 36 KB; reused for the output
  const int tid = threadIdx.x, lane = tid & 63, v = tid >> 6;
  // workgroups are dealt round-robin to the 8 XCDs: give every XCD a contiguous range of
  // frequency quads so that the 4 workgroups sharing a 128-byte line share an L2
  const int nq = (a.nfreq + 3) / 4, per_xcd = (nq + 7) / 8;
  const int quad = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (quad >= nq || (blockIdx.x >> 3) >= per_xcd) return;
  const int f0 = quad * 4;
  if (a.g.band_hi < a.g.N) {
    // spectrum element e = k2 N1 + k1; N1 is a multiple of 4, so a quad lies in one column
    const int k2 = f0 / a.g.N1, k1 = f0 - k2 * a.g.N1;
    int za, zb;
    band_zero_stretch(a.g, k2, za, zb);
    if (zb >= za && k1 >= za && k1 + 3 <= zb) return;       // every spectrum is zero here, and so is G
  }
  const int S = a.S, W = a.W;
  const bool two = S > 32;
  f32x16 gr00 = {0}, gr01 = {0}, gr11 = {0}, gi00 = {0}, gi01 = {0}, gi11 = {0};
  const int ls_s = tid >> 2, ls_f = tid & 3;                 // loader role: station, frequency of the quad
  const bool ld_ok = ls_s < S && f0 + ls_f < a.nfreq;
  // the loader's address is clamped into the buffer and the value masked when it is stored: a load inside the bounds
  // branch gets a wait of its own, and the sixteen loads of a chunk went out one round trip after the other
  const cf* __restrict__ src = a.X + (size_t)min(ls_s, S - 1) * (a.ldw ? a.ldw : W) * a.spec_pitch + min(f0 + ls_f, a.nfreq - 1);
  const int part = lane >> 5, st = lane & 31;
  const float sgn = part ? -1.f : 1.f;
  // rows of C this thread will write (4 entries per 32x32 block), looked up before the main
  // loop so that the latency hides under it
  int prow[3][4];
#pragma unroll
  for (int blk = 0; blk < 3; ++blk)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      const int si = (blk == 2 ? 32 : 0) + (e >> 5), sj = (blk == 0 ? 0 : 32) + (e & 31);
      const int pv = a.pidx[min(si, S - 1) * S + min(sj, S - 1)];
      prow[blk][q] = ((si < S) & (sj < S)) ? pv : -1;
    }
  cf stage[kGW];
  auto fetch = [&](int w0) {                                // one chunk of windows -> registers
#pragma unroll
    for (int w = 0; w < kGW; ++w) stage[w] = src[(size_t)min(w0 + w, W - 1) * a.spec_pitch];      // masked when stored
  };
  fetch(0);
  for (int w0 = 0; w0 < W; w0 += kGW) {
    const int wc = min(kGW, W - w0);
    __syncthreads();                                        // previous chunk fully consumed
#pragma unroll
    for (int w = 0; w < kGW; ++w) {
      float* d = ls + (w * 64 + ls_s) * kGPitch + 2 * ls_f;
      const bool ok = ld_ok & (w0 + w < W);
      d[0] = ok ? stage[w].x : 0.f; d[1] = ok ? stage[w].y : 0.f;
    }
    __syncthreads();
    if (w0 + kGW < W) fetch(w0 + kGW);                      // next chunk's loads fly under the MFMAs
    for (int w = 0; w < wc; ++w) {                          // workgroup-uniform trip count
      const float* r0 = ls + (w * 64 + st) * kGPitch + 2 * v;
      const float a0 = r0[part], x0 = sgn * r0[1 - part];
      gr00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, gr00, 0, 0, 0);
      gi00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, a0, gi00, 0, 0, 0);
      if (two) {
        const float* r1 = r0 + 32 * kGPitch;
        const float a1 = r1[part], x1 = sgn * r1[1 - part];
        gr01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, gr01, 0, 0, 0);
        gi01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, a1, gi01, 0, 0, 0);
        gr11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a1, gr11, 0, 0, 0);
        gi11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, a1, gi11, 0, 0, 0);
      }
    }
  }
  // ---- results: block by block through LDS, 32-byte runs per pair row ----------
  cf* lo = reinterpret_cast<cf*>(ls);                       // [1024 entries][4 freq]
  auto emit = [&](const f32x16& gre, const f32x16& gim, const int (&rows)[4]) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * part;      // C/D map of the 32x32 MFMA
      lo[(i * 32 + st) * 4 + v] = make_float2(gre[r], gim[r]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = rows[q];
      if (p < 0) continue;
      cf* dst = a.C + (size_t)p * a.c_pitch + f0;
      const cf* src_e = lo + (tid + 256 * q) * 4;
      if (f0 + 3 < a.nfreq) {
        float4 lo4 = reinterpret_cast<const float4*>(src_e)[0], hi4 = reinterpret_cast<const float4*>(src_e)[1];
        if (a.accumulate) {
          const float4 o0 = reinterpret_cast<const float4*>(dst)[0], o1 = reinterpret_cast<const float4*>(dst)[1];
          lo4 = make_float4(lo4.x + o0.x, lo4.y + o0.y, lo4.z + o0.z, lo4.w + o0.w);
          hi4 = make_float4(hi4.x + o1.x, hi4.y + o1.y, hi4.z + o1.z, hi4.w + o1.w);
        }
        reinterpret_cast<float4*>(dst)[0] = lo4;
        reinterpret_cast<float4*>(dst)[1] = hi4;
      } else {
        for (int k = 0; k < 4; ++k)
          if (f0 + k < a.nfreq) dst[k] = a.accumulate ? cadd(dst[k], src_e[k]) : src_e[k];
      }
    }
  };
  emit(gr00, gi00, prow[0]);
  if (two) {
    emit(gr01, gi01, prow[1]);
    emit(gr11, gi11, prow[2]);
  }
}

hipError_t launch_gram(const GramArgs& a, hipStream_t s) {
  static const bool mfma = !(getenv("XCX_GRAM_MFMA") && atoi(getenv("XCX_GRAM_MFMA")) == 0);
  if (mfma && a.S <= 64 && (a.c_pitch % 2) == 0 && (reinterpret_cast<size_t>(a.C) % 16) == 0) {
    hipLaunchKernelGGL(k_gram_mfma, dim3(8u * (unsigned)(((a.nfreq + 3) / 4 + 7) / 8)), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  const int nchunks = (a.nfreq + 63) / 64;
  const int per_xcd = (nchunks + 7) / 8;
  const unsigned waves_per_xcd = (unsigned)per_xcd * (unsigned)a.ntiles;
  const dim3 grid(8u * ((waves_per_xcd + kWPB - 1) / kWPB));
  hipLaunchKernelGGL(k_gram, grid, dim3(kWNT * kWPB), 0, s, a);
  return hipGetLastError();
}

}  // namespace xcx
