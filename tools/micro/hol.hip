// Microbenchmark 6: do a wave's LOADS queue behind other waves' STORES in the CU's memory pipeline?
// In the pair pass both wave kinds wait ~5 us (10-12 k cycles) for their loads -- the column waves' loads are L2 hits,
// the row waves' come from the Infinity Cache -- while an idle-chip L2 hit takes ~500 cycles (profiles/r05_stamps.txt).
// Here every CU runs ONE workgroup of 16 waves.  "L" waves: 24 loads of 8 B/lane from a 1 MB slice that every CU of
// the XCD reads (L2 hits), wait, repeat; "S" waves: 15 nt stores of 16 B/lane (1 KB runs) into a stream of their own.
//   mode 0: 8 L + 8 S waves in EVERY CU            mode 1: CUs alternate: 16 L waves | 16 S waves
//   mode 2: 8 L waves per CU, no stores anywhere   mode 3: 8 S waves per CU, no loads
//   mode 4 / 5: as 0 / 1 with PLAIN stores
//   mode 6 / 7: as 0 / 1 with "M" waves instead of S waves: the L pattern on a 192 MB buffer (L2 misses, Infinity-Cache hits)
//   mode 8: 8 M waves per CU alone
//   hipcc -O3 --offload-arch=gfx950 -o hol hol.hip && ./hol
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args { const char* A; const char* M; char* C; size_t per_wave; int iters; int mode; unsigned long long* lat; unsigned long long* tend; float* sink; };

__global__ __launch_bounds__(1024) void k_hol(Args p) {
  extern __shared__ char lds_pad[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = blockIdx.x & 7, cu = blockIdx.x >> 3;
  const int gw = blockIdx.x * 16 + wave;
  bool isL, active = true;
  const int m = p.mode == 8 ? 3 : (p.mode & 3) % 2 + (p.mode < 6 ? (p.mode & 2) : 0);
  const bool plain = p.mode >= 4 && p.mode < 6;
  const bool mwaves = p.mode >= 6;
  if (m == 0) isL = wave < 8;
  else if (m == 1) isL = (cu & 1) == 0;
  else if (m == 2) { isL = true; active = wave < 8; }
  else { isL = false; active = wave < 8; }
  unsigned long long lat = 0;
  float acc = 0.f;
  if (active && isL) {
    const char* a = p.A + ((size_t)x << 20);
    for (int it = 0; it < p.iters; ++it) {
      const size_t base = ((size_t)(gw * 7919 + it * 104729) * 12288) & ((1u << 20) - 1) & ~(size_t)511;
      f2 v[24];
      const unsigned long long t0 = clock64();
#pragma unroll
      for (int j = 0; j < 24; ++j) v[j] = *reinterpret_cast<const f2*>(a + ((base + j * 512) & ((1u << 20) - 1)) + lane * 8);
#pragma unroll
      for (int j = 0; j < 24; ++j) acc += v[j].x;
      asm volatile("" :: "v"(acc));
      lat += clock64() - t0;
    }
  } else if (active && mwaves) {
    for (int it = 0; it < p.iters; ++it) {
      const size_t base = (((size_t)(gw * 7919 + it * 104729) * 12288) % ((size_t)192 << 20)) & ~(size_t)511;
      f2 v[24];
      const unsigned long long t0 = clock64();
#pragma unroll
      for (int j = 0; j < 24; ++j) v[j] = *reinterpret_cast<const f2*>(p.M + (base + j * 512) % ((size_t)192 << 20) + lane * 8);
#pragma unroll
      for (int j = 0; j < 24; ++j) acc += v[j].x;
      asm volatile("" :: "v"(acc));
      lat += clock64() - t0;
    }
  } else if (active) {
    char* c = p.C + (size_t)gw * p.per_wave;
    const f4 w = {(float)lane, 1.f, 2.f, 3.f};
    for (int it = 0; it < p.iters; ++it) {
      char* d = c + ((size_t)it * 15360) % (p.per_wave - 15360);
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        if (plain) *reinterpret_cast<f4*>(d + j * 1024 + lane * 16) = w;
        else __builtin_nontemporal_store(w, reinterpret_cast<f4*>(d + j * 1024 + lane * 16));
      }
    }
  }
  if (lane == 0) {
    p.lat[gw] = active ? lat : 0;
    p.tend[gw] = active ? wall_clock64() : 0;
    if (acc == 12345.f) p.sink[0] = acc;
  }
}

int main() {
  const int nwg = 256, iters = 400;
  const size_t per_wave = (size_t)8 << 20;
  char *A, *C, *M; unsigned long long *lat, *tend; float* sink;
  CK(hipMalloc(&A, 8 << 20)); CK(hipMemset(A, 0, 8 << 20));
  CK(hipMalloc(&C, per_wave * nwg * 16));
  CK(hipMalloc(&M, (size_t)193 << 20)); CK(hipMemset(M, 0, (size_t)193 << 20));
  CK(hipMalloc(&lat, sizeof(unsigned long long) * nwg * 16)); CK(hipMalloc(&tend, sizeof(unsigned long long) * nwg * 16));
  CK(hipMalloc(&sink, 4));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_hol), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10));
  std::vector<unsigned long long> hl(nwg * 16), ht(nwg * 16);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
  for (int mode : {2, 3, 0, 1, 4, 5, 8, 6, 7}) {
    Args a{A, M, C, per_wave, iters, mode, lat, tend, sink};
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_hol, dim3(nwg), dim3(1024), 100 << 10, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(hl.data(), lat, sizeof(unsigned long long) * nwg * 16, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht.data(), tend, sizeof(unsigned long long) * nwg * 16, hipMemcpyDeviceToHost));
    double sl = 0, sm = 0; int nl = 0, ns = 0; unsigned long long t_min = ~0ull, tl = 0, ts = 0;
    const int m = mode == 8 ? 3 : (mode & 3) % 2 + (mode < 6 ? (mode & 2) : 0);
    for (int g = 0; g < nwg * 16; ++g) {
      if (!ht[g]) continue;
      const int wave = g & 15, cu = (g >> 4) >> 3;
      const bool isL = m == 0 ? wave < 8 : m == 1 ? (cu & 1) == 0 : m == 2;
      if (isL) { sl += (double)hl[g] / iters; ++nl; tl = std::max(tl, ht[g]); } else { ++ns; sm += (double)hl[g] / iters; ts = std::max(ts, ht[g]); }
      t_min = std::min(t_min, ht[g]);
    }
    // wall_clock64 ticks at 100 MHz: role completion relative to the kernel's end is what the event time cannot split
    const double lb = (double)nl * iters * 24 * 512, sb = (double)ns * iters * 15 * 1024;
    printf("mode %d: %.3f ms | L waves %d: %.0f cycles per 24-load batch, %.2f TB/s if they ran the whole launch | S waves %d: %.2f TB/s | L done %+.3f ms, S done %+.3f ms before the last wave\n",
           mode, ms, nl, nl ? sl / nl : 0.0, lb / (ms * 1e-3) / 1e12, ns, sb / (ms * 1e-3) / 1e12,
           nl ? (double)(std::max(tl, ts) - tl) / 1e5 : 0.0, ns ? (double)(std::max(tl, ts) - ts) / 1e5 : 0.0);
    if (mode >= 6) printf("        M waves: %.0f cycles per 24-load batch, %.2f TB/s\n", ns ? sm / ns : 0.0, (double)ns * iters * 24 * 512 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
