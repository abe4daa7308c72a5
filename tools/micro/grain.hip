// Microbenchmark 4 (round 4): does a FINER work grain lift the pair pass?  phased.hip's waves (all loads, wait, all
// stores; the kernel's T shapes) plus a compute delay between the arrival of the loads and the stores, which is what
// the real waves have (column wave: ~10 k cycles until its loads are in, ~11 k of exchanges / butterflies / stores;
// row wave: ~10 k + ~17 k).  SPLIT = 2 halves every item (half the loads, half the stores, half the delay) and
// doubles the resident waves (32 per CU): the same registers per CU hold the same bytes, but a wave's life shrinks
// towards the memory latency.
//   hipcc -O3 --offload-arch=gfx950 -o grain grain.hip && ./grain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args { const char* A; size_t nA; char* B; size_t nB; char* C; size_t nC; int nunits; float* sink; int dcol, drow; };
constexpr int kCol = 121, kRow = 96, kItems = kCol + kRow;

__device__ __forceinline__ void delay_cycles(int cyc, float pin) {
  // the sleep must sit BEHIND the loads: pin it to a loaded value
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(pin) : "memory");
  for (int c = 0; c < cyc; c += 64 * 100) __builtin_amdgcn_s_sleep(100);
}

template <int SPLIT, int WGS = 4>
__global__ __launch_bounds__(256, SPLIT == 1 ? 4 : WGS) void k_grain(Args p) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  float acc = 0.f;
  constexpr int NC = 48 / SPLIT, NS = 24 / SPLIT, NR = 30 / SPLIT;
  const long it_end = (long)p.nunits * kItems * SPLIT;
  for (long its = wave0; its < it_end; its += nw) {
    const long it = its / SPLIT;
    const int h = (int)(its % SPLIT);
    const long u = it / kItems;
    const int k = (int)(it % kItems);
    constexpr size_t kUnit = (size_t)1536 * 121 * 8;
    const size_t nring = p.nB / kUnit;
    if (k < kCol) {
      const char* a = p.A + ((size_t)(u * 7919 + k) * 24576) % (p.nA - 24576) / 512 * 512;
      char* tu = p.B + (size_t)(u % nring) * kUnit;
      f2 v[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) v[j] = *reinterpret_cast<const f2*>(a + (h * NC + j) * 512 + lane * 8);
      delay_cycles(p.dcol / SPLIT, v[NC - 1].x);
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int jj = h * NS + j;
        const f2 w = {v[2 * j].x + v[2 * j + 1].y, v[2 * j].y - v[2 * j + 1].x};
        char* dst = tu + (size_t)(jj >> 1) * (121 * 1024) + (size_t)k * 1024 + (jj & 1) * 512 + lane * 8;
        *reinterpret_cast<f2*>(dst) = w;
      }
      acc += v[0].x;
    } else {
      const char* tu = p.B + (size_t)((u + 37) % nring) * kUnit;
      char* c = p.C + ((size_t)it * 24576) % (p.nC - 4 * 6144 - 16384) / 256 * 256;
      const int r = k - kCol;
      f2 v[NR];
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        int k2 = 4 * (h * NR + j) + (lane >> 4); k2 = k2 < 121 ? k2 : 120;
        const char* src = tu + (size_t)(r >> 3) * (121 * 1024) + (size_t)k2 * 1024 + (r & 7) * 128 + (lane & 15) * 8;
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(src));
      }
      delay_cycles(p.drow / SPLIT, v[NR - 1].x);
      const size_t seg = (size_t)(lane >> 4) * 6144 + (lane & 15) * 16;
#pragma unroll
      for (int j = 0; j < NR / 2; ++j) {
        const int jj = h * (NR / 2) + j;
        const f4 w = {v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y};
        __builtin_nontemporal_store(w, reinterpret_cast<f4*>(c + (size_t)(jj / 24) * 4 * 6144 + (jj % 24) * 256 + seg));
      }
      acc += v[0].y;
    }
  }
  if (acc == 12345.678f) *p.sink = acc;
}

int main(int argc, char** argv) {
  const size_t MB = 1 << 20;
  const size_t szA = 2 * MB, szB = 160 * MB, szC = (size_t)24 << 30;
  char *A, *B, *C; float* sink;
  CK(hipMalloc(&A, szA)); CK(hipMalloc(&B, szB)); CK(hipMalloc(&C, szC)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, szA)); CK(hipMemset(B, 0, szB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Args p{A, szA, B, szB, C, szC, 4096, sink, 0, 0};
  auto run = [&](auto kern, int grid, const char* name) {
    for (int w = 0; w < 1; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e0));
    const int reps = 2;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * (double)p.nunits);
    printf("  %-40s delay col %5d row %5d cycles: %.3f us/unit -> %.2f M units/s\n", name, p.dcol, p.drow, us, 1.0 / us);
  };
  const int delays[][2] = {{0, 0}, {4000, 4000}, {8000, 8000}, {11000, 17000}, {16000, 24000}};
  for (auto& d : delays) {
    p.dcol = d[0]; p.drow = d[1];
    run(k_grain<1>, 4096, "16 waves per CU, whole items");
    run(k_grain<2, 6>, 256 * 6 * 4, "24 waves per CU, half items");
    run(k_grain<2, 7>, 256 * 7 * 4, "28 waves per CU, half items");
  }
  return 0;
}
