// Microbenchmark: what the memory system sustains for the pair pass's traffic MIX, with no
// arithmetic at all.  Per "unit" (one pair-window) a workgroup
//   reads  a bytes from region A (small, re-read every pass  -> Infinity-Cache / L2 resident:
//                                 the T tile + the station spectra),
//   writes b bytes to   region B (small ring, rewritten in place -> cache resident: T),
//   writes c bytes to   region C (huge, streamed once with nt stores -> HBM: the CCFs).
// Prints us per unit and TB/s for read-only, write-only, stream-only, the T round trip and
// the full mix, so the ceiling of the two-pass design (a = 2.33 MB, b = 1.49 MB, c = 1.44 MB)
// is measured rather than inferred from the sum of FETCH_SIZE + WRITE_SIZE.
//   hipcc -O3 --offload-arch=gfx950 -o fabric fabric.hip && ./fabric
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float4* A; size_t nA;      // float4 elements
  float4* B; size_t nB;
  float4* C; size_t nC;
  size_t a, b, c;                  // float4 per unit
  int nunits;
  float* sink;
};

__global__ __launch_bounds__(256) void k_mix(Args p) {
  float acc = 0.f;
  for (int u = blockIdx.x; u < p.nunits; u += gridDim.x) {
    const size_t a0 = ((size_t)u * p.a) % (p.nA - p.a + 1);
    const size_t b0 = ((size_t)u * p.b) % (p.nB - p.b + 1);
    const size_t c0 = ((size_t)u * p.c) % (p.nC - p.c + 1);
    const size_t steps = (p.a > p.b ? (p.a > p.c ? p.a : p.c) : (p.b > p.c ? p.b : p.c));
    for (size_t i = threadIdx.x; i < steps; i += 4 * 256) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t j = i + (size_t)k * 256;
        v[k] = (j < p.a) ? p.A[a0 + j] : make_float4(1.f, 2.f, 3.f, 4.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const size_t j = i + (size_t)k * 256;
        if (j < p.b) p.B[b0 + j] = v[k];
        if (j < p.c) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          f4 w = {v[k].x, v[k].y, v[k].z, v[k].w};
          __builtin_nontemporal_store(w, reinterpret_cast<f4*>(&p.C[c0 + j]));
        }
        acc += v[k].x;
      }
    }
  }
  if (acc == 12345.678f) *p.sink = acc;
}

int main() {
  const size_t MB = 1 << 20;
  const size_t szA = 160 * MB, szB = 80 * MB, szC = (size_t)24 << 30;
  float4 *A, *B, *C; float* sink;
  CK(hipMalloc(&A, szA)); CK(hipMalloc(&B, szB)); CK(hipMalloc(&C, szC)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, szA)); CK(hipMemset(B, 0, szB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; double a, b, c; };   // MB per unit
  const Case cases[] = {
      {"read resident (T+spectra) only        ", 2.33, 0, 0},
      {"write resident ring (T) only           ", 0, 1.49, 0},
      {"nt stream write (CCF) only             ", 0, 0, 1.44},
      {"T round trip: read 1.49 + write 1.49   ", 1.49, 1.49, 0},
      {"resident read 2.33 + stream write 1.44 ", 2.33, 0, 1.44},
      {"full mix 2.33 r + 1.49 w + 1.44 stream ", 2.33, 1.49, 1.44},
      {"fused-ideal: 0.84 r + 1.44 stream      ", 0.84, 0, 1.44},
      {"T in L2:   0.84 r + 1.49 w + 1.44 s    ", 0.84, 1.49, 1.44},
  };
  for (int grid : {1024, 2048, 4096}) {
    printf("grid %d x 256 threads\n", grid);
    for (const Case& cs : cases) {
      Args p{};
      p.A = A; p.nA = szA / 16; p.B = B; p.nB = szB / 16; p.C = C; p.nC = szC / 16;
      p.a = (size_t)(cs.a * MB / 16); p.b = (size_t)(cs.b * MB / 16); p.c = (size_t)(cs.c * MB / 16);
      p.nunits = 8192; p.sink = sink;
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, p);
      CK(hipEventRecord(e0));
      const int reps = 3;
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, p);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us_unit = ms * 1e3 / (reps * (double)p.nunits);
      const double bytes = (cs.a + cs.b + cs.c) * MB;
      printf("  %s: %.3f us/unit  %.2f TB/s total (r %.2f, w %.2f)  -> %.2f M units/s\n", cs.name, us_unit,
             bytes / us_unit / 1e6, cs.a * MB / us_unit / 1e6, (cs.b + cs.c) * MB / us_unit / 1e6, 1.0 / us_unit);
    }
  }
  return 0;
}
