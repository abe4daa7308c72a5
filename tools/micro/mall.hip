// Microbenchmark: does a small working set that is rewritten / re-read stay in
// the Infinity Cache (MALL)?  Effective bandwidth of repeated float4 write,
// read and copy passes over footprints from 32 MB to 4 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_write(float4* p, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(v, v, v, v);
}
__global__ void k_read(const float4* p, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = p[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *out = acc;
}
__global__ void k_copy(const float4* a, float4* b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main() {
  const size_t maxb = (size_t)4 << 30;
  float4 *a, *b; float* o;
  hipMalloc(&a, maxb); hipMalloc(&b, maxb); hipMalloc(&o, 4);
  hipMemset(a, 0, maxb); hipMemset(b, 0, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t total = (size_t)16 << 30;   // bytes moved per measurement
  for (size_t mb : {32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    const int reps = (int)(total / bytes);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int w = 0; w < 2; ++w) { if (mode == 0) k_write<<<4096, 256>>>(a, n, 1.f); else if (mode == 1) k_read<<<4096, 256>>>(a, n, o); else k_copy<<<4096, 256>>>(a, b, n); }
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) k_write<<<4096, 256>>>(a, n, (float)r);
        else if (mode == 1) k_read<<<4096, 256>>>(a, n, o);
        else k_copy<<<4096, 256>>>(a, b, n);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("footprint %5zu MB: write %6.2f TB/s  read %6.2f TB/s  copy(r+w) %6.2f TB/s\n", mb,
           total / (ms[0] * 1e-3) / 1e12, total / (ms[1] * 1e-3) / 1e12, 2.0 * total / (ms[2] * 1e-3) / 1e12);
  }
  return 0;
}
