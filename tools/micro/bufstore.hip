// Does a raw buffer store drop out-of-range offsets on gfx950?  (range check of the V#)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(float* base, int nrec_bytes, const int* offs, int n) {
  const unsigned long long ob = (unsigned long long)base;
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32));
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);     // no sign extension
  float* u = (float*)(((unsigned long long)hi << 32) | (unsigned long long)lo);
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(u, 0, nrec_bytes, 0x00020000);
  int i = threadIdx.x;
  if (i < n) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(100.f + i), r, offs[i] * 4, 0, 2);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  hipMemset(d, 0, 4096 * 4);
  std::vector<int> offs = {0, 1, 1023, 1024, 1025, 2000, -1, -5, 1 << 20, -(1 << 20), 500};
  int* doffs; hipMalloc(&doffs, offs.size() * 4);
  hipMemcpy(doffs, offs.data(), offs.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d + 1024, 1024 * 4, doffs, (int)offs.size());     // window = floats [1024, 2048)
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<float> h(4096);
  hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4096; ++i) if (h[i] != 0.f) printf("  [%d] = %g\n", i, h[i]);
  return 0;
}
