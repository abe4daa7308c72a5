// Issue rate of packed vs scalar f32 VALU on gfx950: independent chains, hipEvent-timed.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/pkrate.hip -o /tmp/pkrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int MODE> __global__ void k(float* out, float seed) {
  v2f a[8];
  for (int i = 0; i < 8; ++i) a[i] = v2f{seed + i + threadIdx.x, seed - i};
  v2f b = v2f{seed * 0.5f, seed * 0.25f};
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (MODE == 0) {          // 2 x v_add_f32
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(b.y));
      } else if constexpr (MODE == 1) {   // 1 x v_pk_add_f32
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if constexpr (MODE == 2) {   // 2 x v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(b.y));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(b.y), "v"(b.x));
      } else if constexpr (MODE == 3) {   // 1 x v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
      } else if constexpr (MODE == 4) {   // 1 x v_pk_mul_f32
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      } else if constexpr (MODE == 5) {   // pk_add with op_sel swap + neg (a + i*b)
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "+v"(a[i]) : "v"(b));
      } else if constexpr (MODE == 6) {   // 1 x v_add_f32 only (half the work) for reference
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
      }
    }
  }
  v2f s = a[0];
  for (int i = 1; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <int MODE> void run(const char* name, float* d, int wavesPerSimd) {
  const int blocks = 256 * wavesPerSimd, threads = 256;   // 4 waves per WG -> one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(d, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)ITER * 8 * ((MODE == 0 || MODE == 2) ? 2 : 1);   // per wave
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, wavesPerSimd, ms,
         ms * 1e6 / (instr * wavesPerSimd));
}

int main(int argc, char** argv) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  if (argc > 1) {           // single configuration (for counter runs): pkrate <waves/SIMD>
    const int w = atoi(argv[1]);
    run<6>("v_add_f32 x1", d, w);
    run<1>("v_pk_add_f32", d, w);
    run<2>("v_fma_f32 x2", d, w);
    return 0;
  }
  for (int w : {1, 2, 4}) {
    run<6>("v_add_f32 x1", d, w);
    run<0>("v_add_f32 x2", d, w);
    run<1>("v_pk_add_f32", d, w);
    run<5>("v_pk_add_f32 op_sel/neg", d, w);
    run<2>("v_fma_f32 x2", d, w);
    run<3>("v_pk_fma_f32", d, w);
    run<4>("v_pk_mul_f32", d, w);
  }
  return 0;
}
