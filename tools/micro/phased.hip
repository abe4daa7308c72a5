// Microbenchmark 3: the pair pass's traffic in the PHASE STRUCTURE of a transform wave, no arithmetic.
// fabric2.hip issues its loads and stores as 8-deep independent streams and reaches 1.60 M units/s with the
// pass's access shapes; the real pass runs at 1.27-1.30 M (1.37 M with the arithmetic compiled out).  A
// transform wave cannot stream: it needs ALL its inputs before its first output.  Here every wave is
//   column item: 48 loads of 8 B/lane (two station columns, cache-resident), wait, 24 ring stores (T)
//   row item   : 30 loads of 8 B/lane (T rows), wait, 15 nt stores of 16 B/lane in 4 x 256 B runs (CCF)
// 121 column + 96 row items per unit (pair-window), 4 waves per workgroup, 16 waves per CU.
//   hipcc -O3 --offload-arch=gfx950 -o phased phased.hip && ./phased
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args { const char* A; size_t nA; char* B; size_t nB; char* C; size_t nC; int nunits; float* sink; int stream_mode; long item0, nitems; };

constexpr int kCol = 121, kRow = 96, kItems = kCol + kRow;

// STREAMED = 1: the same bytes per item, but each load's value is stored right away (8-deep), like fabric2
// TSHAPE 0: T stores / loads as contiguous 512-byte instructions; 1: the kernel's layout [res/128][k2][res%128] --
// a column wave stores 12 runs of 1 KB 124 KB apart, a row wave's load instruction takes four 128-byte pieces 1 KB apart;
// 2: [res/64][k2][res%64] (512-byte store runs, pieces 512 B apart); 3: [res/16][k2][res%16] (row waves read one
// contiguous block, column stores are four 128-byte pieces per instruction)
template <int TSHAPE>
__global__ __launch_bounds__(256, 4) void k_phased(Args p) {
  const int lane = threadIdx.x & 63;
  const long wave0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  float acc = 0.f;
  // one-shot launches (item0 / nitems set): every wave takes ONE item, like the dual launches of the pair pass
  const long it_end = p.nitems ? p.item0 + p.nitems : (long)p.nunits * kItems;
  for (long it = p.item0 + wave0; it < it_end; it += nw) {
    const long u = it / kItems;
    const int k = (int)(it % kItems);
    if (k < kCol) {
      // two station columns of 1536 x 8 B = 12 KB each, from a cache-resident pool
      const char* a = p.A + ((size_t)(u * 7919 + k) * 24576) % (p.nA - 24576) / 512 * 512;
      constexpr size_t kUnit = (size_t)1536 * 121 * 8;
      const size_t nring = p.nB / kUnit;
      char* tu = p.B + (size_t)(u % nring) * kUnit;
      char* t = p.B + ((size_t)it * 12288) % (p.nB - 12288) / 512 * 512;
      f2 v[48];
#pragma unroll
      for (int j = 0; j < 48; ++j) v[j] = *reinterpret_cast<const f2*>(a + j * 512 + lane * 8);
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        const f2 w = {v[2 * j].x + v[2 * j + 1].y, v[2 * j].y - v[2 * j + 1].x};
        // instruction j covers res = 64 j .. 64 j + 63 of column k
        char* dst;
        if (TSHAPE == 0) dst = t + j * 512 + lane * 8;
        else if (TSHAPE == 1) dst = tu + (size_t)(j >> 1) * (121 * 1024) + (size_t)k * 1024 + (j & 1) * 512 + lane * 8;
        else if (TSHAPE == 2) dst = tu + (size_t)j * (121 * 512) + (size_t)k * 512 + lane * 8;
        else dst = tu + (size_t)(4 * j + (lane >> 4)) * (121 * 128) + (size_t)k * 128 + (lane & 15) * 8;
        *reinterpret_cast<f2*>(dst) = w;
      }
      acc += v[0].x;
    } else {
      constexpr size_t kUnit = (size_t)1536 * 121 * 8;
      const size_t nring = p.nB / kUnit;
      const char* tu = p.B + (size_t)((u + 37) % nring) * kUnit;        // another unit's T (an earlier launch wrote it)
      const char* t = p.B + ((size_t)(it * 31 + 5) * 15360) % (p.nB - 15360) / 512 * 512;
      char* c = p.C + ((size_t)it * 24576) % (p.nC - 4 * 6144 - 16384) / 256 * 256;
      const int r = k - kCol;                                            // rows res = 16 r .. 16 r + 15
      f2 v[30];
#pragma unroll
      for (int j = 0; j < 30; ++j) {
        int k2 = 4 * j + (lane >> 4); k2 = k2 < 121 ? k2 : 120;
        const char* src;
        if (TSHAPE == 0) src = t + j * 512 + lane * 8;
        else if (TSHAPE == 1) src = tu + (size_t)(r >> 3) * (121 * 1024) + (size_t)k2 * 1024 + (r & 7) * 128 + (lane & 15) * 8;
        else if (TSHAPE == 2) src = tu + (size_t)(r >> 2) * (121 * 512) + (size_t)k2 * 512 + (r & 3) * 128 + (lane & 15) * 8;
        else src = tu + (size_t)r * (121 * 128) + (size_t)k2 * 128 + (lane & 15) * 8;
        v[j] = __builtin_nontemporal_load(reinterpret_cast<const f2*>(src));
      }
      const size_t seg = (size_t)(lane >> 4) * 6144 + (lane & 15) * 16;
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        const f4 w = {v[2 * j].x, v[2 * j].y, v[2 * j + 1].x, v[2 * j + 1].y};
        __builtin_nontemporal_store(w, reinterpret_cast<f4*>(c + (size_t)(j / 24) * 4 * 6144 + (j % 24) * 256 + seg));
      }
      acc += v[0].y;
    }
  }
  if (acc == 12345.678f) *p.sink = acc;
}

int main(int argc, char** argv) {
  const size_t MB = 1 << 20;
  // station-column pool: 2 MB = resident in every XCD's L2 (the station-grouped launches of the real pass: 0.25 MB
  // of spectra per pair-window reach the fabric); 40 MB = Infinity-Cache resident, every column load crosses the fabric
  const size_t szA = (argc > 1 ? atoi(argv[1]) : 2) * MB, szB = 160 * MB, szC = (size_t)24 << 30;
  char *A, *B, *C; float* sink;
  CK(hipMalloc(&A, szA)); CK(hipMalloc(&B, szB)); CK(hipMalloc(&C, szC)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, szA)); CK(hipMemset(B, 0, szB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Args p{A, szA, B, szB, C, szC, 8192, sink, 0};
  const double bytes = kCol * (24576.0 + 12288.0) + kRow * (15360.0 + 15360.0);
  printf("station-column pool %zu MB\n", szA / MB);
  printf("per unit: %.2f MB read from cache-resident pools + %.2f MB ring write + %.2f MB nt stream (L2-level bytes; the\n"
         "station columns mostly hit L2 in the real pass)\n", (kCol * 24576.0 + kRow * 15360.0) / 1e6, kCol * 12288.0 / 1e6, kRow * 15360.0 / 1e6);
  auto run = [&](auto kern, const char* name) {
    const int grid = 4096;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e0));
    const int reps = 3;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / (reps * (double)p.nunits);
    printf("  %-78s %.3f us/unit -> %.2f M units/s\n", name, us, 1.0 / us);
  };
  run(k_phased<0>, "T as contiguous 512-byte instructions (ideal)");
  run(k_phased<1>, "T = [res/128][k2][res%128]: 1 KB store runs, loads of four 128 B pieces 1 KB apart");
  run(k_phased<2>, "T = [res/64][k2][res%64]: 512 B store runs, loads of four 128 B pieces 512 B apart");
  run(k_phased<3>, "T = [res/16][k2][res%16]: stores of four 128 B pieces, loads contiguous");
  run(k_phased<0>, "ideal again");
  // the launch structure of the pair pass: 28 units per launch, one item per wave, launches alternating on two streams
  {
    hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    for (int per : {28, 56, 112}) {
      Args q = p;
      const long items = (long)per * kItems;
      const int grid = (int)((items + 3) / 4);
      const int nl = p.nunits / per;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int l = 0; l < nl; ++l) {
          q.item0 = (long)l * items; q.nitems = items;
          hipLaunchKernelGGL(k_phased<0>, dim3(grid), dim3(256), 0, st[l & 1], q);
        }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) {
          const double us = ms * 1e3 / ((double)nl * per);
          printf("  ideal T shapes, one-shot waves, %3d units per launch, two streams: %.3f us/unit -> %.2f M units/s\n", per, us, 1.0 / us);
        }
      }
    }
  }
  return 0;
}
