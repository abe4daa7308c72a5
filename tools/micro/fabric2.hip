// Microbenchmark 2: the SAME per-unit traffic mix as fabric.hip (2.33 MB resident reads, 1.49 MB
// resident ring writes, 1.44 MB streamed nt writes per pair-window) but issued in the ACCESS
// SHAPES of the pair kernels instead of ideal 16-byte-per-lane streams:
//   reads : 8 B per lane, 512 contiguous bytes per wave instruction        (spectra / T loads)
//   ring  : 8 B per lane, four 128-byte segments 15.5 KB apart              (T tile stores)
//   stream: 4 B per lane, four 64-byte segments 6 KB apart, nt              (CCF stores)
// and in intermediate shapes, to see how much of the 8.3 TB/s the memory pipeline keeps when the
// requests are narrow.   hipcc -O3 --offload-arch=gfx950 -o fabric2 fabric2.hip && ./fabric2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Args { const char* A; size_t nA; char* B; size_t nB; char* C; size_t nC; size_t a, b, c; int nunits; float* sink; };

// RW: bytes per lane of reads (8 or 16); TW: ring-store bytes per lane (8 or 16, segmented when 8);
// CW: stream-store bytes per lane (4, 8 or 16; 4 = four 64-byte segments, 8 = four 128-byte segments)
template <int RW, int TW, int CW>
__global__ __launch_bounds__(256) void k_mix(Args p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int u = blockIdx.x; u < p.nunits; u += gridDim.x) {
    const char* A = p.A + ((size_t)u * p.a) % (p.nA - p.a + 1) / 256 * 256;
    char* B = p.B + ((size_t)u * p.b) % (p.nB - p.b + 1) / 256 * 256;
    char* C = p.C + ((size_t)u * p.c) % (p.nC - p.c + 1) / 256 * 256;
    // each wave moves its quarter of the unit; 8 wave-instructions of each kind in flight
    const size_t ra = p.a / 4, rb = p.b / 4, rc = p.c / 4;
    const size_t nra = ra / (64 * RW), nrb = rb / (64 * (TW == 12 ? 8 : TW)), nrc = rc / (64 * (CW >= 32 ? 16 : CW));
    const size_t steps = nra > nrb ? (nra > nrc ? nra : nrc) : (nrb > nrc ? nrb : nrc);
    for (size_t i = 0; i < steps; i += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t j = i + k;
        v[k] = 1.f;
        if (j < nra) {
          const char* s = A + wave * ra + j * 64 * RW + lane * RW;
          if (RW == 16) { f4 t = *reinterpret_cast<const f4*>(s); v[k] = t.x + t.w; }
          else { f2 t = *reinterpret_cast<const f2*>(s); v[k] = t.x + t.y; }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const size_t j = i + k;
        if (j < nrb) {
          if (TW == 16) *reinterpret_cast<f4*>(B + wave * rb + j * 1024 + lane * 16) = f4{v[k], v[k], v[k], v[k]};
          else if (TW == 12) *reinterpret_cast<f2*>(B + wave * rb + j * 512 + lane * 8) = f2{v[k], v[k]};   // round 3: 512 contiguous bytes
          else {   // 4 segments of 128 B (16 lanes x 8 B), 15488 B apart: the T tile store of a column wave
            const size_t seg = (size_t)(lane >> 4) * 15488 + (lane & 15) * 8;
            const size_t base = (wave * rb + (j / 121) * 4 * 15488 + (j % 121) * 128) % (rb * 4 - 4 * 15488);
            *reinterpret_cast<f2*>(B + base + seg) = f2{v[k], v[k]};
          }
        }
        if (j < nrc) {
          if (CW == 32) {   // 16 B per lane, four 256-byte segments (16 lanes x 16 B) 6144 B apart: WG-cooperative CCF store
            const size_t seg = (size_t)(lane >> 4) * 6144 + (lane & 15) * 16;
            const size_t base = (wave * rc + (j / 24) * 4 * 6144 + (j % 24) * 256) % (rc * 4 - 4 * 6144);
            __builtin_nontemporal_store(f4{v[k], v[k], v[k], v[k]}, reinterpret_cast<f4*>(C + base + seg));
          } else if (CW == 64) {   // 16 B per lane, two 512-byte segments (32 lanes x 16 B) 6144 B apart: 128-row tiles
            const size_t seg = (size_t)(lane >> 5) * 6144 + (lane & 31) * 16;
            const size_t base = (wave * rc + (j / 12) * 2 * 6144 + (j % 12) * 512) % (rc * 4 - 2 * 6144);
            __builtin_nontemporal_store(f4{v[k], v[k], v[k], v[k]}, reinterpret_cast<f4*>(C + base + seg));
          } else if (CW == 16) __builtin_nontemporal_store(f4{v[k], v[k], v[k], v[k]}, reinterpret_cast<f4*>(C + wave * rc + j * 1024 + lane * 16));
          else if (CW == 8) __builtin_nontemporal_store(f2{v[k], v[k]}, reinterpret_cast<f2*>(C + wave * rc + j * 512 + lane * 8));
          else {   // 4 segments of 64 B (16 lanes x 4 B), 6144 B apart: the CCF store of a row wave
            const size_t seg = (size_t)(lane >> 4) * 6144 + (lane & 15) * 4;
            const size_t base = (wave * rc + (j / 96) * 4 * 6144 + (j % 96) * 64) % (rc * 4 - 4 * 6144);
            __builtin_nontemporal_store(v[k], reinterpret_cast<float*>(C + base + seg));
          }
        }
        acc += v[k];
      }
    }
  }
  if (acc == 12345.678f) *p.sink = acc;
}

template <int RW, int TW, int CW> void run(const char* name, Args p, hipEvent_t e0, hipEvent_t e1) {
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_mix<RW, TW, CW>), dim3(2048), dim3(256), 0, 0, p);
  CK(hipEventRecord(e0));
  const int reps = 3;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_mix<RW, TW, CW>), dim3(2048), dim3(256), 0, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / (reps * (double)p.nunits);
  printf("  %-64s %.3f us/unit  %.2f TB/s -> %.2f M units/s\n", name, us, (p.a + p.b + p.c) / us / 1e6, 1.0 / us);
}

int main() {
  const size_t MB = 1 << 20;
  const size_t szA = 160 * MB, szB = 80 * MB, szC = (size_t)24 << 30;
  char *A, *B, *C; float* sink;
  CK(hipMalloc(&A, szA)); CK(hipMalloc(&B, szB)); CK(hipMalloc(&C, szC)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, szA)); CK(hipMemset(B, 0, szB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  Args p{A, szA, B, szB, C, szC, (size_t)(2.33 * MB) / 4096 * 4096, (size_t)(1.49 * MB) / 4096 * 4096, (size_t)(1.44 * MB) / 4096 * 4096, 8192, sink};
  printf("full mix per unit: %.2f MB read + %.2f MB ring write + %.2f MB nt stream write\n", p.a / 1e6, p.b / 1e6, p.c / 1e6);
  run<16, 16, 16>("ideal: 16 B/lane everywhere", p, e0, e1);
  run<8, 16, 16>("reads 8 B/lane", p, e0, e1);
  run<16, 8, 16>("ring stores 8 B/lane in 4 x 128 B segments", p, e0, e1);
  run<16, 16, 8>("stream stores 8 B/lane contiguous", p, e0, e1);
  run<16, 16, 4>("stream stores 4 B/lane in 4 x 64 B segments", p, e0, e1);
  run<8, 8, 4>("kernel shapes: 8 B reads, 8 B segmented ring, 4 B segmented stream", p, e0, e1);
  run<8, 8, 8>("8 B reads, 8 B segmented ring, 8 B contiguous stream", p, e0, e1);
  run<16, 16, 32>("stream stores 16 B/lane in 4 x 256 B segments", p, e0, e1);
  run<8, 8, 32>("8 B reads, 8 B segmented ring, 16 B/lane 4 x 256 B stream", p, e0, e1);
  Args q = p; q.a = 0; q.b = 0;
  run<16, 16, 16>("stream only, 16 B/lane", q, e0, e1);
  run<16, 16, 8>("stream only, 8 B/lane", q, e0, e1);
  run<16, 16, 4>("stream only, 4 B/lane segmented", q, e0, e1);
  run<16, 16, 32>("stream only, 16 B/lane in 4 x 256 B segments", q, e0, e1);
  Args r = p; r.b = 0; r.c = 0;
  run<16, 16, 16>("reads only, 16 B/lane", r, e0, e1);
  run<8, 16, 16>("reads only, 8 B/lane", r, e0, e1);
  // round 3: 1.74 MB of reads per unit (T + the band-limited, station-grouped spectra), T tile stores as 512 contiguous bytes
  Args t = p; t.a = (size_t)(1.74 * MB) / 4096 * 4096;
  printf("round-3 mix per unit: %.2f MB read + %.2f MB ring write + %.2f MB nt stream write\n", t.a / 1e6, t.b / 1e6, t.c / 1e6);
  run<16, 16, 16>("ideal: 16 B/lane everywhere", t, e0, e1);
  run<8, 12, 32>("round-3 shapes: 8 B reads, 8 B x 512 B ring, 16 B 4 x 256 B stream", t, e0, e1);
  run<8, 12, 64>("  ... with a 2 x 512 B stream (128-row tiles)", t, e0, e1);
  run<8, 12, 32>("round-3 shapes again", t, e0, e1);
  run<8, 12, 64>("  ... with a 2 x 512 B stream again", t, e0, e1);
  run<8, 12, 16>("  ... with a contiguous 16 B/lane stream", t, e0, e1);
  run<16, 12, 32>("  ... with 16 B/lane reads", t, e0, e1);
  return 0;
}
