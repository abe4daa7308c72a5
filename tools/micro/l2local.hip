// Microbenchmark: can the intermediate T of a pair-window stay in ONE XCD's 4 MiB L2 when the same
// XCD writes it (column phase) and reads it back (row phase)?  What the answer decides: an
// XCD-local pair pass would keep 2.95 MB of the 4.68 MB a pair-window moves across the L2 <->
// fabric boundary on chip -- if dirty lines that are rewritten in place never leave the L2.
//
// One persistent workgroup per CU (grid 256 x 256 threads); a workgroup learns its XCD from
// HW_REG_XCC_ID; the 32 workgroups of an XCD share a private ring of `slots` x 1.5 MB.  Per
// iteration: every workgroup writes its 48 KB share of the slot (16 B per lane), XCD barrier,
// every workgroup reads ANOTHER workgroup's share with nt loads (L1 bypass), XCD barrier.
// The XCD barrier is an L2-executed atomic (workgroup scope: no sc1) polled with fetch_add(0);
// every spin is bounded, so the grid drains even if the placement is not 32 per XCD.
//   mode 0: T round trip only         mode 1: + 1.44 MB of nt streaming stores per iteration (CCF)
//   mode 2: round trip with sc1 (write-through) stores, the reference point for WRITE_SIZE
//   hipcc -O3 --offload-arch=gfx950 -o l2local l2local.hip && ./l2local [slots] [iters]
//   rocprofv3 --pmc WRITE_SIZE -- ./l2local   /   --pmc FETCH_SIZE   (separate passes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kShare = 48 * 1024;            // bytes per workgroup and phase
constexpr int kWgPerXcd = 32;
constexpr size_t kSlot = (size_t)kShare * kWgPerXcd;   // 1.5 MB

struct Args {
  char* ring;          // [8][slots][kSlot]
  char* stream;        // streaming-store target (mode 1)
  size_t stream_bytes;
  unsigned* ctr;       // [8][32] barrier counters (one 128-byte line per XCD), [8*32 + xcd] arrival tickets
  unsigned* err;       // bounded-spin overruns
  float* sink;
  int slots, iters, mode;
};

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ bool xcd_barrier(unsigned* c, unsigned target, unsigned* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    for (;;) {
      unsigned seen;   // L2-served poll (sc1 bypasses the CU's L1; the compiler turns fetch_add(c, 0) into an L1-cached load)
      asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(c) : "memory");
      if (seen >= target) break;
      __builtin_amdgcn_s_sleep(4);
      if (++spins > 200000) { atomicAdd(err, 1u); ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_l2local(Args a) {
  const unsigned x = xcc_id() & 7;
  __shared__ unsigned me_s;
  if (threadIdx.x == 0) me_s = atomicAdd(a.ctr + 8 * 32 + x, 1u);     // arrival ticket inside the XCD
  __syncthreads();
  const unsigned me = me_s;
  if (me >= kWgPerXcd) { if (threadIdx.x == 0) atomicAdd(a.err, 1000000u); return; }   // not 32 per XCD: do not join
  unsigned* bar = a.ctr + x * 32;
  char* ring = a.ring + (size_t)x * a.slots * kSlot;
  char* cs = a.stream + ((size_t)x * kWgPerXcd + me) * (a.stream_bytes / (8 * kWgPerXcd));
  const size_t cs_len = a.stream_bytes / (8 * kWgPerXcd);
  size_t cs_off = 0;
  float acc = 0.f;
  unsigned gen = 0;
  for (int it = 0; it < a.iters; ++it) {
    char* slot = ring + (size_t)(it % a.slots) * kSlot;
    // column phase: write my share
    f4* w = reinterpret_cast<f4*>(slot + (size_t)me * kShare) + threadIdx.x;
    const f4 v = {(float)it, 1.f, 2.f, 3.f};
#pragma unroll
    for (int k = 0; k < kShare / (256 * 16); ++k) {
      if (a.mode == 2) __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
          __builtin_amdgcn_make_buffer_rsrc(slot, 0, (int)kSlot, 0x00020000),
          (int)(me * kShare + (k * 256 + threadIdx.x) * 16), 0, 16 /* sc1 */);
      else w[k * 256] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    gen += kWgPerXcd;
    if (!xcd_barrier(bar, gen, a.err)) break;
    // row phase: read the share of another workgroup, L1 bypassed
    const f4* r = reinterpret_cast<const f4*>(slot + (size_t)((me + 7) % kWgPerXcd) * kShare) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kShare / (256 * 16); ++k) {
      const f4 t = __builtin_nontemporal_load(r + k * 256);
      acc += t.x + t.w;
    }
    if (a.mode == 1) {   // 1.44 MB per XCD and iteration = 45 KB per workgroup: 11 x 4 KB + a bit
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(cs + cs_off) + k * 256 + threadIdx.x);
      }
      cs_off += 11 * 4096;
      if (cs_off + 11 * 4096 > cs_len) cs_off = 0;
    }
    gen += kWgPerXcd;
    if (!xcd_barrier(bar, gen, a.err)) break;
  }
  if (acc == 12345.678f) *a.sink = acc;
}

int main(int argc, char** argv) {
  const int slots = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  Args a{};
  a.slots = slots; a.iters = iters;
  a.stream_bytes = (size_t)8 << 30;
  CK(hipMalloc(&a.ring, 8 * slots * kSlot));
  CK(hipMalloc(&a.stream, a.stream_bytes));
  CK(hipMalloc(&a.ctr, sizeof(unsigned) * (8 * 32 + 8)));
  CK(hipMalloc(&a.err, 4)); CK(hipMalloc(&a.sink, 4));
  CK(hipMemset(a.ring, 0, 8 * slots * kSlot));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"T round trip in one XCD (plain stores, nt loads)", "  ... + 1.44 MB nt CCF stream per iteration",
                          "T round trip with sc1 (write-through) stores"};
  printf("ring: %d slots x %.2f MB per XCD, %d iterations, 256 persistent workgroups\n", slots, kSlot / 1e6, iters);
  for (int mode = 0; mode < 3; ++mode) {
    a.mode = mode;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(a.ctr, 0, sizeof(unsigned) * (8 * 32 + 8)));
      CK(hipMemset(a.err, 0, 4));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_l2local, dim3(256), dim3(256), 0, 0, a);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned err = 0; CK(hipMemcpy(&err, a.err, 4, hipMemcpyDeviceToHost));
      if (rep == 1) {
        const double us = ms * 1e3 / iters;            // per iteration = one "pair-window" per XCD
        printf("  %-52s %.3f us per unit and XCD -> %.2f M units/s chip-wide, T traffic %.2f TB/s%s\n", names[mode], us,
               8.0 / us, 8.0 * 2.0 * kSlot / us / 1e6, err ? "  [BARRIER OVERRUN / PLACEMENT]" : "");
        if (err) printf("     err word = %u\n", err);
      }
    }
  }
  return 0;
}
