// xcx_internal.h -- shared between the translation units of libxcx.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstddef>

struct xcx_ctx;

// library-internal plan flag (above the 16 public flag bits of include/xcx.h), see xcx_plan_cc1d
constexpr int kPlanShortOperand = 0x10000;
// longest single transform of the instantiated kernels: N = 2 * N1 * N2h = 2 * 3072 * 128
constexpr int kMaxTransformN = 786432;   // == XCX_MAX_SINGLE_TRANSFORM (include/xcx.h; asserted in xcx_api.hip)

namespace xcx {
int set_err(int code, const char* fmt, ...);

// Host-pointer entry points (on_device = 0): the caller's buffer is pinned IN PLACE for the
// duration of the call (hipHostRegister), so the DMA engines move it at PCIe speed (~55 GB/s
// instead of ~9 GB/s through the runtime's pageable staging).  If the registration is refused
// (read-only mapping, ulimit -l, already registered or page-locked) the copy silently stays as it is.
// XCX_PIN=0 turns it off.  The object must outlive the stream synchronisation that ends the copies.
struct HostPin {
  void* ptr = nullptr;
  HostPin(const void* p, size_t bytes) {
    static const bool on = !(getenv("XCX_PIN") && atoi(getenv("XCX_PIN")) == 0);
    if (!on || !p || bytes < ((size_t)1 << 20)) return;       // small copies: not worth a syscall
    if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) == hipSuccess) ptr = const_cast<void*>(p);
    else (void)hipGetLastError();
  }
  ~HostPin() { if (ptr) (void)hipHostUnregister(ptr); }
  HostPin(const HostPin&) = delete;
  HostPin& operator=(const HostPin&) = delete;
};

// hipMalloc that asks the caller's pool back once (xcx_ctx::reclaim) when the device is out of memory
hipError_t ctx_malloc(xcx_ctx* ctx, void** ptr, size_t bytes);
size_t ctx_reclaim(xcx_ctx* ctx);

// comm.hip: ncclReduceScatter of `per` floats per rank on the context's stream (send: nranks * per floats)
int comm_reduce_scatter(xcx_ctx* ctx, const float* send, float* recv, size_t per);

struct DevBuf {
  void* ptr = nullptr;
  size_t bytes = 0;
  unsigned long long gen = 0;              // process-wide allocation number of `ptr` (0: none): identifies the memory's lifetime
  int ensure(xcx_ctx* ctx, size_t need);   // grow-only
  void release();
};
}  // namespace xcx

struct xcx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // The caller may keep freed device buffers in a pool of its own (the Python binding does): the library asks for
  // them back before it gives up on an allocation and before it sizes a workspace from the free device memory
  // (xcx_set_reclaim).  The callback may call xcx_free; it returns the number of bytes it released.
  size_t (*reclaim)(void* user) = nullptr;
  void* reclaim_user = nullptr;
  // Streams of the pair-phase pipelines, shared by every plan of the context and created WITH the context: a
  // process has few hardware queues (4 by default) and HIP deals streams onto them in creation order, so two
  // pipelines created late (after the streams of other plans) can land on ONE queue and lose their overlap --
  // measured 75.7 -> 101 ms per step when a time-shift and two transform plans existed first.
  hipStream_t pipe[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* comm = nullptr;      // ncclComm_t
  int nranks = 1, rank = 0;
  xcx::DevBuf scratch[10];   // grow-only workspaces of the row ops / filters / transforms
  struct xcx_plan* ts_plan = nullptr;   // cached plan of xcx_timeshift
  long ts_n = 0;
  bool ts_direct = false;               // ts_plan correlates against the period-extended kernel (timeshift.hip)
  struct xcx_plan* bs_plan[2] = {nullptr, nullptr};   // cached plans of the Bluestein FFTs (fftn.hip)
  struct xcx_plan* bs_seg_plan[2] = {nullptr, nullptr}; long bs_seg_n[2] = {0, 0}; int bs_seg_next = 0;   // ... and the segmented plans of
                                        // the long axes (chirp convolution in blocks); two: a resample alternates between two lengths
  long bs_key[2][3] = {{0, 0, 0}, {0, 0, 0}};
  int bs_next = 0;
  xcx::DevBuf bs_tab;                    // chirp exp(i pi j^2 / n), j <= n, of the last transform length (fftn.hip)
  long bs_tab_n = 0;
  xcx::DevBuf sg_tab;                    // cached DFT-matrix tables of xcx_spectrogram
  unsigned long long sg_key = 0;
  // cached transition-matrix powers of the chunked SOS scan, one table per (filter, chunk geometry): a preprocess
  // list alternates between its filters (high-pass, anti-alias low-pass, low-pass ...), so one slot would miss
  // on every call
  static constexpr int kSosTabs = 8;
  xcx::DevBuf sos_tab[kSosTabs];
  unsigned long long sos_key[kSosTabs] = {0, 0, 0, 0, 0, 0, 0, 0};
  int sos_next = 0;
  // flow pair kernel (one launch per pair phase, in-kernel batch dependencies): device error
  // flag + pinned host mirror, checked by every synchronising entry point
  void* flow_T = nullptr; size_t flow_T_bytes = 0;   // uncached T ring, shared by the context's plans (grow-only)
  int* flow_err_dev = nullptr;
  int* flow_err_host = nullptr;
  bool flow_ran = false;
};
