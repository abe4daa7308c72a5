// xcx_plan.h -- the plan object and the helpers shared by xcx_api.hip (planner, pair engine, time-stack) and
// seg_plan.hip (segmented plans of windows longer than any transform).  Library-internal.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

#include "../../include/xcx.h"
#include "cc_kernels.h"
#include "wave_launch.h"
#include "xcx_internal.h"

#define HIP_TRY(expr)                                                     \
  do {                                                                    \
    hipError_t e_ = (expr);                                               \
    if (e_ != hipSuccess)                                                 \
      return xcx::set_err(e_ == hipErrorOutOfMemory ? XCX_ENOMEM : XCX_EHIP,   \
                     "%s: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                     __FILE__, __LINE__);                                 \
  } while (0)

using xcx::DevBuf; using xcx::Geom; using xcx::PwDesc; using xcx::cf;
using xcx::WColEntry; using xcx::WRowEntry; using xcx::WFusedEntry;

struct xcx_plan {
  xcx_ctx* ctx = nullptr;
  Geom g{};
  const WColEntry* wcol = nullptr;
  const WRowEntry* wrow = nullptr;
  DevBuf gramC, gramTiles, gramPidx;   // time-stack cross-spectra [P][spec_pitch], tile list, pair lookup
  // xcx_cc1d_stack_finish_reduce: in-band elements of gramC packed for the reduce-scatter, this rank's share of the
  // sum, its share unpacked into cross-spectrum rows, the per-column band table {za, zb, offset}
  DevBuf gramPack, gramRecv, gramOwn, bandTab;
  void* gram_own_zeroed = nullptr; unsigned long long gram_own_gen = 0; int gram_own_band = 0;
  const WFusedEntry* dual = nullptr;    // one-launch I2(b-1) + I1(b) kernel, if built for (N1, N2h)
  // windows too long for one transform (n + max|lag| beyond the planner's table, e.g. the reference's
  // production setting window_length = 86400 s, clip_lag = (0, 9 h): scripts/cc.py:25-29): the
  // correlation is assembled from blocks by a time-stacking sub-plan (see seg_run)
  struct Seg { xcx_plan* sub = nullptr; xcx_plan* subp = nullptr; int Bk = 0, Kb = 0, npr = 0; DevBuf segbuf, subout, stats, ptrs, pairs; };
  Seg* seg = nullptr;
  bool flow = false;                    // whole pair phase in one launch (k_wpair_flow), default when built
  DevBuf flow_sync;                     // cdone / rdone counters
  int flow_R = 4;
  // pending row pass of the dual pipeline
  // dual pipelines ("chains"): each chain owns a stream and a double-buffered T
  // and runs its launches in order; chains interleave on the GPU so that one
  // chain's launch ramps up while another's drains, with small enough batches
  // that the whole T ring stays resident in the 256 MiB Infinity Cache
  struct Chain { hipStream_t stream = nullptr; hipEvent_t done = nullptr; size_t d0 = 0; int B = 0; int slot = 0; };
  static const int kMaxChains = 4;
  Chain chains[kMaxChains];
  int nchains = 2;
  hipEvent_t ev_fwd = nullptr;
  hipEvent_t fwd_last = nullptr; hipStream_t fwd_last_stream = nullptr;   // end of the plan's most recent forward()
  // xcx_cc1d_prepare: spectra of the NEXT batch computed on a side stream while
  // the pair phase of the current one runs; consumed by a matching xcx_cc1d_pairs
  DevBuf spectra_next;
  hipStream_t prep_stream = nullptr;
  hipEvent_t prep_done = nullptr, pairs_done[2] = {nullptr, nullptr};
  bool prep_valid = false; unsigned long long pairs_count = 0;   // pair phases recorded in pairs_done[count & 1]
  const float* prep_traces = nullptr; int prep_S = 0, prep_W = 0; long prep_ss = 0, prep_sw = 0;
#ifdef XCX_STAMPS
  DevBuf stampbuf; size_t stamp_waves = 0;
#endif
  unsigned long long gram_key = 0; int gram_ntiles = 0;
  // multi-call accumulation in gramC (xcx_cc1d_stack_add / _finish)
  unsigned long long acc_key = 0; int acc_count = 0, acc_S = 0, acc_P = 0; long long acc_windows = 0; long acc_ld = 0;
  // multi-day stacks: the spectra of up to acc_cap adds wait in station-major slots [S][acc_cap * acc_W] and go
  // through ONE Gram update (the read-modify-write of the P x N/2 accumulator is amortised over them)
  int acc_pending = 0, acc_W = 0, acc_cap = 1, acc_grams = 0;
  bool acc_broken = false;   // an add / flush of the accumulation failed: only xcx_cc1d_stack_reset is accepted
  int fwd_slot_w = 0, fwd_slot_ld = 0;               // forward(): station-major slot mapping (0: dense)
  DevBuf psum, psq;          // partial statistics of a forward chunk
  int L = 0;                 // retained lags
  int batch = 512;           // pair-windows per launch batch
  bool batch_user = false;   // set through xcx_plan_set_batch
  size_t spec_pitch = 0;     // float2 per spectrum slot
  // whitening
  float wb[4] = {0, 0, 0, 0};
  float fixed_norm = 1.f;
  bool whiten_set = false;
  // q24 storage of the intermediate T (wave_kernels.h): fixed-point scale from the bound of |T| of a whitened plan
  float tq = 0.f, tqi = 0.f;
  bool t_q24 = false;        // the dual launches of plain pair products store T as 2 x 24 bits
  int run_ts = 0;            // T storage of the dual_run in progress (every launch of a run agrees, the draining ones too)
  int run_zb0 = 0;           // ... and its zero-stretch skip (ZB0 of the column body, wave_kernels.h; 0: none)
  // tables
  DevBuf tw_N1, tw_N2h, tw_unt, tw_N;
  // workspaces
  DevBuf Tm_f, spectra, T, descs, dtraces, dout, dpairs;
  std::vector<PwDesc> hdescs;
  std::vector<int> bounds;   // launch-batch starts inside one window's P descriptors (pair_order.h); empty: fixed batches
  // spectrum buffers whose out-of-band stretches hold zeros (band-limited plans: the forward column pass
  // does not store there): {pointer, DevBuf generation, band_hi} of up to two buffers (spectra, spectra_next)
  struct Zeroed { void* ptr = nullptr; unsigned long long gen = 0; int band_hi = 0; } zeroed[2], gram_zeroed;
  int zeroed_next = 0;
  unsigned long long desc_key = 0;
  long long launches = 0, pair_windows = 0;
  // optional per-kernel HIP-event timing (xcx_plan_profile)
  int profiling = 0;         // 0 off, 1 phases only, 2 every kernel launch
  struct Ev { hipEvent_t a, b; int kid; int units; };
  std::vector<Ev> evs;
};


enum { KID_STATS = 0, KID_FWD_ROWS, KID_FWD_COLS, KID_PAIR_COLS, KID_PAIR_ROWS, KID_PAIR_FUSED, KID_GRAM, KID_PAIR_PHASE, KID_COUNT };

struct ProfScope {
  xcx_plan* p; size_t idx; bool on; hipStream_t st;
  ProfScope(xcx_plan* p_, int kid, int units, hipStream_t s = nullptr)
      : p(p_), idx(0), on(p_->profiling >= (kid == KID_PAIR_PHASE ? 1 : 2)), st(s ? s : p_->ctx->stream) {
    if (!on) return;
    xcx_plan::Ev e{nullptr, nullptr, kid, units};
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e.a, st);
    p->evs.push_back(e);
    idx = p->evs.size() - 1;
  }
  ~ProfScope() { if (on) (void)hipEventRecord(p->evs[idx].b, st); }
};

// xcx_api.hip
int xcx_internal_forward(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                         int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid);
int xcx_internal_forward_raw(xcx_plan* p, const float* traces, long stride_s, long stride_w,
                             int W, int w0, int sw_begin, int sw_end, int slot0, int n_valid);
// inverse transforms of P cross-spectra rows (descriptors uploaded by the caller) starting at descriptor `first`
int xcx_internal_stack_inverse(xcx_plan* p, int P, float* dout, int first, const xcx::cf* Csrc);
int xcx_internal_upload(xcx_ctx* ctx, xcx::DevBuf& b, const void* src, size_t bytes);
// seg_plan.hip
int xcx_internal_seg_plan_create(xcx_ctx* ctx, int n, int idx_min, int idx_max, int flags, xcx_plan** out);
int xcx_internal_seg_run(xcx_plan* p, const float* const* xs, const float* const* ys, int rows, float* out, long ld_out,
                         long conv_len = 0, const int* psign = nullptr);
