/*
 * xcx.h -- C ABI of libxcx.so, the MI355X-native cross-correlation engine that
 * replaces xcorr's CPU hot path.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  Bound from Python through ctypes
 * (xcorr_amd/_ffi.py); INTEGRATION.md shows the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative XCX_E* code;
 *     xcx_last_error() returns the message of the calling thread's last error.
 *   - the caller owns every data buffer; the library owns plans, twiddle
 *     tables, workspaces and its HIP streams.  One xcx_ctx <-> one HIP device
 *     <-> one main stream (+ two pipeline streams that the pair phases of all
 *     its plans share); a context and its plans are used from one thread at a
 *     time; contexts share no mutable state (the reference runs one Python
 *     thread per Dask worker, xcorr/scripts/utils.py:62-66).
 *   - `on_device` != 0: data pointers are device pointers obtained from
 *     xcx_malloc(); == 0: host pointers (the library stages through HBM).
 *   - all arrays are C-contiguous float32 with the core dimension last, which
 *     is what every xr.apply_ufunc closure of the reference receives
 *     (xcorr/signal/correlate.py:121-127, filter.py:104-109, ...).
 *
 * Paths are relative to the reference checkout (psmsmets/xcorr).
 */
#ifndef XCX_H
#define XCX_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct xcx_ctx xcx_ctx;
typedef struct xcx_plan xcx_plan;

enum {
  XCX_OK = 0,
  XCX_EINVAL = -1,     /* bad argument (length/shape mismatch, null pointer) */
  XCX_ENOMEM = -2,     /* device allocation failed */
  XCX_EHIP = -3,       /* HIP runtime error */
  XCX_EUNSUPPORTED = -4, /* size outside the planner's table */
  XCX_ENODEV = -5,     /* no usable GPU */
  XCX_ERCCL = -6       /* RCCL error */
};

/* plan flags */
enum {
  XCX_NORMALIZE = 1,   /* divide by ||x|| ||y||   (util/cc.py:76-77)          */
  XCX_UNBIAS = 2,      /* multiply by n/(n-|lag|) (util/cc.py:79,111-141)     */
  XCX_STACK = 4,       /* time-stack: sum over windows (frequency domain)     */
  XCX_ONEBIT = 8,      /* sign(x) per window            (self-defined)        */
  XCX_WHITEN = 16,     /* X/|X| inside a band           (self-defined)        */
  XCX_DEMEAN = 32      /* subtract the window mean (signal/detrend.py:110)    */
};

/* ---- numeric limits of this build ---------------------------------------
 * Every size limit this header speaks of is one of these constants; the library enforces exactly them (static
 * assertions tie them to its kernels) and reports them at run time: xcx_limits(out, XCX_LIM_COUNT) fills
 * out[XCX_LIM_...] (no GPU needed).  tests/test_abi.py compares the two. */
#define XCX_MAX_SINGLE_TRANSFORM 786432L   /* longest transform ONE plan holds (2 x 3072 x 128 points); correlations,
                                              time shifts and DFTs beyond it run block-decomposed ("segmented") */
#define XCX_MAX_CHIRP_POINTS     737280L   /* longest chirp convolution (nin + nout - 1 points) done as one transform */
#define XCX_MAX_ROW_SAMPLES  1073741823L   /* rows of the segmented paths: 2n fits a 32-bit int */
#define XCX_MAX_SPLIT_PARTS        4096L   /* partial transforms of a split DFT (input blocks x output blocks) */
#define XCX_CC2D_MAX_F              256L   /* xcx_cc2d tile rows */
#define XCX_CC2D_MAX_L             4096L   /* xcx_cc2d tile columns */
#define XCX_RAM_MAX_HALF_WIDTH     3072L   /* xcx_ram_normalize */
#define XCX_SOS_MAX_SECTIONS          8L   /* second-order sections of one filter call */
#define XCX_SPECTROGRAM_MAX_WIN     256L   /* window length of the MFMA spectrogram kernels */
#define XCX_SPECTROGRAM_MAX_NFFT   8192L   /* nfft of the MFMA spectrogram kernels */
#define XCX_SPECTROGRAM_SEG_MAX_NFFT 4194304L  /* nfft of xcx_spectrogram_seg's DFT route (2^22) */
#define XCX_ROWS_MAX_BATCH        65535L   /* rows per call of xcx_scale_rows / xcx_absmax_rows (grid y) */
#define XCX_SEG_WORKSPACE_MAX_GB     64L   /* segmented plans: workspace = half of the free device memory, at most this */
enum {
  XCX_LIM_MAX_SINGLE_TRANSFORM = 0, XCX_LIM_MAX_CHIRP_POINTS, XCX_LIM_MAX_ROW_SAMPLES, XCX_LIM_MAX_SPLIT_PARTS,
  XCX_LIM_CC2D_MAX_F, XCX_LIM_CC2D_MAX_L, XCX_LIM_RAM_MAX_HALF_WIDTH, XCX_LIM_SOS_MAX_SECTIONS,
  XCX_LIM_SPECTROGRAM_MAX_WIN, XCX_LIM_SPECTROGRAM_MAX_NFFT, XCX_LIM_SPECTROGRAM_SEG_MAX_NFFT,
  XCX_LIM_ROWS_MAX_BATCH, XCX_LIM_SEG_WORKSPACE_MAX_GB, XCX_LIM_COUNT
};
int xcx_limits(long* out, int count);

/* ---- context / memory / timing ---------------------------------------- */
int xcx_device_count(int* count);
int xcx_init(int device, xcx_ctx** ctx);
/* destroy a context's plans (xcx_plan_destroy) before the context: they run on its streams */
int xcx_destroy(xcx_ctx* ctx);
int xcx_last_error(char* buf, size_t len);
int xcx_device_name(xcx_ctx* ctx, char* buf, size_t len);
int xcx_malloc(xcx_ctx* ctx, size_t bytes, void** dptr);
int xcx_free(xcx_ctx* ctx, void* dptr);
/* A caller that keeps freed device buffers in a pool of its own registers a callback that returns them (through
 * xcx_free) and reports the bytes released.  The library calls it once before an allocation of its own fails with
 * XCX_ENOMEM (plan workspaces, staged host rows, xcx_malloc itself) and before a segmented plan sizes its workspace
 * from the free device memory.  cb = NULL removes it.  Not re-entered from the xcx_free calls it makes. */
int xcx_set_reclaim(xcx_ctx* ctx, size_t (*cb)(void* user), void* user);
/* Release what the context caches between calls: the grow-only workspaces of the row ops, filters and transforms, its
 * tables, and the cached plans of xcx_timeshift and of the any-length transforms (after a day-long rfft, time shift
 * or segmented correlation they hold tens of GB until xcx_destroy).  Plans created by the caller are not touched. */
int xcx_trim(xcx_ctx* ctx);
/* page-locked host memory (hipHostMalloc): device-to-host copies into it run at PCIe speed; the
 * Python mirrors return their result arrays in it (xcorr_amd.Context.pinned_empty) */
int xcx_host_alloc(xcx_ctx* ctx, size_t bytes, void** hptr);
int xcx_host_free(xcx_ctx* ctx, void* hptr);
int xcx_memset(xcx_ctx* ctx, void* dptr, int value, size_t bytes);
int xcx_memcpy_h2d(xcx_ctx* ctx, void* dst, const void* src, size_t bytes);
int xcx_memcpy_d2h(xcx_ctx* ctx, void* dst, const void* src, size_t bytes);
/* asynchronous on the context's stream */
int xcx_memcpy_d2d(xcx_ctx* ctx, void* dst, const void* src, size_t bytes);
/* Measurement aid: overwrite `bytes` of device memory `reps` times with 16-byte-per-lane streaming
 * (non-temporal) stores -- the access shape of the CCF store of the pair pass -- and return the elapsed
 * milliseconds (HIP events): the write ceiling of an output-bound pass on this chip (bench.py). */
int xcx_stream_write_rate(xcx_ctx* ctx, void* dptr, size_t bytes, int reps, float* ms);
/* rows x width_bytes, with independent pitches (bytes) */
int xcx_memcpy2d_d2h(xcx_ctx* ctx, void* dst, size_t dpitch, const void* src,
                     size_t spitch, size_t width_bytes, size_t rows);
int xcx_sync(xcx_ctx* ctx);
/* HIP events on the context's stream (the stream every kernel is launched
 * on); xcx_timer_stop synchronises and returns the elapsed milliseconds. */
int xcx_timer_start(xcx_ctx* ctx);
int xcx_timer_stop(xcx_ctx* ctx, float* ms);

/* ---- cross-correlation plan (replaces util/cc.py:22-79 cc(), and the
 *      _correlate1d closure of signal/correlate.py:108-113) ---------------
 * n        : window length in samples (len(x) == len(y), util/cc.py:60-62)
 * nfft     : transform length, 0 = smallest supported N >= n + max|lag| of the
 *            retained slice (2n-1 for the full lag axis).  The reference uses
 *            exactly 2n-1 (util/cc.py:67); any such N yields the identical
 *            linear cross-correlation on the retained lags (aliases m +- N of a
 *            correlation that vanishes beyond |m| = n-1 cannot reach them).
 * idx_min, idx_max : retained slice of the 2n-1 lags, end exclusive
 *            (core/process.py:155-162 `[index_min:index_max]`,
 *            util/time.py:104-108); idx_max = 0 means 2n-1.
 * Windows for which no single transform holds n + max|lag| points (beyond XCX_MAX_SINGLE_TRANSFORM: the
 * reference's production setting window_length = 86400 s, clip_lag = (0, 9 h), scripts/cc.py:25-29) get a
 * SEGMENTED plan behind the same entry points (not with XCX_WHITEN / XCX_STACK): y is cut into blocks and the
 * lags into chunks of 184 320 samples each (half of the tuned 368 640-point block transform), every distinct window of a call is transformed
 * once, and the rows go through in chunks whose workspace is half of the free device memory, at most
 * XCX_SEG_WORKSPACE_MAX_GB (XCX_SEG_BUDGET_MB overrides; a registered xcx_set_reclaim callback is asked first).
 */
/* What a plan for `need` = n + max|lag| points (nfft != 0: exactly that transform) runs on, without a GPU or a plan:
 * out5 = {N, N1, N2h, 1 if the (N1, N2h) pair has a dual / flow pair kernel, 1 if that kernel has the q24 T storage};
 * N = -1: no single transform holds it (segmented plan).  xcx_plan_fused_list: the instantiated (N1, N2h) pair
 * kernels, out[2 i], out[2 i + 1], *count of them. */
int xcx_plan_geometry(long long need, int nfft, int* out5);
int xcx_plan_fused_list(int* out, int cap, int* count);
int xcx_plan_cc1d(xcx_ctx* ctx, int n, int nfft, int idx_min, int idx_max,
                  int flags, xcx_plan** plan);
int xcx_plan_destroy(xcx_plan* plan);
/* whitening band in Hz (self-defined op): gain 1 in [fmin,fmax], raised-cosine
 * ramps of width ftaper outside, 0 beyond. */
int xcx_plan_set_whiten(xcx_plan* plan, double sampling_rate, double fmin,
                        double fmax, double ftaper);
/* Storage of the pair pass's intermediate T between its two kernels: *ts = 0 float2 (8 bytes per element), 1 "q24"
 * (two 24-bit fixed-point numbers, 6 bytes per element).  q24 is the default for WHITENED plans whose column length
 * has the lane-pair store (N1 in {512, 768, 1024, 1536}) on the dual launches of plain pair products: a whitened
 * spectrum has the known modulus gain(k) * scale, so |T| has a bound and the fixed-point scale never overflows;
 * the arithmetic stays float32.  Measured: +4-6 % pair-windows/s, worst error against the float64 oracle 2.1e-6 of
 * max|cc| (pure-noise CCFs; float2 storage: 1.2e-6), profiles/r05_t24_error_table.txt.  Stacked plans, precomputed
 * cross-spectra and plans without whitening keep float2.  Environment XCX_T_STORAGE=f32 | q24 overrides, read by
 * xcx_plan_set_whiten. */
int xcx_plan_t_storage(xcx_plan* plan, int* ts);
/* Zero-stretch skipping of the pair pass's column kernel (round 5): a whitened spectrum is exactly zero outside the
 * whitening band, i.e. in the middle stretch of every column of the N1 x N2 spectrum layout.  For q24 plans the column
 * kernel is instantiated with that stretch known at compile time -- the 64-point blocks out2[0] .. out2[1] - 1 - out2[0]
 * of the out2[1] = N1 / 64 blocks of every column are neither loaded nor multiplied (results bit-identical: the skipped
 * products are 0 * x).  out2[0] = 0: the generic kernel (no band, float2 storage, or no instantiated range inside the
 * band's stretch; such loads then go to one cached chunk of stored zeros).  14 of 24 blocks for a 0.1-10 Hz band at
 * 50 Hz: +3.7 % pair-windows/s.  Environment XCX_ZSKIP=0 turns it off (A/B). */
int xcx_plan_zero_skip(xcx_plan* plan, int* out2);
/* Host-only (no GPU): the processing order xcx_cc1d_pairs gives a pair list -- launch batches of at most
 * `batch` pairs that touch few stations (csrc/pair_order.h: lines of the affine plane AG(2,8) for <= 64
 * stations, greedy station sets otherwise).  perm[P] = order (indices into `pairs`), bounds[ngroups + 1] =
 * group starts (bounds must hold P + 1 ints).  Exposed for tests and for callers that want to lay their
 * output out in processing order. */
int xcx_pair_groups(const int* pairs, int S, int P, int batch, int* perm, int* bounds, int* ngroups);
/* tuning: pair-windows per launch batch (default 512) */
int xcx_plan_set_batch(xcx_plan* plan, int pair_windows_per_batch);
/* out[0..5] = N, N1, N2, L (= idx_max-idx_min), batch, spectrum floats */
int xcx_plan_info(xcx_plan* plan, int* out6);

/* All pairs x all windows (the hot loop of core/process.py:82-165, batched):
 *   traces : S station records; window w of station s starts at
 *            traces + s*stride_s + w*stride_w (elements) and has n samples
 *   pairs  : P x 2 int32 station indices (i, j): R[k] = sum_t x_i[t+k] x_j[t]
 *   out    : row (p*W + w) at out + (p*W + w)*ld_out, L floats each
 *            (XCX_STACK: row p at out + p*ld_out, sum over the W windows)
 *   ld_out : elements between consecutive rows, >= L; 0 means L.
 */
int xcx_cc1d_pairs(xcx_plan* plan, const float* traces, int S, int W,
                   long stride_s, long stride_w, const int* pairs, int P,
                   float* out, long ld_out, int on_device);

/* Multi-call time-stack (BASELINE configs[3]: 30 days per pair): an XCX_STACK plan accumulates
 * the cross-spectra  G[p][k] += sum_w X_{i,w}[k] conj X_{j,w}[k]  of every call in the frequency
 * domain (one cgemm-shaped update per call on the matrix cores) and pays for the P inverse
 * transforms once, in xcx_cc1d_stack_finish -- by linearity the result equals the sum over the
 * calls of xcx_cc1d_pairs(XCX_STACK) outputs (the reference has no stack function; SURVEY 8c:
 * oracle = sum over windows of util/cc.py:22-79 outputs).  Every add of one accumulation must
 * pass the same S and pair list; W and the traces may differ.  finish writes row p at
 * out + p*ld_out (L floats), returns the number of windows summed in *windows (may be null)
 * and clears the accumulator.  xcx_cc1d_pairs on the same plan is refused while an
 * accumulation is open. */
int xcx_cc1d_stack_add(xcx_plan* plan, const float* traces, int S, int W, long stride_s,
                       long stride_w, const int* pairs, int P, int on_device);
int xcx_cc1d_stack_finish(xcx_plan* plan, const int* pairs, int P, float* out, long ld_out,
                          int on_device, long long* windows);
/* ... across ranks, in the frequency domain (configs[3]; needs xcx_comm_init on the plan's context).  Every rank
 * calls it with the SAME pair list after accumulating its own station-days: the in-band elements of the ranks'
 * cross-spectrum accumulators are summed with ONE ncclReduceScatter (a plan whitened to 0.1-10 Hz at 50 Hz: 1.19 GB
 * per rank instead of the 2.90 GB of P x L float32 stacks that xcx_cc1d_stack_finish + xcx_stack_reduce move), and
 * each rank inverse-transforms only the ceil(P / nranks) accumulator rows it receives.  out: DEVICE [P][ld_out]; only
 * the rows of the pairs this rank owns are written, owned[i] (P host bytes) = 1 for those pairs (every pair is owned
 * by exactly one rank; which one follows the engine's processing order of the pair list, not the caller's).
 * By linearity the owned rows equal the sum over ranks of what xcx_cc1d_stack_finish would have produced. */
int xcx_cc1d_stack_finish_reduce(xcx_plan* plan, const int* pairs, int P, float* out, long ld_out,
                                 unsigned char* owned, long long* windows);
/* Abandon an open accumulation (after a failed add or an error in the driver loop): clears the
 * accumulator state so that the plan accepts xcx_cc1d_pairs or a new series of adds again. */
int xcx_cc1d_stack_reset(xcx_plan* plan);

/* Software pipelining across batches (station-days): compute the spectra of ALL S x W
 * windows of `traces` (device pointer) on a side stream, into a second spectra buffer,
 * while the pair phase of the previous xcx_cc1d_pairs call is still running.  The next
 * xcx_cc1d_pairs call with the same (traces, S, W, strides, on_device = 1) consumes them
 * instead of transforming again; any other call discards them.  The caller must not
 * modify `traces` until xcx_cc1d_prepare_sync() returns (or the consuming call is issued
 * and the context synchronised).  Needs W >= 2 and S*W spectra <= 24 GB; not for XCX_STACK. */
int xcx_cc1d_prepare(xcx_plan* plan, const float* traces, int S, int W, long stride_s,
                     long stride_w);
int xcx_cc1d_prepare_sync(xcx_plan* plan);

/* Batched two-input form (util/cc.py:22 cc(x, y) / correlate1d(in1, in2)):
 * x, y : [batch][n]; out : [batch][L]. */
int xcx_cc1d(xcx_plan* plan, const float* x, const float* y, int batch,
             float* out, int on_device);

/* number of kernel launches and of pair-windows processed since plan
 * creation (bench bookkeeping) */
int xcx_plan_counters(xcx_plan* plan, long long* launches,
                      long long* pair_windows);

/* Per-kernel timing with HIP events recorded on the plan's stream around
 * every launch (bench.py roofline leg).  enable = 2 clears and starts per-launch timing,
 * 1 times only whole pair phases (no perturbation of the overlapping pipelines), 0 stops;
 * xcx_plan_profile_read synchronises and returns, for kernel_id 0..7
 * (k_stats, k_fwd_rows, k_fwd_cols, k_pair_cols, k_pair_rows, k_pair_dual, k_gram,
 * and pair_phase = wall time of a whole pair phase: its pipelines overlap), the summed
 * duration, the launch count and the units (station-windows / pair-windows)
 * those launches processed. */
int xcx_plan_profile(xcx_plan* plan, int enable);
int xcx_plan_profile_read(xcx_plan* plan, int kernel_id, char* name,
                          size_t name_len, double* total_us,
                          long long* launches, long long* units);

/* ---- row-wise signal ops on [batch][n] float32, in place ----------------
 * norm1d  : x / ||x||_2                       (signal/normalize.py:73-80)
 * demean  : x - mean(x)                       (signal/detrend.py:110)
 * detrend_linear : x - (m t + b), least squares (signal/detrend.py:87-99)
 * taper   : x * w, w = ones with `nmax` samples of a periodic Hann window of
 *           length 2*nmax on both flanks      (signal/window.py:92-106,
 *           signal/taper.py:67-69); wtype: 0 = hann, 1 = obspy cosine
 * unbias  : x * n/(n-|lag|), lags idx_min..idx_max-1 of 2n-1
 *                                             (signal/unbias.py:73-85)
 * onebit  : sign(x)                           (self-defined)
 * sosfiltfilt : zero-phase Butterworth, scipy semantics (odd padding of
 *           3*ntaps samples, sosfilt_zi initial state) given `sos`
 *           [n_sections][6] float64           (signal/filter.py:86-96); n_sections <= XCX_SOS_MAX_SECTIONS
 *           in every SOS entry point (an order-16 band-pass)
 * sosfilt : forward-only cascade from rest (ObsPy Trace.filter,
 *           stream/process.py:279-282; unverified vs obspy)
 */
int xcx_norm1d(xcx_ctx* ctx, float* x, long batch, long n, int on_device);
int xcx_demean(xcx_ctx* ctx, float* x, long batch, long n, int on_device);
int xcx_detrend_linear(xcx_ctx* ctx, float* x, long batch, long n,
                       int on_device);
/* the same two detrends from DEVICE rows src into DEVICE rows dst (dst may be src): an
 * original that has to stay (raw traces, re-processed with other settings) needs no copy
 * first.  linear: 0 = subtract the mean, 1 = subtract the least-squares line. */
int xcx_detrend_to(xcx_ctx* ctx, const float* src, float* dst, long batch, long n, int linear);
/* The same, writing only columns [col0, col1) of dst: the mean / trend are those of the WHOLE row
 * (window-sharded front end: a rank needs the preprocessed samples of its window block only). */
int xcx_detrend_to_cols(xcx_ctx* ctx, const float* src, float* dst, long batch, long n, int linear,
                        long col0, long col1);
int xcx_taper(xcx_ctx* ctx, float* x, long batch, long n, long nmax,
              int wtype, int on_device);
int xcx_unbias(xcx_ctx* ctx, float* x, long batch, long npts, long idx_min,
               long idx_max, int on_device);
/* signal/detrend.py:87-110 with skipna=True (the reference's default): mean (linear = 0) or
 * least-squares line (linear != 0) fitted to the finite samples only; NaN samples stay NaN. */
int xcx_detrend_nan(xcx_ctx* ctx, float* x, long batch, long n, int linear, int on_device);
/* signal/window.py:92-106 for any window type: x[:, :nmax] *= flanks[0:nmax],
 * x[:, -nmax:] *= flanks[nmax:2 nmax] (right flank wins on overlap); flanks is HOST memory
 * (scipy.signal.windows.<wtype>(2 nmax, False) cast to float32). */
int xcx_taper_table(xcx_ctx* ctx, float* x, long batch, long n, const float* flanks, long nmax,
                    int on_device);
/* Head of the post-correlation chain in ONE op (core/postprocess.py:133-136,
 * `cc.unbias().demean().taper(max_length)`): rows are the lag slice [idx_min, idx_max) of CCFs of
 * window length npts; x *= n/(n-|lag|) (if `unbias`), x -= mean(x), Hann (wtype 0) / ObsPy-cosine
 * (wtype 1) flanks of taper_nmax samples (0 = none).  Bit-identical to xcx_unbias + xcx_demean +
 * xcx_taper, two passes over the rows instead of five. */
int xcx_post_prep(xcx_ctx* ctx, float* x, long batch, long npts, long idx_min, long idx_max,
                  int unbias, long taper_nmax, int wtype, int on_device);
int xcx_onebit(xcx_ctx* ctx, float* x, long batch, long n, int on_device);
/* Row scaling and row maxima for the ObsPy-named preprocess ops that are a division by one number per trace
 * (stream/process.py:29-94: `remove_sensitivity` = Trace.remove_sensitivity, `normalize` = Trace.normalize):
 * x[r][:] *= factor[r] (factor: `batch` host doubles; the product is formed in float64 and rounded once), and
 * out[r] = max |x[r][:]| (out: `batch` host floats).  batch <= XCX_ROWS_MAX_BATCH per call. */
int xcx_scale_rows(xcx_ctx* ctx, float* x, long batch, long n, const double* factor, int on_device);
int xcx_absmax_rows(xcx_ctx* ctx, const float* x, long batch, long n, float* out, int on_device);
/* Rotation of two components in place (stream/process.py:70-73 `rotate` = obspy Stream.rotate; 'NE->RT' as used by
 * scripts/cc.py:83-84): (a, b) <- (c4[0] a + c4[1] b, c4[2] a + c4[3] b) on two arrays [batch][n] (both host or both
 * device); products in float64, rounded once.  out[r] of xcx_absmax_rows is NaN when the row holds a NaN
 * (np.abs(x).max() semantics). */
int xcx_rotate2(xcx_ctx* ctx, float* a, float* b, long batch, long n, const double* c4, int on_device);
/* Running-absolute-mean temporal normalisation (Bensen et al. 2007 eq. 1; self-defined,
 * the reference has none): x[t] /= mean(|x[t-h .. t+h]|), the window truncated at the
 * row ends and averaged over the samples it holds; 0 where the mean is 0.  h <= XCX_RAM_MAX_HALF_WIDTH. */
int xcx_ram_normalize(xcx_ctx* ctx, float* x, long batch, long n, long half_width,
                      int on_device);
int xcx_sosfiltfilt(xcx_ctx* ctx, float* x, long batch, long n,
                    const double* sos, int n_sections, int on_device);
int xcx_sosfilt(xcx_ctx* ctx, float* x, long batch, long n, const double* sos,
                int n_sections, int on_device);
/* Columns [col0, col1) of xcx_sosfilt on device rows [batch][n], without filtering the whole row: the
 * recursion starts from rest `warm` samples before col0, where warm (xcx_sosfilt_warmup) is the
 * smallest multiple of 1024 with ||A^warm||_inf <= 2^-64 (A = the cascade's zero-input transition):
 * the start state at col0 then equals the from-rest state to below float64 rounding.  Columns
 * [*first_col, col0) are overwritten with intermediate values; columns outside [*first_col, col1)
 * are untouched.  warm < 0 (not stable) or warm >= col0: the row is filtered from its start. */
int xcx_sosfilt_warmup(const double* sos, int n_sections, long* warm);
int xcx_sosfilt_cols(xcx_ctx* ctx, float* x, long batch, long n, const double* sos, int n_sections,
                     long col0, long col1, long* first_col);
/* ObsPy Trace.filter(..., zerophase=True) (xcorr/stream/process.py:279-282 ->
 * obspy.signal.filter): sosfilt from rest, then sosfilt of the reversed result
 * from rest, reversed back.  No padding and no steady-state start values, so it
 * differs from xcx_sosfiltfilt (scipy.signal.sosfiltfilt) near the row ends. */
int xcx_sosfilt_zerophase(xcx_ctx* ctx, float* x, long batch, long n,
                          const double* sos, int n_sections, int on_device);

/* Signal-to-noise pieces of xcorr/signal/snr.py:107-116 (envelope=False): per
 * row, s = max |x| over the signal mask with its index (first occurrence) and
 * noise_rms = sqrt(mean(x^2)) over the noise mask (signal/rms.py:52-58).
 * Masks are n bytes (host pointers); s, noise_rms, argmax are host arrays of
 * `batch` entries.  snr = s / noise_rms is left to the caller (power / dB). */
int xcx_snr(xcx_ctx* ctx, const float* x, long batch, long n,
            const unsigned char* signal_mask, const unsigned char* noise_mask,
            float* s, float* noise_rms, int* argmax, int on_device);

/* Fractional time shift of every row (xcorr/signal/timeshift.py:90-96, fast=True):
 * n = size - (size & 1) (the reference's rfft drops the last sample of an
 * odd-length axis, signal/fft.py:357); y[r] = irfft(rfft(x[r][:n]) *
 * exp(-2 pi i delay[r] f)) with f = linspace(0, fN, n/2+1) and
 * fN = round(n/2 / ((n-1)/sampling_rate)) (signal/fft.py:358).
 * x: [batch][size], y: [batch][n], delay: [batch] seconds (host pointer).
 * Computed as a period-n circular convolution with the analytic Dirichlet
 * kernel on the cross-correlation engine (no length-n FFT is needed), any n >= 2 up to
 * XCX_MAX_ROW_SAMPLES: rows with 2n-1 <= XCX_MAX_SINGLE_TRANSFORM (n <= 393 216) take three
 * transforms of >= 2n-1 points (9.0 us per 359 998-sample row; rows of a call that share a delay
 * share the kernel's transform: 5.3 us); longer rows (the 1.62 M-lag rows of the production
 * container) run the same convolution as ONE pass of the block decomposition of the segmented
 * plans (blocks of 184 320 samples, cost ~ n * ceil(n / 184 320): 66 us per
 * 1.62 M-sample row, 34 us with shared delays; workspace: see xcx_plan_cc1d, segmented plans). */
int xcx_timeshift(xcx_ctx* ctx, const float* x, float* y, long batch, long size,
                  const double* delay, double sampling_rate, int on_device);

/* timeshift(..., fast=False) of xcorr/signal/timeshift.py:90-96: the fft / ifft path
 * (signal/fft.py:104-113, 239-241).  x, y: [batch][size] (every sample kept);
 * delay: [batch] host doubles (seconds); delta: sample spacing (rounded to 1e-10
 * like the reference's frequency axis).  Even size: the same period-`size` shift
 * as xcx_timeshift up to the frequency-axis rounding.  Odd size: the reference's
 * second fftshift leaves the spectrum rolled by one bin and real() of the result
 * is the shifted row times cos(2 pi m / size); reproduced literally. */
int xcx_timeshift_fft(xcx_ctx* ctx, const float* x, float* y, long batch, long size,
                      const double* delay, double delta, int on_device);

/* Two-dimensional cross-correlation of tiles (xcorr/signal/correlate.py:162-291
 * with norm2d of signal/normalize.py:95-155; the inner call of
 * scripts/timelapse.py:243-268 which keeps max and argmax only).
 * tiles: [T][F][L] float32 (host or device per on_device);  pairs: [P][2] host
 * int32 tile indices (a, b).  For every pair
 *   cc[i][j] = sum_{u,v} a[u + i-(F-1)][v + j-(L-1)] b[u][v],  i < 2F-1, j < 2L-1
 * (the fftshift-ed layout of the reference), divided by ||a||_F ||b||_F when
 * normalize != 0.  peak_val[P] / peak_idx[P][2] (host) receive the maximum and
 * its (i, j), first occurrence in row-major order (numpy.argmax).  full, when
 * non-NULL, receives all [P][2F-1][2L-1] values (host or device per on_device).
 * Computed at the padded size P1 x P2 >= (2F-1) x (2L-1) with P1, P2 powers of
 * two; F <= XCX_CC2D_MAX_F, L <= XCX_CC2D_MAX_L (XCX_EUNSUPPORTED beyond).
 * Every tile is transformed once per call; a pair costs one inverse 2-D
 * transform of the product spectrum (column pass, row pass with the arg-max).
 * The pairs run in passes of whole 8 x 8 tile groups; two passes are in flight
 * on the context's pipeline streams (the column pass of one beside the row
 * pass of the other), the context's stream waits for both before the call
 * returns its results.  Workspace on the context (grow-only): the T padded
 * spectra (8 P1 (P2/2 + 16) bytes each) and two halves of the intermediate,
 * 2.2 GB each when the pair list needs more than one pass. */
int xcx_cc2d(xcx_ctx* ctx, const float* tiles, int T, int F, int L,
             const int* pairs, int P, int normalize, float* peak_val,
             int* peak_idx, float* full, int on_device);

/* Discrete Fourier transforms of ARBITRARY length along the last axis with numpy.fft
 * conventions -- the kernels of xcorr/signal/fft.py (fft :111-113, ifft :239-241,
 * rfft :367-374, irfft :487-494; fftshift / coordinates stay with the caller).
 * Complex data is interleaved float32 (re, im).
 *   xcx_fft / xcx_ifft : in [batch][n] complex -> out [batch][n] complex (ifft scales 1/n)
 *   xcx_rfft           : x [batch][size] real  -> out [batch][n/2+1] complex with
 *                        n = size - (size & 1)  (fft.py:357 drops an odd last sample)
 *   xcx_irfft          : in [batch][m] complex -> x [batch][2(m-1)] real  (fft.py:475)
 * Bluestein's chirp convolution on the cross-correlation engine, any n >= 2 up to XCX_MAX_ROW_SAMPLES,
 * three regimes (even-length real transforms are packed into one complex transform of n/2 points first):
 *   nin + nout - 1 <= XCX_MAX_CHIRP_POINTS : ONE convolution (complex n <= 368 640; packed real n <= 737 280 --
 *       the 359 998-sample lag axis: rfft / irfft 8.8 / 8.3 us per row, hilbert 18 us);
 *   beyond that while 2 max(nin, nout) - 1 <= XCX_MAX_SINGLE_TRANSFORM : the DFT sum split over si input x so output
 *       blocks, si * so <= XCX_MAX_SPLIT_PARTS partial transforms accumulated into the output (cost ~ si * so);
 *   beyond every single transform (day-long rows: rfft of 4.32 M samples = 2.16 M complex points) : the chirp
 *       convolution block-decomposed on the segmented plans, cost ~ n * ceil(n / 368 640) (rfft 159 us, hilbert
 *       326 us per 4.32 M-sample row); workspace = half of the free device memory, at most
 *       XCX_SEG_WORKSPACE_MAX_GB (XCX_SEG_BUDGET_MB overrides), rows beyond it go through in several chunks. */
int xcx_fft(xcx_ctx* ctx, const float* in, float* out, long batch, long n, int on_device);
int xcx_ifft(xcx_ctx* ctx, const float* in, float* out, long batch, long n, int on_device);
int xcx_rfft(xcx_ctx* ctx, const float* x, float* out, long batch, long size, int on_device);
int xcx_irfft(xcx_ctx* ctx, const float* in, float* x, long batch, long m, int on_device);

/* Fourier resampling with ObsPy's Trace.resample semantics (registry entry 'resample' of
 * xcorr/stream/process.py:29-94, chain scripts/cc.py:46-113; unverified vs obspy):
 *   y = irfft(interp(rfft(x) * window_half), num) * num / npts,   x [batch][npts] -> y [batch][num],
 * the spectrum linearly interpolated at bin positions j * ratio (ratio = npts * new_rate /
 * (num * old_rate)), clamped at the last bin like numpy.interp.  window_half: HOST float32
 * [npts/2 + 1] (ifftshift(get_window(w, npts))[:npts/2+1]) or NULL.  Any npts, num >= 2 up to XCX_MAX_ROW_SAMPLES:
 * both transforms follow the three regimes of xcx_fft (a day-long row 50 -> 20 Hz: 0.22 ms). */
int xcx_resample(xcx_ctx* ctx, const float* x, float* y, long batch, long npts, long num,
                 double ratio, const float* window_half, int on_device);

/* Analytic signal of every row, scipy.signal.hilbert semantics (xcorr/signal/hilbert.py:69):
 * x [batch][n] real -> out [batch][n] complex, or with envelope != 0 its modulus
 * [batch][n] real (hilbert.py:120; the `envelope=True` branch of signal/snr.py:109).
 * Any n >= 2 up to XCX_MAX_ROW_SAMPLES; even n as two packed real transforms, every length through the three
 * regimes of xcx_fft (18 us per 359 998-sample row, 326 us per 4.32 M-sample row). */
int xcx_hilbert(xcx_ctx* ctx, const float* x, float* out, long batch, long n, int envelope,
                int on_device);

/* Sample-rate changing preprocess ops of the stream registry
 * (xcorr/stream/process.py:29-94; chain in scripts/cc.py:46-113).  Out of place:
 * x [batch][n] -> y [batch][n_out].  decimate / interpolate are ObsPy methods in the
 * reference (not vendored): documented behaviour restated, unverified vs obspy.
 *   xcx_decimate : optional causal SOS low-pass (n_sections > 0: pass ObsPy's
 *                  lowpass_cheby_2 design), then y[i] = x[i*factor], n_out = ceil(n/factor)
 *   xcx_lanczos  : y[i] = sum_m x[m] K(offset + i*step - m), K(u) = sinc(u) sinc(u/a),
 *                  |u| < a; offset and step in input samples
 *                  (obspy lanczos_interpolation: step = new_dt/old_dt)
 *   xcx_block_rms: y[i] = sqrt(mean(x[i*step : i*step+win]^2)), i < nwin
 *                  (xcorr/stream/running_rms.py:104-111) */
int xcx_decimate(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long factor,
                 const double* sos, int n_sections, int on_device);
int xcx_lanczos(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long n_out,
                double offset, double step, int a, int on_device);
int xcx_block_rms(xcx_ctx* ctx, const float* x, float* y, long batch, long n, long win,
                  long step, long nwin, int on_device);

/* Hop-1 spectrogram of every row: scipy.signal.spectrogram(x, nperseg=win_len,
 * noverlap=win_len-1, nfft, window, detrend, mode='psd', return_onesided=True) as used
 * by xcorr/signal/spectrogram.py:124-141, including its zero columns at the outer
 * edges: x [batch][n] -> S [batch][nfft/2+1][n], column edge + t (edge = rint(win_len/2))
 * holds the segment starting at sample t, t <= n - win_len; other columns are 0.
 * window: win_len float64 values (host); scale: 1/(fs sum w^2) ('density') or
 * 1/(sum w)^2 ('spectrum') -- the one-sided doubling of all bins but DC / Nyquist is
 * applied here; detrend: 0 none, 1 'constant' (scipy default).
 * Sliding-window DFT as a Hankel GEMM on fp32 MFMA; 2 <= win_len <= XCX_SPECTROGRAM_MAX_WIN,
 * nfft <= XCX_SPECTROGRAM_MAX_NFFT. */
int xcx_spectrogram(xcx_ctx* ctx, const float* x, float* S, long batch, long n,
                    int win_len, int nfft, const double* window, double scale,
                    int detrend, int on_device);

/* Segmented spectrogram with any hop: scipy.signal.spectrogram(x, nperseg=win_len,
 * noverlap=win_len-hop, nfft, window, detrend, mode='psd') as called by
 * spectrogram_mtc, xcorr/signal/spectrogram.py:266-276.  x [batch][n] ->
 * S [batch][nfft/2+1][nseg], nseg = (n - win_len)/hop + 1, segment s starts at
 * sample s*hop.  window / scale / detrend as for xcx_spectrogram.  Segments within the limits of xcx_spectrogram
 * take its MFMA kernel with a column stride; longer ones are framed on the device and transformed by the
 * any-length DFT (nfft <= XCX_SPECTROGRAM_SEG_MAX_NFFT). */
int xcx_spectrogram_seg(xcx_ctx* ctx, const float* x, float* S, long batch, long n,
                        int win_len, long hop, int nfft, const double* window,
                        double scale, int detrend, int on_device);

/* ---- multi-GPU time-stack reduction (RCCL over xGMI) --------------------
 * One communicator per process/GPU.  id_bytes is the 128-byte ncclUniqueId
 * created by rank 0 with xcx_comm_unique_id() and distributed by the host
 * launcher (torch.distributed store / file / MPI).
 * xcx_stack_reduce sums `count` floats in place across ranks
 * (ncclAllReduce) or leaves each rank with its count/nranks share
 * (ncclReduceScatter, reduce_scatter != 0; recv = stack + rank*count/nranks).
 * Without a communicator on `ctx` it returns XCX_EINVAL (also for one rank:
 * a single-rank world initialises a 1-rank communicator), so a failed or
 * forgotten xcx_comm_init can never pass a partial stack off as the total.
 */
int xcx_comm_unique_id(char* id_bytes128);
int xcx_comm_init(xcx_ctx* ctx, const char* id_bytes128, int nranks, int rank);
int xcx_comm_destroy(xcx_ctx* ctx);
/* What RCCL itself reports for the communicator of `ctx` (ncclCommCount /
 * ncclCommUserRank), not what the host passed to xcx_comm_init: the launcher's
 * evidence that the reduce really spans WORLD_SIZE ranks.  XCX_EINVAL without a
 * communicator. */
int xcx_comm_info(xcx_ctx* ctx, int* nranks, int* rank);
int xcx_stack_reduce(xcx_ctx* ctx, float* stack, size_t count,
                     int reduce_scatter);

#ifdef __cplusplus
}
#endif
#endif /* XCX_H */
