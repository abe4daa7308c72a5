"""
The driver-timed workload of BASELINE.json configs[2]/[3] as ONE object, so that
``bench.py`` (timing) and ``tests/test_gpu_config3.py`` (full-size parity test)
run exactly the same calls.

Per station-day: demean (raw -> work; raw is kept), cosine taper (5 %, <= 30 s),
Butterworth band-pass 0.1-10 Hz (4 corners, forward ``sosfilt`` from rest:
ObsPy's ``Trace.filter`` default, xcorr/stream/process.py:279-282).  Per
window: one-bit + spectral whitening fused into the forward transforms (no
per-window demean: the band-pass has removed the DC), then the cross-correlation of every pair (xcorr/core/process.py:82-165
batched; one-bit / whitening are self-defined, SURVEY TL;DR).
"""
import numpy as np

from . import ops
from .engine import CCPlan

__all__ = ['StationDay', 'FS', 'BAND', 'WHITEN']

FS = 50.0
BAND = (0.1, 10.0)
WHITEN = (FS, 0.1, 10.0, 0.05)        # sampling rate, fmin, fmax, raised-cosine flank width (Hz)
TAPER = (0.05, 30.0)                  # max_percentage, max_length (s)


def band_sos():
    import scipy.signal
    return scipy.signal.butter(4, list(BAND), btype='bandpass', output='sos', fs=FS)


def taper_samples(T):
    return min(int(TAPER[0] * T), int(TAPER[1] * FS), T // 2)


class StationDay:
    """S day-long traces resident in HBM + the plan that correlates them.

    step() = one pass of the hot path over the station-day (what bench.py times).
    """

    def __init__(self, ctx, S, T, n, hop, pairs, index_min=0, index_max=None, preprocess=True,
                 stack=False, batch=None, w_first=0, W=None):
        self.ctx, self.S, self.T, self.n, self.hop = ctx, int(S), int(T), int(n), int(hop)
        self.pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        self.P = len(self.pairs)
        self.W_all = (self.T - self.n) // self.hop + 1
        self.w_first = int(w_first)
        self.W = int(W) if W is not None else self.W_all - self.w_first
        self.preprocess = bool(preprocess)
        self.stack = bool(stack)
        self.plan = CCPlan(ctx, n, index_min, index_max, normalize=True,
                           onebit=self.preprocess, whiten=WHITEN if self.preprocess else None,
                           stack=stack, batch=batch)
        self.L = self.plan.L
        self.ld = (self.L + 31) // 32 * 32           # 128-byte aligned output rows
        self.raw = None
        self.work = None
        self.nrows = self.P if stack else self.P * self.W
        self.out = ctx.empty((self.nrows, self.ld))
        self.sos = band_sos() if self.preprocess else None
        self.taper_n = taper_samples(self.T)
        self.warm = None

    def load(self, raw_host):
        """Make a day of raw traces resident (outside any timed region)."""
        raw_host = np.ascontiguousarray(raw_host, dtype=np.float32)
        assert raw_host.shape == (self.S, self.T)
        if self.raw is None:
            self.raw = self.ctx.array(raw_host)
            self.work = self.ctx.empty((self.S, self.T)) if self.preprocess else self.raw
        else:
            self.raw.copy_from(raw_host)
        self.view = self.work.columns_from(self.w_first * self.hop) if self.w_first else self.work
        return self

    def front(self, ctx=None):
        """Preprocess of the station-day (raw is kept: every step starts from it)."""
        ctx = ctx or self.ctx
        if not self.preprocess:
            return
        c0, c1 = self.w_first * self.hop, (self.w_first + self.W - 1) * self.hop + self.n
        if c0 == 0 and c1 >= self.T:
            ops.demean(ctx, self.raw, out=self.work)          # raw -> work: no separate copy of the day
            ops.taper(ctx, self.work, self.taper_n, wtype='cosine')
            ops.sosfilt(ctx, self.work, self.sos)
            return
        # A block of windows (bench.py --shard windows): only the samples the block needs.  The day mean
        # is still that of the whole row (one read of the day); the band-pass runs over the block plus the
        # warm-up after which its state equals the from-rest state to below float64 rounding (ops.sosfilt).
        if self.warm is None:
            self.warm = ops.sosfilt_warmup(self.sos)
        lo = 0 if (self.warm < 0 or self.warm >= c0) else c0 - self.warm
        ops.demean(ctx, self.raw, out=self.work, cols=(lo, c1))
        ops.taper(ctx, self.work, self.taper_n, wtype='cosine')   # touches the day's two flanks only
        ops.sosfilt(ctx, self.work, self.sos, cols=(c0, c1))

    def correlate(self):
        self.plan.pairs(self.view, self.pairs, self.W, self.hop, out=self.out, ld_out=self.ld,
                        stride_s=self.T)

    def step(self):
        self.front()
        self.correlate()

    def rows(self, p, w):
        """CCF of pair index p, local window w (host copy of the L retained lags)."""
        r = p if self.stack else p * self.W + w
        return self.out.rows_to_host(r, 1, self.L)[0]

    def close(self):
        self.plan.close()
        for b in (self.out, self.work, self.raw):
            if b is not None:
                b.free()
