"""
oracle/ref_numpy.py -- CPU restatement of xcorr's cross-correlation hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The shipped path (``xcorr_amd``) never routes
through this module; it fails loudly when the HIP library is missing.

Every function restates, on plain ``[batch..., n]`` numpy arrays with the core
dimension last, what the cited reference lines compute (paths relative to the
reference checkout, ``xcorr/...``).  All arithmetic is done in float64, which
is what the reference produced on its own numpy (<2) stack where ``np.fft``
promotes float32 input to complex128 ("dtype drift").

Pinning: ``cc``/``lag``/``weight`` are checked against the *imported*
reference file ``xcorr/util/cc.py`` by ``oracle/make_golden.py`` (run in the
dev container, where the reference is mounted) and frozen into
``tests/golden/*.npz``.  The reference ships no golden vectors of its own
(its ``tests/`` are stubs), so every other function here is pinned only by
scipy/numpy themselves: "parity unpinned" by the reference for those rows.
The ops marked SELF-DEFINED (one-bit, whitening, stack) do not exist in the
reference at all.
"""
import numpy as np
import scipy.signal
import scipy.stats

__all__ = [
    'cc', 'cc_fair', 'lag', 'weight', 'weight_full', 'lag_indices',
    'norm1d', 'correlate1d', 'demean', 'detrend_linear', 'window', 'taper',
    'butter_sos', 'sosfiltfilt', 'sosfilt', 'sosfilt_zerophase', 'unbias', 'timeshift',
    'onebit', 'whiten_weights', 'whiten_spectrum', 'whiten_norm',
    'cosine_taper_obspy', 'clip_lag_indices', 'cc_pairs', 'stack',
    'next_fast_len', 'synthetic_traces', 'window_starts', 'all_pairs',
    'postprocess_chain', 'snr',
]


# --------------------------------------------------------------------------
# a1  util/cc.py:22-79   cc()
# --------------------------------------------------------------------------
def cc(x, y, normalize=True, pad=True, unbiased=False, dtype=None):
    """Faithful restatement of ``xcorr.util.cc.cc`` (util/cc.py:60-79).

    Front zero-pad to 2n-1 (:66-71), complex fft of both (:75), divide by the
    product of the L2 norms (:76-77), ``fftshift(real(ifft))`` (:78).
    ``unbiased=True`` multiplies by ``weight(nn, False)`` (:79) -- in the
    reference that line raises (shape 2n-3 vs 2n-1); here the
    closed form over all 2n-1 lags is used (``weight_full``).
    Raises AssertionError on length/dtype mismatch exactly like :62-65.
    """
    x = np.asarray(x)
    y = np.asarray(y)
    n = len(x)
    dtype = dtype or x.dtype
    assert n == len(y), "Vectors `x` and `y` should have the same length!"
    assert x.dtype == y.dtype, "Vectors `x` and `y` should have the same dtype!"
    if pad:
        nn = 2 * n - 1
        xx = np.zeros(nn, dtype=np.float64)
        xx[nn - n:] = x
        yy = np.zeros(nn, dtype=np.float64)
        yy[nn - n:] = y
    else:
        nn = n
        xx = x.astype(np.float64)
        yy = y.astype(np.float64)
    fg = np.fft.fft(xx) * np.conjugate(np.fft.fft(yy))
    if normalize:
        fg = fg / (np.linalg.norm(xx) * np.linalg.norm(yy))
    Rxy = np.fft.fftshift(np.real(np.fft.ifft(fg)))
    if unbiased:
        Rxy = Rxy * weight_full(n)
    return Rxy.astype(dtype)


def next_fast_len(m):
    """Smallest 5-smooth even integer >= m (the 'fair' CPU FFT length)."""
    m = int(m)
    while True:
        k = m
        for p in (2, 3, 5):
            while k % p == 0:
                k //= p
        if k == 1 and m % 2 == 0:
            return m
        m += 1


def cc_fair(x, y, normalize=True, nfft=None, dtype=None):
    """Same linear CC as ``cc(pad=True)`` but through rfft/irfft at a fast
    length N >= 2n-1 ('what a sane CPU path costs').
    out[j] = c[(j-(n-1)) mod N],  j = 0..2n-2.
    """
    x = np.asarray(x)
    y = np.asarray(y)
    n = x.shape[-1]
    dtype = dtype or x.dtype
    N = nfft or next_fast_len(2 * n - 1)
    X = np.fft.rfft(x.astype(np.float64), N, axis=-1)
    Y = np.fft.rfft(y.astype(np.float64), N, axis=-1)
    c = np.fft.irfft(X * np.conj(Y), N, axis=-1)
    out = np.concatenate([c[..., N - (n - 1):], c[..., :n]], axis=-1)
    if normalize:
        den = (np.linalg.norm(x.astype(np.float64), axis=-1, keepdims=True) *
               np.linalg.norm(y.astype(np.float64), axis=-1, keepdims=True))
        out = out / den
    return out.astype(dtype)


# --------------------------------------------------------------------------
# a2  util/cc.py:82-108   lag()
# --------------------------------------------------------------------------
def lag(n, delta, pad=True):
    """``fftshift(fftfreq(nn, 1/(nn*delta)))`` (util/cc.py:107-108)."""
    nn = n * 2 - 1 if pad else n
    return np.fft.fftshift(np.fft.fftfreq(nn, 1 / (nn * delta)))


# --------------------------------------------------------------------------
# a3  util/cc.py:111-141  weight();  signal/unbias.py:30-145
# --------------------------------------------------------------------------
def weight(n, pad=True, clip=None):
    """Literal restatement of util/cc.py:136-141 (with ``np.int`` -> ``int``).
    NOTE the reference quirk: the arange yields nn-2 entries, i.e. 2n-3 for
    pad=True called with the *input* length n (lags -(n-1)..(n-3))."""
    nn = n * 2 - 1 if pad else n
    n = int((nn + 1) / 2)
    w = n / (n - np.abs(np.arange(1 - n, nn - n - 1, 1, np.float64)))
    if clip is not None:
        w[np.where(w > clip)] = clip
    return w


def weight_full(n, clip=None):
    """Closed form over all 2n-1 lags: w[j] = n / (n - |j-(n-1)|).
    Agrees with ``weight(n)`` wherever the reference is defined (first 2n-3)."""
    k = np.arange(-(n - 1), n, dtype=np.float64)
    w = n / (n - np.abs(k))
    if clip is not None:
        w[w > clip] = clip
    return w


def lag_indices(n, delta, lag_first, lag_last):
    """util/time.py:104-108 update_lag_indices."""
    lag_max = (n - 1) * delta
    srate = 1.0 / delta
    imin = int(round((lag_first + lag_max) * srate))
    imax = int(round((lag_last + lag_max) * srate + 1))
    return imin, imax


def clip_lag_indices(n, delta, clip_lag):
    """core/init.py:176-196: argmin |lag-a| : argmin |lag-b| (end exclusive)."""
    lg = lag(n, delta)
    if clip_lag is None:
        return 0, 2 * n - 1
    nmin = int(np.argmin(abs(lg - clip_lag[0])))
    nmax = int(np.argmin(abs(lg - clip_lag[1])))
    return nmin, nmax


def unbias(x, n, index_min=0, index_max=None):
    """signal/unbias.py:73-85: ``x * w[index_min:index_max].astype(x.dtype)``."""
    x = np.asarray(x)
    index_max = 2 * n - 1 if index_max is None else index_max
    w = weight_full(n)[index_min:index_max]
    return x * w.astype(x.dtype)


# --------------------------------------------------------------------------
# a5  signal/normalize.py:73-80   norm1d
# --------------------------------------------------------------------------
def norm1d(x):
    """x / ||x||_2 along the last axis (signal/normalize.py:73-80)."""
    x = np.asarray(x)
    nrm = np.linalg.norm(x.astype(np.float64), axis=-1, keepdims=True)
    with np.errstate(divide='ignore', invalid='ignore'):
        return (x / nrm).astype(x.dtype)


# --------------------------------------------------------------------------
# a4  signal/correlate.py:97-113   correlate1d
# --------------------------------------------------------------------------
def correlate1d(in1, in2, normalize=True, dtype=None):
    """norm1d each (:103-105), fft(n=2N-1) end-padded (:109-110), product with
    conj, ``fftshift(real(ifft))`` (:111-112).  float64 internally."""
    in1 = np.asarray(in1)
    in2 = np.asarray(in2)
    if in1.shape != in2.shape:
        raise ValueError('in1 and in2 should have the same shape!')
    dtype = dtype or in1.dtype
    npad = 2 * in1.shape[-1] - 1
    f = in1.astype(np.float64)
    g = in2.astype(np.float64)
    if normalize:
        f = f / np.linalg.norm(f, axis=-1, keepdims=True)
        g = g / np.linalg.norm(g, axis=-1, keepdims=True)
    F = np.fft.fft(f, n=npad, axis=-1)
    G = np.fft.fft(g, n=npad, axis=-1)
    out = np.fft.fftshift(np.real(np.fft.ifft(F * np.conjugate(G), axis=-1)),
                          axes=-1)
    return out.astype(dtype)


# --------------------------------------------------------------------------
# a7  signal/detrend.py:87-110
# --------------------------------------------------------------------------
def demean(x):
    """``x - x.mean(dim)`` (signal/detrend.py:110)."""
    x = np.asarray(x)
    return (x - x.astype(np.float64).mean(axis=-1, keepdims=True)
            ).astype(x.dtype)


def detrend_linear(x, t=None):
    """Per row ``scipy.stats.linregress(t, y)``; subtract ``m*t+b``
    (signal/detrend.py:87-99).  ``t`` defaults to 0..n-1 (any affine time
    axis yields the same residual)."""
    x = np.asarray(x)
    n = x.shape[-1]
    t = np.arange(n, dtype=np.float64) if t is None else np.asarray(t, np.float64)

    def _row(y):
        m, b, *_ = scipy.stats.linregress(t, y)
        return y - (m * t + b)
    return np.apply_along_axis(_row, -1, x.astype(np.float64)).astype(x.dtype)


# --------------------------------------------------------------------------
# a8  signal/window.py:74-106, signal/taper.py:67-69
# --------------------------------------------------------------------------
def window(npts, sampling_rate, wtype='hann', max_percentage=None,
           max_length=None):
    """``nmax`` rule (window.py:74-90), periodic ``wtype(2*nmax, False)``
    (:92), both flanks always (:105-106; ``side`` is ignored there)."""
    nmax = 0.5 * npts
    if max_percentage:
        nmax = max_percentage * npts
    if max_length:
        nmax = max_length * sampling_rate
    nmax = int(round(nmax))
    win = getattr(scipy.signal.windows, wtype)(nmax * 2, False)
    w = np.ones(npts, dtype=np.float64)
    w[0:nmax] = win[0:nmax]
    w[-nmax:] = win[-nmax:]
    return w


def taper(x, sampling_rate, wtype='hann', max_percentage=None,
          max_length=None):
    """``x * w.astype(x.dtype)`` (signal/taper.py:67-69)."""
    x = np.asarray(x)
    w = window(x.shape[-1], sampling_rate, wtype, max_percentage, max_length)
    return x * w.astype(x.dtype)


# --------------------------------------------------------------------------
# a9  signal/filter.py:86-96
# --------------------------------------------------------------------------
def butter_sos(order, frequency, btype, sampling_rate):
    """``scipy.signal.butter(N, Wn, btype, output='sos', fs=)`` (:86-92)."""
    return scipy.signal.butter(N=order, Wn=frequency, btype=btype,
                               output='sos', fs=sampling_rate)


def sosfiltfilt(sos, x):
    """``sosfiltfilt(sos, x, axis=-1).astype(x.dtype)`` (:95-96)."""
    x = np.asarray(x)
    return scipy.signal.sosfiltfilt(sos, x, axis=-1).astype(x.dtype)


def sosfilt(sos, x):
    """Forward-only cascade from rest (what ObsPy ``Trace.filter`` runs with
    ``zerophase=False``; a13, unverified vs obspy)."""
    x = np.asarray(x)
    return scipy.signal.sosfilt(sos, x.astype(np.float64), axis=-1
                                ).astype(x.dtype)


# --------------------------------------------------------------------------
# a10 signal/timeshift.py:90-96 with signal/fft.py:357-358,475-479
# --------------------------------------------------------------------------
def timeshift(x, delay, sampling_rate):
    """rfft (drops the last sample when the length is odd, fft.py:357),
    multiply by ``exp(-2 pi i delay f)`` with ``f = linspace(0, fN, n//2+1)``
    and ``fN = round(n//2 / (t[n-1]-t[0]))`` (fft.py:358; integer Hz), irfft.
    ``delay`` broadcasts over the leading dims (scalar or array [...])."""
    x = np.asarray(x)
    size = x.shape[-1]
    n = size - 1 if size & 1 else size
    delta = 1.0 / sampling_rate
    fN = np.round(n // 2 / ((n - 1) * delta), decimals=0)
    f = np.linspace(0., fN, n // 2 + 1)
    X = np.fft.rfft(x[..., :n].astype(np.float64), n=n, axis=-1)
    delay = np.asarray(delay, dtype=np.float64)[..., None]
    X = X * np.exp(-1j * 2 * np.pi * delay * f)
    return np.fft.irfft(X, n=n, axis=-1).astype(x.dtype)


def timeshift_fft(x, delay, delta):
    """fast=False (signal/timeshift.py:90-96): ``X = fftshift(fft(x))`` on
    ``f = fftshift(fftfreq(n, round(delta, 10)))`` (fft.py:104-113), phase ramp,
    ``real(ifft(fftshift(X)))`` (fft.py:239-241 -- fftshift again, literally:
    for odd n that leaves the spectrum rolled by one bin)."""
    x = np.asarray(x)
    n = x.shape[-1]
    d = np.round(np.fabs(delta), decimals=10)
    f = np.fft.fftshift(np.fft.fftfreq(n, d=d))
    X = np.fft.fftshift(np.fft.fft(x.astype(np.complex128), n=n, axis=-1), axes=-1)
    delay = np.asarray(delay, dtype=np.float64)[..., None]
    X = X * np.exp(-1j * 2 * np.pi * delay * f)
    y = np.fft.ifft(np.fft.fftshift(X, axes=-1), n=n, axis=-1)
    return np.real(y).astype(x.dtype)


# --------------------------------------------------------------------------
# signal/snr.py:107-116, signal/rms.py:52-58  (envelope=False)
# --------------------------------------------------------------------------
def snr(x, signal, noise):
    """(snr, s, n, argmax): n = sqrt(mean(x[noise]^2)), s = max |x[signal]| and
    its index along the full axis (first occurrence, like ``argmax``)."""
    x = np.asarray(x, dtype=np.float64)
    signal = np.asarray(signal, bool)
    noise = np.asarray(noise, bool)
    n = np.sqrt(np.mean(np.square(x[..., noise]), axis=-1))
    xs = np.abs(x[..., signal])
    am = xs.argmax(axis=-1)
    s = np.take_along_axis(xs, am[..., None], axis=-1)[..., 0]
    idx = np.flatnonzero(signal)[am]
    return s / n, s, n, idx


# --------------------------------------------------------------------------
# SELF-DEFINED ops (no reference; Bensen et al. 2007 conventions)
# --------------------------------------------------------------------------
def onebit(x):
    """sign(x) in {-1, 0, +1}."""
    x = np.asarray(x)
    return np.sign(x).astype(x.dtype)


def ram_normalize(x, half_width):
    """Running-absolute-mean normalisation (Bensen et al. 2007 eq. 1): x / mean(|x|) over
    a centred window of 2h+1 samples, truncated (and re-counted) at the ends."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    c = np.concatenate([np.zeros(x.shape[:-1] + (1,)), np.cumsum(np.abs(x), axis=-1)], axis=-1)
    t = np.arange(n)
    lo = np.maximum(t - half_width, 0)
    hi = np.minimum(t + half_width, n - 1)
    w = (c[..., hi + 1] - c[..., lo]) / (hi - lo + 1)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(w > 0, x / w, 0.0)


def whiten_weights(nfft, sampling_rate, fmin, fmax, ftaper):
    """Band weight g(f) on the rfft grid of length nfft: 1 inside
    [fmin, fmax], raised-cosine ramps of width ``ftaper`` Hz outside, 0
    beyond.  Returned as float64[nfft//2+1]."""
    f = np.arange(nfft // 2 + 1, dtype=np.float64) * (sampling_rate / nfft)
    g = np.zeros_like(f)
    g[(f >= fmin) & (f <= fmax)] = 1.0
    if ftaper > 0:
        lo = (f >= fmin - ftaper) & (f < fmin)
        g[lo] = 0.5 * (1 - np.cos(np.pi * (f[lo] - (fmin - ftaper)) / ftaper))
        hi = (f > fmax) & (f <= fmax + ftaper)
        g[hi] = 0.5 * (1 + np.cos(np.pi * (f[hi] - fmax) / ftaper))
    return g


def whiten_spectrum(X, g):
    """X[k] -> g[k] * X[k] / |X[k]|  (0 where |X| == 0)."""
    a = np.abs(X)
    with np.errstate(divide='ignore', invalid='ignore'):
        Xw = np.where(a > 0, X / a, 0.0)
    return Xw * g


def cosine_taper_obspy(npts, sampling_rate, max_percentage=0.05,
                       max_length=None):
    """ObsPy ``Trace.taper(type='cosine', max_percentage, max_length)``
    restated from its documented behaviour (a13; obspy is a third-party
    dependency absent here, ``obspy>=1.2`` unpinned in setup.cfg:41 --
    UNVERIFIED vs obspy): ``wlen = min(npts//2, int(p*npts),
    int(max_length*fs))`` samples per flank; flank i = 0..wlen-1 is
    ``0.5*(1-cos(pi*i/(wlen-1)))`` (``cosine_taper(2*wlen+1, p=1.0)`` first
    ``wlen`` samples), mirrored on the right; ones between."""
    wlen = min(npts // 2, int(max_percentage * npts))
    if max_length is not None:
        wlen = min(wlen, int(max_length * sampling_rate))
    w = np.ones(npts, dtype=np.float64)
    if wlen > 1:
        i = np.arange(wlen, dtype=np.float64)
        ramp = 0.5 * (1.0 - np.cos(np.pi * i / (wlen - 1)))
        w[:wlen] = ramp
        w[npts - wlen:] = ramp[::-1]
    elif wlen == 1:
        w[0] = 0.0
        w[-1] = 0.0
    return w


def whiten_norm(g, nfft):
    """L2 norm of a real length-nfft signal whose rfft magnitudes are g[k]
    (Parseval): sqrt(sum_k c_k g_k^2 / nfft), c_k = 1 at DC/Nyquist else 2."""
    c = np.full(len(g), 2.0)
    c[0] = 1.0
    if nfft % 2 == 0:
        c[-1] = 1.0
    return float(np.sqrt((c * np.asarray(g, np.float64) ** 2).sum() / nfft))


# --------------------------------------------------------------------------
# Batched pair x window engine semantics (core/process.py:82-165 hot loop,
# signal/correlate.py:108-113), the thing the GPU engine is compared with.
# --------------------------------------------------------------------------
def window_starts(T, n, hop):
    """Whole windows inside one record of T samples."""
    return np.arange(0, T - n + 1, hop, dtype=np.int64)


def all_pairs(S):
    """Upper triangle i<j in row-major order."""
    return np.array([(i, j) for i in range(S) for j in range(i + 1, S)],
                    dtype=np.int32).reshape(-1, 2)


def cc_pairs(traces, pairs, n, hop, normalize=True, index_min=0,
             index_max=None, onebit_=False, whiten=None, nfft=None,
             demean_=False, dtype=np.float32):
    """CC of every (pair, window): out[p, w, :] = cc(x_i^w, x_j^w)[imin:imax].

    Per window: optional demean -> optional one-bit -> rfft(N) -> optional
    whitening (``whiten`` = g[N//2+1]) -> X_i conj(X_j) -> irfft -> rotate to
    lags -(n-1)..(n-1) -> divide by ||x_i|| ||x_j|| (time-domain norms of the
    one-bit / demeaned windows; with whitening the norm is ``whiten_norm``,
    the Parseval norm of the ideal whitened spectrum |X|=g) -> slice.
    Equals ``cc()`` to float64 rounding when no SELF-DEFINED op is enabled.
    """
    traces = np.asarray(traces)
    S, T = traces.shape
    index_max = 2 * n - 1 if index_max is None else index_max
    N = nfft or next_fast_len(2 * n - 1)
    starts = window_starts(T, n, hop)
    W = len(starts)
    out = np.empty((len(pairs), W, index_max - index_min), dtype=dtype)
    for w, s0 in enumerate(starts):
        xw = traces[:, s0:s0 + n].astype(np.float64)
        if demean_:
            xw = xw - xw.mean(axis=-1, keepdims=True)
        if onebit_:
            xw = np.sign(xw)
        X = np.fft.rfft(xw, N, axis=-1)
        if whiten is not None:
            X = whiten_spectrum(X, whiten)
            nrm = np.full(S, whiten_norm(whiten, N))
        else:
            nrm = np.linalg.norm(xw, axis=-1)
        for p, (i, j) in enumerate(pairs):
            c = np.fft.irfft(X[i] * np.conj(X[j]), N)
            r = np.concatenate([c[N - (n - 1):], c[:n]])
            if normalize:
                with np.errstate(divide='ignore', invalid='ignore'):
                    r = r / (nrm[i] * nrm[j])
            out[p, w] = r[index_min:index_max]
    return out


def stack(ccf, mean=False):
    """Time-stack over the window axis: sum (or mean) of cc over windows."""
    s = np.asarray(ccf, dtype=np.float64).sum(axis=-2)
    if mean:
        s = s / np.asarray(ccf).shape[-2]
    return s.astype(np.asarray(ccf).dtype)


def postprocess_chain(ccf, n, sampling_rate, index_min, index_max, delay,
                      frequency=3., order=2, taper1=5., taper2=1.5):
    """core/postprocess.py:133-141 on a [..., L] array:
    unbias -> demean -> taper(max_length=5.) -> timeshift(-delay) ->
    highpass sosfiltfilt -> taper(max_length=1.5)."""
    y = unbias(ccf, n, index_min, index_max)
    y = demean(y)
    y = taper(y, sampling_rate, max_length=taper1)
    y = timeshift(y, delay, sampling_rate)
    sos = butter_sos(order, frequency, 'highpass', sampling_rate)
    y = sosfiltfilt(sos, y)
    y = taper(y, sampling_rate, max_length=taper2)
    return y


# --------------------------------------------------------------------------
# Synthetic inputs -- shared by tests, bench and cpu_baseline
# --------------------------------------------------------------------------
def synthetic_traces(S, T, day=0, dtype=np.float32):
    """x_s[t] = 0.5*c[t+25*(S-1-s)] + N_s(0,1)[t] + (3 + 1e-6 t); common signal
    with 0.5 s moveout per station at 50 Hz."""
    c = np.random.default_rng(day * 10000).standard_normal(T + 25 * S)
    t = np.arange(T, dtype=np.float64)
    out = np.empty((S, T), dtype=dtype)
    for s in range(S):
        nse = np.random.default_rng(day * 10000 + 1000 + s).standard_normal(T)
        off = 25 * (S - 1 - s)
        out[s] = (0.5 * c[off:off + T] + nse + (3.0 + 1e-6 * t)).astype(dtype)
    return out


def norm2d(x):
    """x / ||x||_F over the last two axes (xcorr/signal/normalize.py:95-155)."""
    x = np.asarray(x, dtype=np.float64)
    return x / np.linalg.norm(x, ord='fro', axis=(-2, -1), keepdims=True)


def correlate2d(in1, in2, normalize=True):
    """Two-dimensional linear cross-correlation over the last two axes
    (xcorr/signal/correlate.py:230-246): zero-pad to (2F-1, 2L-1), F conj(G),
    real inverse, fftshift of both axes.  float64."""
    f = np.asarray(in1, dtype=np.float64)
    g = np.asarray(in2, dtype=np.float64)
    assert f.shape == g.shape
    if normalize:
        f, g = norm2d(f), norm2d(g)
    npad = [2 * f.shape[-2] - 1, 2 * f.shape[-1] - 1]
    F = np.fft.fft2(f, s=npad, axes=(-2, -1))
    G = np.fft.fft2(g, s=npad, axes=(-2, -1))
    return np.fft.fftshift(np.real(np.fft.ifft2(F * np.conjugate(G), axes=(-2, -1))), axes=(-2, -1))


def correlate2d_direct(f, g):
    """The defining double sum of correlate2d (no FFT), for tiny tiles:
    cc[i][j] = sum_{u,v} f[u + i-(F-1)][v + j-(L-1)] g[u][v]."""
    f = np.asarray(f, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    Fn, Ln = f.shape
    out = np.zeros((2 * Fn - 1, 2 * Ln - 1))
    for i in range(2 * Fn - 1):
        di = i - (Fn - 1)
        u0, u1 = max(0, -di), min(Fn, Fn - di)
        for j in range(2 * Ln - 1):
            dj = j - (Ln - 1)
            v0, v1 = max(0, -dj), min(Ln, Ln - dj)
            out[i, j] = np.sum(f[u0 + di:u1 + di, v0 + dj:v1 + dj] * g[u0:u1, v0:v1])
    return out


# -- a6: labelled FFT kernels (xcorr/signal/fft.py) ---------------------------
def fft_kernel(f):
    """`_fft` of signal/fft.py:111-113: fftshift(fft(f, n=size)) in complex128."""
    f = np.asarray(f).astype(np.complex128)
    return np.fft.fftshift(np.fft.fft(f, n=f.shape[-1], axis=-1), axes=-1)


def ifft_kernel(F):
    """`_ifft` of signal/fft.py:239-241: ifft(fftshift(F), n=size)."""
    F = np.fft.fftshift(np.asarray(F).astype(np.complex128), axes=-1)
    return np.fft.ifft(F, n=F.shape[-1], axis=-1)


def rfft_kernel(x):
    """signal/fft.py:357, 367-374: rfft over n = size - (size & 1) samples, float64."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1] - 1 if x.shape[-1] & 1 else x.shape[-1]
    return np.fft.rfft(x, n=n, axis=-1)


def irfft_kernel(X):
    """signal/fft.py:475, 487-494: irfft with n = 2 * size - 2."""
    return np.fft.irfft(np.asarray(X).astype(np.complex128), n=None, axis=-1)


def fft_freq(size, delta):
    """frequency coordinate of `fft` (signal/fft.py:104-108)."""
    return np.fft.fftshift(np.fft.fftfreq(size, d=np.round(abs(delta), decimals=10)))


def rfft_freq(coord):
    """frequency coordinate of `rfft` (signal/fft.py:357-359): integer-rounded Nyquist."""
    coord = np.asarray(coord, dtype=np.float64)
    n = coord.size - 1 if coord.size & 1 else coord.size
    fN = np.round(n // 2 / (coord[n - 1] - coord[0]), decimals=0)
    return np.linspace(0., fN, n // 2 + 1)


def irfft_time(freq, zero=0.0):
    """time coordinate of `irfft` (signal/fft.py:475-479)."""
    n = 2 * len(freq) - 2
    return np.linspace(0., (n - 1) / 2 / freq[-1], n) + zero


def hilbert(x):
    """scipy.signal.hilbert along the last axis (xcorr/signal/hilbert.py:69), float64."""
    import scipy.signal
    return scipy.signal.hilbert(np.asarray(x, dtype=np.float64), axis=-1)


def envelope(x):
    """|hilbert(x)| (xcorr/signal/hilbert.py:120)."""
    return np.abs(hilbert(x))


def snr_envelope(x, signal, noise):
    """snr with envelope=True (signal/snr.py:107-113): noise rms of x, peak of |hilbert(x)|."""
    _, _, n, _ = snr(x, signal, noise)
    _, s, _, idx = snr(envelope(x), signal, noise)
    return s / n, s, n, idx


# -- a13: ObsPy-backed stream ops, restated (unverified vs obspy) ------------------
def cheby2_lowpass_sos(freq, sampling_rate, maxorder=12):
    """ObsPy lowpass_cheby_2 design used by Trace.decimate (documented behaviour)."""
    nyquist = sampling_rate * 0.5
    rp, rs, order = 1, 96, 1e99
    ws = min(freq / nyquist, 1.0)
    wp = freq / nyquist
    wn = ws
    while order > maxorder:
        wp = wp * 0.99
        order, wn = scipy.signal.cheb2ord(wp, ws, rp, rs, analog=0)
    z, p, k = scipy.signal.cheby2(order, rs, wn, btype='low', analog=0, output='zpk')
    return scipy.signal.zpk2sos(z, p, k)


def sosfilt_zerophase(sos, x):
    """ObsPy ``Trace.filter(..., zerophase=True)`` (obspy.signal.filter, documented behaviour):
    ``firstpass = sosfilt(sos, data); return sosfilt(sos, firstpass[::-1])[::-1]`` -- both passes
    from rest, no padding (NOT scipy.signal.sosfiltfilt)."""
    x = np.asarray(x, dtype=np.float64)
    first = scipy.signal.sosfilt(sos, x, axis=-1)
    return scipy.signal.sosfilt(sos, first[..., ::-1], axis=-1)[..., ::-1]


def decimate(x, sampling_rate, factor, no_filter=False):
    """Trace.decimate(factor): causal Chebyshev-II low-pass, then x[::factor]."""
    x = np.asarray(x, dtype=np.float64)
    if not no_filter:
        x = scipy.signal.sosfilt(cheby2_lowpass_sos(0.5 * sampling_rate / factor, sampling_rate), x, axis=-1)
    return x[..., ::factor]


def resample_fourier(x, sampling_rate, new_sampling_rate, window='hann'):
    """ObsPy ``Trace.resample(sampling_rate, window, no_filter=True)`` restated with numpy's
    rfft/irfft (ObsPy works on scipy.fftpack's packed real format; same numbers): spectrum times
    ``ifftshift(get_window(window, npts))[:npts//2+1]``, ``np.interp`` of real and imaginary parts
    onto the new frequency grid, ``irfft(., num) * num / npts`` with ``num = int(npts / factor)``.
    Documented behaviour, UNVERIFIED vs obspy (not installed here)."""
    x = np.asarray(x, dtype=np.float64)
    npts = x.shape[-1]
    factor = sampling_rate / float(new_sampling_rate)
    num = int(npts / factor)
    X = np.fft.rfft(x, axis=-1)
    if window is not None:
        X = X * np.fft.ifftshift(scipy.signal.get_window(window, npts))[:npts // 2 + 1]
    df = 1.0 / (npts / sampling_rate)
    f = df * np.arange(0, npts // 2 + 1)
    large_f = (1.0 / num * new_sampling_rate) * np.arange(0, num // 2 + 1)
    flat = X.reshape(-1, X.shape[-1])
    Y = np.stack([np.interp(large_f, f, r.real) + 1j * np.interp(large_f, f, r.imag) for r in flat])
    y = np.fft.irfft(Y, n=num, axis=-1) * (float(num) / float(npts))
    return y.reshape(x.shape[:-1] + (num,))


def lanczos_interpolation(x, offset, step, n_out, a):
    """y[i] = sum_m x[m] sinc(p-m) sinc((p-m)/a), |p-m| < a, p = offset + i*step
    (obspy.signal.interpolation.lanczos_interpolation, documented behaviour)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[-1]
    p = offset + np.arange(n_out) * step
    c = np.floor(p).astype(np.int64)
    y = np.zeros(x.shape[:-1] + (n_out,))
    for d in range(-a + 1, a + 1):
        m = c + d
        u = p - m
        w = np.sinc(u) * np.sinc(u / a)
        ok = (m >= 0) & (m < n) & (np.abs(u) < a)
        y[..., ok] += x[..., m[ok]] * w[ok]
    return y


def running_rms(x, sampling_rate, window=1.):
    """xcorr/stream/running_rms.py:104-111 with ObsPy's inclusive slide windows."""
    x = np.asarray(x, dtype=np.float64)
    step = int(round(window * sampling_rate))
    win = step + 1
    nwin = (x.shape[-1] - win) // step + 1
    return np.stack([np.sqrt(np.sum(x[..., i * step:i * step + win] ** 2, axis=-1) / win) for i in range(nwin)], axis=-1)


# -- f2: spectrogram (xcorr/signal/spectrogram.py:112-141) --------------------------
def spectrogram(x, sampling_rate, duration, padding_factor=2, scaling='density', **kwargs):
    """`_spectrogram` of signal/spectrogram.py:124-141: scipy.signal.spectrogram with
    nperseg = int(duration*fs), noverlap = nperseg-1, nfft = nperseg*padding_factor,
    mode 'psd', zero-padded to the input length (edge = rint(nperseg/2) on the left).
    Returns (freq, Sxx[..., nfreq, n]); float64."""
    x = np.asarray(x, dtype=np.float64)
    win_len = int(duration * sampling_rate)
    nfft = int(win_len * padding_factor)
    edge = int(np.rint(win_len / 2))
    _f, _t, Sxx = scipy.signal.spectrogram(x=x, fs=sampling_rate, nperseg=win_len, noverlap=win_len - 1, nfft=nfft,
                                           scaling=scaling, mode='psd', axis=-1, return_onesided=True, **kwargs)
    npad = [(0, 0)] * (Sxx.ndim - 1) + [(edge, x.shape[-1] - Sxx.shape[-1] - edge)]
    freq = np.linspace(0., sampling_rate / 2, int(nfft / 2 + 1))
    return freq, np.pad(Sxx, pad_width=npad, mode='constant', constant_values=0)


def spectrogram_mtc(x, sampling_rate, duration=None, padding_factor=None, overlap=None, **kwargs):
    """signal/spectrogram.py:253-276: argument defaults as the reference resolves them, then
    scipy.signal.spectrogram(nperseg, noverlap=int(nperseg*overlap), nfft, 'density', 'psd').
    Returns (f, t, Sxx[..., nfreq, nseg]); float64."""
    x = np.asarray(x, dtype=np.float64)
    delta = 1.0 / sampling_rate
    padding_factor = padding_factor if (padding_factor and padding_factor >= 1) else 1
    duration = duration if (duration and duration > delta) else delta
    win_len = int(duration * sampling_rate)
    if win_len < 16:
        raise ValueError('Change duration to have at least 16 sample points!')
    overlap = overlap if overlap and (0. < overlap < 1.) else .9
    win_shift = int(win_len * overlap)
    return scipy.signal.spectrogram(x=x, fs=sampling_rate, nperseg=win_len, noverlap=win_shift,
                                    nfft=int(win_len * padding_factor), scaling='density', mode='psd', axis=-1,
                                    **kwargs)
