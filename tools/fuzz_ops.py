"""Randomised parity sweep of the row ops / filters / FFT helpers against the oracle (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as R
from xcorr_amd import ops
from xcorr_amd.engine import default_context


def rel(got, ref):
    ref = np.asarray(ref)
    d = np.abs(ref).max()
    return float(np.abs(np.asarray(got) - ref).max() / (d if d > 0 else 1.0))


def run(budget=60.0, seed=0, ctx=None):
    rng = np.random.default_rng(seed)
    ctx = ctx or default_context()
    t_end = time.time() + budget
    worst, cases = {}, 0
    while time.time() < t_end:
        kind = rng.choice(['sos', 'rowops', 'fft', 'hilbert', 'ram', 'cc2d', 'lanczos', 'timeshift', 'timeshift_fft',
                           'specseg', 'resample', 'bigfft', 'zerophase', 'postprep', 'decimate', 'polyphase', 'longfft'])
        if kind == 'sos':
            n = int(rng.choice([rng.integers(40, 300), rng.integers(300, 20000), rng.integers(20000, 400000)]))
            b = int(rng.integers(1, 5)) if n > 20000 else int(rng.integers(1, 200))
            order = int(rng.integers(1, 9))
            btype = rng.choice(['lowpass', 'highpass', 'bandpass'])
            if btype == 'bandpass':
                order = min(order, 4)
                f = tuple(sorted(rng.uniform(0.05, 20., 2)));
                if f[1] - f[0] < 0.5: f = (f[0], f[0] + 0.5)
            else:
                f = float(rng.uniform(0.05, 20.))
            sos = R.butter_sos(order, f, btype, 50.)
            x = (rng.standard_normal((b, n)) + rng.uniform(-2, 2)).astype(np.float32)
            e = rel(ops.sosfilt(ctx, x, sos), R.sosfilt(sos, x))
            if n > 3 * (2 * sos.shape[0] + 1):
                e = max(e, rel(ops.sosfiltfilt(ctx, x, sos), R.sosfiltfilt(sos, x)))
            tol = 2e-5 if order > 4 else 1e-5
        elif kind == 'rowops':
            n = int(rng.integers(2, 300000)); b = int(rng.integers(1, 6))
            x = (rng.standard_normal((b, n)) + 3.0 + 1e-4 * np.arange(n)).astype(np.float32)
            e = max(rel(ops.demean(ctx, x), R.demean(x)), rel(ops.norm1d(ctx, x), R.norm1d(x)),
                    rel(ops.detrend_linear(ctx, x), R.detrend_linear(x)) / 2)
            tol = 1e-5
        elif kind == 'fft':
            n = int(rng.integers(2, 30000)); b = int(rng.integers(1, 4))
            x = rng.standard_normal((b, n)).astype(np.float32)
            e = rel(ops.rfft(ctx, x), R.rfft_kernel(x))
            z = (x + 1j * rng.standard_normal((b, n))).astype(np.complex64)
            e = max(e, rel(ops.fft(ctx, z), np.fft.fft(z.astype(np.complex128), axis=-1)))
            tol = 1e-5
        elif kind == 'hilbert':
            n = int(rng.integers(2, 20000)); x = rng.standard_normal((2, n)).astype(np.float32)
            e = rel(ops.hilbert(ctx, x), R.hilbert(x)); tol = 1e-5
        elif kind == 'ram':
            n = int(rng.integers(1, 50000)); h = int(rng.integers(0, 3073))
            x = rng.standard_normal((3, n)).astype(np.float32)
            e = rel(ops.ram_normalize(ctx, x, h), R.ram_normalize(x, h)); tol = 1e-5
        elif kind == 'cc2d':
            F = int(rng.integers(1, 65)); L = int(rng.integers(1, 600)); T = int(rng.integers(2, 5))
            t = np.abs(rng.standard_normal((T, F, L))).astype(np.float32)
            nrm = bool(rng.random() < 0.5)
            if rng.random() < 0.5:
                a, b2 = int(rng.integers(0, T)), int(rng.integers(0, T))
                pk, ix, full = ops.correlate2d(ctx, t, [(a, b2)], normalize=nrm, full=True)
                e = rel(full[0], R.correlate2d(t[a], t[b2], normalize=nrm)); tol = 1e-5
            else:       # a pair list over several tile groups (run order != caller's order): every map and peak in place
                T = int(rng.integers(5, 14)); F = int(rng.integers(1, 9)); L = int(rng.integers(1, 80))
                t = np.abs(rng.standard_normal((T, F, L))).astype(np.float32)
                prs = [(int(rng.integers(0, T)), int(rng.integers(0, T))) for _ in range(int(rng.integers(2, 60)))]
                pk, ix, full = ops.correlate2d(ctx, t, prs, normalize=nrm, full=True)
                e = 0.0
                for q, (a, b2) in enumerate(prs):
                    ref = R.correlate2d(t[a], t[b2], normalize=nrm)
                    e = max(e, rel(full[q], ref), abs(float(pk[q]) - float(ref.max())) / max(abs(float(ref.max())), 1e-30))
                tol = 1e-5
        elif kind == 'lanczos':
            n = int(rng.integers(50, 50000)); a_ = int(rng.integers(1, 25)); step = float(rng.uniform(0.2, 7.))
            off = float(rng.uniform(0, 3))
            if rng.random() < 0.3:       # positions a rounding away from a sample: the centre tap's sine is ~1e-12
                step = float(rng.choice([0.75, 0.750075, 2.5, 1.0 + 1e-13, 0.2 + 1e-14]))
                off = float(rng.choice([0.0, 1e-12, 3.0000000001, 2.0 - 1e-13]))
            nout = max(1, int((n - 1 - off) / step))
            x = rng.standard_normal((2, n)).astype(np.float32)
            e = rel(ops.lanczos(ctx, x, nout, off, step, a_), R.lanczos_interpolation(x, off, step, nout, a_)); tol = 1e-5
        elif kind == 'timeshift_fft':
            n = int(rng.integers(16, 60000)); x = (rng.standard_normal((3, n)) * np.hanning(n)).astype(np.float32)
            d = rng.uniform(-1, 1, 3)
            e = rel(ops.timeshift(ctx, x, d, 50., fast=False), R.timeshift_fft(x, d, 1 / 50.)); tol = 1e-5
        elif kind == 'specseg':
            import scipy.signal
            n = int(rng.integers(64, 20000)); win = int(rng.integers(16, min(n, 700) + 1))
            hop = int(rng.integers(1, win + 1)); nfft = int(win * rng.integers(1, 4) + rng.integers(0, 3))
            x = (rng.standard_normal((2, n)) + 0.5).astype(np.float32)
            w = scipy.signal.get_window(('tukey', .25), win)
            det = bool(rng.random() < 0.7)
            got = ops.spectrogram_seg(ctx, x, win, hop, nfft, w, 1.0 / (50. * (w * w).sum()), det)
            _, _, ref = scipy.signal.spectrogram(x.astype(np.float64), fs=50., nperseg=win, noverlap=win - hop, nfft=nfft,
                                                 detrend='constant' if det else False, scaling='density', mode='psd')
            e = rel(got, ref); tol = 1e-5
        elif kind == 'resample':
            from xcorr_amd import stream
            n = int(rng.integers(64, 120000)); new = float(rng.choice([10., 12.5, 20., 40., 62.5, 100.]))
            x = rng.standard_normal((2, n)).astype(np.float32)
            y, _ = stream.resample(x, 50., new, ctx=ctx)
            e = rel(y, R.resample_fourier(x, 50., new)); tol = 1e-5
        elif kind == 'bigfft':
            n = int(rng.integers(184000, 520000)); x = rng.standard_normal((1, n)).astype(np.float32)
            e = rel(ops.rfft(ctx, x), R.rfft_kernel(x))
            if rng.random() < 0.3:
                e = max(e, rel(ops.hilbert(ctx, x, envelope=True), np.abs(R.hilbert(x))))
            tol = 1e-5
        elif kind == 'zerophase':
            from xcorr_amd import stream
            n = int(rng.integers(200, 300000)); x = (rng.standard_normal((2, n)) + 1.0).astype(np.float32)
            sos = stream.butter_sos('bandpass', 50., int(rng.integers(1, 5)), freqmin=float(rng.uniform(0.05, 2.)), freqmax=float(rng.uniform(4., 20.)))
            e = rel(ops.sosfilt_zerophase(ctx, x, sos), R.sosfilt_zerophase(sos, x)); tol = 2e-5
        elif kind == 'decimate':                 # the filter's last pass stores every factor-th sample (sos.hip: Decim)
            from xcorr_amd import stream
            n = int(rng.choice([rng.integers(40, 5000), rng.integers(5000, 300000), rng.integers(300000, 3000000)]))
            b = int(rng.integers(1, 5)) if n > 20000 else int(rng.integers(1, 40))
            factor = int(rng.integers(2, 17)); fs = float(rng.choice([100., 250., 40.]))
            x = (rng.standard_normal((b, n)) + 0.3).astype(np.float32)
            if rng.random() < 0.5:
                y, _ = stream.decimate(ctx.array(x), fs, factor, ctx=ctx)
                got = y.to_host(); y.free()
            else:
                got, _ = stream.decimate(x, fs, factor, ctx=ctx)
            e = rel(got, R.decimate(x, fs, factor)); tol = 1e-5
        elif kind == 'polyphase':                # rational steps: per-call weight table + polyphase FIR (resample.hip)
            n = int(rng.integers(200, 200000)); a_ = int(rng.integers(1, 33))
            P_, Q_ = int(rng.integers(1, 40)), int(rng.integers(1, 65))
            step = P_ / Q_
            if step < 0.05: step = 1.0
            off = float(rng.choice([0.0, 0.5, float(rng.uniform(0, 3)), 2.0 - 1e-13]))
            nout = max(1, int((n - 1 - off) / step) + int(rng.integers(0, 2)))
            if off + (nout - 1) * step > n - 1: nout -= 1
            x = rng.standard_normal((int(rng.integers(1, 4)), n)).astype(np.float32)
            e = rel(ops.lanczos(ctx, x, nout, off, step, a_), R.lanczos_interpolation(x, off, step, nout, a_)); tol = 1e-5
        elif kind == 'longfft':                  # beyond every single transform: block-decomposed chirp convolution
            if rng.random() < 0.8:               # (expensive on the host: a fifth of the draws)
                continue
            n = int(rng.integers(790000, 2600000)); x = rng.standard_normal((1, n)).astype(np.float32)
            m = n - (n & 1)
            X = ops.rfft(ctx, x)
            e = rel(X, np.fft.rfft(x[:, :m].astype(np.float64), axis=-1))
            e = max(e, rel(ops.irfft(ctx, X), x[:, :m]) / 2)
            tol = 1e-5
        elif kind == 'postprep':
            npts = int(rng.integers(200, 100000)); full = 2 * npts - 1
            i0 = int(rng.integers(0, full - 64)); i1 = int(rng.integers(i0 + 64, full + 1))
            x = (1e-2 * rng.standard_normal((3, i1 - i0))).astype(np.float32)
            nmax = int(rng.integers(1, (i1 - i0) // 2 + 1))
            got = ops.post_prep(ctx, x.copy(), npts, i0, i1, unbiased=True, taper_nmax=nmax)
            r = R.demean(R.unbias(x.astype(np.float64), npts, i0, i1))
            wt = np.ones(i1 - i0); hw = np.hanning(2 * nmax + 1)[1:]           # periodic Hann of 2*nmax points
            import scipy.signal.windows as sw
            win = sw.hann(2 * nmax, False); wt[:nmax] = win[:nmax]; wt[-nmax:] = win[-nmax:]
            e = rel(got, r * wt); tol = 1e-5
        else:
            n = int(rng.integers(16, 60000)); b = int(rng.choice([3, 3, 40, 300])) if n < 8000 else 3
            x = (rng.standard_normal((b, n)) * np.hanning(n)).astype(np.float32)
            # distinct delays, or a handful of values shared by the rows (one Dirichlet kernel per value and launch group)
            nv = int(rng.integers(1, 5))
            d = rng.uniform(-1, 1, b) if rng.random() < 0.5 else rng.uniform(-1, 1, nv)[rng.integers(0, nv, b)]
            got, ref = ops.timeshift(ctx, x, d, 50.), R.timeshift(x, d, 50.)
            e = max(rel(got[r], ref[r]) for r in range(b)); tol = 1e-5
        cases += 1
        worst[kind] = max(worst.get(kind, 0.0), e)
        if not e < tol:
            raise AssertionError(f'{kind}: error {e} (seed {seed}, case {cases})')
    return cases, worst


if __name__ == '__main__':
    c, w = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f'fuzz_ops: {c} cases, worst relative errors', {k: f'{v:.1e}' for k, v in w.items()})
