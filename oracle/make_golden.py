"""
oracle/make_golden.py -- freeze golden vectors from the reference itself.

Run ONLY in the dev container (where the reference checkout is mounted
read-only at /root/reference); the reference never travels to the GPU box, the
small fixtures written to tests/golden/ do.

    /opt/conda/bin/python3.9 oracle/make_golden.py     # numpy 1.26: the
        reference-era stack, np.fft promotes float32 -> complex128

What can be imported: ``xcorr/util/cc.py`` only (numpy-only; loaded standalone
by file path).  Every other module of the reference hard-imports xarray / obspy
(not installed, no network) -- an ordinary ModuleNotFoundError, not a denial --
so the remaining rows of SURVEY 8a are pinned by scipy/numpy restatements only.
``weight()`` uses the removed alias ``np.int``; it is provided here in the
calling process (the reference file is never edited).

This file is test infrastructure (see oracle/ref_numpy.py header).
"""
import importlib.util
import os
import sys

import numpy as np

REF = '/root/reference/xcorr/util/cc.py'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests',
                   'golden')


def load_reference():
    if not hasattr(np, 'int'):
        np.int = int          # util/cc.py:137 needs the removed alias
    spec = importlib.util.spec_from_file_location('ref_cc', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    os.makedirs(OUT, exist_ok=True)
    g = {}
    g['numpy_version'] = np.array(np.__version__)

    # --- known-answer tests -----------------------------------
    x = np.array([1., 2., 3., 4.])
    y = np.array([0., 1., .5, 0.])
    g['kat_x'] = x
    g['kat_y'] = y
    g['kat_raw'] = ref.cc(x, y, normalize=False, pad=True, unbiased=False)
    g['kat_norm'] = ref.cc(x, y, normalize=True, pad=True, unbiased=False)
    g['kat_nopad'] = ref.cc(x, y, normalize=False, pad=False, unbiased=False)
    g['kat_lag'] = ref.lag(4, .02)
    g['kat_lag_nopad'] = ref.lag(4, .02, pad=False)

    # --- random small cases, inputs stored by value -----------------------
    sizes = [1, 2, 3, 5, 8, 16, 31, 100, 257, 1000, 4096]
    g['sizes'] = np.array(sizes)
    for n in sizes:
        rng = np.random.default_rng(100 + n)
        a = rng.standard_normal(n).astype(np.float32)
        b = (0.7 * np.roll(a, n // 7) +
             rng.standard_normal(n).astype(np.float32) + 0.25
             ).astype(np.float32)
        g[f'x_{n}'] = a
        g[f'y_{n}'] = b
        for nrm in (False, True):
            g[f'cc_{n}_norm{int(nrm)}'] = ref.cc(
                a, b, normalize=nrm, pad=True, unbiased=False,
                dtype=np.float64)
            # float32 in -> whatever dtype the reference returns natively
            g[f'cc32_{n}_norm{int(nrm)}'] = ref.cc(
                a, b, normalize=nrm, pad=True, unbiased=False)
        g[f'lag_{n}'] = ref.lag(n, 0.02)
        if n >= 3:
            g[f'weight_{n}'] = ref.weight(n, pad=True)
            g[f'weight_{n}_clip'] = ref.weight(n, pad=True, clip=3.0)

    # --- full-size digest: 1 h @ 50 Hz -----------------------------------
    n = 180000
    rng = np.random.default_rng(12345)
    a = rng.standard_normal(n).astype(np.float32)
    b = (np.roll(a, 250) + 0.5 * rng.standard_normal(n).astype(np.float32)
         ).astype(np.float32)
    r = ref.cc(a, b, True, True, False, np.float32)
    probes = np.array([0, 1, 17, 1000, 89999, 179748, 179749, 179750, 179999,
                       180000, 180249, 250000, 300001, 359000, 359997, 359998])
    g['big_n'] = np.array(n)
    g['big_shape'] = np.array(r.shape)
    g['big_argmax'] = np.array(int(np.argmax(r)))
    g['big_max'] = np.array(float(r.max()))
    g['big_sum'] = np.array(float(r.astype(np.float64).sum()))
    g['big_probe_idx'] = probes
    g['big_probe_val'] = r[probes].astype(np.float64)
    g['big_l2'] = np.array(float(np.sqrt((r.astype(np.float64) ** 2).sum())))
    shift, mx = ref.extract_shift_and_max(r, 0.02)
    g['big_shift_s'] = np.array(float(shift))
    g['big_shift_max'] = np.array(float(mx))

    path = os.path.join(OUT, 'cc_reference.npz')
    np.savez_compressed(path, **g)
    print('wrote', os.path.normpath(path), os.path.getsize(path), 'bytes;',
          'numpy', np.__version__, 'python', sys.version.split()[0])


if __name__ == '__main__':
    main()
