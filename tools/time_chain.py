"""The reference's 'EDH' preprocess list (scripts/cc.py:98-113, without remove_response) on day-long 250 Hz rows,
device-resident, through xcorr_amd.stream.apply_operations -- and the 'BHZ' list's ops on 40 Hz rows (interpolate to
50 Hz with Lanczos a = 20, scripts/cc.py:47-68).  Per-op wall clock, one extra call each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd import stream
from xcorr_amd.engine import default_context
ctx = default_context()
rng = np.random.default_rng(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def run(name, x, fs, ops):
    def once():
        y, f2 = x, fs
        t_ops = []
        for op in ops:
            t0 = time.perf_counter()
            y, f2 = stream.apply_operations(y, f2, [op], ctx=ctx, t0=0.0, starttime=0.0, endtime=86400.0)
            ctx.sync()
            t_ops.append(time.perf_counter() - t0)
        return t_ops
    once()
    t = once()
    tot = sum(t)
    print(f'{name}: {rows} rows of {x.shape[-1]} samples: {tot*1e3:.2f} ms = {tot/rows*1e3:.3f} ms per station-day')
    for op, dt in zip(ops, t):
        print(f'    {str(op[0]):12s} {str(op[1])[:60]:60s} {dt*1e3:8.2f} ms')


T250 = 250 * 86400
x = ctx.array(rng.standard_normal((rows, T250), dtype=np.float32) + 2.0)
run('EDH (250 Hz -> 50 Hz)', x, 250., [
    ('detrend', {'type': 'demean'}), ('filter', {'type': 'highpass', 'freq': .05}), ('decimate', {'factor': 5}),
    ('trim', {}), ('detrend', {'type': 'demean'}), ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.})])
x.free()
T40 = 40 * 86400
x = ctx.array(rng.standard_normal((rows, T40), dtype=np.float32) + 2.0)
run('BHZ (40 Hz -> 50 Hz)', x, 40., [
    ('detrend', {'type': 'demean'}), ('filter', {'type': 'highpass', 'freq': .05}),
    ('interpolate', {'sampling_rate': 50, 'method': 'lanczos', 'a': 20}), ('filter', {'type': 'lowpass', 'freq': 20.}),
    ('trim', {}), ('detrend', {'type': 'demean'}), ('taper', {'type': 'cosine', 'max_percentage': 0.05, 'max_length': 30.})])
