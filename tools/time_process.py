"""Per-call overhead of the reference-style driver: one pair x one day per dataset (core/lazy.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd import core
from xcorr_amd.engine import default_context
from xcorr_amd.synth import synthetic_traces
ctx = default_context()
fs, wl = 50., 3600.
T = int(86400 * fs)
tr = synthetic_traces(2, T)
client = core.ArrayClient({'A': tr[0], 'B': tr[1]}, fs)
for rep in range(3):
    t0 = time.perf_counter()
    ds = core.init('A-B', 0., 86400., sampling_rate=fs, window_length=wl, window_overlap=.5)
    t1 = time.perf_counter()
    core.process(ds, client, ctx=ctx)
    ctx.sync()
    t2 = time.perf_counter()
    print(f'init {1e3*(t1-t0):.1f} ms, process {1e3*(t2-t1):.1f} ms, processed {int((ds.status==1).sum())} of {ds.status.size} steps', flush=True)
import cProfile, pstats
ds = core.init('A-B', 0., 86400., sampling_rate=fs, window_length=wl, window_overlap=.5)
pr = cProfile.Profile(); pr.enable(); core.process(ds, client, ctx=ctx); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
# ... and the post-processing of that container (core/postprocess.py: unbias -> demean -> taper -> timeshift -> filter -> taper)
ds.pair_offset[...] = 0.0049
for rep in range(3):
    t0 = time.perf_counter()
    pp = core.postprocess(ds, ctx=ctx)
    ctx.sync()
    print(f'postprocess {1e3*(time.perf_counter()-t0):.1f} ms for {int((ds.status==1).sum())} rows of {ds.cc.shape[-1]} lags', flush=True)
pr = cProfile.Profile(); pr.enable(); core.postprocess(ds, ctx=ctx); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
