"""Per-kernel timing of one stacked day (config 4 shape: 64 stations, 47 windows, 2016 pairs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xcorr_amd.engine import default_context, CCPlan
from xcorr_amd.synth import synthetic_traces

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, hop, W = 180000, 90000, 47
T = 4320000
ctx = default_context()
tr = synthetic_traces(S, T).astype(np.float32)
d = ctx.array(tr)
pairs = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
plan = CCPlan(ctx, n, normalize=True, stack=True)
out = ctx.empty((len(pairs), (plan.L + 31) // 32 * 32))
step = lambda: (plan.pairs(d, pairs, W, hop, out=out, ld_out=out.shape[-1]), ctx.sync())
step()
t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
print(f'stacked day: {dt*1e3:.2f} ms  ({len(pairs)*W/dt:.0f} pair-windows/s equivalent)')
for k, v in plan.profile_step(step, level=2).items():
    print(f"  {k:14s} total {v['total_us']/1e3:8.3f} ms  launches {v['launches']:4d}  avg {v['avg_us']:9.1f} us")
if os.environ.get('EXPLICIT_CLOSE'):
    plan.close(); del out, d
    print('closed explicitly')
