#!/usr/bin/env python
"""
BASELINE.json configs[4] ("xcorr-timelapse"): 2-D cross-correlation of spectrogram tiles.

    python tools/bench_timelapse.py [--tiles 64] [--steps 3] [--from-ccf]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_timelapse.py --gpus N

SURVEY 8(d) config 5: T_l = 64 tiles float32 [256][4096] (default_rng(7), squares of N(0,1)),
all 2016 tile pairs, correlate2d peak value + position.  --from-ccf builds the tiles the way
scripts/timelapse.py does instead: CCF-like rows -> hop-1 spectrogram (2.5 s, padding 4) ->
a 256-bin band (timed separately).  Tile pairs shard over ranks with no collective (every
rank transforms all tiles).  Prints one JSON line on rank 0.
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--tiles', type=int, default=64)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--from-ccf', action='store_true')
    a = ap.parse_args()
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    from xcorr_amd.rendezvous import from_env
    rdzv = from_env()                          # file rendezvous of the launcher's ranks (no torch)
    import ctypes as C
    import xcorr_amd as X
    from xcorr_amd import _ffi, ops
    nd = C.c_int(0); _ffi.check(_ffi.lib().xcx_device_count(C.byref(nd)))
    ctx = X.Context(local % max(nd.value, 1))
    F, L, T = 256, 4096, a.tiles
    extra = {}
    if a.from_ccf:
        import scipy.signal as ss
        rng = np.random.default_rng(7)
        t = (np.arange(L) - L // 2) / 50.
        ccf = (0.05 * rng.standard_normal((T, L)) + np.exp(-0.5 * (t / 8.) ** 2) * np.cos(2 * np.pi * 4. * t)).astype(np.float32)
        w = ss.get_window(('tukey', .25), 125)
        d = ctx.array(ccf)
        S = ops.spectrogram(ctx, d, 125, 1000, w, 1. / (50. * (w * w).sum())); ctx.sync()
        t0 = time.perf_counter()
        S = ops.spectrogram(ctx, d, 125, 1000, w, 1. / (50. * (w * w).sum())); ctx.sync()
        extra['spectrogram_ms'] = (time.perf_counter() - t0) * 1e3
        tiles_h = np.ascontiguousarray(S.to_host()[:, 20:20 + F, :])      # 1-13.8 Hz band, 256 bins
        del S
        tiles = ctx.array(tiles_h)
    else:
        tiles = ctx.array(np.square(np.random.default_rng(7).standard_normal((T, F, L))).astype(np.float32))
    allp = np.array([(i, j) for i in range(T) for j in range(i + 1, T)], np.int32)
    pairs = np.ascontiguousarray(allp[rank::world])
    run = lambda: ops.correlate2d(ctx, tiles, pairs)
    for _ in range(a.warmup):
        run()
    rdzv.barrier()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(a.steps):
        peak, idx = run()
    ctx.sync()
    rdzv.barrier()
    dt = rdzv.max(time.perf_counter() - t0)
    if rank == 0:
        ms = dt / a.steps * 1e3
        # algorithmic bytes per pair: 2 spectra in + intermediate out/in = 67 MB
        print(json.dumps({
            'metric': 'tile-pairs/s', 'value': len(allp) * a.steps / dt, 'unit': 'tile-pairs/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'configs[4] timelapse: {T} tiles {F}x{L} -> {len(allp)} tile pairs, correlate2d '
                                   f'peak + position, padded transform 512x8192' + (', tiles from spectrogram' if a.from_ccf else ''),
                       'pairs_per_rank': int(len(pairs)), **extra},
            'check': {'peak0': float(peak[0]), 'idx0': [int(v) for v in idx[0]]}}))
    rdzv.close()


if __name__ == '__main__':
    main()
