#!/usr/bin/env python
"""
bench.py -- pair-windows/s of the MI355X cross-correlation engine.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one station-day batch through the whole hot path on one GPU:
preprocess of S day-long traces (demean -> cosine taper -> Butterworth
bandpass), then for each of the W windows the forward transforms (per-window
one-bit + spectral whitening fused) and the full-lag cross-correlation of all
P = S(S-1)/2 pairs; every CCF (L = 2n-1 lags) is written to HBM.  Inputs are
resident in HBM before the timed region; outputs stay in HBM.

Workload (BASELINE.json configs[2], the one the metric is quoted on):
S = 64 stations x 86400 s @ 50 Hz, n = 180000 (3600 s), 50 % overlap ->
W = 47 windows, P = 2016 pairs, 94752 pair-windows per step.

Multi-GPU: one process per GPU; every rank processes its own day (seeded by
rank) -- the path shards over independent station-days / pair-windows with no
data-path collective ("scaling": "weak").  --shard pairs splits the pair list
of ONE day over the ranks instead (strong scaling).  RCCL is only used for the
start/stop rendezvous (and for the time-stack reduce with --stack).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

FS = 50.0
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--stations', type=int, default=64)
    ap.add_argument('--seconds', type=float, default=86400.0)
    ap.add_argument('--window', type=float, default=3600.0)
    ap.add_argument('--overlap', type=float, default=0.5)
    ap.add_argument('--batch', type=int, default=0, help='pair-windows per launch batch')
    ap.add_argument('--shard', choices=['days', 'pairs', 'windows'], default='days')
    ap.add_argument('--clip-lag', type=float, default=0.0,
                    help='retain only lags within +-SECONDS (the reference\'s clip_lag); 0 = full lag')
    ap.add_argument('--no-preprocess', action='store_true',
                    help='configs[1]-style: raw traces, L2 normalisation only')
    ap.add_argument('--stack', action='store_true', help='time-stack mode (configs[3])')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--pipeline-front-end', action='store_true',
                    help='overlap the next step\'s preprocess + forward transforms with the pair phase '
                         '(xcx_cc1d_prepare; measured 20-25 %% SLOWER: the streaming front end evicts the '
                         'cache-resident T ring of the pair pipelines -- kept for A/B)')
    return ap.parse_args()


def synth_day(S, T, day):
    from xcorr_amd.synth import synthetic_traces
    return synthetic_traces(S, T, day=day)


# ---------------------------------------------------------------------------
# CPU baseline: the oracle's faithful restatement of util/cc.py (complex FFT of
# length 2n-1), one process per host core, bounded sample.
# ---------------------------------------------------------------------------
def _cpu_worker(args):
    seed, n, reps, fair = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import ref_numpy as R
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    f = R.cc_fair if fair else R.cc
    f(x[:1000], y[:1000])
    t0 = time.perf_counter()
    for _ in range(reps):
        f(x, y, normalize=True, dtype=np.float32)
    return time.perf_counter() - t0


def cpu_baseline(n, seconds):
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a one-GPU box owns a 16-core share of its host (more workers only
    # oversubscribe it); override with XCX_CPU_CORES
    cores = int(os.environ.get('XCX_CPU_CORES', min(cores, 16)))
    for v in ('OMP_NUM_THREADS', 'MKL_NUM_THREADS', 'OPENBLAS_NUM_THREADS'):
        os.environ[v] = '1'
    ctx = mp.get_context('spawn')
    out = {}
    with ctx.Pool(cores) as pool:
        # calibrate with one call per core, then size the sample
        t1 = max(pool.map(_cpu_worker, [(i, n, 1, False) for i in range(cores)]))
        reps = max(1, int(seconds / max(t1, 1e-3)))
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(100 + i, n, reps, False) for i in range(cores)])
        wall = time.perf_counter() - t0
        out['faithful'] = cores * reps / wall
        out['sample'] = (f'{cores * reps} pair-windows (n={n}, FFT length {2 * n - 1}, '
                         f'{reps} per core) of oracle.ref_numpy.cc in {wall:.1f} s')
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(200 + i, n, 20, True) for i in range(cores)])
        out['fair'] = cores * 20 / (time.perf_counter() - t0)
    out['cores'] = cores
    return out


def main():
    a = parse()
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus and world > 1:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)

    import xcorr_amd as X
    from xcorr_amd import _ffi
    from xcorr_amd import ops
    import ctypes as _C
    ndev = _C.c_int(0)
    _ffi.check(_ffi.lib().xcx_device_count(_C.byref(ndev)))
    # one rank per GPU; the modulo only matters when rehearsing N ranks on a smaller box
    ctx = X.Context(local % max(ndev.value, 1))

    S = a.stations
    T = int(round(a.seconds * FS))
    n = int(round(a.window * FS))
    hop = int(round(n * (1.0 - a.overlap)))
    W = (T - n) // hop + 1
    L = 2 * n - 1
    imin, imax = 0, L
    if a.clip_lag > 0:
        half = min(n - 1, int(round(a.clip_lag * FS)))
        imin, imax = n - 1 - half, n + half
        L = imax - imin
    pairs_all = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
    w_first, W_all = 0, W
    if a.shard == 'pairs' and world > 1:
        pairs = np.ascontiguousarray(pairs_all[rank::world])     # cyclic over the triangle
        day = 0
    elif a.shard == 'windows' and world > 1:
        # strong scaling over the windows of ONE day: contiguous blocks, every rank filters the
        # whole day (2.7 ms) but transforms and correlates only its own windows
        pairs = pairs_all
        day = 0
        w_first = (W_all * rank) // world
        W = (W_all * (rank + 1)) // world - w_first
    else:
        pairs = pairs_all
        day = rank
    P = len(pairs)
    ld = (L + 31) // 32 * 32          # 128-byte aligned output rows

    preprocess = not a.no_preprocess
    whiten = (FS, 0.1, 10.0, 0.05) if preprocess else None
    plan = X.CCPlan(ctx, n, imin, imax, normalize=True, onebit=preprocess, whiten=whiten,
                    stack=a.stack, batch=a.batch or None)

    # ---- synthetic inputs, resident in HBM before the clock starts ---------
    raw_h = synth_day(S, T, day)
    raw = ctx.array(raw_h)
    work = ctx.empty((S, T)) if preprocess else raw
    nrows = P if a.stack else P * W
    out = ctx.empty((nrows, ld))
    del raw_h
    if preprocess:
        import scipy.signal
        sos = scipy.signal.butter(4, [0.1, 10.0], btype='bandpass', output='sos', fs=FS)
        taper_n = min(int(0.05 * T), int(30.0 * FS), T // 2)

    lib = _ffi.lib()
    wview = work.columns_from(w_first * hop) if w_first else work     # this rank's first window

    # time-stack across ranks (configs[3]): the only exchange step of the path,
    # one RCCL all-reduce of the per-rank partial stacks over xGMI
    rccl = None
    if a.stack and dist is not None:
        try:
            uid = [X.Context.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(uid[0], world, rank)
            rccl = True
        except Exception as e:          # e.g. two ranks rehearsing on one GPU
            rccl = f'unavailable: {e}'

    def step():
        if preprocess:
            ops.copy(ctx, work, raw)
            ops.demean(ctx, work)
            ops.taper(ctx, work, taper_n, wtype='cosine')
            ops.sosfilt(ctx, work, sos)
        plan.pairs(wview, pairs, W, hop, out=out, ld_out=ld, stride_s=T)
        if rccl is True:
            ctx.stack_reduce(out, count=nrows * ld)

    # Optional software pipelining across steps (--pipeline-front-end): while the pair phase of step k
    # runs, step k+1 is preprocessed on a second context (own stream) and its windows are
    # transformed on the plan's side stream (xcx_cc1d_prepare).  Every step's full work stays
    # inside the timed region.  Off by default: it measures slower (see --help).
    pipelined = preprocess and not a.stack and a.pipeline_front_end and W >= 2
    ctx2 = X.Context(local % max(ndev.value, 1)) if pipelined else None

    def front():
        plan.prepare_sync()                 # the previous prepare no longer reads `work`
        ops.copy(ctx2, work, raw)
        ops.demean(ctx2, work)
        ops.taper(ctx2, work, taper_n, wtype='cosine')
        ops.sosfilt(ctx2, work, sos)
        ctx2.sync()
        plan.prepare(work, W, hop)

    def run(K):
        if not pipelined:
            for _ in range(K):
                step()
            return
        if K > 0:
            front()
        for k in range(K):
            plan.pairs(work, pairs, W, hop, out=out, ld_out=ld)      # consumes the prepared spectra
            if k + 1 < K:
                front()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    run(a.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    run(a.steps)
    ev_ms = ctx.timer_stop()
    ctx.sync()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        dist.barrier()

    units_rank = P * W * a.steps
    units_all = units_rank * world if a.shard == 'days' else len(pairs_all) * W_all * a.steps
    value = units_all / wall

    # ---- per-kernel timing of one extra step (HIP events on the stream the
    #      kernels are launched on, inside the library) ----------------------
    kern = None
    if not a.no_kernel_timing and rank == 0:
        kern = plan.profile_step(step, level=2)
        phase = plan.profile_step(step, level=1)
        if 'pair_phase' in phase:
            kern['pair_phase'] = phase['pair_phase']

    # ---- context numbers (rank 0, outside the timed region) -----------------
    extras = {}
    if not a.no_kernel_timing and rank == 0:
        # measured device-copy ceiling next to the nominal HBM peak
        src_b = raw
        dst_b = work if work is not raw else ctx.empty(raw.shape)
        ops.copy(ctx, dst_b, src_b); ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            ops.copy(ctx, dst_b, src_b)
        ctx.sync()
        extras['copy_GBps_measured'] = 5 * 2 * src_b.nbytes / (time.perf_counter() - t1) / 1e9
        # PCIe-inclusive rate of the host-pointer form of the same call (bounded sample):
        # traces in pageable host memory, CCFs copied back
        try:
            S_h, W_h = min(S, 8), min(W, 3)
            host_tr = raw.rows_to_host(0, S_h, (W_h - 1) * hop + n)
            prs = np.array([(i, j) for i in range(S_h) for j in range(i + 1, S_h)], np.int32)
            hplan = X.CCPlan(ctx, n, normalize=True)
            hplan.pairs(host_tr, prs[:2], W_h, hop)
            t1 = time.perf_counter()
            hplan.pairs(host_tr, prs, W_h, hop)
            dt_h = time.perf_counter() - t1
            extras['pcie_inclusive'] = {
                'value': len(prs) * W_h / dt_h, 'unit': 'pair-windows/s',
                'sample': f'{len(prs) * W_h} pair-windows, host traces in ({host_tr.nbytes / 1e6:.0f} MB) and '
                          f'full-lag CCFs out ({len(prs) * W_h * L * 4 / 1e6:.0f} MB) through pageable memory'}
            hplan.close()
        except Exception as e:          # context number only
            extras['pcie_inclusive'] = {'error': str(e)}

    res = {
        'metric': 'pair-windows/s' if not a.stack else 'pair-windows/s (stacked: windows summed per pair)',
        'value': value,
        'unit': 'pair-windows/s',
        'n_gpus': world,
        'steps': a.steps,
        'warmup': a.warmup,
        'ms_per_step': 1e3 * wall / a.steps,
        'higher_is_better': True,
        'scaling': 'weak' if a.shard == 'days' else 'strong',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': (f'{S} stations x {a.seconds:.0f} s @ {FS:.0f} Hz, {len(pairs_all)} pairs x {W_all} windows '
                         f'(n={n}, {a.overlap:.0%} overlap), ' + (f'lags +-{a.clip_lag:g} s L={L}, ' if a.clip_lag > 0 else f'full lag L={L}, ')
                         + ('demean+cosine-taper+bandpass(0.1-10 Hz, 4 corners) per station-day, '
                            'one-bit+whitening per window' if preprocess else 'no preprocess')
                         + (', time-stack' if a.stack else '')),
            'fft_length': plan.N, 'pair_batch': plan.batch,
            'shard': a.shard,
            'overlap_next_step_front_end': bool(pipelined), 'per_rank_pair_windows_per_step': P * W,
        },
        'cc_samples_per_s': value * L,
        'event_ms_per_step_rank0': ev_ms / a.steps,
    }
    if a.stack:
        res['rccl_stack_reduce'] = rccl
    if rank == 0:
        # algorithmic bytes per pair-window: every CCF written once
        # + every station window read once (+ the preprocess pass over the day)
        b_pw = 4.0 * L + 4.0 * n * S / len(pairs_all)
        if preprocess:
            b_pw += 2.0 * 4.0 * T * S / (len(pairs_all) * W)
        res['algorithmic_bytes_per_pair_window'] = b_pw
        res['pipeline_GBps'] = (P * W / (1e-3 * ev_ms / a.steps)) * b_pw / 1e9
        res['pipeline_frac_of_hbm_peak'] = res['pipeline_GBps'] / HBM_PEAK_GBS
        if kern:
            # The inverse pass of a pair-window is two launches: k_pair_cols (I1)
            # and k_pair_rows (I2).  I2 writes every algorithmic output byte and
            # is the kernel the roofline is quoted for; `pass` prices the whole
            # two-launch pass against the full algorithmic bytes per pair-window.
            dom = 'k_pair_dual' if 'k_pair_dual' in kern else 'k_pair_rows'
            names = [k for k in ('k_pair_cols', 'k_pair_rows', 'k_pair_dual') if k in kern]
            B = kern[dom]['units_per_launch']
            pmc = {}
            pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_latest.json')
            if os.path.exists(pmc_path):
                try:
                    pmc = json.load(open(pmc_path))
                except Exception:
                    pmc = {}
            whole_pass = dom == 'k_pair_dual'
            alg_dom = B * (b_pw if whole_pass else 4.0 * L)
            # k_pair_dual launches of the two pipelines overlap on the GPU: the time
            # one launch's worth of work takes is the pair-phase wall time (HIP events
            # on the main stream) divided by the number of launches
            if whole_pass and 'pair_phase' in kern:
                eff_us = kern['pair_phase']['total_us'] / kern[dom]['launches']
            else:
                eff_us = kern[dom]['avg_us']
            ach = alg_dom / (eff_us * 1e-6) / 1e9
            tr = pmc.get(dom, {}).get('hbm_bytes_per_launch')
            res['roofline'] = {
                'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'traffic': tr,
                'avg_launch_us': kern[dom]['avg_us'], 'launches': kern[dom]['launches'],
                'effective_launch_us': eff_us,
                'concurrent_pipelines': (kern[dom]['avg_us'] / eff_us) if eff_us else None,
                'algorithmic_bytes_per_launch': alg_dom,
                'pair_windows_per_launch': B,
                'kernels_us': {k: v['avg_us'] for k, v in kern.items()},
                # what the kernel actually moves between L2 and the fabric (Infinity Cache + HBM),
                # from the PMC passes, at the effective launch time: the quantity that saturates
                'traffic_GBps': (tr / (eff_us * 1e-6) / 1e9) if (tr and eff_us) else None,
                'copy_GBps_measured': extras.get('copy_GBps_measured'),
                'frac_of_measured_copy': (ach / extras['copy_GBps_measured']) if extras.get('copy_GBps_measured') else None,
            }
            if 'pcie_inclusive' in extras:
                res['pcie_inclusive'] = extras['pcie_inclusive']
            if not whole_pass:
                pass_us = sum(kern[k]['avg_us'] * kern[k]['launches'] for k in names) / kern[dom]['launches']
                res['roofline']['pass'] = {'kernels': names, 'us_per_launch_set': pass_us,
                                           'achieved': B * b_pw / (pass_us * 1e-6) / 1e9,
                                           'frac': B * b_pw / (pass_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(n, a.cpu_seconds)
            res['cpu_baseline'] = {
                'value': cb['faithful'], 'unit': 'pair-windows/s', 'cores': cb['cores'],
                'kind': 'port', 'sample': cb['sample'],
                'fair_rfft_value': cb['fair'],
            }
            res['speedup_vs_cpu_baseline'] = value / cb['faithful']
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
