#!/usr/bin/env python
"""
bench.py -- pair-windows/s of the MI355X cross-correlation engine.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...             (starts its own N rank processes, xcorr_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one station-day batch through the whole hot path
(xcorr_amd.workload.StationDay.step, the same object tests/test_gpu_config3.py
checks against the oracle): preprocess of S day-long traces (demean -> cosine
taper -> Butterworth bandpass), then for each of the W windows the forward
transforms (per-window one-bit + spectral whitening fused) and the full-lag
cross-correlation of all P = S(S-1)/2 pairs; every CCF (L = 2n-1 lags) is
written to HBM.  Inputs are resident in HBM before the timed region; outputs
stay in HBM.

Workload (BASELINE.json configs[2], the one the metric is quoted on):
S = 64 stations x 86400 s @ 50 Hz, n = 180000 (3600 s), 50 % overlap ->
W = 47 windows, P = 2016 pairs, 94752 pair-windows per step.

Multi-GPU (one process per GPU, no data-path collective -- SURVEY 8e):
  N > 1 default, --shard windows : STRONG scaling.  The ranks split the windows of ONE
      station-day into contiguous blocks; every rank filters the whole day (3 % of a
      step) but transforms and correlates only its own windows.
  --shard pairs : strong scaling over the pair list (every rank transforms all windows).
  --shard days  : weak scaling, every rank its own station-day.
  --stack [--days D] : BASELINE configs[3].  D station-days are dealt to the ranks, every
      rank accumulates its days' cross-spectra in the frequency domain
      (xcx_cc1d_stack_add), inverse-transforms once and the per-rank partial stacks are
      summed with ONE RCCL reduce-scatter over xGMI (xcx_stack_reduce).
The launcher's ranks meet through a directory of files (xcorr_amd/rendezvous.py): no
torch, no MPI.  Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    ap.add_argument('--shard', choices=['auto', 'days', 'pairs', 'windows'], default='auto',
                    help='auto = windows (strong scaling) for N > 1')
    ap.add_argument('--clip-lag', type=float, default=0.0,
                    help='retain only lags within +-SECONDS (the reference\'s clip_lag); 0 = full lag')
    ap.add_argument('--no-preprocess', action='store_true',
                    help='configs[1]-style: raw traces, L2 normalisation only')
    ap.add_argument('--stack', action='store_true', help='time-stack mode (configs[3])')
    ap.add_argument('--days', type=int, default=0,
                    help='--stack: station-days per step, dealt to the ranks (default: one per rank)')
    ap.add_argument('--stack-reduce', choices=['freq', 'time'], default='freq',
                    help='--stack with N > 1: sum the ranks\' partial stacks in the frequency domain (one RCCL '
                         'reduce-scatter of the in-band cross-spectra, each rank inverse-transforms its share of the '
                         'pairs: xcx_cc1d_stack_finish_reduce) or in the time domain (every rank inverse-transforms all '
                         'pairs, then xcx_stack_reduce of the P x L stacks)')
    ap.add_argument('--oversubscribe', action='store_true',
                    help='rehearsal: allow more ranks than visible GPUs (ranks share devices; RCCL will refuse)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--pipeline-front-end', action='store_true',
                    help='overlap the next step\'s preprocess + forward transforms with the pair phase '
                         '(xcx_cc1d_prepare; measured 20-25 %% SLOWER: the streaming front end evicts the '
                         'cache-resident T ring of the pair pipelines -- kept for A/B; --shard days only)')
    return ap.parse_args()


# ---------------------------------------------------------------------------
# CPU baseline: the oracle's faithful restatement of util/cc.py (complex FFT of
# length 2n-1), one process per host core, bounded sample; plus the oracle's
# preprocess of one station-day row.
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


def _cpu_pre_worker(args):
    seed, T = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import ref_numpy as R
    from xcorr_amd import workload
    x = np.random.default_rng(seed).standard_normal(T).astype(np.float32)
    sos = workload.band_sos()
    w = R.cosine_taper_obspy(T, FS, *workload.TAPER)
    t0 = time.perf_counter()
    y = R.demean(x.astype(np.float64)[None])
    y = y * w
    R.sosfilt(sos, y)
    return time.perf_counter() - t0


def cpu_baseline(n, T, seconds, preprocess):
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for qf, pf in (('/sys/fs/cgroup/cpu.max', None),             # a container's CPU share, if smaller
                   ('/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us')):
        try:
            if pf is None:
                q, per = open(qf).read().split()
            else:
                q, per = open(qf).read().strip(), open(pf).read().strip()
            if q not in ('max', '-1'):
                cores = max(1, min(cores, -(-int(q) // int(per))))
            break
        except (OSError, ValueError):
            continue
    cores = int(os.environ.get('XCX_CPU_CORES', cores))     # every core this process may use
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
        if preprocess:
            t0 = time.perf_counter()
            pool.map(_cpu_pre_worker, [(300 + i, T) for i in range(cores)])
            out['preprocess_station_days_per_s'] = cores / (time.perf_counter() - t0)
    out['cores'] = cores
    return out


def lib_sha256():
    from xcorr_amd import _ffi
    path = getattr(_ffi, 'LIB_PATH', os.path.join(ROOT, 'xcorr_amd', 'lib', 'libxcx.so'))
    h = hashlib.sha256()
    with open(path, 'rb') as fh:
        for blk in iter(lambda: fh.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()


def src_sha256():
    """sha256 over the kernel sources (csrc/*.hip, *.h, Makefile, include/xcx.h): identifies the build
    independently of where and when it was compiled (the .so embeds build paths)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'xcorr_amd', 'csrc')
    files = sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.h')) or f == 'Makefile')
    for f in [os.path.join(csrc, f) for f in files] + [os.path.join(ROOT, 'include', 'xcx.h')]:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _launch_module():
    """xcorr_amd/launch.py loaded by path: the launching parent imports nothing of the package."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('_xcx_launch', os.path.join(ROOT, 'xcorr_amd', 'launch.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    a = parse()
    # `python bench.py --gpus N` without a launcher: this process becomes the launcher -- before any GPU call and
    # before xcorr_amd is imported -- and starts N fresh rank processes (xcorr_amd/launch.py); with a launcher
    # (torch.distributed.run, mpirun, ...) WORLD_SIZE must equal --gpus.  Never a silent one-rank run.
    launch = _launch_module()
    if launch.needs_launch(a.gpus) == 'spawn':
        if not a.oversubscribe:
            try:
                ndev = launch.visible_devices()
            except Exception as e:
                raise SystemExit(f'--gpus {a.gpus}: cannot count the visible GPUs ({e})')
            if a.gpus > ndev:
                raise SystemExit(f'--gpus {a.gpus} but {ndev} GPU(s) visible; --oversubscribe lets ranks share '
                                 'devices (rehearsal only)')
        sys.exit(launch.spawn_ranks(a.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    from xcorr_amd import rendezvous
    rdzv = rendezvous.from_env()

    import xcorr_amd as X
    from xcorr_amd import _ffi, ops, workload
    from xcorr_amd.synth import synthetic_traces, fast_day
    import ctypes as _C
    ndev = _C.c_int(0)
    _ffi.check(_ffi.lib().xcx_device_count(_C.byref(ndev)))
    # one rank per GPU; the modulo only matters when rehearsing N ranks on a smaller box (--oversubscribe)
    if local >= max(ndev.value, 1) and not a.oversubscribe:
        raise SystemExit(f'rank {rank}: LOCAL_RANK={local} but {ndev.value} GPU(s) visible (--oversubscribe to share devices)')
    ctx = X.Context(local % max(ndev.value, 1))

    S = a.stations
    T = int(round(a.seconds * FS))
    n = int(round(a.window * FS))
    hop = int(round(n * (1.0 - a.overlap)))
    W_all = (T - n) // hop + 1
    L = 2 * n - 1
    imin, imax = 0, L
    if a.clip_lag > 0:
        half = min(n - 1, int(round(a.clip_lag * FS)))
        imin, imax = n - 1 - half, n + half
        L = imax - imin
    pairs_all = np.array([(i, j) for i in range(S) for j in range(i + 1, S)], np.int32)
    shard = a.shard
    if shard == 'auto':
        shard = 'days' if (world == 1 or a.stack) else 'windows'
    if a.stack and shard != 'days':
        raise SystemExit('--stack shards station-days over the ranks (--shard days)')
    w_first, W = 0, W_all
    pairs, day0 = pairs_all, 0
    if world > 1 and shard == 'pairs':
        pairs = np.ascontiguousarray(pairs_all[rank::world])     # cyclic over the triangle
    elif world > 1 and shard == 'windows':
        w_first = (W_all * rank) // world
        W = (W_all * (rank + 1)) // world - w_first
    elif world > 1:
        day0 = rank
    P = len(pairs)
    preprocess = not a.no_preprocess

    # ---- station-days of this rank, resident in HBM before the clock starts ----
    ndays = 1
    my_days = [day0]
    if a.stack:
        ndays = a.days or world
        from xcorr_amd.shard import shard_days
        my_days = shard_days(ndays, rank, world)
    sd = workload.StationDay(ctx, S, T, n, hop, pairs, imin, imax, preprocess=preprocess, stack=a.stack,
                             batch=a.batch or None, w_first=w_first, W=W)
    plan = sd.plan
    ld = sd.ld
    raws = []
    for d in my_days:
        host = synthetic_traces(S, T, day=d) if len(my_days) == 1 and ndays <= world else fast_day(S, T, d)
        if not raws:
            sd.load(host)
            raws.append(sd.raw)
        else:
            raws.append(ctx.array(host))
        del host
    nrows = sd.nrows
    out = sd.out

    # time-stack across ranks (configs[3]): the only exchange step of the path
    rccl = None
    rccl_ranks = None
    if a.stack and world > 1:
        uid = rdzv.broadcast(X.Context.comm_unique_id() if rank == 0 else b'')
        ctx.comm_init(uid, world, rank)          # a failure here is fatal: no silent partial stacks
        rccl = 'reduce_scatter' if (nrows * ld) % world == 0 else 'all_reduce'
        if a.stack_reduce == 'freq' and ndays >= world:      # every rank has a day to accumulate
            rccl = 'reduce_scatter_freq'
        rccl_ranks, rccl_rank = ctx.comm_info()  # what RCCL itself says, not what we passed in
        if rccl_ranks != world or rccl_rank != rank:
            raise SystemExit(f'RCCL communicator spans {rccl_ranks} ranks (rank {rccl_rank}); '
                             f'launcher says WORLD_SIZE={world}, RANK={rank}')

    def step():
        if not a.stack:
            sd.step()
            return
        if len(my_days) == 1 and world == 1:
            sd.step()                            # single stacked day: xcx_cc1d_pairs(XCX_STACK)
            return
        try:
            for r in raws:
                sd.raw = r
                sd.front()
                plan.stack_add(sd.view, pairs, W, hop, stride_s=T)
            if rccl == 'reduce_scatter_freq':
                owned, _ = plan.stack_finish_reduce(pairs, out, ld_out=ld)     # the sum across ranks, this rank's pairs
                return
            if raws:
                plan.stack_finish(pairs, out=out, ld_out=ld)
            else:
                out.zero()                       # more ranks than days: this rank contributes zeros
        except BaseException:
            plan.stack_reset()                   # do not leave the plan refusing every later call
            raise
        if rccl:
            ctx.stack_reduce(out, count=nrows * ld, reduce_scatter=(rccl == 'reduce_scatter'))

    # Optional software pipelining across steps (--pipeline-front-end), off by default.
    pipelined = preprocess and not a.stack and a.pipeline_front_end and W >= 2 and shard == 'days'
    ctx2 = X.Context(local % max(ndev.value, 1)) if pipelined else None

    def front():
        plan.prepare_sync()                 # the previous prepare no longer reads `work`
        sd.front(ctx2)
        ctx2.sync()
        plan.prepare(sd.work, W, hop)

    def run(K):
        if not pipelined:
            for _ in range(K):
                step()
            return
        if K > 0:
            front()
        for k in range(K):
            sd.correlate()                       # consumes the prepared spectra
            if k + 1 < K:
                front()

    def barrier():
        ctx.sync()
        rdzv.barrier()

    run(a.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.timer_start()
    run(a.steps)
    ev_ms = ctx.timer_stop()
    ctx.sync()
    wall = rdzv.max(time.perf_counter() - t0)
    rdzv.barrier()

    # outside the timed region, before anything else touches `out` (the write-ceiling measurement below overwrites
    # it): the stack the last step left behind is a stack, not zeros (row 0 = pair (0, 1) lies in rank 0's slice of a
    # reduce-scatter as well) -- its maximum sits where the synthetic move-out puts it, 25 samples per station index
    stack_check = None
    if a.stack and rank == 0:
        row0 = out.rows_to_host(0, 1, L)[0]
        i0, j0 = (int(v) for v in pairs_all[0])
        want = n - 1 - imin + 25 * (i0 - j0)
        stack_check = {'row0_absmax': float(np.abs(row0).max()), 'row0_argmax': int(np.argmax(row0)),
                       'row0_argmax_expected': want}
        if not (stack_check['row0_absmax'] > 0 and (not 0 <= want < L or int(np.argmax(row0)) == want)):
            raise SystemExit(f'stacked output failed its sanity check: {stack_check}')

    units_rank = P * W * a.steps * (len(my_days) if a.stack else 1)
    if a.stack:
        units_all = len(pairs_all) * W_all * ndays * a.steps
    elif shard == 'days':
        units_all = units_rank * world
    else:
        units_all = len(pairs_all) * W_all * a.steps
    value = units_all / wall

    # ---- per-kernel timing of one extra step (HIP events on the stream the
    #      kernels are launched on, inside the library) ----------------------
    kern = None
    if not a.no_kernel_timing and rank == 0 and world == 1:      # roofline leg: N = 1 only (a lone rank
        kern = plan.profile_step(step, level=2)                    # cannot re-run a step with a collective)
        phase = plan.profile_step(step, level=1)
        if 'pair_phase' in phase:
            kern['pair_phase'] = phase['pair_phase']

    # ---- context numbers (rank 0, outside the timed region) -----------------
    extras = {}
    if not a.no_kernel_timing and rank == 0 and world == 1:
        # measured device-copy ceiling next to the nominal HBM peak
        src_b = sd.raw
        dst_b = sd.work if sd.work is not sd.raw else ctx.empty(sd.raw.shape)
        ops.copy(ctx, dst_b, src_b); ctx.sync()
        t1 = time.perf_counter()
        for _ in range(5):
            ops.copy(ctx, dst_b, src_b)
        ctx.sync()
        extras['copy_GBps_measured'] = 5 * 2 * src_b.nbytes / (time.perf_counter() - t1) / 1e9
        # ... and the write ceiling: the pass is output-dominated (1.44 MB of CCF per pair-window), so what
        # the chip sustains for a pure stream of 16-byte non-temporal stores bounds it from above
        wbytes = min(out.nbytes, 8 << 30)
        wms = _C.c_float(0)
        _ffi.check(_ffi.lib().xcx_stream_write_rate(ctx.handle, _C.c_void_p(out.ptr), wbytes, 3, _C.byref(wms)))
        extras['nt_write_stream_GBps_measured'] = 3 * wbytes / (wms.value * 1e-3) / 1e9
        # PCIe-inclusive rate of the host-pointer form of the same call (bounded sample)
        try:
            S_h, W_h = min(S, 8), min(W, 3)
            host_tr = sd.raw.rows_to_host(0, S_h, (W_h - 1) * hop + n)
            prs = np.array([(i, j) for i in range(S_h) for j in range(i + 1, S_h)], np.int32)
            hplan = X.CCPlan(ctx, n, normalize=True)
            r = hplan.pairs(host_tr, prs, W_h, hop)          # steady state of a batch loop: the
            del r                                            # page-locked result block is recycled
            t1 = time.perf_counter()
            for _ in range(3):
                r = hplan.pairs(host_tr, prs, W_h, hop)
                del r
            dt_h = (time.perf_counter() - t1) / 3
            extras['pcie_inclusive'] = {
                'value': len(prs) * W_h / dt_h, 'unit': 'pair-windows/s',
                'sample': f'{len(prs) * W_h} pair-windows, host traces in ({host_tr.nbytes / 1e6:.0f} MB) and '
                          f'full-lag CCFs out ({len(prs) * W_h * L * 4 / 1e6:.0f} MB) into page-locked result arrays, per call'}
            # the link's own ceiling: one device -> page-locked host copy of the same size
            nb = len(prs) * W_h * L * 4 // 4 * 4
            hbuf = ctx.pinned_empty((nb // 4,), np.float32)
            dsrc = ctx.empty((nb // 4,))
            dsrc.copy_to(hbuf); ctx.sync()
            t1 = time.perf_counter()
            for _ in range(3):
                dsrc.copy_to(hbuf)
            ctx.sync()
            d2h = 3 * nb / (time.perf_counter() - t1) / 1e9
            extras['pcie_inclusive']['d2h_GBps_measured'] = d2h
            extras['pcie_inclusive']['frac_of_measured_d2h'] = (len(prs) * W_h * L * 4 / dt_h / 1e9) / d2h
            dsrc.free()
            hplan.close()
        except Exception as e:          # context number only
            extras['pcie_inclusive'] = {'error': str(e)}

    workload_str = (f'{S} stations x {a.seconds:.0f} s @ {FS:.0f} Hz, {len(pairs_all)} pairs x {W_all} windows '
                    f'(n={n}, {a.overlap:.0%} overlap), '
                    + (f'lags +-{a.clip_lag:g} s L={L}, ' if a.clip_lag > 0 else f'full lag L={L}, ')
                    + ('demean+cosine-taper+bandpass(0.1-10 Hz, 4 corners) per station-day, '
                       'one-bit+whitening per window' if preprocess else 'no preprocess')
                    + (f', time-stack over {ndays} station-day(s)' if a.stack else ''))
    res = {
        'metric': 'pair-windows/s' if not a.stack else 'pair-windows/s (stacked: windows summed per pair)',
        'value': value,
        'unit': 'pair-windows/s',
        'n_gpus': world,
        'steps': a.steps,
        'warmup': a.warmup,
        'ms_per_step': 1e3 * wall / a.steps,
        'higher_is_better': True,
        'scaling': 'weak' if (shard == 'days' and not (a.stack and a.days)) else 'strong',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': workload_str,
            'fft_length': plan.N, 'pair_batch': plan.batch,
            't_storage': plan.t_storage,      # intermediate of the two-pass inverse: 'q24' = 2 x 24-bit fixed point
            'zero_skip_blocks': ('%d..%d of %d' % (plan.zero_skip[0], plan.zero_skip[1] - 1 - plan.zero_skip[0], plan.zero_skip[1])
                                 if plan.zero_skip[0] else 'none'),   # 64-point spectrum blocks the column pass knows to be zero
            'shard': shard,
            'overlap_next_step_front_end': bool(pipelined), 'per_rank_pair_windows_per_step': units_rank // a.steps,
        },
        'cc_samples_per_s': value * L,
        'event_ms_per_step_rank0': ev_ms / a.steps,
    }
    if a.stack:
        res['rccl_stack_reduce'] = rccl
        res['rccl_ranks'] = rccl_ranks if rccl else None     # ncclCommCount of the communicator used
        res['config']['station_days'] = ndays
        if stack_check is not None:
            res['stack_check'] = stack_check
    if rank == 0:
        sha = lib_sha256()
        res['lib_sha256'] = sha
        try:
            ssha = src_sha256()
        except OSError:
            ssha = None
        res['src_sha256'] = ssha
        # algorithmic bytes per pair-window: every CCF written once
        # + every station window read once (+ the preprocess pass over the day)
        b_pw = 4.0 * L + 4.0 * n * S / len(pairs_all)
        if preprocess:
            b_pw += 2.0 * 4.0 * T * S / (len(pairs_all) * W_all)
        res['algorithmic_bytes_per_pair_window'] = b_pw
        res['pipeline_GBps'] = (units_rank / a.steps / (1e-3 * ev_ms / a.steps)) * b_pw / 1e9
        res['pipeline_frac_of_hbm_peak'] = res['pipeline_GBps'] / HBM_PEAK_GBS
        if kern:
            dom = 'k_pair_dual' if 'k_pair_dual' in kern else 'k_pair_rows'
            names = [k for k in ('k_pair_cols', 'k_pair_rows', 'k_pair_dual') if k in kern]
            B = kern[dom]['units_per_launch']
            # PMC traffic of the dominant kernel: only from a profile of THIS library build on
            # THIS workload (tools/prof.sh stamps pmc_latest.json); anything else is null
            tr, tr_note = None, 'no profiles/pmc_latest.json'
            pmc_path = os.path.join(ROOT, 'profiles', 'pmc_latest.json')
            if os.path.exists(pmc_path):
                try:
                    pmc = json.load(open(pmc_path))
                    st = pmc.get('_stamp', {})
                    same_build = st.get('lib_sha256') == sha or (ssha is not None and st.get('src_sha256') == ssha)
                    if not same_build:
                        tr_note = 'pmc_latest.json was measured on another libxcx.so build'
                    elif st.get('workload') != workload_str or st.get('pair_batch') != plan.batch:
                        tr_note = 'pmc_latest.json was measured on another workload'
                    else:
                        tr = pmc.get(dom, {}).get('hbm_bytes_per_launch')
                        tr_note = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of {st.get('tag')}"
                except Exception as e:
                    tr_note = f'pmc_latest.json unreadable: {e}'
            whole_pass = dom == 'k_pair_dual'
            alg_dom = B * (b_pw if whole_pass else 4.0 * L)
            # k_pair_dual launches of the pipelines overlap on the GPU: the time one launch's
            # worth of work takes is the pair-phase wall time (HIP events on the main stream)
            # divided by the number of launches (profiles/*_overlap.json shows the union)
            if whole_pass and 'pair_phase' in kern:
                eff_us = kern['pair_phase']['total_us'] / kern[dom]['launches']
            else:
                eff_us = kern[dom]['avg_us']
            ach = alg_dom / (eff_us * 1e-6) / 1e9
            copy = extras.get('copy_GBps_measured')
            res['roofline'] = {
                'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS,
                'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                'traffic': tr, 'traffic_source': tr_note,
                # the same algorithmic bytes priced three ways
                'frac_pair_phase': ach / HBM_PEAK_GBS,
                'frac_end_to_end': res['pipeline_frac_of_hbm_peak'],
                'frac_of_measured_copy': (ach / copy) if copy else None,
                'avg_launch_us': kern[dom]['avg_us'], 'launches': kern[dom]['launches'],
                'effective_launch_us': eff_us,
                'concurrent_pipelines': (kern[dom]['avg_us'] / eff_us) if eff_us else None,
                'algorithmic_bytes_per_launch': alg_dom,
                'pair_windows_per_launch': B,
                'kernels_us': {k: v['avg_us'] for k, v in kern.items()},
                'traffic_GBps': (tr / (eff_us * 1e-6) / 1e9) if (tr and eff_us) else None,
                'copy_GBps_measured': copy,
                # the CCF stream alone (algorithmic output bytes / measured nt write rate) bounds the pass at:
                'nt_write_stream_GBps_measured': extras.get('nt_write_stream_GBps_measured'),
                'pair_windows_per_s_at_write_ceiling': (extras['nt_write_stream_GBps_measured'] * 1e9 / (4.0 * L))
                if extras.get('nt_write_stream_GBps_measured') else None,
            }
            if 'pcie_inclusive' in extras:
                res['pcie_inclusive'] = extras['pcie_inclusive']
            if not whole_pass:
                pass_us = sum(kern[k]['avg_us'] * kern[k]['launches'] for k in names) / kern[dom]['launches']
                res['roofline']['pass'] = {'kernels': names, 'us_per_launch_set': pass_us,
                                           'achieved': B * b_pw / (pass_us * 1e-6) / 1e9,
                                           'frac': B * b_pw / (pass_us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        if not a.no_cpu_baseline and world == 1:
            cb = cpu_baseline(n, T, a.cpu_seconds, preprocess)
            res['cpu_baseline'] = {
                'value': cb['faithful'], 'unit': 'pair-windows/s', 'cores': cb['cores'],
                'kind': 'port', 'sample': cb['sample'],
                'fair_rfft_value': cb['fair'],
            }
            if 'preprocess_station_days_per_s' in cb:
                # the preprocess stage on the same host cores,
                # and the whole step (preprocess + faithful cc) priced per pair-window
                pre = cb['preprocess_station_days_per_s']
                res['cpu_baseline']['preprocess_station_days_per_s'] = pre
                t_step = S / pre + len(pairs_all) * W_all / cb['faithful']
                res['cpu_baseline']['whole_step_value'] = len(pairs_all) * W_all / t_step
            res['speedup_vs_cpu_baseline'] = value / cb['faithful']
        print(json.dumps(res), flush=True)
    rdzv.close()


if __name__ == '__main__':
    main()
