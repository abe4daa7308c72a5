"""
Pair x window driver (xcorr/core/process.py:23-185).

``process`` keeps the reference's control flow -- skip processed steps,
``retry_missing``, status -1 on missing data, pair/time offsets, the lag slice
``[index_min:index_max]``, optional bias correction at the end -- but instead
of one ``util.cc.cc`` call per (pair, time) it collects every available window
first and hands the whole batch to the GPU engine in one call
(``CCPlan.pairs`` over the DISTINCT station windows): the hot loop of
core/process.py:82-165 becomes one batched launch sequence in which every
window is transformed once, however many pairs share it.
"""
import hashlib

import numpy as np

from ..engine import CCPlan, default_context

__all__ = ['process', 'process_batched']

try:                                    # 128-bit content key of a window: xxh3 when present (10x blake2b)
    import xxhash as _xx

    def _digest(b):
        return _xx.xxh3_128_digest(b)
except ModuleNotFoundError:             # pragma: no cover
    def _digest(b):
        return hashlib.blake2b(b, digest_size=16).digest()


def process(dataset, client, inventory=None, retry_missing=False,
            test_run=False, hash_waveforms=False, metadata_hash=None, verb=0,
            ctx=None, **kwargs):
    """Process the container in place and return it.

    dataset : :class:`xcorr_amd.core.CCDataset` (see ``init``) or, where xarray is installed, the
        reference's ``xarray.Dataset`` (converted with ``CCDataset.from_xarray`` and updated in place)
    client  : object with ``get_pair_processed_waveforms(pair, time, duration,
        ...)`` returning two (data, starttime_seconds) items or None
        (stream/client.py:1116-1149; :class:`ArrayClient` here).
    hash_waveforms : when the container has a ``hash`` variable (``init(...,
        hash_waveforms=True)``), store the sha256 of the two preprocessed traces per
        (pair, time) (core/process.py:78-79,146-149; util/hasher.py:142-215 -- headers
        rebuilt from the pair string, the start times and the sampling rate).
    """
    from ..util import hasher
    from .init import CCDataset, is_xarray_dataset
    if is_xarray_dataset(dataset):
        # the reference's calling convention (core/process.py:23-30): an xarray.Dataset from xcorr.init /
        # xcorr.read, processed IN PLACE and returned -- converted in, run, written back
        work = CCDataset.from_xarray(dataset)
        process(work, client, inventory=inventory, retry_missing=retry_missing, test_run=test_run,
                hash_waveforms=hash_waveforms, metadata_hash=metadata_hash, verb=verb, ctx=ctx, **kwargs)
        return work.update_xarray(dataset)
    hash_waveforms = hash_waveforms and getattr(dataset, 'hash', None) is not None
    la = dataset.lag_attrs
    npts, imin, imax = la['npts'], la['index_min'], la['index_max']
    wl = dataset.time_attrs['window_length']
    todo, xs, ys = [], [], []
    for ip, p in enumerate(dataset.pair):
        for it, t in enumerate(dataset.time):
            st = dataset.status[ip, it]
            if st != 0 and not (retry_missing and st == -1):
                continue                      # core/process.py:93-104
            w = client.get_pair_processed_waveforms(
                pair=p, time=t, duration=wl,
                sampling_rate=la['sampling_rate'], **kwargs)
            if w is None or len(w) != 2:
                dataset.status[ip, it] = -1   # core/process.py:122-132
                if test_run:
                    break
                continue
            (xa, ta), (xb, tb) = w
            dataset.pair_offset[ip, it] = ta - tb            # :135-138
            dataset.time_offset[ip, it] = ta + wl / 2 - t    # :139-143
            if hash_waveforms:                               # :146-149, the whole traces as received
                ids = str(p).split('-')
                fs = la['sampling_rate']
                dataset.hash[ip, it] = hasher.hash_stream(
                    (hasher.trace_stats(ids[k] if k < len(ids) else '', tk, fs, len(xk)), np.asarray(xk))
                    for k, (xk, tk) in enumerate(((xa, ta), (xb, tb))))
            xs.append(np.asarray(xa[:npts], dtype=np.float32))
            ys.append(np.asarray(xb[:npts], dtype=np.float32))
            todo.append((ip, it))
            if test_run:
                break
        if test_run and todo:
            break
    if todo:
        ctx = ctx or default_context()
        plan = ctx.plan(npts, index_min=imin, index_max=imax,
                        normalize=bool(dataset.cc_attrs['normalize'] == 1), unbiased=False)
        # every distinct station window is transformed ONCE (the reference transforms both
        # inputs of every pair): windows are identified by content, the pair list indexes them.
        # A single-pair container cannot share windows between pairs: no hashing there (hashing
        # 2 x 47 hour-long windows costs more than correlating them)
        uniq, index = [], {}
        share = len(dataset.pair) > 1

        def slot(x):
            if not share:
                uniq.append(x)
                return len(uniq) - 1
            key = (x.shape[0], _digest(np.ascontiguousarray(x).view(np.uint8)))
            if key not in index:
                index[key] = len(uniq)
                uniq.append(x)
            return index[key]
        pairs = np.array([(slot(a), slot(b)) for a, b in zip(xs, ys)], np.int32)
        # windows gathered straight into page-locked memory (recycled per context): one host copy,
        # H2D at link speed, no per-call page registration
        buf = ctx.pinned_empty((len(uniq), npts), np.float32)
        for i, x in enumerate(uniq):
            buf[i] = x
        out = plan.pairs(buf, pairs, 1, npts)[:, 0]                  # :155-162, batched
        del buf
        for k, (ip, it) in enumerate(todo):
            np.copyto(dataset.cc[ip, it], out[k], casting='same_kind')      # one pass (astype + assign were two)
            dataset.status[ip, it] = 1                       # :165
    if dataset.cc_attrs['bias_correct'] == 1 and dataset.cc_attrs['unbiased'] == 0:
        from .. import ops
        flat = dataset.cc.reshape(-1, dataset.cc.shape[-1])
        dataset.cc[...] = ops.unbias(ctx or default_context(), flat, npts,
                                     imin, imax).reshape(dataset.cc.shape)
        dataset.cc_attrs['unbiased'] = np.byte(1)            # :178-180
    return dataset


def process_batched(traces, sampling_rate, window_length, window_overlap=0.5,
                    pairs=None, clip_lag=None, normalize=True, onebit=False,
                    whiten=None, demean=False, stack=False, ctx=None):
    """Device-resident fast path for regularly sampled, gap-free records:
    traces [S, T] (numpy) -> cc [P, W, L] (or [P, L] with stack=True).
    Windows start at k*hop (all whole windows inside the record); equivalent
    to ``init`` + ``process`` on an :class:`ArrayClient` over the same
    windows, without the per-window host copies."""
    from .init import init
    traces = np.ascontiguousarray(traces, dtype=np.float32)
    S, T = traces.shape
    n = int(window_length * sampling_rate)
    hop = int(round(n * (1 - window_overlap)))
    W = (T - n) // hop + 1
    if pairs is None:
        pairs = [(i, j) for i in range(S) for j in range(i + 1, S)]
    lay = init('x', 0., 1., sampling_rate=sampling_rate, window_length=window_length,
               window_overlap=window_overlap, clip_lag=clip_lag)
    imin, imax = lay.lag_attrs['index_min'], lay.lag_attrs['index_max']
    ctx = ctx or default_context()
    plan = CCPlan(ctx, n, index_min=imin, index_max=imax, normalize=normalize,
                  onebit=onebit, whiten=whiten, demean=demean, stack=stack)
    out = plan.pairs(traces, np.asarray(pairs, np.int32), W, hop)
    plan.close()
    return out, lay.lag
