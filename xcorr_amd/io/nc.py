import datetime as _dt
import hashlib
import os
import tempfile
import warnings

import numpy as np

from ..core.init import CCDataset

__all__ = ['ncfile', 'write', 'read', 'validate', 'merge', 'mfread']

_EPOCH = 'seconds since 1970-01-01 00:00:00'


def _utc(t):
    """POSIX seconds / datetime64 / datetime / pandas.Timestamp -> aware datetime."""
    if isinstance(t, (int, float, np.integer, np.floating)):
        return _dt.datetime.fromtimestamp(float(t), _dt.timezone.utc)
    if isinstance(t, np.datetime64):
        return _dt.datetime.fromtimestamp(float((t - np.datetime64(0, 's')) / np.timedelta64(1, 's')), _dt.timezone.utc)
    if hasattr(t, 'to_pydatetime'):
        t = t.to_pydatetime()
    if isinstance(t, _dt.datetime):
        return t if t.tzinfo else t.replace(tzinfo=_dt.timezone.utc)
    if isinstance(t, str):
        from ..core.init import _seconds
        return _dt.datetime.fromtimestamp(_seconds(t)[0], _dt.timezone.utc)
    raise TypeError('``time`` should be a timestamp')


def ncfile(pair, time, root=None):
    """``{root}/{year}/{pair}/{pair}.{year}.{doy:03d}.nc`` (io/filesystem.py:60-95)."""
    if not isinstance(pair, str):
        raise TypeError('``pair`` should be of type `str`!')
    if root and not isinstance(root, str):
        raise TypeError('``root`` should be of type `str`!')
    t = _utc(time)
    year = f'{t.year:04d}'
    base = os.path.join(root, year, pair) if root else os.path.join(year, pair)
    return os.path.join(base, '{p}.{y:04d}.{d:03d}.nc'.format(p=pair, y=t.year, d=t.timetuple().tm_yday))


# The digest lives under its OWN attribute name: the reference's ``sha256_hash`` is ``hash_Dataset`` over
# metadata and every variable of a NetCDF-4 file (xcorr/io/write.py:88-137) -- a different scheme over a
# different container, so reusing the name would make each side reject the other's files as corrupt.
HASH_ATTR = 'xcorr_amd_sha256'
# Files of earlier builds of this writer carry ``sha256_hash`` (their two-variable digest); the reference writes
# ``sha256_hash`` and ``sha256_hash_metadata`` (io/write.py:98-137).  Neither is this module's digest: they are
# dropped on read and never written back, so no file leaves here with a digest that no longer matches its data.
_STALE_HASH_ATTRS = ('sha256_hash', 'sha256_hash_metadata')


def _data_hash(ds):
    """sha256 over every variable and coordinate of the dataset (native byte order, C order)."""
    h = hashlib.sha256()
    h.update('\x00'.join(ds.pair).encode())
    for name in ('time', 'lag', 'cc', 'status', 'pair_offset', 'time_offset'):
        a = np.ascontiguousarray(getattr(ds, name))
        h.update(name.encode())
        h.update(a.astype(a.dtype.newbyteorder('=')).view(np.uint8))
    if getattr(ds, 'hash', None) is not None:
        h.update(b'hash')
        for row in ds.hash:
            for v in row:
                h.update(str(v).ljust(64)[:64].encode())
    return h.hexdigest()


def _attr(v):
    if isinstance(v, (str, bytes)):
        return v
    if isinstance(v, (dict, list, tuple)):
        import json
        return json.dumps(v, default=str)
    a = np.asarray(v)
    if a.dtype == object or a.dtype.kind in 'US':
        return str(v)
    if a.size == 0:
        return ''
    if a.dtype.kind == 'b':
        return a.astype(np.int8)
    if a.dtype.kind in 'iu' and a.dtype.itemsize > 4:          # classic NetCDF has no 64-bit integers
        return a.astype(np.int32) if np.all(np.abs(a) < 2 ** 31) else a.astype(np.float64)
    if a.dtype.kind == 'u':
        return a.astype(np.int32)
    return a


def write(data, path, close=True, force_write=False, hash_data=True, verb=0):
    """Write a :class:`CCDataset` to `path` through a temporary file that replaces the destination
    (io/write.py:112-184).  Like the reference, a dataset without a single processed step
    (no ``status == 1``) is not written unless ``force_write`` (io/write.py:96-104).
    Returns True when a file was written."""
    from scipy.io import netcdf_file
    if not isinstance(data, CCDataset):
        raise TypeError('data should be a CCDataset')
    if not force_write and not np.any(data.status == 1):
        warnings.warn('Dataset contains no data. No need to save it.', UserWarning)
        return False
    abspath, file = os.path.split(os.path.abspath(path))
    os.makedirs(abspath, exist_ok=True)
    fd, tmp = tempfile.mkstemp(prefix=file + '.', suffix='.tmp', dir=abspath)
    os.close(fd)
    try:
        with netcdf_file(tmp, 'w', version=2) as nc:
            names = [str(p) for p in data.pair]
            slen = max(1, max(len(n) for n in names))
            nc.createDimension('pair', len(names))
            nc.createDimension('time', len(data.time))
            nc.createDimension('lag', len(data.lag))
            nc.createDimension('pair_strlen', slen)
            v = nc.createVariable('pair', 'c', ('pair', 'pair_strlen'))
            v[:] = np.array([list(n.ljust(slen)) for n in names], dtype='S1')
            for k, a in getattr(data, 'pair_attrs', {}).items():
                setattr(v, k, _attr(a))
            v = nc.createVariable('time', 'd', ('time',))
            v[:] = np.asarray(data.time, dtype=np.float64)
            v.units = _EPOCH if getattr(data, 'time_is_datetime', False) else 's'
            for k, a in data.time_attrs.items():
                setattr(v, k, _attr(a if a is not None else 'None'))
            v = nc.createVariable('lag', 'd', ('lag',))
            v[:] = data.lag
            for k, a in data.lag_attrs.items():
                setattr(v, k, _attr(a))
            v = nc.createVariable('cc', data.cc.dtype.char, ('pair', 'time', 'lag'))
            v[:] = data.cc
            for k, a in data.cc_attrs.items():
                setattr(v, k, _attr(a))
            v = nc.createVariable('status', 'b', ('pair', 'time'))
            v[:] = data.status
            v.long_name, v.standard_name, v.units = 'processing status', 'processing_status', '-'
            v.flag_values = np.array([-1, 0, 1], np.byte)
            v.flag_meanings = 'missing_data not_processed processed'
            for name, desc in (('pair_offset', 'offset = receiver[0].starttime - receiver[1].starttime'),
                               ('time_offset', 'offset = receiver[0].starttime - time + window_length/2')):
                v = nc.createVariable(name, 'd', ('pair', 'time'))
                v[:] = getattr(data, name)
                v.units, v.description = 's', desc
            if getattr(data, 'hash', None) is not None:            # core/init.py:252-272
                nc.createDimension('hash_strlen', 64)
                v = nc.createVariable('hash', 'c', ('pair', 'time', 'hash_strlen'))
                v[:] = np.array([[list(str(h).ljust(64)[:64]) for h in row] for row in data.hash], dtype='S1')
                v.long_name, v.standard_name, v.units = ('pair preprocessed stream hash',
                                                         'pair_preprocessed_stream_hash', '-')
            for k, a in data.attrs.items():
                if k != HASH_ATTR and k not in _STALE_HASH_ATTRS:     # never a digest computed over other data
                    setattr(nc, k, _attr(a))
            nc.Conventions = 'CF-1.9'
            nc.xcorr_amd_format = 'NetCDF-3 64-bit offset; layout of xcorr/core/init.py:237-320'
            if hash_data:
                setattr(nc, HASH_ATTR, _data_hash(data))
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise
    if verb > 0:
        print(f'Write data as "{path}": done')
    return True


def read(path, fast=False, quick_and_dirty=False, extract=False):
    """Open a file written by :func:`write` -> :class:`CCDataset`, or None when it does not exist
    (io/read.py returns ``None`` for a missing file; core/lazy.py:53-56 relies on it).
    The data hash is verified unless ``quick_and_dirty``.  ``extract`` masks the estimates of
    steps with ``status != 1`` with NaN (io/read.py:88-90)."""
    from scipy.io import netcdf_file
    if not os.path.isfile(path):
        return None
    with netcdf_file(path, 'r', mmap=False) as nc:
        def attrs(v, skip=()):
            out = {}
            for k, a in v._attributes.items():
                if k in skip:
                    continue
                out[k] = a.decode() if isinstance(a, bytes) else (a.item() if getattr(a, 'size', 0) == 1 and np.ndim(a) == 0 else a)
            return out
        pairs = [b''.join(row).decode().rstrip() for row in nc.variables['pair'][:]]
        tv = nc.variables['time']
        la = attrs(nc.variables['lag'])
        for k in ('npts', 'index_min', 'index_max'):
            la[k] = int(np.asarray(la[k]).reshape(-1)[0])
        for k in ('sampling_rate', 'delta'):
            la[k] = float(np.asarray(la[k]).reshape(-1)[0])
        ca = attrs(nc.variables['cc'])
        for k in ('normalize', 'bias_correct', 'unbiased'):
            if k in ca:
                ca[k] = np.byte(np.asarray(ca[k]).reshape(-1)[0])
        ta = attrs(tv, skip=('units',))
        for k in ('window_length', 'window_overlap'):
            if k in ta:
                ta[k] = float(np.asarray(ta[k]).reshape(-1)[0])
        if ta.get('closed') == 'None':
            ta['closed'] = None
        cc = np.array(nc.variables['cc'][:])
        cc = cc.astype(cc.dtype.newbyteorder('='))             # the file is big-endian
        ds = CCDataset(pairs, np.array(tv[:], dtype=np.float64), np.array(nc.variables['lag'][:], dtype=np.float64), la, ca, ta,
                       dtype=cc.dtype, hash_waveforms='hash' in nc.variables)
        if ds.hash is not None:
            hv = nc.variables['hash'][:]
            for i in range(hv.shape[0]):
                for j in range(hv.shape[1]):
                    ds.hash[i, j] = b''.join(hv[i, j]).decode().rstrip()
        ds.cc[...] = cc
        ds.status[...] = nc.variables['status'][:]
        ds.pair_offset[...] = nc.variables['pair_offset'][:]
        ds.time_offset[...] = nc.variables['time_offset'][:]
        ds.time_is_datetime = attrs(tv).get('units') == _EPOCH
        ds.pair_attrs = attrs(nc.variables['pair'])
        ga = attrs(nc)
        stored = ga.pop(HASH_ATTR, None)
        for k in _STALE_HASH_ATTRS:       # digests of another scheme (earlier builds of this writer, or the
            ga.pop(k, None)               # reference's): not verifiable here, and stale once rewritten
        ds.attrs = ga
    if stored is not None and not quick_and_dirty and stored != _data_hash(ds):
        raise ValueError(f'{path}: data sha256 hash does not match the stored one')
    if extract:
        ds.cc[ds.status != 1] = np.nan
    return ds


def merge(*datasets, extract=False, quick_and_dirty=False):
    """Combine containers (or paths / glob strings of files written by :func:`write`) by their
    coordinates -- ``xarray.combine_by_coords(join='outer')`` of core/merge.py:53-79 for this
    layout: the union of the pairs and of the window times (sorted), one common lag axis; steps no
    input holds keep ``status`` 0 and zero estimates (NaN with ``extract``); a step present in
    several inputs is taken from the one where it is processed.  Global attributes of the first
    input, ``title`` extended by the covered day range, ``history`` = 'Merged @ <now>'."""
    import glob as _glob
    dsets = []
    for d in datasets:
        if isinstance(d, CCDataset):
            dsets.append(d)
        elif isinstance(d, (str, os.PathLike)):
            paths = sorted(_glob.glob(str(d))) if any(c in str(d) for c in '*?[') else [str(d)]
            for pth in paths:
                r = read(pth, quick_and_dirty=quick_and_dirty)
                if r is not None:
                    dsets.append(r)
        elif isinstance(d, (list, tuple)):
            sub = merge(*d, quick_and_dirty=quick_and_dirty) if d else None
            if sub is not None:
                dsets.append(sub)
        else:
            raise TypeError('datasets should be CCDataset objects or paths')
    if not dsets:
        return None
    first = dsets[0]
    for d in dsets[1:]:
        if d.lag.shape != first.lag.shape or not np.allclose(d.lag, first.lag, rtol=0, atol=1e-9):
            raise ValueError('datasets have different lag axes')
        for k in ('sampling_rate', 'npts', 'index_min', 'index_max'):
            if d.lag_attrs[k] != first.lag_attrs[k]:
                raise ValueError(f'datasets differ in lag attribute "{k}"')
    pairs = sorted({str(p) for d in dsets for p in d.pair})
    times = np.unique(np.concatenate([np.asarray(d.time, np.float64) for d in dsets]))
    out = CCDataset(pairs, times, first.lag, first.lag_attrs, first.cc_attrs, first.time_attrs, dtype=first.cc.dtype,
                    hash_waveforms=any(getattr(d, 'hash', None) is not None for d in dsets))
    out.time_is_datetime = getattr(first, 'time_is_datetime', False)
    out.pair_attrs = dict(getattr(first, 'pair_attrs', {}))
    pidx = {p: i for i, p in enumerate(pairs)}
    for d in dsets:
        ti = np.searchsorted(times, np.asarray(d.time, np.float64))
        for ip, p in enumerate(d.pair):
            row = pidx[str(p)]
            take = (d.status[ip] == 1) | (out.status[row, ti] != 1)       # a processed step is never overwritten by a pending one
            cols = ti[take]
            out.cc[row, cols] = d.cc[ip, take]
            out.status[row, cols] = d.status[ip, take]
            out.pair_offset[row, cols] = d.pair_offset[ip, take]
            out.time_offset[row, cols] = d.time_offset[ip, take]
            if out.hash is not None and getattr(d, 'hash', None) is not None:
                out.hash[row, cols] = d.hash[ip, take]
    out.attrs = {k: v for k, v in first.attrs.items() if k != HASH_ATTR and k not in _STALE_HASH_ATTRS}
    t0, t1 = _utc(times[0]), _utc(times[-1])
    s0, s1 = t0.strftime('%Y.%j'), t1.strftime('%Y.%j')
    title = str(out.attrs.get('title', '')).split(' - ')[0]
    out.attrs['title'] = (title + ' - ' + s0 + (' to {}'.format(s1) if s0 != s1 else '')).strip()
    out.attrs['history'] = 'Merged @ {}'.format(_dt.datetime.now(_dt.timezone.utc).strftime('%Y-%m-%d %H:%M:%S'))
    if extract:
        out.cc[out.status != 1] = np.nan
    return out


def mfread(paths, extract=False, quick_and_dirty=True):
    """Read several files (glob string or list) as ONE container (io/read.py:95-200: the
    reference opens them lazily with ``xarray.open_mfdataset(combine='by_coords', join='outer')``
    after ``validate(quick_and_dirty=True)``; here they are read and merged eagerly)."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    return merge(*paths, extract=extract, quick_and_dirty=quick_and_dirty)


def validate(ds, raise_error=True):
    """Shape / coordinate consistency of a container (the checks of io/validate.py that do not
    need xarray): cc(pair,time,lag), lag slice attrs, status range."""
    ok = (ds.cc.shape == (len(ds.pair), len(ds.time), len(ds.lag))
          and ds.status.shape == ds.cc.shape[:2]
          and ds.lag_attrs['index_max'] - ds.lag_attrs['index_min'] == len(ds.lag)
          and np.all(np.isin(ds.status, (-1, 0, 1))))
    if not ok and raise_error:
        raise ValueError('dataset layout is not a valid xcorr cross-correlation container')
    return bool(ok)
