"""
Host-side handles over the C ABI: Context (one HIP device + stream),
DeviceArray (caller-owned HBM buffer) and CCPlan (cross-correlation plan).

This is plumbing only; all arithmetic happens in libxcx.so.
"""
import ctypes as C
import os
import time
import weakref

import numpy as np

from . import _ffi
from ._ffi import check, ptr

__all__ = ['Context', 'DeviceArray', 'CCPlan', 'default_context']


class DeviceArray:
    """A contiguous device buffer with numpy-like shape/dtype metadata."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = ctx._pool_take(self.nbytes)
        if self.ptr is None:
            p = C.c_void_p()
            check(_ffi.lib().xcx_malloc(ctx.handle, self.nbytes, C.byref(p)))
            self.ptr = p.value
        ctx._children.add(self)

    def columns_from(self, col0):
        """Non-owning view of a 2-D buffer starting at column `col0`: shape [rows, cols-col0],
        same row pitch (pass ``stride_s=self.shape[-1]`` to the consumer)."""
        if len(self.shape) != 2 or not 0 <= col0 < self.shape[1]:
            raise ValueError('columns_from needs a 2-D buffer and a column inside it')
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.dtype = self.ctx, self.dtype
        v.shape = (self.shape[0], self.shape[1] - int(col0))
        v.nbytes = 0
        v.ptr = self.ptr + int(col0) * self.dtype.itemsize
        v._owner = self                      # keeps the allocation alive; the view frees nothing
        v._pitch = getattr(self, '_pitch', self.shape[1] * self.dtype.itemsize)    # bytes from row to row
        return v

    def free(self):
        if getattr(self, '_owner', None) is not None:
            self.ptr = None
            return
        # a context that is already closed has released everything it owned
        # (finalisers may run in any order at interpreter exit)
        if self.ptr and self.ctx.handle:
            if not self.ctx._pool_give(self.ptr, self.nbytes):
                check(_ffi.lib().xcx_free(self.ctx.handle, C.c_void_p(self.ptr)))
        self.ptr = None

    close = free

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def copy_from(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError('size mismatch in copy_from')
        check(_ffi.lib().xcx_memcpy_h2d(self.ctx.handle, C.c_void_p(self.ptr),
                                        ptr(a), self.nbytes))
        return self

    def to_host(self):
        # page-locked from the context's pool for a megabyte and more: D2H at link speed
        out = (self.ctx.pinned_empty(self.shape, self.dtype) if self.nbytes >= (1 << 20)
               else np.empty(self.shape, dtype=self.dtype))
        check(_ffi.lib().xcx_memcpy_d2h(self.ctx.handle, ptr(out),
                                        C.c_void_p(self.ptr), self.nbytes))
        return out

    def copy_to(self, out):
        """Copy into an existing host array of the same size (e.g. page-locked, `Context.pinned_empty`)."""
        if out.nbytes != self.nbytes or not out.flags['C_CONTIGUOUS']:
            raise ValueError('copy_to needs a contiguous host array of the same size')
        check(_ffi.lib().xcx_memcpy_d2h(self.ctx.handle, ptr(out), C.c_void_p(self.ptr), self.nbytes))
        return out

    def rows_to_host(self, row0, nrows, width):
        """Copy rows [row0, row0+nrows) of a 2-D buffer, first `width` cols."""
        pitch = getattr(self, '_pitch', self.shape[-1] * self.dtype.itemsize)       # a column view keeps its owner's pitch
        if width > self.shape[-1] or row0 < 0 or row0 + nrows > self.shape[0]:
            raise ValueError('rows_to_host: rows / width outside the buffer')
        out = np.empty((nrows, width), dtype=self.dtype)
        check(_ffi.lib().xcx_memcpy2d_d2h(
            self.ctx.handle, ptr(out), width * self.dtype.itemsize,
            C.c_void_p(self.ptr + row0 * pitch), pitch,
            width * self.dtype.itemsize, nrows))
        return out

    def zero(self):
        check(_ffi.lib().xcx_memset(self.ctx.handle, C.c_void_p(self.ptr), 0,
                                    self.nbytes))
        return self


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        check(_ffi.lib().xcx_init(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self._children = weakref.WeakSet()      # plans and arrays that must go first
        self._pin_pool = {}                     # capacity -> [addresses] of idle page-locked blocks
        self._pin_idle = 0                      # bytes held by those idle blocks
        self._pin_live = 0                      # bytes of page-locked blocks inside live result arrays
        self._plans = {}                        # (n, lag slice, flags) -> CCPlan kept for repeated driver calls
        # Freed device buffers of a chain of row ops are handed to the next result of the same size instead of going
        # through hipFree / hipMalloc (milliseconds each for day-long rows, and hipFree drains the device): exact-size
        # free lists in least-recently-freed order, bounded three ways -- XCX_DEV_POOL_MB (default 2048; 0 = off;
        # single blocks up to half of it; the oldest blocks make room for a new one), an age limit (blocks idle for
        # XCX_DEV_POOL_AGE_S seconds, default 30, go back to the driver at the next take / give: a chain whose row
        # lengths keep changing -- trim, decimate, resample -- never reuses its sizes), and the library itself, which
        # asks for everything back before one of its own allocations fails or a segmented plan sizes its workspace
        # from the free device memory (xcx_set_reclaim).
        self._dev_pool = {}                     # nbytes -> [(device pointer, time freed)], oldest first
        self._dev_pool_bytes = 0
        self._dev_pool_cap = max(0, int(os.environ.get('XCX_DEV_POOL_MB', '2048'))) << 20
        self._dev_pool_age = float(os.environ.get('XCX_DEV_POOL_AGE_S', '30'))
        wself = weakref.ref(self)                # no cycle through the callback object

        def _cb(_user):
            c = wself()
            return c._reclaim() if c is not None else 0
        self._reclaim_cb = _ffi.RECLAIM_CB(_cb)
        check(_ffi.lib().xcx_set_reclaim(self.handle, C.cast(self._reclaim_cb, C.c_void_p), None))

    def _reclaim(self):
        """Called BY the library (xcx_set_reclaim): hand every pooled device buffer back, report the bytes."""
        try:
            got = self._dev_pool_bytes
            self.trim_pool()
            return got
        except Exception:
            return 0

    def _pool_expire(self, now):
        if not self._dev_pool_bytes or self._dev_pool_age <= 0:
            return
        for nbytes in list(self._dev_pool):
            blocks = self._dev_pool[nbytes]
            while blocks and now - blocks[0][1] > self._dev_pool_age:
                p, _ = blocks.pop(0)
                self._dev_pool_bytes -= nbytes
                _ffi.lib().xcx_free(self.handle, C.c_void_p(p))
            if not blocks:
                del self._dev_pool[nbytes]

    def _pool_take(self, nbytes):
        self._pool_expire(time.monotonic())
        blocks = self._dev_pool.get(nbytes)
        if not blocks:
            return None
        p, _ = blocks.pop()                     # the most recently freed block of that size
        if not blocks:
            del self._dev_pool[nbytes]
        self._dev_pool_bytes -= nbytes
        # the previous owner's kernels may still be queued on the context's streams, and host copies run on another
        # (blocking) stream: wait once here -- what hipFree used to do implicitly
        check(_ffi.lib().xcx_sync(self.handle))
        return p

    def _pool_give(self, p, nbytes):
        if nbytes < (1 << 20) or nbytes > self._dev_pool_cap // 2:
            return False
        now = time.monotonic()
        self._pool_expire(now)
        while self._dev_pool_bytes + nbytes > self._dev_pool_cap and self._dev_pool_bytes:
            # least recently freed block of any size makes room
            size = min(self._dev_pool, key=lambda k: self._dev_pool[k][0][1])
            q, _ = self._dev_pool[size].pop(0)
            if not self._dev_pool[size]:
                del self._dev_pool[size]
            self._dev_pool_bytes -= size
            _ffi.lib().xcx_free(self.handle, C.c_void_p(q))
        self._dev_pool.setdefault(nbytes, []).append((p, now))
        self._dev_pool_bytes += nbytes
        return True

    def trim_pool(self):
        """Return every pooled device buffer to the driver."""
        pool, self._dev_pool, self._dev_pool_bytes = self._dev_pool, {}, 0
        for blocks in pool.values():
            for p, _ in blocks:
                _ffi.lib().xcx_free(self.handle, C.c_void_p(p))

    def trim(self):
        """Release everything the context caches between calls: the pooled device buffers of this binding and the
        library's own workspaces, tables and cached plans (xcx_trim).  Plans and arrays of the caller stay."""
        self.trim_pool()
        check(_ffi.lib().xcx_trim(self.handle))

    def plan(self, n, index_min=0, index_max=None, **flags):
        """A :class:`CCPlan` owned by the context and reused by later calls with the same arguments
        (the reference-style driver makes one call per pair and day: plan workspaces are allocated
        once, not per call).  Do not close it; `Context.close` does."""
        key = (int(n), int(index_min), int(index_max or 0), tuple(sorted(flags.items())))
        p = self._plans.get(key)
        if p is None or not p.handle:
            if len(self._plans) >= 8:                       # a handful of geometries at most
                self._plans.pop(next(iter(self._plans))).close()
            p = self._plans[key] = CCPlan(self, n, index_min=index_min, index_max=index_max, **flags)
        return p

    def close(self):
        if self.handle:
            self._plans = {}
            for child in list(self._children):
                child.close()
            self.trim_pool()
            _ffi.lib().xcx_set_reclaim(self.handle, None, None)
            for blocks in self._pin_pool.values():
                for addr in blocks:
                    _ffi.lib().xcx_host_free(self.handle, C.c_void_p(addr))
            self._pin_pool = {}
            self._pin_idle = 0
            _ffi.lib().xcx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def name(self):
        buf = C.create_string_buffer(256)
        check(_ffi.lib().xcx_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def array(self, a, dtype=None):
        a = np.ascontiguousarray(a, dtype=dtype)
        return DeviceArray(self, a.shape, a.dtype).copy_from(a)

    # --- page-locked result arrays ------------------------------------------
    PIN_MAX = 2 << 30          # larger results come back in ordinary (pageable) numpy memory
    PIN_IDLE_MAX = 4 << 30     # idle page-locked bytes kept for reuse, per context
    PIN_LIVE_MAX = 16 << 30    # page-locked bytes in live arrays + idle blocks, per context: beyond it results are pageable
    PIN_GRANULE = 2 << 20

    def pinned_empty(self, shape, dtype=np.float32):
        """numpy array in page-locked host memory (xcx_host_alloc): the library's D2H into it runs at
        PCIe speed.  Blocks are recycled through a small per-context pool when the array is
        garbage-collected (page-locking 100 MB costs as much as copying it).

        Page-locked memory cannot be swapped, so the pool is bounded: requests are rounded up to 2 MiB
        (not to a power of two), results above PIN_MAX and requests the driver refuses fall back to
        ``np.empty``, at most PIN_IDLE_MAX idle bytes (and 4 blocks per size) are kept, and a block
        whose array outlives the context is still freed (hipHostFree needs no context)."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if nbytes > self.PIN_MAX or not self.handle:
            return np.empty(shape, dtype)
        g = self.PIN_GRANULE
        cap = max(g, (nbytes + g - 1) // g * g)
        pool = self._pin_pool.setdefault(cap, [])
        if pool:
            addr = pool.pop()
            self._pin_idle -= cap
        else:
            if self._pin_live + self._pin_idle + cap > self.PIN_LIVE_MAX:
                return np.empty(shape, dtype)            # page-locked memory cannot be swapped: bounded per context
            p = C.c_void_p()
            if _ffi.lib().xcx_host_alloc(self.handle, cap, C.byref(p)) != 0 or not p.value:
                return np.empty(shape, dtype)            # driver refused (limits, fragmentation): pageable result
            addr = p.value
        self._pin_live += cap
        buf = (C.c_char * nbytes).from_address(addr)
        arr = np.frombuffer(buf, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape)
        ctx_ref = weakref.ref(self)
        idle_max = self.PIN_IDLE_MAX

        def _release(addr=addr, cap=cap):
            c = ctx_ref()
            if c is not None:
                c._pin_live -= cap
            if c is not None and c.handle:
                lst = c._pin_pool.setdefault(cap, [])
                if len(lst) < 4 and c._pin_idle + cap <= idle_max:
                    lst.append(addr)
                    c._pin_idle += cap
                    return
            _ffi.lib().xcx_host_free(c.handle if c is not None and c.handle else None, C.c_void_p(addr))
        weakref.finalize(buf, _release)
        return arr

    def sync(self):
        check(_ffi.lib().xcx_sync(self.handle))

    def timer_start(self):
        check(_ffi.lib().xcx_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        check(_ffi.lib().xcx_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    # --- RCCL time-stack reduce -------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(_ffi.lib().xcx_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid, nranks, rank):
        check(_ffi.lib().xcx_comm_init(self.handle, uid, nranks, rank))

    def comm_info(self):
        """(nranks, rank) as RCCL reports them for this context's communicator."""
        n, r = C.c_int(0), C.c_int(0)
        check(_ffi.lib().xcx_comm_info(self.handle, C.byref(n), C.byref(r)))
        return n.value, r.value

    def stack_reduce(self, stack, count=None, reduce_scatter=False):
        n = int(count if count is not None else np.prod(stack.shape))
        check(_ffi.lib().xcx_stack_reduce(self.handle, ptr(stack), n,
                                          int(bool(reduce_scatter))))


def _check_out(out, nrows, ld, L):
    """Refuse an output buffer the library would overrun: it writes `nrows` rows of L float32
    at a pitch of `ld` floats (the last row needs only L)."""
    if ld < L:
        raise ValueError(f'ld_out ({ld}) < L ({L})')
    need = 4 * ((nrows - 1) * ld + L)
    if isinstance(out, DeviceArray):
        if out.dtype != np.float32:
            raise ValueError('`out` must be float32')
        own = getattr(out, '_owner', None)       # a column view: what is left of the owner's allocation
        have = out.nbytes if own is None else own.nbytes - (out.ptr - own.ptr)
    else:
        if not isinstance(out, np.ndarray) or out.dtype != np.float32 or not out.flags['C_CONTIGUOUS']:
            raise ValueError('`out` must be a C-contiguous float32 numpy array or a DeviceArray')
        if not out.flags['WRITEABLE']:
            raise ValueError('`out` is read-only')
        have = out.nbytes
    if have < need:
        raise ValueError(f'`out` holds {have} bytes, the call writes {need} '
                         f'({nrows} rows x {L} lags at a pitch of {ld} floats)')


_default = {}


def default_context(device=0):
    """Process-wide context per device (created on first use)."""
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]


class CCPlan:
    """Plan for the batched cross-correlation (include/xcx.h: xcx_plan_cc1d).

    n: window length; index_min/index_max: retained lag slice of the 2n-1 lags
    (core/process.py:155-162); flags: normalize / unbias / stack / onebit /
    whiten / demean.
    """

    def __init__(self, ctx, n, index_min=0, index_max=None, normalize=True,
                 unbiased=False, stack=False, onebit=False, whiten=None,
                 demean=False, nfft=0, batch=None):
        self.ctx = ctx
        self.n = int(n)
        flags = 0
        flags |= _ffi.NORMALIZE if normalize else 0
        flags |= _ffi.UNBIAS if unbiased else 0
        flags |= _ffi.STACK if stack else 0
        flags |= _ffi.ONEBIT if onebit else 0
        flags |= _ffi.WHITEN if whiten else 0
        flags |= _ffi.DEMEAN if demean else 0
        self.flags = flags
        h = C.c_void_p()
        check(_ffi.lib().xcx_plan_cc1d(ctx.handle, self.n, int(nfft),
                                       int(index_min), int(index_max or 0),
                                       flags, C.byref(h)))
        self.handle = h
        ctx._children.add(self)
        if whiten:
            fs, fmin, fmax, ftaper = whiten
            check(_ffi.lib().xcx_plan_set_whiten(h, fs, fmin, fmax, ftaper))
        if batch:
            check(_ffi.lib().xcx_plan_set_batch(h, int(batch)))
        info = (C.c_int * 6)()
        check(_ffi.lib().xcx_plan_info(h, info))
        self.N, self.N1, self.N2, self.L, self.batch, self.spec_floats = info[:]
        self.stack = bool(stack)

    def close(self):
        if self.handle and self.ctx.handle:
            _ffi.lib().xcx_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def t_storage(self):
        """'q24' when the pair pass keeps its intermediate as 24-bit fixed point (whitened plans), else 'f32'."""
        ts = C.c_int(0)
        check(_ffi.lib().xcx_plan_t_storage(self.handle, C.byref(ts)))
        return 'q24' if ts.value == 1 else 'f32'

    @property
    def zero_skip(self):
        """(first, count): the 64-point blocks first .. count - 1 - first of every spectrum column that the pair pass's
        column kernel knows to be zero and does not load (band-limited q24 plans); first = 0: none."""
        o = (C.c_int * 2)()
        check(_ffi.lib().xcx_plan_zero_skip(self.handle, o))
        return int(o[0]), int(o[1])

    def counters(self):
        a, b = C.c_longlong(), C.c_longlong()
        check(_ffi.lib().xcx_plan_counters(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def profile_step(self, fn, level=2):
        """Run fn() once with HIP-event timing enabled (level 2: every kernel
        launch; level 1: only the pair phase as a whole, which does not perturb
        the overlapping pipelines); returns {kernel: {avg_us, total_us,
        launches, units_per_launch}}."""
        L = _ffi.lib()
        check(L.xcx_plan_profile(self.handle, int(level)))
        fn()
        out = {}
        for kid in range(8):
            name = C.create_string_buffer(64)
            tot, nl, nu = C.c_double(), C.c_longlong(), C.c_longlong()
            check(L.xcx_plan_profile_read(self.handle, kid, name, 64,
                                          C.byref(tot), C.byref(nl),
                                          C.byref(nu)))
            if nl.value:
                out[name.value.decode()] = {
                    'total_us': tot.value, 'launches': nl.value,
                    'avg_us': tot.value / nl.value,
                    'units_per_launch': nu.value / nl.value}
        check(L.xcx_plan_profile(self.handle, 0))
        return out

    def prepare(self, traces, W, stride_w, stride_s=None):
        """Start transforming all windows of the device-resident `traces` [S, T] on a side
        stream; the next matching pairs() call uses the result (xcx_cc1d_prepare)."""
        if not isinstance(traces, DeviceArray):
            raise ValueError('prepare() needs device-resident traces')
        S, T = traces.shape
        stride_s = T if stride_s is None else stride_s
        if (W - 1) * stride_w + self.n > T:
            raise ValueError('windows exceed the trace length')
        check(_ffi.lib().xcx_cc1d_prepare(self.handle, ptr(traces), S, int(W), int(stride_s), int(stride_w)))

    def prepare_sync(self):
        """Block until the last prepare() has finished reading its traces."""
        check(_ffi.lib().xcx_cc1d_prepare_sync(self.handle))

    def pairs(self, traces, pairs, W, stride_w, out=None, ld_out=0,
              stride_s=None):
        """All pairs x W windows.  `traces`: numpy [S, T] or DeviceArray [S, T];
        window w of station s = traces[s, w*stride_w : w*stride_w + n].
        Returns `out` (numpy [P, W, L] / [P, L] if stacking, or the DeviceArray
        passed in)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        P = pairs.shape[0]
        S, T = traces.shape
        stride_s = T if stride_s is None else stride_s
        if (W - 1) * stride_w + self.n > T:
            raise ValueError('windows exceed the trace length')
        on_dev = isinstance(traces, DeviceArray)
        nrows = P if self.stack else P * W
        if on_dev:
            if not isinstance(out, DeviceArray):
                raise ValueError('device traces need a DeviceArray `out`')
        else:
            traces = np.ascontiguousarray(traces, dtype=np.float32)
            if out is None:
                shape = (P, self.L) if self.stack else (P, W, self.L)
                out = self.ctx.pinned_empty(shape)
        _check_out(out, nrows, int(ld_out) or self.L, self.L)
        check(_ffi.lib().xcx_cc1d_pairs(
            self.handle, ptr(traces), S, W, stride_s, stride_w, ptr(pairs), P,
            ptr(out), int(ld_out), int(on_dev)))
        return out

    def stack_add(self, traces, pairs, W, stride_w, stride_s=None):
        """Accumulate the cross-spectra of W windows of every pair (xcx_cc1d_stack_add): one
        call per station-day of a multi-day stack; finish with :meth:`stack_finish`."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        S, T = traces.shape
        stride_s = T if stride_s is None else stride_s
        if (W - 1) * stride_w + self.n > T:
            raise ValueError('windows exceed the trace length')
        on_dev = isinstance(traces, DeviceArray)
        if not on_dev:
            traces = np.ascontiguousarray(traces, dtype=np.float32)
        check(_ffi.lib().xcx_cc1d_stack_add(self.handle, ptr(traces), S, int(W), int(stride_s),
                                            int(stride_w), ptr(pairs), len(pairs), int(on_dev)))

    def stack_reset(self):
        """Abandon an open accumulation (xcx_cc1d_stack_reset): call it from the error path of a driver loop."""
        check(_ffi.lib().xcx_cc1d_stack_reset(self.handle))

    def stack_finish(self, pairs, out=None, ld_out=0):
        """Inverse-transform the accumulated cross-spectra -> (out [P, L], windows summed)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        P = len(pairs)
        if out is None:
            out = self.ctx.pinned_empty((P, self.L))
        on_dev = isinstance(out, DeviceArray)
        _check_out(out, P, int(ld_out) or self.L, self.L)
        nw = C.c_longlong(0)
        check(_ffi.lib().xcx_cc1d_stack_finish(self.handle, ptr(pairs), P, ptr(out), int(ld_out),
                                               int(on_dev), C.byref(nw)))
        return out, nw.value

    def stack_finish_reduce(self, pairs, out, ld_out=0):
        """The multi-rank finish of an accumulation (xcx_cc1d_stack_finish_reduce): the ranks' cross-spectrum
        accumulators are summed in the frequency domain with one RCCL reduce-scatter of their in-band elements and
        every rank inverse-transforms only the pairs it receives.  `out`: DeviceArray [P, ld]; returns
        (owned, windows): owned[i] is True for the pairs whose rows of `out` this rank wrote (the finished stacks)."""
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        P = len(pairs)
        if not isinstance(out, DeviceArray):
            raise ValueError('stack_finish_reduce needs a DeviceArray `out`')
        _check_out(out, P, int(ld_out) or self.L, self.L)
        owned = np.zeros(P, np.uint8)
        nw = C.c_longlong(0)
        check(_ffi.lib().xcx_cc1d_stack_finish_reduce(self.handle, ptr(pairs), P, ptr(out), int(ld_out), ptr(owned),
                                                      C.byref(nw)))
        return owned.astype(bool), nw.value

    def cc(self, x, y, out=None):
        """Batched two-input form: x, y [batch, n] -> [batch, L]."""
        on_dev = isinstance(x, DeviceArray)
        if not on_dev:
            x = np.ascontiguousarray(x, dtype=np.float32)
            y = np.ascontiguousarray(y, dtype=np.float32)
        if x.shape != y.shape or x.shape[-1] != self.n:
            raise ValueError('x and y must both be [batch, n]')
        batch = int(np.prod(x.shape[:-1])) if len(x.shape) > 1 else 1
        if out is None:
            if on_dev:
                out = self.ctx.empty((batch, self.L))
            else:
                out = self.ctx.pinned_empty(x.shape[:-1] + (self.L,))
        _check_out(out, batch, self.L, self.L)
        check(_ffi.lib().xcx_cc1d(self.handle, ptr(x), ptr(y), batch, ptr(out),
                                  int(on_dev)))
        return out
