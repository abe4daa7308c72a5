"""
ctypes binding of libxcx.so (C ABI: include/xcx.h).

The HIP library is the product path.  There is NO CPU fallback: if the shared
library is missing or no GPU is visible, importing works (so that host-side
logic and the ABI surface can be tested without a GPU) but every compute call
raises ``XcxError`` loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('XCX_LIB_PATH') or os.path.join(_HERE, 'lib', 'libxcx.so')   # override: A/B builds

# plan flags (include/xcx.h)
NORMALIZE, UNBIAS, STACK, ONEBIT, WHITEN, DEMEAN = 1, 2, 4, 8, 16, 32

XCX_OK, XCX_EINVAL, XCX_ENOMEM, XCX_EHIP, XCX_EUNSUPPORTED, XCX_ENODEV, \
    XCX_ERCCL = 0, -1, -2, -3, -4, -5, -6


class XcxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libxcx error {code}: {msg}')
        self.code = code


_vp, _i, _l, _sz = C.c_void_p, C.c_int, C.c_long, C.c_size_t
RECLAIM_CB = C.CFUNCTYPE(C.c_size_t, C.c_void_p)        # xcx_set_reclaim callback: size_t cb(void* user)
_pf, _pi, _pd = C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_double)

# name -> argtypes; every function returns int.  Kept in one table so that
# tests can check the library exports exactly what include/xcx.h declares.
SIGNATURES = {
    'xcx_device_count': [_pi],
    'xcx_limits': [C.POINTER(C.c_long), _i],
    'xcx_plan_geometry': [C.c_longlong, _i, _pi],
    'xcx_plan_t_storage': [_vp, _pi],
    'xcx_plan_zero_skip': [_vp, _pi],
    'xcx_plan_fused_list': [_pi, _i, _pi],
    'xcx_init': [_i, C.POINTER(_vp)],
    'xcx_destroy': [_vp],
    'xcx_last_error': [C.c_char_p, _sz],
    'xcx_device_name': [_vp, C.c_char_p, _sz],
    'xcx_malloc': [_vp, _sz, C.POINTER(_vp)],
    'xcx_free': [_vp, _vp],
    'xcx_set_reclaim': [_vp, _vp, _vp],
    'xcx_trim': [_vp],
    'xcx_host_alloc': [_vp, _sz, C.POINTER(_vp)],
    'xcx_host_free': [_vp, _vp],
    'xcx_memset': [_vp, _vp, _i, _sz],
    'xcx_memcpy_h2d': [_vp, _vp, _vp, _sz],
    'xcx_memcpy_d2h': [_vp, _vp, _vp, _sz],
    'xcx_memcpy_d2d': [_vp, _vp, _vp, _sz],
    'xcx_stream_write_rate': [_vp, _vp, C.c_size_t, _i, C.POINTER(C.c_float)],
    'xcx_memcpy2d_d2h': [_vp, _vp, _sz, _vp, _sz, _sz, _sz],
    'xcx_sync': [_vp],
    'xcx_timer_start': [_vp],
    'xcx_timer_stop': [_vp, _pf],
    'xcx_plan_cc1d': [_vp, _i, _i, _i, _i, _i, C.POINTER(_vp)],
    'xcx_plan_destroy': [_vp],
    'xcx_plan_set_whiten': [_vp, C.c_double, C.c_double, C.c_double,
                            C.c_double],
    'xcx_plan_set_batch': [_vp, _i],
    'xcx_pair_groups': [_vp, _i, _i, _i, _vp, _vp, C.POINTER(_i)],
    'xcx_plan_info': [_vp, _pi],
    'xcx_cc1d_pairs': [_vp, _vp, _i, _i, _l, _l, _vp, _i, _vp, _l, _i],
    'xcx_cc1d': [_vp, _vp, _vp, _i, _vp, _i],
    'xcx_plan_counters': [_vp, C.POINTER(C.c_longlong),
                          C.POINTER(C.c_longlong)],
    'xcx_plan_profile': [_vp, _i],
    'xcx_plan_profile_read': [_vp, _i, C.c_char_p, _sz, _pd,
                              C.POINTER(C.c_longlong),
                              C.POINTER(C.c_longlong)],
    'xcx_norm1d': [_vp, _vp, _l, _l, _i],
    'xcx_demean': [_vp, _vp, _l, _l, _i],
    'xcx_detrend_linear': [_vp, _vp, _l, _l, _i],
    'xcx_detrend_to': [_vp, _vp, _vp, _l, _l, _i],
    'xcx_detrend_to_cols': [_vp, _vp, _vp, _l, _l, _i, _l, _l],
    'xcx_taper': [_vp, _vp, _l, _l, _l, _i, _i],
    'xcx_unbias': [_vp, _vp, _l, _l, _l, _l, _i],
    'xcx_onebit': [_vp, _vp, _l, _l, _i],
    'xcx_scale_rows': [_vp, _vp, _l, _l, _vp, _i],
    'xcx_rotate2': [_vp, _vp, _vp, _l, _l, _vp, _i],
    'xcx_absmax_rows': [_vp, _vp, _l, _l, _vp, _i],
    'xcx_detrend_nan': [_vp, _vp, _l, _l, _i, _i],
    'xcx_taper_table': [_vp, _vp, _l, _l, _vp, _l, _i],
    'xcx_post_prep': [_vp, _vp, _l, _l, _l, _l, _i, _l, _i, _i],
    'xcx_sosfiltfilt': [_vp, _vp, _l, _l, _vp, _i, _i],
    'xcx_sosfilt': [_vp, _vp, _l, _l, _vp, _i, _i],
    'xcx_sosfilt_warmup': [_vp, _i, C.POINTER(_l)],
    'xcx_sosfilt_cols': [_vp, _vp, _l, _l, _vp, _i, _l, _l, C.POINTER(_l)],
    'xcx_sosfilt_zerophase': [_vp, _vp, _l, _l, _vp, _i, _i],
    'xcx_snr': [_vp, _vp, _l, _l, _vp, _vp, _vp, _vp, _vp, _i],
    'xcx_timeshift': [_vp, _vp, _vp, _l, _l, _vp, C.c_double, _i],
    'xcx_timeshift_fft': [_vp, _vp, _vp, _l, _l, _vp, C.c_double, _i],
    'xcx_cc2d': [_vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i],
    'xcx_fft': [_vp, _vp, _vp, _l, _l, _i],
    'xcx_ifft': [_vp, _vp, _vp, _l, _l, _i],
    'xcx_rfft': [_vp, _vp, _vp, _l, _l, _i],
    'xcx_irfft': [_vp, _vp, _vp, _l, _l, _i],
    'xcx_resample': [_vp, _vp, _vp, _l, _l, _l, C.c_double, _vp, _i],
    'xcx_hilbert': [_vp, _vp, _vp, _l, _l, _i, _i],
    'xcx_decimate': [_vp, _vp, _vp, _l, _l, _l, _vp, _i, _i],
    'xcx_lanczos': [_vp, _vp, _vp, _l, _l, _l, C.c_double, C.c_double, _i, _i],
    'xcx_block_rms': [_vp, _vp, _vp, _l, _l, _l, _l, _l, _i],
    'xcx_spectrogram': [_vp, _vp, _vp, _l, _l, _i, _i, _vp, C.c_double, _i, _i],
    'xcx_spectrogram_seg': [_vp, _vp, _vp, _l, _l, _i, _l, _i, _vp, C.c_double, _i, _i],
    'xcx_ram_normalize': [_vp, _vp, _l, _l, _l, _i],
    'xcx_cc1d_stack_add': [_vp, _vp, _i, _i, _l, _l, _vp, _i, _i],
    'xcx_cc1d_stack_finish': [_vp, _vp, _i, _vp, _l, _i, _vp],
    'xcx_cc1d_stack_finish_reduce': [_vp, _vp, _i, _vp, _l, _vp, _vp],
    'xcx_cc1d_stack_reset': [_vp],
    'xcx_cc1d_prepare': [_vp, _vp, _i, _i, _l, _l],
    'xcx_cc1d_prepare_sync': [_vp],
    'xcx_comm_unique_id': [C.c_char_p],
    'xcx_comm_init': [_vp, C.c_char_p, _i, _i],
    'xcx_comm_destroy': [_vp],
    'xcx_comm_info': [_vp, C.POINTER(_i), C.POINTER(_i)],
    'xcx_stack_reduce': [_vp, _vp, _sz, _i],
}

_lib = None


def lib():
    """Load libxcx.so (once).  Raises XcxError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XcxError(XCX_ENODEV, (
                f'{LIB_PATH} not found: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C xcorr_amd/csrc -j8`.  xcorr_amd has no CPU fallback.'))
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().xcx_last_error(buf, 512)
    return buf.value.decode(errors='replace')


def check(rc):
    if rc != 0:
        raise XcxError(rc, last_error())


def ptr(a):
    """Pointer of a numpy array / DeviceArray / int / None as c_void_p."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(_vp)
    if hasattr(a, 'ptr'):
        return _vp(a.ptr)
    return _vp(int(a))
