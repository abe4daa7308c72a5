"""CPU: the C-ABI library loads and exports exactly what include/xcx.h
declares; no compute calls are made (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, 'include', 'xcx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(xcx_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from xcorr_amd import _ffi
    lib = _ffi.lib()                       # raises loudly if not built
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in xcx.h but not exported'
    # the ctypes table binds exactly the header's surface
    assert sorted(_ffi.SIGNATURES) == syms


def test_no_torch_in_abi_or_library():
    src = open(os.path.join(ROOT, 'include', 'xcx.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)        # declarations only
    assert 'torch' not in src.lower() and 'at::' not in src and '#include <stddef.h>' in src
    import subprocess
    out = subprocess.run(['ldd', os.path.join(ROOT, 'xcorr_amd', 'lib', 'libxcx.so')],
                         capture_output=True, text=True).stdout
    assert 'torch' not in out and 'libamdhip64' in out and 'librccl' in out


def test_errors_are_reported_not_thrown():
    from xcorr_amd import _ffi
    lib = _ffi.lib()
    n = C.c_int(-1)
    rc = lib.xcx_device_count(C.byref(n))
    assert rc in (_ffi.XCX_OK, _ffi.XCX_ENODEV)
    if rc != 0:
        assert 'hipGetDeviceCount' in _ffi.last_error()
    assert lib.xcx_plan_destroy(None) == 0
    assert lib.xcx_destroy(None) == 0
    assert lib.xcx_plan_info(None, None) == _ffi.XCX_EINVAL
    assert 'null' in _ffi.last_error()


def test_product_path_has_no_cpu_fallback():
    """The package never imports the oracle, and a compute call without a GPU
    raises instead of silently computing on the host."""
    import subprocess
    import sys
    code = ('import sys; import xcorr_amd, xcorr_amd.util.cc, xcorr_amd.signal, '
            'xcorr_amd.core, xcorr_amd.ops, xcorr_amd.shard; '
            'assert not any(m.startswith("oracle") for m in sys.modules), "oracle imported"')
    subprocess.run([sys.executable, '-c', code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'xcorr_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt, f'{f} mentions the oracle'
    from xcorr_amd import _ffi
    n = C.c_int(0)
    if _ffi.lib().xcx_device_count(C.byref(n)) != 0 or n.value == 0:
        import numpy as np
        import xcorr_amd.util.cc as ucc
        with pytest.raises(_ffi.XcxError):
            ucc.cc(np.ones(8, np.float32), np.ones(8, np.float32))


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/xcx.h compiles as C99 with no C++ (or torch) in sight, and a
    plain-C caller links against libxcx.so."""
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / 'hdr.c'
    src.write_text('#include "xcx.h"\nint main(void) { return XCX_OK; }\n')
    subprocess.run([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(root, 'include'),
                    '-c', str(src), '-o', str(tmp_path / 'hdr.o')], check=True)
    lib = os.path.join(root, 'xcorr_amd', 'lib')
    subprocess.run([gcc, '-std=c99', '-Wall', '-O1', '-I', os.path.join(root, 'include'),
                    os.path.join(root, 'examples', 'abi_demo.c'), '-L', lib, '-lxcx', f'-Wl,-rpath,{lib}', '-lm',
                    '-o', str(tmp_path / 'abi_demo')], check=True)


def test_header_limits_are_what_the_library_enforces():
    """Every numeric limit the header quotes is a named constant, the library reports the values it was built with
    (xcx_limits, no GPU needed), and the prose refers to the names -- a binder reading include/xcx.h reads the
    contract of THIS build."""
    from xcorr_amd import _ffi
    src = open(os.path.join(ROOT, 'include', 'xcx.h')).read()
    macros = dict(re.findall(r'#define\s+(XCX_[A-Z0-9_]+)\s+(\d+)L\b', src))
    order = re.search(r'enum\s*\{\s*(XCX_LIM_MAX_SINGLE_TRANSFORM.*?)XCX_LIM_COUNT\s*\}', src, flags=re.S).group(1)
    names = [n.split('=')[0].strip() for n in order.replace('\n', ' ').split(',') if n.strip()]
    assert len(names) >= 12 and len(names) == len(macros)
    out = (C.c_long * len(names))()
    assert _ffi.lib().xcx_limits(out, len(names)) == 0
    for i, lim in enumerate(names):
        macro = 'XCX_' + lim[len('XCX_LIM_'):]
        assert macro in macros, f'{lim} has no #define {macro}'
        assert out[i] == int(macros[macro]), f'{macro}: header {macros[macro]}, library {out[i]}'
    # the entry-point comments quote limits by name, not by stale numbers
    prose = ' '.join(re.findall(r'/\*.*?\*/', src, flags=re.S))
    for stale in ('n <= 393216)', '3n <= 737280', 'square of n / 184320', 'F <= 256, L <= 4096', 'h <= 3072'):
        assert stale not in prose, f'stale limit text in include/xcx.h: {stale!r}'
    for macro in macros:
        assert prose.count(macro) >= 1, f'{macro} is defined but no entry point documents it'
    assert _ffi.lib().xcx_limits(None, 1) == _ffi.XCX_EINVAL


def test_reclaim_hook_is_part_of_the_context(monkeypatch):
    """freed device buffers wait in a pool of the Python binding; the library must be able to get them
    back (xcx_set_reclaim) -- the pool gives everything up when asked, oldest blocks first when it is full."""
    from xcorr_amd import engine
    calls = []

    class FakeLib:
        def xcx_free(self, h, p):
            calls.append(p.value)
            return 0

        def xcx_sync(self, h):
            return 0

    monkeypatch.setattr(engine._ffi, 'lib', lambda: FakeLib())
    ctx = engine.Context.__new__(engine.Context)
    ctx.handle = 1
    ctx._dev_pool, ctx._dev_pool_bytes = {}, 0
    ctx._dev_pool_cap, ctx._dev_pool_age = 8 << 20, 30.0
    assert not ctx._pool_give(0x1000, 100)                  # small blocks are not pooled
    assert not ctx._pool_give(0x1000, 5 << 20)              # nor blocks above half the cap
    assert ctx._pool_give(0x1000, 3 << 20) and ctx._pool_give(0x2000, 3 << 20)
    assert ctx._pool_give(0x3000, 4 << 20)                  # over the cap: the OLDEST block makes room
    assert calls == [0x1000] and ctx._dev_pool_bytes == 7 << 20
    assert ctx._pool_take(3 << 20) == 0x2000 and ctx._pool_take(3 << 20) is None
    assert ctx._reclaim() == 4 << 20 and calls == [0x1000, 0x3000] and ctx._dev_pool_bytes == 0
    ctx._pool_give(0x4000, 2 << 20)
    ctx._dev_pool[2 << 20][0] = (0x4000, -1e9)              # idle since long ago: expires at the next take / give
    assert ctx._pool_take(1 << 20) is None and calls[-1] == 0x4000 and not ctx._dev_pool
    ctx.handle = None                                       # no __del__ work on the fake


def test_every_pair_kernel_instantiation_is_reachable():
    """walk the planner (xcx_plan_geometry, no GPU) over every transform length it can pick --
    each instantiated dual / flow pair kernel is selected by some plan, and the list in csrc/Makefile (WFUSED) is
    the list the library carries.  Geometries without a dual kernel run the separate column / row launches."""
    from xcorr_amd import _ffi
    lib = _ffi.lib()
    cnt = C.c_int(0)
    buf = (C.c_int * 128)()
    assert lib.xcx_plan_fused_list(buf, 128, C.byref(cnt)) == 0
    fused = {(buf[2 * i], buf[2 * i + 1]) for i in range(cnt.value)}
    mk = open(os.path.join(ROOT, 'xcorr_amd', 'csrc', 'Makefile')).read()
    listed = {tuple(int(v) for v in f.split(':')) for f in re.search(r'^WFUSED = (.*)$', mk, flags=re.M).group(1).split()}
    assert fused == listed and len(fused) == cnt.value
    out = (C.c_int * 5)()
    chosen, lengths = set(), set()
    need = 1
    while True:                                   # every N the planner can return for nfft = 0, smallest first
        assert lib.xcx_plan_geometry(need, 0, out) == 0
        if out[0] < 0:
            break
        assert out[0] >= need and out[0] == 2 * out[1] * out[2]
        lengths.add(out[0])
        chosen.add((out[1], out[2]))
        assert bool(out[3]) == ((out[1], out[2]) in fused)
        # a forced nfft gives the same geometry as the automatic choice of that length
        forced = (C.c_int * 5)()
        assert lib.xcx_plan_geometry(1, out[0], forced) == 0 and list(forced) == list(out)
        need = out[0] + 1
    limits = (C.c_long * 1)()
    assert lib.xcx_limits(limits, 1) == 0 and max(lengths) == limits[0]       # XCX_MAX_SINGLE_TRANSFORM
    assert fused <= chosen, f'never selected: {sorted(fused - chosen)}'
    assert (1536, 120) in fused and out[0] == -1
    assert lib.xcx_plan_geometry(368640 - 100, 0, out) == 0 and list(out) == [368640, 1536, 120, 1, 1]   # the headline plan
