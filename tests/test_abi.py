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
