"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/pwv_hip.h
declares; argument validation works without a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'pwv_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pwv_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported(built_lib):
    from pwv_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 16
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared          # the binding list mirrors the header
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_packed_sizes_and_column_map(built_lib):
    from pwv_amd import _lib
    lib = built_lib
    assert lib.pwv_version() == _lib.HEADER_VERSION == 301      # PWV_HIP_VERSION: 3xx = pwv_persist_args begins with struct_size
    base = lib.pwv_layer_packed_floats(0, 0)
    assert base == 4 * 16 * 64 * 4 + 2 * 8 * 64 * 4 + 64               # filter|gate + dense + dense bias
    assert lib.pwv_layer_packed_floats(1, 0) == base + 8192 + 128     # + skip + skip bias
    assert lib.pwv_layer_packed_floats(0, 80) == base + 10240         # + per-sample cond
    assert lib.pwv_head_packed_floats(1) == lib.pwv_head_packed_floats(4) == 8192 + 128 + 16384 + 128 + 2 * 4 * 64 + 4
    cmap = _lib.proj_column_map()
    assert sorted(cmap) == list(range(128))                           # a permutation of F|G channels
    # lane half h owns the 16-byte chunks at float offsets 8g + 4h (v_mfma 32x32 C/D layout)
    assert cmap[:8] == [0, 1, 2, 3, 8, 9, 10, 11] and cmap[64:68] == [4, 5, 6, 7]


def test_argument_validation_no_gpu(built_lib):
    from pwv_amd import _lib
    lib = built_lib
    assert lib.pwv_causal_conv_f32(None, None, None, 1, 8, 4, 4, 2, 1, None) == -1
    assert b'NULL' in lib.pwv_last_error()
    assert lib.pwv_linear_f32(1, 1, None, 1, 4, 7, 8, 0, None) == -1          # K not a multiple of 8
    assert b'multiple of 8' in lib.pwv_last_error()
    a = _lib.LayerArgs()
    a.G = 3
    assert lib.pwv_wavenet_layer_f32(ctypes.byref(a), None) == -1
    assert b'out of range' in lib.pwv_last_error()
    h = _lib.HeadArgs()
    h.G, h.N, h.T, h.Q = 1, 1, 8, 9
    assert lib.pwv_wavenet_head_f32(ctypes.byref(h), None) == -1
    assert lib.pwv_pack_layer_f32(1, 1, 1, None, None, None, None, None, 0, 40, 0, 1, None) == -1   # cond C != 80
    with pytest.raises(_lib.PwvError):
        _lib.check(-1, 'x')
    # pwv_persist_args carries its own size (PWV_HIP_VERSION 3xx): a struct that does not say how long it is is refused before
    # anything in it is used; the ctypes mirror fills the field in by itself
    pa = _lib.PersistArgs()
    assert pa.struct_size == ctypes.sizeof(_lib.PersistArgs)
    pa.struct_size = 0
    assert lib.pwv_persist_workspace_bytes(ctypes.byref(pa)) == 0 and b'struct_size' in lib.pwv_last_error()
    pa.struct_size = 16
    assert lib.pwv_persist_workspace_bytes(ctypes.byref(pa)) == 0 and b'struct_size' in lib.pwv_last_error()
    assert lib.pwv_wavenet_stack_persist_f32(ctypes.byref(pa), None) == -1 and b'struct_size' in lib.pwv_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No silent fallback: a missing libpwv_hip.so is a hard error."""
    from pwv_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.PwvError, match='no CPU fallback'):
        _lib.lib()


def test_header_is_plain_c():
    """include/pwv_hip.h is the drop-in boundary: it must compile as C99 (no C++ / torch / HIP types in it), and the
    ctypes mirrors in _lib.py must have the sizes the C compiler gives the structs."""
    import ctypes
    import os
    import re
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stdio.h>\n#include "pwv_hip.h"\nint main(void){ printf("%zu %zu %zu\\n", sizeof(pwv_layer_args), '
           'sizeof(pwv_head_args), sizeof(pwv_stack_args)); printf("%zu\\n", sizeof(pwv_persist_args)); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        c, exe = os.path.join(d, 't.c'), os.path.join(d, 't')
        with open(c, 'w') as f:
            f.write(src)
        subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I' + os.path.join(root, 'include'), c, '-o', exe])
        sizes = [int(x) for x in re.findall(r'\d+', subprocess.check_output([exe]).decode())]
    from pwv_amd import _lib
    assert sizes == [ctypes.sizeof(_lib.LayerArgs), ctypes.sizeof(_lib.HeadArgs), ctypes.sizeof(_lib.StackArgs), ctypes.sizeof(_lib.PersistArgs)]


def test_plain_c_client_builds(tmp_path):
    """examples/c_abi_smoke.c: the boundary is usable from C99 with nothing but the header, the library and the HIP
    runtime (it is RUN on the GPU by tests/test_gpu_parity.py::test_plain_c_client_runs)."""
    from pwv_amd import _lib
    from tests.util import build_c_abi_smoke
    _lib.build_library()
    res = build_c_abi_smoke(str(tmp_path / 'c_abi_smoke'))
    assert res.returncode == 0, res.stdout
