"""ctypes binding of libpwv_hip.so (C ABI declared in include/pwv_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (or ``build_library()`` here)
with ``hipcc --offload-arch=gfx950``.  There is NO fallback: if the shared object is
missing or a call fails, a ``PwvError`` is raised -- the product path never routes
through a CPU implementation.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_int, c_int64, c_size_t, c_uint64, c_void_p

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_REPO_ROOT = os.path.dirname(_PKG_DIR)
LIB_PATH = os.environ.get('PWV_LIB') or os.path.join(_PKG_DIR, 'libpwv_hip.so')   # PWV_LIB: A/B another build
CSRC = [os.path.join(_PKG_DIR, 'csrc', f) for f in ('pwv_layer.hip', 'pwv_layer_f16.hip', 'pwv_layer_h16.hip', 'pwv_misc.hip',
                                                    'pwv_stack_persist.hip', 'pwv_norm.hip', 'pwv_audio.hip')]

HEADER_VERSION = 301          # PWV_HIP_VERSION of include/pwv_hip.h these ctypes mirrors were written against
FIRST_FOLD_FLOATS = 2048      # PWV_FIRST_FOLD_FLOATS
PWV_MAX_NETS = 2
PREC_F32, PREC_F16X3, PREC_F16 = 0, 1, 2
OUT_RESIDUAL, OUT_GATED = 0, 1
HEAD_IN_GATED, HEAD_IN_SKIPSUM = 0, 1

# every symbol include/pwv_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTED_SYMBOLS = (
    'pwv_last_error', 'pwv_version', 'pwv_device_cus', 'pwv_causal_conv_f32', 'pwv_linear_f32',
    'pwv_upsample_repeat_f32', 'pwv_crop_time_f32', 'pwv_logistic_noise_f32', 'pwv_logistic_noise_stream_f32', 'pwv_iaf_front_f32',
    'pwv_layer_packed_floats', 'pwv_pack_layer_f32', 'pwv_proj_column_map', 'pwv_wavenet_layer_f32',
    'pwv_head_packed_floats', 'pwv_pack_head_f32', 'pwv_wavenet_head_f32', 'pwv_wavenet_stack_f32',
    'pwv_iaf_front_f16', 'pwv_cond_to_f16', 'pwv_tile32_floats', 'pwv_rows_to_tile32_f32', 'pwv_tile32_to_rows_f32',
    'pwv_linear_split_f32', 'pwv_cond_project_f32', 'pwv_pack_first_fold_f16x3', 'pwv_pack_first_fold_f32', 'pwv_cond_split_f16', 'pwv_range_flag', 'pwv_status_words_alloc', 'pwv_status_words_free', 'pwv_range_check_f32', 'pwv_range_stats_f32',
    'pwv_persist_workspace_bytes', 'pwv_persist_short_input', 'pwv_persist_status', 'pwv_wavenet_stack_persist_f32',
    'pwv_wav_to_mel_db_f32', 'pwv_pack_proj_f32', 'pwv_instance_norm_workspace_bytes', 'pwv_instance_norm_f32', 'pwv_channel_affine_f32', 'pwv_add_f32', 'pwv_gate_f32',
)


class PwvError(RuntimeError):
    pass


class PwvRangeError(PwvError):
    """A split-fp16 ('f16x3') forward met an operand beyond fp16's exponent range (include/pwv_hip.h, range guard):
    its result is not trustworthy; rerun with precision='f32'."""


class PwvPersistError(PwvError):
    """A persistent stack launch gave up (csrc/pwv_stack_persist.hip: a bounded poll ran out, e.g. because another process
    held CUs and its workgroups were not all resident): its outputs are invalid.  The per-layer path is used from then on;
    rerun the forward."""


class LayerArgs(Structure):
    _fields_ = [
        ('G', c_int),
        ('x_in', c_void_p * PWV_MAX_NETS),
        ('x_out', c_void_p * PWV_MAX_NETS),
        ('packed', c_void_p * PWV_MAX_NETS),
        ('proj', c_void_p * PWV_MAX_NETS),
        ('proj_row_stride', c_int),
        ('cond', c_void_p),
        ('cond_channels', c_int),
        ('skip', c_void_p * PWV_MAX_NETS),
        ('skip_init', c_int),
        ('N', c_int), ('T', c_int), ('dilation', c_int),
        ('cond_hop', c_int), ('cond_offset', c_int), ('cond_frames', c_int),
        ('out_mode', c_int),
        ('precision', c_int),
        ('max_workgroups', c_int),
        ('x_first', c_void_p),
        ('causal_filter', c_void_p * PWV_MAX_NETS),
        ('first_fold', c_void_p * PWV_MAX_NETS),
        ('head_packed', c_void_p * PWV_MAX_NETS),
        ('head_out', c_void_p * PWV_MAX_NETS),
        ('head_q', c_int),
        ('x_limit', ctypes.c_float),
        ('range_flag', c_void_p),
    ]


class HeadArgs(Structure):
    _fields_ = [
        ('G', c_int),
        ('in_', c_void_p * PWV_MAX_NETS),
        ('packed', c_void_p * PWV_MAX_NETS),
        ('out', c_void_p * PWV_MAX_NETS),
        ('N', c_int), ('T', c_int), ('Q', c_int),
        ('in_mode', c_int),
        ('precision', c_int),
        ('max_workgroups', c_int),
    ]


class StackArgs(Structure):
    _fields_ = [
        ('G', c_int),
        ('n_layers', c_int),
        ('dilations', POINTER(c_int)),
        ('buf0', c_void_p * PWV_MAX_NETS),
        ('buf1', c_void_p * PWV_MAX_NETS),
        ('packed_layers', c_void_p * PWV_MAX_NETS),
        ('packed_layer_stride', c_size_t),
        ('proj', c_void_p * PWV_MAX_NETS),
        ('proj_row_stride', c_int),
        ('cond', c_void_p),
        ('cond_channels', c_int),
        ('skip', c_void_p * PWV_MAX_NETS),
        ('packed_head', c_void_p * PWV_MAX_NETS),
        ('out', c_void_p * PWV_MAX_NETS),
        ('Q', c_int),
        ('N', c_int), ('T', c_int),
        ('cond_hop', c_int), ('cond_offset', c_int), ('cond_frames', c_int),
        ('precision', c_int),
        ('max_workgroups', c_int),
        ('ev_begin', c_void_p * PWV_MAX_NETS),
        ('ev_end', c_void_p * PWV_MAX_NETS),
        ('x_first', c_void_p),
        ('causal_filter', c_void_p * PWV_MAX_NETS),
        ('separate_head', c_int),
        ('x_limit', ctypes.c_float),
        ('range_flag', c_void_p),
        ('first_fold', c_void_p * PWV_MAX_NETS),
    ]


class PersistArgs(Structure):
    _fields_ = [
        ('struct_size', c_size_t),      # set by __init__
        ('G', c_int),
        ('n_layers', c_int),
        ('dilations', POINTER(c_int)),
        ('x_ring', c_void_p * PWV_MAX_NETS),
        ('ring_stride', c_size_t),
        ('ring_rotation', c_int),
        ('packed_layers', c_void_p * PWV_MAX_NETS),
        ('packed_layer_stride', c_size_t),
        ('proj', c_void_p * PWV_MAX_NETS),
        ('proj_row_stride', c_int),
        ('N', c_int), ('T', c_int),
        ('cond_hop', c_int), ('cond_offset', c_int), ('cond_frames', c_int),
        ('workspace', c_void_p),
        ('workspace_bytes', c_size_t),
        ('workspace_clean', c_int),
        ('precision', c_int),
        ('max_workgroups', c_int),
        ('min_units_per_workgroup', c_int),
        ('x_first', c_void_p),
        ('causal_filter', c_void_p * PWV_MAX_NETS),
        ('x_limit', ctypes.c_float),
        ('range_flag', c_void_p),
        ('first_fold', c_void_p * PWV_MAX_NETS),
        ('status', c_void_p),
        ('tail_layer', c_void_p * PWV_MAX_NETS),
        ('tail_head', c_void_p * PWV_MAX_NETS),
        ('tail_out', c_void_p * PWV_MAX_NETS),
        ('tail_q', c_int),
        ('tail_dilation', c_int),
        ('affine_x', c_void_p),
        ('affine_out', c_void_p),
    ]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(PersistArgs)



# per-source extra flags (none in the product; tools/probes/regw/README.md: the register-stationary probe kernel needs
# `-mllvm -amdgpu-mfma-vgpr-form=1`, which is why the sources are compiled one by one)
EXTRA_FLAGS = {}
# extra sources for probe builds: PWV_EXTRA_SRC="path[:flag,flag...] ..."
for _e in os.environ.get('PWV_EXTRA_SRC', '').split():
    _src, _, _fl = _e.partition(':')
    CSRC.append(os.path.abspath(_src))
    EXTRA_FLAGS[os.path.basename(_src)] = [f for f in _fl.split(',') if f]


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into the in-tree shared library (one object per source, compiled in parallel and
    reused while the source and the headers are older than it)."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(_PKG_DIR, 'csrc', h) for h in ('pwv_common.h', 'pwv_layer_common.h', 'pwv_f16x3.h')] + [os.path.join(_REPO_ROOT, 'include', 'pwv_hip.h')]
    if os.environ.get('PWV_LIB'):
        return LIB_PATH            # an explicitly chosen library is never rebuilt
    if not force and os.path.exists(LIB_PATH):
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in CSRC + hdrs):
            return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(_PKG_DIR, 'csrc', '_obj')
    os.makedirs(objdir, exist_ok=True)
    # ONE gfx950 code object for the XNACK mode an MI355X runs in by default (xnack-): code built for a known mode instead of
    # 'either' is 0.5 % faster on the bench step; xnack+ objects (XNACK-on runs) are not available on the GPU pool
    base = [hipcc, '--offload-arch=gfx950:xnack-', '-O3', '-std=c++17', '-fPIC',
            '-I' + os.path.join(_REPO_ROOT, 'include'), '-I' + os.path.join(_PKG_DIR, 'csrc')]
    extra = os.environ.get('PWV_CXXFLAGS', '').split()
    hdr_time = max(os.path.getmtime(h) for h in hdrs + [os.path.abspath(__file__)])

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ('.' + '_'.join(extra).replace('/', '_') if extra else '') + '.o')
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_time):
            return obj, None
        cmd = base + EXTRA_FLAGS.get(os.path.basename(src), []) + extra + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return obj, (res.stdout if res.returncode != 0 else None)

    with ThreadPoolExecutor(max_workers=min(len(CSRC), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, CSRC))
    for obj, err in results:
        if err is not None:
            raise PwvError('hipcc failed:\n' + err)
    cmd = [hipcc, '--offload-arch=gfx950:xnack-', '-shared', '-fPIC', '-o', LIB_PATH] + [o for o, _ in results]
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise PwvError('hipcc (link) failed:\n' + res.stdout)
    return LIB_PATH


_lib = None


def _declare(lib):
    f32p = c_void_p
    lib.pwv_last_error.restype = c_char_p
    lib.pwv_last_error.argtypes = []
    lib.pwv_version.restype = c_int
    lib.pwv_device_cus.restype = c_int
    lib.pwv_causal_conv_f32.argtypes = [f32p, f32p, f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.pwv_linear_f32.argtypes = [f32p, f32p, f32p, f32p, c_int, c_int, c_int, c_int, c_void_p]
    lib.pwv_linear_split_f32.argtypes = lib.pwv_linear_f32.argtypes
    lib.pwv_upsample_repeat_f32.argtypes = [f32p, f32p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.pwv_crop_time_f32.argtypes = [f32p, f32p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.pwv_logistic_noise_f32.argtypes = [f32p, c_int64, c_uint64, c_uint64, c_void_p]
    lib.pwv_iaf_front_f32.argtypes = [f32p, f32p, f32p, c_int, f32p, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                      c_int, c_int, c_int, c_int, c_void_p]
    lib.pwv_iaf_front_f16.argtypes = lib.pwv_iaf_front_f32.argtypes
    lib.pwv_cond_to_f16.argtypes = [f32p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.pwv_cond_split_f16.argtypes = lib.pwv_cond_to_f16.argtypes
    lib.pwv_tile32_floats.restype = c_size_t
    lib.pwv_tile32_floats.argtypes = [c_int64, c_int]
    lib.pwv_rows_to_tile32_f32.argtypes = [f32p, f32p, c_int64, c_int, c_void_p]
    lib.pwv_tile32_to_rows_f32.argtypes = [f32p, f32p, c_int64, c_int, c_void_p]
    lib.pwv_layer_packed_floats.restype = c_size_t
    lib.pwv_layer_packed_floats.argtypes = [c_int, c_int]
    lib.pwv_pack_layer_f32.argtypes = [f32p] * 8 + [c_int, c_int, c_int, f32p, c_void_p]
    lib.pwv_proj_column_map.argtypes = [POINTER(c_int)]
    lib.pwv_wavenet_layer_f32.argtypes = [POINTER(LayerArgs), c_void_p]
    lib.pwv_head_packed_floats.restype = c_size_t
    lib.pwv_head_packed_floats.argtypes = [c_int]
    lib.pwv_pack_head_f32.argtypes = [f32p] * 6 + [c_int, c_int, f32p, c_void_p]
    lib.pwv_pack_first_fold_f16x3.argtypes = [f32p, f32p, f32p, f32p, c_void_p]
    lib.pwv_pack_first_fold_f32.argtypes = [f32p, f32p, f32p, f32p, c_void_p]
    lib.pwv_wavenet_head_f32.argtypes = [POINTER(HeadArgs), c_void_p]
    lib.pwv_wavenet_stack_f32.argtypes = [POINTER(StackArgs), POINTER(c_void_p)]
    lib.pwv_pack_proj_f32.argtypes = [f32p, f32p, f32p, f32p, c_int, c_int, c_int, f32p, f32p, c_void_p]
    lib.pwv_wav_to_mel_db_f32.argtypes = [f32p, f32p, f32p, f32p, c_int, c_int, c_int, c_int, c_int] + [ctypes.c_float] * 4 + [c_int, c_void_p]
    lib.pwv_instance_norm_workspace_bytes.restype = c_size_t
    lib.pwv_instance_norm_workspace_bytes.argtypes = [c_int, c_int, c_int]
    lib.pwv_instance_norm_f32.argtypes = [f32p, f32p, c_int, c_int, c_int, f32p, f32p, ctypes.c_float, c_void_p, c_size_t, c_void_p]
    lib.pwv_channel_affine_f32.argtypes = [f32p, f32p, c_int64, c_int, f32p, f32p, c_int, c_int, c_void_p]
    lib.pwv_add_f32.argtypes = [f32p, f32p, f32p, c_int64, c_void_p]
    lib.pwv_gate_f32.argtypes = [f32p, f32p, f32p, c_int64, c_void_p]
    lib.pwv_persist_workspace_bytes.restype = c_size_t
    lib.pwv_persist_workspace_bytes.argtypes = [POINTER(PersistArgs)]
    lib.pwv_persist_short_input.restype = c_int
    lib.pwv_persist_short_input.argtypes = [POINTER(PersistArgs)]
    lib.pwv_persist_status.argtypes = [POINTER(c_void_p)]
    lib.pwv_wavenet_stack_persist_f32.argtypes = [POINTER(PersistArgs), c_void_p]
    lib.pwv_range_stats_f32.argtypes = [f32p] * 8 + [c_int, f32p, c_void_p]
    lib.pwv_range_flag.argtypes = [POINTER(c_void_p)]
    lib.pwv_logistic_noise_stream_f32.argtypes = [f32p, c_int64, c_void_p, c_void_p]
    lib.pwv_cond_project_f32.argtypes = [f32p, f32p, c_int, f32p, f32p, f32p, f32p, c_int, c_int, c_int, ctypes.c_float, c_void_p, c_void_p]
    lib.pwv_status_words_alloc.argtypes = [POINTER(c_void_p)]
    lib.pwv_status_words_free.argtypes = [c_void_p]
    lib.pwv_range_check_f32.argtypes = [f32p, c_int64, ctypes.c_float, c_void_p, c_void_p]
    for name in EXPORTED_SYMBOLS:      # fails loudly (AttributeError) if a symbol is missing
        getattr(lib, name)
    return lib


def lib():
    """Load libpwv_hip.so (once).  Raises PwvError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PwvError('libpwv_hip.so not built at %s -- run `python -c "import __graft_entry__ as g; '
                           'g.build()"` (hipcc, gfx950). There is no CPU fallback.' % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  Load torch's first so our NEEDED entry binds to the runtime that owns the
        # device context, streams and allocations we are handed.
        import torch
        hip_rt = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
        try:
            loaded = _declare(ctypes.CDLL(LIB_PATH))
        except OSError as e:
            raise PwvError('cannot load %s: %s' % (LIB_PATH, e))
        # a major version step changes an argument struct's layout (include/pwv_hip.h): never call across one
        if loaded.pwv_version() // 100 != HEADER_VERSION // 100:
            raise PwvError('%s is version %d, this binding was written against %d: rebuild it (`python -c "import __graft_entry__ as g; g.build()"`)'
                           % (LIB_PATH, loaded.pwv_version(), HEADER_VERSION))
        _lib = loaded
    return _lib


def check(code: int, what: str = ''):
    if code != 0:
        msg = lib().pwv_last_error()
        raise PwvError('%s failed (%d): %s' % (what or 'libpwv_hip call', code, msg.decode() if msg else '?'))


def proj_column_map():
    arr = (c_int * 128)()
    check(lib().pwv_proj_column_map(arr), 'pwv_proj_column_map')
    return list(arr)
