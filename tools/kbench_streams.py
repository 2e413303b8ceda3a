#!/usr/bin/env python
"""Experiment: one G=2 launch per layer on one stream vs G=1 launches on two streams (one per net)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwv_amd import _lib  # noqa: E402
from pwv_amd._lib import LayerArgs, check  # noqa: E402

_lib.build_library()
lib = _lib.lib()
dev = torch.device('cuda', 0)
rows, layers, prec = int(os.environ.get('ROWS', 160000)), 30, 1
nf = lib.pwv_layer_packed_floats(0, 0)
xs = [[torch.randn(rows, 64, device=dev) for _ in range(2)] for _ in range(2)]
packed = [torch.randn(nf, device=dev) * 0.05 for _ in range(2)]
proj = [torch.randn(128, device=dev) * 0.1 for _ in range(2)]
dil = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512] * 3


def args_for(nets, j, wgs):
    a = LayerArgs()
    a.G = len(nets)
    a.proj_row_stride, a.N, a.T, a.dilation, a.precision, a.skip_init = 128, 1, rows, dil[j], prec, 1
    a.out_mode = _lib.OUT_RESIDUAL
    a.max_workgroups = wgs
    for i, g in enumerate(nets):
        a.x_in[i], a.x_out[i] = xs[g][j & 1].data_ptr(), xs[g][(j & 1) ^ 1].data_ptr()
        a.packed[i], a.proj[i] = packed[g].data_ptr(), proj[g].data_ptr()
    return a


def run_single(wgs=0):
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for j in range(layers):
        a = args_for([0, 1], j, wgs)
        check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), s))


streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run_two(wgs=0):
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    for j in range(layers):
        for g in range(2):
            a = args_for([g], j, wgs)
            check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), ctypes.c_void_p(streams[g].cuda_stream)))
    for st in streams:
        cur.wait_stream(st)


def timeit(fn, *a):
    for _ in range(2):
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / layers * 1e3


print('one stream, G=2 per launch           : %.1f us per layer (both nets)' % timeit(run_single))
for w in (96, 112, 128, 144, 160, 192):
    print('two streams, G=1, %3d WGs per launch  : %.1f us per layer (both nets)' % (w, timeit(run_two, w)))
