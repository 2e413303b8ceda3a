#!/usr/bin/env python
"""Micro-benchmark of the fused layer / head kernels alone (random packed weights; timing only).
   python tools/kbench.py [--rows 160000] [--iters 50] [--wgs 0]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwv_amd import _lib  # noqa: E402
from pwv_amd._lib import HeadArgs, LayerArgs, check  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=160000)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--wgs', type=int, default=0)
    ap.add_argument('--dilation', type=int, default=64)
    ap.add_argument('--G', type=int, default=2)
    ap.add_argument('--zero', action='store_true', help='all-zero activations and weights (power / DVFS probe)')
    ap.add_argument('--cold-weights', type=int, default=1, help='rotate through this many distinct packed weight buffers')
    ap.add_argument('--precision', type=int, default=0, help='0 = f32, 1 = f16x3, 2 = f16')
    args = ap.parse_args()
    _lib.build_library()
    lib = _lib.lib()
    dev = torch.device('cuda', 0)
    G, rows = args.G, args.rows
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nf = lib.pwv_layer_packed_floats(0, 0)
    scale = 0.0 if args.zero else 1.0
    xs = [[torch.randn(rows, 64, device=dev) * scale for _ in range(2)] for _ in range(G)]
    if args.precision == 2:      # fp16 rows
        xs = [[x.half() for x in pair] for pair in xs]
    packed = [torch.randn(nf, device=dev) * 0.05 * scale for _ in range(G)]
    rot = [[torch.randn(nf, device=dev) * 0.05 * scale for _ in range(G)] for _ in range(args.cold_weights)]
    proj = [torch.randn(128, device=dev) * 0.1 * scale for _ in range(G)]
    a = LayerArgs()
    a.G = G
    a.proj_row_stride = 128
    a.N, a.T, a.dilation = 1, rows, args.dilation
    a.precision = args.precision
    a.max_workgroups = args.wgs
    a.skip_init = 1
    for mode, name in ((_lib.OUT_RESIDUAL, 'layer_residual'), (_lib.OUT_GATED, 'layer_gated')):
        a.out_mode = mode
        for g in range(G):
            a.x_in[g] = xs[g][0].data_ptr()
            a.x_out[g] = xs[g][1].data_ptr()
            a.packed[g] = packed[g].data_ptr()
            a.proj[g] = proj[g].data_ptr()
        for _ in range(5):
            check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for it in range(args.iters):
            if args.cold_weights > 1:
                for g in range(G):
                    a.packed[g] = rot[it % args.cold_weights][g].data_ptr()
            check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), s))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.iters * 1e3
        flop = rows * G * (2 * (2 * 64 * 128 + (64 * 64 if mode == _lib.OUT_RESIDUAL else 0)))
        print('%-15s %8.2f us  %7.2f TFLOP/s  %7.1f GB/s(alg)' % (name, us, flop / us / 1e6, rows * G * 512 / us / 1e3))
    # head
    hf = lib.pwv_head_packed_floats(1)
    hp_ = [torch.randn(hf, device=dev) * 0.05 for _ in range(G)]
    outs = [torch.empty(rows, 1, device=dev) for _ in range(G)]
    h = HeadArgs()
    h.G = G
    h.N, h.T, h.Q = 1, rows, 1
    h.in_mode = _lib.HEAD_IN_GATED
    h.precision = args.precision
    h.max_workgroups = args.wgs
    for g in range(G):
        h.in_[g] = xs[g][0].data_ptr()
        h.packed[g] = hp_[g].data_ptr()
        h.out[g] = outs[g].data_ptr()
    for _ in range(5):
        check(lib.pwv_wavenet_head_f32(ctypes.byref(h), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(args.iters):
        check(lib.pwv_wavenet_head_f32(ctypes.byref(h), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    flop = rows * G * 2 * (64 * 128 + 128 * 128 + 128)
    print('%-15s %8.2f us  %7.2f TFLOP/s' % ('head', us, flop / us / 1e6))


if __name__ == '__main__':
    main()
