#!/usr/bin/env python
"""Phase timeline of the f16x3 layer kernel: builds a -DPWV_TRACE copy of the library in /tmp, runs one
launch and prints, per wave of workgroups 0/1, the s_memtime deltas between phase stamps.
stamps: 0 unit start | 1 P (+cond) loads issued | 2 all loads landed (trace build waits here) | 3 x[t-d] split
        4 GEMM1 pair0 (+x[t] split) | 5 GEMM1 pair1 (+gate pair0) | 6 next-unit loads issued + GEMM2 (+gate pair1)
        7 stores issued"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pwv_amd import _lib  # noqa: E402

so = '/tmp/libpwv_trace.so'
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-DPWV_TRACE'] + os.environ.get('PWV_TRACE_FLAGS', '').split() + [
       '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'parallel-wavenet-vocoder_amd', 'csrc'), '-o', so] + _lib.CSRC
subprocess.check_call(cmd)
_lib.LIB_PATH = so
lib = _lib.lib()
dev = torch.device('cuda', 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
G = 2
trace = torch.zeros(4096 + 4 * 1024, dtype=torch.int64, device=dev)
os.environ['PWV_TRACE_PTR'] = str(trace.data_ptr())
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nf = lib.pwv_layer_packed_floats(0, 0)
xs = [[torch.randn(rows, 64, device=dev) for _ in range(2)] for _ in range(G)]
packed = [torch.randn(nf, device=dev) * 0.05 for _ in range(G)]
proj = [torch.randn(128, device=dev) * 0.1 for _ in range(G)]
a = _lib.LayerArgs()
a.G, a.proj_row_stride, a.N, a.T, a.dilation, a.precision, a.skip_init = G, 128, 1, rows, 64, 1, 1
a.out_mode = _lib.OUT_RESIDUAL
for g in range(G):
    a.x_in[g], a.x_out[g] = xs[g][0].data_ptr(), xs[g][1].data_ptr()
    a.packed[g], a.proj[g] = packed[g].data_ptr(), proj[g].data_ptr()
for _ in range(3):
    trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.pwv_wavenet_layer_f32(ctypes.byref(a), s))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
full = trace.cpu().numpy()
rt = full[4096 + 1024:4096 + 1024 + 512].reshape(-1, 2)
rt = rt[rt[:, 0] > 0]
print('chip-wide clock (100 MHz): first WG start -> last WG end %.2f us; WG starts spread over %.2f us, ends over %.2f us; '
      'mean WG lifetime %.2f us' % ((rt[:, 1].max() - rt[:, 0].min()) / 100.0, (rt[:, 0].max() - rt[:, 0].min()) / 100.0,
                                     (rt[:, 1].max() - rt[:, 1].min()) / 100.0, (rt[:, 1] - rt[:, 0]).mean() / 100.0))
wgt = full[4096:4096 + 1024].reshape(-1, 4)
wgt = wgt[wgt[:, 0] > 0]
dur = wgt[:, 2] - wgt[:, 0]          # per-WG: entry -> last wave done (s_memtime is per-XCD: only differences within a WG mean anything)
fill = wgt[:, 1] - wgt[:, 0]
print('kernel event time %.1f us; per-WG duration ticks: min %d mean %d max %d; LDS fill mean %d max %d; %d WGs' % (
    ms * 1e3, dur.min(), dur.mean(), dur.max(), fill.mean(), fill.max(), len(dur)))
import numpy as np
print('   if ticks are 100 MHz-independent shader cycles: max WG duration = %.1f us at 2.2 GHz' % (dur.max() / 2.2e3))
# same-XCD spread: WG ids b, b+8, ... share an XCD (round-robin dispatch): compare their start times
for x in range(2):
    sel = wgt[x::8]
    print('   XCD %d: start spread %d ticks, end spread %d ticks, first start -> last end %d ticks' % (
        x, sel[:, 0].max() - sel[:, 0].min(), sel[:, 2].max() - sel[:, 2].min(), sel[:, 2].max() - sel[:, 0].min()))
# per-XCD view of the workgroup durations (WG b runs on XCD b % 8 under round-robin dispatch) and of the chip-wide end times
rt_all = full[4096 + 1024:4096 + 1024 + 2 * len(wgt)].reshape(-1, 2)
for x in range(8):
    d = dur[x::8]
    d = d[d > 0.5 * dur.mean()]          # (the last workgroups have no units)
    e = rt_all[x::8][:len(dur[x::8])]
    e = e[e[:, 0] > 0]
    print('   XCD %d: WG duration ticks mean %6d min %6d max %6d (%d WGs); chip-clock lifetime mean %.2f us, last end at +%.2f us' % (
        x, d.mean(), d.min(), d.max(), len(d), (e[:, 1] - e[:, 0]).mean() / 100.0, (e[:, 1].max() - rt[:, 0].min()) / 100.0))
t = full[:2048].reshape(2, 8, 8, 16)
for b in range(2):
    t0 = t[b][t[b] > 0].min()
    for w in range(8):
        print('wg %d wave %d' % (b, w))
        for u in range(8):
            st = t[b, w, u, :8]
            if st[0] == 0:
                continue
            d = [int(st[i + 1] - st[i]) for i in range(7)]
            print('   unit %d  start %7d | Pissue %5d wait %6d splitb %5d g1p0 %5d g1p1 %5d g2 %5d store %5d | total %6d' % (
                u, st[0] - t0, *d, st[7] - st[0]))
