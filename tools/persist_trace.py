"""GPU: per-wave cycle accounting of the persistent stack kernel (needs the -DPWV_PTRACE build:
   hipcc ... -DPWV_PTRACE -o /tmp/libpwv_ptrace.so ; PWV_LIB=/tmp/libpwv_ptrace.so python tools/persist_trace.py [rows] [layers] [G])"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pwv_amd import engine
from pwv_amd.modules import WaveNet
from pwv_amd.variables import VariableStore


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    dev = torch.device('cuda', 0)
    d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    dil = (d10 * 3)[:L]
    store = VariableStore(device=dev, seed=3)
    kw = dict(batch_size=1, dilations=dil, filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    hop = 80
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, rows, 1), generator=g).to(dev)
    frames = torch.rand((1, rows // hop + 1, 80), generator=g).to(dev)
    cond = engine.RepeatedCondition(frames, hop, hop // 2, rows)
    trace = torch.zeros((2 * 256 * 8, 32), dtype=torch.int64, device=dev)
    os.environ['PWV_PTRACE_PTR'] = str(trace.data_ptr())
    engine.PERSIST = True
    engine.PERSIST_UNITS_PER_WAVE = float(os.environ.get('UPW', '2'))
    for _ in range(3):
        engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    trace.zero_()
    engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    raw = trace.cpu().numpy()
    diag = raw[2048:]
    bad = diag[diag[:, 0] != 0]
    if len(bad):
        print('GIVE-UPS: %d waves; by code %s' % (len(bad), {int(c): int((bad[:, 0] == c).sum()) for c in np.unique(bad[:, 0])}))
        for c in np.unique(bad[:, 0]):
            b = bad[bad[:, 0] == c]
            pairs = {}
            for r in b:
                pairs[(int(r[1]), int(r[2]))] = pairs.get((int(r[1]), int(r[2])), 0) + 1
            print('  code %d: (need, value) -> count: %s' % (c, sorted(pairs.items())[:12]))
        st = raw[:2048]
        st = st[(st[:, 6] > 0) | (st[:, 30] != 0)]
        print('  final (layer, unit) of dead waves: %s' % sorted(set((int(r[31] >> 32), int(r[31] & 0xffffffff)) for r in st if r[30]))[:40])
    t = raw[:2048].astype(np.float64)
    t = t[t[:, 6] > 0]
    names = ['loop total', 'TOP wait vmcnt(0)', 'RAW spins', 'WAR spins', 'leave_layer', 'weight-ready spins']
    print('%d waves, units per wave: mean %.1f (min %d, max %d); status %d' % (len(t), t[:, 6].mean(), t[:, 6].min(), t[:, 6].max(), engine.persist_status()))
    tot = t[:, 0]
    print('loop cycles per wave: mean %.0f, min %.0f, max %.0f; per unit %.0f' % (tot.mean(), tot.min(), tot.max(), (tot / t[:, 6]).mean()))
    for k in range(1, 6):
        print('  %-22s %6.1f %% of the loop (mean %.0f cycles per unit, max wave %.1f %%)'
              % (names[k], 100 * t[:, k].sum() / tot.sum(), (t[:, k] / t[:, 6]).mean(), 100 * (t[:, k] / tot).max()))
    ph = ['TOP (P + flag loads, wait, publish)', 'split x[t-d]', 'GEMM1 pair 0 (48 MFMA)', 'GEMM1 pair 1 (48 MFMA) + gate', 'acc2 init + prefetch issue',
          'GEMM2 (24 MFMA) + gate', 'WAR check + stores', 'leave_layer / bookkeeping']
    for k, name in enumerate(ph):
        print('  phase %-38s %6.0f cycles per unit (%4.1f %%)' % (name, (t[:, 16 + k] / t[:, 6]).mean(), 100 * t[:, 16 + k].sum() / tot.sum()))
    for k, name in enumerate(['unit_rows + P loads issued', 'claim read, locate, deps, flag loads issued', 'wait vmcnt(0)', 'publish, claim, deferred loads', 'flag evaluation']):
        print('    TOP part %-46s %6.0f cycles per unit' % (name, (t[:, 24 + k] / t[:, 6]).mean()))
    print('  units that had to spin on RAW: %.2f %%' % (100 * t[:, 7].sum() / t[:, 6].sum()))
    # chip-wide 100 MHz clock: absolute picture in microseconds from the first wave's entry
    t0 = t[:, 13].min()
    ent, st, en = (t[:, 13] - t0) / 100.0, (t[:, 14] - t0) / 100.0, (t[:, 15] - t0) / 100.0
    print('kernel entry: %.1f .. %.1f us; loop start: %.1f .. %.1f us; loop end: %.1f .. %.1f us' % (ent.min(), ent.max(), st.min(), st.max(), en.min(), en.max()))
    clk = tot / ((t[:, 15] - t[:, 14]) / 100.0) / 1e3
    print('shader clock during the loop (s_memtime cycles / s_memrealtime): mean %.3f GHz (min %.3f, max %.3f)' % (clk.mean(), clk.min(), clk.max()))
    for xcc in range(8):
        m = t[:, 11] == xcc
        if m.any():
            print('  XCD %d: %d waves, loop start %.1f..%.1f us, end %.1f..%.1f us, %.0f cycles per unit'
                  % (xcc, m.sum(), st[m].min(), st[m].max(), en[m].min(), en[m].max(), (tot[m] / t[m, 6]).mean()))


if __name__ == '__main__':
    main()
