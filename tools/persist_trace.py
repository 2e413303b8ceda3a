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
    trace = torch.zeros((2 * 256 * 8, 24), dtype=torch.int64, device=dev)
    os.environ['PWV_PTRACE_PTR'] = str(trace.data_ptr())
    engine.PERSIST = True
    for _ in range(3):
        engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    trace.zero_()
    engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    t = trace.cpu().numpy().astype(np.float64)
    t = t[t[:, 5] > 0]
    # per wave: [0] loop cycles, [1] drain at the top, [2] RAW spins (incl. the publish in front), [3] WAR spins, [4] settle,
    # [5] units, [6] units whose rows were not prefetched, [7]/[8] first task at / last task done at (s_memtime), [12] gave up
    print('%d waves, units per wave: mean %.1f (min %d, max %d); status %d; waves that gave up: %d'
          % (len(t), t[:, 5].mean(), t[:, 5].min(), t[:, 5].max(), engine.persist_status(), int(t[:, 18].sum())))
    tot = t[:, 0]
    print('loop cycles per wave: mean %.0f, min %.0f, max %.0f; per unit %.0f (steady: two waves share a SIMD)'
          % (tot.mean(), tot.min(), tot.max(), (tot / t[:, 5]).mean()))
    for k, name in ((1, 'drain vmcnt(0) at the top'), (2, 'not-prefetched path (publish + RAW spins)'), (3, 'WAR spins'), (4, 'settle (publish / leave / refill issue)')):
        print('  %-44s %6.2f %% of the loop (mean %.0f cycles per unit, worst wave %.1f %%)'
              % (name, 100 * t[:, k].sum() / tot.sum(), (t[:, k] / t[:, 5]).mean(), 100 * (t[:, k] / tot).max()))
    print('  units whose rows were NOT prefetched: %.2f %%' % (100 * t[:, 6].sum() / t[:, 5].sum()))
    # round 5: where a unit of the general loop spends its time (cycles per unit, means over the waves; [1] [2] [3] are inside [9] / [12])
    for k, name in ((9, 'top: P row requested, next task located, look-back row split'), (1, 'drain vmcnt(0): the rows, the P row, the previous stores'),
                    (10, 'GEMM1 (filter|gate, K = 128; gate of pair 0)'), (11, 'GEMM2 (dense; gate of pair 1; next rows requested)'),
                    (12, 'stores + moving on (incl. the waits for the next task)')):
        print('  phase %-68s %7.0f cycles per unit' % (name, (t[:, k] / t[:, 5]).mean()))
    t0 = t[:, 20].min()
    st, en = (t[:, 20] - t0) / 100.0, (t[:, 19] - t0) / 100.0
    print('loop start %.1f .. %.1f us, loop end %.1f .. %.1f us (chip-wide 100 MHz clock)' % (st.min(), st.max(), en.min(), en.max()))
    clk = tot / ((t[:, 19] - t[:, 20]) / 100.0) / 1e3
    print('shader clock during the loop (s_memtime cycles / s_memrealtime): mean %.3f GHz (min %.3f, max %.3f)' % (clk.mean(), clk.min(), clk.max()))
    # workgroups of one XCD take consecutive ranges (16 per XCD and net at 256 CUs): is the spread systematic per XCD?
    for g in sorted(set(int(r[17]) // 16 for r in t)):
        m = np.array([int(r[17]) // 16 == g for r in t])
        print('  ranges %3d..%3d (one XCD): waves finish at %.1f .. %.1f us, mean loop %.0f cycles' % (16 * g, 16 * g + 15, en[m].min(), en[m].max(), tot[m].mean()))
    per_wg = {}
    for r in t:
        per_wg.setdefault((int(r[16]), int(r[17])), []).append(r[0])
    wg_tot = np.array([max(v) for v in per_wg.values()])
    print('workgroups: %d; slowest wave per workgroup: mean %.0f, min %.0f, max %.0f cycles (spread %.1f %%)'
          % (len(wg_tot), wg_tot.mean(), wg_tot.min(), wg_tot.max(), 100 * (wg_tot.max() - wg_tot.min()) / wg_tot.mean()))


def xcd_series(reps=8):
    """Per-XCD finish times of `reps` consecutive launches in one process: is the clock pattern over the XCDs stable?"""
    rows, L, G = 160000, 10, 2
    dev = torch.device('cuda', 0)
    d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    store = VariableStore(device=dev, seed=3)
    kw = dict(batch_size=1, dilations=d10[:L], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, rows, 1), generator=g).to(dev)
    frames = torch.rand((1, rows // 80 + 1, 80), generator=g).to(dev)
    cond = engine.RepeatedCondition(frames, 80, 40, rows)
    trace = torch.zeros((2 * 256 * 8, 24), dtype=torch.int64, device=dev)
    os.environ['PWV_PTRACE_PTR'] = str(trace.data_ptr())
    engine.PERSIST = True
    for _ in range(20):
        engine.run_nets(nets, x, cond)
    for k in range(reps):
        torch.cuda.synchronize()
        trace.zero_()
        engine.run_nets(nets, x, cond)
        if k % 2:
            for _ in range(3):
                engine.run_nets(nets, x, cond)      # (every other sample a few launches later)
        torch.cuda.synchronize()
        t = trace.cpu().numpy().astype(np.float64)
        t = t[t[:, 5] > 0]
        t0 = t[:, 20].min()
        en = (t[:, 19] - t0) / 100.0
        fin = [en[(t[:, 17] // 16) == x_].max() for x_ in range(8)]
        print('launch %d: last finish per XCD (us): %s  | spread %.1f us' % (k, ' '.join('%6.1f' % v for v in fin), max(fin) - min(fin)))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'xcd':
        xcd_series()
    else:
        main()
