"""GPU: the persistent stack kernel against the per-layer launches -- bitwise, and timed.
usage: python tools/persist_check.py [rows] [layers] [G] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pwv_amd import engine
from pwv_amd.modules import WaveNet
from pwv_amd.variables import VariableStore


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    prec = sys.argv[5] if len(sys.argv) > 5 else None
    dev = torch.device('cuda', 0)
    d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    dil = (d10 * 3)[:L]
    store = VariableStore(device=dev, seed=3)
    kw = dict(batch_size=1, dilations=dil, filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    hop = 80
    assert rows % hop == 0
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, rows, 1), generator=g).to(dev)
    frames = torch.rand((1, rows // hop + 1, 80), generator=g).to(dev)
    cond = engine.RepeatedCondition(frames, hop, hop // 2, rows)
    engine.run_nets(nets, x, cond, precision=prec)          # creates variables
    for name in list(store.vars):
        if store.vars[name].dim() == 1:
            store.vars[name].normal_(0, 0.1)
    store.version += 1
    engine.clear_plan_cache()

    def run(persist):
        engine.PERSIST = persist
        outs = engine.run_nets(nets, x, cond, precision=prec)
        torch.cuda.synchronize()
        return [o.clone() for o in outs]

    ref = run(False)
    ok = True
    for k in range(3):
        got = run(True)
        st = engine.persist_status()
        same = all(torch.equal(a, b) for a, b in zip(ref, got))
        nbad = sum(int((a != b).sum()) for a, b in zip(ref, got))
        print('persist run %d: status %d, bitwise equal: %s (%d differing elements, max|diff| %.3g)'
              % (k, st, same, nbad, max(float((a - b).abs().max()) for a, b in zip(ref, got))), flush=True)
        ok = ok and same and st == 0
        if st != 0:
            break
    if not ok:
        # where do they differ?
        for gi, (a, b) in enumerate(zip(ref, got)):
            idx = (a != b).reshape(-1).nonzero().reshape(-1)
            if idx.numel():
                print('net %d: first/last differing rows %d .. %d, count %d' % (gi, int(idx[0]), int(idx[-1]), idx.numel()))
                un = torch.unique(idx // 32)
                print('  differing units: %d, first few %s' % (un.numel(), un[:20].tolist()))
    for persist in (False, True):
        engine.PERSIST = persist
        for _ in range(2):
            engine.run_nets(nets, x, cond, precision=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            engine.run_nets(nets, x, cond, precision=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if persist:
            log = engine.EVENT_LOG = []
            for _ in range(reps):
                engine.run_nets(nets, x, cond, precision=prec)
            torch.cuda.synchronize()
            engine.EVENT_LOG = None
            ts = [en[1].elapsed_time(en[2]) * 1e3 for en in log if en[0] == 'persist']
            print('persistent launch alone (zero kernel + %d layers): %s us = %.1f us per layer-pair' % (L - 2, ['%.0f' % v for v in ts], min(ts) / (L - 2)))
        print('%s: %.3f ms per stack call (%d nets x %d layers x %d rows) = %.1f us per layer-pair, status %d'
              % ('persistent' if persist else 'per-layer ', ms, G, L, rows, ms * 1e3 / L, engine.persist_status()), flush=True)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
