"""GPU: random shapes through the persistent launch against the per-layer launches, bit for bit (the protocol's stress test).

    python tools/persist_fuzz.py [--cases 200] [--seed 1] [--max-rows 40000] [--load]

Every case: random (utterances, length, layers, nets, units per workgroup, layers per run, arithmetic) -- lengths that are no multiple of 32,
utterance starts inside units, one to seven units per workgroup (the short-input instantiation) and a few beyond (the general one) -- the per-layer
result once, then the persistent launch FOUR times on one workspace (the control words must be left clean), each compared with torch.equal.
--load runs a second stream of unrelated memory traffic beside it (uneven load: hand-off bugs hide on an idle chip).
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pwv_amd import engine, _lib
from pwv_amd.modules import WaveNet
from pwv_amd.variables import VariableStore

D10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--max-rows', type=int, default=40000)
    ap.add_argument('--load', action='store_true')
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    dev = torch.device('cuda', 0)
    side = torch.cuda.Stream()
    junk = torch.empty((64 << 20) // 4, device=dev)
    nets_cache = {}
    fails = 0
    short = general = 0
    for case in range(a.cases):
        G = rnd.choice((1, 2, 2))
        L = rnd.choice((3, 5, 8, 10, 12, 15, 30))
        n = rnd.choice((1, 1, 1, 2, 3))
        rows = rnd.randint(64, a.max_rows)
        t = max(40, rows // n)
        min_units = rnd.choice((0, 0, 1, 2, 3, 5, 7))
        max_layers = rnd.choice((32, 32, 32, 4, 7))
        prec = rnd.choice(('f16x3', 'f16x3', 'f32'))
        key = (L, G)
        if key not in nets_cache:
            store = VariableStore(device=dev, seed=3 + L)
            kw = dict(batch_size=1, dilations=(D10 * 3)[:L], filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
                      quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
            nets_cache[key] = (store, [WaveNet(name='n%d' % g, **kw) for g in range(G)], False)
        store, nets, init = nets_cache[key]
        g = torch.Generator().manual_seed(case)
        x = torch.randn((n, t, 1), generator=g).to(dev)
        frames = torch.rand((n, (t + 39) // 80 + 1, 80), generator=g).to(dev)
        cond = engine.RepeatedCondition(frames, 80, 40, t)
        if not init:
            engine.run_nets(nets, x, cond, precision=prec)
            for name in list(store.vars):
                if store.vars[name].dim() == 1:
                    store.vars[name].normal_(0, 0.1)
            store.version += 1
            nets_cache[key] = (store, nets, True)
        engine.PERSIST = False
        ref = [o.clone() for o in engine.run_nets(nets, x, cond, precision=prec)]
        engine.PERSIST, engine.PERSIST_MIN_UNITS, engine.PERSIST_MAX_LAYERS = True, min_units, max_layers
        log = engine.EVENT_LOG = []
        ok = True
        for rep in range(4):
            if a.load:
                with torch.cuda.stream(side):
                    junk.add_(1.0)
            got = engine.run_nets(nets, x, cond, precision=prec)
            torch.cuda.synchronize()
            if engine.persist_status() != 0:
                print('case %d: GIVE-UP status %d' % (case, engine.persist_status()))
                engine.resume_persist()
                ok = False
                break
            for r_, g_ in zip(ref, got):
                if not torch.equal(r_, g_):
                    ok = False
        engine.EVENT_LOG = None
        kinds = sorted({e[7] for e in log if e[0] == 'persist'})
        short += 1 in kinds
        general += 0 in kinds
        if not ok:
            fails += 1
            print('case %d FAILED: n %d t %d L %d G %d min_units %d max_layers %d %s (instantiations %s)' % (case, n, t, L, G, min_units, max_layers, prec, kinds), flush=True)
    print('%d cases, %d failed; %d took the short-input instantiation, %d the general one' % (a.cases, fails, short, general))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
