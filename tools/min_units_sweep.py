"""Short inputs: units per workgroup and layer of the persistent launch (engine.PERSIST_MIN_UNITS) against the step time, default model,
HIP-graph replay.  python tools/min_units_sweep.py [--length 16000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--length', type=int, default=16000)
    ap.add_argument('--case', default='bench/c3')
    a = ap.parse_args()
    import torch
    from pwv_amd import engine
    from pwv_amd.graph import GraphedVocoder
    from pwv_amd.hparam import hparam as hp
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    hp.set_hparam_yaml(a.case)
    dev = torch.device('cuda', 0)
    store = VariableStore(device=dev, seed=2)
    mel = (torch.rand((1, 1 + a.length // hp.signal.hop_length, hp.signal.n_mels)) * 2 - 1).to(dev)
    model = IAFVocoder(batch_size=1, length=a.length, store=store)
    model(None, mel, is_training=False)
    for rounds in range(2):
        for mu in (1, 2, 3, 4, 5, 6, 8):
            engine.PERSIST_MIN_UNITS = mu
            g = GraphedVocoder(model)
            for _ in range(5):
                g(mel)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(40):
                g(mel)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 40 * 1e3
            g.verify()
            print('min_units %d: %.4f ms per forward' % (mu, ms), flush=True)


if __name__ == '__main__':
    main()
