"""What the COMPOSED path costs ("correct but unbenchmarked"): a model outside the fused kernels' shape runs every
op of modules.py:129-259 as its own HIP kernel (engine / modules.WaveNet._call_unfused).  Times the default architecture on the fused
path next to (a) the same architecture forced onto the composed path and (b) a narrower net (R = D = 32, S = 64) that can only run there.

    python tools/unfused_bench.py [--length 16000]
"""
import argparse
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(model, mel, steps=5):
    import torch
    model(None, mel, is_training=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model(None, mel, is_training=False, verify=False)
    torch.cuda.synchronize()
    model.verify()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--length', type=int, default=16000)
    a = ap.parse_args()
    import torch
    from pwv_amd import modules
    from pwv_amd.hparam import hparam as hp
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    dev = torch.device('cuda', 0)
    warnings.simplefilter('ignore')
    hp.set_hparam_yaml('default')
    mel = (torch.rand((1, 1 + a.length // hp.signal.hop_length, hp.signal.n_mels)) * 2 - 1).to(dev)
    rows = []
    m = IAFVocoder(batch_size=1, length=a.length, store=VariableStore(device=dev, seed=2))
    rows.append(('default architecture, fused path (one launch per flow)', timed(m, mel, 20)))
    real = modules.WaveNet.fused_supported
    modules.WaveNet.fused_supported = lambda self, cond: False
    try:
        m2 = IAFVocoder(batch_size=1, length=a.length, store=VariableStore(device=dev, seed=2))
        rows.append(('default architecture forced onto the composed path', timed(m2, mel)))
    finally:
        modules.WaveNet.fused_supported = real
    hp.model.residual_channels = hp.model.dilation_channels = 32
    hp.model.skip_channels = 64
    m3 = IAFVocoder(batch_size=1, length=a.length, store=VariableStore(device=dev, seed=2))
    rows.append(('R = D = 32, S = 64 (composed path only)', timed(m3, mel)))
    for name, ms in rows:
        print('%-62s %9.2f ms per forward = %8.2f M samples/s' % (name, ms, a.length / ms / 1e3))


if __name__ == '__main__':
    main()
