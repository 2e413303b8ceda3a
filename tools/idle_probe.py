#!/usr/bin/env python
"""Is idle time free on a power-capped chip?  Adds a spin kernel of X us (one wave, no memory traffic) behind every flow of
the headline forward (graph replay) and reports the step time: if 4 x X us of added idle cost (much) less than 4 x X us,
the chip wins the idle time back as clock, and REMOVING idle time (flow boundaries, prologue) will return as little.
   python tools/idle_probe.py [case] [us ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwv_amd import modules  # noqa: E402
from pwv_amd.graph import GraphedVocoder  # noqa: E402
from pwv_amd.hparam import hparam as hp  # noqa: E402
from pwv_amd.models import IAFVocoder  # noqa: E402
from pwv_amd.variables import VariableStore  # noqa: E402

dev = torch.device('cuda', 0)
case = sys.argv[1] if len(sys.argv) > 1 else 'bench/c3'
delays = [float(v) for v in sys.argv[2:]] or [0, 25, 50, 100, 0]
hp.set_hparam_yaml(case)
length, n = int(hp.generate.length), int(hp.generate.batch_size)
store = VariableStore(device=dev, seed=2)
mel = torch.rand((n, 1 + length // int(hp.signal.hop_length), int(hp.signal.n_mels)), device=dev) * 2 - 1
z = torch.randn((n, length, 1), device=dev)

# cycles of torch.cuda._sleep per microsecond
torch.cuda._sleep(1000)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 10_000_000 / (e0.elapsed_time(e1) * 1e3)

orig = modules.LinearIAFLayer.__call__
state = {'us': 0.0}


def patched(self, input, condition=None):
    out = orig(self, input, condition)
    if state['us'] > 0:
        torch.cuda._sleep(int(state['us'] * cyc_per_us))
    return out


modules.LinearIAFLayer.__call__ = patched
for us in delays:
    state['us'] = us
    model = IAFVocoder(batch_size=n, length=length, store=store)
    g = GraphedVocoder(model)
    for _ in range(5):
        g(mel, z=z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 40
    for _ in range(iters):
        g(mel, z=z)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print('%s: +%5.1f us of idle behind each of the %d flows -> %.4f ms per step' % (case, us, int(hp.model.n_iaf), ms), flush=True)
