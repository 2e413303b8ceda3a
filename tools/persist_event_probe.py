#!/usr/bin/env python
"""How long does a persistent launch take in a host-enqueued forward, by HIP events, forward after forward?  (bench.py's
live roofline figure is taken this way; this prints the series.)  python tools/persist_event_probe.py [forwards]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwv_amd import engine  # noqa: E402
from pwv_amd.graph import GraphedVocoder  # noqa: E402
from pwv_amd.hparam import hparam as hp  # noqa: E402
from pwv_amd.models import IAFVocoder  # noqa: E402
from pwv_amd.variables import VariableStore  # noqa: E402

dev = torch.device('cuda', 0)
hp.set_hparam_yaml('bench/c3')
n, length = 1, 160000
store = VariableStore(device=dev, seed=2)
model = IAFVocoder(batch_size=n, length=length, store=store)
mel = torch.rand((n, model.t_mel, 80), device=dev) * 2 - 1
g = GraphedVocoder(model)
for _ in range(100):          # heat the chip the way the timed loop does
    g(mel)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    g(mel)
torch.cuda.synchronize()
print('graph replay: %.3f ms per forward' % ((time.perf_counter() - t0) / 50 * 1e3))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
engine.EVENT_LOG = []
t0 = time.perf_counter()
for _ in range(reps):
    model(None, mel, is_training=False, verify=False)
torch.cuda.synchronize()
print('host-enqueued: %.3f ms per forward' % ((time.perf_counter() - t0) / reps * 1e3))
log, engine.EVENT_LOG = engine.EVENT_LOG, None
per = [en[1].elapsed_time(en[2]) * 1e3 for en in log if en[0] == 'persist']
for k in range(reps):
    print('forward %2d: persistent launches %s us' % (k, ' '.join('%.0f' % v for v in per[4 * k:4 * k + 4])))
