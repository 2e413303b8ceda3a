#!/usr/bin/env python
"""Eager launches vs HIP-graph replay of the same forward (pwv_amd/graph.py), bit-compared.
   python tools/graph_bench.py [case] [length]     e.g.  bench/c3   |   bench/c3 16000   |   bench/c1"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pwv_amd.graph import GraphedVocoder  # noqa: E402
from pwv_amd.hparam import hparam as hp  # noqa: E402
from pwv_amd.models import IAFVocoder  # noqa: E402
from pwv_amd.variables import VariableStore  # noqa: E402

dev = torch.device('cuda', 0)
case = sys.argv[1] if len(sys.argv) > 1 else 'bench/c3'
hp.set_hparam_yaml(case)
if len(sys.argv) > 2:
    hp.generate.length = int(sys.argv[2])
length, n = int(hp.generate.length), int(hp.generate.batch_size)
store = VariableStore(device=dev, seed=2)
model = IAFVocoder(batch_size=n, length=length, store=store)
mel = torch.rand((n, model.t_mel, int(hp.signal.n_mels)), device=dev) * 2 - 1
z = torch.randn((n, length, 1), device=dev)


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


y_eager = model(None, mel, is_training=False, z=z).clone()
graphed = GraphedVocoder(model)
y_graph = graphed(mel, z=z).clone()
t_e = timeit(lambda: model(None, mel, is_training=False, z=z, verify=False))      # enqueue-only: the timed loop must not synchronise
t_g = timeit(lambda: graphed(mel, z=z))
print('%s: %d x %d samples | eager %.3f ms (%.1f M samples/s) | graph %.3f ms (%.1f M samples/s) | bit-identical: %s' % (
    case, n, length, t_e, n * length / t_e / 1e3, t_g, n * length / t_g / 1e3, torch.equal(y_eager, y_graph)))
