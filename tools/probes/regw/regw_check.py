#!/usr/bin/env python
"""Writes the output of one seeded forward of a 1-flow model (two 10-layer nets) on the PER-LAYER path to OUT.npy, so that
two processes with different environments (e.g. PWV_REGW=0 / 1: the register-stationary layer kernel) can be compared bit for bit:
   PWV_PERSIST=0 PWV_REGW=0 python tools/probes/regw/regw_check.py /tmp/a.npy; PWV_PERSIST=0 PWV_REGW=1 python tools/probes/regw/regw_check.py /tmp/b.npy
   python tools/probes/regw/regw_check.py --cmp /tmp/a.npy /tmp/b.npy"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np  # noqa: E402


def main():
    if sys.argv[1] == '--cmp':
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        print('bitwise equal' if same else 'DIFFERENT: max abs diff %g' % float(np.abs(a - b).max()), a.shape, float(np.abs(a).max()))
        sys.exit(0 if same else 1)
    import torch
    from oracle import iaf_oracle as O
    from tests.util import run_vocoder_hip
    shapes = [(1, 32000), (3, 4000 + 80), (2, 80 * 13)]
    outs = []
    for n, length in shapes:
        cfg = O.ModelConfig(dilations=[[1, 2, 4, 8, 16, 32, 64, 128, 256, 512]], n_iaf=1)
        weights = O.init_weights(cfg, seed=2)
        mel, z = O.synthetic_inputs(n, length, cfg)
        outs.append(run_vocoder_hip(cfg, weights, mel, z, torch.device('cuda', 0)).reshape(-1))
    np.save(sys.argv[1], np.concatenate(outs))


if __name__ == '__main__':
    main()
