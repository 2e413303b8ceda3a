#!/usr/bin/env python
"""Max |HIP - oracle_fp64| of both arithmetic variants on the golden fixtures + a longer default-model run."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import iaf_oracle as O  # noqa: E402
from oracle.make_golden import VOCODER_CASES  # noqa: E402
from tests.util import run_vocoder_hip  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
dev = torch.device('cuda', 0)
for name in sorted(VOCODER_CASES):
    z = np.load(os.path.join(GOLD, name + '.npz'))
    cfg = O.ModelConfig(**json.loads(str(z['cfg'])))
    w = O.init_weights(cfg, seed=int(z['weight_seed']))
    errs = {}
    for prec in ('f32', 'f16x3'):
        got = run_vocoder_hip(cfg, w, z['mel'], z['z'], dev, precision=prec)
        errs[prec] = float(np.abs(got - z['y']).max())
    y32 = O.iaf_vocoder_forward(w, z['mel'], z['z'], cfg, dtype=np.float32)
    print('%-24s |y|max %.3f  f32 %.3e  f16x3 %.3e  (numpy fp32 oracle %.3e)' % (
        name, float(np.abs(z['y']).max()), errs['f32'], errs['f16x3'], float(np.abs(y32 - z['y']).max())))
if len(sys.argv) > 1:
    L = int(sys.argv[1])
    cfg = O.ModelConfig()
    w = O.init_weights(cfg, seed=2)
    mel, zz = O.synthetic_inputs(1, L, cfg)
    want = O.iaf_vocoder_forward(w, mel, zz, cfg)
    for prec in ('f32', 'f16x3'):
        got = run_vocoder_hip(cfg, w, mel, zz, dev, precision=prec)
        print('default model L=%d  %s: max err %.3e  rms err %.3e' % (L, prec, np.abs(got - want).max(), np.sqrt(((got - want) ** 2).mean())))
