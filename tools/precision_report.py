#!/usr/bin/env python
"""Max |HIP - oracle_fp64| of both arithmetic variants on the golden fixtures + a longer default-model run, and (round 6, VERDICT r05
item 7) the RANGE GUARD of the split-fp16 arithmetic quantified: per operand class the limit the guard derives against what a forward
shows, on the glorot weights every parity test uses and on a "trained-like" set (per-tensor gains log-uniform in [0.5, 4], heavy-tailed
biases: Student-t with 3 degrees of freedom x 0.1), each with the f16x3 / f32 error against the fp64 oracle.
   python tools/precision_report.py [L]            goldens (+ default model at L samples)
   python tools/precision_report.py --range [L]    the range table (default L = 16000); --json FILE writes it"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import iaf_oracle as O  # noqa: E402
from oracle.make_golden import VOCODER_CASES  # noqa: E402
from tests.util import run_vocoder_hip  # noqa: E402



def trained_like(weights, seed=7):
    """The glorot set re-scaled the way training moves weights apart: every matrix by its own gain, log-uniform in [0.5, 4]; every bias
    heavy-tailed (Student-t, 3 degrees of freedom, x 0.1)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name in sorted(weights):
        v = weights[name]
        if v.ndim == 1:
            out[name] = (0.1 * rng.standard_t(3, size=v.shape)).astype(v.dtype)
        else:
            out[name] = (v * np.exp(rng.uniform(np.log(0.5), np.log(4.0)))).astype(v.dtype)
    return out


def range_table(L, json_path=None):
    from pwv_amd import engine
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    from tests.util import set_hparams
    cfg = O.ModelConfig()
    base = O.init_weights(cfg, seed=2)
    mel, zz = O.synthetic_inputs(1, L, cfg)
    report = {}
    for label, w in (('glorot', base), ('trained_like', trained_like(base))):
        set_hparams(cfg)
        store = VariableStore(device=dev)
        store.load_dict(w)
        model = IAFVocoder(batch_size=1, length=L, store=store, precision='f16x3')
        tm, tz = torch.from_numpy(mel).to(dev), torch.from_numpy(zz).to(dev)
        rep = engine.range_report(lambda: model(None, tm, is_training=False, z=tz, verify=False))
        tripped = engine.range_flag_raised()
        engine.clear_range_flag()
        want = O.iaf_vocoder_forward(w, mel, zz, cfg)
        errs = {}
        for prec in ('f32', 'f16x3'):
            try:
                got = run_vocoder_hip(cfg, w, mel, zz, dev, precision=prec)
                errs[prec] = float(np.abs(got - want).max())
            except Exception as e:      # (the guard refusing the split-fp16 arithmetic is an answer too)
                errs[prec] = type(e).__name__
                engine.clear_range_flag()
        rep.update(guard_tripped=bool(tripped), out_max=float(np.abs(want).max()), err_vs_fp64=errs)
        report[label] = rep
        print('%s weights, default model, L = %d: |y|max %.3g, guard %s, range_margin %.3g, err vs fp64: %s' % (
            label, L, rep['out_max'], 'TRIPPED' if tripped else 'quiet', rep['range_margin'], errs))
        for k, c in sorted(rep['classes'].items()):
            print('    %-13s limit %10.4g   observed / bounded %10.4g   margin %8.3g x' % (k, c['limit'], c['observed'], c['margin']))
    if json_path:
        with open(json_path, 'w') as f:
            json.dump(report, f, indent=1, sort_keys=True)
    return report


GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
dev = torch.device('cuda', 0)
if len(sys.argv) > 1 and sys.argv[1] == '--range':
    rest = [a for a in sys.argv[2:]]
    jp = rest[rest.index('--json') + 1] if '--json' in rest else None
    nums = [a for a in rest if a.isdigit()]
    range_table(int(nums[0]) if nums else 16000, jp)
    sys.exit(0)
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
