"""Generate tests/golden/*.npz: seeded inputs + float64 expected outputs of the oracle.

PARITY UNPINNED upstream: the reference ships no golden vectors and TensorFlow cannot be
imported here, so these fixtures pin the ORACLE (two independent conv formulations + KATs in
tests/test_oracle.py keep the oracle honest) and then the HIP path against it.
Weights are regenerated from (config, seed) with init_weights; their SHA-256 is stored so any
drift of the generator is caught.   Run:  python -m oracle.make_golden
"""
import hashlib
import json
import os

import numpy as np

from . import iaf_oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def weights_digest(w):
    h = hashlib.sha256()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


def cfg_dict(cfg):
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(cfg).items()}


VOCODER_CASES = {
    'vocoder_repeat': dict(cfg=dict(dilations=[[1, 2, 4], [1, 2, 4, 8]], n_iaf=2), n=2, length=480),
    'vocoder_tconv': dict(cfg=dict(dilations=[[1, 2, 4], [1, 2, 4, 8]], n_iaf=2, cond_upsample_method='transposed_conv'),
                          n=2, length=320),
    'vocoder_shared': dict(cfg=dict(dilations=[[1, 2, 4], [1, 2, 4, 8]], n_iaf=2, shared_nets=True), n=1, length=400),
    'vocoder_skipconn': dict(cfg=dict(dilations=[[1, 2, 4, 8, 16]], n_iaf=1, use_skip_connection=True), n=2, length=320),
    'vocoder_default_0p25s': dict(cfg=dict(), n=1, length=4000),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.RandomState(123)
    conv = {}
    for i, (N, T, Cin, Cout, W, d) in enumerate([(1, 10, 1, 4, 2, 1), (3, 33, 5, 7, 2, 3), (3, 37, 5, 7, 3, 4),
                                                 (1, 64, 8, 8, 2, 8), (2, 10, 4, 4, 2, 512), (3, 37, 8, 12, 3, 16)]):
        x = rng.randn(N, T, Cin).astype(np.float32)
        f = rng.randn(W, Cin, Cout).astype(np.float32)
        y = O.causal_conv_literal(x.astype(np.float64), f.astype(np.float64), d)
        conv['x%d' % i], conv['f%d' % i], conv['d%d' % i], conv['y%d' % i] = x, f, np.int64(d), y
    np.savez_compressed(os.path.join(OUT, 'causal_conv.npz'), **conv)

    for name, case in VOCODER_CASES.items():
        cfg = O.ModelConfig(**case['cfg'])
        w = O.init_weights(cfg, seed=2)
        mel, z = O.synthetic_inputs(case['n'], case['length'], cfg)
        y, flows = O.iaf_vocoder_forward(w, mel, z, cfg, return_flows=True)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), cfg=json.dumps(case['cfg']), weight_seed=np.int64(2),
                            weights_sha256=weights_digest(w), mel=mel, z=z, y=y,
                            flow0=flows[0])
        print(name, y.shape, float(np.abs(y).max()))


if __name__ == '__main__':
    main()
