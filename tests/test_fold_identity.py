"""CPU: the algebra behind the folded layer 0 (pwv_pack_first_fold_f16x3 / _f32, HISTORY.md section 4 "Layer 0 folded"), checked with the
oracle's own causal convolutions in fp64.

The causal layer of a scalar-input net is h[t] = x[t-1] w0 + x[t] w1 (modules.py:174-183, filter [2, 1, R], no bias), so the
filter / gate convolutions of layer 0 over h (modules.py:216-222, filter [2, R, D], dilation d),

    F[t] = h[t-d] W[0] + h[t] W[1],

are linear maps of the four scalars x[t-d-1], x[t-d], x[t-1], x[t]:  F[t] = sum_q s_q[t] M[q],  M[2 tap + c] = cf[c] @ W[tap].
This pins the index convention the pack kernels use (q = 2 * tap + c; tap 0 <-> the look-back row, c 0 <-> x[. - 1]) and the
zero-padding left of the utterance start, on the CPU; the HIP side is checked against the unfolded kernels and the oracle in
tests/test_gpu_persist.py::test_folded_layer0_is_the_same_function_on_every_path."""
import numpy as np
import pytest

from oracle import iaf_oracle as O


@pytest.mark.parametrize('d', [1, 2, 5, 512])
@pytest.mark.parametrize('n,t', [(1, 40), (3, 700)])
def test_layer0_convolution_is_a_map_of_four_scalars(n, t, d):
    rng = np.random.RandomState(d + t)
    R, D = 64, 64
    x = rng.randn(n, t, 1)
    cf = rng.randn(2, 1, R) * 0.3                  # causal layer filter [W = 2, 1, R]
    w = rng.randn(2, R, D) * 0.1                   # layer 0 filter (or gate) [W = 2, R, D]
    h = O.causal_conv_literal(x, cf, 1)            # modules.py:179-180
    want = O.causal_conv_literal(h, w, d)          # modules.py:218 (221 for the gate)
    assert np.abs(want - O.causal_conv_direct(h, w, d)).max() <= 1e-12

    def shifted(k):                                # x[t - k], zero left of the utterance start (modules.py:32)
        out = np.zeros_like(x[..., 0])
        if k < t:
            out[:, k:] = x[:, :t - k, 0]
        return out
    s = [shifted(d + 1), shifted(d), shifted(1), shifted(0)]           # q = 0..3: the B operand's k values
    m = np.stack([cf[c, 0] @ w[tap] for tap in (0, 1) for c in (0, 1)])   # [4, D], q = 2 * tap + c
    got = sum(s[q][..., None] * m[q][None, None, :] for q in range(4))
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_fold_is_exact_at_the_left_edge():
    """Rows t < d have no look-back row (h[t-d] = 0) and row t = d has h[0] = x[0] w1 only: the shifted scalars carry exactly
    these zeros, so the folded form needs no special case at the utterance start."""
    t, d = 12, 4
    x = np.arange(1, t + 1, dtype=np.float64).reshape(1, t, 1)
    cf = np.stack([np.full((1, 3), 2.0), np.full((1, 3), 5.0)])        # w0 = 2, w1 = 5
    w = np.stack([np.eye(3), 10 * np.eye(3)])
    h = O.causal_conv_literal(x, cf, 1)
    f = O.causal_conv_literal(h, w, d)
    assert f[0, 0, 0] == 10 * (5 * 1)                                   # t = 0: h[0] = x[0] w1 only, no look-back
    assert f[0, d, 0] == (5 * 1) + 10 * (2 * 4 + 5 * 5)                 # t = d: h[0] + 10 h[d]
    assert f[0, d + 1, 0] == (2 * 1 + 5 * 2) + 10 * (2 * 5 + 5 * 6)
