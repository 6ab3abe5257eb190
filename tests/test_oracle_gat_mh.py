"""The multi-head GAT extension has no reference implementation (parity unpinned); its
float64 oracle is pinned against itself: analytic backward == central finite differences
of the loss, for every parameter kind (W, a_l, a_r) and for the input features."""
import numpy as np

import gat_mh_oracle as go
import partition_oracle as po


def test_gat_mh_backward_matches_finite_differences():
    rng = np.random.default_rng(0)
    V, E = 30, 120
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    dims, heads = [7, 12, 5], [3, 2]
    X = rng.standard_normal((V, dims[0]))
    labels = rng.integers(0, dims[-1], V)
    params = []
    for l in range(2):
        width = dims[l + 1] * (heads[l] if l == 1 else 1)      # last layer: K_out heads of C each
        params.append([rng.standard_normal((dims[l] if l == 0 else dims[1], width)) * 0.4,
                       rng.standard_normal(width) * 0.4, rng.standard_normal(width) * 0.4])
    _, _, loss, _, grads = go.epoch(g, X, labels, params, heads)

    def loss_of(ps, Xin=X):
        return go.epoch(g, Xin, labels, ps, heads)[2]

    eps = 1e-6
    for l in range(2):
        for pi, key in ((0, "dW"), (1, "da_l"), (2, "da_r")):
            P = params[l][pi]
            flat = P.reshape(-1)
            for idx in rng.choice(flat.size, size=min(6, flat.size), replace=False):
                old = flat[idx]
                flat[idx] = old + eps
                lp = loss_of(params)
                flat[idx] = old - eps
                lm = loss_of(params)
                flat[idx] = old
                num = (lp - lm) / (2 * eps)
                ana = grads[l][key].reshape(-1)[idx]
                assert abs(num - ana) < 1e-5 * max(1.0, abs(num)), (l, key, idx, num, ana)
    # gradient w.r.t. the input features of layer 0
    for _ in range(5):
        i, j = rng.integers(0, V), rng.integers(0, dims[0])
        Xp, Xm = X.copy(), X.copy()
        Xp[i, j] += eps
        Xm[i, j] -= eps
        num = (loss_of(params, Xp) - loss_of(params, Xm)) / (2 * eps)
        assert abs(num - grads[0]["dH"][i, j]) < 1e-5 * max(1.0, abs(num))
    assert np.isfinite(loss)


def test_gat_mh_attention_is_a_distribution():
    rng = np.random.default_rng(1)
    V = 20
    s, d = rng.integers(0, V, 60), rng.integers(0, V, 60)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    fw = go.layer_forward(g, rng.standard_normal((V, 5)), rng.standard_normal((5, 8)), rng.standard_normal(8),
                          rng.standard_normal(8), 4)
    sums = np.zeros((V, 4))
    np.add.at(sums, fw["dst"], fw["alpha"])
    assert np.allclose(sums, 1.0)
