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


def test_gat_mh_oracle_matches_torch_autograd():
    """A second, independent pin of the definition: the same forward written with torch tensor ops (float64, CPU) and
    differentiated by torch's autograd must reproduce EVERY gradient the oracle's hand-written backward returns (the
    finite-difference test above samples a few entries; this one covers all of them, and an engine nobody here wrote)."""
    import torch
    rng = np.random.default_rng(5)
    V, E = 40, 300
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    dims, heads = [9, 16, 6], [4, 2]
    X = rng.standard_normal((V, dims[0]))
    labels = rng.integers(0, dims[-1], V)
    params = []
    for l in range(2):
        width = dims[l + 1] * (heads[l] if l == 1 else 1)
        params.append([rng.standard_normal((dims[l], width)) * 0.4, rng.standard_normal(width) * 0.4,
                       rng.standard_normal(width) * 0.4])
    fws, Hs, loss, _, grads = go.epoch(g, X, labels, params, heads)

    N = V
    ptr = g["colPtr"].astype(np.int64)
    dst = torch.tensor(np.concatenate([np.repeat(np.arange(N), np.diff(ptr)), np.arange(N)]))
    src = torch.tensor(np.concatenate([g["rowIdx"].astype(np.int64), np.arange(N)]))
    tX = torch.tensor(X, dtype=torch.float64, requires_grad=True)
    tp = [[torch.tensor(p, dtype=torch.float64, requires_grad=True) for p in ps] for ps in params]
    H = tX
    for l, (W, a_l, a_r) in enumerate(tp):
        K = heads[l]
        Z3 = (H @ W).reshape(N, K, -1)
        el = (Z3 * a_l.reshape(K, -1)).sum(-1)
        er = (Z3 * a_r.reshape(K, -1)).sum(-1)
        sc = torch.nn.functional.leaky_relu(el[src] + er[dst], 0.2)
        m = torch.full((N, K), -float("inf"), dtype=torch.float64).scatter_reduce(0, dst[:, None].expand(-1, K), sc, "amax")
        p = torch.exp(sc - m[dst].detach())
        den = torch.zeros((N, K), dtype=torch.float64).index_add(0, dst, p)
        alpha = p / den[dst]
        O = torch.zeros_like(Z3).index_add(0, dst, alpha[:, :, None] * Z3[src])
        H = torch.nn.functional.elu(O.reshape(N, -1)) if l == 0 else O.mean(1)
    tl = torch.nn.functional.cross_entropy(H, torch.tensor(labels), reduction="sum")
    tl.backward()
    assert abs(tl.item() - loss) < 1e-9 * max(1.0, abs(loss))
    for l in range(2):
        for pi, key in ((0, "dW"), (1, "da_l"), (2, "da_r")):
            a, b = tp[l][pi].grad.numpy().reshape(-1), grads[l][key].reshape(-1)
            assert np.abs(a - b).max() < 1e-10 * max(1.0, np.abs(b).max()), (l, key)
    assert np.abs(tX.grad.numpy() - grads[0]["dH"]).max() < 1e-10 * max(1.0, np.abs(grads[0]["dH"]).max())
