"""gat_mh_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Self-defined float64 oracle for the multi-head GAT *extension* (BASELINE.json config 3
wording: "8-head, per-edge attention softmax + weighted SpMM").  The reference has NO
implementation of this (its GAT is a single-head prototype whose edge score depends on the
destination only and has no softmax: SURVEY.md 0-6, CPU_comm.cpp:190-242), so parity with
the reference is UNPINNED by construction; this file is the definition the HIP kernels are
checked against, and tests/test_oracle_gat_mh.py pins the definition's own backward pass
with central finite differences.

Definition (Velickovic et al. 2018, as in common GATConv implementations; no bias, no
dropout, negative slope 0.2, self edge included in every neighbourhood):
    Z  = H W                                 W: F_in x (K*D)
    el[u,k] = <Z[u,k,:], a_l[k,:]>           er[v,k] = <Z[v,k,:], a_r[k,:]>
    s[e,k]  = LeakyReLU_0.2(el[src e,k] + er[dst e,k])      e in in(v) + {v->v}
    alpha   = softmax over the in-edges of v (per head)
    O[v,k,:] = sum_e alpha[e,k] Z[src e,k,:]
  hidden layer: H' = ELU(O) (heads concatenated);  last layer: logits = mean_k O[:,k,:].
  Loss gradient at the logits: softmax(logits) - onehot (like Engine::predictGAT).
Duplicate edges are kept (multigraph), like everywhere else on this path.
"""
import numpy as np

SLOPE = 0.2


def _lrelu(x):
    return np.where(x > 0, x, SLOPE * x)


def _edges(g):
    """(src, dst) of all in-edges incl. one self edge per vertex; ghost-free partitions only."""
    N = int(g["localVtxCnt"])
    ptr = g["colPtr"].astype(np.int64)
    dst = np.repeat(np.arange(N), np.diff(ptr))
    src = g["rowIdx"].astype(np.int64)
    assert src.size == 0 or src.max() < N, "single-partition oracle"
    return np.concatenate([src, np.arange(N)]), np.concatenate([dst, np.arange(N)]), N


def layer_forward(g, H, W, a_l, a_r, K):
    src, dst, N = _edges(g)
    Z = H.astype(np.float64) @ W.astype(np.float64)
    D = Z.shape[1] // K
    Z3 = Z.reshape(N, K, D)
    el = (Z3 * a_l.reshape(K, D)).sum(-1)
    er = (Z3 * a_r.reshape(K, D)).sum(-1)
    s = _lrelu(el[src] + er[dst])                       # E' x K
    m = np.full((N, K), -np.inf)
    np.maximum.at(m, dst, s)
    p = np.exp(s - m[dst])
    den = np.zeros((N, K))
    np.add.at(den, dst, p)
    alpha = p / den[dst]
    O = np.zeros((N, K, D))
    np.add.at(O, dst, alpha[:, :, None] * Z3[src])
    return dict(Z=Z, el=el, er=er, m=m, den=den, alpha=alpha, O=O.reshape(N, K * D), src=src, dst=dst, K=K, D=D)


def layer_backward(g, H, W, a_l, a_r, fw, dO):
    src, dst, K, D = fw["src"], fw["dst"], fw["K"], fw["D"]
    N = H.shape[0]
    Z3 = fw["Z"].reshape(N, K, D)
    dO3 = dO.astype(np.float64).reshape(N, K, D)
    alpha = fw["alpha"]
    dalpha = (dO3[dst] * Z3[src]).sum(-1)                  # E' x K
    t = np.zeros((N, K))
    np.add.at(t, dst, alpha * dalpha)
    ds = alpha * (dalpha - t[dst])
    pre = fw["el"][src] + fw["er"][dst]
    dpre = ds * np.where(pre > 0, 1.0, SLOPE)
    d_el = np.zeros((N, K))
    d_er = np.zeros((N, K))
    np.add.at(d_el, src, dpre)
    np.add.at(d_er, dst, dpre)
    dZ3 = np.zeros((N, K, D))
    np.add.at(dZ3, src, alpha[:, :, None] * dO3[dst])
    dZ3 += d_el[:, :, None] * a_l.reshape(1, K, D) + d_er[:, :, None] * a_r.reshape(1, K, D)
    da_l = (d_el[:, :, None] * Z3).sum(0).reshape(-1)
    da_r = (d_er[:, :, None] * Z3).sum(0).reshape(-1)
    dZ = dZ3.reshape(N, K * D)
    dW = H.astype(np.float64).T @ dZ
    dH = dZ @ W.astype(np.float64).T
    return dict(dZ=dZ, dW=dW, dH=dH, da_l=da_l, da_r=da_r, t=t, d_el=d_el, d_er=d_er)


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def elu_grad(x):
    return np.where(x > 0, 1.0, np.exp(np.minimum(x, 0)))


def epoch(g, X, labels, params, heads):
    """params: list of (W, a_l, a_r); heads: K per layer.  Returns forward caches, the loss
    (mean cross entropy over all vertices, for the finite-difference check) and gradients."""
    L = len(params)
    H = X.astype(np.float64)
    fws, Hs = [], [H]
    for l, (W, a_l, a_r) in enumerate(params):
        fw = layer_forward(g, H, W, a_l, a_r, heads[l])
        fws.append(fw)
        if l < L - 1:
            H = elu(fw["O"])
        else:
            N = H.shape[0]
            H = fw["O"].reshape(N, heads[l], -1).mean(1)      # logits
        Hs.append(H)
    logits = Hs[-1]
    N, C = logits.shape
    e = np.exp(logits - logits.max(1, keepdims=True))
    prob = e / e.sum(1, keepdims=True)
    onehot = np.eye(C)[labels]
    loss = -np.log((prob * onehot).sum(1)).sum()
    grads = [None] * L
    dlogits = prob - onehot                                   # predictGAT-style gradient
    dO = np.repeat(dlogits[:, None, :] / heads[-1], heads[-1], axis=1).reshape(N, -1)
    for l in range(L - 1, -1, -1):
        W, a_l, a_r = params[l]
        bw = layer_backward(g, Hs[l], W, a_l, a_r, fws[l], dO)
        grads[l] = bw
        if l > 0:
            dO = bw["dH"] * elu_grad(fws[l - 1]["O"])
    return fws, Hs, loss, dlogits, grads
