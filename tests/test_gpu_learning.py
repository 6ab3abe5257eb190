"""End-to-end sanity beyond parity: the engine actually learns.  A planted-partition graph (communities = classes,
features = noisy community indicators) must be classified well after a few dozen epochs -- in the reference's order,
in the transform-first order, with the reference GAT prototype and with the multi-head GAT extension.  Validation
accuracy is the reference's own statistic (CPUComm::getTrainStat rows [0.66 N, 0.76 N))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _task(V=6000, C=6, deg=12, F=24, seed=4):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, C, V)
    members = [np.nonzero(y == c)[0] for c in range(C)]
    s = rng.integers(0, V, V * deg // 2)
    same = rng.random(s.size) < 0.8
    d = np.where(same, [members[y[v]][rng.integers(0, members[y[v]].size)] for v in s], rng.integers(0, V, s.size))
    s, d = np.concatenate([s, d]), np.concatenate([d, s])
    X = rng.standard_normal((V, F)).astype(np.float32)
    X[np.arange(V), y] += 1.0                                       # weak per-vertex signal, the neighbourhood makes it strong
    return s.astype(np.uint32), d.astype(np.uint32), X, y.astype(np.uint32), C


@pytest.mark.parametrize("mode", ["gcn", "gcn_transform_first", "gat", "gatmh"])
def test_planted_communities_are_learned(mode):
    import dorylus_amd as da
    s, d, X, y, C = _task()
    V, F = X.shape
    part = da.Partition.build(s, d, np.zeros(V, np.int32), 0, 1)
    gnn = {"gcn": da.GCN, "gcn_transform_first": da.GCN, "gat": da.GAT, "gatmh": da.GATMH}[mode]
    ctx = da.Context(0)
    ctx.configure(gnn, [F, 16, C], V)
    if mode == "gatmh":
        ctx.gatmh_heads([4, 1])
    if mode == "gcn_transform_first":
        ctx.set_option("gcn_transform_first", 2)
    part.upload(ctx)
    ctx.preallocate()
    ctx.upload(0, "x" if gnn == da.GCN else "h", X)
    ctx.labels_upload(y)
    ctx.weights_init_xavier()
    if mode == "gatmh":                                             # attention vectors start small and random
        rng = np.random.default_rng(1)
        for l, zw in ((0, 16), (1, C)):
            ctx.weight_set(l, "a_l", (rng.standard_normal(zw) * 0.1).astype(np.float32))
            ctx.weight_set(l, "a_r", (rng.standard_normal(zw) * 0.1).astype(np.float32))
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    acc = []
    for _ in range(12):
        eng.run(5)
        if gnn == da.GCN:
            a, l, n = ctx.train_stat()
            acc.append(a / n)
        else:                                                       # GAT variants: accuracy from the logits of the last epoch
            logits = ctx.download(1, "ah" if mode == "gat" else "logits")
            lo, hi = int(V * 0.66), int(V * 0.66) + int(V * 0.1)
            acc.append(float((logits[lo:hi].argmax(1) == y[lo:hi]).mean()))
    eng.close()
    ctx.close()
    print(f"\n{mode}: validation accuracy {acc[0]:.3f} -> {acc[-1]:.3f}")
    if mode == "gat":
        # The reference's GAT is a prototype: unnormalised edge weights lrelu(z_dst . a), an attention
        # vector the weight server never updates (weightserver.cpp:112-116) and softmax applied to the aggregate of the
        # last layer.  The mirror reproduces it stage by stage (tests/test_gpu_parity.py); it is not expected to learn --
        # only to stay finite.
        assert all(np.isfinite(a) for a in acc)
        return
    assert acc[-1] > 0.85 and acc[-1] > acc[0], acc
