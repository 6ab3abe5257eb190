"""Epoch graph (hipGraph replay of one recorded epoch, include/dorylus_hip.h): the replayed
epochs must leave exactly the bits the eager epochs leave -- same kernels, same order, the
only difference being where Adam's step size comes from -- and a launch-bound epoch
(Cora-sized graph, BASELINE.json configs[0] shape) must not get slower."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(da, gnn, dims, heads=None, V=2708, E=5278, seed=5):
    import partition_oracle as po
    rng = np.random.default_rng(seed)
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    s, d = np.concatenate([s, d]), np.concatenate([d, s])
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = da.Context(0)
    ctx.configure(gnn, dims, V)
    if heads:
        ctx.gatmh_heads(heads)
    ctx.graph_upload(g)
    ctx.preallocate()
    ctx.fill_uniform(0, "x" if gnn == da.GCN else "h", 3, -1.0, 1.0, g["localToGlobal"])
    ctx.labels_upload(rng.integers(0, dims[-1], V).astype(np.uint32))
    ctx.weights_init_xavier()
    ctx.adam_config(0.01)
    return ctx


def _state(ctx, da, gnn, L):
    out = {}
    names = {da.GCN: ["z", "h", "grad", "aTg", "g"], da.GAT: ["z", "ah", "h", "grad", "aTg"],
             da.GATMH: ["z", "o", "dz", "el", "t"]}[gnn]
    for l in range(L):
        out[("w", l)] = ctx.weight_get(l, "w")
        out[("dw", l)] = ctx.weight_grad_get(l, "w")
        for nm in names:
            try:
                out[(nm, l)] = ctx.download(l, nm)
            except da.DoryError:
                pass
    return out


@pytest.mark.parametrize("which", ["gcn", "gcn_tf", "gat", "gatmh"])
def test_replayed_epochs_are_bit_identical_to_eager(which):
    import dorylus_amd as da
    gnn = {"gcn": da.GCN, "gcn_tf": da.GCN, "gat": da.GAT, "gatmh": da.GATMH}[which]
    dims = [1433, 16, 7] if which != "gatmh" else [1433, 32, 7]
    heads = [4, 1] if which == "gatmh" else None
    EPOCHS = 6
    states, times = [], []
    for graph in (0, 1):
        ctx = _setup(da, gnn, dims, heads)
        if which == "gcn_tf":
            ctx.set_option("gcn_transform_first", 2)      # the recorded epoch follows the transform-first stage order
        ctx.set_option("epoch_graph", graph)
        eng = da.NativeEngine(ctx)
        ms = eng.run(EPOCHS)          # graph: 1 eager epoch, then record once and replay 5 times
        ms2 = eng.run(4)              # a second call keeps replaying (and keeps Adam's iteration count going)
        times.append(np.concatenate([ms, ms2]))
        states.append(_state(ctx, da, gnn, 2))
        if graph:                     # switching the option off drops the recording and goes back to eager
            ctx.set_option("epoch_graph", 0)
            eng.run(1)
        eng.close()
        ctx.close()
    assert states[0].keys() == states[1].keys() and len(states[0]) >= 8
    for k in states[0]:
        assert np.array_equal(states[0][k], states[1][k]), k
    eager, replay = np.median(times[0][2:]), np.median(times[1][2:])
    print(f"\n{which}: eager {eager*1e3:.0f} us/epoch, replayed {replay*1e3:.0f} us/epoch")
    assert replay < eager * 1.25


def test_epoch_graph_manual_api_and_errors():
    import dorylus_amd as da
    ctx = _setup(da, da.GCN, [64, 16, 4], V=300, E=900)
    eng = da.NativeEngine(ctx)
    # recording before any eager epoch: lazily sized buffers are missing -> clean error, not a stuck stream
    ctx.set_option("epoch_graph", 0)
    with pytest.raises(da.DoryError):
        ctx.epoch_graph_launch(1)     # nothing recorded
    eng.run(1)
    w_before = ctx.weight_get(0, "w")
    ctx.epoch_graph_begin()
    with pytest.raises(da.DoryError):
        ctx.epoch_graph_begin()       # already recording
    # the calls of one GCN epoch, recorded by hand (Engine::runEpoch order)
    ctx.aggregate(0, da.FORWARD); ctx.apply_vertex(0, da.FORWARD)
    ctx.aggregate(1, da.FORWARD); ctx.apply_vertex(1, da.FORWARD); ctx.weight_update(1)
    ctx.aggregate(1, da.BACKWARD); ctx.apply_vertex(0, da.BACKWARD); ctx.weight_update(0)
    ctx.epoch_graph_end()
    assert np.array_equal(ctx.weight_get(0, "w"), w_before)     # recording executed nothing
    ctx.epoch_graph_launch(3)
    ctx.sync()
    w_graph = ctx.weight_get(0, "w")
    ctx.epoch_graph_drop()
    eng.close()
    ctx.close()
    # same four epochs eagerly
    ctx = _setup(da, da.GCN, [64, 16, 4], V=300, E=900)
    eng = da.NativeEngine(ctx)
    eng.run(4)
    assert np.array_equal(ctx.weight_get(0, "w"), w_graph)
    eng.close()
    ctx.close()
