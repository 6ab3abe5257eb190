"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs.  Tolerance: 1e-4 relative on fp32 activations
(BASELINE.json north_star); index arrays are consumed bit-exact."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def da():
    import dorylus_amd
    return dorylus_amd


def _golden_partitions(name):
    import partition_oracle as po
    d = os.path.join(ROOT, "tests", "golden", name)
    bins = sorted(glob.glob(os.path.join(d, "graph.*.bin")), key=lambda p: int(p.split(".")[-2]))
    gs = [po.parse_graph_bin(open(b, "rb").read()) for b in bins]
    parts = np.loadtxt(os.path.join(d, "graph.bsnap.parts"), dtype=np.int64, ndmin=1)
    return gs, parts


@pytest.mark.parametrize("F", [7, 16, 41, 128, 602, 1433])
@pytest.mark.parametrize("case", ["parts_toy60_p1", "parts_toy60_p2", "parts_toy97_p8_und", "parts_toy40_p3_empty"])
def test_aggregate_gcn_forward_backward_vs_oracle(da, case, F):
    """K1 on the reference-built CSC/CSR of the golden partitions, ghosts included."""
    import orc
    from helpers import make_ctx, rel_err
    gs, parts = _golden_partitions(case)
    rng = np.random.default_rng(F)
    for r, g in enumerate(gs):
        N = g["localVtxCnt"]
        ctx = make_ctx(da, g, [F, F, 3], g["globalVtxCnt"], node_id=r, num_nodes=len(gs))
        x = rng.standard_normal((N, F)).astype(np.float32)
        fg = rng.standard_normal((g["srcGhostCnt"], F)).astype(np.float32)
        gr = rng.standard_normal((N, F)).astype(np.float32)
        bg = rng.standard_normal((g["dstGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "x", x)
        ctx.upload(0, "fg", fg)
        ctx.upload(1, "grad", gr)
        ctx.upload(0, "bg", bg)
        ctx.set_option("spmm_variant", 0)
        # interior rows first, boundary rows in a second launch (what K1 does under an exchange in flight): same bits
        ctx.set_option("spmm_blk_force_split", 1)
        ctx.aggregate(0, da.FORWARD)
        ctx.aggregate(1, da.BACKWARD)
        split_out = (ctx.download(0, "ah"), ctx.download(0, "aTg"))
        ctx.set_option("spmm_blk_force_split", 0)
        for order in (2, 1, 0):
            ctx.set_option("spmm_order", order)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            ah = ctx.download(0, "ah")
            aTg = ctx.download(0, "aTg")
            assert np.array_equal(ah, split_out[0]) and np.array_equal(aTg, split_out[1])
            ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x, fg)
            ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr, bg)
            assert rel_err(ah, ref_f) < 1e-5, (case, r, F, order)
            assert rel_err(aTg, ref_b) < 1e-5, (case, r, F, order)
        ctx.close()


@pytest.mark.parametrize("F", [16, 41, 128, 602, 1433])
@pytest.mark.parametrize("case", ["parts_toy60_p1", "parts_toy60_p2", "parts_toy97_p8_und", "parts_toy40_p3_empty"])
def test_aggregate_blocked_variant_vs_oracle(da, case, F):
    """K1b (source-blocked, XCD-aware) == oracle, forward (CSC) and backward (CSR), ghosts included."""
    import orc
    from helpers import make_ctx, rel_err
    gs, parts = _golden_partitions(case)
    rng = np.random.default_rng(F + 1)
    for r, g in enumerate(gs):
        N = g["localVtxCnt"]
        ctx = make_ctx(da, g, [F, F, 3], g["globalVtxCnt"], node_id=r, num_nodes=len(gs))
        ctx.set_option("spmm_variant", 1)
        x = rng.standard_normal((N, F)).astype(np.float32)
        fg = rng.standard_normal((g["srcGhostCnt"], F)).astype(np.float32)
        gr = rng.standard_normal((N, F)).astype(np.float32)
        bg = rng.standard_normal((g["dstGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "x", x); ctx.upload(0, "fg", fg); ctx.upload(1, "grad", gr); ctx.upload(0, "bg", bg)
        ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x, fg)
        ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr, bg)
        for grp, split in ((32, 0), (16, 0), (16, 1), (32, 1)):   # split: local-source blocks launched apart
            ctx.set_option("spmm_blk_group", grp)
            ctx.set_option("spmm_blk_force_split", split)
            ctx.set_option("spmm_blk_nb", 8 * split)        # toy graphs: several blocks so that some are local-only
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            assert rel_err(ctx.download(0, "ah"), ref_f) < 1e-5, (case, r, F, grp)
            assert rel_err(ctx.download(0, "aTg"), ref_b) < 1e-5, (case, r, F, grp)
        ctx.close()


@pytest.mark.parametrize("F", [41, 128, 602, 1433])
@pytest.mark.parametrize("case", ["parts_toy60_p1", "parts_toy60_p2", "parts_toy97_p8_und", "parts_toy40_p3_empty"])
def test_aggregate_sweep_variant_vs_oracle(da, case, F):
    """K1s (register-accumulating sweep over the blocked adjacency, gated per XCD) == oracle on the reference-built
    partitions, forward (CSC) and backward (CSR), ghosts included: with ghost rows the local-source blocks and the
    ghost blocks are two launches (out += sum), with and without an exchange in flight the same bits."""
    import orc
    from helpers import make_ctx, rel_err
    gs, parts = _golden_partitions(case)
    rng = np.random.default_rng(F + 2)
    for r, g in enumerate(gs):
        N = g["localVtxCnt"]
        ctx = make_ctx(da, g, [F, F, 3], g["globalVtxCnt"], node_id=r, num_nodes=len(gs))
        ctx.set_option("spmm_variant", 2)
        x = rng.standard_normal((N, F)).astype(np.float32)
        fg = rng.standard_normal((g["srcGhostCnt"], F)).astype(np.float32)
        gr = rng.standard_normal((N, F)).astype(np.float32)
        bg = rng.standard_normal((g["dstGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "x", x); ctx.upload(0, "fg", fg); ctx.upload(1, "grad", gr); ctx.upload(0, "bg", bg)
        ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x, fg)
        ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr, bg)
        prev = None
        for grp, nb in ((32, 8), (16, 8), (32, 16), (16, 24)):   # toy graphs: explicit block counts (else K1 takes them)
            ctx.set_option("spmm_blk_group", grp)
            ctx.set_option("spmm_blk_nb", nb)
            for split in (0, 1):
                ctx.set_option("spmm_blk_force_split", split)
                ctx.aggregate(0, da.FORWARD)
                ctx.aggregate(1, da.BACKWARD)
                got = (ctx.download(0, "ah"), ctx.download(0, "aTg"))
                assert rel_err(got[0], ref_f) < 1e-5, (case, r, F, grp, nb)
                assert rel_err(got[1], ref_b) < 1e-5, (case, r, F, grp, nb)
                if split:
                    assert np.array_equal(got[0], prev[0]) and np.array_equal(got[1], prev[1])
                prev = got
        ctx.close()


def test_sweep_variant_several_sweeps_and_unit_weights(da):
    """K1s on a graph large enough for several sweeps per slab (more workgroup tasks per XCD than CUs), a ragged last
    sweep, rows that end inside a workgroup, empty rows, F = 128 and 602; and the unit-weight form with a
    per-destination factor (the reference GAT's aggregation) against the explicit per-edge values through K1."""
    import orc
    import partition_oracle as po
    from helpers import make_ctx, rel_err
    rng = np.random.default_rng(31)
    V, E = 90001, 1200000
    s = rng.integers(0, V, E)
    d = rng.integers(0, V - 700, E)          # the last 700 vertices have no in-edges
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    for F in (128, 602):
        ctx = make_ctx(da, g, [F, F, 3], V)
        x = rng.standard_normal((V, F)).astype(np.float32)
        gr = rng.standard_normal((V, F)).astype(np.float32)
        ctx.upload(0, "x", x)
        ctx.upload(1, "grad", gr)
        ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x)
        ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr)
        outs = {}
        for variant, nb in ((2, 0), (2, 40), (1, 0)):
            ctx.set_option("spmm_variant", variant)
            ctx.set_option("spmm_blk_nb", nb)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            outs[(variant, nb)] = (ctx.download(0, "ah"), ctx.download(0, "aTg"))
            assert rel_err(outs[(variant, nb)][0], ref_f) < 1e-5, (F, variant, nb)
            assert rel_err(outs[(variant, nb)][1], ref_b) < 1e-5, (F, variant, nb)
        # two rows of a lane group as one stream of entries (the default from three slabs on) and the plain row loop,
        # forced either way at both widths: the additions inside a row keep their order -> the same bits
        ctx.set_option("spmm_variant", 2)
        base = None
        for pair in (-1, 0, 1):
            ctx.set_option("spmm_sweep_pair", pair)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            got = (ctx.download(0, "ah"), ctx.download(0, "aTg"))
            if base is None:
                base = got
                assert rel_err(got[0], ref_f) < 1e-5 and rel_err(got[1], ref_b) < 1e-5
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), (F, pair)
        ctx.set_option("spmm_sweep_pair", -1)
        ctx.close()
    # reference GAT prototype, whole epoch on this graph: the unit-weight sweep with row factors (default) AND the general
    # K1 path, each against the C oracle's epoch (not against each other)
    from helpers import oracle_gat_epoch
    dims = [64, 128, 16]
    H0 = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = (np.arange(V) % dims[-1]).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(2)]
    As = [(rng.standard_normal((dims[i + 1], 1)) / 2).astype(np.float32) for i in range(2)]
    T, dWs, _ = oracle_gat_epoch(g, H0, labels, Ws, As)
    for variant in (2, 0):
        ctx = make_ctx(da, g, dims, V, gnn=da.GAT)
        ctx.set_option("spmm_variant", variant)
        ctx.upload(0, "h", H0)
        ctx.labels_upload(labels)
        for l in range(2):
            ctx.weight_set(l, "w", Ws[l])
            ctx.weight_set(l, "a_i", As[l])
        ctx.adam_config(0.01)
        eng = da.NativeEngine(ctx)
        eng.run(1)
        for l in range(2):
            for nm in ("z", "ah", "aTg"):
                got = ctx.download(l, nm)
                assert np.isfinite(got).all(), (variant, nm, l)
                # aTg: unnormalised edge weights (the prototype has no softmax) make it a sum of large terms of both signs
                assert rel_err(got, T[f"{nm}{l}"]) < (RTOL if nm != "aTg" else 1e-3), (variant, nm, l, rel_err(got, T[f"{nm}{l}"]))
            assert rel_err(ctx.weight_grad_get(l), dWs[l]) < 1e-3, (variant, l)
        eng.close()
        ctx.close()


@pytest.mark.parametrize("F", [41, 602])
def test_hub_partition_built_by_the_reference(da, F):
    """tests/golden/parts_hub3000_p2: graph.<id>.bin bytes written by the reference's own DataLoader, with one
    destination of 11 000 in-edges and one source of 10 000 out-edges (beyond K1's 8 192-edge row clamp and the blocked
    kernels' 2 048-edge segment clamp), two partitions with ghosts: K1 + long-row kernels, K1b + long-segment kernels,
    and the default variant (which hands hub graphs to K1b) against the oracle, forward and backward."""
    import orc
    from helpers import make_ctx, rel_err
    gs, parts = _golden_partitions("parts_hub3000_p2")
    assert max(np.diff(g["colPtr"].astype(np.int64)).max() for g in gs) > 8192
    assert max(np.diff(g["rowPtr"].astype(np.int64)).max() for g in gs) > 8192
    rng = np.random.default_rng(F)
    for r, g in enumerate(gs):
        N = g["localVtxCnt"]
        ctx = make_ctx(da, g, [F, F, 3], g["globalVtxCnt"], node_id=r, num_nodes=len(gs))
        x = rng.standard_normal((N, F)).astype(np.float32)
        fg = rng.standard_normal((g["srcGhostCnt"], F)).astype(np.float32)
        gr = rng.standard_normal((N, F)).astype(np.float32)
        bg = rng.standard_normal((g["dstGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "x", x); ctx.upload(0, "fg", fg); ctx.upload(1, "grad", gr); ctx.upload(0, "bg", bg)
        ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x, fg)
        ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr, bg)
        for variant, nb in ((0, 0), (1, 8), (2, 8), (2, 16)):
            ctx.set_option("spmm_variant", variant)
            ctx.set_option("spmm_blk_nb", nb)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            # 10 000-term sums in chunk / block order instead of edge order: the path's 1e-4 bar
            assert rel_err(ctx.download(0, "ah"), ref_f) < RTOL, (r, F, variant, nb)
            assert rel_err(ctx.download(0, "aTg"), ref_b) < RTOL, (r, F, variant, nb)
        ctx.close()


def test_blocked_variant_many_blocks(da):
    """enough source rows for several rounds of 8 blocks (nb = 16+), skewed degrees."""
    import orc
    import partition_oracle as po
    from helpers import make_ctx, rel_err
    rng = np.random.default_rng(9)
    V, E, F = 40000, 400000, 64
    s = rng.integers(0, V, E)
    s[: E // 8] = rng.integers(0, 50, E // 8)                          # sources concentrated in block 0
    d = np.where(rng.random(E) < 0.05, 7, rng.integers(0, V, E))      # one hub destination (> 2048 edges per block)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = make_ctx(da, g, [F, 8, 3], V)
    x = rng.standard_normal((V, F)).astype(np.float32)
    ctx.upload(0, "x", x)
    ref = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x)
    for variant, grp, nb in ((0, 32, 0), (1, 32, 0), (1, 16, 24), (1, 8, 8), (1, 16, 0), (1, 32, 24)):
        ctx.set_option("spmm_variant", variant)
        ctx.set_option("spmm_blk_group", grp)
        ctx.set_option("spmm_blk_nb", nb)
        ctx.aggregate(0, da.FORWARD)
        assert rel_err(ctx.download(0, "ah"), ref) < 1e-5, (variant, grp, nb)
    ctx.close()


def test_k1_hub_rows_are_split(da):
    """K1 (row gather) with rows far beyond LONG_ROW_CLAMP = 8192 edges: the clamped main launch plus the
    workgroup-per-chunk kernels give the oracle's aggregate, forward and backward, several widths."""
    import orc
    import partition_oracle as po
    from helpers import make_ctx, rel_err
    rng = np.random.default_rng(21)
    V, E = 3000, 120000
    s = rng.integers(0, V, E)
    d = rng.integers(0, V, E)
    d[:40000] = 7            # hub destination: in-degree ~40k (5 chunks beyond the clamp, last one ragged)
    s[40000:70000] = 11      # hub source: out-degree ~30k
    d[70000:79000] = 13      # just above the clamp: one short chunk
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    assert np.diff(g["colPtr"].astype(np.int64)).max() > 8192 and np.diff(g["rowPtr"].astype(np.int64)).max() > 8192
    for F in (16, 128, 602):
        ctx = make_ctx(da, g, [F, F, 3], V)
        ctx.set_option("spmm_variant", 0)
        x = rng.standard_normal((V, F)).astype(np.float32)
        gr = rng.standard_normal((V, F)).astype(np.float32)
        ctx.upload(0, "x", x)
        ctx.upload(1, "grad", gr)
        for slab in (0, 64):
            ctx.set_option("spmm_slab", slab)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x)
            ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr)
            # 30-40k-term sums in chunk order instead of edge order: the 1e-4 bar of the path, not the 1e-5 of short rows
            assert rel_err(ctx.download(0, "ah"), ref_f) < RTOL, (F, slab)
            assert rel_err(ctx.download(0, "aTg"), ref_b) < RTOL, (F, slab)
        ctx.close()


@pytest.mark.parametrize("slab", [0, 32, 64, 128, 256])
def test_aggregate_feature_slabs(da, slab):
    import orc
    from helpers import make_ctx, rel_err
    gs, _ = _golden_partitions("parts_toy60_p2")
    g = gs[0]
    F = 602
    rng = np.random.default_rng(1)
    ctx = make_ctx(da, g, [F, 8, 3], g["globalVtxCnt"], node_id=0, num_nodes=2)
    x = rng.standard_normal((g["localVtxCnt"], F)).astype(np.float32)
    fg = rng.standard_normal((g["srcGhostCnt"], F)).astype(np.float32)
    ctx.upload(0, "x", x)
    ctx.upload(0, "fg", fg)
    ctx.set_option("spmm_variant", 0)
    ctx.set_option("spmm_slab", slab)
    ctx.aggregate(0, da.FORWARD)
    ref = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x, fg)
    assert rel_err(ctx.download(0, "ah"), ref) < 1e-5
    ctx.close()


def _run_gcn_epochs(da, gs, parts, dims, X, labels, Ws, epochs=1, lr=0.01, native_plan=None, transform_first=0, blk_nb=0):
    """P partitions as P contexts on one GPU; the transport between them is a host
    copy of the packed buffers (pack/unpack kernels + plan are the code under test)."""
    import torch
    from halo_plan_ref import halo_plan
    from helpers import make_ctx
    P = len(gs)
    V = int(gs[0]["globalVtxCnt"])
    ctxs, engs, plans = [], [], []
    for r, g in enumerate(gs):
        ctx = make_ctx(da, g, dims, V, node_id=r, num_nodes=P, options={"spmm_blk_nb": blk_nb})
        ctx.upload(0, "x", X[g["localToGlobal"]])
        ctx.upload(0, "fg", X[g["srcGhost"]].reshape(g["srcGhostCnt"], dims[0]))
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l, W in enumerate(Ws):
            ctx.weight_set(l, "w", W)
        ctx.adam_config(lr)
        ctx.set_option("gcn_transform_first", int(transform_first))
        pl = halo_plan(g, parts, r, P)
        if native_plan is not None:
            # the C++ path bench.py / graphserver use: dory_partition_upload computes both plans
            native_plan[r].upload(ctx, parts)
        else:
            for d in (0, 1):
                ctx.halo_plan(d, pl[d][0], pl[d][1])
        ctxs.append(ctx)
        plans.append(pl)
    L = len(dims) - 1

    def exchange(layer, d):
        # widths travel padded (ld); transport = device-to-device copies by torch
        tfl = ctxs[0].transform_first_layer(layer)
        if d == 0:
            src_name, src_layer = ("xw", layer) if tfl else ("h", layer - 1)
        else:
            src_name, src_layer = ("g", layer) if tfl else ("grad", layer)
        _, _, ld, _ = ctxs[0].info(src_layer, src_name)
        send = [torch.zeros(max(1, sum(len(x) for x in plans[r][d][0])) * ld, device="cuda") for r in range(P)]
        recv = [torch.zeros(max(1, sum(len(x) for x in plans[r][d][1])) * ld, device="cuda") for r in range(P)]
        torch.cuda.synchronize()   # (the contexts' streams are non-blocking: torch's zero fills must have landed before a pack kernel writes)
        for r in range(P):
            ctxs[r].halo_pack(layer, d, send[r].data_ptr())
            ctxs[r].sync()
        for r in range(P):
            soff = np.concatenate([[0], np.cumsum([len(x) for x in plans[r][d][0]])])
            for p in range(P):
                roff = np.concatenate([[0], np.cumsum([len(x) for x in plans[p][d][1]])])
                n = len(plans[r][d][0][p])
                assert n == len(plans[p][d][1][r])
                recv[p][roff[r] * ld:(roff[r] + n) * ld] = send[r][soff[p] * ld:(soff[p] + n) * ld]
        torch.cuda.synchronize()
        for r in range(P):
            ctxs[r].halo_unpack(layer, d, recv[r].data_ptr())
            ctxs[r].sync()

    stats = []
    for ep in range(epochs):
        # stage order of the reference epoch
        for l in range(L):
            if l > 0:
                exchange(l, da.FORWARD)
            for c in ctxs:
                c.aggregate(l, da.FORWARD)
                c.apply_vertex(l, da.FORWARD)
        stats.append([c.train_stat() for c in ctxs])
        dWs = {}
        for l in range(L - 1, 0, -1):
            exchange(l, da.BACKWARD)
            for c in ctxs:
                c.aggregate(l, da.BACKWARD)
                c.apply_vertex(l - 1, da.BACKWARD)
        if ctxs[0].transform_first_layer(0):
            exchange(0, da.BACKWARD)              # ghost rows of g0
            for c in ctxs:
                c.aggregate(0, da.BACKWARD)       # dW0 = X^T (A^T g0)
        for l in range(L):                        # every gradient exists now, whichever stage produced it
            dWs[l] = sum(c.weight_grad_get(l) for c in ctxs)
        if epochs > 1:
            for c in ctxs:
                for l in range(L - 1, -1, -1):
                    c.weight_update(l)
    return ctxs, dWs, stats


@pytest.mark.parametrize("case,dims", [
    ("parts_toy60_p1", [13, 8, 5]),
    ("parts_toy60_p2", [602, 128, 41]),
    ("parts_toy60_p4_hash", [33, 16, 7]),
    ("parts_toy97_p8_und", [20, 12, 9, 4]),
    ("parts_toy97_p8_und", [300, 64, 64, 25]),     # BASELINE config 4 shape: Amazon GCN 3-layer, 8 partitions
    ("parts_toy97_p8_und", [256, 48, 51]),         # BASELINE config 5 shape: Friendster GCN 2-layer, 8 partitions
])
@pytest.mark.parametrize("blk_nb", [0, 8])   # 8: the source-blocked K1b kernels even on these L2-sized graphs
def test_gcn_epoch_vs_oracle(da, case, dims, blk_nb):
    """Whole forward+backward epoch, every named tensor, P partitions with halo."""
    from helpers import assert_parity, oracle_gcn_epoch, rel_err
    gs, parts = _golden_partitions(case)
    V = int(gs[0]["globalVtxCnt"])
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32)
          for i in range(len(dims) - 1)]
    ctxs, dWs, stats = _run_gcn_epochs(da, gs, parts, dims, X, labels, Ws, blk_nb=blk_nb)
    T, dW = oracle_gcn_epoch(gs, parts, X, labels, Ws, V)
    L = len(dims) - 1
    for r, c in enumerate(ctxs):
        if gs[r]["localVtxCnt"] == 0:
            continue
        for l in range(L):
            assert_parity(c.download(l, "ah"), T[r][f"ah{l}"], (r, l, "ah"))
            if l < L - 1:
                assert_parity(c.download(l, "z"), T[r][f"z{l}"], 'c.download(l')
                assert_parity(c.download(l, "h"), T[r][f"h{l}"], 'c.download(l')
                assert_parity(c.download(l, "aTg"), T[r][f"aTg{l}"], 'c.download(l')
                assert_parity(c.download(l, "g"), T[r][f"g{l}"], 'c.download(l')
            if l > 0:
                assert_parity(c.download(l, "grad"), T[r][f"grad{l}"], (r, l, "grad"))
                assert_parity(c.download(l, "fg"), T[r][f"fg{l}"], 'c.download(l')
                assert_parity(c.download(l - 1, "bg"), T[r][f"bg{l-1}"], 'c.download(l - 1')
        assert_parity(c.download(L - 1, "g"), T[r]["d"], 'c.download(L - 1')
        a, lo, n = stats[0][r]
        assert abs(a - T[r]["acc"]) < 1e-3 and abs(lo - T[r]["loss"]) < 1e-3 * max(1.0, abs(T[r]["loss"]))
    for l in range(L):
        assert_parity(dWs[l], dW[l], ("dW", l))
    # halo rows are bit-exact copies of the owner's rows (fg[slot(gvid)] == owner.h[lvid(gvid)])
    owner_row = {}
    for r, g in enumerate(gs):
        for lv, gv in enumerate(g["localToGlobal"]):
            owner_row[int(gv)] = (r, lv)
    for l in range(1, L):
        hs = [c.download(l - 1, "h") for c in ctxs]
        grs = [c.download(l, "grad") for c in ctxs]
        for r, c in enumerate(ctxs):
            fg, bg = c.download(l, "fg"), c.download(l - 1, "bg")
            for k, gv in enumerate(gs[r]["srcGhost"]):
                o, lv = owner_row[int(gv)]
                assert np.array_equal(fg[k], hs[o][lv])
            for k, gv in enumerate(gs[r]["dstGhost"]):
                o, lv = owner_row[int(gv)]
                assert np.array_equal(bg[k], grs[o][lv])
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("case,dims", [
    ("parts_toy60_p1", [13, 8, 5]),
    ("parts_toy60_p2", [602, 128, 41]),
    ("parts_toy60_p4_hash", [33, 16, 7]),
    ("parts_toy60_p4_hash", [300, 64, 64, 25]),      # 3 layers: only layer 0 changes order
    ("parts_toy40_p3_empty", [20, 12, 4]),
])
@pytest.mark.parametrize("mode", [1, 2])    # 1: layer 0 only, 2: every layer that narrows
def test_gcn_epoch_transform_first_vs_oracle(da, case, dims, mode):
    """Option gcn_transform_first: z0 = A(X W0), dW0 = X^T(A^T g0) -- every tensor the reference order also
    produces (all but ah0) and the summed weight gradients against the same oracle epoch, P partitions."""
    from helpers import oracle_gcn_epoch, rel_err
    gs, parts = _golden_partitions(case)
    V = int(gs[0]["globalVtxCnt"])
    rng = np.random.default_rng(7)
    X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32)
          for i in range(len(dims) - 1)]
    ctxs, dWs, stats = _run_gcn_epochs(da, gs, parts, dims, X, labels, Ws, transform_first=mode, blk_nb=8 if mode == 2 else 0)
    assert ctxs[0].transform_first_active()
    tfl = [ctxs[0].transform_first_layer(l) for l in range(len(dims) - 1)]
    assert tfl[0] and (mode == 2 or not any(tfl[1:]))
    assert tfl == [mode == 2 and dims[l] > dims[l + 1] or (l == 0) for l in range(len(dims) - 1)]
    T, dW = oracle_gcn_epoch(gs, parts, X, labels, Ws, V)
    L = len(dims) - 1
    for r, c in enumerate(ctxs):
        if gs[r]["localVtxCnt"] == 0:
            continue
        for l in range(L):
            if not tfl[l]:
                assert rel_err(c.download(l, "ah"), T[r][f"ah{l}"]) < RTOL, (r, l, "ah")
            if l < L - 1:
                assert rel_err(c.download(l, "z"), T[r][f"z{l}"]) < RTOL, (r, l, "z")
                assert rel_err(c.download(l, "h"), T[r][f"h{l}"]) < RTOL
                assert rel_err(c.download(l, "aTg"), T[r][f"aTg{l}"]) < RTOL
                assert rel_err(c.download(l, "g"), T[r][f"g{l}"]) < RTOL
        assert rel_err(c.download(L - 1, "g"), T[r]["d"]) < RTOL
    for l in range(L):
        assert rel_err(dWs[l], dW[l]) < RTOL, ("dW", l)
    for c in ctxs:
        c.close()


def test_transform_first_engine_epochs_match_reference_order(da):
    """dory_engine_run in both orders: same weights after 3 epochs (within fp32 rounding), and the mode is
    ignored when the first hidden layer is not narrower than the input."""
    from helpers import rel_err
    import partition_oracle as po
    rng = np.random.default_rng(3)
    V, E = 500, 6000
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    out = {}
    for tf in (0, 1, 2):
        ctx = da.Context(0)
        ctx.configure(da.GCN, [96, 32, 6], V)
        ctx.set_option("gcn_transform_first", tf)
        ctx.graph_upload(g)
        ctx.preallocate()
        ctx.fill_uniform(0, "x", 3, -1.0, 1.0, g["localToGlobal"])
        ctx.labels_upload(rng.integers(0, 6, V).astype(np.uint32) * 0 + (np.arange(V) % 6).astype(np.uint32))
        ctx.weights_init_xavier()
        ctx.adam_config(0.01)
        assert ctx.transform_first_active() == bool(tf)
        eng = da.NativeEngine(ctx)
        eng.run(3)
        out[tf] = [ctx.weight_get(l, "w") for l in range(2)] + [ctx.download(0, "h")]
        eng.close()
        ctx.close()
    for tf in (1, 2):
        for a, b in zip(out[0], out[tf]):
            assert rel_err(a, b) < 1e-4, tf
    ctx = da.Context(0)
    ctx.configure(da.GCN, [16, 32, 6], V)
    ctx.set_option("gcn_transform_first", 1)
    assert not ctx.transform_first_active()      # 16 -> 32: aggregating first is already the narrow side
    ctx.close()
    # partitions built with undirected = 1 carry the reference's ghost-degree quirk: csrVal != cscVal^T,
    # dory_partition_upload marks them and the mode stays off
    for und, want in ((0, True), (1, False)):
        part = da.Partition.build(s.astype(np.uint32), d.astype(np.uint32), np.zeros(V, np.int32), 0, 1, undirected=und)
        ctx = da.Context(0)
        ctx.configure(da.GCN, [96, 32, 6], V)
        ctx.set_option("gcn_transform_first", 1)
        part.upload(ctx)
        assert ctx.transform_first_active() == want
        ctx.close()


def test_gcn_epoch_native_partition_and_plan(da):
    """Same epoch check, but partitions + halo plans come from the C++ host layer
    (Partition.build -> dory_partition_upload), the path bench.py takes at N > 1."""
    from helpers import oracle_gcn_epoch, random_graph, rel_err
    V, P, dims = 500, 4, [40, 24, 6]
    s, d = random_graph(21, V, 6000)
    parts = (np.arange(V, dtype=np.int64) * P // V).astype(np.int32)
    nparts = [da.Partition.build(s, d, parts, r, P) for r in range(P)]
    gs = [p.view() for p in nparts]
    rng = np.random.default_rng(3)
    X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(2)]
    ctxs, dWs, stats = _run_gcn_epochs(da, gs, parts, dims, X, labels, Ws, native_plan=nparts)
    T, dW = oracle_gcn_epoch(gs, parts, X, labels, Ws, V)
    for r, c in enumerate(ctxs):
        assert rel_err(c.download(1, "ah"), T[r]["ah1"]) < RTOL
        assert rel_err(c.download(0, "aTg"), T[r]["aTg0"]) < RTOL
        assert np.array_equal(c.download(1, "fg"), np.concatenate(
            [ctxs[parts[gv]].download(0, "h")[np.searchsorted(gs[parts[gv]]["localToGlobal"], gv)][None]
             for gv in gs[r]["srcGhost"]] + [np.zeros((0, dims[1]), np.float32)]))
    for l in range(2):
        assert rel_err(dWs[l], dW[l]) < RTOL
    for c in ctxs:
        c.close()


def test_gcn_numpy_gnn_fixture(da, golden_dir):
    """Against the reference's own Python GCN (fixture from miscs/numpy-gnn)."""
    import partition_oracle as po
    from helpers import assert_parity, make_ctx, rel_err
    z = np.load(os.path.join(golden_dir, "numpy_gnn_epoch.npz"))
    V = int(z["V"])
    g = po.preprocess(z["src"].astype(np.uint32), z["dst"].astype(np.uint32), np.zeros(V, np.int64), 0, 1)
    dims = [z["X"].shape[1], z["W0"].shape[1], z["W1"].shape[1]]
    ctx = make_ctx(da, g, dims, V)
    ctx.upload(0, "x", z["X"])
    ctx.weight_set(0, "w", z["W0"])
    ctx.weight_set(1, "w", z["W1"])
    ctx.labels_upload(z["labels"].astype(np.uint32))
    ctx.aggregate(0, da.FORWARD)
    ctx.apply_vertex(0, da.FORWARD)
    ctx.aggregate(1, da.FORWARD)
    assert_parity(ctx.download(0, "ah"), z["ah0"], 'ctx.download(0')
    assert_parity(ctx.download(0, "z"), z["z0"], 'ctx.download(0')
    assert_parity(ctx.download(0, "h"), z["h0"], 'ctx.download(0')
    assert_parity(ctx.download(1, "ah"), z["ah1"], 'ctx.download(1')
    ctx.apply_vertex(1, da.FORWARD)
    assert_parity(ctx.download(1, "z"), z["z1"], 'ctx.download(1')
    ctx.close()


@pytest.mark.parametrize("name", ["numpy_gnn_reddit_dims", "numpy_gnn_amazon_dims", "numpy_gnn_hub4k", "numpy_gnn_20k"])
def test_gcn_numpy_gnn_fixture_baseline_widths(da, golden_dir, name):
    """The reference's Python GCN at the widths of BASELINE configs 2 and 4 (602-128-41 / 300-64-64-25, 1 500 vertices):
    the forward half of the epoch through the C-ABI on the fixture's sampled rows."""
    import partition_oracle as po
    from helpers import assert_parity, make_ctx, rel_err
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    V = int(z["V"])
    dims = [int(x) for x in z["dims"]]
    L = len(dims) - 1
    g = po.preprocess(z["src"].astype(np.uint32), z["dst"].astype(np.uint32), np.zeros(V, np.int64), 0, 1)
    rows = z["sample"]
    for variant, nb in ((2, 8), (0, 0)):          # the sweep kernel (forced on this L2-sized graph) and the row gather
        ctx = make_ctx(da, g, dims, V)
        ctx.set_option("spmm_variant", variant)
        ctx.set_option("spmm_blk_nb", nb)
        ctx.upload(0, "x", z["X_q64"].astype(np.float32) / np.float32(64))
        for l in range(L):
            ctx.weight_set(l, "w", z[f"W{l}"])
        ctx.labels_upload(z["labels"].astype(np.uint32))
        for l in range(L):
            ctx.aggregate(l, da.FORWARD)
            ctx.apply_vertex(l, da.FORWARD)
        # forward tensors only: the reference's C++ last layer differs from its Python model behind z_L (maskout of the
        # non-training rows and the 1/(V*0.66) scale, CPU_comm.cpp:126-146) -- the backward half at these widths is
        # checked against the C oracle by test_gcn_epoch_vs_oracle, and the oracle against this fixture on the CPU
        for l in range(L):
            assert_parity(ctx.download(l, "ah")[rows], z[f"ah{l}"], (name, variant, l))
            assert_parity(ctx.download(l, "z")[rows], z[f"z{l}"], (name, variant, l))
            if l < L - 1:
                assert_parity(ctx.download(l, "h")[rows], z[f"h{l}"], 'ctx.download(l')
        ctx.close()


def test_gcn_numpy_gnn_fixture_backward_half(da, golden_dir):
    """Backward half against the reference's Python GCN directly (not via the C oracle): the fixture's grad1 is uploaded
    as "grad"@1, then GA bwd -> aTg0 and AV bwd -> g0, dW0 come from the HIP path and are compared with the fixture's own
    numbers.  dW1 / grad1 of the C++ last layer differ from the Python model by CPUComm's maskout (the float-count quirk,
    CPU_comm.cpp:464-471) and the 1/(V*0.66) scale: those two are checked against the fixture's d with exactly that
    mask and scale applied on the host."""
    import partition_oracle as po
    from helpers import assert_parity, make_ctx, rel_err
    z = np.load(os.path.join(golden_dir, "numpy_gnn_epoch.npz"))
    V = int(z["V"])
    g = po.preprocess(z["src"].astype(np.uint32), z["dst"].astype(np.uint32), np.zeros(V, np.int64), 0, 1)
    dims = [z["X"].shape[1], z["W0"].shape[1], z["W1"].shape[1]]
    for variant, nb in ((2, 8), (1, 8), (0, 0)):
        ctx = make_ctx(da, g, dims, V, options={"spmm_variant": variant, "spmm_blk_nb": nb})
        ctx.upload(0, "x", z["X"])
        ctx.weight_set(0, "w", z["W0"])
        ctx.weight_set(1, "w", z["W1"])
        ctx.labels_upload(z["labels"].astype(np.uint32))
        for l in range(2):
            ctx.aggregate(l, da.FORWARD)
            ctx.apply_vertex(l, da.FORWARD)
        # the C++ last layer on the fixture's d: maskout zeroes (V - stt) floats from dense offset stt*C, then / (V*0.66)
        C = dims[-1]
        stt = int(V * 0.66)
        d = z["d"].astype(np.float64).copy().reshape(-1)
        d[stt * C: stt * C + (V - stt)] = 0.0
        d = d.reshape(V, C) / np.float32(V * 0.66)
        assert_parity(ctx.download(1, "g"), d, variant)
        assert_parity(ctx.weight_grad_get(1), z["ah1"].T @ d, variant)
        assert_parity(ctx.download(1, "grad"), d @ z["W1"].astype(np.float64).T, variant)
        # backward half from the fixture's own gradient
        ctx.upload(1, "grad", z["grad1"].astype(np.float32))
        ctx.aggregate(1, da.BACKWARD)
        assert_parity(ctx.download(0, "aTg"), z["aTg0"], variant)
        ctx.apply_vertex(0, da.BACKWARD)
        assert_parity(ctx.download(0, "g"), z["g0"], variant)
        assert_parity(ctx.weight_grad_get(0), z["dW0"], variant)
        ctx.close()


@pytest.mark.parametrize("name", ["numpy_gnn_reddit_dims", "numpy_gnn_amazon_dims", "numpy_gnn_hub4k", "numpy_gnn_20k"])
def test_gcn_numpy_gnn_fixture_backward_half_baseline_widths(da, golden_dir, name):
    """The same at the widths of BASELINE configs 2 and 4 (any depth): upload the fixture's complete grad_{L-1}, run the
    backward stages of every layer below through the C-ABI, compare aTg_l, g_l, grad_l (sampled rows) and the complete
    dW_l with the reference's Python GCN."""
    import partition_oracle as po
    from helpers import assert_parity, make_ctx, rel_err
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    V = int(z["V"])
    dims = [int(x) for x in z["dims"]]
    L = len(dims) - 1
    g = po.preprocess(z["src"].astype(np.uint32), z["dst"].astype(np.uint32), np.zeros(V, np.int64), 0, 1)
    rows = z["sample"]
    for variant, nb in ((2, 8), (0, 0)):
        ctx = make_ctx(da, g, dims, V, options={"spmm_variant": variant, "spmm_blk_nb": nb})
        ctx.upload(0, "x", z["X_q64"].astype(np.float32) / np.float32(64))
        for l in range(L):
            ctx.weight_set(l, "w", z[f"W{l}"])
        ctx.labels_upload(z["labels"].astype(np.uint32))
        for l in range(L):
            ctx.aggregate(l, da.FORWARD)
            ctx.apply_vertex(l, da.FORWARD)
        ctx.upload(L - 1, "grad", z[f"grad{L-1}_full"])
        for l in range(L - 1, 0, -1):
            ctx.aggregate(l, da.BACKWARD)                       # GA bwd: aTg@(l-1) = A^T grad@l
            assert_parity(ctx.download(l - 1, "aTg")[rows], z[f"aTg{l-1}"], (name, variant, l))
            ctx.apply_vertex(l - 1, da.BACKWARD)                # AV bwd: g, dW, grad@(l-1)
            assert_parity(ctx.download(l - 1, "g")[rows], z[f"g{l-1}"], (name, variant, l))
            assert_parity(ctx.weight_grad_get(l - 1), z[f"dW{l-1}"], (name, variant, l))
            if l - 1 > 0:
                assert_parity(ctx.download(l - 1, "grad")[rows], z[f"grad{l-1}"], (name, variant, l))
        ctx.close()


def test_gcn_cache_ah0_bit_identical_and_invalidated(da):
    """Opt-in gcn_cache_ah0: three learning epochs with and without the kept layer-0 aggregate give the same bits in
    ah0, z0, every dW and every weight; re-uploading x (or refilling fg@0, or touching an option) recomputes."""
    from helpers import random_graph
    V, P, dims = 3000, 2, [96, 32, 6]
    s, d = random_graph(33, V, 40000)
    parts = (np.arange(V, dtype=np.int64) * P // V).astype(np.int32)
    part = da.Partition.build(s, d, parts, 0, P)                 # rank 0 of 2: ghost rows in fg@0 as well
    g = part.view()
    N = int(g["localVtxCnt"])
    rng = np.random.default_rng(4)
    X = rng.uniform(-1, 1, (N, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], N).astype(np.uint32)
    res = {}
    for cache in (0, 1):
        ctx = da.Context(0)
        ctx.configure(da.GCN, dims, V)
        ctx.set_option("spmm_blk_nb", 8)
        part.upload(ctx)
        ctx.preallocate()
        ctx.upload(0, "x", X)
        ctx.fill_uniform(0, "fg", 9, -1.0, 1.0, g["srcGhost"])
        ctx.labels_upload(labels)
        ctx.weights_init_xavier()
        ctx.adam_config(0.01)
        ctx.set_option("gcn_cache_ah0", cache)
        eng = da.NativeEngine(ctx)
        out = []
        for _ in range(3):
            eng.run(1)
            out.append([ctx.download(0, "ah"), ctx.download(0, "z"), ctx.weight_grad_get(0), ctx.weight_grad_get(1),
                        ctx.weight_get(0), ctx.weight_get(1)])
        assert ctx.get_option("gcn_cache_ah0_skips") == (2 if cache else 0)
        if cache:
            # invalidation: new x -> recomputed (equals a fresh aggregate), and kept again afterwards
            X2 = (X * np.float32(0.5)).astype(np.float32)
            ctx.upload(0, "x", X2)
            skips = ctx.get_option("gcn_cache_ah0_skips")
            ctx.aggregate(0, da.FORWARD)
            assert ctx.get_option("gcn_cache_ah0_skips") == skips
            a2 = ctx.download(0, "ah")
            ctx.set_option("gcn_cache_ah0", 0)
            ctx.aggregate(0, da.FORWARD)
            assert np.array_equal(ctx.download(0, "ah"), a2) and not np.array_equal(a2, out[0][0])
            ctx.set_option("gcn_cache_ah0", 1)
            ctx.aggregate(0, da.FORWARD)                          # (set_option invalidated: computes)
            ctx.aggregate(0, da.FORWARD)                          # kept
            assert ctx.get_option("gcn_cache_ah0_skips") == skips + 1
            ctx.fill_uniform(0, "fg", 10, -1.0, 1.0, g["srcGhost"])
            ctx.aggregate(0, da.FORWARD)                          # ghost rows changed: computes
            assert ctx.get_option("gcn_cache_ah0_skips") == skips + 1
            assert not np.array_equal(ctx.download(0, "ah"), a2)
        res[cache] = out
        eng.close()
        ctx.close()
    for ep in range(3):
        for a, b in zip(res[0][ep], res[1][ep]):
            assert np.array_equal(a, b), ep


def test_adam_and_xavier_vs_oracle(da):
    import orc
    import partition_oracle as po
    from helpers import make_ctx, random_graph, rel_err
    V, dims = 64, [10, 6, 4]
    s, d = random_graph(3, V, 300)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = make_ctx(da, g, dims, V)
    ctx.weights_init_xavier()
    W = [ctx.weight_get(0), ctx.weight_get(1)]
    assert np.array_equal(W[0], orc.xavier(10, 6)) and np.array_equal(W[1], orc.xavier(6, 4))
    rng = np.random.default_rng(0)
    ctx.upload(0, "x", rng.uniform(-1, 1, (V, 10)).astype(np.float32))
    ctx.labels_upload(rng.integers(0, 4, V).astype(np.uint32))
    ctx.adam_config(0.01)
    eng = da.Engine(ctx, da.GCN, 2)
    m = [np.zeros_like(w) for w in W]
    v = [np.zeros_like(w) for w in W]
    for ep in range(1, 4):
        ctx.aggregate(0, da.FORWARD); ctx.apply_vertex(0, da.FORWARD)
        ctx.aggregate(1, da.FORWARD); ctx.apply_vertex(1, da.FORWARD)
        ctx.aggregate(1, da.BACKWARD); ctx.apply_vertex(0, da.BACKWARD)
        grads = [ctx.weight_grad_get(0), ctx.weight_grad_get(1)]
        ctx.weight_update(1)
        ctx.weight_update(0)
        for l in (1, 0):
            orc.adam_update(W[l], grads[l], m[l], v[l], 0.01, ep)
        for l in (0, 1):
            assert rel_err(ctx.weight_get(l), W[l]) < 1e-6, (ep, l)
    ctx.close()


def test_adam_reference_known_answer(da):
    """K7 on the one output of the reference's real AdamOptimizer on record (AdamOptimizer.cpp linked into
    a harness, w = 0.5, g = 0.1, lr = 0.01, first iteration -> 0.49000031): every weight, both layers, bit for bit."""
    import partition_oracle as po
    from helpers import make_ctx, random_graph
    V, dims = 64, [10, 6, 4]
    s, d = random_graph(3, V, 300)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = make_ctx(da, g, dims, V)
    ctx.adam_config(0.01)
    for l in (0, 1):
        ctx.weight_set(l, "w", np.full((dims[l], dims[l + 1]), 0.5, np.float32))
        ctx.weight_grad_set(l, np.full((dims[l], dims[l + 1]), 0.1, np.float32))
    ctx.weight_update(1)
    ctx.weight_update(0)
    for l in (0, 1):
        w = ctx.weight_get(l)
        assert np.all(w == np.float32(0.49000031)), (l, w.ravel()[:3])
    ctx.close()


def test_engine_epoch_matches_manual_order(da):
    """dorylus_amd.Engine (reference stage order / chunk state machine) == manual calls."""
    import partition_oracle as po
    from helpers import make_ctx, random_graph
    V, dims = 80, [12, 8, 5]
    s, d = random_graph(5, V, 500)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (V, 12)).astype(np.float32)
    lab = rng.integers(0, 5, V).astype(np.uint32)
    out = []
    for mode in ("engine", "manual"):
        ctx = make_ctx(da, g, dims, V)
        ctx.weights_init_xavier()
        ctx.upload(0, "x", X)
        ctx.labels_upload(lab)
        ctx.adam_config(0.01)
        if mode == "engine":
            eng = da.Engine(ctx, da.GCN, 2)
            for ep in range(3):
                eng.run_epoch(ep + 1)
        else:
            for ep in range(3):
                ctx.aggregate(0, 0); ctx.apply_vertex(0, 0)
                ctx.aggregate(1, 0); ctx.apply_vertex(1, 0); ctx.weight_update(1)
                ctx.aggregate(1, 1); ctx.apply_vertex(0, 1); ctx.weight_update(0)
        out.append([ctx.weight_get(0), ctx.weight_get(1)])
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("nb", [0, 8])   # 8: force the source-blocked unit-weight path (toy graphs otherwise take K1)
@pytest.mark.parametrize("case,dims", [("parts_toy60_p1", [9, 6, 4]), ("parts_toy60_p2", [30, 16, 5]),
                                       ("parts_toy60_p2", [30, 64, 5])])
def test_gat_stages_vs_oracle(da, case, dims, nb):
    """K5 + GAT aggregate forward/backward (parity vs reference unpinned; oracle = restatement)."""
    import orc
    from helpers import make_ctx, rel_err
    gs, parts = _golden_partitions(case)
    rng = np.random.default_rng(11)
    for r, g in enumerate(gs):
        N, E = g["localVtxCnt"], g["localInEdgeCnt"]
        ctx = make_ctx(da, g, dims, g["globalVtxCnt"], gnn=da.GAT, node_id=r, num_nodes=len(gs),
                       options={"spmm_blk_nb": nb})
        F = dims[1]
        h = rng.uniform(-1, 1, (N, dims[0])).astype(np.float32)
        W = (rng.standard_normal((dims[0], F)) / 3).astype(np.float32)
        a = rng.standard_normal((F, 1)).astype(np.float32)
        ctx.upload(0, "h", h)
        ctx.weight_set(0, "w", W)
        ctx.weight_set(0, "a_i", a)
        ctx.apply_vertex(0, da.FORWARD)
        z = ctx.download(0, "z")
        assert rel_err(z, orc.sgemm(h, W)) < RTOL
        fgz = rng.uniform(-1, 1, (g["srcGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "fg_z", fgz)
        ctx.apply_edge(1, da.FORWARD)
        az_ref, A_ref = orc.edge_forward_gat(g["colPtr"], z, a)
        assert rel_err(ctx.download(0, "az").ravel(), az_ref) < RTOL
        assert rel_err(ctx.download(0, "A").ravel(), A_ref) < RTOL
        ctx.aggregate(1, da.FORWARD)
        A = ctx.download(0, "A").ravel()
        assert rel_err(ctx.download(0, "ah"), orc.aggregate_gat_fwd(g["colPtr"], g["rowIdx"], A, z, fgz)) < RTOL
        # backward
        grad = rng.uniform(-1, 1, (N, F)).astype(np.float32)
        bgd = rng.uniform(-1, 1, (g["dstGhostCnt"], F)).astype(np.float32)
        ctx.upload(0, "grad", grad)
        ctx.upload(0, "bg_d", bgd)
        ctx.apply_edge(1, da.BACKWARD)
        az = ctx.download(0, "az").ravel()
        dA_ref, da_ref = orc.edge_backward_gat(g["colPtr"], grad, az, z, a)
        assert rel_err(ctx.download(0, "dA").ravel(), dA_ref) < RTOL
        assert rel_err(ctx.weight_grad_get(0, "a_i").ravel(), da_ref) < 5e-4
        ctx.aggregate(1, da.BACKWARD)
        dA = ctx.download(0, "dA").ravel()
        ref = orc.aggregate_gat_bwd(g["rowPtr"], g["colIdx"], g["csrVal"], grad, bgd,
                                    g["colPtr"], g["rowIdx"], dA, z, fgz)
        assert rel_err(ctx.download(0, "aTg"), ref) < RTOL
        ctx.apply_vertex(0, da.BACKWARD)
        assert rel_err(ctx.weight_grad_get(0), orc.sgemm(h, ctx.download(0, "aTg"), ta=True)) < RTOL
        ctx.close()


@pytest.mark.parametrize("reuse", [1, 0])
@pytest.mark.parametrize("nb", [0, 8])
def test_gat_engine_epoch_vs_oracle(da, nb, reuse):
    """Whole GAT epoch through the C++ Engine (stage order of pipeline.cpp for GAT) == oracle; with the forward's neighbour
    sum reused by the backward's dA-weighted aggregation (gat_reuse_nsum, round 6: two aggregations per layer instead of
    three where a blocked layout gathers with unit weights) and with every aggregation gathered afresh."""
    import orc
    import partition_oracle as po
    from helpers import make_ctx, oracle_gat_epoch, random_graph, rel_err
    V, dims = 300, [24, 16, 6]
    s, d = random_graph(4, V, 2500)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    rng = np.random.default_rng(2)
    H0 = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(2)]
    As = [(rng.standard_normal((dims[i + 1], 1)) / 2).astype(np.float32) for i in range(2)]
    ctx = make_ctx(da, g, dims, V, gnn=da.GAT, options={"spmm_blk_nb": nb, "gat_reuse_nsum": reuse})
    ctx.upload(0, "h", H0)
    ctx.labels_upload(labels)
    for l in range(2):
        ctx.weight_set(l, "w", Ws[l])
        ctx.weight_set(l, "a_i", As[l])
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    ctx.timing_enable(True)
    eng.run(1)
    ms, launches = ctx.timing_get("spmm")
    ctx.timing_enable(False)
    if nb:      # the blocked layouts' unit-weight gathers: the reuse replaces one sweep per layer by a row-wise kernel (same count, cheaper)
        assert launches >= 6
    T, dWs, das = oracle_gat_epoch(g, H0, labels, Ws, As)
    for l in range(2):
        assert rel_err(ctx.download(l, "z"), T[f"z{l}"]) < RTOL, l
        assert rel_err(ctx.download(l, "az").ravel(), T[f"az{l}"]) < RTOL, l
        assert rel_err(ctx.download(l, "ah"), T[f"ah{l}"]) < RTOL, l
        assert rel_err(ctx.download(l, "grad"), T[f"grad{l}"]) < RTOL, l
        assert rel_err(ctx.download(l, "dA").ravel(), T[f"dA{l}"]) < RTOL, l
        assert rel_err(ctx.download(l, "aTg"), T[f"aTg{l}"]) < RTOL, l
        assert rel_err(ctx.weight_grad_get(l), dWs[l]) < RTOL, l
        assert rel_err(ctx.weight_grad_get(l, "a_i").ravel(), das[l]) < 5e-4, l
    # "w" took one Adam step, "a_i" is left as the (faked) weight server leaves it
    m = [np.zeros_like(w) for w in Ws]
    v = [np.zeros_like(w) for w in Ws]
    for l in (1, 0):
        W = Ws[l].copy()
        orc.adam_update(W, dWs[l], m[l], v[l], 0.01, 1)
        assert rel_err(ctx.weight_get(l), W) < 1e-5
        assert np.array_equal(ctx.weight_get(l, "a_i"), As[l])
    eng.close()
    ctx.close()


@pytest.mark.parametrize("case", ["parts_toy60_p2", "parts_toy97_p8_und"])
@pytest.mark.parametrize("nb", [0, 8])
def test_gat_epoch_partitions_vs_oracle(da, case, nb):
    """The reference's GAT prototype ACROSS partitions, end to end: P contexts in one process walk the Engine's GAT stage
    order (AV -> SC -> AE -> GA, predict, backwards), `z` travels forward into fg_z and `grad` backward into bg_d through
    the pack / unpack kernels and the halo plans (Engine::scatterGAT / ghostReceiverGAT, gat_ops.cpp:277-435) -- every
    named tensor of every partition against the oracle's epoch over the same reference-built partitions, ghost rows
    bit-equal to the owners' rows."""
    import torch
    from halo_plan_ref import halo_plan
    from helpers import assert_parity, make_ctx, oracle_gat_epoch_parts
    gs, parts = _golden_partitions(case)
    P, V = len(gs), int(gs[0]["globalVtxCnt"])
    dims = [20, 16, 6]
    L = 2
    rng = np.random.default_rng(11)
    H0 = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(L)]
    As = [(rng.standard_normal((dims[i + 1], 1)) / 2).astype(np.float32) for i in range(L)]
    ctxs, plans = [], []
    for r, g in enumerate(gs):
        ctx = make_ctx(da, g, dims, V, gnn=da.GAT, node_id=r, num_nodes=P, options={"spmm_blk_nb": nb})
        ctx.upload(0, "h", H0[g["localToGlobal"]])
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l in range(L):
            ctx.weight_set(l, "w", Ws[l])
            ctx.weight_set(l, "a_i", As[l])
        pl = halo_plan(g, parts, r, P)
        for dd in (0, 1):
            ctx.halo_plan(dd, pl[dd][0], pl[dd][1])
        ctxs.append(ctx)
        plans.append(pl)

    def exchange(layer, dd):   # SC of GAT layer `layer` (1-based like the Engine's chunk.layer after incLayerGAT)
        _, _, ld, _ = ctxs[0].info(layer - 1, "z" if dd == 0 else "grad")
        send = [torch.zeros(max(1, sum(len(x) for x in plans[r][dd][0])) * ld, device="cuda") for r in range(P)]
        recv = [torch.zeros(max(1, sum(len(x) for x in plans[r][dd][1])) * ld, device="cuda") for r in range(P)]
        torch.cuda.synchronize()   # (the contexts' streams are non-blocking: torch's zero fills must have landed before a pack kernel writes)
        for r in range(P):
            ctxs[r].halo_pack(layer, dd, send[r].data_ptr())
            ctxs[r].sync()
        for r in range(P):
            soff = np.concatenate([[0], np.cumsum([len(x) for x in plans[r][dd][0]])])
            for p in range(P):
                roff = np.concatenate([[0], np.cumsum([len(x) for x in plans[p][dd][1]])])
                n = len(plans[r][dd][0][p])
                recv[p][roff[r] * ld:(roff[r] + n) * ld] = send[r][soff[p] * ld:(soff[p] + n) * ld]
        torch.cuda.synchronize()
        for r in range(P):
            ctxs[r].halo_unpack(layer, dd, recv[r].data_ptr())
            ctxs[r].sync()

    for l in range(L):                                  # pipeline.cpp order for GAT (host/engine.cpp: runEpoch)
        for c in ctxs:
            c.apply_vertex(l, da.FORWARD)               # AV
        exchange(l + 1, da.FORWARD)                     # SC
        for c in ctxs:
            c.apply_edge(l + 1, da.FORWARD)             # AE
            c.aggregate(l + 1, da.FORWARD)              # GA
    for c in ctxs:
        c.predict_gat(L)
    for l in range(L - 1, -1, -1):
        exchange(l + 1, da.BACKWARD)
        for c in ctxs:
            c.apply_edge(l + 1, da.BACKWARD)
            c.aggregate(l + 1, da.BACKWARD)
            c.apply_vertex(l, da.BACKWARD)
    T, dWs, das = oracle_gat_epoch_parts(gs, parts, H0, labels, Ws, As)
    for r, c in enumerate(ctxs):
        if gs[r]["localVtxCnt"] == 0:
            continue
        for l in range(L):
            for nm in ("z", "ah", "grad", "aTg"):
                assert_parity(c.download(l, nm), T[r][f"{nm}{l}"], (case, nb, r, l, nm))
            assert_parity(c.download(l, "az").ravel(), T[r][f"az{l}"], (case, nb, r, l, "az"))
            assert_parity(c.download(l, "dA").ravel(), T[r][f"dA{l}"], (case, nb, r, l, "dA"))
            if gs[r]["srcGhostCnt"]:
                assert_parity(c.download(l, "fg_z"), T[r][f"fg_z{l}"], (case, nb, r, l, "fg_z"))
            if gs[r]["dstGhostCnt"]:
                assert_parity(c.download(l, "bg_d"), T[r][f"bg_d{l}"], (case, nb, r, l, "bg_d"))
    # ghost rows are the owners' rows bit for bit (fg_z[slot(gvid)] == owner.z[lvid(gvid)], bg_d likewise for grad)
    for l in range(L):
        g2row = [{}, {}]
        for r, c in enumerate(ctxs):
            zz, gg = c.download(l, "z"), c.download(l, "grad")
            for i, gv in enumerate(gs[r]["localToGlobal"]):
                g2row[0][int(gv)] = zz[i]
                g2row[1][int(gv)] = gg[i]
        for r, c in enumerate(ctxs):
            if gs[r]["srcGhostCnt"]:
                assert np.array_equal(c.download(l, "fg_z"), np.stack([g2row[0][int(gv)] for gv in gs[r]["srcGhost"]])), (r, l, "fg_z bits")
            if gs[r]["dstGhostCnt"]:
                assert np.array_equal(c.download(l, "bg_d"), np.stack([g2row[1][int(gv)] for gv in gs[r]["dstGhost"]])), (r, l, "bg_d bits")
    for l in range(L):
        assert_parity(sum(c.weight_grad_get(l) for c in ctxs), dWs[l], (case, nb, "dW", l))
        assert_parity(sum(c.weight_grad_get(l, "a_i") for c in ctxs).ravel(), das[l], (case, nb, "da", l), rtol=5e-4, atol_frac=5e-5)
    for c in ctxs:
        c.close()


def test_tanh_matches_libm(da):
    """dory_tanh (csrc/ctx.hpp: odd polynomial below 0.35, 1 - 2 / (1 + exp 2x) above; 13 instructions for libm's ~40) in the
    transform's epilogue: h = tanh(z) on the z the GPU itself holds, from 1e-6 to saturation, within 1e-6 relative of libm
    (the oracle's tanhf, CPU_comm.cpp:265-274 `std::tanh`)."""
    import partition_oracle as po
    from helpers import make_ctx, random_graph
    V, dims = 4096, [64, 128, 8]
    s, d = random_graph(3, V, 20000)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    rng = np.random.default_rng(1)
    X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    X *= (10.0 ** rng.uniform(-6, 1.6, (V, 1))).astype(np.float32)     # rows of very different magnitude: z from 1e-6 to +-40
    ctx = make_ctx(da, g, dims, V)
    ctx.upload(0, "x", X)
    ctx.weight_set(0, "w", (rng.standard_normal((dims[0], dims[1])) / 4).astype(np.float32))
    ctx.weight_set(1, "w", np.zeros((dims[1], dims[2]), np.float32))
    ctx.aggregate(0, da.FORWARD)
    ctx.apply_vertex(0, da.FORWARD)
    z, h = ctx.download(0, "z").astype(np.float64), ctx.download(0, "h").astype(np.float64)
    ref = np.tanh(z)
    nz = ref != 0
    assert np.abs(z).max() > 20 and (np.abs(z) < 1e-3).mean() > 0.001 and ((np.abs(z) > 0.2) & (np.abs(z) < 0.6)).mean() > 0.01   # the premise: all three regimes are there
    assert (np.abs(h - ref)[nz] / np.abs(ref[nz])).max() < 1e-6
    assert np.array_equal(h[~nz], ref[~nz]) and np.abs(h).max() <= 1.0
    ctx.close()


def test_fill_uniform_matches_host_twin(da):
    import partition_oracle as po
    from helpers import make_ctx, random_graph, splitmix_uniform
    V = 50
    s, d = random_graph(2, V, 100)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = make_ctx(da, g, [37, 8, 3], V)
    ids = np.arange(V, dtype=np.uint32)[::-1].copy()
    ctx.fill_uniform(0, "x", 12345, -1.0, 1.0, ids)
    assert np.array_equal(ctx.download(0, "x"), splitmix_uniform(12345, ids, 37))
    ctx.close()


def test_errors_do_not_abort(da):
    import partition_oracle as po
    from helpers import random_graph
    ctx = da.Context(0)
    with pytest.raises(da.DoryError):
        ctx.preallocate()                       # not configured
    ctx.configure(da.GCN, [4, 3, 2], 10)
    s, d = random_graph(1, 10, 20)
    g = po.preprocess(s, d, np.zeros(10, np.int64), 0, 1)
    bad = dict(g)
    bad["rowIdx"] = g["rowIdx"].copy()
    if bad["rowIdx"].size:
        bad["rowIdx"][0] = 1000
        with pytest.raises(da.DoryError):
            ctx.graph_upload(bad)
    ctx.graph_upload(g)
    ctx.preallocate()
    with pytest.raises(da.DoryError):
        ctx.aggregate(7, da.FORWARD)
    with pytest.raises(da.DoryError):
        ctx.download(0, "nope")
    ctx.close()


def test_rccl_communicator_inside_the_library(da):
    """The RCCL calls the library makes itself (the N > 1 path cannot run on a one-GPU box: RCCL refuses two ranks on
    one device): unique id, a one-rank communicator created next to torch's own RCCL, and an epoch after it."""
    import partition_oracle as po
    from helpers import make_ctx
    rng = np.random.default_rng(5)
    V = 200
    s, d = rng.integers(0, V, 1500), rng.integers(0, V, 1500)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    ctx = make_ctx(da, g, [12, 8, 3], V)
    uid = ctx.comm_unique_id()
    assert uid.shape == (128,) and uid.any()
    ctx.comm_init(uid, 0, 1)
    with pytest.raises(da.DoryError):
        ctx.comm_init(uid, 1, 1)                      # rank out of range: refused before RCCL sees it
    ctx.fill_uniform(0, "x", 1)
    ctx.labels_upload((np.arange(V) % 3).astype(np.uint32))
    ctx.weights_init_xavier()
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    eng.run(2)
    assert np.isfinite(ctx.weight_get(0, "w")).all()
    eng.close()
    ctx.close()


def test_sweep_variant_random_shapes(da):
    """K1s against the oracle on a spread of shapes the fixed cases do not hit: vertex counts around the layout's group
    and sweep boundaries, empty and very long rows (pieces), odd widths, explicit and automatic block counts, forced rows
    per lane group (process-wide option, restored) -- forward (CSC) and backward (CSR) on every one."""
    import orc
    import partition_oracle as po
    from helpers import make_ctx, rel_err
    rng = np.random.default_rng(2026)
    cases = [  # V, E, F, nb, rows, skew
        (9, 40, 32, 8, 0, 0), (65, 900, 96, 8, 2, 0), (1023, 9000, 64, 16, 0, 0), (1025, 30000, 128, 8, 4, 1),
        (8193, 100000, 41, 24, 0, 0), (8200, 160000, 300, 16, 8, 2), (20000, 400000, 256, 0, 0, 0),
        (33000, 700000, 132, 40, 6, 1), (70001, 1500000, 64, 0, 10, 2), (120000, 2000000, 602, 0, 0, 1),
        # dense: a lane group's entries of one step exceed a staging pass (127), with rows in pairs (3 slabs) and without
        (2000, 400000, 384, 8, 0, 0), (2000, 400000, 128, 8, 0, 0), (3000, 500000, 64, 8, 0, 2),
    ]
    try:
        for V, E, F, nb, rows, skew in cases:
            if skew == 0:
                s, d = rng.integers(0, V, E), rng.integers(0, V, E)
            elif skew == 1:   # a few hubs: rows far above twice the mean degree (cut into pieces), many empty rows
                d = np.where(rng.random(E) < 0.3, rng.integers(0, max(1, V // 500), E), rng.integers(0, V // 2 + 1, E))
                s = rng.integers(0, V, E)
            else:             # power law on both sides
                s = (V * rng.random(E) ** 3).astype(np.int64) % V
                d = (V * rng.random(E) ** 3).astype(np.int64) % V
            g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
            ctx = make_ctx(da, g, [F, F, 3], V, options={"spmm_variant": 2, "spmm_blk_nb": nb, "spmm_sweep_rows": rows})
            x = rng.standard_normal((V, F)).astype(np.float32)
            gr = rng.standard_normal((V, F)).astype(np.float32)
            ctx.upload(0, "x", x)
            ctx.upload(1, "grad", gr)
            ctx.aggregate(0, da.FORWARD)
            ctx.aggregate(1, da.BACKWARD)
            ref_f = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], x)
            ref_b = orc.aggregate_gcn(g["rowPtr"], g["colIdx"], g["csrVal"], g["norm"], gr)
            tol = 5e-5 if skew == 2 else 1e-5      # power law: rows of 1e4-1e5 terms, summed in pieces (fp32 reassociation)
            assert rel_err(ctx.download(0, "ah"), ref_f) < tol, (V, E, F, nb, rows, skew)
            assert rel_err(ctx.download(0, "aTg"), ref_b) < tol, (V, E, F, nb, rows, skew)
            ctx.close()
    finally:
        c = da.Context(0)
        c.set_option("spmm_sweep_rows", 0)
        c.close()


def _poison_padding(ctx, layer, name):
    """NaN into the padding columns [cols, ld) of a device tensor, through its dory_tensor_info pointer"""
    import ctypes as C
    rows, cols, ld, p = ctx.info(layer, name)
    if ld == cols or rows == 0:
        return 0
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    pad = np.full((rows, ld - cols), np.nan, np.float32)
    ctx.sync()
    rc = hip.hipMemcpy2D(C.c_void_p(p + cols * 4), ld * 4, pad.ctypes.data_as(C.c_void_p), (ld - cols) * 4, (ld - cols) * 4, rows, 1)
    assert rc == 0
    return ld - cols


def test_gemm_does_not_read_operand_padding(da):
    """K2's operand tiles come in 16-byte pieces of 16-deep k-tiles, so the last k-tile of a row-major operand covers columns
    past K (602 -> 608, 24 -> 32, 12 -> 16): whatever lies there must not reach the product.  The tensors' padding is poisoned
    with NaN right before every transform (NN with the tanh epilogue, TN split-K, NT) and the results are compared with the
    oracle's -- a missing mask or a wrong out-of-range bound shows up as NaN, not as a 1e-7 difference."""
    import orc
    import partition_oracle as po
    from helpers import assert_parity, make_ctx
    V, dims = 777, [50, 24, 12, 7]
    ids = np.arange(V)
    g = po.preprocess(ids, ids, np.zeros(V, np.int64), 0, 1)     # one self edge per vertex: the graph does not matter here
    rng = np.random.default_rng(77)
    ctx = make_ctx(da, g, dims, V)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(3)]
    for l in range(3):
        ctx.weight_set(l, "w", Ws[l])
    ah = [rng.uniform(-1, 1, (V, dims[l])).astype(np.float32) for l in range(3)]
    # hidden layers, forward: z = ah W (+ h = tanh z)
    for l in (0, 1):
        ctx.upload(l, "ah", ah[l])
        assert _poison_padding(ctx, l, "ah") > 0
        ctx.apply_vertex(l, da.FORWARD)
        z = ctx.download(l, "z")
        assert np.isfinite(z).all()
        assert_parity(z, orc.sgemm(ah[l], Ws[l]), what=f"z@{l}")
        assert_parity(ctx.download(l, "h"), np.tanh(orc.sgemm(ah[l], Ws[l]).astype(np.float64)).astype(np.float32), what=f"h@{l}")
    # layer 1, backward: g = aTg * (1 - tanh^2 z); dW = ah^T g (TN); grad = g W^T (NT, K = 12 of a 16-deep tile)
    aTg = rng.uniform(-1, 1, (V, dims[2])).astype(np.float32)
    ctx.upload(1, "aTg", aTg)
    for name in ("ah", "aTg", "g", "z"):
        _poison_padding(ctx, 1, name)
    ctx.apply_vertex(1, da.BACKWARD)
    z1 = orc.sgemm(ah[1], Ws[1]).astype(np.float64)
    g1 = (aTg * (1.0 - np.tanh(z1) ** 2)).astype(np.float32)
    dW = ctx.weight_grad_get(1)
    grad = ctx.download(1, "grad")
    assert np.isfinite(dW).all() and np.isfinite(grad).all()
    assert_parity(dW, orc.sgemm(ah[1], g1, ta=True), what="dW@1")
    assert_parity(grad, orc.sgemm(g1, Ws[1], tb=True), what="grad@1")
    ctx.close()


def test_gat_lazy_edge_tensors_are_the_eager_ones(da):
    """GAT prototype, gat_lazy_edge_tensors (round 6): az / A / dA hold one value per DESTINATION, so the stages keep the
    per-vertex values and write the per-edge tensors only when somebody reads them (download, raw pointer, K1's per-edge
    path).  After the same epoch the tensors a caller can see are the eager run's, bit for bit -- through
    dory_tensor_download and through the dory_tensor_info pointer -- and a caller's own "az" still drives the backward."""
    import ctypes as C
    import partition_oracle as po
    from helpers import make_ctx, random_graph
    V, dims = 300, [24, 16, 6]
    s, d = random_graph(4, V, 2500)
    g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    rng = np.random.default_rng(2)
    H0 = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(2)]
    As = [(rng.standard_normal((dims[i + 1], 1)) / 2).astype(np.float32) for i in range(2)]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    got = {}
    for lazy in (0, 1):
        for nb in (0, 8):                       # K1 (reads the per-edge values itself) and the blocked layouts (never do)
            ctx = make_ctx(da, g, dims, V, gnn=da.GAT, options={"spmm_blk_nb": nb, "gat_lazy_edge_tensors": lazy})
            ctx.upload(0, "h", H0)
            ctx.labels_upload(labels)
            for l in range(2):
                ctx.weight_set(l, "w", Ws[l])
                ctx.weight_set(l, "a_i", As[l])
            da.NativeEngine(ctx).run(1)
            out = {f"{nm}{l}": ctx.download(l, nm) for l in range(2) for nm in ("az", "dA", "ah", "aTg")}
            out["A"] = ctx.download(0, "A")      # (= forwardAdj.values: the scores of the layer whose edge stage ran last)
            rows, cols, ld, ptr = ctx.info(1, "az")
            raw = np.empty(rows * max(ld, 1), np.float32)
            ctx.sync()
            assert hip.hipMemcpy(raw.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), raw.nbytes, 2) == 0
            out["az1_raw"] = raw
            for k, v in out.items():
                key = (nb, k)
                if key in got:
                    assert np.array_equal(got[key], v), (lazy, nb, k)
                got[key] = v
            ctx.close()
    assert np.array_equal(got[(8, "az1_raw")][:got[(8, "az1")].size], got[(8, "az1")].ravel())
