"""The multi-rank path as far as one GPU allows: two (and three) *processes* share device 0, each is one rank running
`dory_engine_run` -- partition upload with the C++ halo plans, pack -> exchange -> unpack on the comm stream with the
event ordering of the overlapped schedule, interior-source SpMM blocks under the exchange, gradient sum, Adam -- and
only the bytes travel differently: `dory_comm_set_host_transport` hands the packed rows and the weight gradients to
gloo instead of RCCL (two ranks cannot share one GPU under RCCL).  Every named tensor of every rank is compared with
the oracle's epoch over the same partitions (reference semantics: gcn_ops.cpp:204-362 for the exchange,
weighttensor.cpp:131-166 + AdamOptimizer.cpp:29-51 for the update)."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        import dorylus_amd as da
        import orc
        from helpers import oracle_gcn_epoch, rel_err
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dims, V, E, epochs, opts = case["dims"], case["V"], case["E"], case["epochs"], case.get("opts", {})
        rng = np.random.default_rng(case["seed"])
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        src, dst = np.concatenate([s, d]), np.concatenate([d, s])
        if case["parts"] == "block":
            parts = (np.arange(V, dtype=np.int64) * world // V).astype(np.int32)
        else:
            parts = rng.integers(0, world, V).astype(np.int32)
        X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
        labels = rng.integers(0, dims[-1], V).astype(np.uint32)
        L = len(dims) - 1
        Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(L)]
        all_parts = [da.Partition.build(src, dst, parts, r, world) for r in range(world)]   # every rank: oracle side
        gs = [p.view() for p in all_parts]
        part, g = all_parts[rank], gs[rank]

        ctx = da.Context(0)
        ctx.configure(da.GCN, dims, V, rank, world)
        for k, v in opts.items():
            ctx.set_option(k, v)
        part.upload(ctx, parts)                    # adjacency + both halo plans (dory_partition_upload)
        ctx.preallocate()
        ctx.upload(0, "x", X[g["localToGlobal"]])
        if g["srcGhostCnt"]:
            ctx.upload(0, "fg", X[g["srcGhost"]].reshape(g["srcGhostCnt"], dims[0]))
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l, W in enumerate(Ws):
            ctx.weight_set(l, "w", W)
        ctx.adam_config(0.01)

        calls = {"a2a": 0, "ar": 0, "bytes": 0}

        def alltoallv(send, sc, so, recv, rc, ro):
            calls["a2a"] += 1
            reqs, keep = [], []
            for p in range(world):
                if p == rank:
                    assert sc[p] == 0 and rc[p] == 0
                    continue
                if rc[p]:
                    t = torch.empty(int(rc[p]), dtype=torch.float32)
                    keep.append((t, int(ro[p]), int(rc[p])))
                    reqs.append(dist.irecv(t, p))
                if sc[p]:
                    t = torch.from_numpy(send[int(so[p]):int(so[p] + sc[p])].copy())
                    calls["bytes"] += 4 * int(sc[p])
                    reqs.append(dist.isend(t, p))
            for r_ in reqs:
                r_.wait()
            for t, o, n in keep:
                recv[o:o + n] = t.numpy()

        def allreduce(buf):
            calls["ar"] += 1
            t = torch.from_numpy(buf.copy())
            dist.all_reduce(t)
            buf[:] = t.numpy()
        ctx.set_host_transport(alltoallv, allreduce)

        eng = da.NativeEngine(ctx)
        eng.run(epochs)
        ctx.sync()
        # the validation statistics summed over the ranks (the weight servers' updateGlobalAccLoss): one more all-reduce of 3 floats
        loc = ctx.train_stat()
        glob = ctx.train_stat_global()
        tl = torch.tensor([loc[0], loc[1], float(loc[2])], dtype=torch.float64)
        dist.all_reduce(tl)
        stat_err = max(abs(glob[0] - tl[0].item()), abs(glob[1] - tl[1].item()) / max(1.0, abs(tl[1].item())), abs(glob[2] - tl[2].item()))

        # ---- oracle: the same epochs over the same partitions, Adam on the summed gradients ----
        Wo = [w.copy() for w in Ws]
        m = [np.zeros_like(w) for w in Ws]
        v = [np.zeros_like(w) for w in Ws]
        T = dW = None
        for ep in range(epochs):
            T, dW = oracle_gcn_epoch(gs, parts, X, labels, Wo, V)
            for l in range(L - 1, -1, -1):
                orc.adam_update(Wo[l], dW[l], m[l], v[l], 0.01, ep + 1)
        errs = {}
        if g["localVtxCnt"]:
            for l in range(L):
                errs[f"ah{l}"] = rel_err(ctx.download(l, "ah"), T[rank][f"ah{l}"])
                if l < L - 1:
                    errs[f"h{l}"] = rel_err(ctx.download(l, "h"), T[rank][f"h{l}"])
                    errs[f"aTg{l}"] = rel_err(ctx.download(l, "aTg"), T[rank][f"aTg{l}"])
                if l > 0:
                    errs[f"grad{l}"] = rel_err(ctx.download(l, "grad"), T[rank][f"grad{l}"])
                    if g["srcGhostCnt"]:
                        # ghost rows are the owners' rows bit for bit (they were computed on the other process)
                        errs[f"fg{l}"] = rel_err(ctx.download(l, "fg"), T[rank][f"fg{l}"])
                    if g["dstGhostCnt"]:
                        errs[f"bg{l-1}"] = rel_err(ctx.download(l - 1, "bg"), T[rank][f"bg{l-1}"])
        for l in range(L):
            errs[f"dW{l}"] = rel_err(ctx.weight_grad_get(l), dW[l])        # the all-reduced sum, on every rank
            errs[f"W{l}"] = rel_err(ctx.weight_get(l), Wo[l])             # after `epochs` identical Adam steps
        # the exchanged ghost rows themselves: identical bits on owner and receiver
        own_h = ctx.download(0, "h")
        allh = [None] * world
        dist.all_gather_object(allh, (g["localToGlobal"], own_h))
        if g["srcGhostCnt"]:
            g2row = {}
            for l2g, hh in allh:
                for i, gv in enumerate(l2g):
                    g2row[int(gv)] = hh[i]
            want = np.stack([g2row[int(gv)] for gv in g["srcGhost"]])
            errs["fg1_bits"] = 0.0 if np.array_equal(ctx.download(1, "fg"), want) else 1.0
        expect_a2a = epochs * 2 * (L - 1)
        errs["stat_global"] = stat_err
        ok_calls = calls["a2a"] == expect_a2a and calls["ar"] == epochs * L + 1
        eng.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, errs, ok_calls, calls))
    except Exception:
        q.put((rank, traceback.format_exc(), False, None))


CASES = {
    "two_ranks_reddit_dims": dict(dims=[602, 128, 41], V=3000, E=40000, seed=1, epochs=2, parts="block"),
    "two_ranks_hash_parts_3layer": dict(dims=[300, 64, 64, 25], V=2500, E=30000, seed=2, epochs=2, parts="hash"),
    "three_ranks_sweep_blocks": dict(dims=[128, 128, 16], V=30000, E=300000, seed=3, epochs=1, parts="block",
                                     opts={"spmm_blk_nb": 16}),   # K1s in two launches: local-source blocks under the exchange
    "two_ranks_no_overlap": dict(dims=[64, 32, 8], V=2000, E=20000, seed=4, epochs=1, parts="hash", opts={"halo_overlap": 0}),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_epochs_two_processes_one_gpu(name):
    import torch.multiprocessing as mp
    case = CASES[name]
    world = 3 if name.startswith("three") else 2
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, errs, ok_calls, calls in sorted(res, key=lambda t: t[0]):
        assert isinstance(errs, dict), f"rank {rank} failed:\n{errs}"
        assert ok_calls, (rank, calls)
        bad = {k: e for k, e in errs.items() if not e < RTOL}
        assert not bad, (rank, bad)


def _worker_gat(rank, world, port, case, q):
    """the reference's GAT prototype as `world` processes on one GPU: dory_engine_run with DORY_GAT, z forward / grad backward
    over the host transport (Engine::scatterGAT / ghostReceiverGAT, gat_ops.cpp:277-435)."""
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        import dorylus_amd as da
        from helpers import elem_err, oracle_gat_epoch_parts, rel_err
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dims, V, E = case["dims"], case["V"], case["E"]
        rng = np.random.default_rng(case["seed"])
        s, d = rng.integers(0, V, E), rng.integers(0, V, E)
        src, dst = np.concatenate([s, d]), np.concatenate([d, s])
        parts = (np.arange(V, dtype=np.int64) * world // V).astype(np.int32) if case["parts"] == "block" else rng.integers(0, world, V).astype(np.int32)
        H0 = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
        labels = rng.integers(0, dims[-1], V).astype(np.uint32)
        L = len(dims) - 1
        Ws = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(L)]
        As = [(rng.standard_normal((dims[i + 1], 1)) / 2).astype(np.float32) for i in range(L)]
        all_parts = [da.Partition.build(src, dst, parts, r, world) for r in range(world)]
        gs = [p.view() for p in all_parts]
        part, g = all_parts[rank], gs[rank]
        ctx = da.Context(0)
        ctx.configure(da.GAT, dims, V, rank, world)
        for k, v in case.get("opts", {}).items():
            ctx.set_option(k, v)
        part.upload(ctx, parts)
        ctx.preallocate()
        ctx.upload(0, "h", H0[g["localToGlobal"]])
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l in range(L):
            ctx.weight_set(l, "w", Ws[l])
            ctx.weight_set(l, "a_i", As[l])
        ctx.adam_config(0.01)
        calls = {"a2a": 0, "ar": 0}

        def alltoallv(send, sc, so, recv, rc, ro):
            calls["a2a"] += 1
            reqs, keep = [], []
            for p in range(world):
                if p == rank:
                    continue
                if rc[p]:
                    t = torch.empty(int(rc[p]), dtype=torch.float32)
                    keep.append((t, int(ro[p]), int(rc[p])))
                    reqs.append(dist.irecv(t, p))
                if sc[p]:
                    reqs.append(dist.isend(torch.from_numpy(send[int(so[p]):int(so[p] + sc[p])].copy()), p))
            for r_ in reqs:
                r_.wait()
            for t, o, n in keep:
                recv[o:o + n] = t.numpy()

        def allreduce(buf):
            calls["ar"] += 1
            t = torch.from_numpy(buf.copy())
            dist.all_reduce(t)
            buf[:] = t.numpy()
        ctx.set_host_transport(alltoallv, allreduce)
        eng = da.NativeEngine(ctx)
        eng.run(1)
        ctx.sync()
        T, dWs, das = oracle_gat_epoch_parts(gs, parts, H0, labels, Ws, As)
        errs = {}
        if g["localVtxCnt"]:
            for l in range(L):
                for nm in ("z", "ah", "grad", "aTg"):
                    a, b = ctx.download(l, nm), T[rank][f"{nm}{l}"]
                    errs[f"{nm}{l}"] = rel_err(a, b)
                    # element-wise criterion (tests/helpers.py) on the activations; grad = softmax - onehot is a difference of O(1)
                    # numbers (its entries go down to 1e-10 on confidently classified rows while its rounding error stays at
                    # eps x 1), so it and what is aggregated from it keep the max-norm bound
                    if nm in ("z", "ah"):
                        errs[f"{nm}{l}_elem"] = 1e-4 * elem_err(a, b)
                if g["srcGhostCnt"]:
                    errs[f"fg_z{l}"] = rel_err(ctx.download(l, "fg_z"), T[rank][f"fg_z{l}"])
                if g["dstGhostCnt"]:
                    errs[f"bg_d{l}"] = rel_err(ctx.download(l, "bg_d"), T[rank][f"bg_d{l}"])
        for l in range(L):
            errs[f"dW{l}"] = rel_err(ctx.weight_grad_get(l), dWs[l])          # the all-reduced sum, on every rank
        # ghost rows of the last layer's z: the owner's bits
        own = ctx.download(L - 1, "z")
        allz = [None] * world
        dist.all_gather_object(allz, (g["localToGlobal"], own))
        if g["srcGhostCnt"]:
            g2row = {int(gv): zz[i] for l2g, zz in allz for i, gv in enumerate(l2g)}
            want = np.stack([g2row[int(gv)] for gv in g["srcGhost"]])
            errs["fg_z_bits"] = 0.0 if np.array_equal(ctx.download(L - 1, "fg_z"), want) else 1.0
        ok_calls = calls["a2a"] == 2 * L and calls["ar"] >= L
        eng.close()
        ctx.close()
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, errs, ok_calls, calls))
    except Exception:
        q.put((rank, traceback.format_exc(), False, None))


@pytest.mark.parametrize("world,parts", [(2, "block"), (3, "hash")])
def test_gat_prototype_engine_epoch_processes_one_gpu(world, parts):
    import torch.multiprocessing as mp
    case = dict(dims=[24, 16, 6], V=2400, E=26000, seed=6, parts=parts, opts={"spmm_blk_nb": 8})
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker_gat, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank, errs, ok_calls, calls in sorted(res, key=lambda t: t[0]):
        assert isinstance(errs, dict), f"rank {rank} failed:\n{errs}"
        assert ok_calls, (rank, calls)
        bad = {k: e for k, e in errs.items() if not e < RTOL}
        assert not bad, (rank, bad)
