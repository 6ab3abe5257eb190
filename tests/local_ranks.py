"""P ranks in ONE process on ONE device over the in-process device transport (dory_comm_init_local): one context and one
host thread per rank, whole epochs inside the C++ Engine (host/engine.cpp) -- the overlapped halo schedule with copies
that really run on the comm streams beside the aggregation (Engine::scatterGCN + ghostReceiverGCN, gcn_ops.cpp:204-362).
Used by tests/test_gpu_local_transport.py and tools/local_transport_run.py."""
import threading

import numpy as np

TIMING_FAMS = ("spmm", "gemm", "halo", "halo_deferred", "halo_waited", "halo_hidden", "spmm_beside_halo", "allreduce")


def run_local(da, parts_objs, parts_vec, dims, gnn, epochs, setup, opts=None, timing=False, warm_epochs=0, downloads=(), pre=None,
              wnames=None):
    """parts_objs: da.Partition per rank; setup(ctx, rank, view) uploads inputs / weights / labels.  Returns a dict:
    tensors[rank][(layer, name)] for `downloads`, weights[rank][layer][name], wgrads likewise, timing (summed over the ranks), gates."""
    P = len(parts_objs)
    V = int(len(parts_vec))
    ctxs = []
    for r, part in enumerate(parts_objs):
        ctx = da.Context(0)
        ctx.configure(gnn, dims, V, r, P)
        if pre:
            pre(ctx)                          # (e.g. dory_gatmh_heads: before the tensors are laid out)
        for k, v in ((opts[r] if isinstance(opts, (list, tuple)) else opts) or {}).items():     # one dict for all ranks, or one per rank
            ctx.set_option(k, v)
        part.upload(ctx, parts_vec)          # adjacency + both halo plans (host/partition.cpp)
        ctx.preallocate()
        setup(ctx, r, part.view())
        ctx.adam_config(0.01)
        ctxs.append(ctx)
    for c in ctxs:
        c.sync()
    da.Context.comm_init_local(ctxs)
    engs = [da.NativeEngine(c) for c in ctxs]
    errors = [None] * P
    epoch_ms = [None] * P

    stats = [None] * P

    def drive(r, n):
        try:
            epoch_ms[r] = engs[r].run(n)
            if gnn == da.GCN:              # the weight servers' sum of the validation statistics over the nodes: a collective
                stats[r] = (ctxs[r].train_stat(), ctxs[r].train_stat_global())
            ctxs[r].sync()
        except Exception as e:      # a failing rank must not leave the others waiting for the test's timeout silently
            errors[r] = e

    def all_ranks(n):
        th = [threading.Thread(target=drive, args=(r, n)) for r in range(P)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        bad = [(r, e) for r, e in enumerate(errors) if e is not None]
        if bad:
            raise RuntimeError("rank(s) failed: " + "; ".join(f"{r}: {e}" for r, e in bad))

    if warm_epochs:
        all_ranks(warm_epochs)
    if timing:
        for c in ctxs:
            c.timing_reset()
            c.timing_enable(True)
    all_ranks(epochs)
    out = {"tensors": [], "weights": [], "wgrads": [], "epoch_ms": [np.asarray(m) for m in epoch_ms], "views": [p.view() for p in parts_objs]}
    L = len(dims) - 1
    wnames = wnames or (("w", "a_i") if gnn == da.GAT else ("w",))
    for r, c in enumerate(ctxs):
        vw = out["views"][r]
        N, Gs, Gd = int(vw["localVtxCnt"]), int(vw["srcGhostCnt"]), int(vw["dstGhostCnt"])
        skip = lambda nm: (nm in ("fg", "fg_z") and not Gs) or (nm in ("bg", "bg_d") and not Gd)
        out["tensors"].append({(l, nm): c.download(l, nm) for (l, nm) in downloads if not skip(nm)} if N else {})
        out["weights"].append([{nm: c.weight_get(l, nm) for nm in wnames} for l in range(L)])
        out["wgrads"].append([{nm: c.weight_grad_get(l, nm) for nm in wnames} for l in range(L)])
    if timing:
        tm = {f: [0.0, 0] for f in TIMING_FAMS}
        for c in ctxs:
            for f in TIMING_FAMS:
                ms, n = c.timing_get(f)
                tm[f][0] += ms
                tm[f][1] += n
        out["timing"] = {f: {"ms": round(v[0], 4), "launches": v[1]} for f, v in tm.items()}
    out["stats"] = stats
    out["gates"] = [{"timeouts": int(c.get_option("spmm_gate_timeouts")), "ungated_launches": int(c.get_option("spmm_ungated_launches"))}
                    for c in ctxs]
    for e in engs:
        e.close()
    for c in ctxs:
        c.sync()
    for c in ctxs:
        c.close()
    return out
