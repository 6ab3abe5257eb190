
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
RTOL = 1e-4
def run(P, dims, heads, sweep, rows=0, sflags=0, serial=False):
    """P partitions (one context each, ghost rows moved by pack / unpack + a device copy, i.e. everything of the
    multi-GPU path but RCCL itself): forward exchange of z, scores of the ghost sources recomputed locally, the
    backward sweep in its two phases with dO and st shipped in between -- against the single-partition float64 oracle."""
    import torch
    import dorylus_amd as da
    import gat_mh_oracle as go
    import partition_oracle as po
    from halo_plan_ref import halo_plan
    from helpers import rel_err
    V, E = 240, 2600
    rng = np.random.default_rng(17)
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    parts = (rng.permutation(V) % P).astype(np.int64)                 # scattered ownership: many ghosts
    g_all = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
    gs = [po.preprocess(s, d, parts, r, P) for r in range(P)]
    X = rng.uniform(-1, 1, (V, dims[0])).astype(np.float32)
    labels = rng.integers(0, dims[-1], V).astype(np.uint32)
    params = []
    for l in range(2):
        zw = dims[l + 1] * (heads[l] if l == 1 else 1)
        params.append([(rng.standard_normal((dims[l], zw)) / np.sqrt(dims[l])).astype(np.float32),
                       (rng.standard_normal(zw) * 0.3).astype(np.float32),
                       (rng.standard_normal(zw) * 0.3).astype(np.float32)])
    ctxs, plans = [], []
    for r, g in enumerate(gs):
        ctx = da.Context(0)
        ctx.configure(da.GATMH, dims, V, r, P)
        ctx.gatmh_heads(heads)
        ctx.set_option("spmm_blk_nb", 8)
        ctx.set_option("gatmh_sweep", sweep)
        ctx.set_option("gatmh_sweep_rows", rows)
        ctx.set_option("spmm_sweep_flags", sflags)
        ctx.graph_upload(g)
        ctx.preallocate()
        ctx.upload(0, "h", X[g["localToGlobal"]])
        ctx.labels_upload(labels[g["localToGlobal"]])
        for l, (W, al, ar) in enumerate(params):
            ctx.weight_set(l, "w", W); ctx.weight_set(l, "a_l", al); ctx.weight_set(l, "a_r", ar)
        pl = halo_plan(g, parts, r, P)
        for dd in (0, 1):
            ctx.halo_plan(dd, pl[dd][0], pl[dd][1])
        ctxs.append(ctx)
        plans.append(pl)

    def exchange(layer, src_name, ghost_name, dd):
        _, _, ld, _ = ctxs[0].info(layer, src_name)
        send = [torch.zeros(max(1, sum(len(x) for x in plans[r][dd][0])) * ld, device="cuda") for r in range(P)]
        recv = [torch.zeros(max(1, sum(len(x) for x in plans[r][dd][1])) * ld, device="cuda") for r in range(P)]
        torch.cuda.synchronize()   # (the contexts' streams are non-blocking: torch's zero fills must have landed before a pack kernel writes)
        for r in range(P):
            ctxs[r].halo_pack_tensor(layer, src_name, dd, send[r].data_ptr())
            ctxs[r].sync()
        for r in range(P):
            soff = np.concatenate([[0], np.cumsum([len(x) for x in plans[r][dd][0]])])
            for p in range(P):
                roff = np.concatenate([[0], np.cumsum([len(x) for x in plans[p][dd][1]])])
                n = len(plans[r][dd][0][p])
                recv[p][roff[r] * ld:(roff[r] + n) * ld] = send[r][soff[p] * ld:(soff[p] + n) * ld]
        torch.cuda.synchronize()
        for r in range(P):
            ctxs[r].halo_unpack_tensor(layer, ghost_name, dd, recv[r].data_ptr())
            ctxs[r].sync()

    L = 2
    for l in range(L):
        for c in ctxs:
            c.apply_vertex(l, da.FORWARD)
        exchange(l, "z", "fg_z", da.FORWARD)
        for c in ctxs:
            c.apply_edge(l + 1, da.FORWARD)
            c.aggregate(l + 1, da.FORWARD)
            if serial:
                c.sync()
        if l == 1 and P == 4 and not serial:
            for c in ctxs: c.sync()
            conc = [(c.download(1, "o").copy(), c.download(1, "den").copy(), c.download(1, "m").copy()) for c in ctxs]
            for c in ctxs:
                c.aggregate(l + 1, da.FORWARD); c.sync()
            for r, c in enumerate(ctxs):
                o2, d2, m2 = c.download(1, "o"), c.download(1, "den"), c.download(1, "m")
                badrows = np.where(np.abs(conc[r][0] - o2).max(1) > 1e-6)[0]
                print("rank", r, "rows differing concurrent vs serial rerun:", len(badrows), list(badrows[:20]), "den diff rows", int((np.abs(conc[r][1]-d2).max(1) > 1e-6).sum()), "m diff rows", int((np.abs(conc[r][2]-m2).max(1) > 0).sum()))
    for c in ctxs:
        c.predict_gat(L)
    for l in range(L - 1, -1, -1):
        for c in ctxs:
            c.set_option("gatmh_bwd_phase", 1)
            c.aggregate(l + 1, da.BACKWARD)
        exchange(l, "do", "bg_do", da.BACKWARD)
        exchange(l, "st", "bg_st", da.BACKWARD)
        for c in ctxs:
            c.set_option("gatmh_bwd_phase", 2)
            c.aggregate(l + 1, da.BACKWARD)
            c.apply_vertex(l, da.BACKWARD)

    fws, Hs, loss, dlogits, grads = go.epoch(g_all, X, labels, [[p.astype(np.float64) for p in ps] for ps in params], heads)

    def gathered(layer, name):
        out = None
        for r, g in enumerate(gs):
            t = ctxs[r].download(layer, name)
            if out is None:
                out = np.zeros((V, t.shape[1]), np.float32)
            out[g["localToGlobal"]] = t
        return out


    for l in range(L):
        for nm, ref in (("z", fws[l]["Z"]), ("o", fws[l]["O"]), ("t", grads[l]["t"]), ("del", grads[l]["d_el"]), ("der", grads[l]["d_er"]), ("dz", grads[l]["dZ"]), ("m+lden", None), ("do", None)):
            if ref is None: continue
            got = gathered(l, nm)
            per_rank = [float(np.abs(got[g["localToGlobal"]] - ref[g["localToGlobal"]]).max() / np.abs(ref).max()) for g in gs]
            if rel_err(got, ref) > 1e-3: print(P, sweep, rows, sflags, serial, l, nm, "%.2e" % rel_err(got, ref), ["%.1e" % x for x in per_rank])
        lse = np.log(gathered(l, "den").astype(np.float64)) + gathered(l, "m")
        if 0: print("   lse err", np.abs(lse - (np.log(fws[l]["den"]) + fws[l]["m"])).max())
    l = 1
    got, ref = gathered(l, "o"), fws[l]["O"]
    if rel_err(got, ref) > 1e-3:
        K = heads[l]
        el = fws[l]["el"]; er = fws[l]["er"]
        for r, g in enumerate(gs):
            N = int(g["localVtxCnt"])
            l2g = g["localToGlobal"]
            ptr = g["colPtr"].astype(np.int64); idx = g["rowIdx"].astype(np.int64)
            allg = np.concatenate([l2g, np.asarray(g["srcGhost"], np.int64)])
            mg = ctxs[r].download(l, "m"); dg = ctxs[r].download(l, "den")
            nbad = 0
            for v in range(N):
                gv = l2g[v]
                e = np.abs(got[gv] - ref[gv]).max()
                if e < 1e-4: continue
                nbad += 1
                src = allg[idx[ptr[v]:ptr[v+1]]]
                loc = idx[ptr[v]:ptr[v+1]] < N
                sc = el[src] + er[gv][None, :]; sc = np.where(sc > 0, sc, 0.2 * sc)
                sself = el[gv] + er[gv]; sself = np.where(sself > 0, sself, 0.2 * sself)
                p = np.exp(sc - mg[v][None, :])
                if nbad <= 4 and r in (1, 2):
                    print("rank", r, "row", v, "err %.2e" % e, "den_gpu", dg[v][:K], "self", np.exp(sself - mg[v]), "local", p[loc].sum(0), "ghost", p[~loc].sum(0), "nl", loc.sum(), "ng", (~loc).sum(), "ghost ids", idx[ptr[v]:ptr[v+1]][~loc] - N)
            print("rank", r, "bad rows", nbad, "of", N, "Gsrc", len(g["srcGhost"]))
    print("gate timeouts", [c.get_option("spmm_gate_timeouts") for c in ctxs], "ungated", [c.get_option("spmm_ungated_launches") for c in ctxs])
    for c in ctxs:
        c.close()

sweep = int(sys.argv[1]); serial = bool(int(sys.argv[2])); sflags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for P in (2, 4):
    for dims, heads in (([24, 32, 8], [4, 2]), ([24, 128, 6], [8, 1])):
        run(P, dims, heads, sweep, serial=serial, sflags=sflags)
print("done", sweep, serial)
