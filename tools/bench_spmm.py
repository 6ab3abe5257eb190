#!/usr/bin/env python3
"""Micro-benchmark of K1 (SpMM) at Reddit scale through the C-ABI.  GPU box only.
  python tools/bench_spmm.py [--scale 1.0] [--F 602 128] [--graph uniform|rmat]
Prints ms/launch, edges/s, gather GB/s (E*ld*4/t) and compulsory GB/s."""
import argparse
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (first: one HIP runtime)
import dorylus_amd as da  # noqa: E402


def synth_csc(N, E, kind, seed=42, window=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if kind == "uniform":
        dst = torch.randint(0, N, (E,), device="cuda", generator=g)
        src = torch.randint(0, window or N, (E,), device="cuda", generator=g, dtype=torch.int32)
    else:  # rmat-like skew: product of uniform powers concentrates on low ids
        u = torch.rand(E, device="cuda", generator=g)
        dst = (u.pow(1.5) * N).long().clamp_(max=N - 1)
        u = torch.rand(E, device="cuda", generator=g)
        src = (u.pow(1.5) * N).to(torch.int32).clamp_(max=N - 1)
        perm = torch.randperm(N, device="cuda", generator=g)
        dst = perm[dst]
        src = perm[src.long()].to(torch.int32)
    deg = torch.bincount(dst, minlength=N)
    ptr = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
    ptr[1:] = torch.cumsum(deg, 0)
    val = torch.rand(E, device="cuda", generator=g) * 0.01
    return ptr.cpu().numpy().astype(np.uint64), src.cpu().numpy().astype(np.uint32), val.cpu().numpy(), int(deg.max())


_FORM = {}


def ctx_form(ctx):
    return _FORM.get("f", "-")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--F", type=int, nargs="+", default=[602, 128])
    ap.add_argument("--graph", default="uniform")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--slabs", type=int, nargs="+", default=[0, 256, 128, 64, 32])
    ap.add_argument("--variants", type=int, nargs="+", default=[0])
    ap.add_argument("--groups", type=int, nargs="+", default=[16])
    ap.add_argument("--nbs", type=int, nargs="+", default=[0])
    ap.add_argument("--forms", type=int, nargs="+", default=[1])
    ap.add_argument("--window", type=int, default=0, help="draw sources from [0, window): L2-resident gather probe")
    ap.add_argument("--opt", nargs="*", default=[], help="extra context options key=value (e.g. spmm_blk_force_split=1)")
    a = ap.parse_args()
    N = 232965
    E = int(114615892 * a.scale)
    t0 = time.time()
    ptr, idx, val, maxdeg = synth_csc(N, E, a.graph, window=a.window)
    print(f"graph {a.graph}: N={N} E={E} maxdeg={maxdeg} gen {time.time()-t0:.1f}s", flush=True)
    g = dict(localVtxCnt=N, srcGhostCnt=0, dstGhostCnt=0, colPtr=ptr, rowIdx=idx, cscVal=val,
             rowPtr=ptr, colIdx=idx, csrVal=val, norm=np.full(N, 0.002, np.float32))
    for F in a.F:
        ctx = da.Context(0)
        ctx.configure(da.GCN, [F, 8, 4], N)
        for kv in a.opt:                      # before the upload: some options shape the blocked adjacency
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ctx.graph_upload(g)
        ctx.preallocate()
        ctx.fill_uniform(0, "x", 1)
        _, _, ld, _ = ctx.info(0, "x")
        comp = E * 8 + 8 * (N + 1) + 4 * N + 4 * F * N + 4 * F * N
        for variant in a.variants:
          for order in ((2, 0) if variant == 0 else (1,)):
           for grp, nbq in ([(16, 0)] if variant == 0 else [(g_ + 100 * f_, n_) for f_ in a.forms for g_ in a.groups for n_ in a.nbs]):
            for slab in (a.slabs if variant == 0 else [0]):
                if slab and slab >= ld:
                    continue
                _FORM["f"] = grp // 100
                grp = grp % 100
                ctx.set_option("spmm_blk_group", grp)
                ctx.set_option("spmm_blk_nb", nbq)
                ctx.set_option("spmm_variant", variant)
                ctx.set_option("spmm_order", order)
                ctx.set_option("spmm_slab", slab)
                ctx.aggregate(0, da.FORWARD)
                ctx.sync()
                ctx.timing_reset()
                ctx.timing_enable(True)
                for _ in range(a.iters):
                    ctx.aggregate(0, da.FORWARD)
                ctx.sync()
                ms, n = ctx.timing_get("spmm")
                ctx.timing_enable(False)
                t = ms / n * 1e-3
                print(f"F={F} ld={ld} variant={variant} form={ctx_form(ctx)} grp={grp} nb={nbq} order={order} slab={slab:4d}: {t*1e3:8.3f} ms  "
                      f"{E/t/1e9:7.2f} Gedge/s  gather {E*ld*4/t/1e12:6.2f} TB/s  "
                      f"compulsory {comp/t/1e12:6.3f} TB/s", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
