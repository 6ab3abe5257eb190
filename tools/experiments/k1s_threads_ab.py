#!/usr/bin/env python3
"""Round 4, experiment 1: K1s with 20 waves per CU (96 registers, 8 rows per lane group) against the product form (one
1024-thread workgroup, 16 waves, 10 rows).  First as two 640-thread workgroups per CU (never co-resident: residency_probe),
then as five 256-thread workgroups per CU, with and without the loader wave, then as one 768-thread workgroup with 20 rows.
NOT RUNNABLE ON THE PRODUCT BUILD: it needs the option `spmm_sweep_threads` of profiles/r04_k1s_wg_threads_experiment.patch
(apply it to csrc/spmm.hip, ctx.hpp, abi_stages.hip, abi_context.hip, host/sweep_deal.cpp of commit c603f01 and rebuild);
results: profiles/r04_k1s_wg_threads_ab.txt, profiles/r04_experiments.txt item 1.  GPU box only.
  python tools/experiments/k1s_threads_ab.py            # correctness on a partition with ghost rows, then timing A/B"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import dorylus_amd as da  # noqa: E402


def correctness():
    import orc
    from helpers import random_graph, rel_err
    for nb, F, V, E in ((3, 200, 30000, 500000), (12, 602, 60000, 3000000), (0, 128, 120000, 6000000)):
        P = 2
        s, d = random_graph(77 + nb, V, E)
        parts = (np.arange(V, dtype=np.int64) * P // V).astype(np.int32)
        part = da.Partition.build(s, d, parts, 1, P)
        g = part.view()
        N, Gs = int(g["localVtxCnt"]), int(g["srcGhostCnt"])
        rng = np.random.default_rng(nb)
        X = rng.uniform(-1, 1, (N, F)).astype(np.float32)
        FG = rng.uniform(-1, 1, (Gs, F)).astype(np.float32)
        ref = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], X, FG)
        outs = {}
        for nt in (256, 1024, 768):
            ctx = da.Context(0)
            ctx.configure(da.GCN, [F, 8, 4], V)
            ctx.set_option("spmm_variant", 2)
            ctx.set_option("spmm_blk_nb", nb)
            ctx.set_option("spmm_sweep_threads", abs(nt))
            ctx.set_option("spmm_sweep_loader", 1 if nt > 0 else 0)     # (-256: five workgroups per CU without the loader wave)
            ctx.set_option("spmm_sweep_window_kb", 2432)
            part.upload(ctx)
            ctx.preallocate()
            ctx.upload(0, "x", X)
            ctx.upload(0, "fg", FG)
            ctx.aggregate(0, da.FORWARD)
            outs[nt] = ctx.download(0, "ah")
            ctx.aggregate(0, da.FORWARD)
            again = np.array_equal(ctx.download(0, "ah"), outs[nt])
            print(f"nb={nb} F={F} V={V} threads={nt}: rel_err vs oracle {rel_err(outs[nt], ref):.2e} run-to-run identical {again} "
                  f"gate timeouts {ctx.get_option('spmm_gate_timeouts')} ungated {ctx.get_option('spmm_ungated_launches')}", flush=True)
            ctx.close()
        print(f"   256 == 1024 bit for bit: {np.array_equal(outs[256], outs[1024])}; 768: {np.array_equal(outs[768], outs[1024])}", flush=True)


def timing(rounds=2, iters=5):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_spmm import synth_csc
    N, E = 232965, 114615892
    ptr, idx, val, _ = synth_csc(N, E, "uniform")
    g = dict(localVtxCnt=N, srcGhostCnt=0, dstGhostCnt=0, colPtr=ptr, rowIdx=idx, cscVal=val,
             rowPtr=ptr, colIdx=idx, csrVal=val, norm=np.full(N, 0.002, np.float32))
    for F in (602, 128):
        ctxs = {}
        for nt in (256, 1024, 768):
            ctx = da.Context(0)
            ctx.configure(da.GCN, [F, 8, 4], N)
            ctx.set_option("spmm_sweep_threads", abs(nt))
            ctx.set_option("spmm_sweep_loader", 1 if nt > 0 else 0)
            ctx.graph_upload(g)
            ctx.preallocate()
            ctx.fill_uniform(0, "x", 1)
            ctx.aggregate(0, da.FORWARD)
            ctx.sync()
            ctxs[nt] = ctx
        ld = ctxs[768].info(0, "x")[2]
        for r in range(rounds):
            for nt in (256, 1024, 768):
                ctx = ctxs[nt]
                ctx.timing_reset()
                ctx.timing_enable(True)
                for _ in range(iters):
                    ctx.aggregate(0, da.FORWARD)
                ctx.sync()
                ms, n = ctx.timing_get("spmm")
                ctx.timing_enable(False)
                t = ms / n * 1e-3
                print(f"F={F} threads={nt} round {r}: {t*1e3:8.3f} ms  gather {E*ld*4/t/1e12:6.2f} TB/s  timeouts "
                      f"{ctx.get_option('spmm_gate_timeouts')} ungated {ctx.get_option('spmm_ungated_launches')}", flush=True)
        a = ctxs[1024].download(0, "ah")
        b = ctxs[768].download(0, "ah")
        print(f"F={F}: 768 == 1024 bit for bit on the full graph: {np.array_equal(a, b)}  max|diff| {np.abs(a-b).max():.3e}", flush=True)
        for c in ctxs.values():
            c.close()


if __name__ == "__main__":
    t0 = time.time()
    correctness()
    timing()
    print(f"done in {time.time()-t0:.0f} s")
