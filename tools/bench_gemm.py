#!/usr/bin/env python3
"""Micro-benchmark of K2 (fp32 MFMA GEMM) at Reddit shapes through the C-ABI.  GPU box only.
  python tools/bench_gemm.py [--N 232965] [--dims 602 128 41]
apply_vertex(0, FORWARD)  = NN  ah0 (N x d0) * W0 (d0 x d1) -> z0, h0 = tanh(z0)
apply_vertex(0, BACKWARD) = tanh' + TN  ah0^T (d0 x N) * g0 (N x d1) -> dW0   (split-K + reduce)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401
import dorylus_amd as da  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=232965)
    ap.add_argument("--dims", type=int, nargs=3, default=[602, 128, 41])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--opt", nargs="*", default=[], help="context options key=value, e.g. gemm_persistent=1")
    a = ap.parse_args()
    N, dims = a.N, a.dims
    ptr = np.arange(N + 1, dtype=np.uint64)            # one edge per vertex: the graph does not matter here
    idx = np.arange(N, dtype=np.uint32)
    val = np.ones(N, np.float32)
    g = dict(localVtxCnt=N, srcGhostCnt=0, dstGhostCnt=0, colPtr=ptr, rowIdx=idx, cscVal=val, rowPtr=ptr, colIdx=idx,
             csrVal=val, norm=np.full(N, 0.5, np.float32))
    ctx = da.Context(0)
    ctx.configure(da.GCN, dims, N)
    for kv in a.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    ctx.graph_upload(g)
    ctx.preallocate()
    ctx.weights_init_xavier()
    ctx.labels_upload(np.zeros(N, np.uint32))
    ctx.fill_uniform(0, "ah", 3)
    ctx.fill_uniform(0, "aTg", 4)
    for name, layer, direction, flops in (("NN  %dx%d * %dx%d + tanh" % (N, dims[0], dims[0], dims[1]), 0, da.FORWARD, 2.0 * N * dims[0] * dims[1]),
                                          ("TN  %dx%d^T * %dx%d (split-K)" % (N, dims[0], N, dims[1]), 0, da.BACKWARD, 2.0 * N * dims[0] * dims[1])):
        ctx.apply_vertex(layer, direction)
        ctx.sync()
        ctx.timing_reset()
        ctx.timing_enable(True)
        for _ in range(a.iters):
            ctx.apply_vertex(layer, direction)
        ctx.sync()
        ms, n = ctx.timing_get("gemm")
        ctx.timing_enable(False)
        t = ms / a.iters * 1e-3
        print(f"{name}: {t*1e6:8.1f} us  {flops/t/1e12:6.1f} TFLOP/s  ({flops/t/157e12*100:4.1f} % of the fp32 MFMA peak)  [{n} timed launches]", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
