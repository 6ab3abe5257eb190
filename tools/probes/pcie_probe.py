"""Host<->device cost at the C-ABI boundary (dory_tensor_upload / dory_tensor_download), Reddit shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa
import dorylus_amd as da
N = 232965
ctx = da.Context(0)
ctx.configure(da.GCN, [602, 128, 41], N)
g = dict(localVtxCnt=N, srcGhostCnt=0, dstGhostCnt=0, colPtr=np.zeros(N + 1, np.uint64), rowIdx=np.zeros(0, np.uint32),
         cscVal=np.zeros(0, np.float32), rowPtr=np.zeros(N + 1, np.uint64), colIdx=np.zeros(0, np.uint32),
         csrVal=np.zeros(0, np.float32), norm=np.ones(N, np.float32))
ctx.graph_upload(g)
ctx.preallocate()
X = np.random.default_rng(0).random((N, 602), dtype=np.float32)
for _ in range(3):
    t0 = time.perf_counter(); ctx.upload(0, "x", X); ctx.sync(); t1 = time.perf_counter()
    Z = ctx.download(0, "ah"); t2 = time.perf_counter()
    H = ctx.download(0, "h"); t3 = time.perf_counter()
    print(f"upload x {X.nbytes/1e6:.0f} MB: {(t1-t0)*1e3:.1f} ms ({X.nbytes/(t1-t0)/1e9:.1f} GB/s); "
          f"download ah {Z.nbytes/1e6:.0f} MB: {(t2-t1)*1e3:.1f} ms ({Z.nbytes/(t2-t1)/1e9:.1f} GB/s); "
          f"download h {H.nbytes/1e6:.0f} MB: {(t3-t2)*1e3:.1f} ms", flush=True)
