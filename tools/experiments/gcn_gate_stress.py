import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (repo root)
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import dorylus_amd as da, orc, partition_oracle as po
from helpers import make_ctx, rel_err
V, E, P = 240, 2600, int(sys.argv[1]); F = int(sys.argv[2]); rows = int(sys.argv[3])
for rep in range(3):
    rng = np.random.default_rng(17 + rep)
    s, d = rng.integers(0, V, E), rng.integers(0, V, E)
    parts = (rng.permutation(V) % P).astype(np.int64)
    gs = [po.preprocess(s, d, parts, r, P) for r in range(P)]
    X = rng.uniform(-1, 1, (V, F)).astype(np.float32)
    ctxs = []
    for r, g in enumerate(gs):
        c = make_ctx(da, g, [F, 8, 4], V, node_id=r, num_nodes=P, options={"spmm_blk_nb": 8, "spmm_sweep_rows": rows})
        c.upload(0, "x", X[g["localToGlobal"]]); c.upload(0, "fg", X[g["srcGhost"]].reshape(g["srcGhostCnt"], F))
        ctxs.append(c)
    for it in range(3):
        for c in ctxs:
            c.aggregate(0, da.FORWARD)
    errs = []
    for r, (c, g) in enumerate(zip(ctxs, gs)):
        ref = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], X[g["localToGlobal"]], X[g["srcGhost"]].reshape(g["srcGhostCnt"], F))
        errs.append("%.1e" % rel_err(c.download(0, "ah"), ref))
    print(P, F, rows, errs, "timeouts", [c.get_option("spmm_gate_timeouts") for c in ctxs], "ungated", [c.get_option("spmm_ungated_launches") for c in ctxs])
    for c in ctxs: c.close()
