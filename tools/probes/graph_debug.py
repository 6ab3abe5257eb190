import faulthandler, sys, os, time
faulthandler.dump_traceback_later(50, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import numpy as np
import torch
import dorylus_amd as da
import partition_oracle as po
def P(*a):
    print(*a, flush=True)
rng = np.random.default_rng(5)
V, E = 300, 900
s, d = rng.integers(0, V, E), rng.integers(0, V, E)
g = po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)
ctx = da.Context(0)
ctx.configure(da.GCN, [64, 16, 4], V)
ctx.graph_upload(g); ctx.preallocate()
ctx.fill_uniform(0, "x", 3, -1.0, 1.0, g["localToGlobal"])
ctx.labels_upload(rng.integers(0, 4, V).astype(np.uint32))
ctx.weights_init_xavier(); ctx.adam_config(0.01)
eng = da.NativeEngine(ctx)
eng.run(1); P("eager ok")
ctx.epoch_graph_begin(); P("begin ok")
ctx.aggregate(0, da.FORWARD); P("agg0")
ctx.apply_vertex(0, da.FORWARD); P("av0")
ctx.aggregate(1, da.FORWARD); P("agg1")
ctx.apply_vertex(1, da.FORWARD); P("av1")
ctx.weight_update(1); P("wu1")
ctx.aggregate(1, da.BACKWARD); P("agg1b")
ctx.apply_vertex(0, da.BACKWARD); P("av0b")
ctx.weight_update(0); P("wu0")
ctx.epoch_graph_end(); P("end ok")
ctx.epoch_graph_launch(1); P("launched")
ctx.sync(); P("synced")
ctx.epoch_graph_launch(3); ctx.sync(); P("3 more ok")
