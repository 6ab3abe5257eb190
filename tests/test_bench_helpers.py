"""bench.py's host-side helpers (no GPU): the per-rank record generator for configs 4 / 5 must describe ONE global graph
whichever rank asks, and every rank must get all records incident to its block."""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_incident_records_of_all_ranks_are_one_graph():
    import bench
    V, E, P = 1003, 40000, 4
    per_rank = [bench.synth_incident_edges(V, E, r, P) for r in range(P)]
    blk = lambda v: v.astype(np.int64) * P // V                       # noqa: E731  (bench.py's contiguous blocks)
    union = collections.Counter()
    for r, (s, d) in enumerate(per_rank):
        assert s.dtype == np.uint32 and s.shape == d.shape
        assert np.all((blk(s) == r) | (blk(d) == r))                  # nothing a rank does not need
        lo, hi = bench.block_bounds(V, P, r)
        assert np.all(blk(np.arange(lo, hi, dtype=np.uint32)) == r)   # block_bounds agrees with the parts vector
        # records between two blocks appear in both ranks' lists: count each once (from the lower rank)
        for a, b in zip(s.tolist(), d.tolist()):
            if min(int(blk(np.uint32(a))), int(blk(np.uint32(b)))) == r:
                union[(a, b)] += 1
    total = sum(union.values())
    assert abs(total - E) < 0.02 * E                                  # ~E directed records overall
    for r, (s, d) in enumerate(per_rank):
        mine = collections.Counter(zip(s.tolist(), d.tolist()))
        want = collections.Counter({k: c for k, c in union.items()
                                    if int(k[0]) * P // V == r or int(k[1]) * P // V == r})
        assert mine == want                                           # exactly the global graph's records at this block
    # symmetric: every record has its reverse (the generator emits undirected pairs in both directions)
    assert all(union[(b, a)] == c for (a, b), c in union.items())


def test_incident_records_expected_in_edge_count():
    import bench
    V, E, P = 50000, 2000000, 8
    s, d = bench.synth_incident_edges(V, E, 0, P)
    lo, hi = bench.block_bounds(V, P, 0)
    nnz_in = int(((d >= lo) & (d < hi)).sum())
    assert abs(nnz_in - E / P) < 0.02 * E / P                         # the block's in-edges are its share of the graph
