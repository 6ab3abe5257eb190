"""The destination side of the K1s layout (dorylus_amd/host/sweep_deal.cpp, used by csrc/spmm.hip:build_blocked_sweep)
without a GPU: every item gets exactly one position, the positions are laid out for whole sweeps (8 XCDs x S sweeps x
sweep_tiles workgroups x 32 lane groups x R), only groups of the last sweeps carry fewer rows, every XCD gets the same
share, and groups with the same number of rows carry nearly the same weight (serpentine deal of items sorted by
descending weight)."""
import ctypes

import numpy as np
import pytest

import dorylus_amd._lib as L


def deal(items, R, tiles):
    fn = L.load().dory_sweep_deal
    npos = np.zeros(1, np.uint32)
    assert fn(items, R, tiles, npos.ctypes.data_as(ctypes.c_void_p), None, None) == 0
    cap = np.zeros(max(1, int(npos[0]) // R), np.uint32)
    pos = np.zeros(items, np.uint32)
    assert fn(items, R, tiles, npos.ctypes.data_as(ctypes.c_void_p), cap.ctypes.data_as(ctypes.c_void_p),
              pos.ctypes.data_as(ctypes.c_void_p)) == 0
    return int(npos[0]), cap, pos


@pytest.mark.parametrize("items,R,tiles", [(232965, 10, 32), (232965, 8, 32), (116483, 10, 32), (29121, 4, 32),
                                           (3000, 2, 32), (8, 2, 32), (1, 10, 32), (250000, 10, 28), (1000003, 6, 32)])
def test_deal_structure(items, R, tiles):
    npos, cap, pos = deal(items, R, tiles)
    GS = tiles * 32
    assert npos % (8 * GS * R) == 0 or npos == 8           # whole sweeps on every XCD
    S = npos // (8 * GS * R)
    assert cap.sum() == items and cap.max() <= R
    assert len(np.unique(pos)) == items and pos.max() < npos
    # an item sits in a group at a row index below that group's count
    g, r = pos // R, pos % R
    assert np.all(r < cap[g])
    assert np.array_equal(np.bincount(g, minlength=len(cap)).astype(np.uint32), cap)
    # only the last sweeps are short: per XCD the sweeps' rows per group never increase, and all but the last are full
    capx = cap.reshape(8, S, GS)
    per_sweep_max = capx.max(axis=2)
    assert np.all(per_sweep_max[:, :-1] == R) if S > 1 else True
    assert np.all(capx[:, :-1, :] == R) if S > 1 else True
    # inside the last sweep the groups differ by at most one row; the XCDs get the same share (+-1 row per group overall)
    lastsw = capx[:, -1, :]
    assert lastsw.max() - lastsw.min() <= 1
    per_xcd = capx.sum(axis=(1, 2))
    assert per_xcd.max() - per_xcd.min() <= max(1, GS // 8)
    # the minimum number of group-rows: sum over sweeps of rows per group = ceil(ceil(items / 8) / GS)
    need = -(-(-(-items // 8)) // GS)
    assert int(per_sweep_max.max(axis=0).sum()) == max(1, need)


def test_deal_balances_weight():
    """Poisson-like degrees (a uniform graph) and a skewed set: groups with the same number of rows differ by little."""
    rng = np.random.default_rng(5)
    skew = (rng.pareto(2.0, 60000) * 200 + 20).astype(np.int64)
    skew = np.minimum(skew, 2 * int(skew.mean()) + 1)          # build_blocked_sweep cuts rows above 2 x the mean into pieces
    for w, tol in ((np.sort(rng.poisson(492, 232965))[::-1], 0.02), (np.sort(skew)[::-1], 0.10)):
        items = len(w)
        npos, cap, pos = deal(items, 10, 32)
        gsum = np.bincount(pos // 10, weights=w.astype(np.float64), minlength=len(cap))
        for c in np.unique(cap):
            if c == 0:
                continue
            s = gsum[cap == c]
            assert (s.max() - s.min()) / s.mean() < tol, (c, s.min(), s.max(), s.mean())


# ---- the deal of the loader layouts (round 3): capacities differ on purpose, items go by weight ---------------------------
def deal_weighted(w, R, tiles, relief):
    fn = L.load().dory_sweep_deal_weighted
    w = np.ascontiguousarray(w, np.uint64)
    npos = np.zeros(1, np.uint32)
    assert fn(w.size, w.ctypes.data_as(ctypes.c_void_p), R, tiles, relief, npos.ctypes.data_as(ctypes.c_void_p), None, None) == 0
    cap = np.zeros(max(1, int(npos[0]) // R), np.uint32)
    pos = np.zeros(w.size, np.uint32)
    assert fn(w.size, w.ctypes.data_as(ctypes.c_void_p), R, tiles, relief, npos.ctypes.data_as(ctypes.c_void_p),
              cap.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p)) == 0
    return int(npos[0]), cap, pos


def _weights(kind, n, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        w = rng.poisson(492, n)
    else:   # skewed: a power law with a long tail, a few rows near the split limit (2 x the mean)
        w = np.minimum((rng.pareto(1.3, n) + 1) * 120, 984).astype(np.int64)
    return np.sort(w.astype(np.uint64))[::-1]


@pytest.mark.parametrize("kind", ["uniform", "skewed"])
@pytest.mark.parametrize("items,R,relief", [(232965, 10, 3), (232965, 10, 0), (116483, 10, 3), (29121, 4, 3), (3000, 2, 3),
                                            (250000, 8, 2), (97, 2, 3)])
def test_weighted_deal_relieves_the_loader_groups_and_balances_by_rows(kind, items, R, relief):
    w = _weights(kind, items)
    npos, cap, pos = deal_weighted(w, R, 32, relief)
    GS = 32 * 32
    S = npos // (8 * GS * R)
    assert cap.sum() == items and cap.max() <= R
    assert len(np.unique(pos)) == items and pos.max() < npos
    g, r = pos // R, pos % R
    assert np.all(r < cap[g])
    assert np.array_equal(np.bincount(g, minlength=len(cap)).astype(np.uint32), cap)
    # never a sweep more than without relief
    npos0, cap0, _ = deal_weighted(w, R, 32, 0)
    assert npos == npos0
    # loader groups (lane groups 0 and 1 of a workgroup) never carry more rows than their neighbours of the same sweep
    capx = cap.reshape(8, S, 32, 32)
    assert np.all(capx[..., :2].max(axis=-1) <= capx[..., 2:].max(axis=-1))
    if relief and items >= 100000 and R >= 8:
        full = capx[:, 0]                                  # a full sweep: ordinary groups R rows, loader groups fewer
        assert full[..., 2:].max() == R
        assert full[..., :2].max() < R or np.array_equal(cap, cap0)      # (no relief only where it would have cost a sweep)
    # edges in proportion to the rows: load per row slot is level across the groups of the full sweeps
    load = np.bincount(g, weights=w.astype(np.float64), minlength=len(cap))
    fullmask = (cap >= max(1, R - 4)) if items >= 100000 else (cap > 0)
    per_row = load[fullmask] / cap[fullmask]
    if items >= 100000:
        assert per_row.max() / per_row.mean() < (1.03 if kind == "uniform" else 1.25), (per_row.max(), per_row.mean())
