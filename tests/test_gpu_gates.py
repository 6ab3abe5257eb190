"""K1s gates when their residency assumption is false.  A sweep = the G workgroups of an XCD that K1s keeps in step
assumes G whole CUs per XCD; while dory_debug_occupy_cus holds 96 CUs with a sleeping kernel on the comm stream (what a
co-tenant's kernel, or RCCL holding more CUs than reserved, does) fewer are there.  Required behaviour: results
bit-identical to the unmasked run (placement is for speed only), the polling gives up within its bound, the timeout is
COUNTED (dory_get_option "spmm_gate_timeouts" / "spmm_ungated_launches"), the context backs off -- later launches do
not pay the timeout again -- and a context on a normal stream counts nothing."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(V, E, seed):
    from helpers import random_graph
    import partition_oracle as po
    s, d = random_graph(seed, V, E)
    return po.preprocess(s, d, np.zeros(V, np.int64), 0, 1)


def _ctx(da, g, V, F, stream=None):
    ctx = da.Context(0)
    ctx.configure(da.GCN, [F, 8, 4], V)
    ctx.set_option("spmm_variant", 2)
    ctx.set_option("spmm_blk_nb", 12)        # a graph this small would take K1: force the sweep over 12 source blocks
    if stream is not None:
        ctx.set_streams(compute=stream)
    ctx.graph_upload(g)
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 3)
    return ctx


@pytest.mark.timeout(300)
def test_gate_timeouts_counted_and_results_unchanged_under_cu_mask():
    import dorylus_amd as da
    V, E, F = 60000, 1500000, 128
    g = _graph(V, E, 5)
    # reference run on the context's own stream: gates work, nothing is counted
    ctx = _ctx(da, g, V, F)
    for _ in range(3):
        ctx.aggregate(0, da.FORWARD)
    ref = ctx.download(0, "ah")
    assert ctx.get_option("spmm_gate_timeouts") == 0 and ctx.get_option("spmm_ungated_launches") == 0
    ctx.close()
    # the same while a sleeping kernel holds 96 of the 256 CUs (12 per XCD) for the whole sequence
    ctx = _ctx(da, g, V, F)
    ctx.debug_occupy_cus(96, 400000)
    time.sleep(0.05)
    # times below are HIP-event times of the launches themselves (the host clock of a shared box proves nothing); the
    # behaviour under test is in the device counters, the times only bound the polling
    ctx.timing_reset()
    ctx.timing_enable(True)
    ctx.aggregate(0, da.FORWARD)
    timeouts = ctx.get_option("spmm_gate_timeouts")         # (synchronises the compute stream only; the occupier sleeps on)
    first_ms, n_first = ctx.timing_get("spmm")
    assert timeouts >= 1                                    # the sweep's workgroups were not co-resident: counted
    assert n_first == 1 and first_ms < 100.0                # bounded polling: ~3 ms per timed-out gate class, not round 2's 0.1 s per gate
    for _ in range(6):                                      # inside the back-off: no gates, no new timeouts
        ctx.aggregate(0, da.FORWARD)
    assert ctx.get_option("spmm_gate_timeouts") == timeouts
    assert ctx.get_option("spmm_ungated_launches") >= 6
    all_ms, n_all = ctx.timing_get("spmm")
    ctx.timing_enable(False)
    assert n_all == 7 and (all_ms - first_ms) / 6 < 50.0    # ungated launches run at the unsynchronised rate, nobody polls
    assert np.array_equal(ctx.download(0, "ah"), ref)       # same bits whatever the placement (waits for the occupier)
    tot_ms, n = ctx.timing_get("spmm_gate_timeouts")        # the same counters through dory_timing_get
    assert n == timeouts and tot_ms >= 6
    ctx.sync()                                              # the occupier is gone
    for _ in range(20):                                     # past the back-off horizon the gates are tried again and hold
        ctx.aggregate(0, da.FORWARD)
    assert np.array_equal(ctx.download(0, "ah"), ref)
    assert ctx.get_option("spmm_gate_timeouts") == timeouts
    assert ctx.get_option("spmm_ungated_launches") <= 17
    ctx.close()


@pytest.mark.timeout(300)
def test_gate_backoff_of_launches_beside_an_exchange_is_a_class_of_its_own():
    """A timeout of a launch that runs beside an exchange (flag 32: CUs left to RCCL kernels, which may hold more than
    reserved) must not switch off the gates of the launches that run alone -- the ghost-block launch right behind it waits
    for the exchange and has the chip to itself -- and the other way round."""
    import dorylus_amd as da
    V, E, F = 60000, 1500000, 128
    g = _graph(V, E, 5)
    ctx = _ctx(da, g, V, F)
    ctx.aggregate(0, da.FORWARD)
    ref = ctx.download(0, "ah")
    ctx.set_option("spmm_sweep_flags", 32)                  # what launch_spmm_sweep sets when an exchange is in flight
    ctx.debug_occupy_cus(96, 150000)
    time.sleep(0.01)
    ctx.aggregate(0, da.FORWARD)
    timeouts = ctx.get_option("spmm_gate_timeouts")
    assert timeouts >= 1
    ctx.sync()                                              # the occupier is gone
    ctx.set_option("spmm_sweep_flags", 0)
    ungated = ctx.get_option("spmm_ungated_launches")
    for _ in range(4):                                      # launches that run alone: gated, although inside the other class's back-off
        ctx.aggregate(0, da.FORWARD)
    assert ctx.get_option("spmm_ungated_launches") == ungated and ctx.get_option("spmm_gate_timeouts") == timeouts
    ctx.set_option("spmm_sweep_flags", 32)
    ctx.aggregate(0, da.FORWARD)                            # the class that timed out is still backing off
    assert ctx.get_option("spmm_ungated_launches") == ungated + 1
    assert np.array_equal(ctx.download(0, "ah"), ref)
    ctx.close()


def test_sweep_knobs_are_per_context():
    """spmm_sweep_rows / spmm_sweep_pair were process-wide in round 2: a second context must not inherit them."""
    import dorylus_amd as da
    V, E, F = 20000, 400000, 128
    g = _graph(V, E, 6)
    a = _ctx(da, g, V, F)
    a.aggregate(0, da.FORWARD)
    ref = a.download(0, "ah")
    b = da.Context(0)
    b.configure(da.GCN, [F, 8, 4], V)
    b.set_option("spmm_variant", 2)
    b.set_option("spmm_blk_nb", 12)
    b.set_option("spmm_sweep_rows", 4)
    b.set_option("spmm_sweep_pair", 1)
    b.graph_upload(g)
    b.preallocate()
    b.fill_uniform(0, "x", 3)
    b.aggregate(0, da.FORWARD)
    from helpers import rel_err
    assert rel_err(b.download(0, "ah"), ref) < 1e-5         # another deal of the rows: same sums to reassociation
    assert a.get_option("spmm_sweep_rows") == 0 and a.get_option("spmm_sweep_pair") == -1
    a.aggregate(0, da.FORWARD)
    assert np.array_equal(a.download(0, "ah"), ref)         # context a still runs its own layout
    a.close()
    b.close()


@pytest.mark.parametrize("nb", [2, 3, 5, 12])
def test_loader_wave_same_bits_as_without_it(nb):
    """The loader wave only changes how entries and offsets reach LDS (and, through its relief, which position a row
    takes): a row's sum is formed in the same order either way -- block order, edge order inside a block -- so K1s with and
    without it must agree bit for bit, on a partition with ghost rows (two launches, the ghost one possibly a single
    block), few and many blocks, F with a ragged last slab, and with the oracle to 1e-5."""
    import dorylus_amd as da
    import orc
    from helpers import random_graph, rel_err
    V, P, F = 30000, 2, 200
    s, d = random_graph(77 + nb, V, 500000)
    parts = (np.arange(V, dtype=np.int64) * P // V).astype(np.int32)
    part = da.Partition.build(s, d, parts, 1, P)
    g = part.view()
    N, Gs = int(g["localVtxCnt"]), int(g["srcGhostCnt"])
    rng = np.random.default_rng(nb)
    X = rng.uniform(-1, 1, (N, F)).astype(np.float32)
    FG = rng.uniform(-1, 1, (Gs, F)).astype(np.float32)
    ref = orc.aggregate_gcn(g["colPtr"], g["rowIdx"], g["cscVal"], g["norm"], X, FG)
    outs = {}
    for loader in (1, 0):
        ctx = da.Context(0)
        ctx.configure(da.GCN, [F, 8, 4], V)
        ctx.set_option("spmm_variant", 2)
        ctx.set_option("spmm_blk_nb", nb)
        ctx.set_option("spmm_sweep_loader", loader)
        part.upload(ctx)
        ctx.preallocate()
        ctx.upload(0, "x", X)
        ctx.upload(0, "fg", FG)
        ctx.aggregate(0, da.FORWARD)
        outs[loader] = ctx.download(0, "ah")
        ctx.aggregate(0, da.FORWARD)
        assert np.array_equal(ctx.download(0, "ah"), outs[loader])          # run to run
        assert ctx.get_option("spmm_gate_timeouts") == 0
        ctx.close()
    assert rel_err(outs[1], ref) < 1e-5
    assert np.array_equal(outs[1], outs[0])


@pytest.mark.timeout(300)
def test_gate_backoff_ends_under_epoch_graph_replay():
    """The back-off compares the context's K1s launch number with a horizon on the device.  As a kernel argument that
    number was frozen into a recorded epoch (round 3): after one timeout during a replay every recorded launch stayed
    ungated in all later replays.  It now lives on the device and the last workgroup of a launch bumps it, so replays
    advance it: a timeout costs SWEEP_BACKOFF launches without gates, then the recorded launches gate again."""
    import dorylus_amd as da
    V, E = 60000, 1500000
    g = _graph(V, E, 5)
    ctx = da.Context(0)
    ctx.configure(da.GCN, [128, 64, 8], V)
    ctx.set_option("spmm_variant", 2)
    ctx.set_option("spmm_blk_nb", 12)
    ctx.graph_upload(g)
    ctx.preallocate()
    ctx.fill_uniform(0, "x", 3)
    ctx.labels_upload(np.random.default_rng(1).integers(0, 8, V).astype(np.uint32))
    ctx.weights_init_xavier()
    ctx.adam_config(0.01)
    eng = da.NativeEngine(ctx)
    eng.run(1)                                              # eager: every lazily sized buffer exists
    assert ctx.get_option("spmm_gate_timeouts") == 0
    ctx.epoch_graph_begin()
    _one_epoch(ctx, da)
    ctx.epoch_graph_end()
    ctx.epoch_graph_launch(2)
    ctx.sync()
    assert ctx.get_option("spmm_gate_timeouts") == 0 and ctx.get_option("spmm_ungated_launches") == 0
    ctx.debug_occupy_cus(96, 300000)                        # a co-tenant holds 12 CUs per XCD while one epoch replays
    time.sleep(0.05)
    ctx.epoch_graph_launch(1)
    timeouts = ctx.get_option("spmm_gate_timeouts")
    assert timeouts >= 1
    ctx.sync()                                              # the occupier is gone
    ctx.epoch_graph_launch(12)                              # 3 K1s launches per epoch: far past the 16-launch back-off
    u1 = ctx.get_option("spmm_ungated_launches")
    ctx.epoch_graph_launch(6)
    assert ctx.get_option("spmm_ungated_launches") == u1    # the recorded launches gate again
    assert ctx.get_option("spmm_gate_timeouts") == timeouts
    assert 1 <= u1 <= 2 * 17 + 3                            # (one back-off per class that timed out, plus the launch that did)
    eng.close()
    ctx.close()


def _one_epoch(ctx, da):
    """the stage calls of one 2-layer GCN epoch in Engine::runEpoch's order (what dory_engine_run issues)"""
    ctx.aggregate(0, da.FORWARD)
    ctx.apply_vertex(0, da.FORWARD)
    ctx.aggregate(1, da.FORWARD)
    ctx.apply_vertex(1, da.FORWARD)
    ctx.weight_update(1)
    ctx.aggregate(1, da.BACKWARD)
    ctx.apply_vertex(0, da.BACKWARD)
    ctx.weight_update(0)


def test_xcd_placement_check_and_measured_fallback():
    """K1s assumes workgroup id & 7 = XCD and eight XCDs.  dory_create checks it (HW_REG_XCC_ID of 2 048 probe workgroups); on
    this device the check must hold.  When it does not (forced here by option), the first repeatable K1s launch runs gated
    and ungated, the faster form is kept (spmm_xcd_policy 0 / 8), nothing is counted as a gate timeout, and the bits are
    those of the normal context."""
    import dorylus_amd as da
    V, E, F = 60000, 1500000, 128
    g = _graph(V, E, 9)
    ctx = _ctx(da, g, V, F)
    assert ctx.get_option("spmm_xcd_mapping_ok") == 1 and ctx.get_option("spmm_xcd_count") == 8
    ctx.aggregate(0, da.FORWARD)
    assert ctx.get_option("spmm_xcd_policy") == -1          # nothing to decide
    ref = ctx.download(0, "ah")
    ctx.close()
    ctx = _ctx(da, g, V, F)
    ctx.set_option("spmm_xcd_assume_mismatch", 1)
    for _ in range(3):
        ctx.aggregate(0, da.FORWARD)
    pol = ctx.get_option("spmm_xcd_policy")
    assert pol in (0, 8)
    assert ctx.get_option("spmm_xcd_gated_us") > 0 and ctx.get_option("spmm_xcd_ungated_us") > 0
    # the choice follows the measurement
    assert (pol == 0) == (ctx.get_option("spmm_xcd_gated_us") <= ctx.get_option("spmm_xcd_ungated_us"))
    assert np.array_equal(ctx.download(0, "ah"), ref)
    assert ctx.get_option("spmm_gate_timeouts") == 0 and ctx.get_option("spmm_ungated_launches") == 0
    ctx.close()
