"""N>1 path on CPU: world_size-2 (and 3) `gloo` processes run the halo exchange the way
the GPU ranks do -- per-peer send lists from the partition, all-to-all-v of packed rows,
scatter into ghost slots by the host-side receive plan (host/partition.cpp) -- and check
the semantic oracle of the reference's scatter/ghostReceiver pair
(gcn_ops.cpp:204-362): fg[slot(gvid)] == owner.h[lvid(gvid)], both directions.
The RCCL calls themselves are replaced by gloo isend/irecv; counts, offsets, list order
and slot maps are the product's."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, seed, ret):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dorylus_amd as da
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        V, E, F = 120, 900, 7
        src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
        parts = rng.integers(0, world, V).astype(np.int32)
        H = rng.standard_normal((V, F)).astype(np.float32)           # "h" / "grad" rows by global id
        part = da.Partition.build(src, dst, parts, rank, world)
        g = part.view()
        local = H[g["localToGlobal"]]
        for direction, ghost_key, list_key in ((0, "srcGhost", "fwdLists"), (1, "dstGhost", "bwdLists")):
            send_lists = g[list_key]
            recv_slots = part.recv_plan(parts, direction)
            # counts must agree pairwise: what I send to q == what q expects from me
            mine = torch.tensor([len(x) for x in send_lists], dtype=torch.int64)
            allc = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allc, mine)
            assert [int(allc[q][rank]) for q in range(world)] == [len(x) for x in recv_slots]
            send = [torch.from_numpy(np.ascontiguousarray(local[np.asarray(l, np.int64)])) for l in send_lists]
            recv = [torch.empty((len(s), F), dtype=torch.float32) for s in recv_slots]
            # grouped point-to-point, the shape of the ncclSend/ncclRecv group in csrc/abi_comm.hip
            reqs = []
            for q in range(world):
                if q == rank:
                    continue
                if len(send_lists[q]):
                    reqs.append(dist.isend(send[q], q))
                if len(recv_slots[q]):
                    reqs.append(dist.irecv(recv[q], q))
            for r_ in reqs:
                r_.wait()
            ghost = np.full((len(g[ghost_key]), F), np.nan, np.float32)
            for q in range(world):
                if len(recv_slots[q]):
                    ghost[recv_slots[q]] = recv[q].numpy()
            assert np.array_equal(ghost, H[g[ghost_key]])             # owner's rows, bit-exact
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_gloo(world):
    import dorylus_amd
    if not os.path.exists(dorylus_amd.LIB_PATH):
        pytest.skip("library not built")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 7 + world, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {r: "ok" for r in range(world)}
