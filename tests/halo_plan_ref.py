"""Test-side restatement of the halo-exchange plan (the product computes it in host/partition.cpp: dory_partition_upload /
Partition.recv_plan); two GPU tests build their multi-context plans with it.  From a partition's graph.<id>.bin fields +
the .parts vector.

Send side: the per-peer lists the reference stores (forwardGhostsList /
backwardGhostsList, graph/dataloader.cpp:277-297).  Receive side: the reference
looks every incoming row up by global id (ghostReceiverGCN, gcn_ops.cpp:310-318);
because local ids ascend with global id on the sender and ghost slots ascend with
global id on the receiver, peer p's k-th row lands in the k-th ghost slot owned
by p -- a static scatter index.
"""
import numpy as np


def halo_plan(g, parts, node_id, num_nodes):
    parts = np.asarray(parts)
    plan = {}
    for d, (ghost_key, list_key) in enumerate((("srcGhost", "fwdLists"), ("dstGhost", "bwdLists"))):
        owner = parts[g[ghost_key]] if len(g[ghost_key]) else np.zeros(0, np.int64)
        slots = np.arange(len(owner), dtype=np.uint32)
        recv = [slots[owner == p] if p != node_id else np.zeros(0, np.uint32) for p in range(num_nodes)]
        send = [np.asarray(g[list_key][p], np.uint32) for p in range(num_nodes)]
        plan[d] = (send, recv)
    return plan
