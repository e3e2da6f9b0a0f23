"""N > 1 path on CPU: two gloo ranks shard a unit list, 'process' their shard, gather on rank 0 and restore the order."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_items, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from instarevive_amd import parallel as P
    r, w, _ = P.init_distributed("gloo")
    assert (r, w) == (rank, world)
    mine = P.shard(list(range(n_items)), r, w)
    # stand-in for the per-image path: an image whose pixels encode the unit id
    local = torch.stack([torch.full((4, 6, 3), i, dtype=torch.uint8) for i in mine]) if mine else torch.zeros((0, 4, 6, 3), dtype=torch.uint8)
    out = P.gather_uint8(local, dst=0)
    tmax = P.max_over_ranks(1.0 + rank)
    if rank == 0:
        inv = P.unshard_order(n_items, w)
        ids = [int(out[inv[i], 0, 0, 0]) for i in range(n_items)]
        q.put((ids, tmax, out.shape[0]))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_shard_and_gather():
    for n_items in (5, 4, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
        for p in procs:
            p.start()
        ids, tmax, n = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert ids == list(range(n_items)) and n == n_items and tmax == 2.0


def test_shard_is_a_partition():
    from instarevive_amd.parallel import shard, unshard_order
    items = list(range(11))
    for world in (1, 2, 3, 8):
        parts = [shard(items, r, world) for r in range(world)]
        flat = [i for p in parts for i in p]
        assert sorted(flat) == items
        inv = unshard_order(len(items), world)
        assert [flat[inv[i]] for i in items] == items
