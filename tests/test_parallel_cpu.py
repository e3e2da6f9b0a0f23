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
    plan = P.GatherPlan(local, dst=0)   # the reusable form bench.py keeps in its timed step: counts exchanged once, one gather per call
    for _ in range(2):
        again = plan.gather(local)
        assert (again is None) == (rank != 0)
        if rank == 0:
            assert torch.equal(again, out)
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


# ---------------------------------------------------------------------------------------------------------------------
# Tile-level sharding of ONE image: parallel.sharded_tiled_process over an oracle-backed engine.
def _tile_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from instarevive_amd import parallel as P
    from tests.golden._det import det_input
    from tests.support.oracle_engine import OracleTileEngine
    torch.set_num_threads(2)
    P.init_distributed("gloo")
    eng = OracleTileEngine(det_input(9, (1, 20, 64), -1, 1))
    imgs = [(det_input(150, (128, 192, 3)) * 255).numpy().astype(np.uint8)]   # latent 16 x 24, tile 8, stride 5: 3 x 5 tiles, snapped edges
    preds, stage1 = P.sharded_tiled_process(eng, imgs)
    if rank == 0:
        q.put((preds[0], stage1[0]))
    else:
        assert preds is None and stage1 is None
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharding_is_bit_identical_to_one_rank():
    """Two (and three) gloo ranks share the 15 tiles of one image: the re-assembled uint8 image equals, bit for bit, both the
    single-rank run of the same five phases and the oracle's process(tiled=True) in one piece."""
    import numpy as np
    from instarevive_amd import parallel as P
    from tests.golden._det import det_input
    from tests.support.oracle_engine import OracleTileEngine
    eng = OracleTileEngine(det_input(9, (1, 20, 64), -1, 1))
    imgs = [(det_input(150, (128, 192, 3)) * 255).numpy().astype(np.uint8)]
    assert eng.count(128, 192) == 15
    nt = torch.get_num_threads()
    torch.set_num_threads(2)   # as the ranks below: the CPU oracle's convolutions sum in a thread-count-dependent order
    try:
        one, one_s1 = P.sharded_tiled_process(eng, imgs, rank=0, world=1)
        ref, ref_s1 = eng.reference(imgs)
    finally:
        torch.set_num_threads(nt)
    assert np.array_equal(one[0], ref[0]) and np.array_equal(one_s1[0], ref_s1[0])
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_tile_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got, got_s1 = q.get(timeout=600)
        for p in procs:
            p.join(timeout=600)
            assert p.exitcode == 0
        assert np.array_equal(got, one[0]) and np.array_equal(got_s1, one_s1[0]), world
    assert one[0].std() > 1.0


def test_row_shards_cover_the_rows_in_whole_blocks():
    from instarevive_amd.parallel import row_shards
    for n_rows, world in ((130560, 8), (130560, 3), (65536, 8), (1024, 8), (256, 3), (128, 2)):
        sh = row_shards(n_rows, world)
        assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == n_rows
        assert all(a % 128 == 0 and b % 128 == 0 and a <= b for a, b in sh)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in sh]
        assert max(sizes) - min(sizes) <= 128


# ---------------------------------------------------------------------------------------------------------------------
# sharded_encode: the ranks must agree on the softmax-overflow fallback (a rank only sees its own rows).
class _FakeShardEngine:
    """Stands in for pipeline.HipTileEngine's two-part encode: 'fast' rows are row % 128, 'fallback' rows -1 - row % 128 (exact in bf16); a rank whose own
    rows contain `bad_row` overflows, recomputes ALL rows by the fallback and reports it (encode_overflow), as ir_tiled_encode_part does."""

    def __init__(self, bad_row):
        self.bad_row, self.calls, self.T = bad_row, [], 512

    def can_shard_encode(self, imgs):
        return True

    def encode(self, imgs):
        raise AssertionError("the replicated encode must not run")

    def _rows(self, slow):
        v = (torch.arange(self.T) % 128).float()
        return (-1.0 - v if slow else v)[:, None].expand(self.T, 4).contiguous().to(torch.bfloat16)

    def encode_part0(self, imgs, r0, r1, force_fallback=False):
        self.calls.append((r0, r1, force_fallback))
        self._over = int(force_fallback or (self.bad_row is not None and r0 <= self.bad_row < r1))
        o = torch.zeros(self.T, 4, dtype=torch.bfloat16)
        if self._over:
            o[:] = self._rows(True)
        else:
            o[r0:r1] = self._rows(False)[r0:r1]
        return torch.zeros(1), o, torch.zeros(1)

    def encode_overflow(self):
        return self._over

    def encode_part1(self, control, attn_o, attn_res):
        return attn_o.float().clone()


def _overflow_worker(rank, world, port, bad_row, forced, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if forced:
        os.environ["IR_FORCE_COLLECTIVES"] = "1"
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from instarevive_amd import parallel as P
    P.init_distributed("gloo")
    assert dist.is_initialized()
    eng = _FakeShardEngine(bad_row)
    _, init = P.sharded_encode(eng, [np.zeros((64 * 8, 8 * 8, 3), np.uint8)], rank, world)   # 64 x 8 latent = 512 tokens
    q.put((rank, init[:, 0].tolist(), eng.calls))
    if forced and world == 1:   # the other exchange steps through the (one-rank) collective as well
        t = torch.arange(24, dtype=torch.float32).reshape(3, 2, 4)
        assert torch.equal(P._exchange_tiles(t, 3, 0, 1, to_all=True), t) and torch.equal(P._exchange_tiles(t, 3, 0, 1, to_all=False), t)
        plan = P.GatherPlan(torch.zeros(2, 4, 4, 3, dtype=torch.uint8))
        assert plan.on and torch.equal(plan.gather(torch.ones(2, 4, 4, 3, dtype=torch.uint8)), torch.ones(2, 4, 4, 3, dtype=torch.uint8))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_encode_ranks_agree_on_the_overflow_fallback():
    fast = [float(r % 128) for r in range(512)]
    slow = [-1.0 - r % 128 for r in range(512)]
    for world, bad_row, forced in ((2, None, False), (2, 300, False), (2, 5, False), (1, None, True), (1, 7, True)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_overflow_worker, args=(r, world, port, bad_row, forced, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for rank, rows, calls in res:
            assert rows == (fast if bad_row is None else slow), (world, bad_row, rank)
            mine = bad_row is not None and calls[0][0] <= bad_row < calls[0][1]
            # a rank whose own rows did not overflow repeats part 0 with the fallback forced; nobody else does
            assert [c[2] for c in calls] == ([False] if (bad_row is None or mine) else [False, True]), (world, bad_row, rank, calls)
