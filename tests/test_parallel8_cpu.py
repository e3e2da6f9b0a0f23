"""8-rank rehearsal of the multi-GPU path without hardware (UNMEASURED ON HARDWARE: no 8-GPU node is available to the builder, the
shapes below only prove the control flow and the bit-identity of the exchange steps): gloo, world = 8, with the shapes BASELINE.json's configs[3] and a
4K --tiled frame produce - GatherPlan over 64 images (8 per rank) and over an uneven 61, sharded_tiled_process over 45 tiles (6-6-6-6-6-5-5-5) and
sharded_encode over 8 / 10 row blocks - each bit-identical to the one-rank result. Reference loop: test_scripts/inference.py:119-153,261."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from tests.test_parallel_cpu import ROOT, _FakeShardEngine, _free_port, _worker

WORLD = 8


def _run(target, args_of_rank, n_results, timeout=900):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=args_of_rank(r, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=timeout) for _ in range(n_results)]
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return res


def test_eight_ranks_gather_64_and_61_items():
    """configs[3]'s shape: 64 images over 8 ranks (8 each, the even gather = one slab), and 61 (five ranks hold 8, three hold 7: the padded gather)."""
    for n_items in (64, 61):
        port = _free_port()
        (ids, tmax, n), = _run(_worker, lambda r, q: (r, WORLD, port, n_items, q), 1)
        assert ids == list(range(n_items)) and n == n_items and tmax == float(WORLD)


def _tile_worker8(rank, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from instarevive_amd import parallel as P
    from tests.golden._det import det_input
    from tests.support.oracle_engine import OracleTileEngine
    torch.set_num_threads(1)
    P.init_distributed("gloo")
    eng = OracleTileEngine(det_input(9, (1, 20, 64), -1, 1))
    imgs = [(det_input(151, (192, 384, 3)) * 255).numpy().astype(np.uint8)]
    mine = len(eng_tiles(eng, 192, 384)[rank::WORLD])
    preds, stage1 = P.sharded_tiled_process(eng, imgs)
    if rank == 0:
        q.put(("img", preds[0], stage1[0]))
    else:
        assert preds is None and stage1 is None
    q.put(("count", rank, mine))
    dist.barrier()
    dist.destroy_process_group()


def eng_tiles(eng, h, w):
    from oracle import glue as oglue
    return oglue.sliding_windows(h // 8, w // 8, eng.tl, eng.sl)


def test_eight_ranks_share_45_tiles():
    """A 4K frame has 45 tiles of 512 px at stride 448 (5 x 9); the same grid at the oracle engine's scale (latent 24 x 48, tile 8, stride 5) over 8
    gloo ranks: five ranks take 6 tiles, three take 5, and the re-assembled uint8 image equals the one-rank run bit for bit."""
    from instarevive_amd import parallel as P
    from tests.golden._det import det_input
    from tests.support.oracle_engine import OracleTileEngine
    eng = OracleTileEngine(det_input(9, (1, 20, 64), -1, 1))
    imgs = [(det_input(151, (192, 384, 3)) * 255).numpy().astype(np.uint8)]
    assert eng.count(192, 384) == 45
    nt = torch.get_num_threads()
    torch.set_num_threads(1)   # as the ranks: the CPU oracle's convolutions sum in a thread-count-dependent order
    try:
        one, one_s1 = P.sharded_tiled_process(eng, imgs, rank=0, world=1)
    finally:
        torch.set_num_threads(nt)
    port = _free_port()
    res = _run(_tile_worker8, lambda r, q: (r, port, q), WORLD + 1)
    counts = sorted(c for kind, *rest in res if kind == "count" for c in [rest[1]])
    assert counts == [5, 5, 5, 6, 6, 6, 6, 6]
    (got, got_s1), = [(a, b) for kind, a, b in res if kind == "img"]
    assert np.array_equal(got, one[0]) and np.array_equal(got_s1, one_s1[0])
    assert one[0].std() > 1.0


class _Engine8(_FakeShardEngine):
    def __init__(self, bad_row, T, replicated_ok=False):
        super().__init__(bad_row)
        self.T, self.replicated_ok = T, replicated_ok

    def encode(self, imgs):
        if not self.replicated_ok:
            raise AssertionError("the replicated encode must not run")
        self.calls.append("replicated")
        return torch.zeros(1), self._rows(False).float()


def _encode_worker8(rank, port, T, bad_row, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(WORLD), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from instarevive_amd import parallel as P
    P.init_distributed("gloo")
    eng = _Engine8(bad_row, T, replicated_ok=T // 128 < WORLD)
    lat_h = T // 8
    _, init = P.sharded_encode(eng, [np.zeros((lat_h * 8, 8 * 8, 3), np.uint8)], rank, WORLD)   # lat_h x 8 latent = T tokens
    q.put((rank, init[:, 0].tolist(), eng.calls))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_sharded_encode_row_blocks():
    """The encoder's mid-block attention rows over 8 ranks: 8 blocks of 128 (one each), 10 blocks (2-2-1-1-1-1-1-1), with and without an overflow on one
    rank's rows; and 4 blocks (fewer than ranks): every rank takes the replicated encode instead of holding an empty shard."""
    from instarevive_amd.parallel import row_shards
    for T, bad_row in ((1024, None), (1280, None), (1280, 700), (512, None)):
        port = _free_port()
        res = sorted(_run(_encode_worker8, lambda r, q: (r, port, T, bad_row, q), WORLD))
        fast = [float(r % 128) for r in range(T)]
        slow = [-1.0 - r % 128 for r in range(T)]
        shards = row_shards(T, WORLD)
        for rank, rows, calls in res:
            assert rows == (fast if bad_row is None else slow), (T, bad_row, rank)
            if T // 128 < WORLD:
                assert calls == ["replicated"]
                continue
            assert (calls[0][0], calls[0][1]) == shards[rank]
            mine = bad_row is not None and calls[0][0] <= bad_row < calls[0][1]
            assert [c[2] for c in calls] == ([False] if (bad_row is None or mine) else [False, True]), (T, bad_row, rank, calls)
        if T == 1280:
            assert sorted(b - a for a, b in shards) == [128] * 6 + [256] * 2
