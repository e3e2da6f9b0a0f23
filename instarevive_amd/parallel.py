"""One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).

The path shards by independent units:
  * images (test_scripts/inference.py:261, one process() per file): every rank runs the full four-stage path on its own images
    with replicated weights, NO data-path collective; finished uint8 images are gathered on rank 0 (gather_uint8) where the caller
    wants the batch in one place (BASELINE.json configs[3]);
  * tiles of one large image under --tiled (inference.py:128-134,139-152): sharded_tiled_process() below, whose only exchange steps
    are one all-gather of latent tiles between the two loops and the final gather of pixel tiles for re-assembly.
Timings are reduced with max_over_ranks."""
import os
from typing import List, Sequence

import torch


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def force_collectives() -> bool:
    """IR_FORCE_COLLECTIVES=1: a ONE-process group still initialises its backend and every exchange step (GatherPlan.gather,
    _exchange_tiles, sharded_encode) goes through the collective instead of the single-rank shortcut. This is how a one-GPU box runs the
    RCCL ("nccl") branch on device tensors (tests/support/rccl_single_rank_worker.py); it changes no result."""
    return os.environ.get("IR_FORCE_COLLECTIVES", "") not in ("", "0")


def init_distributed(backend: str = None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT). No-op for one process."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if (world == 1 and not force_collectives()) or dist.is_initialized():
        return rank, world, local
    if world == 1:   # forced one-process group: the rendezvous the launcher would have set
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return rank, world, local


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin unit assignment: rank r takes items r, r+world, ...."""
    return list(items[rank::world])


def agree_on_list(items: Sequence[str], src: int = 0) -> List[str]:
    """Every rank returns rank `src`'s list (one broadcast of the pickled list). Unit ownership and the pairing of per-image
    collectives both follow from the list, so ranks must not work from listings that differ (directory changed between the ranks'
    walks, different mount views)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(items)
    box = [list(items)]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def unshard_order(n_items: int, world: int) -> List[int]:
    """Position in the rank-major concatenation of every original item index (inverse of `shard` after a gather)."""
    order = [i for r in range(world) for i in range(r, n_items, world)]
    inv = [0] * n_items
    for pos, i in enumerate(order):
        inv[i] = pos
    return inv


def max_over_ranks(value: float, device=None) -> float:
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


class GatherPlan:
    """The gather of per-rank uint8 image batches [n_r, H, W, 3] on rank `dst`, with everything that does not depend on the pixel
    values done ONCE: the per-rank counts n_r (they may differ by one) are exchanged in the constructor and the receive buffers are
    allocated there, so gather() is exactly one RCCL gather of device tensors (<= 12.6 MB per 2048 x 2048 image: latency-, not
    bandwidth-bound) with no host synchronisation - what bench.py keeps inside its timed step."""

    def __init__(self, like: torch.Tensor, dst: int = 0):
        import torch.distributed as dist
        self.on = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives())
        self.dst = dst
        if not self.on:
            return
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.dev = _comm_device(like.device)   # RCCL moves device tensors; gloo (CPU tests, single-GPU rehearsals) gathers host tensors
        counts = [torch.zeros(1, dtype=torch.int64, device=self.dev) for _ in range(self.world)]
        dist.all_gather(counts, torch.tensor([like.shape[0]], dtype=torch.int64, device=self.dev))
        self.counts = [int(c) for c in counts]
        self.shape, self.nmax = tuple(like.shape[1:]), max(self.counts)
        self.even = min(self.counts) == self.nmax
        self.send = None if self.even else torch.zeros((self.nmax,) + self.shape, dtype=torch.uint8, device=self.dev)
        # one receive slab, the gather list = its per-rank slices: with equal counts the slab IS the rank-major concatenation
        self.slab = torch.empty((self.world * self.nmax,) + self.shape, dtype=torch.uint8, device=self.dev) if self.rank == dst else None
        self.bufs = [self.slab[r * self.nmax:(r + 1) * self.nmax] for r in range(self.world)] if self.rank == dst else None

    def gather(self, local: torch.Tensor):
        """-> the rank-major concatenation on rank `dst` (a view of the plan's receive buffers when every rank holds the same
        count), None elsewhere."""
        import torch.distributed as dist
        if not self.on:
            return local
        if local.shape[0] != self.counts[self.rank] or tuple(local.shape[1:]) != self.shape:
            raise ValueError("GatherPlan.gather: batch shape differs from the one the plan was built for")
        src = local if local.device == self.dev else local.to(self.dev)
        if not self.even:
            self.send[: local.shape[0]] = src
            src = self.send
        dist.gather(src.contiguous(), self.bufs, dst=self.dst)
        if self.rank != self.dst:
            return None
        out = self.slab if self.even else torch.cat([b[:c] for b, c in zip(self.bufs, self.counts)], dim=0)
        return out if out.device == local.device else out.to(local.device)


def gather_uint8(local: torch.Tensor, dst: int = 0):
    """One-shot form of GatherPlan (counts exchanged on every call): returns the rank-major concatenation on `dst`, None elsewhere."""
    return GatherPlan(local, dst).gather(local)


# ------------------------------------------------------------------------------------------------ tile sharding of ONE image
def _comm_device(local_device):
    import torch.distributed as dist
    return local_device if dist.get_backend() == "nccl" else torch.device("cpu")


def _exchange_tiles(local: torch.Tensor, n_tiles: int, rank: int, world: int, to_all: bool, dst: int = 0):
    """local: this rank's tiles [k_r, ...] (tiles rank, rank+world, ... of the loop order). Returns all n_tiles tiles in loop order
    [n_tiles, ...] on every rank (to_all, one all_gather) or on rank `dst` only (one gather; None elsewhere). Ranks hold
    ceil/floor(n_tiles / world) tiles, so the buffers are padded to the maximum and the padding is dropped on arrival."""
    import torch.distributed as dist
    if world == 1 and not (force_collectives() and dist.is_initialized()):
        return local
    kmax = (n_tiles + world - 1) // world
    dev = _comm_device(local.device)
    send = torch.zeros((kmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    send[: local.shape[0]] = local.to(dev)
    if to_all:
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(parts, send)
    else:
        parts = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, parts, dst=dst)
        if rank != dst:
            return None
    out = torch.empty((n_tiles,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    for r in range(world):
        out[r::world] = parts[r][: len(range(r, n_tiles, world))]   # tile i sits at index i // world of rank i % world
    return out.to(local.device)


def row_shards(n_rows: int, world: int, quantum: int = 128):
    """Contiguous [row0, row1) per rank over n_rows rows in whole `quantum`-row blocks (n_rows % quantum == 0): blocks dealt as evenly as
    possible, the first ranks take the remainder."""
    blocks = n_rows // quantum
    out, b0 = [], 0
    for r in range(world):
        nb = blocks // world + (1 if r < blocks % world else 0)
        out.append((b0 * quantum, (b0 + nb) * quantum))
        b0 += nb
    return out


def sharded_encode(engine, control_imgs, rank: int, world: int):
    """SwinIR + VAE encode of the frame with the encoder's mid-block attention (T^2 * 512: 35 of the 75 TFLOP every rank used to repeat at
    4K) split over the ranks by query rows; one all_gather of the attention rows. Every row is bit-identical to the unsharded launch's
    (whole 128-query workgroups; an overflow of the fixed softmax reference on any rank sends every rank to the rescaling kernel, as in
    the unsharded launch), hence so is everything behind it. Under IR_FLAG_FP8 the sharded form still runs the bf16 attention. Falls back to the replicated encode when the engine cannot split
    (several images, token count not a multiple of 128, an engine without the two-part encode)."""
    import torch.distributed as dist
    if (world == 1 and not (force_collectives() and dist.is_initialized())) or not hasattr(engine, "encode_part0") or not engine.can_shard_encode(control_imgs):
        return engine.encode(control_imgs)             # replicated: SwinIR and the VAE encoder are untiled in the reference
    h, w = control_imgs[0].shape[:2]
    T = (h // 8) * (w // 8)
    if T // 128 < world:   # fewer 128-row blocks than ranks (a frame below 8 x 128 latent pixels on 8 ranks): a rank would hold no rows - every rank
        return engine.encode(control_imgs)   # sees the same shape and takes the replicated encode
    shards = row_shards(T, world)
    r0, r1 = shards[rank]
    control, attn_o, attn_res = engine.encode_part0(control_imgs, r0, r1)
    kmax = max(b - a for a, b in shards)
    dev = _comm_device(attn_o.device)
    if hasattr(engine, "encode_overflow"):
        # The unsharded launch recomputes EVERY row with the rescaling kernel as soon as any row overflows the fixed softmax reference; a
        # rank only sees its own rows, so the ranks agree on the flag first (MAX). When it is set, every rank holds all rows from the
        # rescaling kernel (its own fallback, or part 0 repeated with the fallback forced) and nothing is exchanged.
        mine = engine.encode_overflow()
        flag = torch.tensor([mine], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            if not mine:
                control, attn_o, attn_res = engine.encode_part0(control_imgs, r0, r1, force_fallback=True)
            return control, engine.encode_part1(control, attn_o, attn_res)
    bits = attn_o.view(torch.uint8)                    # raw bytes: every backend moves uint8 (gloo takes neither bfloat16 nor int16)
    send = torch.zeros((kmax, bits.shape[1]), dtype=torch.uint8, device=dev)
    send[: r1 - r0] = bits[r0:r1].to(dev)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    for r, (a, b) in enumerate(shards):
        if r != rank:
            bits[a:b] = parts[r][: b - a].to(bits.device)
    return control, engine.encode_part1(control, attn_o, attn_res)


def sharded_tiled_process(engine, control_imgs, rank: int = None, world: int = None, dst: int = 0):
    """process(..., tiled=True) of ONE image batch with its tiles sharded over the ranks (test_scripts/inference.py:119-153; SURVEY.md
    section 8(e), "tile-level sharding of one large image"). `engine` supplies the five phases (pipeline.HipTileEngine on a GPU).
    Exchange steps: one all_gather of the encoder's mid-block attention rows (sharded_encode), one all_gather of the x0 latent tiles between the two loops (every rank needs the blended latent for its own
    decoder tiles) and one gather of the decoded pixel tiles on rank `dst`, which re-assembles the image. Both sums run over ALL
    tiles in the reference's loop order on the receiving side, so the result equals the single-rank result bit for bit.
    Returns (preds, stage1_preds) on rank `dst`, (None, None) elsewhere."""
    import torch.distributed as dist
    if rank is None or world is None:
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
    control, init = sharded_encode(engine, control_imgs, rank, world)
    h, w = control.shape[-2:]
    n_tiles = engine.count(h, w)
    x0_all = _exchange_tiles(engine.dit_tiles(init, rank, world), n_tiles, rank, world, to_all=True)
    nb = engine.blend_latent(x0_all)
    px_all = _exchange_tiles(engine.decode_tiles(nb, control, rank, world), n_tiles, rank, world, to_all=False, dst=dst)
    if rank != dst:
        return None, None
    return engine.blend_pixels(px_all), engine.stage1()
