#!/usr/bin/env python3
"""Drop-in for the reference's SR command line (test_scripts/inference.py, documented in README.md:57,63):

    python inference.py --ckpt weights/InstaRevive_v1.ckpt --input DIR --output DIR [--sr_scale F] [--tiled
        --tile_size 512 --tile_stride 448] [--color_fix_type wavelet|adain|none] [--disable_preprocess_model] ...

Same flags, defaults, file naming (`<stem>_<i>.png` under the input's relative path) and artefacts:
  * DiT:      flat state dict in diffusers Transformer2DModel key layout  (--ckpt; the reference parses --ckpt but
              hard-codes ./weights/InstaRevive_v1.ckpt, inference.py:239-240 — here --ckpt is honoured)
  * SwinIR:   ./weights/general_swinir_v1.ckpt with ./configs/swinir.yaml     (override: --swinir_ckpt / --swinir_config)
  * VAE:      diffusers folder 'stabilityai/sd-vae-ft-ema'                     (override: --vae)
  * prompt:   {'caption_embeds': [1,300,4096], 'emb_mask': [1,300]} .pth       (override: --prompt_embeds)
  * scheduler: only alphas_cumprod[400] is consumed                            (override: --dit_config folder)
All compute runs on the MI355X through hand-written HIP kernels; `--device cpu|mps` is rejected (no CPU path).
With torchrun (WORLD_SIZE > 1) the file list is sharded over the ranks, one process per GPU (images are independent); with
--tiled --shard_tiles the tiles of each image are sharded instead (one RCCL all-gather of latent tiles + one gather of pixel tiles).
"""
import math
import os
import threading
import time
from argparse import ArgumentParser, Namespace
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Callable, Iterable, Iterator, List

import numpy as np
import torch
from PIL import Image

DEFAULT_PROMPT = ("./output/tmp/real-world image, realistic, high quality, photograph, film, professional, 4k, "
                  "highly detailed_300token.pth")


def parse_args() -> Namespace:
    parser = ArgumentParser()
    parser.add_argument("--ckpt", required=True, type=str, help="full checkpoint path", default="./weights/InstaRevive_v1.ckpt")
    parser.add_argument("--input", type=str, required=True)
    parser.add_argument("--sr_scale", type=float, default=1)
    parser.add_argument("--repeat_times", type=int, default=1)
    parser.add_argument("--disable_preprocess_model", action="store_true")
    # patch-based sampling
    parser.add_argument("--tiled", action="store_true")
    parser.add_argument("--tile_size", type=int, default=512)
    parser.add_argument("--tile_stride", type=int, default=448)
    # latent image guidance (accepted and inert, like the reference)
    parser.add_argument("--use_guidance", action="store_true")
    parser.add_argument("--g_scale", type=float, default=0.0)
    parser.add_argument("--g_t_start", type=int, default=1001)
    parser.add_argument("--g_t_stop", type=int, default=-1)
    parser.add_argument("--g_space", type=str, default="latent")
    parser.add_argument("--g_repeat", type=int, default=5)
    parser.add_argument("--color_fix_type", type=str, default="wavelet", choices=["wavelet", "adain", "none"])
    parser.add_argument("--output", type=str, required=True)
    parser.add_argument("--show_lq", action="store_true")
    parser.add_argument("--skip_if_exist", action="store_true")
    parser.add_argument("--seed", type=int, default=231)
    parser.add_argument("--device", type=str, default="cuda", choices=["cpu", "cuda", "mps"])
    parser.add_argument("--use_prompt", action="store_true")
    parser.add_argument("--use_center_crop", action="store_true")
    # locations the reference hard-codes
    parser.add_argument("--swinir_ckpt", type=str, default="./weights/general_swinir_v1.ckpt")
    parser.add_argument("--swinir_config", type=str, default="./configs/swinir.yaml")
    parser.add_argument("--vae", type=str, default="stabilityai/sd-vae-ft-ema")
    parser.add_argument("--dit_config", type=str, default="PixArt-alpha/PixArt-Alpha-DMD-XL-2-512x512")
    parser.add_argument("--prompt_embeds", type=str, default=DEFAULT_PROMPT)
    # extensions (defaults reproduce the reference's behaviour)
    parser.add_argument("--batch_size", type=int, default=1, help="consecutive files of equal network-input size per process() call")
    parser.add_argument("--shard_tiles", action="store_true", help="with --tiled under torchrun: spread the TILES of each image over the "
                        "GPUs (one large image at a time) instead of spreading the files")
    parser.add_argument("--fp8", type=str, default="off", choices=["off", "default", "auto", "qualified", "all", "attention"], help="BASELINE.json configs[4]: fp8 (e4m3) MFMA "
                        "operands. default (= auto): the operand set is chosen ON THE LOADED WEIGHTS at start-up - one 512 x 512 calibration image, every part alone "
                        "against the bf16 pass (about a second; cached per weight set) - so that the result stays within 0.1 dB PSNR of the bf16 / reference path up to a "
                        "30 dB reference: on flat-softmax weights that is the attention products + two decoder conv levels (about 15 %% faster), on weights with "
                        "heavy-tailed channels / peaky attention fewer parts or none (instarevive_amd/fp8_select.py). qualified = that flat-softmax set without "
                        "calibration; attention = the attention products only; all = every part (about 22 %% faster, 42 dB against the reference path: out of the "
                        "tolerance). off is the default")
    parser.add_argument("--png_compress_level", type=int, default=None, choices=range(0, 10), metavar="0..9", help="zlib level of the saved PNGs; default: PIL's own "
                        "(6, what the reference writes). The pixels are the same at every level; 1 costs about a third of the encoder time - for ranks whose CPU share "
                        "cannot keep up with the GPU (the CLI says so at start-up)")
    parser.add_argument("--workers", type=int, default=-1, help="host threads that decode / resize the inputs and resize / PNG-encode the results "
                        "around the GPU (PIL releases the GIL there); -1 = this process's CPU share, 0 = everything on the main thread like the reference")
    return parser.parse_args()


def cpu_share() -> int:
    """Cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota when one is set (a GPU box hands each job a
    share - 16 cores per GPU - of a host whose affinity mask still shows every core; PNG encoders beyond the quota only time-slice against
    the thread that launches the kernels)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            with open(path) as f:
                quota, period = parse(f.read())
            if period is None:
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = f.read().strip()
            if quota not in ("max", "-1") and int(quota) > 0 and int(period) > 0:
                cores = min(cores, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return cores


def default_workers(local_world: int = 1) -> int:
    """Host threads for this rank: its part of the CPU share, less the cores kept for the thread that feeds the GPU and the copy engine's callbacks
    (two; one when the rank's part is 8 cores or fewer - 8 ranks on a 64-core host - where the feeding thread, asleep in stream waits most of the time,
    shares a core with an encoder rather than take a quarter of the rank's budget), at most 16 (one MI355X produces ~8 results of 2048 x 2048 a second and
    a PNG of that size costs 0.8 - 1.7 core-seconds; $IR_WORKERS overrides)."""
    if os.environ.get("IR_WORKERS"):
        return max(0, int(os.environ["IR_WORKERS"]))
    part = cpu_share() // max(local_world, 1)
    return max(1, min(16, part - (2 if part > 8 else 1)))


PNG_CORE_SECONDS_2048 = 1.25   # PIL's default encoder (compress_level 6) on one 2048 x 2048 RGB result: 0.8 - 1.7 core-seconds by content (profiles/r05_cli_rate_ab.txt)
GPU_FILES_PER_SECOND_2048 = 8.4   # what one MI355X delivers at 2048 x 2048 (bench.py `value`)


def host_keeps_up(workers: int, out_pixels: int, compress_level) -> str:
    """'' when `workers` encoder threads keep ahead of one GPU for results of out_pixels pixels, else the sentence the CLI prints: the rank is then
    host-bound (8 ranks on a 64-core host: 7 threads encode ~5.6 files/s of 2048 x 2048 at PIL's default level against ~8.4 from the GPU). The estimate
    scales the measured level-6 cost by the pixel count; level 1 costs about a third of it (larger files, the same pixels)."""
    if workers <= 0 or out_pixels <= 0:
        return ""
    scale = out_pixels / float(2048 * 2048)
    cost = PNG_CORE_SECONDS_2048 * scale * (1.0 if compress_level is None or compress_level >= 6 else (0.35 if compress_level <= 1 else 0.6))
    host_rate, gpu_rate = workers / cost, GPU_FILES_PER_SECOND_2048 / scale
    if host_rate >= gpu_rate:
        return ""
    return (f"host-bound: {workers} encoder threads write ~{host_rate:.1f} files/s of {out_pixels / 1e6:.1f} Mpixel against ~{gpu_rate:.1f} from the GPU - give the rank more "
            f"cores (--workers / $IR_WORKERS), or trade file size for speed with --png_compress_level 1 (same pixels, lossless)")


def check_device(device: str) -> str:
    if device != "cuda" or not torch.cuda.is_available():
        raise SystemExit(f"device '{device}' requested / no GPU visible: this build runs on MI355X (ROCm) only and has no CPU or MPS path")
    print(f"using device {device}")
    return device


# ---------------------------------------------------------------------------------------------------------------------
# The reference runs one file at a time through load -> resize -> pad -> process() -> crop -> resize -> save, all on one host thread
# (test_scripts/inference.py:261-346). Here the same per-file arithmetic is cut into three pieces around the GPU call:
#   read_job()    everything before process(): decode, --sr_scale bicubic, auto_resize / centre crop, pad to 64       (:263-291)
#   process_stream()  the GPU path, transfers of neighbouring jobs overlapped with compute
#   write_job()   everything after: un-pad, LANCZOS back to the LQ size, optional LQ | stage-1 | result strip, PNG    (:323-346)
# so that a directory streams through the GPU instead of alternating between PIL and the device.
@dataclass
class Job:
    save_path: str
    lq: Image.Image            # the (sr_scale-d) LQ image: target size of the saved result and left panel of --show_lq
    net_in: np.ndarray         # what process() receives: HWC uint8, edges multiples of 64 (or 512 x 512 under --use_center_crop)
    valid_hw: tuple            # un-padded extent of net_in, () under --use_center_crop (nothing to remove)


def read_job(file_path: str, repeat: int, args: Namespace) -> Job:
    from instarevive_amd.utils import auto_resize, center_crop_arr, get_file_name_parts, pad
    lq = Image.open(file_path).convert("RGB")
    if args.sr_scale != 1:
        lq = lq.resize(tuple(math.ceil(edge * args.sr_scale) for edge in lq.size), Image.BICUBIC)
    if args.use_center_crop and not args.tiled:
        net_in = np.array(center_crop_arr(lq, 512))
    else:
        fitted = auto_resize(lq, args.tile_size if args.tiled else 512)
        net_in = pad(np.array(fitted), scale=64)
    # --use_center_crop switches the un-padding / resize-back of the result off, also next to --tiled (inference.py:326-346)
    valid = () if args.use_center_crop else (fitted.height, fitted.width)
    folder, stem, _ = get_file_name_parts(os.path.join(args.output, os.path.relpath(file_path, args.input)))
    return Job(os.path.join(folder, f"{stem}_{repeat}.png"), lq, net_in, valid)


def write_job(job: Job, pred: np.ndarray, stage1_pred, args: Namespace) -> None:
    def back_to_lq(img):
        if not job.valid_hw:
            return img
        img = img[:job.valid_hw[0], :job.valid_hw[1], :]
        return np.array(Image.fromarray(img).resize(job.lq.size, Image.LANCZOS))

    os.makedirs(os.path.dirname(job.save_path) or ".", exist_ok=True)
    result = back_to_lq(pred)
    if args.show_lq:
        panels = [np.array(job.lq) if job.valid_hw else job.net_in]
        if not args.disable_preprocess_model:
            panels.append(back_to_lq(stage1_pred))
        result = np.concatenate(panels + [result], axis=1)
    lvl = getattr(args, "png_compress_level", None)
    if lvl is None:
        Image.fromarray(result).save(job.save_path)
    else:
        Image.fromarray(result).save(job.save_path, compress_level=lvl)
    print(f"save to {job.save_path}")


class HostPools:
    """The host side of the stream (one main thread around an 8 images/s GPU delivers 0.6 - 1.2 files/s). Readers run
    read_job() ahead of the GPU, writers run write_job() behind it; both keep the reference's per-file arithmetic and file names
    (test_scripts/inference.py:263-291,323-346) - only WHEN a file is decoded or encoded changes, so the PNGs are pixel-identical to the
    one-thread run. Order is preserved on the read side (batches_of() groups CONSECUTIVE files, and the result list pairs with the job
    list by position); writes are independent files and may finish in any order. Both queues are bounded: at most `depth` decoded inputs
    wait for the GPU and at most `depth` results wait for an encoder, so memory stays at a few dozen images whatever the folder size.
    workers = 0 runs everything inline on the caller's thread (the reference's behaviour)."""

    def __init__(self, workers: int):
        self.workers = max(workers, 0)
        self.depth = max(2 * self.workers, 2)
        # reading (decode + bicubic + pad) is ~10x cheaper than writing (LANCZOS + PNG deflate): a quarter of the threads keeps up
        self.readers = ThreadPoolExecutor(max(1, self.workers // 4), thread_name_prefix="ir-read") if self.workers else None
        self.writers = ThreadPoolExecutor(self.workers, thread_name_prefix="ir-write") if self.workers else None
        self.slots = threading.Semaphore(self.depth)
        self.pending = deque()
        self.written = 0

    def read_ahead(self, fn: Callable, items: Iterable) -> Iterator:
        """fn(item) for every item, results in the items' order, up to `depth` calls running or finished ahead of the consumer."""
        if not self.readers:
            for item in items:
                yield fn(item)
            return
        ahead = deque()
        for item in items:
            ahead.append(self.readers.submit(fn, item))
            if len(ahead) >= self.depth:
                yield ahead.popleft().result()
        while ahead:
            yield ahead.popleft().result()

    def write_behind(self, fn: Callable, *a) -> None:
        """fn(*a) on a writer thread; blocks while `depth` writes are outstanding. A failed write is re-raised by the next call / drain()."""
        self.written += 1
        if not self.writers:
            fn(*a)
            return
        while self.pending and self.pending[0].done():
            self.pending.popleft().result()
        self.slots.acquire()

        def run():
            try:
                fn(*a)
            finally:
                self.slots.release()

        self.pending.append(self.writers.submit(run))

    def drain(self) -> None:
        while self.pending:
            self.pending.popleft().result()
        for pool in (self.readers, self.writers):
            if pool:
                pool.shutdown(wait=True)


def batches_of(jobs: Iterable[Job], limit: int) -> Iterator[List[Job]]:
    """Consecutive jobs of equal network-input shape, at most `limit` per batch (limit 1 = the reference's one image per call)."""
    group: List[Job] = []
    for job in jobs:
        if group and (len(group) >= limit or job.net_in.shape != group[0].net_in.shape):
            yield group
            group = []
        group.append(job)
    if group:
        yield group


def load_models(args: Namespace, device: torch.device) -> SimpleNamespace:
    from instarevive_amd.models import AutoencoderKL, DDPMScheduler, Transformer2DModel
    from instarevive_amd.utils import instantiate_from_config, load_state_dict, load_yaml
    m = SimpleNamespace()
    m.noise_scheduler = DDPMScheduler.from_pretrained(args.dit_config, subfolder="scheduler")
    m.vae = AutoencoderKL.from_pretrained(args.vae).to(torch.float32).to(device)
    m.model = Transformer2DModel.from_pretrained(args.dit_config, subfolder="transformer")
    m.model.load_state_dict(torch.load(args.ckpt, map_location="cpu"))
    m.preprocess_model = instantiate_from_config(load_yaml(args.swinir_config))
    load_state_dict(m.preprocess_model, torch.load(args.swinir_ckpt, map_location="cpu"), strict=True)
    m.model.to(device)
    m.preprocess_model.to(device)
    prompt = torch.load(args.prompt_embeds, map_location="cpu")
    embeds = prompt["caption_embeds"]
    m.y = embeds.to(device, torch.float32).reshape(1, -1, embeds.shape[-1])
    # [1,1,L]: a 3-D mask reaches the cross-attention as an ADDITIVE bias (diffusers semantics, inference.py:274-277)
    m.y_mask = prompt["emb_mask"].to(device, torch.float32).reshape(1, 1, -1)
    return m


def main() -> None:
    from instarevive_amd import parallel
    from instarevive_amd.pipeline import HipTileEngine, process_stream
    from instarevive_amd.utils import list_image_files
    args = parse_args()
    torch.manual_seed(args.seed)  # the path is deterministic; kept for surface compatibility (pl.seed_everything)
    args.device = check_device(args.device)
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    m = load_models(args, device)
    if args.fp8 != "off":
        from instarevive_amd import _lib as L
        if args.shard_tiles:
            raise SystemExit("--fp8 is not offered together with --shard_tiles (the sharded encode runs the bf16 attention)")
        m.vae.enable_fp8(True)                                   # packs + uploads the fp8 weight forms ...
        ctx = m.model.ctx
        ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")    # ... the mode itself is switched per call (IR_FLAG_FP8)
        if args.fp8 in ("default", "auto"):   # the operand set these weights allow (calibrated once per weight set, cached)
            from instarevive_amd import fp8_select
            fmask = fp8_select.auto_mask(m.preprocess_model, m.vae, m.model, m.y, m.y_mask, log=lambda t: print(f"[rank {rank}] {t}"))
            m.vae.enable_fp8(True)
            ctx.check(ctx.lib.ir_set_fp8(ctx.h, 0), "ir_set_fp8")
        else:
            fmask = {"qualified": L.FP8_MASK_QUALIFIED, "all": L.FP8_MASK_ALL, "attention": L.FP8_MASK_ATTENTION}[args.fp8]
        ctx.check(ctx.lib.ir_set_fp8_mask(ctx.h, fmask), "ir_set_fp8_mask")
        if fmask == 0:
            print(f"[rank {rank}] fp8: no operand part holds the tolerance on these weights - running bf16 throughout")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if os.environ.get("IR_SWITCH_INTERVAL"):   # experiment knob: how long a worker thread may keep the GIL while the thread that feeds the GPU waits for it
        import sys
        sys.setswitchinterval(float(os.environ["IR_SWITCH_INTERVAL"]))
    pools = HostPools(default_workers(local_world) if args.workers < 0 else args.workers)
    note = host_keeps_up(pools.workers, int(512 * 512 * max(args.sr_scale, 1.0) ** 2), args.png_compress_level)   # priced on a 512 x 512 LQ file at this --sr_scale
    if note:
        print(f"[rank {rank}] {note}")
    if not os.path.isdir(args.input):
        raise SystemExit(f"--input {args.input} is not a directory")
    # os.walk order, like the reference (no sorting) for one process. With several ranks the list is what decides which rank owns
    # which file (and, under --shard_tiles, which image a collective belongs to), and os.walk promises no order across processes:
    # every rank sorts its listing and checks it against rank 0's.
    files = list_image_files(args.input, follow_links=True)
    if world > 1:
        files = parallel.agree_on_list(sorted(files))
    common = dict(color_fix_type=args.color_fix_type, disable_preprocess_model=args.disable_preprocess_model, tile_size=args.tile_size,
                  tile_stride=args.tile_stride, preprocess_model=m.preprocess_model, vae=m.vae, y=m.y, y_mask=m.y_mask,
                  noise_scheduler=m.noise_scheduler)
    if args.shard_tiles and args.tiled and world > 1:
        # one large image at a time, its tiles spread over the GPUs; rank 0 re-assembles and writes
        engine = HipTileEngine(m.model, m.vae, m.preprocess_model, m.y, m.y_mask, args.color_fix_type, args.disable_preprocess_model,
                               args.tile_size, args.tile_stride, m.noise_scheduler)
        for job in pools.read_ahead(lambda pi: read_job(pi[0], pi[1], args), [(p, i) for p in files for i in range(args.repeat_times)]):
            preds, stage1 = parallel.sharded_tiled_process(engine, [job.net_in], rank, world)
            if rank == 0:
                pools.write_behind(write_job, job, preds[0], stage1[0], args)
        pools.drain()
        return
    mine = parallel.shard(files, rank, world)       # images are independent: no collective on the data path
    t0 = time.perf_counter()
    jobs = pools.read_ahead(lambda pi: read_job(pi[0], pi[1], args), [(p, i) for p in mine for i in range(args.repeat_times)])
    todo: List[List[Job]] = []

    def feed():
        for group in batches_of(jobs, max(args.batch_size, 1)):
            todo.append(group)
            yield [j.net_in for j in group]

    first = None    # (time, files) when the first result left the GPU: what follows is the steady state (no library / workspace warm-up in it)
    for preds, stage1 in process_stream(m.model, feed(), tiled=args.tiled, return_stage1=args.show_lq and not args.disable_preprocess_model,
                                        fp8=args.fp8 != "off", **common):
        group = todo.pop(0)
        last_result = time.perf_counter()
        if first is None:
            first = (last_result, len(group))
        for k, job in enumerate(group):
            pools.write_behind(write_job, job, preds[k], stage1[k] if stage1 else None, args)
    pools.drain()
    t1 = time.perf_counter()
    if pools.written:
        # first read submitted -> last PNG closed, model loading excluded (bench.py --cli_files parses this line)
        rest, dt_rest = pools.written - first[1], t1 - first[0]
        print(f"[rank {rank}] wrote {pools.written} files in {t1 - t0:.3f} s = {pools.written / (t1 - t0):.3f} files/s ({pools.workers} host threads); "
              f"after the first result: {rest} files in {dt_rest:.3f} s = {rest / dt_rest:.3f} files/s"
              + (f"; results left the GPU at {rest / (last_result - first[0]):.3f} /s (the rest is the encoders' drain of the last files)" if rest and last_result > first[0] else ""))


if __name__ == "__main__":
    main()
