"""Worker of test_cli_gpu.py::test_tile_sharding_two_ranks_on_one_gpu: one rank of a gloo group (all ranks share the one GPU) runs
parallel.sharded_tiled_process on the same synthetic frame; rank 0 writes the re-assembled prediction to argv[1]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from instarevive_amd import parallel  # noqa: E402
from instarevive_amd.pipeline import HipTileEngine  # noqa: E402


def main():
    out_path, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world, _ = parallel.env_rank_world()
    torch.cuda.set_device(0)
    if world > 1:
        parallel.init_distributed("gloo")
    swin, vae, dit, _sched, _sds = bench.build_models(torch.device("cuda", 0), lambda m: None)
    y, mask = bench.synthetic_prompt()
    eng = HipTileEngine(dit, vae, swin, y.cuda(), mask.cuda(), "wavelet", False, 512, 448)
    img = bench.synthetic_lq(1, h, w, 51)[0].numpy()
    preds, stage1 = parallel.sharded_tiled_process(eng, [img])
    if rank == 0:
        np.save(out_path, np.stack([preds[0], stage1[0]]))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
