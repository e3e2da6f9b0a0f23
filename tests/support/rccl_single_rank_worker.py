"""Worker of test_cli_gpu.py::test_rccl_branch_single_rank: a ONE-process "nccl" (= RCCL) group on this box's one GPU with
IR_FORCE_COLLECTIVES=1, so that every exchange step of the multi-GPU paths runs through RCCL on DEVICE tensors - the branch gloo
rehearsals never take (parallel._comm_device): GatherPlan.gather (cfg-4's step), _exchange_tiles (all_gather + gather) and
sharded_encode (all_reduce of the overflow flag + all_gather of the attention rows) inside sharded_tiled_process. Each result must equal
the no-collective path's. The process group is initialised BEFORE anything else touches the GPU (as bench.self_launch's ranks do).
argv: out.json"""
import json
import os
import sys

os.environ["IR_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from instarevive_amd import parallel
    rank, world, local = parallel.init_distributed("nccl")
    assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    import bench
    from instarevive_amd.pipeline import HipTileEngine, process
    dev = torch.device("cuda", local)
    out = {"backend": dist.get_backend(), "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "world": world}
    # 1. cfg-4's step: the uint8 results of a rank's batch gathered on rank 0 - device tensors through RCCL
    g = torch.Generator().manual_seed(3)
    batch = torch.randint(0, 256, (3, 256, 320, 3), dtype=torch.uint8, generator=g).to(dev)
    plan = parallel.GatherPlan(batch)
    assert plan.on and plan.dev == dev, "the forced plan must gather device tensors"
    got = plan.gather(batch)
    torch.cuda.synchronize()
    out["gather_equal"] = bool(torch.equal(got, batch)) and got.device == dev
    t = parallel.max_over_ranks(1.25, dev)
    out["max_over_ranks"] = t
    # 2. tile sharding of one frame: forced collectives against the plain single-process tiled path
    swin, vae, dit, _sched, _sds = bench.build_models(dev, lambda m: None)
    y, mask = bench.synthetic_prompt()
    yc, mc = y.to(dev), mask.to(dev)
    img = bench.synthetic_lq(1, 1024, 1536, 51)[0].numpy()
    eng = HipTileEngine(dit, vae, swin, yc, mc, "wavelet", False, 512, 448)
    assert eng.can_shard_encode([img])
    preds, stage1 = parallel.sharded_tiled_process(eng, [img], rank, world)
    want, want1 = process(dit, [img], 1, "wavelet", False, True, 512, 448, preprocess_model=swin, vae=vae, y=yc, y_mask=mc)
    out["tiles"] = eng.count(1024, 1536)
    out["sharded_equal"] = bool(np.array_equal(preds[0], want[0]) and np.array_equal(stage1[0], want1[0]))
    out["image_std"] = float(want[0].std())
    out["encode_overflow"] = eng.encode_overflow()
    dist.barrier()
    dist.destroy_process_group()
    with open(sys.argv[1], "w") as f:
        json.dump(out, f)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
