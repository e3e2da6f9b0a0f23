"""Drop-in surface end to end: write checkpoints in the reference's own formats (SwinIR .ckpt wrapped in {"state_dict"} with
`module.` prefixes + yaml config, diffusers VAE folder with config.json + safetensors, diffusers transformer folder +
flat DiT .ckpt, scheduler_config.json, prompt-embedding .pth), run the command line on PNG files and compare the saved
images with the oracle run on the same files."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import dit as odit
from oracle import glue as oglue
from oracle import swinir as oswin
from oracle import vae as ovae
from tests.golden._det import det_input, det_state_dict
from tests.test_models_gpu import DIT_SMALL, SWIN_SMALL, VAE_SMALL, _psnr_u8

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_artifacts(d):
    from safetensors.torch import save_file
    sws = det_state_dict(oswin.state_dict_shapes(SWIN_SMALL), seed=101)
    svae = det_state_dict(ovae.state_dict_shapes(VAE_SMALL), seed=202)
    sdit = det_state_dict(odit.state_dict_shapes(DIT_SMALL), seed=404)
    os.makedirs(d / "weights", exist_ok=True)
    full = dict(sws)  # the released file also carries the derived buffers; mimic that for every other block
    full["layers.0.residual_group.blocks.0.attn.relative_position_index"] = torch.zeros(64, 64, dtype=torch.long)
    full["layers.0.residual_group.blocks.1.attn_mask"] = torch.zeros(64, 64, 64)
    torch.save({"state_dict": {"module." + k: v for k, v in full.items()}}, d / "weights" / "swinir.ckpt")
    (d / "swinir.yaml").write_text(
        "target: diffusion.model.swinir.SwinIR\nparams:\n  img_size: 64\n  patch_size: 1\n  in_chans: 3\n  embed_dim: 60\n"
        "  depths: [2, 2]\n  num_heads: [6, 6]\n  window_size: 8\n  mlp_ratio: 2\n  sf: 8\n  img_range: 1.0\n"
        "  upsampler: \"nearest+conv\"\n  resi_connection: \"1conv\"\n  unshuffle: True\n  unshuffle_scale: 8\n")
    os.makedirs(d / "vae", exist_ok=True)
    (d / "vae" / "config.json").write_text(json.dumps({"_class_name": "AutoencoderKL", "in_channels": 3, "out_channels": 3, "latent_channels": 4,
                                                       "block_out_channels": [32, 64, 128, 128], "layers_per_block": 2, "norm_num_groups": 32,
                                                       "scaling_factor": 0.18215, "act_fn": "silu", "sample_size": 256}))
    save_file({k: v.contiguous() for k, v in svae.items()}, str(d / "vae" / "diffusion_pytorch_model.safetensors"))
    os.makedirs(d / "pixart" / "transformer", exist_ok=True)
    os.makedirs(d / "pixart" / "scheduler", exist_ok=True)
    (d / "pixart" / "transformer" / "config.json").write_text(json.dumps({
        "_class_name": "Transformer2DModel", "num_attention_heads": 4, "attention_head_dim": 72, "in_channels": 4, "out_channels": 8, "num_layers": 2,
        "cross_attention_dim": 288, "attention_bias": True, "sample_size": 16, "patch_size": 2, "activation_fn": "gelu-approximate",
        "norm_type": "ada_norm_single", "norm_elementwise_affine": False, "norm_eps": 1e-6, "caption_channels": 64, "num_embeds_ada_norm": 1000}))
    (d / "pixart" / "scheduler" / "scheduler_config.json").write_text(json.dumps({"_class_name": "DDPMScheduler", "num_train_timesteps": 1000,
                                                                                 "beta_start": 0.0001, "beta_end": 0.02, "beta_schedule": "linear"}))
    torch.save(sdit, d / "weights" / "dit.ckpt")
    y = det_input(9, (1, 20, 64), -1, 1)
    mask = torch.zeros(1, 20)
    mask[:, :13] = 1
    torch.save({"caption_embeds": y, "emb_mask": mask}, d / "prompt.pth")
    return sws, svae, sdit, y, mask


def test_cli_matches_oracle(tmp_path):
    d = tmp_path
    sws, svae, sdit, y, mask = _write_artifacts(d)
    os.makedirs(d / "in" / "sub", exist_ok=True)
    imgs = {"a.png": (det_input(70, (64, 64, 3)) * 255).numpy().astype(np.uint8), "sub/b.png": (det_input(71, (40, 56, 3)) * 255).numpy().astype(np.uint8)}
    for k, v in imgs.items():
        Image.fromarray(v).save(d / "in" / k)
    cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--ckpt", str(d / "weights" / "dit.ckpt"), "--input", str(d / "in"), "--output",
           str(d / "out"), "--swinir_ckpt", str(d / "weights" / "swinir.ckpt"), "--swinir_config", str(d / "swinir.yaml"), "--vae", str(d / "vae"),
           "--dit_config", str(d / "pixart"), "--prompt_embeds", str(d / "prompt.pth")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "save to" in r.stdout
    for k, v in imgs.items():
        out = d / "out" / (os.path.splitext(k)[0] + "_0.png")
        assert out.exists(), (k, r.stdout)
        got = np.array(Image.open(out).convert("RGB"))
        # oracle on the reference's host pre/post-processing (inference.py:263-291, 326-346) around process()
        lq = Image.fromarray(v)
        rs = oglue.auto_resize(lq, 512)
        x = oglue.pad(np.array(rs), 64)
        ref, _ = oglue.process([x], lambda t: oswin.swinir_forward(sws, t, SWIN_SMALL), lambda t: ovae.vae_encode_mean(svae, t, VAE_SMALL),
                               lambda lat, tt, yy, mm: odit.dit_forward(sdit, lat, tt, yy, mm, DIT_SMALL), lambda z: ovae.vae_decode(svae, z, VAE_SMALL),
                               oglue.alphas_cumprod_diffusers(), y.reshape(1, 20, 64), mask.reshape(1, 1, 20))
        want = np.array(Image.fromarray(ref[0][:rs.height, :rs.width]).resize(lq.size, Image.LANCZOS))
        assert got.shape == want.shape == v.shape
        p = _psnr_u8([got], [want])
        print(f"cli {k}: PSNR vs oracle {p:.2f} dB")
        assert p >= 47.0  # measured 52.3 and 53.1 dB on these two files


def test_cli_threaded_run_writes_the_single_thread_pixels(tmp_path):
    """the reader / writer pools of inference.py (--workers) change WHEN a file is decoded or encoded, never what is
    computed: 14 files of five sizes (so batches_of() regroups while reads run ahead), --sr_scale 2 --show_lq, once with --workers 0 (the
    reference's one-thread order) and once with 4 threads; every PNG must hold the same pixels, under the same name, and the summary
    line bench.py --cli_files parses must be printed."""
    from tools import cli_artifacts as A
    d = tmp_path
    _write_artifacts(d)
    os.makedirs(d / "in" / "deep" / "er", exist_ok=True)
    sizes = [(64, 64), (40, 56), (64, 64), (64, 64), (33, 90), (64, 64), (40, 56), (72, 72), (64, 64), (64, 64), (50, 50), (64, 64), (64, 64), (40, 56)]
    for i, hw in enumerate(sizes):
        sub = ("", "deep/", "deep/er/")[i % 3]
        Image.fromarray((det_input(300 + i, hw + (3,)) * 255).numpy().astype(np.uint8)).save(d / "in" / f"{sub}im{i:02d}.png")
    outs = {}
    for workers in (0, 4):
        cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--ckpt", str(d / "weights" / "dit.ckpt"), "--input", str(d / "in"), "--output",
               str(d / f"out{workers}"), "--swinir_ckpt", str(d / "weights" / "swinir.ckpt"), "--swinir_config", str(d / "swinir.yaml"), "--vae", str(d / "vae"),
               "--dit_config", str(d / "pixart"), "--prompt_embeds", str(d / "prompt.pth"), "--sr_scale", "2", "--show_lq", "--batch_size", "3",
               "--workers", str(workers)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rate = A.parse_cli_rate(r.stdout)
        assert len(rate) == 1 and rate[0]["files"] == len(sizes) and rate[0]["workers"] == workers and rate[0]["files_per_s"] > 0, r.stdout[-500:]
        found = {}
        for root, _, names in os.walk(d / f"out{workers}"):
            for nm in names:
                found[os.path.relpath(os.path.join(root, nm), d / f"out{workers}")] = np.array(Image.open(os.path.join(root, nm)))
        outs[workers] = found
    assert sorted(outs[0]) == sorted(outs[4]) and len(outs[0]) == len(sizes)
    for k in outs[0]:
        assert outs[0][k].shape == outs[4][k].shape and np.array_equal(outs[0][k], outs[4][k]), k


def test_cli_fp8_flag(tmp_path):
    """inference.py --fp8 (BASELINE.json configs[4] from the command line; default off): the flag is accepted, the fp8 weight forms are packed and
    the run succeeds on the reduced test models - whose widths the fp8 kernels mostly do not take, so the result must stay within fp8's error of the
    plain run (identical where no part was eligible) - and an unknown operand set is refused by the parser."""
    d = tmp_path
    _write_artifacts(d)
    os.makedirs(d / "in", exist_ok=True)
    for i in range(2):
        Image.fromarray((det_input(400 + i, (64, 64, 3)) * 255).numpy().astype(np.uint8)).save(d / "in" / f"x{i}.png")
    base = [sys.executable, os.path.join(ROOT, "inference.py"), "--ckpt", str(d / "weights" / "dit.ckpt"), "--input", str(d / "in"), "--swinir_ckpt",
            str(d / "weights" / "swinir.ckpt"), "--swinir_config", str(d / "swinir.yaml"), "--vae", str(d / "vae"), "--dit_config", str(d / "pixart"),
            "--prompt_embeds", str(d / "prompt.pth")]
    outs = {}
    for name, extra in (("plain", []), ("fp8", ["--fp8", "all"])):
        r = subprocess.run(base + ["--output", str(d / name)] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[name] = [np.array(Image.open(d / name / f"x{i}_0.png")) for i in range(2)]
    p = _psnr_u8(outs["fp8"], outs["plain"])
    print(f"cli --fp8 all vs plain on the reduced models: {p:.2f} dB")
    assert p >= 38.0
    r = subprocess.run(base + ["--output", str(d / "bad"), "--fp8", "int4"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "invalid choice" in r.stderr


def test_eval_batch_harness_matches_oracle(tmp_path):
    """the batched evaluation harness (eval_batch.py, the role of test_dmd_general.py:112-192) on five images of
    different sizes in batches of 2 — centre crop (center_crop_arr), B > 1 through process_stream, result + condition folders, .jpg -> .png —
    against the oracle run image by image."""
    d = tmp_path
    sws, svae, sdit, y, mask = _write_artifacts(d)
    os.makedirs(d / "lq" / "sub", exist_ok=True)
    srcs = {"a.png": (70, 90), "b.jpg": (64, 64), "sub/c.png": (150, 130), "d.png": (64, 100), "e.png": (97, 71)}
    for i, (k, hw) in enumerate(srcs.items()):
        Image.fromarray((det_input(80 + i, hw + (3,)) * 255).numpy().astype(np.uint8)).save(d / "lq" / k, quality=95)
    cmd = [sys.executable, os.path.join(ROOT, "eval_batch.py"), "--ckpt", str(d / "weights" / "dit.ckpt"), "--input", str(d / "lq"), "--output",
           str(d / "res"), "--cond_output", str(d / "cond"), "--batch_size", "2", "--image_size", "64", "--swinir_ckpt", str(d / "weights" / "swinir.ckpt"),
           "--swinir_config", str(d / "swinir.yaml"), "--vae", str(d / "vae"), "--dit_config", str(d / "pixart"), "--prompt_embeds", str(d / "prompt.pth")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    from instarevive_amd.utils import center_crop_arr
    for k in srcs:
        name = os.path.splitext(k)[0] + ".png"
        got, cond = np.array(Image.open(d / "res" / name).convert("RGB")), np.array(Image.open(d / "cond" / name).convert("RGB"))
        x = center_crop_arr(Image.open(d / "lq" / k).convert("RGB"), 64)
        ref, ref1 = oglue.process([x], lambda t: oswin.swinir_forward(sws, t, SWIN_SMALL), lambda t: ovae.vae_encode_mean(svae, t, VAE_SMALL),
                                  lambda lat, tt, yy, mm: odit.dit_forward(sdit, lat, tt, yy, mm, DIT_SMALL), lambda z: ovae.vae_decode(svae, z, VAE_SMALL),
                                  oglue.alphas_cumprod_diffusers(), y.reshape(1, 20, 64), mask.reshape(1, 1, 20))
        assert got.shape == (64, 64, 3)
        p, p1 = _psnr_u8([got], ref), _psnr_u8([cond], ref1)
        print(f"eval_batch {k}: PSNR vs oracle {p:.2f} dB (condition image {p1:.2f} dB)")
        assert p >= 45.0 and p1 >= 50.0


def test_bench_self_launches_its_ranks():
    """BASELINE.json configs[3] control flow: `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must start its own two ranks as
    fresh child processes, shard one batch per rank, gather the uint8 results on rank 0 inside the step and print ONE JSON line.
    On this one-GPU box the ranks share the device and the collectives go through the host (IR_BENCH_BACKEND=gloo); the driver's
    multi-GPU runs use one GPU per rank over RCCL with the same code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IR_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "2", "--lq", "128", "--sr_scale", "2", "--steps", "2", "--warmup", "1",
           "--no_cpu_baseline", "--no_host_rate"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["gathered_images"] == 4 and line["verified"] is True
    assert line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert "roofline" in line and line["roofline"]["per_kernel"]


def test_bench_round5_legs_on_a_small_workload():
    """bench.py's round-5 additions through the real script on a small workload (128 x 128 LQ, sr_scale 2): the clock / power trace object (its fields
    are null on a box without readable hwmon nodes, never missing), `executed_frac` beside `frac` on every per-kernel row that has one, no `frac` above 1,
    the --logit_gain leg (weights re-uploaded with every self-attention logit x 4, the attention-fallback counter read through the C ABI, weights restored)
    and a `roofline.traffic_source` that is either a PMC file of THESE kernel sources or the reason why there is none."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--lq", "128", "--sr_scale", "2", "--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--no_host_rate",
           "--cli_files", "0", "--logit_gain", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "clock_mhz" in line and "power_w" in line and "power_trace" in line
    rows = line["roofline"]["per_kernel"]
    assert rows and all(("frac" not in v) or (v["frac"] <= 1.0 and "executed_frac" in v) for v in rows.values()), rows
    pk = line["peaky_attention"]
    assert len(pk) == 1 and pk[0]["logit_gain"] == 4 and pk[0]["fallback_launches_per_step"] >= 0 and pk[0]["ms_per_step"] > 0 and pk[0]["output_std"] > 1.0
    assert line["verified"] is True and line["value"] > 0
    assert line["roofline"]["traffic"] is None     # the PMC file describes the 2048 x 2048 workload only


def test_make_prompt_writes_the_reference_prompt_file(tmp_path):
    """SURVEY.md section 8(f) N3, the producer of --prompt_embeds: tools/make_prompt.py (tokenizer from a local folder -> HIP T5 encoder ->
    {'caption_embeds', 'emb_mask'} as test_scripts/test_controlnet.py:389-395 saves them) on a folder in the DeepFloyd layout: a
    SentencePiece model trained here on a few sentences, config.json and safetensors weights of a reduced T5 v1.1 encoder. The saved
    embeddings are checked against the oracle on the ids the same tokenizer yields, and the file is fed to the restoration CLI's loader."""
    import sentencepiece as spm
    from safetensors.torch import save_file
    from transformers import T5Tokenizer
    from oracle import t5 as ot5
    d = tmp_path / "t5"
    os.makedirs(d)
    corpus = d / "corpus.txt"
    corpus.write_text("\n".join(["a high quality photo of a human face", "highly detailed portrait, sharp focus, 8k", "restore the image cleanly",
                                 "a photo of a cat and a dog in the garden", "clean, realistic, natural skin texture"] * 4))
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(d / "spiece"), vocab_size=60, model_type="unigram", pad_id=0, eos_id=1, unk_id=2,
                                   bos_id=-1, hard_vocab_limit=False, minloglevel=2)
    os.remove(corpus)
    tok = T5Tokenizer.from_pretrained(str(d))   # spiece.model alone, as in the DeepFloyd folder (+ the 100 sentinel ids T5 adds)
    cfg = dict(d_model=64, d_kv=16, num_heads=4, d_ff=128, num_layers=2, vocab_size=len(tok))
    sd = det_state_dict(ot5.state_dict_shapes(cfg), seed=717)
    sd["shared.weight"] = sd["shared.weight"] * 8.0
    for k in sd:
        if k.endswith("SelfAttention.q.weight"):
            sd[k] = sd[k] * cfg["d_kv"] ** -0.5
    (d / "config.json").write_text(json.dumps(dict(cfg, feed_forward_proj="gated-gelu", layer_norm_epsilon=1e-6, relative_attention_num_buckets=32,
                                                   relative_attention_max_distance=128, model_type="t5")))
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d / "model.safetensors"))
    out = tmp_path / "prompt.pth"
    prompt = "A high quality PHOTO of a human face, <b>highly detailed</b> https://example.com/x"
    for clean in (False, True):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "make_prompt.py"), "--t5", str(d), "--prompt", prompt, "--max_length", "24", "--out", str(out)]
        r = subprocess.run(cmd + (["--clean"] if clean else []), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        f = torch.load(out, map_location="cpu")
        assert set(f) == {"caption_embeds", "emb_mask"} and tuple(f["caption_embeds"].shape) == (1, 24, 64) and tuple(f["emb_mask"].shape) == (1, 24)
        from instarevive_amd.captions import text_preprocessing
        want = tok(text_preprocessing(prompt) if clean else prompt, max_length=24, padding="max_length", truncation=True, return_tensors="pt")
        assert torch.equal(f["emb_mask"], want["attention_mask"]) and 2 < int(f["emb_mask"].sum()) <= 24
        ref = ot5.t5_encode(sd, want["input_ids"], want["attention_mask"], cfg)
        n = int(f["emb_mask"].sum())
        err = float((f["caption_embeds"][:, :n] - ref[:, :n]).norm() / ref[:, :n].norm())
        print(f"make_prompt clean={clean}: {n} tokens, rel-L2 vs oracle {err:.4f}")
        assert err <= 0.015
    # the CLI's loader reshapes it as the reference does (inference.py:256-259,273-277)
    y = f["caption_embeds"].reshape(1, -1, f["caption_embeds"].shape[-1])
    assert tuple(y.shape) == (1, 24, 64) and tuple(f["emb_mask"].reshape(1, 1, -1).shape) == (1, 1, 24)


def test_tile_sharding_two_ranks_on_one_gpu(tmp_path):
    """SURVEY.md section 8(e), tile-level sharding of ONE frame with real processes and collectives: two gloo ranks (sharing this box's one GPU)
    run parallel.sharded_tiled_process on a 1024 x 1536 frame - the encoder's mid-block attention split by query rows (one all_gather of the
    rows), the DiT and decoder tiles dealt round-robin (all_gather of the latent tiles, gather of the pixel tiles) - and rank 0's re-assembled
    image must equal the single-process result bit for bit."""
    worker = os.path.join(ROOT, "tests", "support", "shard_tiles_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, two = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r = subprocess.run([sys.executable, worker, one, "1024", "1536"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           worker, two, "1024", "1536"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert a.shape == b.shape == (2, 1024, 1536, 3) and a.std() > 1.0
    assert np.array_equal(a, b), f"max |diff| {np.abs(a.astype(int) - b.astype(int)).max()}"


def test_rccl_branch_single_rank(tmp_path):
    """The "nccl" (RCCL) branch of instarevive_amd/parallel.py, which a one-GPU box never takes with world_size 1 and which the gloo
    rehearsals replace by host copies: a fresh child process initialises a ONE-rank RCCL group before touching the GPU and runs, with
    IR_FORCE_COLLECTIVES=1, GatherPlan.gather (cfg-4's in-step gather), _exchange_tiles and sharded_encode on device tensors; every result
    must equal the no-collective path (tests/support/rccl_single_rank_worker.py)."""
    import json
    worker = os.path.join(ROOT, "tests", "support", "rccl_single_rank_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = str(tmp_path / "rccl.json")
    r = subprocess.run([sys.executable, worker, out], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    d = json.load(open(out))
    print(f"RCCL single-rank smoke: {d}")
    assert d["backend"] == "nccl" and d["world"] == 1 and d["gather_equal"] and d["sharded_equal"] and d["tiles"] == 12 and d["image_std"] > 1.0
    assert d["max_over_ranks"] == 1.25


def test_bench_multi_gpu_branch_through_rccl_on_one_rank():
    """bench.py's multi-GPU code path - init_process_group("nccl"), GatherPlan's RCCL gather of the uint8 results INSIDE the timed step, the
    MAX all-reduce of the step time, the MIN all-reduce of `verified`, the barriers - with ONE rank and IR_FORCE_COLLECTIVES=1: what the
    driver's --gpus 2 / 4 / 8 runs execute, on device tensors, on the box that exists."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "IR_BENCH_BACKEND")}
    env.update(IR_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--no_host_rate"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"bench.py through a one-rank RCCL group: {line['ms_per_step']} ms per step, verified {line.get('verified')}, gathered {line.get('gathered_images')}, "
          f"RCCL {line.get('rccl_version')}")
    assert line["n_gpus"] == 1 and line["verified"] is True and line["gathered_images"] == 1 and line["world_size"] == 1
    assert line["rccl_version"] not in (None, "gloo") and 100.0 < line["ms_per_step"] < 200.0
