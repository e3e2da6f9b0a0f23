"""Build the HIP shared library (gfx950 only) in-tree with hipcc.

`python -m instarevive_amd.build` or `instarevive_amd.build.build()` compiles every file under csrc/ into
csrc/libinstarevive_hip.so. hipcc cross-compiles without a GPU, so this also runs on CPU-only boxes.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libinstarevive_hip.so")
SOURCES = ["igemm.hip", "conv_s1.hip", "conv_s1_fp8.hip", "norm.hip", "attention.hip", "attn_d512.hip", "attn_fp8.hip", "attn_d512_fp8.hip", "swin_fused.hip", "elementwise.hip", "vae_io.hip", "t5.hip", "unet.hip", "api.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=fast"] + os.environ.get("IR_EXTRA_HIPCC_FLAGS", "").split()


def source_hash() -> str:
    """16 hex digits over every kernel source / header of the library (csrc/*.hip|cpp|h + include/instarevive_hip.h), in name order: what a
    measurement file (profiles/rNN_pmc_kernels.json) records so that bench.py can refuse it once a kernel has changed."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")))
    for f in files + [os.path.join(os.path.dirname(HERE), "include", "instarevive_hip.h")]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "instarevive_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers)
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or os.path.getmtime(obj) < hdr_time:
            cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", src, "-o", obj]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
