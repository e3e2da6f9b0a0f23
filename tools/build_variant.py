#!/usr/bin/env python3
"""Build a diagnostic variant of the HIP library: one source recompiled with extra -D flags, everything else from the in-tree objects.

    python tools/build_variant.py attn_fp8.hip tools/libir_f8st.so -DIR_STAMPS_F8          # phase stamps (tools/dbg/attn8_stamps.py)
    python tools/build_variant.py attn_fp8.hip tools/libir_f8ko48.so -DIR_KO_F8=48         # knock-out: no score MFMAs
    INSTAREVIVE_HIP_LIB=$PWD/tools/libir_f8st.so python tools/dbg/attn8_stamps.py

Run `python -m instarevive_amd.build` first (the other objects must exist). Variant libraries are git-ignored."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instarevive_amd.build as B


def main():
    src, out, flags = sys.argv[1], os.path.abspath(sys.argv[2]), sys.argv[3:]
    objs = []
    for s in B.SOURCES:
        obj = os.path.join(B.CSRC, s.rsplit(".", 1)[0] + ".o")
        if s == src:
            obj = out + ".o"
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + flags + (["-x", "hip"] if s.endswith(".cpp") else []) +
                                  ["-c", os.path.join(B.CSRC, s), "-o", obj])
        objs.append(obj)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    os.remove(out + ".o")
    print(out)


if __name__ == "__main__":
    main()
