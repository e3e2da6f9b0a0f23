#!/usr/bin/env python3
"""Static check of the hand-pinned MFMA streams: hipcc's hazard recogniser does not look into `asm` statements, so a VALU instruction it places
directly in front of an asm MFMA that reads its result (SrcA / SrcB / SrcC in arch VGPRs) is a silent hazard (seen once: the zeroed C operand of
flash_attn_x72_kernel's score MFMAs). Compiles the given .hip files to ISA and reports, per kernel, every asm MFMA
whose VGPR operand was written by a VALU instruction fewer than MIN_WAIT wait states earlier (s_nop N counts N + 1), and every asm VALU
instruction that reads the result of a transcendental (v_exp / v_rcp / v_rsq / v_sqrt / v_log) issued fewer than MIN_TRANS wait states earlier
(the second hazard of the round: a full-rate consumer two slots behind v_exp_f32 read stale lanes).

    python tools/mfma_hazard_scan.py [file.hip ...]      # default: every kernel source with asm MFMAs; exit code 1 if anything is flagged"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "instarevive_amd", "csrc")
MIN_WAIT = 3
MIN_TRANS = 2
TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_log_", "v_sin_", "v_cos_")
DEFAULT = ["attn_d512.hip", "conv_s1.hip", "conv_s1_fp8.hip", "attn_fp8.hip", "attn_d512_fp8.hip", "attention.hip", "vae_io.hip", "swin_fused.hip", "igemm.hip"]


def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return [int(m.group(1))]
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return []


def aregs(text):
    """AGPR indices named in an instruction's operand text: a7, a[4:7]."""
    out = set()
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\ba(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def scan(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=fast", "-S", "--cuda-device-only",
               "-I" + os.path.join(ROOT, "include"), path, "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read().splitlines()
    flagged, kernel, last, lastt, clock, in_asm, n_mfma = [], None, {}, {}, 0, False, 0
    # third check: kernels keep state in LITERAL accumulation registers across separate asm statements (a[0:255] of the one-wave-per-
    # SIMD kernels, a[0:175] of flash_attn_x72_kernel); nothing but a clobber list tells hipcc so. Every AGPR an asm statement of a kernel names
    # is "owned" by the asm; a compiler-emitted instruction of the same kernel that touches an owned AGPR (an MFMA accumulator of its own, a spill)
    # is flagged.
    owned, foreign = {}, {}
    for line in text:
        t = line.strip()
        if re.match(r"^_Z\w+:", t) or re.match(r"^[A-Za-z_]\w*:\s*(;.*)?$", t) and not t.startswith(".L"):
            kernel, last, lastt, clock = t.split(":")[0], {}, {}, 0
            continue
        if kernel and t and not t.startswith((";", ".")) and not t.endswith(":"):
            ar = aregs(t.split(";")[0])
            if ar:
                if in_asm:
                    owned.setdefault(kernel, set()).update(ar)
                else:
                    foreign.setdefault(kernel, []).append((t.split(";")[0].strip(), ar))
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        ins = t.split(";")[0].strip()
        op = ins.split()[0]
        args = [a.strip() for a in ins[len(op):].split(",")] if len(ins) > len(op) else []
        if op == "s_nop":
            clock += int(args[0]) + 1
            continue
        clock += 1
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            n_mfma += 1
            if in_asm:
                for a in args[1:4]:
                    for r in regs(a):
                        if r in last and clock - last[r][0] - 1 < MIN_WAIT:
                            flagged.append((kernel, ins, last[r][1], clock - last[r][0] - 1))
            continue
        if op.startswith("v_") and in_asm and not op.startswith(TRANS) and not op.startswith("v_accvgpr"):
            for a in args[1:]:
                for r in regs(a):
                    if r in lastt and clock - lastt[r][0] - 1 < MIN_TRANS:
                        flagged.append((kernel, ins, lastt[r][1], clock - lastt[r][0] - 1))
        if op.startswith("v_") and not op.startswith("v_accvgpr_write") and not op.startswith("v_cmp") and args:
            for r in regs(args[0]):
                last[r] = (clock, ins)
                if op.startswith(TRANS):
                    lastt[r] = (clock, ins)
                else:
                    lastt.pop(r, None)
    # only sources whose asm names accumulation registers LITERALLY ("a[%c..." / a clobber list); a kernel whose asm takes "+a" operands lets the
    # compiler allocate them, and its own v_accvgpr_* around them are legitimate (attention.hip's 4-wave kernels)
    with open(path) as f:
        src = f.read()
    if "a[%c" not in src and "IR_AGPR" not in src:
        foreign = {}
    for k, items in foreign.items():
        for ins, ar in items:
            hit = ar & owned.get(k, set())
            if hit:
                flagged.append((k, ins, f"compiler-emitted instruction touches a{min(hit)}..a{max(hit)}, which the kernel's asm statements own", 0))
    return flagged, n_mfma


def scan_all(files=None, workers=4):
    """-> {file: (flagged, n_mfma)}; the files compile side by side"""
    from concurrent.futures import ThreadPoolExecutor
    files = files or [os.path.join(CSRC, f) for f in DEFAULT]
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return dict(zip(files, ex.map(scan, files)))


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in DEFAULT]
    bad = 0
    for f, (flagged, n) in scan_all(files).items():
        print(f"{os.path.basename(f)}: {n} MFMAs, {len(flagged)} operand(s) of asm MFMAs / asm VALU instructions read too soon after the VALU / transcendental instruction that wrote them")
        seen = set()
        for k, ins, w, d in flagged:
            if (k, ins, w) in seen:
                continue
            seen.add((k, ins, w))
            print(f"   {k[:50]}: `{w}` -> {d} wait state(s) -> `{ins}`")
        bad += len(flagged)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
