#!/usr/bin/env python3
"""files/s of the drop-in command line (inference.py --sr_scale 4 as a child process over K synthetic 512 x 512 PNGs, full-size seeded weights written in the
reference's file formats) under different host settings: --workers N and the GIL switch interval (IR_SWITCH_INTERVAL).

    python tools/cli_rate_ab.py [--files 64] [--configs "w=-1" "w=-1,si=0.0005" "w=10" ...]"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--configs", nargs="+", default=["w=-1", "w=-1,si=0.0005", "w=-1,si=0.0001", "w=10", "w=12,si=0.0005", "w=-1"])
    a = ap.parse_args()
    from tools import cli_artifacts as A
    d = tempfile.mkdtemp(prefix="ir_cli_ab_")
    try:
        flags = A.write_full_artifacts(d)
        A.write_lq_pngs(os.path.join(d, "in"), a.files)
        for cfg in a.configs:
            kv = dict(x.split("=") for x in cfg.split(","))
            env = dict(os.environ)
            if "si" in kv:
                env["IR_SWITCH_INTERVAL"] = kv["si"]
            out = os.path.join(d, "out")
            shutil.rmtree(out, ignore_errors=True)
            cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "--input", os.path.join(d, "in"), "--output", out, "--sr_scale", "4", "--workers", kv.get("w", "-1")] + flags
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
            rate = A.parse_cli_rate(r.stdout)
            if r.returncode or not rate:
                print(cfg, "FAILED", r.stderr[-500:])
                continue
            c = rate[0]
            print(f"{cfg:22s} {c['files_per_s']:6.2f} files/s overall, {c['steady_files_per_s']:6.2f} after the first result, results left the GPU at {c.get('result_rate', float('nan')):.2f} /s ({c['workers']} threads; child wall {time.time() - t0:.1f} s)", flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
