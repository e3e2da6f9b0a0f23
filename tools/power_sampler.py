#!/usr/bin/env python3
"""Clock / power trace of the GPU during a bench run ("the four largest kernels run at the power-limited clock" had no file
behind it). A CHILD process that never touches the HIP runtime: it polls the amdgpu hwmon / sysfs nodes of one card at a fixed period and
appends `t_unix sclk_mhz power_w [mclk_mhz temp_c]` lines to a file until its stdin closes (the parent exits or closes the pipe).

    python tools/power_sampler.py --out trace.txt [--card N] [--period 0.02]        (started by bench.py before its first GPU call)

Nodes read (whatever exists; a missing one is reported as nan): hwmon*/freq1_input (Hz, gfx clock), hwmon*/power1_average or power1_input
(microwatt, socket power), hwmon*/freq2_input (memory clock), hwmon*/temp1_input (millidegree). No rocm-smi subprocess per sample (50 ms each)."""
import argparse
import glob
import os
import sys
import threading
import time


def find_card(index):
    """The `index`-th amdgpu card with a hwmon directory (cards sorted by number): a GPU box shows its one visible device as such a card."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int("".join(ch for ch in p.split("/")[4] if ch.isdigit()) or 0)):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw:
            cards.append((dev, hw[0]))
    if not cards:
        return None, None
    return cards[min(index, len(cards) - 1)]


def read_num(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) * scale
    except (OSError, ValueError):
        return float("nan")


def summarise(path, t0, t1):
    """Mean / min / max of the samples with t0 <= t <= t1 -> dict (None when the file has no usable sample in the window)."""
    rows = []
    try:
        with open(path) as f:
            for line in f:
                p = line.split()
                if len(p) >= 3 and not line.startswith("#"):
                    rows.append([float(v) for v in p[:5]])
    except OSError:
        return None
    win = [r for r in rows if t0 <= r[0] <= t1]
    if not win:
        return None
    out = {"samples": len(win), "window_s": round(t1 - t0, 3)}
    for name, col in (("clock_mhz", 1), ("power_w", 2), ("mclk_mhz", 3), ("temp_c", 4)):
        vals = [r[col] for r in win if len(r) > col and r[col] == r[col]]
        if vals:
            out[name] = round(sum(vals) / len(vals), 1)
            out[name + "_min"], out[name + "_max"] = round(min(vals), 1), round(max(vals), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--card", type=int, default=0)
    ap.add_argument("--period", type=float, default=0.02)
    a = ap.parse_args()
    dev, hw = find_card(a.card)
    stop = threading.Event()
    threading.Thread(target=lambda: (sys.stdin.read(), stop.set()), daemon=True).start()   # parent gone / pipe closed -> stop
    with open(a.out, "w") as f:
        f.write(f"# device {dev} hwmon {hw}; columns: t_unix sclk_mhz power_w mclk_mhz temp_c\n")
        if hw is None:
            return
        power = next((p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)), os.path.join(hw, "power1_average"))
        while not stop.is_set():
            f.write(f"{time.time():.4f} {read_num(os.path.join(hw, 'freq1_input'), 1e-6):.0f} {read_num(power, 1e-6):.1f} "
                    f"{read_num(os.path.join(hw, 'freq2_input'), 1e-6):.0f} {read_num(os.path.join(hw, 'temp1_input'), 1e-3):.1f}\n")
            f.flush()
            stop.wait(a.period)


if __name__ == "__main__":
    main()
