#!/usr/bin/env python3
"""Clock / power trace of the GPU during a bench run ("the four largest kernels run at the power-limited clock" had no file
behind it). A CHILD process that never touches the HIP runtime: it polls the amdgpu hwmon / sysfs nodes of EVERY card at a fixed period and
appends `t_unix card sclk_mhz power_w mclk_mhz temp_c` lines (one per card and sample) to a file until its stdin closes (the parent exits or closes the pipe).

    python tools/power_sampler.py --out trace.txt [--card N] [--period 0.02]        (started by bench.py before its first GPU call)

Nodes read (whatever exists; a missing one is reported as nan): hwmon*/freq1_input (Hz, gfx clock), hwmon*/power1_average or power1_input
(microwatt, socket power), hwmon*/freq2_input (memory clock), hwmon*/temp1_input (millidegree). No rocm-smi subprocess per sample (50 ms each)."""
import argparse
import glob
import os
import sys
import threading
import time


def find_cards():
    """Every amdgpu card with a hwmon directory, sorted by card number. A GPU box exposes ALL the host's cards in sysfs whatever device the
    job may open, so every card is sampled and summarise() picks the busy one."""
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int("".join(ch for ch in p.split("/")[4] if ch.isdigit()) or 0)):
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        if hw:
            cards.append((dev, hw[0]))
    return cards


def read_num(path, scale):
    try:
        with open(path) as f:
            return float(f.read().strip()) * scale
    except (OSError, ValueError):
        return float("nan")


def summarise(path, t0, t1, bdf=None):
    """Mean / min / max over the samples with t0 <= t <= t1 of the card that drew the most power in that window -> dict (None when the file
    has no usable sample in the window). Lines: t_unix card sclk_mhz power_w mclk_mhz temp_c."""
    rows, paths = {}, []
    try:
        with open(path) as f:
            for line in f:
                if line.startswith("# cards "):
                    paths = [t.strip(" '[],") for t in line[len("# cards "):].split(";")[0].split(",")]
                p = line.split()
                if len(p) >= 4 and not line.startswith("#"):
                    v = [float(x) for x in p[:6]]
                    if t0 <= v[0] <= t1:
                        rows.setdefault(int(v[1]), []).append(v)
    except (OSError, ValueError):
        return None
    if not rows:
        return None
    mean = lambda xs: sum(xs) / len(xs) if xs else float("nan")
    by_bdf = [i for i, pth in enumerate(paths) if bdf and pth.lower().rstrip("/").endswith(bdf.lower()) and i in rows]
    busy = by_bdf[0] if by_bdf else max(rows, key=lambda c: (mean([r[3] for r in rows[c] if r[3] == r[3]]) if any(r[3] == r[3] for r in rows[c]) else -1.0,
                                    mean([r[2] for r in rows[c] if r[2] == r[2]]) if any(r[2] == r[2] for r in rows[c]) else -1.0))
    win = rows[busy]
    out = {"card": busy, "cards_sampled": len(rows), "samples": len(win), "window_s": round(t1 - t0, 3),
           "selection": (f"PCI address {bdf} of the device this process runs on" if by_bdf else
                         "the card with the highest mean socket power in the window (sysfs shows every card of the host; no PCI address match)")}
    for name, col in (("clock_mhz", 2), ("power_w", 3), ("mclk_mhz", 4), ("temp_c", 5)):
        vals = [r[col] for r in win if len(r) > col and r[col] == r[col]]
        if vals:
            out[name] = round(mean(vals), 1)
            out[name + "_min"], out[name + "_max"] = round(min(vals), 1), round(max(vals), 1)
    others = [round(mean([r[3] for r in rows[c] if r[3] == r[3]]), 1) for c in sorted(rows) if c != busy and any(r[3] == r[3] for r in rows[c])]
    if others:
        out["other_cards_power_w"] = others
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--card", type=int, default=0, help="unused (kept for the bench's command line): every card is sampled")
    ap.add_argument("--period", type=float, default=0.02)
    a = ap.parse_args()
    cards = find_cards()
    stop = threading.Event()
    threading.Thread(target=lambda: (sys.stdin.read(), stop.set()), daemon=True).start()   # parent gone / pipe closed -> stop
    with open(a.out, "w") as f:
        f.write(f"# cards {[os.path.realpath(c[0]) for c in cards]}; columns: t_unix card sclk_mhz power_w mclk_mhz temp_c\n")
        if not cards:
            return
        nodes = []
        for dev, hw in cards:
            power = next((p for p in (os.path.join(hw, "power1_average"), os.path.join(hw, "power1_input")) if os.path.exists(p)), os.path.join(hw, "power1_average"))
            nodes.append((os.path.join(hw, "freq1_input"), power, os.path.join(hw, "freq2_input"), os.path.join(hw, "temp1_input")))
        while not stop.is_set():
            t = time.time()
            for i, (fq, pw, mq, tp) in enumerate(nodes):
                f.write(f"{t:.4f} {i} {read_num(fq, 1e-6):.0f} {read_num(pw, 1e-6):.1f} {read_num(mq, 1e-6):.0f} {read_num(tp, 1e-3):.1f}\n")
            f.flush()
            stop.wait(a.period)


if __name__ == "__main__":
    main()
