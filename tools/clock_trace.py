"""Engine clock and board power of GPU 0 while a command runs: samples the amdgpu hwmon / sysfs files every 50 ms
(freq1_input = sclk in Hz, power1_average | power1_input in microwatts, temp) and prints their distribution over the samples taken
while the command was running (all, and the loaded part).  No privileges needed.   usage: python tools/clock_trace.py <command...>"""
import glob
import os
import subprocess
import sys
import threading
import time


def find():
    """hwmon / sysfs files of EVERY amdgpu card: a box holds several, the visible GPU is not always card0 -- the card whose
    board power peaks highest during the command is the one reported"""
    cards = []
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
        if not hw:
            continue
        out = {}
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
            f = os.path.join(hw[0], name)
            if os.path.exists(f):
                out[name] = f
        f = os.path.join(card, "gpu_busy_percent")
        if os.path.exists(f):
            out["gpu_busy_percent"] = f
        if out:
            out["card"] = card
            cards.append(out)
    return cards


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    cards = find()
    print("cards:", [c["card"] for c in cards], flush=True)
    per_card, stop = [[] for _ in cards], threading.Event()

    def loop():
        while not stop.is_set():
            for files, samples in zip(cards, per_card):
                row = {"t": time.perf_counter()}
                for k, f in files.items():
                    if k == "card":
                        continue
                    v = read(f)
                    if v is not None and v.lstrip("-").isdigit():
                        row[k] = int(v)
                samples.append(row)
            time.sleep(0.05)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    t0 = time.perf_counter()
    rc = subprocess.call(sys.argv[1:])
    t1 = time.perf_counter()
    stop.set()
    th.join()
    peak = [max((r.get("power1_average", r.get("power1_input", 0)) for r in rows), default=0) for rows in per_card]
    pick = peak.index(max(peak)) if peak else 0
    files, samples = (cards[pick], per_card[pick]) if cards else ({}, [])
    print(f"peak board power per card (W): {[round(p * 1e-6) for p in peak]} -> reporting {files.get('card')}")
    print(f"command rc={rc}, {t1 - t0:.1f} s, {len(samples)} samples")
    # the loaded part: samples whose board power is at least 60 % of the maximum seen (gpu_busy_percent is a slow average)
    pk = next((k for k in ("power1_average", "power1_input") if k in files), None)
    pmax = max((s.get(pk, 0) for s in samples), default=0) if pk else 0
    busy = [s for s in samples if pk and s.get(pk, 0) >= 0.6 * pmax]
    for label, rows in (("all samples", samples), ("board power >= 60 % of its maximum in the run", busy)):
        print(f"-- {label}: {len(rows)}")
        for k, scale, unit in (("freq1_input", 1e-6, "MHz sclk"), ("freq2_input", 1e-6, "MHz mclk"), ("power1_average", 1e-6, "W"),
                               ("power1_input", 1e-6, "W"), ("temp1_input", 1e-3, "C"), ("temp2_input", 1e-3, "C"), ("gpu_busy_percent", 1, "%")):
            v = [r[k] * scale for r in rows if k in r]
            if v:
                v.sort()
                print(f"   {k:16s} min {v[0]:9.1f}  p10 {v[len(v) // 10]:9.1f}  median {v[len(v) // 2]:9.1f}  mean {sum(v) / len(v):9.1f}  "
                      f"p90 {v[(9 * len(v)) // 10]:9.1f}  max {v[-1]:9.1f}  {unit}")


if __name__ == "__main__":
    main()
