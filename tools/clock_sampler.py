"""Side process that samples the GPU's clocks, power and temperature from sysfs (no HIP, no ROCm library: it never initialises
the GPU) while a measurement runs; tools/ladder.sh starts it first and stops it last.
  python tools/clock_sampler.py OUT.csv [PERIOD_S]      (stops on SIGTERM / SIGINT, or when OUT.csv.stop appears)
Columns: t_s, then for every card found: sclk_MHz (hwmon freq1_input or the starred pp_dpm_sclk level), mclk_MHz, fclk_MHz,
power_W, temp_C. Missing files give empty fields: the pool's boxes do not all expose the same nodes.
"""
import glob
import os
import re
import signal
import sys
import time


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def starred_mhz(text):
    for ln in text.splitlines():
        if ln.rstrip().endswith("*"):
            m = re.search(r"(\d+)\s*Mhz", ln, re.I)
            if m:
                return m.group(1)
    return ""


def main():
    out = sys.argv[1]
    period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    cards = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
            continue
        hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
        cards.append((dev, hw[0] if hw else ""))
    stop = {"now": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.update(now=True))
    signal.signal(signal.SIGINT, lambda *_: stop.update(now=True))
    t0 = time.time()
    with open(out, "w") as f:
        # (which card is whose: the PCI address behind every card; tools/ladder_probe.py prints the address of the GPU it ran on)
        f.write("# " + " ".join(f"card{i}={os.path.basename(os.path.realpath(dev))}" for i, (dev, _) in enumerate(cards)) + "\n")
        f.write("t_s" + "".join(f",card{i}_sclk_MHz,card{i}_mclk_MHz,card{i}_fclk_MHz,card{i}_power_W,card{i}_temp_C" for i in range(len(cards))) + "\n")
        while not stop["now"] and not os.path.exists(out + ".stop"):
            row = [f"{time.time() - t0:.3f}"]
            for dev, hw in cards:
                s = read(os.path.join(hw, "freq1_input")) if hw else ""
                sclk = str(int(s) // 1000000) if s.isdigit() else starred_mhz(read(os.path.join(dev, "pp_dpm_sclk")))
                mclk = starred_mhz(read(os.path.join(dev, "pp_dpm_mclk")))
                fclk = starred_mhz(read(os.path.join(dev, "pp_dpm_fclk")))
                p = read(os.path.join(hw, "power1_average")) or read(os.path.join(hw, "power1_input")) if hw else ""
                power = f"{int(p) / 1e6:.1f}" if p.isdigit() else ""
                t = read(os.path.join(hw, "temp1_input")) if hw else ""
                temp = f"{int(t) / 1e3:.1f}" if t.isdigit() else ""
                row += [sclk, mclk, fclk, power, temp]
            f.write(",".join(row) + "\n")
            f.flush()
            time.sleep(period)


if __name__ == "__main__":
    main()
