"""Read-only sampler of what the GPU reports about itself while another process streams: the DPM levels of the memory /
fabric / shader / SoC clocks (sysfs pp_dpm_*), power and temperatures (hwmon), every --period seconds, as JSON lines with
wall-clock stamps that the probe scripts also print.  Changes nothing.
    python profiles/gpu_state_sampler.py --seconds 120 > gpurun_out/r4ae_state.jsonl"""
import argparse
import glob
import json
import os
import time


def active_level(text):
    # "0: 900Mhz\n1: 1300Mhz *\n" -> 1300
    for ln in text.splitlines():
        if ln.rstrip().endswith("*"):
            try:
                return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
            except (IndexError, ValueError):
                return ln.strip()
    return text.strip()[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--period", type=float, default=0.1)
    args = ap.parse_args()
    files = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(card + "/pp_dpm_mclk") and not glob.glob(card + "/hwmon/hwmon*/power1_*"):
            continue
        tag = os.path.basename(os.path.dirname(card))
        for n in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_sclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent",
                  "power_dpm_force_performance_level", "current_link_speed"):
            if os.path.exists(card + "/" + n):
                files[tag + ":" + n] = card + "/" + n
        for hw in glob.glob(card + "/hwmon/hwmon*"):
            for f in glob.glob(hw + "/power1_*") + glob.glob(hw + "/temp*_input") + glob.glob(hw + "/freq*_input"):
                if f.endswith(("_average", "_input", "_cap")):
                    files[tag + ":" + os.path.basename(f)] = f
    print(json.dumps({"files": sorted(files)}), flush=True)
    t_end = time.time() + args.seconds
    last = None
    while time.time() < t_end:
        rec = {}
        for k, f in files.items():
            try:
                with open(f) as fh:
                    txt = fh.read()
                rec[k] = active_level(txt) if "pp_dpm" in k else txt.strip()
            except OSError as e:
                rec[k] = "err %d" % e.errno
        if rec != last:                        # only changes are written
            print(json.dumps({"t": round(time.time(), 3), **rec}), flush=True)
            last = rec
        time.sleep(args.period)


if __name__ == "__main__":
    main()
