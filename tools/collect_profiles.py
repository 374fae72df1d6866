#!/usr/bin/env python3
"""Copies the summaries of a tools/profiles_r02.sh run (gpurun_out/prof_r02/) into profiles/r02_*."""
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02")
DST = os.path.join(ROOT, "profiles")


def load(f):
    t = open(f).read()
    return json.loads(t[t.index('{"metric"'):])


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


def main():
    for f in sorted(glob.glob(os.path.join(SRC, "*.json"))):
        name = os.path.basename(f)[:-5]
        if name.startswith("trace"):
            continue
        try:
            d = load(f)
        except Exception as e:  # noqa: BLE001
            print("skipped", name, e)
            continue
        json.dump(d, open(os.path.join(DST, f"r02_{name}.json"), "w"), indent=1)
        r = d["roofline"]
        pl = d.get("pipelined_one_frame_per_launch")
        print(f"{name:18s} serial {d['serial']['kernel_ms_median']:.4f} / {d['serial']['ms_per_step']:.4f} ms ({d['serial']['value']:.0f})  "
              + (f"pipelined 2x1 {pl['ms_per_step']:.4f} ms ({pl['value']:.0f})  " if pl else "")
              + f"throughput {d['overlapped'].get('launches_in_flight', 2)}x{d['overlapped'].get('frames_per_launch', 1)} "
              f"{d['overlapped']['ms_per_step']:.4f} ms ({d['value']:.0f}, {d['fps']:.0f} fps)  traffic {(r.get('traffic') or 0) / 1e9:.2f} GB  "
              f"composited {d['config']['composited_samples_per_frame'] / 1e6:.1f} M fetched {d['config']['fetched_samples_per_frame'] / 1e6:.1f} M "
              f"flavour {d['config'].get('kernel_flavour_resolved')}")
    for t, dst in (("trace", "r02_c3_overlapped_kernel_stats.csv"), ("trace_serial", "r02_c3_serial_kernel_stats.csv")):
        f = newest(os.path.join(SRC, t, "*", "*kernel_stats.csv"))
        if f:
            shutil.copyfile(f, os.path.join(DST, dst))
    d = load(os.path.join(SRC, "c3_default.json"))
    json.dump({"workload": "C3", "tf": "default", "air": "exact0", "n_gpus": 1, "fetch_size_bytes_raw": d["pmc"]["FETCH_SIZE"] * 1024,
               "write_size_bytes": d["pmc"]["WRITE_SIZE"] * 1024, "hbm_bytes_per_launch": d["roofline"]["traffic"],
               "note": "fallback only: bench.py measures the traffic live (rocprofv3 --pmc passes on the same scene) and uses this file, "
                       "labelled STALE, when those passes cannot run. Source: profiles/r02_c3_default.json"},
              open(os.path.join(DST, "pmc_traffic_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
