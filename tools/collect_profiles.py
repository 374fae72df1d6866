#!/usr/bin/env python3
"""Copies the summaries of a tools/profiles_rNN.sh run (gpurun_out/prof_rNN/) into profiles/rNN_*:
    python tools/collect_profiles.py 03"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "02"
SRC = os.path.join(ROOT, "gpurun_out", f"prof_r{RND}")
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
        if name.startswith(("trace", "pmc_probe")):
            continue
        try:
            d = load(f)
        except Exception as e:  # noqa: BLE001
            print("skipped", name, e)
            continue
        json.dump(d, open(os.path.join(DST, f"r{RND}_{name}.json"), "w"), indent=1)
        r = d["roofline"]
        pl = d.get("pipelined_one_frame_per_launch")
        cfg = d["config"]
        comp = cfg.get("composited_samples_frame0", cfg.get("composited_samples_per_frame", 0))
        fet = cfg.get("fetched_samples_frame0", cfg.get("fetched_samples_per_frame", 0))
        ov = d["overlapped"]
        print(f"{name:22s} serial {d['serial']['kernel_ms_median']:.4f} / {d['serial']['ms_per_step']:.4f} ms ({d['serial']['value']:.0f})  "
              + (f"pipelined 2x1 {pl['ms_per_step']:.4f} ms ({pl['value']:.0f})  " if pl else "")
              + f"batched {ov.get('launches_in_flight', 2)}x{ov.get('frames_per_launch', 1)} "
              f"{ov['ms_per_step']:.4f} ms ({ov['value']:.0f})  traffic {(r.get('traffic') or 0) / 1e9:.2f} GB  "
              f"frame 0: composited {comp / 1e6:.1f} M fetched {fet / 1e6:.1f} M flavour {cfg.get('kernel_flavour_resolved')}")
    if RND == "02":
        pairs = (("trace", "r02_c3_overlapped_kernel_stats.csv"), ("trace_serial", "r02_c3_serial_kernel_stats.csv"))
    else:
        pairs = (("trace", f"r{RND}_c3_bench_kernel_stats.csv"), ("trace_serial", f"r{RND}_c3_serial_kernel_stats.csv"))
    for t, dst in pairs:
        f = newest(os.path.join(SRC, t, "*", "*kernel_stats.csv"))
        if f:
            shutil.copyfile(f, os.path.join(DST, dst))
    for js in glob.glob(os.path.join(SRC, "pmc_probe_*.json")):
        shutil.copyfile(js, os.path.join(DST, f"r{RND}_" + os.path.basename(js)))
    for txt in glob.glob(os.path.join(SRC, "*.txt")):
        if os.path.basename(txt).startswith(("block_trace", "valu_issue", "struct_buffer", "stream_gap", "launch_gap")):
            shutil.copyfile(txt, os.path.join(DST, f"r{RND}_" + os.path.basename(txt)))
    f = newest(os.path.join(SRC, "ubench_pmc", "*", "*counter_collection.csv"))
    if f:
        shutil.copyfile(f, os.path.join(DST, f"r{RND}_valu_issue_counters.csv"))
    if RND == "02":
        d = load(os.path.join(SRC, "c3_default.json"))
        json.dump({"workload": "C3", "tf": "default", "air": "exact0", "n_gpus": 1, "fetch_size_bytes_raw": d["pmc"]["FETCH_SIZE"] * 1024,
                   "write_size_bytes": d["pmc"]["WRITE_SIZE"] * 1024, "hbm_bytes_per_launch": d["roofline"]["traffic"],
                   "note": "round 2 only: bench.py used this file, labelled STALE, when its live counter passes could not run; since "
                           "round 3 no committed file stands in for a measurement. Source: profiles/r02_c3_default.json"},
                  open(os.path.join(DST, "pmc_traffic_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
