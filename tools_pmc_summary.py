#!/usr/bin/env python3
"""Summarises a tools_prof.sh output directory into the two files committed under profiles/:

    python tools_pmc_summary.py gpurun_out/prof_<tag> profiles/r01_<tag> "<description>" <workload> <tf> [--latest]

  <out>_kernel_stats.csv  the rocprofv3 --kernel-trace --stats table (per-kernel calls / total / average ns)
  <out>_pmc.json          mean per march-kernel launch of every PMC counter collected (one counter group per run)
--latest also rewrites profiles/pmc_traffic_latest.json, which bench.py reads for roofline.traffic
(HBM bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB: the gfx950 correction of MI355X_MICROARCH.md).
"""
import csv
import glob
import json
import os
import shutil
import sys


def main():
    src, out, desc = sys.argv[1], sys.argv[2], sys.argv[3]
    workload, tf = sys.argv[4], sys.argv[5]
    latest = "--latest" in sys.argv[6:]
    stats = glob.glob(os.path.join(src, "trace", "**", "*_kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copyfile(stats[0], out + "_kernel_stats.csv")
    counters = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True)):
        per = {}
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "march" not in row["Kernel_Name"]:
                    continue
                per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, d in per.items():
            vals = list(d.values())
            counters[name] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals)}
    doc = {"kernel": desc, "workload": workload, "tf": tf, "counters": counters}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        fetch_kib = counters["FETCH_SIZE"]["mean_per_launch"]
        write_kib = counters["WRITE_SIZE"]["mean_per_launch"]
        doc["hbm_bytes_per_launch"] = (2.0 * fetch_kib + write_kib) * 1024.0
        doc["hbm_bytes_note"] = "(2 x FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE correction for 16-B-per-lane reads"
    with open(out + "_pmc.json", "w") as fh:
        json.dump(doc, fh, indent=1)
    if latest and "hbm_bytes_per_launch" in doc:
        lp = os.path.join(os.path.dirname(out), "pmc_traffic_latest.json")
        with open(lp, "w") as fh:
            json.dump({"workload": workload, "tf": tf, "n_gpus": 1, "fetch_size_bytes_raw": fetch_kib * 1024.0,
                       "write_size_bytes": write_kib * 1024.0, "hbm_bytes_per_launch": doc["hbm_bytes_per_launch"],
                       "note": "FETCH_SIZE x2 (gfx950 reports half the bytes of 16-B-per-lane reads, MI355X_MICROARCH.md "
                               "HBM section; uncalibrated for gathers, so this is an upper estimate) + WRITE_SIZE; "
                               "Infinity-Cache hits are included in FETCH_SIZE. Source: profiles/"
                               + os.path.basename(out) + "_pmc.json"}, fh, indent=1)
    print(json.dumps({k: v["mean_per_launch"] for k, v in counters.items()}, indent=1))


if __name__ == "__main__":
    main()
