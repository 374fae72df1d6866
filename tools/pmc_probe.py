#!/usr/bin/env python3
"""Any rocprofv3 counters for the march kernel of a bench.py scene: one `--pmc` pass per group (kernel trace only beside it),
mean per launch over the profiled one-at-a-time launches of `bench.py --pmc-child` (first launch dropped).

    python tools/pmc_probe.py --flavour 17 --groups "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM;SQ_INST_LEVEL_LDS SQ_INSTS_LDS"
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--tf", default="default")
    ap.add_argument("--air", default="exact0")
    ap.add_argument("--flavour", type=int, default=0)
    ap.add_argument("--vol-n", type=int, default=0)
    ap.add_argument("--layout", type=int, default=0)
    ap.add_argument("--arith", default="separate")
    ap.add_argument("--groups", required=True, help="counter groups, ';' between passes")
    a = ap.parse_args()
    passes = [g.split() for g in a.groups.split(";") if g.strip()]
    if not any("GRBM_GUI_ACTIVE" in p for p in passes):
        passes.append(["GRBM_GUI_ACTIVE"])
    out, note = bench.live_pmc(a, passes)
    out["_note"] = note
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
