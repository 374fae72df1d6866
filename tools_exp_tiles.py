#!/usr/bin/env python3
"""Experiment: march-kernel time of ONE rank's share of the frame (tile t owned by rank t mod N) on one GPU, for
several kernel flavours -- what each GPU of an N-GPU run has to do, without the gather.

    python tools_exp_tiles.py [--workload C3] [--tf default] [--flavours 0,6]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--tf", default="default")
    ap.add_argument("--flavours", default="0,6")
    a = ap.parse_args()
    import bench
    from volumerendering_amd import capi, host, synth
    n, W, H, vname = bench.WORKLOADS[a.workload]
    app = host.Application(W, H, 0)
    variant, vols = bench.build_scene(app, host, synth, capi, a.workload, a.tf)
    ctx = app.context()
    for fl in [int(x) for x in a.flavours.split(",")]:
        ctx.set_kernel_flavour(fl)
        for world in (1, 2, 4, 8):
            worst = 0.0
            per_rank = []
            for rank in range(world):
                for _ in range(3):
                    ctx.render_tiles(variant, rank, world)
                ctx.reset_kernel_times()
                for _ in range(15):
                    ctx.render_tiles(variant, rank, world)
                t = float(np.median(ctx.kernel_times()))
                worst = max(worst, t)
                per_rank.append(t)
            print(f"flavour {fl} world {world}: slowest rank's kernel {worst:.4f} ms   (ranks: "
                  + " ".join(f"{t:.3f}" for t in per_rank) + ")", flush=True)


if __name__ == "__main__":
    main()
