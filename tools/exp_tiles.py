#!/usr/bin/env python3
"""Experiment: what each GPU of an N-GPU run has to do, measured on ONE GPU -- the march kernel of one rank's share of
the frame (tile t owned by rank t mod N), without the gather, for several kernel flavours:
  * kernel ms of the slowest rank, one launch at a time (HIP events), and
  * ms per frame with two launches in flight on two streams, as bench.py drives them (wall clock, rank 0's share).

    python tools/exp_tiles.py [--workload C3] [--tf default] [--flavours 0,6]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--tf", default="default")
    ap.add_argument("--flavours", default="0,6")
    ap.add_argument("--worlds", default="1,2,4,8")
    a = ap.parse_args()
    import torch  # first: torch's HIP runtime has to initialise before libvr_hip.so's
    torch.cuda.init()
    from volumerendering_amd import capi, host, workloads as wl
    n, W, H, vname = wl.WORKLOADS[a.workload]
    app = host.Application(W, H, 0)
    variant, vols = wl.build_scene(app, a.workload, a.tf)
    ctx = app.context()
    streams = [ctx.stream(i) for i in range(4)]  # the context's own streams, probed to run side by side
    for fl in [int(x) for x in a.flavours.split(",")]:
        ctx.set_kernel_flavour(fl)
        for world in [int(x) for x in a.worlds.split(",")]:
            worst, per_rank = 0.0, []
            for rank in range(world):
                for _ in range(3):
                    ctx.render_tiles(variant, rank, world)
                ctx.reset_kernel_times()
                for _ in range(15):
                    ctx.render_tiles(variant, rank, world)
                t = float(np.median(ctx.kernel_times()))
                worst = max(worst, t)
                per_rank.append(t)
            ran = ctx.last_kernel_flavour()
            nfl = ctx.tile_count(0, world) * capi.TILE * capi.TILE * 4
            bufs = [torch.zeros(nfl, dtype=torch.float32, device="cuda") for _ in range(4)]

            def burst(k, depth):
                for i in range(k):
                    ctx.render_tiles_async(variant, 0, world, bufs[i % depth].data_ptr(), streams[i % depth])
                torch.cuda.synchronize()

            res = []
            for depth in (2, 3, 4):
                ctx.hint_frames_in_flight(depth)
                burst(8, depth)
                t0 = time.perf_counter()
                burst(120, depth)
                res.append((time.perf_counter() - t0) / 120 * 1e3)
            ctx.hint_frames_in_flight(1)
            print(f"flavour {fl} (ran {ran}) world {world}: slowest rank's kernel {worst:.4f} ms; rank 0 with 2 / 3 / 4 in flight "
                  + " / ".join(f"{t:.4f}" for t in res) + " ms/frame   (ranks: " + " ".join(f"{t:.3f}" for t in per_rank) + ")", flush=True)


if __name__ == "__main__":
    main()
