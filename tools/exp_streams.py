#!/usr/bin/env python3
"""Experiment: which pairs of streams really overlap two frames (HIP maps streams onto a few hardware queues)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from volumerendering_amd import capi, host, workloads as wl
n, W, H, vname = wl.WORKLOADS["C3"]
app = host.Application(W, H, 0)
variant, vols = wl.build_scene(app, "C3", "default", quiet=True)
ctx = app.context()
bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
def run(streams, tag):
    def burst(k):
        for i in range(k):
            ctx.render_async(variant, bufs[i % len(streams)].data_ptr(), streams[i % len(streams)])
        torch.cuda.synchronize()
    burst(10)
    t0 = time.perf_counter(); burst(100); dt = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{tag:60s} {dt:.4f} ms/frame", flush=True)
null = torch.cuda.current_stream().cuda_stream
ts = [torch.cuda.Stream() for _ in range(6)]
run([null], "null stream alone")
run([null, ts[0].cuda_stream], "null + torch stream 0 (what bench.py uses)")
run([ts[0].cuda_stream, ts[1].cuda_stream], "torch streams 0,1")
run([ts[1].cuda_stream, ts[2].cuda_stream], "torch streams 1,2")
run([ts[2].cuda_stream, ts[3].cuda_stream], "torch streams 2,3")
run([ts[0].cuda_stream, ts[2].cuda_stream], "torch streams 0,2")
run([ts[0].cuda_stream, ts[3].cuda_stream], "torch streams 0,3")
run([ts[0].cuda_stream, ts[1].cuda_stream, ts[2].cuda_stream], "torch streams 0,1,2")
hp = [torch.cuda.Stream(priority=-1) for _ in range(2)]
run([hp[0].cuda_stream, ts[0].cuda_stream], "high-priority + normal torch stream")
run([hp[0].cuda_stream, hp[1].cuda_stream], "two high-priority torch streams")
