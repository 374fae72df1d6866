#!/usr/bin/env python3
"""What one frame at a time costs BETWEEN the march kernels.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/launch_gap.py --run [--flavour 17] [--frames 200]
    python3 tools/launch_gap.py --read OUT

--run renders the C3 turntable one frame at a time on one stream (bench.py's `serial` leg: the host enqueues ahead) and prints
the wall time per frame next to the kernels' own spans (first packet start .. last packet end, from the launches' records);
--read takes the kernel trace of that run and prints, for the march kernel, the duration the tracer saw and the idle time of
the stream between the end of one launch and the start of the next -- the part of `ms_per_step` that no kernel work explains --
and which other kernels ran in between.
"""
import argparse
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import numpy as np
    import torch
    from volumerendering_amd import host, workloads as wl

    n, W, H, vname = wl.WORKLOADS[args.workload]
    app = host.Application(W, H, 0)
    variant, vols = wl.build_scene(app, args.workload, "default", "exact0", quiet=True)
    ctx = app.context()
    if args.flavour:
        ctx.set_kernel_flavour(args.flavour)
    cam = app.camera()
    us = []
    for g in range(args.frames + args.warm):  # bench.py's turntable: Camera::Rotate(2 px, 0) per frame
        app.OnUpdate()
        us.append(app.uniforms())
        cam.Rotate(2.0, 0.0)
    s = ctx.stream(0)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctx.hint_frames_in_flight(1)
    for g in range(args.warm):
        ctx.set_uniforms(us[g])
        ctx.render_async(variant, frame.data_ptr(), s)
    torch.cuda.synchronize()
    ctx.reset_kernel_times()
    t0 = time.perf_counter()
    for g in range(args.warm, args.warm + args.frames):
        ctx.set_uniforms(us[g])
        ctx.render_async(variant, frame.data_ptr(), s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    kt = np.asarray(ctx.kernel_times(min(args.frames, 256)))
    tr = ctx.block_trace().astype(np.int64)  # the LAST launch of the stream-ordered run (the sort of the launch before ran beside its start)
    t0_, t1_ = (tr[:, 3] - tr[:, 3].min()) / 100.0, (tr[:, 4] - tr[:, 3].min()) / 100.0
    st = np.sort(t0_)
    n_slots = int((t0_ < 5.0).sum())
    print(f"last launch: span {t1_.max():.1f} us; packets started in the first 5 us: {n_slots}; the 3072 earliest starts: "
          f"#3000 {st[min(2999, len(st) - 1)]:.1f}  #3048 {st[min(3047, len(st) - 1)]:.1f}  #3060 {st[min(3059, len(st) - 1)]:.1f}  #3072 {st[min(3071, len(st) - 1)]:.1f} us")
    late = np.argsort(-t1_)[:8]
    print("   last to end (start / end / duration us):", "  ".join(f"{t0_[i]:.0f}/{t1_[i]:.0f}/{t1_[i] - t0_[i]:.0f}" for i in late))
    dur = t1_ - t0_
    busy = np.sort(dur[tr[:, 2] > 0])[::-1]
    print(f"   sampling packets {len(busy)}: duration p50 {np.median(busy):.0f}  #3072 longest {busy[min(3071, len(busy) - 1)]:.0f}  sum {busy.sum() / 1e3:.1f} ms; "
          f"sum of all packets {dur.sum() / 1e3:.1f} ms = {dur.sum() / 3072:.1f} us per wavefront slot")
    print(f"frames {args.frames}: wall {1e3 * (t2 - t0) / args.frames:.4f} ms per frame (host enqueue {1e6 * (t1 - t0) / args.frames:.1f} us per frame), "
          f"kernel span median {np.median(kt):.4f} mean {kt.mean():.4f} ms, flavour ran {ctx.last_kernel_flavour()}")


def read(path):
    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv below " + path)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    march = [i for i, r in enumerate(rows) if "march_" in r[2]]
    if len(march) < 10:
        sys.exit("fewer than 10 march launches in the trace")
    import numpy as np
    # the timed frames: the last block of launches whose neighbours are less than 2 ms apart
    dur, gap, between = [], [], {}
    for a, b in zip(march[:-1], march[1:]):
        g = rows[b][0] - rows[a][1]
        if g > 200_000:  # a synchronisation of the host in between: not the steady state
            continue
        dur.append(rows[a][1] - rows[a][0])
        gap.append(g)
        for k in range(a + 1, b):
            n = rows[k][2].split("(")[0][:60]
            d = between.setdefault(n, [0, 0, set()])
            d[0] += 1
            d[1] += rows[k][1] - rows[k][0]
            d[2].add(rows[k][3])
    dur, gap = np.asarray(dur) / 1e3, np.asarray(gap) / 1e3
    print(f"{len(dur)} back-to-back march launches ({rows[march[0]][2][:70]} ...)")
    print(f"  kernel duration (tracer): median {np.median(dur):.1f} us, mean {dur.mean():.1f}")
    print(f"  end -> next start: median {np.median(gap):.1f} us, mean {gap.mean():.1f}, p10 {np.percentile(gap, 10):.1f}, p90 {np.percentile(gap, 90):.1f}")
    print(f"  start -> next start: median {np.median(dur + gap):.1f} us")
    for n, (cnt, t, q) in sorted(between.items(), key=lambda x: -x[1][1]):
        print(f"  in between: {n}: {cnt} launches, {t / 1e3 / max(cnt, 1):.1f} us each, queues {sorted(q)}")
    print("  queues of the march launches:", sorted({rows[i][3] for i in march}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--read", default="")
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--flavour", type=int, default=17)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--warm", type=int, default=40)
    a = ap.parse_args()
    if a.run:
        run(a)
    elif a.read:
        read(a.read)
    else:
        ap.print_help()
