#!/usr/bin/env python3
"""Headline benchmark: Gsamples/s + fps of the volume ray-march compositing loop (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W           (N = 1 here; N > 1 under torch.distributed.run)

A "step" is one rendered frame of the hot path on synthetic input that is already resident in HBM.  At N = 1 the
workload is BASELINE.json configs[2] -- the one the metric is quoted on: ct-phantom-512 (512^3 RGBA32F voxels,
2 GiB), 1920x1080, BasicVolLightApp shader (TF lookup + interpolated central-difference gradient + Blinn-Phong
shade + opacity cut-off), default ramp TFs (R = 4096), 1/512 x 886 steps, camera distance 1.2 / yaw .6 / pitch .35.
At N > 1 the same frame is image-tile partitioned (64x64 tiles, tile t owned by rank t mod N), each rank renders
its tiles from its own replica of the volume and an RCCL gather over xGMI assembles the frame on rank 0
("strong" scaling: total work fixed).  value = composited samples of the whole frame / wall time per frame.

Rank 0 prints ONE JSON line; it also carries `roofline` (effective-gather bytes of the march kernel against the
8 TB/s HBM peak, kernel time from HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on the
host cores of this box on a bounded pixel sample of the same frame; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)

# per-composited-sample algorithmic bytes: the f32 footprint of one trilinear cell
BYTES_PER_SAMPLE = {"BASIC": 32, "LIGHT": 128, "VOLUME_MASK": 288, "THREE_FILES": 64, "MULTI_CTRT": 160, "TF_CALIB": 48,
                    "ILLUSTRATIVE": 160}

# which kernel a resolved flavour runs (include/vr.h, vr_set_kernel_flavour)
KERNEL_OF_FLAVOUR = {1: "march_kernel (no skipping)", 2: "march_wtb_light_kernel", 3: "march_wtb_light_kernel",
                     6: "march_kernel (one lane per ray)", 7: "march_dp_kernel (4 lanes per ray)",
                     8: "march_dp_kernel (2 lanes per ray)", 9: "march_kernel (one lane per ray, pipelined)",
                     10: "march_dp_kernel (4 lanes per ray, pipelined)", 11: "march_dp_kernel (2 lanes per ray, pipelined)"}

WORKLOADS = {
    # name: (volume N, W, H, variant)
    "C1": (64, 256, 256, "BASIC"),
    "C2": (256, 1024, 1024, "BASIC"),
    "C3": (512, 1920, 1080, "LIGHT"),
    "C4": (512, 1920, 1080, "VOLUME_MASK"),
    "C5": (1024, 3840, 2160, "LIGHT"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_scene(app, host, synth, capi, workload, tf_kind, vol_n=0):
    """Generates the synthetic inputs, runs the reference's data-prep order through the C++ host classes and
    starts the scene on `app` (uploads happen here, outside any timed region)."""
    n, W, H, vname = WORKLOADS[workload]
    full_n = n
    n = vol_n or n
    variant = capi.VARIANT_NAMES.index(vname)
    t0 = time.time()
    raw = synth.sphere_raw_fast(n) if workload == "C1" else synth.ct_phantom_raw_fast(n)
    ct = host.VolumeFile.from_raw(raw)
    del raw
    vols = [ct]
    if vname == "VOLUME_MASK":
        mask = host.VolumeFile.from_vec4(synth.mask_vec4_fast(n), 1)
        dose = host.VolumeFile.from_raw(synth.dose_raw())
        vols = [mask, dose, ct]
    app.OnStart(variant, vols)  # NormalizeData / PreComputeGradient in the scene's own order + uploads
    if tf_kind in ("thin", "zero"):  # control points (0,0),(R-1,0.002): no ray terminates
        for which in range(2 if vname == "VOLUME_MASK" else 1):
            otf = app.scene_opacity_tf(which)
            # "zero" (experiment): opacity identically 0 -> every sample is an identity blend: pure traversal cost
            otf.SetControlPoint(1, otf.GetTextureResolution() - 1, 0.002 if tf_kind == "thin" else 0.0)
    cam = app.camera()
    cam.SetOrbit(0.35, 0.6, 1.2)
    if vol_n:
        app.set_params(steps_count=int(math.sqrt(3) * full_n), step_size=1.0 / full_n)
    app.OnUpdate()
    log(f"[bench] scene {workload} ({vname}, {n}^3, {W}x{H}, tf={tf_kind}) ready in {time.time() - t0:.1f}s")
    return variant, vols


def cpu_baseline(app, capi, variant, vols, W, H, budget_s=15.0):
    """Times the CPU oracle (a port: plain-C restatement of the WGSL) on a bounded, regular sub-grid of the SAME
    frame, on all host cores.  Reported baseline only; never part of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import host_ref as hr
    import oracle_binding as ob

    # threads actually used: the CPUs this process may run on, capped at the GPU box's per-GPU CPU share
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    u = app.uniforms()
    uo = hr.Uniforms.from_buffer_copy(bytes(u))
    volumes = [v.data() for v in vols]
    tfs = []
    for which in range(2 if len(vols) == 3 else 1):
        tfs.append((app.scene_opacity_tf(which).table(), app.scene_color_tf(which).table()))
    # calibrate the stride on a coarse grid, then size the sample for ~budget_s of wall time
    stride = 32
    value, sample = None, ""
    for _ in range(3):
        ys, xs = np.meshgrid(np.arange(stride // 2, H, stride), np.arange(stride // 2, W, stride), indexing="ij")
        pxy = np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)
        t0 = time.perf_counter()
        _, n = ob.render_pixels(variant, uo, volumes, tfs, W, H, pxy, nthreads=cores)
        dt = time.perf_counter() - t0
        value = n / dt / 1e9 if dt > 0 else 0.0
        sample = (f"one pixel out of every {stride} x {stride} block of the {W}x{H} frame ({pxy.shape[0]} rays, {n} composited "
                  f"samples, {dt:.2f} s)")
        if dt >= budget_s / 4 or stride <= 2:
            break
        stride = max(2, int(stride / math.sqrt(min(16.0, (budget_s / 2) / max(dt, 1e-3)))))
    return {"value": round(value, 6), "unit": "Gsamples/s", "cores": cores, "kind": "port", "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # frames take ~0.5 ms: 200 of them still run in a blink, and the
    ap.add_argument("--warmup", type=int, default=20)   # fill / drain of the two-frame pipeline stops mattering
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--tf", default="default", choices=["default", "thin", "zero"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flavour", type=int, default=0)
    ap.add_argument("--vol-n", type=int, default=0, help="experiment: smaller volume, same frame and stepping")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="frames in flight on one GPU: 2 = alternate two streams and two frame buffers, so that the next "
                         "frame fills the machine while the longest rays of the previous one drain (N = 1 only)")
    ap.add_argument("--exp-mode", type=int, default=0, help="experiment: fragmentMode 1-4 (ray set-up only)")
    ap.add_argument("--exp-steps", type=int, default=-1, help="experiment: override stepsCount")
    args = ap.parse_args()

    import torch
    from volumerendering_amd import capi, host, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # VR_BENCH_DEVICE / VR_BENCH_BACKEND exist only to rehearse the N > 1 code path on a one-GPU box (all ranks on
    # device 0, gloo through host memory); the real multi-GPU run uses one GPU per rank and RCCL.
    device_index = int(os.environ.get("VR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("VR_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    # VR_BENCH_SELF_GATHER=1 (rehearsal): take the multi-rank code path -- tile render, RCCL gather, un-permute -- with
    # a world of one, so that the RCCL calls and their stream ordering can be exercised on a one-GPU box
    multi = world > 1 or bool(os.environ.get("VR_BENCH_SELF_GATHER"))
    dist = None
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    comm_dev = "cuda" if backend == "nccl" else "cpu"

    n, W, H, vname = WORKLOADS[args.workload]
    app = host.Application(W, H, device_index)
    variant, vols = build_scene(app, host, synth, capi, args.workload, args.tf, args.vol_n)
    ctx = app.context()
    if args.flavour:
        ctx.set_kernel_flavour(args.flavour)
    if args.exp_mode or args.exp_steps >= 0:  # experiments only: not the BASELINE workload any more
        app.set_params(fragment_mode=args.exp_mode, steps_count=args.exp_steps)
        app.OnUpdate()
    steps_count, step_size = app.stepping()
    stream = torch.cuda.current_stream().cuda_stream

    tpr_max = ctx.tile_count(0, world)
    tile_floats = capi.TILE * capi.TILE * 4
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    in_flight = args.in_flight
    # two streams and two sets of buffers, used alternately: frame k+1 starts while frame k's longest rays drain
    # (and, with N > 1, while frame k's tiles travel); --in-flight 1 keeps everything on one stream
    nbuf = in_flight if not multi else min(in_flight, 2)
    streams2 = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nbuf - 1)]
    for st in streams2[1:]:
        st.wait_stream(streams2[0])
    frames2 = [frame] + [torch.zeros_like(frame) for _ in range(nbuf - 1)]
    if nbuf == 1:
        streams2, frames2 = streams2 * 2, frames2 * 2
    if multi:
        # every frame's gather and un-permute completes inside the timed region
        my_tiles = [torch.zeros((tpr_max * tile_floats,), dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros((world, tpr_max * tile_floats), dtype=torch.float32, device="cuda") for _ in range(2)] \
            if rank == 0 else [None, None]
        gather_list = [[g[r] for r in range(world)] for g in gathered] if rank == 0 else [None, None]

    def finish(b, work):
        """Frame in buffer set b: wait for its gather (orders the current stream behind it, no host block), un-permute."""
        if work is not None:
            work.wait()
        if rank == 0:
            ctx.unpack_tiles_async(gathered[b].data_ptr(), world, frames2[b].data_ptr(), torch.cuda.current_stream().cuda_stream)

    def run_frames(n_frames):
        if not multi:
            for k in range(n_frames):
                ctx.render_async(variant, frames2[k % nbuf].data_ptr(), streams2[k % nbuf].cuda_stream)
            return
        pending = [None, None]  # per buffer set: (work handle,) of the frame that last used it
        for k in range(n_frames):
            b = k & 1
            with torch.cuda.stream(streams2[b]):
                if pending[b] is not None:  # frame k-2 used these buffers: its tiles must have left before they are reused
                    finish(b, pending[b][0])
                ctx.render_tiles_async(variant, rank, world, my_tiles[b].data_ptr(), streams2[b].cuda_stream)
                if backend == "nccl":
                    # RCCL over xGMI: every peer sends straight to the root (7 links in parallel, not a ring); the
                    # collective is ordered behind this stream's render and runs on RCCL's own stream
                    work = dist.gather(my_tiles[b], gather_list[b], dst=0, async_op=True)
                else:  # rehearsal only: through host memory, synchronous
                    host_list = [torch.empty(my_tiles[b].numel()) for _ in range(world)] if rank == 0 else None
                    dist.gather(my_tiles[b].cpu(), host_list, dst=0)
                    if rank == 0:
                        gathered[b].copy_(torch.stack(host_list))
                    work = None
                pending[b] = (work,)
        for b in ((n_frames & 1), ((n_frames + 1) & 1)):  # the older of the two outstanding frames first
            if pending[b] is not None:
                with torch.cuda.stream(streams2[b]):
                    finish(b, pending[b][0])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    run_frames(args.warmup)
    sync_all()
    ctx.reset_kernel_times()
    t0 = time.perf_counter()
    run_frames(args.steps)
    sync_all()
    dt = time.perf_counter() - t0

    # composited samples / covered pixels / samples whose voxels were fetched, for this rank's share of the frame
    my_samples, my_covered, my_fetched = ctx.counters()
    ktimes = ctx.kernel_times(min(args.steps, 256))
    kernel_ms = float(np.mean(ktimes)) if len(ktimes) else float("nan")
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        s = torch.tensor([my_samples, my_covered, my_fetched], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples, covered, total_fetched = int(s[0].item()), int(s[1].item()), int(s[2].item())
    else:
        total_samples, covered, total_fetched = my_samples, my_covered, my_fetched

    ms_per_step = dt / args.steps * 1e3
    value = total_samples / (dt / args.steps) / 1e9
    bs = BYTES_PER_SAMPLE[vname]
    # dominant kernel = march_kernel; algorithmic bytes per launch = the samples this rank actually FETCHED * B_s
    # (samples the exact empty-space test skips need no voxel bytes and are not counted here, although they are
    # composited samples of the metric) + the 16 B/pixel frame write of the pixels it owns (ray set-up is fused
    # into the kernel: no ray-end image is read)
    owned_px = W * H if not multi else ctx.tile_count(rank, world) * capi.TILE * capi.TILE
    alg_bytes = my_fetched * bs + 16 * owned_px
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms == kernel_ms and kernel_ms > 0 else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == args.workload and tj.get("tf") == args.tf and tj.get("n_gpus") == world:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    ran = ctx.last_kernel_flavour()
    # second denominator: what a plain device-to-device copy reaches on this GPU right now
    d2d_gbs = None
    if rank == 0:
        try:
            src = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB
            dstb = torch.empty_like(src)
            dstb.copy_(src)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dstb.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            d2d_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9  # bytes read + written
            del src, dstb
        except Exception:
            d2d_gbs = None

    out = {
        "metric": "Gsamples/s", "value": round(value, 4), "unit": "Gsamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "fps": round(1e3 / ms_per_step, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: ct-phantom-{n} RGBA32F voxels, {W}x{H}, {vname} shader, TF {args.tf}, "
                        f"step 1/{round(1 / step_size)} x {steps_count}, camera d=1.2 yaw=.6 pitch=.35",
            "partition": ("single GPU" if not multi else
                          f"64x64 image tiles interleaved over {world} GPUs + RCCL gather") +
                         (f", {nbuf} frames in flight" if nbuf > 1 else ", one frame at a time"),
            "composited_samples_per_frame": total_samples, "fetched_samples_per_frame": total_fetched,
            "covered_pixels": covered, "kernel_flavour": args.flavour, "kernel_flavour_resolved": ran,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 2) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
            "kernel": KERNEL_OF_FLAVOUR.get(ran, "march_kernel"), "kernel_ms": round(kernel_ms, 4), "bytes_per_sample": bs,
            "algorithmic_bytes_per_launch": alg_bytes,
            # with two frames in flight two launches overlap: each one's own duration (kernel_ms, what rocprofv3 reports
            # too) is longer than the time the GPU spends per frame; the rate of the overlapped pair is given as well
            "d2d_copy_gbs": round(d2d_gbs, 1) if d2d_gbs else None,
            "frac_of_d2d_copy": round(achieved / d2d_gbs, 4) if (achieved and d2d_gbs) else None,
            "launches_in_flight": nbuf,
            "achieved_per_frame_time": round(alg_bytes / (ms_per_step * 1e-3) / 1e9, 2),
        },
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(app, capi, variant, vols, W, H)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and os.environ.get("VR_BENCH_CHECK_FRAME"):
        # rehearsal aid: the gathered frame must equal a single-rank render of the same scene, bit for bit
        ctx.render_async(variant, 0, stream)
        torch.cuda.synchronize()
        ref, _, _ = ctx.download()
        same = all(bool(np.array_equal(ref.view(np.uint32), f.cpu().numpy().view(np.uint32))) for f in frames2)
        out["config"]["frame_equals_single_rank_render"] = same
    if rank == 0:
        print(json.dumps(out), flush=True)
    app.close()


if __name__ == "__main__":
    main()
