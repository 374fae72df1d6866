#!/usr/bin/env python3
"""Headline benchmark: Gsamples/s + fps of the volume ray-march compositing loop (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W           (N = 1 here; N > 1 under torch.distributed.run)

A "step" is one rendered frame of the hot path on synthetic input that is already resident in HBM.  At N = 1 the
workload is BASELINE.json configs[2] -- the one the metric is quoted on: ct-phantom-512 (512^3 RGBA32F voxels,
2 GiB), 1920x1080, BasicVolLightApp shader (TF lookup + interpolated central-difference gradient + Blinn-Phong
shade + opacity cut-off), default ramp TFs (R = 4096), 1/512 x 886 steps, camera distance 1.2 / yaw .6 / pitch .35.
At N > 1 the same frame is image-tile partitioned (64x64 tiles, tile t owned by rank t mod N), each rank renders
its tiles from its own replica of the volume and an RCCL gather over xGMI assembles the frame on rank 0
("strong" scaling: total work fixed).

Three legs are timed in one run, W warm-up + exactly K timed frames each, bracketed by barrier + device synchronise:
    serial      one frame at a time (the reference's interactive loop, Application.cpp:332-379): latency
    pipelined_one_frame_per_launch
                two launches in flight on two streams, one frame each (the next frame fills the SIMDs the longest rays
                of the previous one leave idle): the throughput leg of round 1 and early round 2, kept for comparison
    overlapped  two launches in flight, FOUR frames per launch (vr_render_batch_async / vr_mgpu_frames_async: one grid
                marches four frames, their workgroups interleaved so that the long ray chains of all four start first):
                throughput, at eight frames of delay.  `value` / `ms_per_step` are this leg's.  Every frame is marched
                in full and checked bit for bit against the one-at-a-time leg's frame.
value = composited samples of the whole frame / wall time per frame.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
    roofline      HBM roofline of the march kernel from MEASURED fabric bytes (rocprofv3 PMC passes run by this script
                  on the same scene: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) over the median HIP-event kernel
                  time of the serial leg; the effective-gather figure (algorithmic bytes that L1/L2 mostly serve) and the
                  VALU-issue occupancy are separate, clearly named fields
    parity        the same frame rendered by the CPU oracle, compared with the GPU frame (max abs, bit equality, counts)
    cpu_baseline  the oracle timed on the host cores of this box (all cores and one thread); N = 1 only
    regimes       (C3 only) the four {exact-0 air, noisy air} x {default ramp, zero-prefix TF} numbers, serial leg
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
N_SIMDS = 256 * 4      # CUs x SIMDs per CU (same table)

# which kernel a resolved flavour runs (include/vr.h, vr_set_kernel_flavour)
KERNEL_OF_FLAVOUR = {1: "march_kernel (no skipping)", 2: "march_wtb_light_kernel", 3: "march_wtb_light_kernel",
                     6: "march_kernel (one lane per ray)", 7: "march_dp_kernel (4 lanes per ray)",
                     8: "march_dp_kernel (2 lanes per ray)", 9: "march_kernel (one lane per ray, pipelined)",
                     10: "march_dp_kernel (4 lanes per ray, pipelined)", 11: "march_dp_kernel (2 lanes per ray, pipelined)",
                     12: "march_den_kernel (density plane, one lane per ray)"}

PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_RD"],
              ["GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"], ["TA_BUSY_avr", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]]


PMC_EXTRA = [["TCP_PENDING_STALL_CYCLES_sum"],
             ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INST_CYCLES_VMEM", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
             ["SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"]]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class stdout_to_stderr:
    """File-descriptor level: whatever native libraries print on stdout inside the block (RCCL's version banner, gloo's
    connection notes) goes to stderr, so that stdout carries the ONE JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def host_cores():
    """CPUs this process can really use: its affinity mask, cut down to the cgroup's CPU quota when one is set (the GPU
    boxes expose every host CPU in the mask but grant a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except Exception:  # noqa: BLE001
        pass
    return n


# ------------------------------------------------------------------------------------------------ PMC child / passes
def pmc_child(args):
    """`bench.py --pmc-child`: what the rocprofv3 counter passes run -- the same scene, a few synchronous frames (one
    launch at a time: counters of overlapping dispatches cannot be told apart), no torch, no timing."""
    from volumerendering_amd import host, workloads as wl
    n, W, H, vname = wl.WORKLOADS[args.workload]
    with host.Application(W, H, 0) as app:
        wl.build_scene(app, args.workload, args.tf, args.air, args.vol_n, quiet=True)
        if args.flavour:
            app.context().set_kernel_flavour(args.flavour)
        app.context().set_volume_layout(args.layout)
        app.context().set_arithmetic(1 if args.arith == "fused" else 0)
        if args.frames_per_launch > 1:
            # launches of several frames, one launch at a time, into the frame buffers of helper contexts (no torch here)
            from volumerendering_amd import capi
            ctx = app.context()
            n = min(4, args.frames_per_launch)
            others = [capi.Context(W, H, 0) for _ in range(n)]
            try:
                u = app.uniforms()
                for _ in range(5):
                    ctx.render_batch_async(capi.VARIANT_NAMES.index(vname), [u] * n, [o.frame_device_ptr() for o in others])  # the context's own stream
                    ctx.counters()  # waits for the launch
            finally:
                for o in others:
                    o.close()
            return
        for _ in range(4):
            app.OnRender()


def live_pmc(args, passes=PMC_PASSES, timeout_s=150, frames_per_launch=0):
    """Runs rocprofv3 --pmc passes (one counter group per run, with --kernel-trace only) on `bench.py --pmc-child` and
    returns {counter: mean per march-kernel launch}.  Empty dict when rocprofv3 is unavailable or a pass fails."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {}, "rocprofv3 not found"
    out = {}
    keep = os.environ.get("VR_BENCH_PMC_DIR")  # keep the raw csv files there (for profiles/)
    tmp = keep or tempfile.mkdtemp(prefix="vr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    note = ""
    for i, ctrs in enumerate(passes):
        d = os.path.join(tmp, f"pass{i}")
        cmd = [prof, "--pmc", *ctrs, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--pmc-child", "--workload", args.workload, "--tf", args.tf, "--air", args.air,
               "--flavour", str(args.flavour), "--vol-n", str(args.vol_n), "--layout", str(args.layout), "--arith", args.arith]
        if frames_per_launch > 1:
            cmd += ["--frames-per-launch", str(frames_per_launch)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        except Exception as e:  # noqa: BLE001
            note += f"pass {ctrs[0]}: {type(e).__name__}; "
            continue
        if r.returncode != 0:
            note += f"pass {ctrs[0]}: rc {r.returncode} {r.stderr.decode(errors='replace')[-160:]!r}; "
            continue
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per = {}
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if "march" not in row["Kernel_Name"]:
                        continue
                    per.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), 0.0)
                    per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
            for name, dd in per.items():
                vals = [dd[k] for k in sorted(dd)][1:] or list(dd.values())  # drop the first (cold) launch
                out[name] = sum(vals) / len(vals)
        # duration of the PROFILED march launches of this pass (they run slower than un-profiled ones: counters are only
        # comparable with a time base taken under the same profiler)
        if "GRBM_GUI_ACTIVE" in ctrs:
            for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
                with open(f, newline="") as fh:
                    durs = [(int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                            for r in csv.DictReader(fh) if "march" in r["Kernel_Name"]]
                durs = [x[1] for x in sorted(durs)][1:]
                if durs:
                    out["_profiled_march_ms"] = sum(durs) / len(durs) / 1e6
    if not keep:
        shutil.rmtree(tmp, ignore_errors=True)
    return out, note


# ------------------------------------------------------------------------------------------------ CPU oracle legs
def oracle_legs(app, variant, vols, W, H, gpu_frame, gpu_samples, budget_s=25.0, fused=False):
    """The CPU oracle (a port: plain-C restatement of the WGSL) on the SAME frame: all host cores on the densest
    regular pixel grid that fits the budget (the whole frame for C1-C4), compared pixel by pixel with the GPU frame,
    and one thread on a sparser grid.  Checker and reported baseline only; never part of the product path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import host_ref as hr
    import oracle_binding as ob
    from volumerendering_amd import workloads as wl

    ob.set_arithmetic(1 if fused else 0)  # the checker's mode follows the mode of the frame it checks
    cores = host_cores()
    ub, volumes, tfs = wl.oracle_inputs(app, vols)
    uo = hr.Uniforms.from_buffer_copy(ub)

    def grid(stride):
        ys, xs = np.meshgrid(np.arange(stride // 2, H, stride), np.arange(stride // 2, W, stride), indexing="ij")
        return np.stack([xs.ravel(), ys.ravel()], axis=1).astype(np.int32)

    def run(stride, threads):
        pxy = grid(stride)
        t0 = time.perf_counter()
        out, n = ob.render_pixels(variant, uo, volumes, tfs, W, H, pxy, nthreads=threads)
        return pxy, out, n, time.perf_counter() - t0

    # calibrate on 1 pixel of every 16 x 16, then the densest grid that fits the budget
    _, _, n_c, dt_c = run(16, cores)
    per_px = dt_c / max(1, len(grid(16)))
    stride = 1
    while stride < 16 and per_px * (W // stride) * (H // stride) > budget_s:
        stride += 1
    pxy, out, n_all, dt_all = run(stride, cores)
    got = gpu_frame[pxy[:, 1], pxy[:, 0]]
    bit_equal = bool(np.array_equal(got.view(np.uint32), out.view(np.uint32)))
    with np.errstate(invalid="ignore"):
        max_abs = float(np.nanmax(np.abs(got - out))) if out.size else 0.0
    parity = {"max_abs": max_abs, "bit_equal": bit_equal, "pixels": int(pxy.shape[0]),
              "of_frame": "whole frame" if stride == 1 else f"one pixel of every {stride} x {stride}",
              "oracle_composited_samples": int(n_all), "tolerance": 1e-4}
    if stride == 1:
        parity["samples_equal"] = bool(n_all == gpu_samples)
    # one thread: a grid sized for ~10 s, from a one-thread calibration on 1 pixel of every 32 x 32
    _, _, n_c1, dt_c1 = run(32, 1)
    rate1 = n_c1 / dt_c1 if dt_c1 > 0 else 1.0
    s1 = stride
    while s1 < 32 and (n_all * (stride / s1) ** 2) / max(rate1, 1.0) > 10.0:
        s1 += 1
    p1, _, n_1, dt_1 = run(s1, 1)
    base = {"value": round(n_all / dt_all / 1e9, 6), "unit": "Gsamples/s", "cores": cores,
            "cores_note": f"threads used = CPUs in this process's affinity mask ({len(os.sched_getaffinity(0))}), cut to the cgroup CPU quota when "
                          "one is readable; speedup_over_one_thread says how many of them the box really grants",
            "kind": "port",
            "implementation": "oracle/vr_oracle.c (scalar C restatement of the WGSL, pthread over pixels)",
            "sample": f"{parity['of_frame']} of the {W}x{H} frame ({pxy.shape[0]} rays, {n_all} composited samples, {dt_all:.2f} s)",
            "speedup_over_one_thread": round((n_all / dt_all) / (n_1 / dt_1), 2) if dt_1 > 0 and n_1 else None,
            "one_thread": {"value": round(n_1 / dt_1 / 1e9, 6), "cores": 1,
                           "sample": f"one pixel of every {s1} x {s1} ({p1.shape[0]} rays, {n_1} composited samples, {dt_1:.2f} s)"}}
    return parity, base


# ------------------------------------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C3", choices=["C1", "C2", "C3", "C4", "C5"])
    ap.add_argument("--tf", default="default", choices=["default", "thin", "prefix", "zero"])
    ap.add_argument("--air", default="exact0", choices=["exact0", "noisy"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 counter passes (roofline.traffic = null)")
    ap.add_argument("--no-regimes", action="store_true", help="skip the air x TF regime table (C3)")
    ap.add_argument("--pmc-extra", action="store_true", help="more counter passes (L1 / TA / wait states); all counters go into `pmc`")
    ap.add_argument("--flavour", type=int, default=0)
    ap.add_argument("--arith", default="separate", choices=["separate", "fused"],
                    help="vr_set_arithmetic for the headline legs (the other mode is timed as the `arith_ab` leg)")
    ap.add_argument("--layout", type=int, default=0, choices=[0, 1, 2],
                    help="vr_set_volume_layout: 0 density plane for .a fetches, 1 the reference's vec4 voxels only, 2 = 0 + lit "
                         "gradients derived on the fly from the plane")
    ap.add_argument("--vol-n", type=int, default=0, help="experiment: smaller volume, same frame and stepping")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="frames in flight in the overlapped leg (1 = that leg is a second serial leg)")
    ap.add_argument("--exp-mode", type=int, default=0, help="experiment: fragmentMode 1-4 (ray set-up only)")
    ap.add_argument("--exp-steps", type=int, default=-1, help="experiment: override stepsCount")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="throughput leg: frames marched by one launch (vr_render_batch_async / vr_mgpu_frames_async; 1..4, "
                         "0 = 4).  The leg with one frame per launch is timed and reported beside it")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)

    import torch
    from volumerendering_amd import capi, host, workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # VR_BENCH_DEVICE / VR_BENCH_BACKEND=gloo exist only to rehearse the N > 1 bookkeeping on a one-GPU box (all ranks on
    # device 0, tiles gathered with gloo through host memory: RCCL refuses two ranks on one device); the real multi-GPU
    # run uses one GPU per rank and the C++ frame loop of libvr_mgpu.so (ncclGather over xGMI).
    device_index = int(os.environ.get("VR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("VR_BENCH_BACKEND", "rccl")
    torch.cuda.set_device(device_index)
    # VR_BENCH_SELF_GATHER=1 (rehearsal): take the multi-rank code path -- tile render, RCCL gather, un-permute -- with
    # a world of one, so that the RCCL calls and their stream ordering can be exercised on a one-GPU box
    multi = world > 1 or bool(os.environ.get("VR_BENCH_SELF_GATHER"))
    dist = None
    if multi:
        # torch.distributed (gloo) is the CONTROL plane only: it carries the 128-byte RCCL id from rank 0 to the others and
        # provides the barrier of the timing contract; no frame data goes through it (except in the gloo rehearsal)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist_mod
        dist = dist_mod
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo")

    n, W, H, vname = wl.WORKLOADS[args.workload]
    app = host.Application(W, H, device_index)
    variant, vols = wl.build_scene(app, args.workload, args.tf, args.air, args.vol_n)
    ctx = app.context()
    if args.flavour:
        ctx.set_kernel_flavour(args.flavour)
    ctx.set_volume_layout(args.layout)
    ctx.set_arithmetic(1 if args.arith == "fused" else 0)
    if args.exp_mode or args.exp_steps >= 0:  # experiments only: not the BASELINE workload any more
        app.set_params(fragment_mode=args.exp_mode, steps_count=args.exp_steps)
        app.OnUpdate()
    steps_count, step_size = app.stepping()

    tpr_max = ctx.tile_count(0, world)
    tile_floats = capi.TILE * capi.TILE * 4
    # the C++ multi-GPU loop keeps as many frames in flight as it has buffer sets (VR_MGPU_SLOTS, default 2)
    n_slots = max(1, min(4, int(os.environ.get("VR_MGPU_SLOTS", "2"))))
    max_flight = n_slots if (multi and backend == "rccl") else (2 if multi else max(2, args.in_flight))
    # context-owned streams probed to really run side by side (two arbitrary HIP streams may share a hardware queue and
    # serialise: vr_stream, include/vr.h)
    streams = [ctx.stream(i) for i in range(max_flight)]
    torch.cuda.synchronize()
    frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(max_flight)] if not multi else []
    mg = None
    if multi and backend == "rccl":
        from volumerendering_amd import mgpu
        idt = torch.zeros(mgpu.ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            idt = torch.tensor(list(mgpu.unique_id()), dtype=torch.uint8)
        with stdout_to_stderr():
            dist.broadcast(idt, 0)
            mg = mgpu.MultiGpu(ctx.h, rank, world, bytes(idt.tolist()))  # ncclCommInitRank: collective
            mg.frame_async(variant)  # first collective (RCCL sets its channels up here), outside every timed region
            mg.wait()
    elif multi:  # gloo rehearsal
        my_tiles = [torch.zeros((tpr_max * tile_floats,), dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros((world, tpr_max * tile_floats), dtype=torch.float32, device="cuda") for _ in range(2)] \
            if rank == 0 else [None, None]
        frames = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]

    # frames per launch of the throughput leg (the one-frame-at-a-time leg never batches)
    part_world = max(world, int(os.environ.get("VR_MGPU_EXP_SHARE", "1"))) if multi else 1
    fpl = max(1, min(4, args.frames_per_launch if args.frames_per_launch > 0 else 4))
    if multi and mg is None:
        fpl = 1  # (the gloo rehearsal renders frame by frame)
    batch_frames = [[torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(fpl)] for _ in range(max_flight)] \
        if (fpl > 1 and not multi) else []

    if mg is not None and fpl > 1:  # size the loop's buffer sets for fpl frames per launch outside every timed region
        with stdout_to_stderr():
            for _ in range(max_flight):  # (every buffer set: the final comparison reads all of their frames)
                mg.frames_async(variant, [app.uniforms()] * fpl)
            mg.wait()

    batch_written = set()  # (buffer set, frame of the launch) the batched launches have written so far

    def run_frames(n_frames, nbuf, fpl=1):
        if not multi:
            if fpl > 1:  # one launch carries fpl frames (same camera in this bench; each frame marched in full)
                u, k, launch = app.uniforms(), 0, 0
                while k < n_frames:
                    n = min(fpl, n_frames - k)
                    ctx.render_batch_async(variant, [u] * n, [t.data_ptr() for t in batch_frames[launch % nbuf][:n]], streams[launch % nbuf])
                    batch_written.update((launch % nbuf, j) for j in range(n))
                    k += n
                    launch += 1
                return
            for k in range(n_frames):
                ctx.render_async(variant, frames[k % nbuf].data_ptr(), streams[k % nbuf])
            return
        if mg is not None:
            # the C++ loop: every rank renders its tiles, ncclGather, the root un-permutes; two buffer sets, so up to two
            # launches are in flight -- nbuf = 1 waits for every frame before the next one is enqueued
            if fpl > 1 and nbuf > 1:
                u, k = app.uniforms(), 0
                while k < n_frames:
                    n = min(fpl, n_frames - k)
                    mg.frames_async(variant, [u] * n)
                    k += n
                mg.wait()
                return
            for _ in range(n_frames):
                mg.frame_async(variant)
                if nbuf == 1:
                    mg.wait()
            mg.wait()
            return
        for k in range(n_frames):  # gloo rehearsal: synchronous, through host memory
            b = k & 1
            ctx.render_tiles_async(variant, rank, world, my_tiles[b].data_ptr(), streams[0])
            host_list = [torch.empty(my_tiles[b].numel()) for _ in range(world)] if rank == 0 else None
            dist.gather(my_tiles[b].cpu(), host_list, dst=0)
            if rank == 0:
                gathered[b].copy_(torch.stack(host_list))
                ctx.unpack_tiles_async(gathered[b].data_ptr(), world, frames[b].data_ptr(), streams[0])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_leg(nbuf, n_warm, n_steps, min_events=0, fpl=1):
        """W warm-up + exactly K timed frames, barrier + synchronise on both sides; the MAX over ranks of the wall time.
        Returns (seconds, HIP-event kernel durations of the timed launches [+ extra serial frames up to min_events])."""
        ctx.hint_frames_in_flight(nbuf)  # what this leg's caller does: steers the default kernel choice (vr.h)
        run_frames(n_warm, nbuf, fpl)
        sync_all()
        ctx.reset_kernel_times()
        t0 = time.perf_counter()
        run_frames(n_steps, nbuf, fpl)
        sync_all()
        dt = time.perf_counter() - t0
        if n_steps < min_events:  # the median below wants >= 20 event-timed frames (outside the K-step region)
            run_frames(min_events - n_steps, nbuf, fpl)
            sync_all()
        kt = ctx.kernel_times(min(max(n_steps, min_events), 256))
        if dist is not None:  # MAX over ranks
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kt

    dt_serial, kt_serial = timed_leg(1, args.warmup, args.steps, min_events=20)
    # The legs' launches are timed from their own per-workgroup records (vr_kernel_times' default: two HIP events around
    # every launch cost the frame's stream 11 us, which the one-frame-at-a-time leg would pay).  The roofline's denominator is
    # measured the contract's way: HIP events on the launch stream around each of 20 more one-at-a-time frames, outside the
    # timed regions.
    ctx.set_kernel_timing(True)
    _, kt_events = timed_leg(1, 3, 20)
    ctx.set_kernel_timing(False)
    nbuf_over = max_flight if mg is not None else min(args.in_flight, max_flight)
    if nbuf_over == 1 and args.frames_per_launch <= 0:
        fpl = 1  # (--in-flight 1: one frame at a time in this leg too, unless --frames-per-launch asks for batched launches)
    # two launches in flight, one frame each (round 1's and early round 2's throughput leg; kept for comparison) ...
    dt_pipe, kt_pipe = timed_leg(nbuf_over, args.warmup, args.steps) if fpl > 1 else (None, None)
    # ... and the throughput leg proper: two launches in flight, fpl frames per launch
    dt_over, kt_over = timed_leg(nbuf_over, args.warmup, args.steps, fpl=fpl)
    if batch_frames:  # the frames of the batched launches must equal the one-at-a-time leg's frame bit for bit
        if not all(bool(torch.equal(batch_frames[b][j].view(torch.int32), frames[0].view(torch.int32))) for b, j in sorted(batch_written)):
            raise SystemExit("bench.py: a frame of a batched launch differs from the single-frame render")

    # composited samples / covered pixels / samples whose voxels were fetched, for this rank's share of the frame
    my_samples, my_covered, my_fetched = ctx.counters()
    if mg is not None:  # summed over the ranks by the C++ driver (ncclAllReduce)
        (total_samples, covered, total_fetched), _ = mg.reduce(0.0)
    elif dist is not None:
        s = torch.tensor([my_samples, my_covered, my_fetched], dtype=torch.int64)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples, covered, total_fetched = int(s[0].item()), int(s[1].item()), int(s[2].item())
    else:
        total_samples, covered, total_fetched = my_samples, my_covered, my_fetched
    ran = ctx.last_kernel_flavour()
    layout_flags = ctx.volume_layout(2 if vname == "VOLUME_MASK" else 0)

    def leg(dt, kt, nbuf):
        ms = dt / args.steps * 1e3
        return {"frames_in_flight": nbuf, "ms_per_step": round(ms, 4), "fps": round(1e3 / ms, 2),
                "value": round(total_samples / (dt / args.steps) / 1e9, 3),
                "fetched_gsamples_per_s": round(total_fetched / (dt / args.steps) / 1e9, 3),
                "kernel_ms_median": round(float(np.median(kt)), 4) if len(kt) else None,
                "kernel_ms_mean": round(float(np.mean(kt)), 4) if len(kt) else None,
                "kernel_ms_p10_p90": [round(float(np.percentile(kt, 10)), 4), round(float(np.percentile(kt, 90)), 4)] if len(kt) else None,
                "kernel_events": int(len(kt))}

    serial, over = leg(dt_serial, kt_serial, 1), leg(dt_over, kt_over, nbuf_over)
    serial["kernel_ms_source"] = "first workgroup start .. last workgroup end of each timed launch (100 MHz device clock, from the launch's records)"
    serial["kernel_ms_hip_events_median"] = round(float(np.median(kt_events)), 4) if len(kt_events) else None
    serial["kernel_ms_hip_events_note"] = "HIP events on the launch stream around 20 further one-at-a-time launches; the roofline's denominator"
    pipelined = None
    if dt_pipe is not None:
        pipelined = leg(dt_pipe, kt_pipe, nbuf_over)
        pipelined["launches_in_flight"], pipelined["frames_per_launch"] = nbuf_over, 1
    over["launches_in_flight"] = nbuf_over
    over["frames_per_launch"] = fpl
    over["frames_in_flight"] = nbuf_over * fpl
    if fpl > 1:  # one event pair per LAUNCH: the durations are those of launches that carry fpl frames each
        over["kernel_ms_note"] = f"kernel_ms_* are per launch of {fpl} frames"
    # the other arithmetic mode, same scene, both legs (20 frames each; outside the K-step regions above)
    arith_ab = None
    if not multi:
        other = "separate" if args.arith == "fused" else "fused"
        ctx.set_arithmetic(1 if other == "fused" else 0)
        dt_s2, kt_s2 = timed_leg(1, 5, 20)
        dt_o2, kt_o2 = timed_leg(nbuf_over, 5, 20, fpl=fpl)
        cs2, _, fs2 = ctx.counters()
        arith_ab = {"arithmetic": other,
                    "serial": {"ms_per_step": round(dt_s2 / 20 * 1e3, 4), "kernel_ms_median": round(float(np.median(kt_s2)), 4),
                               "value": round(cs2 / (dt_s2 / 20) / 1e9, 3)},
                    "overlapped": {"frames_in_flight": nbuf_over, "ms_per_step": round(dt_o2 / 20 * 1e3, 4),
                                   "value": round(cs2 / (dt_o2 / 20) / 1e9, 3)},
                    "composited_samples_per_frame": cs2, "fetched_samples_per_frame": fs2,
                    "note": "vr_set_arithmetic: per-sample a*b+c with one rounding (fused) instead of two; bit-exact against the "
                            "oracle's mode of the same name (tests/), ray placement identical in both modes"}
        ctx.set_arithmetic(1 if args.arith == "fused" else 0)
        ctx.render_async(variant, frames[0].data_ptr(), streams[0])  # frames[0] = the headline mode's frame again
        torch.cuda.synchronize()
    kernel_ms = serial["kernel_ms_hip_events_median"] or serial["kernel_ms_median"]
    bs = wl.BYTES_PER_SAMPLE[vname]
    owned_px = W * H if not multi else ctx.tile_count(rank, world) * capi.TILE * capi.TILE
    alg_fetched = my_fetched * bs + 16 * owned_px   # what the kernel's loads ask for (mostly served by L1 / L2)
    alg_composited = my_samples * bs + 16 * owned_px  # SURVEY 8d's figure: every composited sample priced as a fetch
    vol_bytes = sum(int(np.prod(v.GetSize())) * 16 for v in vols)

    if mg is not None:
        # every frame of the last launch into each buffer set (the throughput leg ran last)
        gpu_frames = [mg.download_batch_frame(w, f, W, H) for w in range(min(2, n_slots)) for f in range(fpl)] if rank == 0 else []
    else:
        gpu_frames = [f.cpu().numpy() for f in frames[:2]] if rank == 0 else []
    gpu_frame = gpu_frames[0] if gpu_frames else None

    # ---- roofline: measured fabric bytes (rocprofv3 PMC passes on the same scene, launched from here) -------------
    pmc, pmc_note = ({}, "skipped")
    if rank == 0 and world == 1 and not multi and not args.no_live_pmc:
        t0 = time.time()
        pmc, pmc_note = live_pmc(args, PMC_PASSES + (PMC_EXTRA if args.pmc_extra else []))
        log(f"[bench] rocprofv3 counter passes: {time.time() - t0:.1f}s {pmc_note}")
    traffic = None
    traffic_source = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        traffic_source = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this script on the same scene, one launch at "
                          "a time; (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 FETCH_SIZE correction; counts "
                          "Infinity-Cache hits too: an upper estimate of HBM bytes)")
    else:
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
        try:
            tj = json.load(open(tpath))
            if (tj.get("workload"), tj.get("tf"), tj.get("air", "exact0"), tj.get("n_gpus")) == (args.workload, args.tf, args.air, world):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = f"STALE: committed profiles/pmc_traffic_latest.json (live passes: {pmc_note})"
        except Exception:  # noqa: BLE001
            pass
    achieved = traffic / (kernel_ms * 1e-3) / 1e9 if (traffic and kernel_ms) else None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
        "kernel": KERNEL_OF_FLAVOUR.get(ran, "march_kernel"), "kernel_ms": kernel_ms,
        "kernel_ms_is": "median HIP-event duration of 20 one-at-a-time launches, events recorded on the launch stream",
        "frac_overlapped": round(traffic / (over["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "bytes_per_sample": bs,
        "effective_gather": {
            "note": "algorithmic bytes the loads ask for; L1 / L2 serve most of them, so this is NOT a roofline fraction",
            "fetched_bytes_per_launch": alg_fetched, "composited_bytes_per_launch": alg_composited,
            "fetched_gbs": round(alg_fetched / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else None,
            "composited_gbs": round(alg_composited / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else None},
        "compulsory_floor_bytes": vol_bytes + 16 * owned_px,
        "compulsory_floor_note": "stored volume bytes + frame write; exact empty-space skipping never touches inert bricks, so "
                                 "measured traffic may be below this",
    }
    if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
        roofline["valu"] = {
            "insts_per_launch": pmc.get("SQ_INSTS_VALU"), "active_quad_cycles": pmc["SQ_ACTIVE_INST_VALU"],
            "gpu_cycles_per_launch": cyc,
            "busy_frac": round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * cyc), 4),
            "note": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of the launch's SIMD-cycles "
                    "spent issuing vector ALU work (serial, profiled launches)"}
        # effective shader clock: cycles of the profiled launches over THEIR duration (kernel trace of the same pass)
        prof_ms = pmc.get("_profiled_march_ms")
        clock_hz = cyc / (prof_ms * 1e-3) if prof_ms else None
        roofline["valu"]["profiled_kernel_ms"] = round(prof_ms, 4) if prof_ms else None
        over_cyc = over["ms_per_step"] * 1e-3 * clock_hz if clock_hz else None
        roofline["valu"]["busy_frac_overlapped"] = round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * over_cyc), 4) if over_cyc else None
        roofline["valu"]["clock_ghz"] = round(clock_hz / 1e9, 3) if clock_hz else None
        # the vector-memory (texture addresser / L1) data path: every lane's bytes are returned at 64 B / clk / CU
        tf_bytes = {"BASIC": 40, "LIGHT": 40, "LIGHT_INSHADER": 40, "TF_CALIB": 40}.get(vname, 80)
        l1_bytes = my_fetched * (bs + tf_bytes) + my_samples // 1  # + one distance-field byte per executed step (lower bound)
        l1_peak = 64.0 * 256 * clock_hz / 1e9 if clock_hz else None
        roofline["l1"] = {
            "bytes_per_launch": l1_bytes, "peak_gbs": round(l1_peak, 1) if l1_peak else None,
            "achieved_gbs": round(l1_bytes / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else None,
            "frac": round(l1_bytes / (kernel_ms * 1e-3) / 1e9 / l1_peak, 4) if (l1_peak and kernel_ms) else None,
            "frac_overlapped": round(l1_bytes / (over["ms_per_step"] * 1e-3) / 1e9 / l1_peak, 4) if l1_peak else None,
            "ta_busy_frac": round(pmc["TA_BUSY_avr"] / cyc, 4) if "TA_BUSY_avr" in pmc else None,
            "ta_busy_frac_overlapped": round(pmc["TA_BUSY_avr"] / over_cyc, 4) if ("TA_BUSY_avr" in pmc and over_cyc) else None,
            "note": "bytes the lanes receive from L1 (corner voxels + transfer-function texels of every fetched sample) against the "
                    "64 B/clk/CU return path of the vector memory pipeline at the measured clock; TA_BUSY_avr = busy cycles of the "
                    "texture addressers per launch"}
        # the same two units measured on launches of `fpl` frames (one launch at a time: counters of overlapping dispatches
        # cannot be told apart), instead of scaled from the one-frame launches
        if fpl > 1 and not args.no_live_pmc:
            bp, bnote = live_pmc(args, [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"], ["GRBM_GUI_ACTIVE", "TA_BUSY_avr"]], frames_per_launch=fpl)
            if "SQ_ACTIVE_INST_VALU" in bp and "GRBM_GUI_ACTIVE" in bp:
                bcyc = bp["GRBM_GUI_ACTIVE"] / 8.0
                roofline["batched_launch_measured"] = {
                    "frames_per_launch": fpl, "launches_in_flight": 1,
                    "valu_busy_frac": round(bp["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * bcyc), 4),
                    "ta_busy_frac": round(bp["TA_BUSY_avr"] / bcyc, 4) if "TA_BUSY_avr" in bp else None,
                    "valu_insts_per_frame": round(bp.get("SQ_INSTS_VALU", 0.0) / fpl),
                    "profiled_ms_per_frame": round(bp["_profiled_march_ms"] / fpl, 4) if "_profiled_march_ms" in bp else None,
                    "note": "rocprofv3 --pmc passes on launches that carry several frames, one launch at a time; the throughput "
                            "leg keeps two such launches in flight"}
            elif bnote:
                log(f"[bench] batched counter passes: {bnote}")
        fr = {"hbm": roofline["frac_overlapped"] or 0, "valu-issue": roofline["valu"]["busy_frac_overlapped"] or 0,
              "l1-return-path": roofline["l1"]["ta_busy_frac_overlapped"] or roofline["l1"]["frac_overlapped"] or 0}
        roofline["limiter"] = max(fr, key=fr.get) + " (throughput leg); longest-ray tail on top of it in the serial leg"
        roofline["limiter_fractions_overlapped"] = fr
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
        roofline["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4)
    # second denominator: what a plain device-to-device copy reaches on this GPU right now
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
            d2d = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9  # bytes read + written
            roofline["d2d_copy_gbs"] = round(d2d, 1)
            roofline["frac_of_d2d_copy"] = round(achieved / d2d, 4) if achieved else None
            del src, dstb
        except Exception:  # noqa: BLE001
            pass

    part = "single GPU" if not multi else f"64x64 image tiles interleaved over {world} GPUs + " + \
        ((mg.backend() + " (C++ frame loop, libvr_mgpu.so)") if mg is not None else "gloo gather through host memory (rehearsal)")
    out = {
        "metric": "Gsamples/s", "value": over["value"], "unit": "Gsamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": over["ms_per_step"], "fps": over["fps"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: ct-phantom-{n} RGBA32F voxels, {W}x{H}, {vname} shader, TF {args.tf}, air {args.air}, "
                        f"step 1/{round(1 / step_size)} x {steps_count}, camera d=1.2 yaw=.6 pitch=.35",
            "partition": part,
            "value_is": f"throughput leg: {nbuf_over} launches in flight x {fpl} frame(s) per launch, every frame marched in full; "
                        "one frame at a time (the latency an interactive viewer sees) in `serial`",
            "composited_samples_per_frame": total_samples, "fetched_samples_per_frame": total_fetched,
            "covered_pixels": covered, "kernel_flavour": args.flavour, "kernel_flavour_resolved": ran,
            "volume_layout": ("density plane for .a fetches" + (", corner gradients derived on the fly" if layout_flags & 4 else ""))
                             if args.layout != 1 else "reference vec4 voxels only",
        },
        "serial": serial, "overlapped": over, "roofline": roofline,
    }
    out["config"]["arithmetic"] = args.arith
    if pipelined:
        out["pipelined_one_frame_per_launch"] = pipelined
    if arith_ab:
        out["arith_ab"] = arith_ab
    if args.pmc_extra:
        out["pmc"] = {"per": "march-kernel launch, mean of the profiled launches (one at a time), first launch dropped",
                      **{k: v for k, v in pmc.items() if not k.startswith("_")}}

    # ---- the regime table (C3): {exact-0 air, noisy air} x {default ramp, zero-prefix TF}, serial leg -------------
    if rank == 0 and not multi and args.workload == "C3" and not args.no_regimes and not args.vol_n:
        current = (args.air, args.tf)

        def scene(air, tf):
            nonlocal app, variant, vols, ctx, current
            if current == (air, tf):
                return
            if current == (air, "default") and tf != "default":
                wl.apply_tf(app, vname, tf)  # same volume: edit the table through the control-point surface
                app.OnUpdate()
            else:
                app.close()
                vols = None
                app = host.Application(W, H, device_index)
                variant, vols = wl.build_scene(app, "C3", tf, air, quiet=True)
                ctx = app.context()
                if args.flavour:
                    ctx.set_kernel_flavour(args.flavour)
                ctx.set_volume_layout(args.layout)
                ctx.set_arithmetic(1 if args.arith == "fused" else 0)
                streams[:] = [ctx.stream(i) for i in range(max_flight)]  # the streams belong to the context
            current = (air, tf)

        regimes = []
        ctx.hint_frames_in_flight(1)
        for air in wl.AIR_KINDS:
            for tf in ("default", "prefix"):
                scene(air, tf)
                ctx.set_kernel_timing(True)  # HIP events on the launch stream (a rebuilt scene has a new context)
                for _ in range(3):
                    ctx.render_async(variant, frames[1].data_ptr(), streams[0])
                torch.cuda.synchronize()
                ctx.reset_kernel_times()
                for _ in range(20):
                    ctx.render_async(variant, frames[1].data_ptr(), streams[0])
                torch.cuda.synchronize()
                ms = float(np.median(ctx.kernel_times(20)))
                cs, _, fs = ctx.counters()
                regimes.append({"air": air, "tf": tf, "kernel_ms": round(ms, 4), "composited_gsamples_per_s": round(cs / ms / 1e6, 2),
                                "fetched_gsamples_per_s": round(fs / ms / 1e6, 2), "composited": cs, "fetched": fs,
                                "kernel_flavour_resolved": ctx.last_kernel_flavour()})
        out["regimes"] = regimes
        out["regimes_note"] = ("one frame at a time, median HIP-event kernel ms of 20 frames; noisy air = raw 0..80 outside the body; "
                               "prefix = preset-style opacity table with a real zero prefix (workloads.py)")
        scene(args.air, args.tf)  # the CPU legs below want the headline scene again
        ctx.set_kernel_timing(False)

    if rank == 0 and world == 1 and not multi and not args.no_cpu_baseline:
        parity, base = oracle_legs(app, variant, vols, W, H, gpu_frame, total_samples, fused=(args.arith == "fused"))
        parity["arithmetic"] = args.arith
        out["parity"] = parity
        out["cpu_baseline"] = base
    if rank == 0 and multi and not os.environ.get("VR_MGPU_EXP_SHARE"):
        # the gathered frames must equal a single-rank render of the same scene, bit for bit (cheap: one more frame)
        ctx.render_async(variant, 0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ref, _, _ = ctx.download()
        out["config"]["frame_equals_single_rank_render"] = all(bool(np.array_equal(ref.view(np.uint32), f.view(np.uint32))) for f in gpu_frames)
    if mg is not None:
        mg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    app.close()


if __name__ == "__main__":
    main()
