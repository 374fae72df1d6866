#!/usr/bin/env python3
"""Headline benchmark: Gsamples/s + fps of the volume ray-march compositing loop (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1: this process.  N > 1: one process per GPU -- under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
environment) or, when those are absent, spawned by this script itself BEFORE anything touches a GPU (`launch_ranks`).

A "step" is one rendered frame of the hot path on synthetic input that is already resident in HBM.  At N = 1 the
workload is BASELINE.json configs[2] -- the one the metric is quoted on: ct-phantom-512 (512^3 RGBA32F voxels,
2 GiB), 1920x1080, BasicVolLightApp shader (TF lookup + interpolated central-difference gradient + Blinn-Phong
shade + opacity cut-off), default ramp TFs (R = 4096), 1/512 x 886 steps.  At N > 1 the same frame is image-tile
partitioned (64x64 tiles, tile t owned by rank t mod N), each rank renders its tiles from its own replica of the
volume and an RCCL gather over xGMI assembles the frame on rank 0 ("strong" scaling: total work fixed).

Camera: a TURNTABLE.  Frame g of every leg is rendered with the camera of BASELINE.md (distance 1.2, yaw .6, pitch .35)
rotated g times by Camera::Rotate(2 px, 0) -- yaw + g * 0.01 rad, what the reference's mouse handler does between two
frames (App/src/Application.cpp:381-457 -> App/src/Camera.cpp:146-152): no two frames of a leg, and no two frames of one
batched launch, share uniforms.  Every leg starts at g = 0, so the legs render the same sequence.

Legs, W warm-up + exactly K timed frames each, bracketed by barrier + device synchronise:
    serial      one frame at a time (the reference's interactive loop, Application.cpp:332-379) -- SURVEY 8d's t_frame.
                `value` / `ms_per_step` / `fps` ARE THIS LEG'S.
    serial_with_present
                the same + the output merge into BGRA8 (vr_present_async, PipelineBuilder.cpp:142-154) behind every frame
    pipelined_one_frame_per_launch
                two launches in flight on two streams, one frame each
    overlapped  two launches in flight, FOUR frames per launch (vr_render_batch_async / vr_mgpu_frames_async), four
                different cameras per launch: throughput at eight frames of delay, reported beside the headline
value = composited samples of the K timed frames (counted per camera in an untimed pass) / wall time of the leg.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
    roofline      HBM roofline of the march kernel from MEASURED fabric bytes (rocprofv3 PMC passes run by this script
                  on the same turntable: FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) over the median HIP-event kernel
                  time of the serial leg; `turntable_vs_identical` has the same counters for identical frames
    parity        frame 0 rendered by the CPU oracle, compared with the GPU frame (max abs, bit equality, counts)
    cpu_baseline  the oracle timed on the host cores of this box (all cores and one thread, one pixel grid); N = 1 only
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
# roofline.valu.busy_frac of a kernel that saturates the vector ALUs, for this kernel's instruction mix (about half plain, half
# packed / converting instructions): tools/ubench/valu_issue.hip under the same counters reads 1.5-1.7 for v_fma_f32 alone and
# about 1.2 for v_pk_fma_f32 alone (profiles/r03_valu_issue_*.txt)
VALU_SATURATION = 1.4

# which kernel a resolved flavour runs (include/vr.h, vr_set_kernel_flavour)
KERNEL_OF_FLAVOUR = {1: "march_kernel (no skipping)", 2: "march_wtb_light_kernel", 3: "march_wtb_light_kernel",
                     6: "march_kernel (one lane per ray)", 7: "march_dp_kernel (4 lanes per ray)",
                     8: "march_dp_kernel (2 lanes per ray)", 9: "march_kernel (one lane per ray, pipelined)",
                     10: "march_dp_kernel (4 lanes per ray, pipelined)", 11: "march_dp_kernel (2 lanes per ray, pipelined)",
                     12: "march_pw_kernel (persistent wavefronts, TF slot 0 in LDS)",
                     13: "march_pw_kernel (persistent wavefronts, TF slot 0 in LDS, corner loads pipelined)",
                     18: "march_kernel (one lane per ray, slot arithmetic from LDS tables)",
                     16: "march_p2_kernel (persistent wavefronts, indexed corner loads two steps ahead, slot tables in LDS, no skipping)",
                     17: "march_p2_kernel (persistent wavefronts, approach loop, indexed corner loads two steps ahead, slot tables in LDS, skipping decided ahead of the loads)"}

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
TURN_PX = 2.0  # Camera::Rotate(2 px, 0) per frame: yaw += 2 * m_RotateSens (0.005 rad, App/src/Camera.h:63) = 0.573 degrees


def camera_sequence(app, n, turntable=True):
    """Uniforms of frames 0 .. n-1: the scene's camera rotated once more per frame by Camera::Rotate (what the reference's
    mouse handler does, App/src/Application.cpp:410-416).  Leaves the camera where it was."""
    from volumerendering_amd import workloads as wl
    cam = app.camera()
    us = []
    for g in range(n):
        app.OnUpdate()
        us.append(app.uniforms())
        if turntable:
            cam.Rotate(TURN_PX, 0.0)
    cam.SetOrbit(*wl.CAMERA)
    app.OnUpdate()
    return us


def pmc_child(args):
    """`bench.py --pmc-child`: what the rocprofv3 counter passes run -- the same scene, a few synchronous frames (one
    launch at a time: counters of overlapping dispatches cannot be told apart) of the same turntable, no torch, no timing."""
    from volumerendering_amd import capi, host, workloads as wl
    n, W, H, vname = wl.WORKLOADS[args.workload]
    with host.Application(W, H, 0) as app:
        wl.build_scene(app, args.workload, args.tf, args.air, args.vol_n, quiet=True)
        ctx = app.context()
        if args.flavour:
            ctx.set_kernel_flavour(args.flavour)
        ctx.set_volume_layout(args.layout)
        ctx.set_arithmetic(1 if args.arith == "fused" else 0)
        variant = capi.VARIANT_NAMES.index(vname)
        if args.frames_per_launch > 1:
            # launches of several frames (different cameras), one launch at a time, into the frame buffers of helper contexts
            n = min(4, args.frames_per_launch)
            us = camera_sequence(app, 5 * n, not args.pmc_identical)
            others = [capi.Context(W, H, 0) for _ in range(n)]
            try:
                for k in range(5):
                    ctx.render_batch_async(variant, us[k * n:(k + 1) * n], [o.frame_device_ptr() for o in others])  # the context's own stream
                    ctx.counters()  # waits for the launch
            finally:
                for o in others:
                    o.close()
            return
        us = camera_sequence(app, 6, not args.pmc_identical)
        for u in us:
            ctx.set_uniforms(u)
            ctx.render(variant)


def live_pmc(args, passes=PMC_PASSES, timeout_s=150, frames_per_launch=0, identical=False):
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
        if identical:
            cmd += ["--pmc-identical"]
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
    _, _, n_a1, dt_a1 = run(s1, cores)  # all cores on the one-thread leg's pixel grid: the two rates of `speedup_over_one_thread`
    base = {"value": round(n_all / dt_all / 1e9, 6), "unit": "Gsamples/s", "cores": cores,
            "cores_note": f"threads used = CPUs in this process's affinity mask ({len(os.sched_getaffinity(0))}), cut to the cgroup CPU quota when "
                          "one is readable; speedup_over_one_thread says how many of them the box really grants",
            "kind": "port",
            "implementation": "oracle/vr_oracle.c (scalar C restatement of the WGSL, pthread over pixels)",
            "sample": f"{parity['of_frame']} of the {W}x{H} frame ({pxy.shape[0]} rays, {n_all} composited samples, {dt_all:.2f} s)",
            "speedup_over_one_thread": round((n_a1 / dt_a1) / (n_1 / dt_1), 2) if dt_1 > 0 and n_1 and dt_a1 > 0 else None,
            "speedup_note": f"both rates on one pixel grid (one pixel of every {s1} x {s1}): {cores} threads {n_a1 / dt_a1 / 1e9:.6f}, "
                            f"one thread {n_1 / dt_1 / 1e9:.6f} Gsamples/s",
            "one_thread": {"value": round(n_1 / dt_1 / 1e9, 6), "cores": 1,
                           "sample": f"one pixel of every {s1} x {s1} ({p1.shape[0]} rays, {n_1} composited samples, {dt_1:.2f} s)"}}
    return parity, base


# ------------------------------------------------------------------------------------------------ N > 1 without a launcher
def rank_environments(n_ranks, base_env=None, port=None):
    """The environment of each of the n_ranks child processes `bench.py --gpus N` starts when it was not launched by
    torch.distributed.run: one process per GPU, rank r on device r, rendezvous on 127.0.0.1 (the container hostname may not
    resolve).  Pure: tests/test_bench_launcher.py checks it on CPU."""
    base = dict(os.environ if base_env is None else base_env)
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    envs = []
    for r in range(n_ranks):
        e = dict(base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "VR_BENCH_SPAWNED": "1"})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
        envs.append(e)
    return envs


def launch_ranks(n_ranks, argv):
    """Starts the ranks as fresh child processes (this process has not touched a GPU and never will), relays rank 0's JSON
    line, and exits with the worst return code.  A rank that fails takes the others down (exact PIDs only)."""
    envs = rank_environments(n_ranks)
    cmd = [sys.executable, os.path.abspath(__file__), *argv]
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = b""
    rcs = [None] * n_ranks
    try:
        while any(rc is None for rc in rcs):
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    if r == 0:
                        try:
                            o, _ = pr.communicate(timeout=0.5)
                            out0 += o or b""
                            rcs[0] = pr.returncode
                        except subprocess.TimeoutExpired:
                            pass
                    else:
                        rcs[r] = pr.poll()
            bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
            if bad:
                for r, pr in enumerate(procs):
                    if rcs[r] is None:
                        pr.terminate()
                log(f"[bench] rank(s) {bad} failed (rc {[rcs[r] for r in bad]}): stopping the others")
                for r, pr in enumerate(procs):
                    if rcs[r] is None:
                        try:
                            pr.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            pr.kill()
                        rcs[r] = pr.returncode
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    sys.stdout.write(out0.decode(errors="replace"))
    sys.stdout.flush()
    return max(abs(rc or 0) for rc in rcs)


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
    ap.add_argument("--layout", type=int, default=0, choices=[0, 1, 2, 3],
                    help="vr_set_volume_layout: 0 bricked copy of voxels + density plane (default), 3 x-fastest voxels + plane (round 2's "
                         "default), 1 the reference's vec4 voxels only, 2 = 3 + lit gradients derived on the fly from the plane")
    ap.add_argument("--vol-n", type=int, default=0, help="experiment: smaller volume, same frame and stepping")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="launches in flight in the pipelined / batched legs")
    ap.add_argument("--exp-mode", type=int, default=0, help="experiment: fragmentMode 1-4 (ray set-up only)")
    ap.add_argument("--exp-steps", type=int, default=-1, help="experiment: override stepsCount")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="batched leg: frames marched by one launch (vr_render_batch_async / vr_mgpu_frames_async; 1..4, 0 = 4)")
    ap.add_argument("--identical-frames", action="store_true",
                    help="experiment: every frame with frame 0's camera (what rounds 1-2 measured) instead of the turntable")
    ap.add_argument("--turn-frames", type=int, default=628,
                    help="frames of the full-turn leg (one frame at a time, yaw + 0.01 rad per frame: 628 = 2 pi; 0 = skip)")
    ap.add_argument("--settle", type=int, default=24,
                    help="untimed LAUNCHES in front of every leg's warm-up (twice as many with launches in flight): the context measures its "
                         "candidate kernels on live frames (vr.h, flavour 0) and the leg then times the steady state")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-identical", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: start the ranks ourselves, before anything in this process touches a GPU
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    from volumerendering_amd import capi, host, workloads as wl

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # VR_BENCH_DEVICE / VR_BENCH_BACKEND=gloo exist only to rehearse the N > 1 bookkeeping on a one-GPU box (all ranks on
    # device 0, tiles gathered with gloo through host memory: RCCL refuses two ranks on one device); the real multi-GPU
    # run uses one GPU per rank and the C++ frame loop of libvr_mgpu.so (ncclGather over xGMI).
    device_index = int(os.environ.get("VR_BENCH_DEVICE", local_rank))
    backend = os.environ.get("VR_BENCH_BACKEND", "rccl")
    if device_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants device {device_index}, but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(device_index)
    # VR_BENCH_SELF_GATHER=1 (rehearsal): take the multi-rank code path -- tile render, RCCL gather, un-permute -- with
    # a world of one, so that the RCCL calls and their stream ordering can be exercised on a one-GPU box
    multi = world > 1 or bool(os.environ.get("VR_BENCH_SELF_GATHER"))
    launched_by = ("bench.py itself (one child process per rank, started before any GPU call)" if os.environ.get("VR_BENCH_SPAWNED") else
                   ("an external launcher (RANK / WORLD_SIZE in the environment)" if "WORLD_SIZE" in os.environ else
                    "nothing: a world of one through the multi-rank code path (VR_BENCH_SELF_GATHER rehearsal)"))
    share = int(os.environ.get("VR_MGPU_EXP_SHARE", "1")) if (multi and world == 1) else 1
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

    # ---- the turntable: uniforms of frame g, and (below) what each of them composites -----------------------------------
    fpl = max(1, min(4, args.frames_per_launch if args.frames_per_launch > 0 else 4))
    n_seq = args.warmup + max(args.steps, 20) + 2 * fpl
    us = camera_sequence(app, n_seq, turntable=not args.identical_frames)
    u0 = us[0]
    assert bytes(u0) == bytes(app.uniforms())  # frame 0 = the scene's own camera (the parity frame)

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
    d_present = torch.zeros((H, W), dtype=torch.int32, device="cuda")
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

    if multi and mg is None:
        fpl = 1  # (the gloo rehearsal renders frame by frame)
    batch_frames = [[torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(fpl)] for _ in range(max_flight)] \
        if (fpl > 1 and not multi) else []

    if mg is not None and fpl > 1:  # size the loop's buffer sets for fpl frames per launch outside every timed region
        with stdout_to_stderr():
            for _ in range(max_flight):  # (every buffer set: the final comparison reads all of their frames)
                mg.frames_async(variant, us[:fpl])
            mg.wait()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- composited samples / covered pixels / fetched samples of every camera of the sequence (untimed) -----------------
    counts = []  # per frame g: (composited, covered, fetched), summed over the ranks
    for g in range(n_seq):
        ctx.set_uniforms(us[g])
        if mg is not None:
            mg.frame_async(variant)
            (c_, v_, f_), _ = mg.reduce(0.0)
        elif multi:
            ctx.render_tiles_async(variant, rank, world, my_tiles[0].data_ptr(), streams[0])
            s_ = torch.tensor(list(ctx.counters()), dtype=torch.int64)
            dist.all_reduce(s_, op=dist.ReduceOp.SUM)
            c_, v_, f_ = (int(x) for x in s_.tolist())
        else:
            ctx.render_async(variant, frames[0].data_ptr(), streams[0])
            c_, v_, f_ = ctx.counters()
        counts.append((c_, v_, f_))
        if args.identical_frames:
            counts = counts * n_seq
            break
    ctx.set_uniforms(u0)
    sync_all()

    last_choice = {}
    full_choice = []  # [vr_kernel_choice, flavour that ran] behind the last batched launch that carried all its frames
    last_batch = {}  # (buffer set, frame of the launch) -> g of the camera the batched leg rendered into it last
    last_single = {}  # buffer set -> g, one frame per launch

    def run_frames(g, n_frames, nbuf, fpl=1, present=False):
        """Enqueues frames g .. g + n_frames - 1 of the turntable; returns the next g."""
        end = g + n_frames
        if not multi:
            if fpl > 1:  # one launch carries up to fpl frames, each with its own camera
                launch = 0
                while g < end:
                    k = min(fpl, end - g)
                    b = launch % nbuf
                    ctx.render_batch_async(variant, [us[(g + j) % n_seq] for j in range(k)], [t.data_ptr() for t in batch_frames[b][:k]], streams[b])
                    if k == fpl:  # (a last launch with fewer frames is another launch shape, with a measured choice of its own)
                        full_choice[:] = [ctx.kernel_choice(), ctx.last_kernel_flavour()]
                    for j in range(k):
                        last_batch[(b, j)] = (g + j) % n_seq
                    g += k
                    launch += 1
                return g
            k = 0
            while g < end:
                b = k % nbuf
                ctx.set_uniforms(us[g % n_seq])
                ctx.render_async(variant, frames[b].data_ptr(), streams[b])
                if present:
                    ctx.present_async(d_present.data_ptr(), frames[b].data_ptr(), streams[b])
                last_single[b] = g % n_seq
                g += 1
                k += 1
            return g
        if mg is not None:
            # the C++ loop: every rank renders its tiles, ncclGather, the root's output pass; two buffer sets, so up to two
            # launches are in flight -- nbuf = 1: one frame at a time ON THE DEVICE (vr_mgpu_set_frames_in_flight(1), set by
            # timed_leg: every stage of a launch on one stream), the host enqueues ahead as the one-GPU leg does
            if fpl > 1 and nbuf > 1:
                while g < end:
                    k = min(fpl, end - g)
                    b = mg.frames_async(variant, [us[(g + j) % n_seq] for j in range(k)])
                    for j in range(k):
                        last_batch[(b, j)] = (g + j) % n_seq
                    g += k
                mg.wait()
                return g
            while g < end:
                ctx.set_uniforms(us[g % n_seq])
                b = mg.frame_async(variant)
                last_single[b] = g % n_seq
                g += 1
            mg.wait()
            return g
        while g < end:  # gloo rehearsal: synchronous, through host memory
            b = g & 1
            ctx.set_uniforms(us[g % n_seq])
            ctx.render_tiles_async(variant, rank, world, my_tiles[b].data_ptr(), streams[0])
            host_list = [torch.empty(my_tiles[b].numel()) for _ in range(world)] if rank == 0 else None
            torch.cuda.synchronize()  # (the render is on the context's stream, the copy below on torch's: nothing else orders them)
            dist.gather(my_tiles[b].cpu(), host_list, dst=0)
            if rank == 0:
                gathered[b].copy_(torch.stack(host_list))
                torch.cuda.synchronize()  # (as above: torch's copy, then the context's stream)
                ctx.unpack_tiles_async(gathered[b].data_ptr(), world, frames[b].data_ptr(), streams[0])
                last_single[b] = g % n_seq
            g += 1
        return g

    def timed_leg(nbuf, n_warm, n_steps, min_events=0, fpl=1, present=False):
        """W warm-up + exactly K timed frames of the turntable (frames 0 .. W+K-1), barrier + synchronise on both sides; the MAX
        over ranks of the wall time.  Returns (seconds, kernel durations of the timed launches [+ extra frames up to
        min_events], composited samples of the K timed frames, fetched samples of them)."""
        ctx.hint_frames_in_flight(nbuf)  # what this leg's caller does: steers the default kernel choice (vr.h)
        if mg is not None:
            mg.set_frames_in_flight(1 if nbuf == 1 else 0)
        if args.settle > 0:  # (untimed, and in front of the W warm-up frames: the leg itself is W + K frames from camera 0)
            run_frames(0, args.settle * fpl * (1 if nbuf == 1 else 2), nbuf, fpl, present)  # (counted in LAUNCHES: 2 x with launches in flight)
            sync_all()
        g = run_frames(0, n_warm, nbuf, fpl, present)
        sync_all()
        ctx.reset_kernel_times()
        t0 = time.perf_counter()
        g1 = run_frames(g, n_steps, nbuf, fpl, present)
        sync_all()
        dt = time.perf_counter() - t0
        comp = sum(counts[x % n_seq][0] for x in range(g, g1))
        fetched = sum(counts[x % n_seq][2] for x in range(g, g1))
        if n_steps < min_events:  # the median below wants >= 20 event-timed frames (outside the K-step region)
            run_frames(g1, min_events - n_steps, nbuf, fpl, present)
            sync_all()
        kt = ctx.kernel_times(min(max(n_steps, min_events), 256))
        cands, cms, chosen = ctx.kernel_choice()  # what the default's measured choice knows of this leg's launch shape
        ran_last = ctx.last_kernel_flavour()
        if fpl > 1 and not multi and full_choice:
            (cands, cms, chosen), ran_last = full_choice
        last_choice.clear()
        last_choice.update({"candidates": cands, "ms_per_launch": [round(x, 4) for x in cms], "kept": cands[chosen] if chosen >= 0 else None,
                            "ran_last": ran_last})
        if dist is not None:  # MAX over ranks
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kt, comp, fetched, dict(last_choice)

    def leg(res, nbuf, steps):
        dt, kt, comp, fetched = res[:4]
        ms = dt / steps * 1e3
        return {"launches_in_flight": nbuf, "kernel_choice": res[4] if len(res) > 4 else None, "ms_per_step": round(ms, 4), "fps": round(1e3 / ms, 2),
                "value": round(comp / dt / 1e9, 3), "fetched_gsamples_per_s": round(fetched / dt / 1e9, 3),
                "composited_samples_timed": comp,
                "kernel_ms_median": round(float(np.median(kt)), 4) if len(kt) else None,
                "kernel_ms_mean": round(float(np.mean(kt)), 4) if len(kt) else None,
                "kernel_ms_p10_p90": [round(float(np.percentile(kt, 10)), 4), round(float(np.percentile(kt, 90)), 4)] if len(kt) else None,
                "kernel_events": int(len(kt))}

    # ---- the legs -----------------------------------------------------------------------------------------------------
    res_serial = timed_leg(1, args.warmup, args.steps, min_events=20)
    # The legs' launches are timed from their own per-workgroup records (vr_kernel_times' default: two HIP events around
    # every launch cost the frame's stream 11 us, which the one-frame-at-a-time leg would pay).  The roofline's denominator is
    # measured the contract's way: HIP events on the launch stream around each of 20 more one-at-a-time frames, outside the
    # timed regions.
    ctx.set_kernel_timing(True)
    _, kt_events, _, _, _ = timed_leg(1, 3, 20)
    ctx.set_kernel_timing(False)
    res_present = timed_leg(1, args.warmup, args.steps, present=True) if not multi else None
    nbuf_over = max_flight if mg is not None else min(args.in_flight, max_flight)
    if nbuf_over == 1 and args.frames_per_launch <= 0:
        fpl = 1  # (--in-flight 1: one frame at a time in this leg too, unless --frames-per-launch asks for batched launches)
    # two launches in flight, one frame each ...
    res_pipe = timed_leg(nbuf_over, args.warmup, args.steps) if (fpl > 1 and nbuf_over > 1) else None
    # ... and the throughput leg proper: two launches in flight, fpl frames per launch, fpl different cameras
    res_over = timed_leg(nbuf_over, args.warmup, args.steps, fpl=fpl)

    # ---- SURVEY 8d's t_frame, literally: >= 20 SYNCHRONOUS vr_render calls (the host waits for every frame), each timed by the
    # HIP events vr_render records around the frame on its stream; 3 warm-ups; the median.  Beside it: the stream-ordered
    # `serial` leg above (the host enqueues ahead and never waits inside the timed region), whose wall / K is `value`.
    sync_8d = None
    if not multi:
        ctx.hint_frames_in_flight(1)
        n_sync = max(20, min(args.steps, 100))
        tms, kms = [], []
        for g in range(3 + n_sync):
            ctx.set_uniforms(us[g % n_seq])
            ctx.render(variant)
            k_ms, t_ms = ctx.last_timing()
            if g >= 3:
                tms.append(t_ms)
                kms.append(k_ms)
        med = float(np.median(tms))
        comp = [counts[g % n_seq][0] for g in range(3, 3 + n_sync)]
        sync_8d = {"calls": n_sync, "warmup": 3, "t_frame_ms_median": round(med, 4), "t_frame_ms_p10_p90": [round(float(np.percentile(tms, 10)), 4), round(float(np.percentile(tms, 90)), 4)],
                   "kernel_ms_median": round(float(np.median(kms)), 4), "fps": round(1e3 / med, 2),
                   "value": round(float(np.median(comp)) / med / 1e6, 3),
                   "note": "SURVEY 8d: median of hipEvent-timed synchronous vr_render calls (events on the frame's stream around ray set-up + march + "
                           "record keeping), turntable cameras 3 .. 3 + calls; value = median composited samples / median t_frame"}

    # ---- the full turn: one frame at a time, every camera of a 2 pi turntable (yaw + 0.01 rad per frame) -----------------------
    full_turn = None
    if not multi and args.turn_frames > 0 and not args.identical_frames:
        ctx.hint_frames_in_flight(1)
        us_t = camera_sequence(app, args.turn_frames, turntable=True)
        comp_t = []
        for u in us_t:  # untimed: what every camera composites
            ctx.set_uniforms(u)
            ctx.render_async(variant, frames[0].data_ptr(), streams[0])
            comp_t.append(ctx.counters()[0])
        torch.cuda.synchronize()
        ctx.reset_kernel_times()
        t0 = time.perf_counter()
        for u in us_t:
            ctx.set_uniforms(u)
            ctx.render_async(variant, frames[0].data_ptr(), streams[0])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = ctx.kernel_times(256)
        full_turn = {"frames": len(us_t), "yaw_step_rad": 0.01, "ms_per_frame_mean": round(dt / len(us_t) * 1e3, 4),
                     "value": round(sum(comp_t) / dt / 1e9, 3), "fps": round(len(us_t) / dt, 2),
                     "composited_samples_per_frame_min_max": [min(comp_t), max(comp_t)],
                     "kernel_ms_last_256_p10_p50_p90": [round(float(np.percentile(kt, q)), 4) for q in (10, 50, 90)] if len(kt) else None,
                     "kernel_flavour_resolved_last": ctx.last_kernel_flavour(),
                     "note": "one frame at a time on one stream (as `serial`), the WHOLE turn: composited samples of all frames / wall time"}
        ctx.set_uniforms(u0)

    # every frame the batched launches left behind must equal a single-frame render with the same uniforms, bit for bit
    batched_equal = None
    if batch_frames and last_batch:
        batched_equal = True
        check = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        for (b, j), g in sorted(last_batch.items()):
            ctx.set_uniforms(us[g])
            ctx.render_async(variant, check.data_ptr(), streams[0])
            torch.cuda.synchronize()
            if not bool(torch.equal(batch_frames[b][j].view(torch.int32), check.view(torch.int32))):
                batched_equal = False
        del check
        if not batched_equal:
            raise SystemExit("bench.py: a frame of a batched launch differs from the single-frame render with the same uniforms")

    # multi-rank: what the root assembled -- every frame of the last batched launch into each buffer set (the batched leg ran
    # last), or the last frames of the gloo rehearsal -- with the camera index of each, for the comparison at the end
    gpu_frames, frames_g = [], []
    if rank == 0 and multi:
        if mg is not None:
            for (b, j), g in sorted((last_batch if fpl > 1 else {(b_, 0): g_ for b_, g_ in last_single.items()}).items()):
                gpu_frames.append(mg.download_batch_frame(b, j, W, H))
                frames_g.append(g)
        else:
            for b, g in sorted(last_single.items()):
                gpu_frames.append(frames[b].cpu().numpy())
                frames_g.append(g)

    serial, over = leg(res_serial, 1, args.steps), leg(res_over, nbuf_over, args.steps)
    serial["kernel_ms_source"] = "first workgroup start .. last workgroup end of each timed launch (100 MHz device clock, from the launch's records)"
    serial["kernel_ms_hip_events_median"] = round(float(np.median(kt_events)), 4) if len(kt_events) else None
    serial["kernel_ms_hip_events_note"] = "HIP events on the launch stream around 20 further one-at-a-time launches; the roofline's denominator"
    with_present = None
    if res_present is not None:
        with_present = leg(res_present, 1, args.steps)
        with_present["note"] = ("serial leg + vr_present_async behind every frame: output merge over the white background into BGRA8Unorm "
                                "(App/src/renderer/PipelineBuilder.cpp:142-154), 16 B read + 4 B written per pixel")
    pipelined = None
    if res_pipe is not None:
        pipelined = leg(res_pipe, nbuf_over, args.steps)
        pipelined["frames_per_launch"] = 1
    over["frames_per_launch"] = fpl
    over["frames_in_flight"] = nbuf_over * fpl
    over["frames_of_batched_launches_equal_single_renders"] = batched_equal
    if fpl > 1:  # one record span per LAUNCH: the durations are those of launches that carry fpl frames each
        over["kernel_ms_note"] = f"kernel_ms_* are per launch of {fpl} frames"

    # frame 0 again (every rank: its counters are this rank's share of the parity frame)
    ctx.set_uniforms(u0)
    ctx.hint_frames_in_flight(1)
    if mg is not None:
        mg.frame_async(variant)
        mg.wait()
    elif multi:
        ctx.render_tiles_async(variant, rank, world, my_tiles[0].data_ptr(), streams[0])
    else:
        ctx.render_async(variant, frames[0].data_ptr(), streams[0])
    torch.cuda.synchronize()
    my_samples, my_covered, my_fetched = ctx.counters()
    total_samples, covered, total_fetched = counts[0]
    ran = ctx.last_kernel_flavour()
    layout_flags = ctx.volume_layout(2 if vname == "VOLUME_MASK" else 0)

    # the other arithmetic mode, same scene, both legs (20 frames each; outside the K-step regions above)
    arith_ab = None
    if not multi:
        other = "separate" if args.arith == "fused" else "fused"
        ctx.set_arithmetic(1 if other == "fused" else 0)
        r_s2 = timed_leg(1, 5, 20)
        r_o2 = timed_leg(nbuf_over, 5, 20, fpl=fpl)
        arith_ab = {"arithmetic": other,
                    "serial": {"ms_per_step": round(r_s2[0] / 20 * 1e3, 4), "kernel_ms_median": round(float(np.median(r_s2[1])), 4),
                               "value": round(r_s2[2] / r_s2[0] / 1e9, 3)},
                    "overlapped": {"frames_in_flight": nbuf_over * fpl, "ms_per_step": round(r_o2[0] / 20 * 1e3, 4),
                                   "value": round(r_o2[2] / r_o2[0] / 1e9, 3)},
                    "note": "vr_set_arithmetic: per-sample a*b+c with one rounding (fused) instead of two; bit-exact against the "
                            "oracle's mode of the same name (tests/), ray placement identical in both modes: the sample counts of "
                            "the headline mode's cameras are used for both"}
        ctx.set_arithmetic(1 if args.arith == "fused" else 0)
        ctx.set_uniforms(u0)
        ctx.render_async(variant, frames[0].data_ptr(), streams[0])  # frames[0] = the headline mode's frame 0 again
        torch.cuda.synchronize()
    kernel_ms = serial["kernel_ms_hip_events_median"] or serial["kernel_ms_median"]
    bs = wl.BYTES_PER_SAMPLE[vname]
    part_world = max(world, share)
    owned_px = W * H if not multi else ctx.tile_count(rank, part_world) * capi.TILE * capi.TILE
    alg_fetched = my_fetched * bs + 16 * owned_px   # what the kernel's loads ask for (mostly served by L1 / L2)
    alg_composited = my_samples * bs + 16 * owned_px  # SURVEY 8d's figure: every composited sample priced as a fetch
    vol_bytes = sum(int(np.prod(v.GetSize())) * 16 for v in vols)

    gpu_frame = frames[0].cpu().numpy() if (rank == 0 and not multi) else None

    # ---- roofline: measured fabric bytes (rocprofv3 PMC passes on the same turntable, launched from here) ---------------
    pmc, pmc_note, pmc_same = {}, "skipped", {}
    # (the counter passes profile a handful of launches of a fresh process: they are told which kernel the serial leg's measured
    # choice kept, so that what is profiled is what was timed and not the first candidates of a new trial)
    pargs = argparse.Namespace(**vars(args))
    pargs.flavour = args.flavour or ran
    if rank == 0 and world == 1 and not multi and not args.no_live_pmc:
        t0 = time.time()
        pmc, pmc_note = live_pmc(pargs, PMC_PASSES + (PMC_EXTRA if args.pmc_extra else []), identical=args.identical_frames)
        if not args.identical_frames:  # the same frames four times over: what the caches make of identical uniforms
            pmc_same, _ = live_pmc(pargs, [["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"],
                                          ["TA_BUSY_avr", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]], identical=True)
        log(f"[bench] rocprofv3 counter passes: {time.time() - t0:.1f}s {pmc_note}")
    traffic = None
    traffic_source = None
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        traffic_source = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this script on the same scene and turntable, one launch "
                          "at a time; (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 FETCH_SIZE correction; counts "
                          "Infinity-Cache hits too: an upper estimate of HBM bytes)")
    else:
        # (no committed file stands in for a measurement: a rank's share, a rehearsal or a skipped pass has no traffic figure)
        traffic_source = f"not measured ({'multi-rank / rank-share run' if multi else pmc_note})"
    achieved = traffic / (kernel_ms * 1e-3) / 1e9 if (traffic and kernel_ms) else None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic, "traffic_source": traffic_source,
        "kernel": KERNEL_OF_FLAVOUR.get(ran, "march_kernel"), "kernel_ms": kernel_ms,
        "kernel_ms_is": "median HIP-event duration of 20 one-at-a-time launches, events recorded on the launch stream",
        "frac_overlapped": round(traffic / (over["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
        "bytes_per_sample": bs,
        "effective_gather": {
            "note": "algorithmic bytes the loads ask for (frame 0); L1 / L2 serve most of them, so this is NOT a roofline fraction",
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
            "busy_frac_at_saturation": VALU_SATURATION,
            "note": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).  NOT a fraction of a limit of 1: the counter "
                    "charges every instruction 4 cycles, a CDNA4 SIMD issues a plain one every ~1.7-1.85 cycles and a packed / "
                    "converting one every ~2.7 at 4-5 wavefronts per SIMD (tools/ubench/valu_issue.hip under the same counters: a "
                    "kernel of nothing but v_fma_f32 reads 1.5-1.7, of v_pk_fma_f32 about 1.2); busy_frac_at_saturation is that "
                    "calibration for this kernel's instruction mix"}
        # effective shader clock: cycles of the profiled launches over THEIR duration (kernel trace of the same pass)
        prof_ms = pmc.get("_profiled_march_ms")
        clock_hz = cyc / (prof_ms * 1e-3) if prof_ms else None
        roofline["valu"]["profiled_kernel_ms"] = round(prof_ms, 4) if prof_ms else None
        over_cyc = over["ms_per_step"] * 1e-3 * clock_hz if clock_hz else None
        roofline["valu"]["busy_frac_overlapped"] = round(pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * over_cyc), 4) if over_cyc else None
        roofline["valu"]["clock_ghz"] = round(clock_hz / 1e9, 3) if clock_hz else None
        # the vector-memory (texture addresser / L1) data path: every lane's bytes are returned at 64 B / clk / CU
        tf_bytes = {"BASIC": 40, "LIGHT": 40, "LIGHT_INSHADER": 40, "TF_CALIB": 40}.get(vname, 80)
        if ran in (12, 13, 16, 17):
            tf_bytes -= 40  # TF slot 0 comes from LDS
        l1_bytes = my_fetched * (bs + tf_bytes) + my_samples // 1  # + one distance-field byte per executed step (lower bound)
        l1_peak = 64.0 * 256 * clock_hz / 1e9 if clock_hz else None
        roofline["l1"] = {
            "bytes_per_launch": l1_bytes, "peak_gbs": round(l1_peak, 1) if l1_peak else None,
            "achieved_gbs": round(l1_bytes / (kernel_ms * 1e-3) / 1e9, 1) if kernel_ms else None,
            "frac": round(l1_bytes / (kernel_ms * 1e-3) / 1e9 / l1_peak, 4) if (l1_peak and kernel_ms) else None,
            "frac_overlapped": round(l1_bytes / (over["ms_per_step"] * 1e-3) / 1e9 / l1_peak, 4) if l1_peak else None,
            "ta_busy_frac": round(pmc["TA_BUSY_avr"] / cyc, 4) if "TA_BUSY_avr" in pmc else None,
            "ta_busy_frac_overlapped": round(pmc["TA_BUSY_avr"] / over_cyc, 4) if ("TA_BUSY_avr" in pmc and over_cyc) else None,
            "note": "bytes the lanes receive from L1 (corner voxels + transfer-function texels of every fetched sample of frame 0) against "
                    "the 64 B/clk/CU return path of the vector memory pipeline at the measured clock; TA_BUSY_avr = busy cycles of the "
                    "texture addressers per launch"}
        # the same two units measured on launches of `fpl` frames (one launch at a time: counters of overlapping dispatches
        # cannot be told apart), instead of scaled from the one-frame launches
        if fpl > 1 and not args.no_live_pmc:
            bargs = argparse.Namespace(**vars(args))
            bargs.flavour = args.flavour or (over.get("kernel_choice") or {}).get("kept") or 0
            bp, bnote = live_pmc(bargs, [["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"], ["GRBM_GUI_ACTIVE", "TA_BUSY_avr"]], frames_per_launch=fpl,
                                 identical=args.identical_frames)
            if "SQ_ACTIVE_INST_VALU" in bp and "GRBM_GUI_ACTIVE" in bp:
                bcyc = bp["GRBM_GUI_ACTIVE"] / 8.0
                roofline["batched_launch_measured"] = {
                    "frames_per_launch": fpl, "launches_in_flight": 1,
                    "valu_busy_frac": round(bp["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMDS * bcyc), 4),
                    "ta_busy_frac": round(bp["TA_BUSY_avr"] / bcyc, 4) if "TA_BUSY_avr" in bp else None,
                    "valu_insts_per_frame": round(bp.get("SQ_INSTS_VALU", 0.0) / fpl),
                    "profiled_ms_per_frame": round(bp["_profiled_march_ms"] / fpl, 4) if "_profiled_march_ms" in bp else None,
                    "note": "rocprofv3 --pmc passes on launches that carry several frames (different cameras), one launch at a time; "
                            "the batched leg keeps two such launches in flight"}
            elif bnote:
                log(f"[bench] batched counter passes: {bnote}")
        fr = {"hbm": roofline["frac_overlapped"] or 0,
              "valu-issue": (roofline["valu"]["busy_frac_overlapped"] or 0) / VALU_SATURATION,
              "l1-return-path": roofline["l1"]["ta_busy_frac_overlapped"] or roofline["l1"]["frac_overlapped"] or 0}
        roofline["limiter"] = max(fr, key=fr.get) + " (batched leg); longest-ray tail on top of it in the serial leg"
        roofline["limiter_fractions_overlapped"] = fr
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
        roofline["l2_hit_rate"] = round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4)
    if pmc_same:
        def side(d):
            cyc_ = d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            return {"traffic_bytes": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 if ("FETCH_SIZE" in d and "WRITE_SIZE" in d) else None,
                    "l2_hit_rate": round(d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in d else None,
                    "ta_busy_frac": round(d["TA_BUSY_avr"] / cyc_, 4) if ("TA_BUSY_avr" in d and cyc_) else None,
                    "l1_accesses": d.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "profiled_kernel_ms": d.get("_profiled_march_ms")}
        roofline["turntable_vs_identical"] = {
            "turntable": side(pmc), "identical": side(pmc_same),
            "note": "same counter passes on the same scene, one launch at a time: frames of the turntable (yaw + 0.01 rad per frame) "
                    "against the same frame rendered again and again (what rounds 1-2 profiled)"}
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

    part = "single GPU" if not multi else f"64x64 image tiles interleaved over {part_world} GPUs + " + \
        ((mg.backend() + " (C++ frame loop, libvr_mgpu.so)") if mg is not None else "gloo gather through host memory (rehearsal)")
    out = {
        "metric": "Gsamples/s", "value": serial["value"], "unit": "Gsamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": serial["ms_per_step"], "fps": serial["fps"],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: ct-phantom-{n} RGBA32F voxels, {W}x{H}, {vname} shader, TF {args.tf}, air {args.air}, "
                        f"step 1/{round(1 / step_size)} x {steps_count}, camera d=1.2 yaw=.6 pitch=.35 + turntable",
            "camera": ("every frame with frame 0's camera (--identical-frames)" if args.identical_frames else
                       "turntable: frame g = the BASELINE camera after g x Camera::Rotate(2 px, 0), yaw + 0.01 rad per frame "
                       "(App/src/Application.cpp:410-416, App/src/Camera.cpp:146-152); every leg renders frames 0 .. W+K-1; frames of "
                       "one batched launch have different uniforms"),
            "partition": part,
            "value_is": "the `serial` leg: ONE FRAME AT A TIME, stream-ordered on one stream (the host enqueues ahead and never waits inside "
                        "the timed region), composited samples of the K timed turntable frames / wall time between barrier + synchronise.  "
                        "SURVEY 8d's t_frame to the letter (median of hipEvent-timed synchronous vr_render calls) is `sync_8d`; every camera "
                        f"of a 2 pi turn is `full_turn`; the pipelined ({nbuf_over} launches x 1 frame) and batched ({nbuf_over} launches x "
                        f"{fpl} frames) legs are extras, in `pipelined_one_frame_per_launch` / `overlapped`",
            "composited_samples_frame0": total_samples, "fetched_samples_frame0": total_fetched, "covered_pixels_frame0": covered,
            "composited_samples_per_frame_min_max": [min(c[0] for c in counts), max(c[0] for c in counts)],
            "kernel_flavour": args.flavour, "kernel_flavour_resolved": ran,
            "volume_layout": {0: "bricked copy (4 x 4 x 4 voxel bricks, brick-linear) of the vec4 voxels + density plane",
                              1: "reference vec4 voxels only, x fastest",
                              2: "x-fastest density plane, corner gradients derived on the fly" if layout_flags & 4 else "x-fastest vec4 voxels + density plane",
                              3: "x-fastest vec4 voxels + density plane (round 2's default)"}[args.layout],
        },
        "serial": serial, "overlapped": over, "roofline": roofline,
    }
    out["config"]["arithmetic"] = args.arith
    if sync_8d:
        out["sync_8d"] = sync_8d
    if full_turn:
        out["full_turn"] = full_turn
    if with_present:
        out["serial_with_present"] = with_present
    if pipelined:
        out["pipelined_one_frame_per_launch"] = pipelined
    if arith_ab:
        out["arith_ab"] = arith_ab
    if args.pmc_extra:
        out["pmc"] = {"per": "march-kernel launch, mean of the profiled launches (one at a time), first launch dropped",
                      **{k: v for k, v in pmc.items() if not k.startswith("_")}}
    if multi:
        # what the run really was: ranks RCCL saw, the device of every rank, the stage timeline of rank 0
        devs = [None] * world
        if dist is not None:
            dist.all_gather_object(devs, {"rank": rank, "device": device_index,
                                          "name": torch.cuda.get_device_name(device_index), "pid": os.getpid()})
        out["config"]["ranks"] = devs
        out["config"]["launched_by"] = launched_by
        if mg is not None:
            out["config"]["rccl_nranks"] = mg.comm_count()
            # stage timeline of this rank (rank 0 prints its own): 20 one-at-a-time frames and 20 batched launches, events on
            out["rank0_stage_timeline"] = stage_timeline(mg, ctx, variant, us, fpl, dist)

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
        out["regimes_note"] = ("one frame at a time, frame 0's camera, median HIP-event kernel ms of 20 frames; noisy air = raw 0..80 outside "
                               "the body; prefix = preset-style opacity table with a real zero prefix (workloads.py)")
        scene(args.air, args.tf)  # the CPU legs below want the headline scene again
        ctx.set_kernel_timing(False)

    if rank == 0 and world == 1 and not multi and not args.no_cpu_baseline:
        parity, base = oracle_legs(app, variant, vols, W, H, gpu_frame, total_samples, fused=(args.arith == "fused"))
        parity["arithmetic"] = args.arith
        parity["frame"] = "frame 0 of the turntable (the BASELINE camera)"
        out["parity"] = parity
        out["cpu_baseline"] = base
    if rank == 0 and multi and share == 1:
        # the gathered frames must equal single-rank renders of the same cameras, bit for bit (cheap: a few more frames)
        ok = True
        for fr_, g in zip(gpu_frames, frames_g):
            ctx.set_uniforms(us[g])
            ctx.render_async(variant, 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ref, _, _ = ctx.download()
            ok = ok and bool(np.array_equal(ref.view(np.uint32), fr_.view(np.uint32)))
        if mg is not None:  # and one frame through the one-frame form of the loop
            ctx.set_uniforms(u0)
            b = mg.frame_async(variant)
            got = mg.download(b, W, H)
            ctx.render_async(variant, 0, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ref, _, _ = ctx.download()
            ok = ok and bool(np.array_equal(ref.view(np.uint32), got.view(np.uint32)))
        out["config"]["frame_equals_single_rank_render"] = ok
        out["config"]["frames_compared"] = len(gpu_frames) + (1 if mg is not None else 0)
    elif mg is not None and share == 1:
        ctx.set_uniforms(u0)
        mg.frame_async(variant)  # (collective: the root's comparison frame above)
        mg.wait()
    if mg is not None:
        mg.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    app.close()


def stage_timeline(mg, ctx, variant, us, fpl, dist):
    """Rank 0's three stages (march / gather / root output) per launch, from HIP events of the C++ loop (vr_mgpu_stage_times):
    medians over 20 one-at-a-time frames and over 20 launches of `fpl` frames, for the un-permuted float frame and for the
    BGRA8 frame presented straight from the gathered tiles.  Collective: every rank runs the same launches."""
    from volumerendering_amd import mgpu
    res = {}
    mg.set_stage_timing(True)
    for name, output in (("float_frame_unpermuted", mgpu.OUT_FRAME), ("bgra8_presented_from_tiles", mgpu.OUT_PRESENT)):
        mg.set_output(output)
        rows = {"one_frame_per_launch": [], f"{fpl}_frames_per_launch": []}
        for k in range(23):
            ctx.set_uniforms(us[k % len(us)])
            b = mg.frame_async(variant)
            mg.wait()
            if k >= 3:
                rows["one_frame_per_launch"].append(mg.stage_times(0, b))
        if fpl > 1:
            for k in range(23):
                b = mg.frames_async(variant, [us[(k * fpl + j) % len(us)] for j in range(fpl)])
                mg.wait()
                if k >= 3:
                    rows[f"{fpl}_frames_per_launch"].append(mg.stage_times(0, b))
        res[name] = {key: dict(zip(("march_ms", "gather_ms", "output_ms", "total_ms"),
                                   (round(float(np.median([r[i] for r in v])), 4) for i in range(4))))
                     for key, v in rows.items() if v}
    mg.set_output(mgpu.OUT_FRAME)
    mg.set_stage_timing(False)
    res["note"] = ("this rank's HIP events around its own stages, one launch at a time (median of 20): march = its tiles rendered, gather = "
                   "ncclGather of every rank's segments complete on it, output = the root's pass over the gathered segments")
    return res


if __name__ == "__main__":
    main()
