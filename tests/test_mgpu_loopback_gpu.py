"""The multi-rank logic of the C++ frame loop (csrc/mgpu/vr_mgpu.cpp) with a world of 2..8 ranks ON ONE GPU.

RCCL refuses two ranks on one device, so the shipped libvr_mgpu.so can only be run here with a world of one
(tests/test_mgpu_gpu.py: that covers the RCCL calls themselves).  What a world of one cannot reach -- segment offsets in
the gather buffer, ranks that own fewer tiles than rank 0 (or none), the interleaved un-permute over several segments, the
buffer sets of pipelined frames with several ranks, the counter reduction -- is exercised here by linking THE SAME source
file against an in-process loopback communicator (tests/loopback_comm/loopback_comm.cpp: gather = stream-ordered
device-to-device copies, same group semantics) instead of librccl.  Every rank's context lives on device 0.  The assembled
frame must equal vr_render's bit for bit.  Transport over xGMI with N > 1 stays unmeasured on hardware here (the driver's
SCALE run is the only multi-GPU run)."""
import os
import subprocess

import numpy as np
import pytest

import host_ref as hr
import vrtest as vt
from volumerendering_amd import build, capi, mgpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_build")
PKG = os.path.join(ROOT, "volumerendering_amd")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_loopback() -> str:
    os.makedirs(OUT, exist_ok=True)
    build.build_hip()
    comm = os.path.join(OUT, "libloopback_comm.so")
    loop = os.path.join(OUT, "libvr_mgpu_loopback.so")
    common = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
              "-I", os.path.join(ROCM, "include")]
    rpaths = ["-Wl,-rpath," + OUT, "-Wl,-rpath," + PKG, "-Wl,-rpath," + os.path.join(ROCM, "lib")]
    subprocess.run(common + ["-o", comm, os.path.join(HERE, "loopback_comm", "loopback_comm.cpp"), "-L", os.path.join(ROCM, "lib"),
                             "-lamdhip64"] + rpaths, check=True)
    subprocess.run(common + ["-o", loop, os.path.join(PKG, "csrc", "mgpu", "vr_mgpu.cpp"), "-L", OUT, "-lloopback_comm", "-L", PKG,
                             "-lvr_hip", "-L", os.path.join(ROCM, "lib"), "-lamdhip64"] + rpaths, check=True)
    return loop


def test_loopback_build_never_touches_rccl():
    """(CPU) the test double builds, exports the whole vr_mgpu ABI, and is not linked against librccl; the SHIPPED library is."""
    loop = build_loopback()
    needed = subprocess.run(["readelf", "-d", loop], capture_output=True, text=True, check=True).stdout
    assert "libloopback_comm.so" in needed and "librccl" not in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", loop], capture_output=True, text=True, check=True).stdout
    for s in mgpu.ABI_SYMBOLS:
        assert f" T {s}" in syms, s
    shipped = subprocess.run(["readelf", "-d", build.build_mgpu()], capture_output=True, text=True, check=True).stdout
    assert "librccl" in shipped and "loopback" not in shipped


@pytest.fixture(scope="module")
def lib():
    return mgpu.bind(build_loopback())


def scene(ctx, W, H, variant, yaw=0.6):
    vols, tfs = vt.scene(variant, n=24)
    step, count = hr.stepping_params(24, 24, 24)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=yaw)
    for i, v in enumerate(vols):
        ctx.volume_upload(i, v)
    for i, t in enumerate(tfs):
        ctx.tf_upload(i, t[0], t[1])
    ctx.set_uniforms(vt.to_capi_uniforms(u))
    return u


def reference(W, H, variant, yaw=0.6):
    with capi.Context(W, H, 0) as ctx:
        scene(ctx, W, H, variant, yaw)
        ctx.render(variant)
        frame, _, n = ctx.download()
        return frame, n, ctx.covered_pixels()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 5, 8])
@pytest.mark.parametrize("variant", [capi.LIGHT, capi.VOLUME_MASK])
def test_n_ranks_assemble_the_single_gpu_frame(lib, world, variant):
    W, H = 200, 150   # 4 x 3 ragged tiles: with 5 and 8 ranks the ranks own different numbers of tiles
    ref, n_ref, cov = reference(W, H, variant)
    with mgpu.MultiGpu.local(W, H, [0] * world, _lib=lib) as m:
        assert m.world() == world and m.local_ranks() == world and "RCCL 0.0.0" in m.backend()
        for r in range(world):
            scene(m.context(r), W, H, variant)
        slots = [m.frame_async(variant) for _ in range(5)]
        assert slots == [0, 1, 0, 1, 0]
        m.wait()
        for which in (0, 1):
            assert np.array_equal(vt.bits(m.download(which, W, H)), vt.bits(ref)), which
        (comp, covered, fetched), mx = m.reduce(1.25)
        assert comp == n_ref and covered == cov and 0 < fetched <= comp and mx == 1.25
        # every rank rendered only its own tiles: the ranks' composited samples add up, none of them has them all
        per_rank = [m.context(r).counters()[0] for r in range(world)]
        assert sum(per_rank) == n_ref and max(per_rank) < n_ref


@pytest.mark.gpu
def test_more_ranks_than_tiles(lib):
    W, H = 64, 64   # one tile: ranks 1 and 2 own nothing and still take part in every collective
    ref, n_ref, cov = reference(W, H, capi.BASIC)
    with mgpu.MultiGpu.local(W, H, [0, 0, 0], _lib=lib) as m:
        for r in range(3):
            scene(m.context(r), W, H, capi.BASIC)
        for _ in range(3):
            m.frame_async(capi.BASIC)
        assert np.array_equal(vt.bits(m.download(0, W, H)), vt.bits(ref))
        (comp, covered, _), _ = m.reduce()
        assert comp == n_ref and covered == cov
        assert [m.context(r).counters()[0] for r in range(3)] == [n_ref, 0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("n_slots", [1, 2, 4])
def test_pipelined_frames_land_in_their_buffer_sets(lib, n_slots, monkeypatch):
    """A different camera per frame, nothing waited for in between: frame k is found in buffer set k mod slots."""
    monkeypatch.setenv("VR_MGPU_SLOTS", str(n_slots))
    W, H, world = 192, 128, 4
    yaws = [0.1, 0.6, 1.1, 1.6, 2.1, 2.6]
    refs = [reference(W, H, capi.LIGHT, y)[0] for y in yaws]
    with mgpu.MultiGpu.local(W, H, [0] * world, _lib=lib) as m:
        us = None
        for r in range(world):
            us = scene(m.context(r), W, H, capi.LIGHT)
        landed = {}
        for k, y in enumerate(yaws):
            step, count = hr.stepping_params(24, 24, 24)
            u = vt.to_capi_uniforms(hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=y))
            for r in range(world):
                m.context(r).set_uniforms(u)
            landed[m.frame_async(capi.LIGHT)] = k
        assert sorted(landed) == list(range(n_slots))
        m.wait()
        for slot, k in landed.items():
            assert np.array_equal(vt.bits(m.download(slot, W, H)), vt.bits(refs[k])), (slot, k)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 8])
def test_several_frames_per_launch(lib, world):
    """vr_mgpu_frames_async: 1..4 frames (different cameras) per launch and rank, one gather for all of them, strided
    un-permute: every assembled frame equals the single-GPU render of its camera; launches alternate between the buffer sets
    and the buffer sets are re-sized on the way up."""
    W, H = 200, 150
    yaws = [0.2, 0.9, 1.7, 2.4, 3.0, 3.9, 4.5]
    refs = [reference(W, H, capi.LIGHT, y) for y in yaws]
    step, count = hr.stepping_params(24, 24, 24)
    us = [vt.to_capi_uniforms(hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=y)) for y in yaws]
    with mgpu.MultiGpu.local(W, H, [0] * world, _lib=lib) as m:
        for r in range(world):
            scene(m.context(r), W, H, capi.LIGHT)
        m.frame_async(capi.LIGHT)                       # the one-frame form still works beside it
        sets = [(0, 1), (1, 3), (3, 7), (0, 4)]         # launches of 1, 2, 4 and 4 frames
        landed = []
        for lo, hi in sets:
            landed.append((m.frames_async(capi.LIGHT, us[lo:hi]), lo, hi))
        assert [b for b, _, _ in landed] == [1, 0, 1, 0]
        m.wait()
        for b, lo, hi in landed[-2:]:                   # the last launch into each buffer set is still there
            for f in range(hi - lo):
                got = m.download_batch_frame(b, f, W, H)
                assert np.array_equal(vt.bits(got), vt.bits(refs[lo + f][0])), (b, f)
        (comp, covered, _), _ = m.reduce()              # counters: the LAST frame of the last launch
        assert comp == refs[3][1] and covered == refs[3][2]
        with pytest.raises(capi.VrError):
            m.frames_async(capi.LIGHT, us[:5])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3, 8])
def test_root_presents_through_the_permutation(lib, world):
    """VR_MGPU_OUT_PRESENT: the root writes the presented BGRA8 frame straight from the gathered tile-major segments
    (vr_present_tiles_async) -- equal, byte for byte, to un-permuting first and presenting the assembled frame
    (vr_download's present of the single-GPU render); with both outputs on, the float frame is still there.  Also the
    per-stage timeline, ncclCommCount and the device query."""
    W, H = 200, 150
    yaws = [0.3, 1.2, 2.0]
    step, count = hr.stepping_params(24, 24, 24)
    us = [vt.to_capi_uniforms(hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=y)) for y in yaws]
    refs = []
    with capi.Context(W, H, 0) as ctx:
        scene(ctx, W, H, capi.LIGHT)
        for u in us:
            ctx.set_uniforms(u)
            ctx.render(capi.LIGHT)
            frag, bgra, _ = ctx.download(present=True)
            refs.append((frag, bgra))
    with mgpu.MultiGpu.local(W, H, [0] * world, _lib=lib) as m:
        assert m.comm_count() == world and [m.device(r) for r in range(world)] == [0] * world
        for r in range(world):
            scene(m.context(r), W, H, capi.LIGHT)
        m.set_output(mgpu.OUT_PRESENT)
        m.set_stage_timing(True)
        b = m.frames_async(capi.LIGHT, us)
        for f in range(3):
            assert np.array_equal(m.download_present(b, f, W, H), refs[f][1]), f
        march, gather, output, total = m.stage_times(0, b)
        assert march > 0 and gather >= 0 and output > 0 and total >= march + output - 1e-3
        if world > 1:
            assert m.stage_times(world - 1, b)[0] > 0
        m.set_output(mgpu.OUT_FRAME | mgpu.OUT_PRESENT)
        for r in range(world):
            m.context(r).set_uniforms(us[1])
        b = m.frame_async(capi.LIGHT)
        assert np.array_equal(m.download_present(b, 0, W, H), refs[1][1])
        assert np.array_equal(vt.bits(m.download(b, W, H)), vt.bits(refs[1][0]))
        m.set_stage_timing(False)
        with pytest.raises(capi.VrError):
            m.stage_times(0, b)
        with pytest.raises(capi.VrError):
            m.set_output(0)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3])
def test_one_frame_at_a_time_on_the_device(lib, world):
    """vr_mgpu_set_frames_in_flight(1): march, gather and output pass of every launch on one stream per rank -- launches
    enqueued back to back with no wait between them run one after the other on the device; the frames are the single-GPU
    frames in both modes, and switching back and forth drains cleanly."""
    W, H = 200, 150
    yaws = [0.3, 1.2, 2.0, 2.6]
    step, count = hr.stepping_params(24, 24, 24)
    us = [vt.to_capi_uniforms(hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=y)) for y in yaws]
    refs = []
    with capi.Context(W, H, 0) as ctx:
        scene(ctx, W, H, capi.LIGHT)
        for u in us:
            ctx.set_uniforms(u)
            ctx.render(capi.LIGHT)
            refs.append(ctx.download()[0])
    with mgpu.MultiGpu.local(W, H, [0] * world, _lib=lib) as m:
        for r in range(world):
            scene(m.context(r), W, H, capi.LIGHT)
        for mode in (1, 0, 1):
            m.set_frames_in_flight(mode)
            used = []
            for g in (0, 1):          # two launches back to back: buffer sets 0 and 1
                for r in range(world):
                    m.context(r).set_uniforms(us[2 * (mode & 1) + g])
                used.append(m.frame_async(capi.LIGHT))
            m.wait()
            assert used[0] != used[1]
            for g, b in enumerate(used):
                assert np.array_equal(vt.bits(m.download(b, W, H)), vt.bits(refs[2 * (mode & 1) + g])), (mode, g)
        with pytest.raises(capi.VrError):
            m.set_frames_in_flight(-1)
