"""Every BASELINE.json configuration (C1..C5) at its FULL size: the HIP frame, through the C++ host surface and the
C ABI, against the CPU oracle on the same scene bench.py measures (volumerendering_amd/workloads.py builds both).

Bar: bit-exact f32 frames and identical composited-sample / covered-pixel counts (the <= 1e-4 max-abs tolerance of
BASELINE.json is asserted as well, so that a future relaxation is visible).  C1-C4: whole frame, default and thin TF.
C5 (1024^3 = 16 GiB of voxels: the 64-bit address kernels, 1773 steps, 3840x2160): whole frame for the default TF,
every 16th tile for the thin one, plus the size-independent properties -- the no-skipping kernel and the skipping
kernels agree bit for bit, and the image-tile partition at world 2 / 8 reassembles the single-GPU frame exactly.

PARITY UNPINNED (see oracle/vr_oracle.h): the reference holds no fixtures and cannot run here; the oracle is the
plain-C restatement of the WGSL.
"""
import os

import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi, host, tiles, workloads as wl

pytestmark = pytest.mark.gpu

NTHREADS = min(len(os.sched_getaffinity(0)), 32)


def oracle_frame(app, variant, vols, W, H):
    ub, volumes, tfs = wl.oracle_inputs(app, vols)
    return ob.render(variant, hr.Uniforms.from_buffer_copy(ub), volumes, tfs, W, H, nthreads=NTHREADS)


def oracle_pixels(app, variant, vols, W, H, pxy):
    ub, volumes, tfs = wl.oracle_inputs(app, vols)
    return ob.render_pixels(variant, hr.Uniforms.from_buffer_copy(ub), volumes, tfs, W, H, pxy, nthreads=NTHREADS)


def gpu_frame(app):
    app.OnRender()
    frag, _, n = app.ReadFrame()
    ctx = app.context()
    return frag, n, ctx.covered_pixels(), ctx.counters()[2]


def assert_frame_equal(frag, ref, what):
    diff = float(np.max(np.abs(frag - ref)))
    assert np.isfinite(ref).all(), what
    assert diff <= 1e-4, (what, diff)
    assert np.array_equal(vt.bits(frag), vt.bits(ref)), (what, diff)


@pytest.mark.parametrize("workload", ["C1", "C2", "C3", "C4"])
def test_config_full_frame_vs_oracle(workload):
    n, W, H, vname = wl.WORKLOADS[workload]
    with host.Application(W, H, 0) as app:
        variant, vols = wl.build_scene(app, workload, "default", quiet=True)
        for tf in ("default", "thin"):
            if tf != "default":
                wl.apply_tf(app, vname, tf)
                app.OnUpdate()
            frag, n_gpu, cov_gpu, fetched = gpu_frame(app)
            ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
            assert_frame_equal(frag, ref, (workload, tf))
            assert (n_gpu, cov_gpu) == (n_ref, cov_ref), (workload, tf)
            assert n_gpu > 0 and 0 < fetched <= n_gpu
            steps, _ = app.stepping()
            assert steps == int(np.sqrt(3.0) * n)  # MiniApp.h:46-54


def test_c3_noisy_air_and_zero_prefix_tf_vs_oracle():
    """The regime real CT data lives in: no voxel class is exactly 0 (air = raw 0..80), preset-style opacity table with a
    real zero prefix; and the worst case -- noisy air under the default ramp, where nothing can be skipped."""
    n, W, H, vname = wl.WORKLOADS["C3"]
    with host.Application(W, H, 0) as app:
        variant, vols = wl.build_scene(app, "C3", "default", air="noisy", quiet=True)
        frag, n_gpu, cov_gpu, fetched = gpu_frame(app)
        ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
        assert_frame_equal(frag, ref, "C3 noisy/default")
        assert (n_gpu, cov_gpu) == (n_ref, cov_ref)
        wl.apply_tf(app, vname, "prefix")
        app.OnUpdate()
        frag, n_gpu, cov_gpu, fetched2 = gpu_frame(app)
        ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
        assert_frame_equal(frag, ref, "C3 noisy/prefix")
        assert (n_gpu, cov_gpu) == (n_ref, cov_ref)
        assert fetched2 < n_gpu  # the zero prefix makes air inert although it is not exactly 0


@pytest.mark.parametrize("workload", ["C3", "C4"])
def test_config_flavours_and_tile_partition_agree_at_full_size(workload):
    """Every kernel flavour, FORCED, at full size against the ORACLE's frame (and the plain no-skipping kernel's counts) --
    the two-steps-ahead kernels (16 / 17: what the default picks for these frames) included, with the flavour that really ran
    asserted so that a silent fallback cannot pass; and the image-tile partition (world 2 and 8) reassembles the single-GPU
    frame exactly."""
    n, W, H, vname = wl.WORKLOADS[workload]
    with host.Application(W, H, 0) as app:
        variant, vols = wl.build_scene(app, workload, "default", quiet=True)
        ctx = app.context()
        ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
        ctx.set_kernel_flavour(1)
        base, n_base, cov_base, _ = gpu_frame(app)
        assert_frame_equal(base, ref, (workload, 1))
        assert (n_base, cov_base) == (n_ref, cov_ref)
        for fl in (0, 6, 11, 10, 12, 13, 16, 17, 18):
            ctx.set_kernel_flavour(fl)
            for rep in range(2 if fl else 5):  # (the default settles on its kernel after a few launches; every one is checked)
                frag, n_f, cov_f, _ = gpu_frame(app)
                assert_frame_equal(frag, ref, (workload, fl, rep))
                assert (n_f, cov_f) == (n_ref, cov_ref), (workload, fl)
            ran = ctx.last_kernel_flavour()
            if fl in (6, 10, 11, 12):
                assert ran == fl, (workload, fl, ran)
            elif fl == 18:
                assert ran == (18 if vname in ("LIGHT", "BASIC") else 6), (workload, fl, ran)
            elif fl == 13:
                assert ran == (13 if vname == "LIGHT" else 12) or ran == 13, (workload, fl, ran)
            elif fl == 16:
                assert ran == (17 if vname == "VOLUME_MASK" else 16), (workload, fl, ran)
            elif fl == 17:
                assert ran == 17, (workload, fl, ran)
        # the same with two launches in flight announced (half-size workgroups, two per CU) and with the thin table
        ctx.set_kernel_flavour(17)
        ctx.hint_frames_in_flight(2)
        frag, n_f, cov_f, _ = gpu_frame(app)
        assert_frame_equal(frag, ref, (workload, "17 in flight"))
        ctx.hint_frames_in_flight(1)
        ctx.set_kernel_flavour(0)
        for world in (2, 8):
            tpr = tiles.tile_count(W, H, 0, world)
            gathered = np.zeros((world, tpr, tiles.TILE, tiles.TILE, 4), dtype=np.float32)
            total = 0
            for r in range(world):
                ctx.render_tiles(variant, r, world)
                nt = ctx.tile_count(r, world)
                t, cnt = ctx.download_tiles(nt)
                gathered[r, :nt] = t
                total += cnt
            assert total == n_base
            assert np.array_equal(vt.bits(tiles.unpack(gathered, W, H, world)), vt.bits(base)), (workload, world)
        for fl in (0, 17):
            ctx.set_kernel_flavour(fl)
            assert_batched_frames_equal(app, variant, W, H, base, n_base)
        ctx.set_kernel_flavour(0)


def assert_batched_frames_equal(app, variant, W, H, frame, n_samples):
    """Three frames in one launch (vr_render_batch_async), the scene's own uniforms each: all equal the single-frame render."""
    ctx = app.context()
    others = [capi.Context(W, H, 0) for _ in range(3)]
    try:
        ctx.render_batch_async(variant, [app.uniforms()] * 3, [o.frame_device_ptr() for o in others], ctx.stream(0))
        assert ctx.counters()[0] == n_samples
        for o in others:
            got, _, _ = o.download()
            assert np.array_equal(vt.bits(got), vt.bits(frame))
    finally:
        for o in others:
            o.close()


def test_c5_16gib_volume_64bit_addressing_vs_oracle():
    """C5: 1024^3 RGBA32F voxels (16 GiB, byte offsets beyond 32 bits), 3840x2160, 1773 steps."""
    n, W, H, vname = wl.WORKLOADS["C5"]
    with host.Application(W, H, 0) as app:
        variant, vols = wl.build_scene(app, "C5", "default", quiet=True)
        ctx = app.context()
        frag, n_gpu, cov_gpu, fetched = gpu_frame(app)
        ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
        assert_frame_equal(frag, ref, "C5 default")
        assert (n_gpu, cov_gpu) == (n_ref, cov_ref)
        del ref
        # no-skipping kernel and the depth-parallel kernels: same bits, same counts
        # (`frag` has just been compared with the oracle: equality with it is equality with the oracle.  16 / 17: the moving gather
        # window of march_p2_kernel, the volume being four windows long)
        for fl in (1, 11, 12, 17, 16, 18):
            ctx.set_kernel_flavour(fl)
            f2, n2, cov2, _ = gpu_frame(app)
            assert np.array_equal(vt.bits(f2), vt.bits(frag)), fl
            assert (n2, cov2) == (n_gpu, cov_gpu), fl
            assert ctx.last_kernel_flavour() == fl, (fl, ctx.last_kernel_flavour())  # (18 on C5: 64-bit addresses from the tables' slots)
        ctx.set_kernel_flavour(0)
        # one rank's share of an 8-GPU partition = the same pixels of the full frame
        world = 8
        for r in (0, 5):
            ctx.render_tiles(variant, r, world)
            nt = ctx.tile_count(r, world)
            t, _ = ctx.download_tiles(nt)
            assert np.array_equal(vt.bits(t), vt.bits(tiles.pack(frag, r, world))), r
        assert_batched_frames_equal(app, variant, W, H, frag, n_gpu)   # the 64-bit-offset batch kernel
        ctx.set_kernel_flavour(17)
        assert_batched_frames_equal(app, variant, W, H, frag, n_gpu)   # ... and the persistent one's (frame, packet) queue
        ctx.set_kernel_flavour(0)
        # thin TF (no ray terminates: the longest accumulation chains): every 16th 64x64 tile against the oracle
        wl.apply_tf(app, vname, "thin")
        app.OnUpdate()
        frag, n_gpu, cov_gpu, _ = gpu_frame(app)
        tx, ty = tiles.tiles_xy(W, H)
        pts = []
        for t in range(3, tx * ty, 16):
            y0, x0 = (t // tx) * tiles.TILE, (t % tx) * tiles.TILE
            ys, xs = np.meshgrid(np.arange(y0, min(y0 + tiles.TILE, H)), np.arange(x0, min(x0 + tiles.TILE, W)), indexing="ij")
            pts.append(np.stack([xs.ravel(), ys.ravel()], axis=1))
        pxy = np.concatenate(pts).astype(np.int32)
        out, _ = oracle_pixels(app, variant, vols, W, H, pxy)
        got = frag[pxy[:, 1], pxy[:, 0]]
        assert np.array_equal(vt.bits(got), vt.bits(out))
        assert float(np.max(np.abs(got - out))) <= 1e-4


@pytest.mark.parametrize("workload", ["C1", "C2", "C3", "C4"])
def test_config_fused_arithmetic_vs_fused_oracle_and_vs_default(workload):
    """vr_set_arithmetic(VR_ARITH_FUSED) at full size: bit-exact against the oracle's fused mode; against the default
    (separately rounded) GPU frame the difference stays within BASELINE.json's 1e-4 except where one of the shader's hard
    thresholds (opacity cut-off, IsInSampleCoords) lands on the other side of a rounding -- those pixels are counted and
    must be rare (they are the reference's own discontinuities, not an error of either mode)."""
    n, W, H, vname = wl.WORKLOADS[workload]
    with host.Application(W, H, 0) as app:
        variant, vols = wl.build_scene(app, workload, "default", quiet=True)
        ctx = app.context()
        sep, n_sep, cov_sep, _ = gpu_frame(app)
        ctx.set_arithmetic(capi.ARITH_FUSED)
        fus, n_fus, cov_fus, fetched = gpu_frame(app)
        with ob.arithmetic(ob.FUSED):
            ref, n_ref, cov_ref = oracle_frame(app, variant, vols, W, H)
        assert_frame_equal(fus, ref, (workload, "fused"))
        assert (n_fus, cov_fus) == (n_ref, cov_ref) and cov_fus == cov_sep
        d = np.max(np.abs(fus - sep), axis=2)
        over = int((d > 1e-4).sum())
        print(f"{workload}: fused vs separate max abs {d.max():.3e}, pixels over 1e-4: {over} of {cov_sep}, "
              f"composited samples {n_fus} vs {n_sep}")
        assert over <= 1e-3 * cov_sep
        assert abs(n_fus - n_sep) <= 1e-4 * n_sep
