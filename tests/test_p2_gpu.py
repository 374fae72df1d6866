"""march_p2_kernel (csrc/vr_p2.h, flavours 16 / 17) beyond the lit / unlit, below-4-GiB, one-frame-per-launch case: the
three-volume composite (VolumeMaskApp.wgsl:182-214, BASELINE config 4), the moving gather window of volumes of 4 GiB and
more (BASELINE config 5; forced onto small volumes here with VR_EXP_P2_WINDOW), launches of several frames, launches in
flight.  Every frame against the CPU oracle, bit for bit; per-packet records against march_kernel's."""
import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi

pytestmark = pytest.mark.gpu
f32 = np.float32


def zero_prefix_tf(res, zeros, top=0.3):
    o = np.zeros(res, dtype=f32)
    if zeros < res:
        o[zeros:] = np.linspace(0.0, top, res - zeros, dtype=f32)
    return o, hr.default_color_tf(res)


def check(ctx, variant, u, vols, tfs, W, H):
    if (ctx.width, ctx.height) != (W, H):
        ctx.resize(W, H)
    frag, _, n = vt.gpu_render(ctx, variant, u, vols, tfs)
    ref, n_ref, cov_ref = ob.render(variant, u, vols, tfs, W, H, nthreads=8)
    assert np.isfinite(ref).all()
    assert float(np.max(np.abs(frag - ref))) <= 1e-4
    assert np.array_equal(vt.bits(frag), vt.bits(ref)), f"max abs diff {np.max(np.abs(frag - ref))}"
    assert n == n_ref and ctx.covered_pixels() == cov_ref
    return frag, n


CAMERAS = [dict(), dict(yaw=1.0, pitch=-0.4, distance=1.6), dict(yaw=-2.3, pitch=0.7, distance=0.6), dict(yaw=0.0, pitch=0.0, distance=1.1),
           dict(yaw=3.14159, pitch=0.05, distance=0.9), dict(clip_x=(0.5, 0.1), clip_z=(0.0, 0.45)), dict(clip_y=(0.3, 0.35), yaw=0.4),
           dict(toggles=(1, 1, 0, 0), yaw=2.0), dict(steps_count=3), dict(steps_count=1)]


def composite_scene(n, zeros):
    """VolumeMaskApp's three volumes: two small structures in the mask (most bricks' mask record is 0: the on-demand path
    is taken by a few packets only, beside packets that never fetch the mask), the dose, the CT with its gradient."""
    vols, tfs = vt.scene(capi.VOLUME_MASK, n=n)
    return vols, [zero_prefix_tf(64, zeros, top=0.6), tfs[1]]


@pytest.mark.parametrize("zeros", [9, 0])
def test_three_volume_composite_two_steps_ahead(zeros):
    """Flavour 17 on VOLUME_MASK: frames bit-equal to the oracle, per-packet composited / covered / fetched counts equal to
    march_kernel's; 16 (the no-skip form) resolves to 17 for this shader (its mask records ARE the skipping's)."""
    n, W, H = 48, 200, 120
    step, count = hr.stepping_params(n, n, n)
    vols, tfs = composite_scene(n, zeros)
    with capi.Context(W, H, 0) as ctx:
        for kw in CAMERAS:
            args = dict(steps_count=count, step_size=step)
            args.update(kw)
            u = hr.make_uniforms(W, H, **args)
            recs = {}
            for fl in (6, 17, 16):
                ctx.set_kernel_flavour(fl)
                check(ctx, capi.VOLUME_MASK, u, vols, tfs, W, H)
                assert ctx.last_kernel_flavour() == (6 if fl == 6 else 17)
                recs[fl] = ctx.block_trace().astype(np.uint64)
            a, b = recs[6], recs[17]
            assert a.shape == b.shape and a.shape[0] > 0
            assert np.array_equal(a[:, :3], b[:, :3]), kw
        # fused arithmetic: its own oracle
        ctx.set_arithmetic(capi.ARITH_FUSED)
        ctx.set_kernel_flavour(17)
        with ob.arithmetic(ob.FUSED):
            for kw in CAMERAS[:4]:
                args = dict(steps_count=count, step_size=step)
                args.update(kw)
                check(ctx, capi.VOLUME_MASK, hr.make_uniforms(W, H, **args), vols, tfs, W, H)
        assert ctx.last_kernel_flavour() == 17


def test_three_volume_composite_without_brick_records_falls_back():
    """A light uniform that is not finite switches exact skipping off (the host cannot prove rgb * 0 == 0; this shader never
    reads the uniform, its light is a constant): no brick records, no on-demand mask fetch -- 17 resolves to the persistent
    kernel without the pipeline (12) and the frame is still the oracle's."""
    n, W, H = 24, 96, 80
    step, count = hr.stepping_params(n, n, n)
    vols, tfs = vt.scene(capi.VOLUME_MASK, n=n)
    with capi.Context(W, H, 0) as ctx:
        ctx.set_kernel_flavour(17)
        u = hr.make_uniforms(W, H, steps_count=count, step_size=step, light_pos=(float("inf"), 5.0, 0.0, 1.0))
        check(ctx, capi.VOLUME_MASK, u, vols, tfs, W, H)
        assert ctx.last_kernel_flavour() == 12


@pytest.mark.parametrize("variant", [capi.LIGHT, capi.BASIC, capi.VOLUME_MASK])
@pytest.mark.parametrize("slabs", [3, 4, 7, 1000])
def test_moving_gather_window(variant, slabs, monkeypatch):
    """Volumes of 4 GiB and more are gathered through a window of whole z-slabs of bricks that follows the packet
    (march_p2_kernel<.., WIN>).  VR_EXP_P2_WINDOW forces that form onto a 48^3 volume with windows of 3 (the base corner's
    slab may be the window's first or second: it moves every few steps, and a packet whose rays span three slabs leaves
    the pipelined loop for the plain one), 4 and 7 slabs and one that holds the whole volume: cameras along z (every slab is
    crossed), oblique, from inside, clipped.  Frames bit-equal to the oracle, records equal to march_kernel's."""
    n, W, H = 48, 200, 120
    slab = ((n + 3) // 4) ** 2 * 64
    monkeypatch.setenv("VR_EXP_P2_WINDOW", str(slabs * slab + 17))
    step, count = hr.stepping_params(n, n, n)
    if variant == capi.VOLUME_MASK:
        vols, tfs = composite_scene(n, 9)
    else:
        vols, tfs = vt.scene(variant, n=n)
        tfs = [zero_prefix_tf(64, 9, top=0.6)]
    with capi.Context(W, H, 0) as ctx:
        for kw in CAMERAS[:8]:
            args = dict(steps_count=count, step_size=step)
            args.update(kw)
            u = hr.make_uniforms(W, H, **args)
            recs = {}
            for fl in (6, 17, 16):
                ctx.set_kernel_flavour(fl)
                check(ctx, variant, u, vols, tfs, W, H)
                assert ctx.last_kernel_flavour() == (fl if (fl == 6 or variant != capi.VOLUME_MASK) else 17)
                recs[fl] = ctx.block_trace().astype(np.uint64)
            assert np.array_equal(recs[6][:, :3], recs[17][:, :3]), kw
            assert np.array_equal(recs[6][:, :2], recs[16][:, :2]), kw


def _batch_uniforms(W, H, count, step):
    return [hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=0.6),
            hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=1.4, pitch=-0.2, clip_x=(0.2, 0.1)),
            hr.make_uniforms(W, H, steps_count=count // 2, step_size=step * 2, yaw=2.9, distance=1.6, toggles=(1, 0, 0, 0)),
            hr.make_uniforms(W, H, steps_count=count, step_size=step, yaw=0.1, fragment_mode=2)]


@pytest.mark.parametrize("variant", [capi.BASIC, capi.LIGHT, capi.VOLUME_MASK])
@pytest.mark.parametrize("flavour", [16, 17])
@pytest.mark.parametrize("window", [0, 4])
def test_frames_of_one_persistent_launch(variant, flavour, window, monkeypatch):
    """vr_render_batch_async through march_p2_kernel<.., BATCH>: the queue hands out (frame, packet) items; n = 1..4 frames with
    different cameras, clip boxes, step counts and a debug mode, each bit-equal to its single-frame render; launch after launch
    (the queue heads must be zero again each time); with and without the moving window."""
    n, W, H = 24, 136, 100
    if window:
        monkeypatch.setenv("VR_EXP_P2_WINDOW", str(window * ((n + 3) // 4) ** 2 * 64 + 5))
    step, count = hr.stepping_params(n, n, n)
    us = _batch_uniforms(W, H, count, step)
    if variant == capi.VOLUME_MASK:
        vols, tfs = composite_scene(n, 5)
    else:
        vols, tfs = vt.scene(variant, n=n)
    with capi.Context(W, H, 0) as ctx:
        ctx.set_kernel_flavour(6)
        refs = [vt.gpu_render(ctx, variant, u, vols, tfs) for u in us]
        others = [capi.Context(W, H, 0) for _ in range(4)]
        try:
            ctx.set_kernel_flavour(flavour)
            for rnd in range(3):
                for k in (1, 2, 3, 4):
                    order = [(j + rnd) % 4 for j in range(k)]
                    ctx.render_batch_async(variant, [vt.to_capi_uniforms(us[j]) for j in order],
                                           [others[j].frame_device_ptr() for j in range(k)], ctx.stream(0))
                    comp = ctx.counters()[0]
                    assert ctx.last_kernel_flavour() == (17 if variant == capi.VOLUME_MASK else flavour)
                    assert comp == refs[order[-1]][2], (k, rnd)
                    ctx.resize(W, H)
                    for j, src in enumerate(order):
                        got, _, _ = others[j].download()
                        assert np.array_equal(vt.bits(got), vt.bits(refs[src][0])), (k, rnd, j)
        finally:
            for o in others:
                o.close()


@pytest.mark.parametrize("variant", [capi.LIGHT, capi.VOLUME_MASK])
def test_persistent_launches_in_flight(variant):
    """vr_hint_frames_in_flight(2): march_p2_kernel is launched as two half-size workgroups per CU so that the next launch moves
    in as this one drains; bursts of launches on two streams, 1080p (several packets per wavefront) and a small frame (fewer
    packets than wavefronts), equal to the one-at-a-time frames."""
    n = 24
    step, count = hr.stepping_params(n, n, n)
    if variant == capi.VOLUME_MASK:
        vols, tfs = composite_scene(n, 5)
    else:
        vols, tfs = vt.scene(variant, n=n)
    for W, H in ((1920, 1080), (40, 24)):
        us = [hr.make_uniforms(W, H, steps_count=count, step_size=step, distance=0.9, yaw=0.3 * k) for k in range(4)]
        with capi.Context(W, H, 0) as ctx:
            ctx.set_kernel_flavour(6)
            refs = [vt.gpu_render(ctx, variant, u, vols, tfs) for u in us]
            outs = [capi.Context(W, H, 0) for _ in range(4)]
            try:
                for fl in (17, 16):
                    ctx.set_kernel_flavour(fl)
                    ctx.hint_frames_in_flight(2)
                    for burst in range(3):
                        for k, u in enumerate(us):
                            ctx.set_uniforms(vt.to_capi_uniforms(u))
                            ctx.render_async(variant, outs[k].frame_device_ptr(), ctx.stream(k & 1))
                        comp = ctx.counters()[0]   # (waits for the last launch)
                        assert comp == refs[3][2]
                        ctx.resize(W, H)           # drains the device
                        for k in range(4):
                            got, _, _ = outs[k].download()
                            assert np.array_equal(vt.bits(got), vt.bits(refs[k][0])), (fl, burst, k)
                    assert ctx.last_kernel_flavour() == (17 if variant == capi.VOLUME_MASK else fl)
                    ctx.hint_frames_in_flight(1)
            finally:
                for o in outs:
                    o.close()


@pytest.mark.parametrize("host_wait", ["0", "1"])
def test_one_frame_at_a_time_on_one_stream_far_ahead_of_the_device(monkeypatch, host_wait):
    """The stream-ordered loop of bench.py's serial leg: 40 launches enqueued on one stream without waiting for any, four cameras in
    turn into four buffers -- more launches than record slots (8) and launch-order buffers (16), so every launch takes an earlier
    launch's order and re-uses a record slot whose sort it must be ordered behind (one wait for the younger of the two sorts; with
    VR_EXP_HOST_ORDER_WAIT=1 on the host).  Every frame equals its reference, the last launch's counts too; default and forced kernels."""
    monkeypatch.setenv("VR_EXP_HOST_ORDER_WAIT", host_wait)
    n, W, H = 24, 640, 360
    step, count = hr.stepping_params(n, n, n)
    vols, tfs = vt.scene(capi.LIGHT, n=n)
    us = [hr.make_uniforms(W, H, steps_count=count, step_size=step, distance=0.9, yaw=0.3 * k) for k in range(4)]
    with capi.Context(W, H, 0) as ctx:
        ctx.set_kernel_flavour(6)
        refs = [vt.gpu_render(ctx, capi.LIGHT, u, vols, tfs) for u in us]
        outs = [capi.Context(W, H, 0) for _ in range(4)]
        try:
            for fl in (0, 17, 6):
                ctx.set_kernel_flavour(fl)
                ctx.hint_frames_in_flight(1)
                for k in range(40):
                    ctx.set_uniforms(vt.to_capi_uniforms(us[k & 3]))
                    ctx.render_async(capi.LIGHT, outs[k & 3].frame_device_ptr(), ctx.stream(0))
                assert ctx.counters()[0] == refs[3][2]   # (waits for the last launch: camera 39 & 3)
                ctx.resize(W, H)                          # drains the device
                for k in range(4):
                    got, _, _ = outs[k].download()
                    assert np.array_equal(vt.bits(got), vt.bits(refs[k][0])), (fl, k)
        finally:
            for o in outs:
                o.close()


# ---- the default's measured kernel choice (flavour 0; csrc/vr_api.hip: tune_pick) -------------------------------------------------
def test_measured_kernel_choice_settles_and_changes_nothing():
    """Flavour 0 tries the eligible kernel forms in turn on the caller's own frames and keeps the fastest.  Whatever runs, every
    frame is the reference frame, bit for bit; the choice is made within a bounded number of launches, is one of the candidates,
    is then stable (the flavour that ran no longer changes), has a measured cost for every candidate, and a new table or a new
    frames-in-flight hint opens a trial of its own."""
    n, W, H = 48, 640, 400
    vols, tfs = vt.scene(capi.LIGHT, n=n)
    step, count = hr.stepping_params(n, n, n)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step, distance=1.0)
    with capi.Context(W, H, 0) as ctx:
        ctx.set_kernel_flavour(6)
        ref, _, n_ref = vt.gpu_render(ctx, capi.LIGHT, u, vols, tfs)
        ctx.set_kernel_flavour(0)
        ran = []
        for k in range(40):
            ctx.render(capi.LIGHT)
            frag, _, ns = ctx.download()
            assert ns == n_ref and np.array_equal(vt.bits(frag), vt.bits(ref)), k
            ran.append(ctx.last_kernel_flavour())
        cands, ms, chosen = ctx.kernel_choice()
        assert len(cands) >= 2 and chosen >= 0, (cands, ms, chosen, ran)
        assert all(m > 0.0 for m in ms), (cands, ms)
        assert set(ran) <= set(cands), (ran, cands)
        assert len(set(ran)) >= 2                      # more than one form really ran during the trial
        assert len(set(ran[-8:])) == 1 and ran[-1] == cands[chosen], (ran, cands, chosen)
        assert ms[chosen] <= min(ms) * 1.03             # the kept one is the fastest measured (the prior keeps a 2 % bonus)
        # a new table: the scene's key changes, a new trial runs, the frames are the new reference's
        o2, c2 = zero_prefix_tf(64, 20, top=0.5)
        ctx.tf_upload(0, o2, c2)
        ctx.set_kernel_flavour(6)
        ctx.render(capi.LIGHT)
        ref2, _, n2 = ctx.download()
        ctx.set_kernel_flavour(0)
        ran2 = []
        for k in range(40):
            ctx.render(capi.LIGHT)
            frag, _, ns = ctx.download()
            assert ns == n2 and np.array_equal(vt.bits(frag), vt.bits(ref2)), k
            ran2.append(ctx.last_kernel_flavour())
        assert len(set(ran2)) >= 2 and len(set(ran2[-8:])) == 1
        # frames in flight: its own trial (asynchronous launches on two streams), same frames
        ctx.hint_frames_in_flight(2)
        outs = [capi.Context(W, H, 0) for _ in range(2)]
        try:
            for k in range(48):
                ctx.render_async(capi.LIGHT, outs[k & 1].frame_device_ptr(), ctx.stream(k & 1))
                if k % 8 == 7:
                    ctx.counters()
                    for o in outs:
                        got, _, _ = o.download()
                        assert np.array_equal(vt.bits(got), vt.bits(ref2)), k
            cands3, ms3, chosen3 = ctx.kernel_choice()
            assert chosen3 >= 0 and all(m > 0.0 for m in ms3), (cands3, ms3, chosen3)
        finally:
            ctx.hint_frames_in_flight(1)
            for o in outs:
                o.close()


def test_measured_kernel_choice_can_be_switched_off(monkeypatch):
    monkeypatch.setenv("VR_EXP_TUNE", "0")
    n, W, H = 24, 320, 200
    vols, tfs = vt.scene(capi.LIGHT, n=n)
    step, count = hr.stepping_params(n, n, n)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step)
    with capi.Context(W, H, 0) as ctx:
        ran = set()
        for _ in range(12):
            vt.gpu_render(ctx, capi.LIGHT, u, vols, tfs)
            ran.add(ctx.last_kernel_flavour())
        assert ctx.kernel_choice()[0] == []


@pytest.mark.parametrize("variant", [capi.LIGHT, capi.BASIC, capi.VOLUME_MASK])
@pytest.mark.parametrize("flavour", [17, 16])
def test_rank_shares_through_the_two_steps_ahead_kernel(variant, flavour):
    """A rank's share of the tiles (packed, tile t -> rank t mod world) marched by the persistent kernel: the tiles of every rank,
    un-permuted, are the single-GPU frame bit for bit and the counts add up -- world 2, 3 and 8, a ragged viewport (tiles that
    hang over both edges), a viewport of one tile, launch after launch on one context (the queue heads), and several frames per
    launch.  (The measured choice keeps this kernel for batched shares: bench.py's multi-rank legs.)"""
    n = 24
    step, count = hr.stepping_params(n, n, n)
    if variant == capi.VOLUME_MASK:
        vols, tfs = composite_scene(n, 5)
    else:
        vols, tfs = vt.scene(variant, n=n)
    for W, H in ((200, 150), (1920, 1080), (64, 64)):
        u = hr.make_uniforms(W, H, steps_count=count, step_size=step, distance=1.0)
        with capi.Context(W, H, 0) as ctx:
            ctx.set_kernel_flavour(6)
            full, _, n_full = vt.gpu_render(ctx, variant, u, vols, tfs)
            ctx.set_kernel_flavour(flavour)
            tx, ty = (W + 63) // 64, (H + 63) // 64
            for world in (2, 3, 8):
                frame = np.zeros_like(full)
                total = 0
                for rank in range(world):
                    ctx.render_tiles(variant, rank, world)
                    cnt = ctx.tile_count(rank, world)
                    if cnt:
                        assert ctx.last_kernel_flavour() == (17 if variant == capi.VOLUME_MASK else flavour)
                    tl, ns = ctx.download_tiles(cnt)
                    total += ns
                    for i in range(cnt):
                        t = rank + i * world
                        y0, x0 = (t // tx) * 64, (t % tx) * 64
                        h, w = min(64, H - y0), min(64, W - x0)
                        frame[y0:y0 + h, x0:x0 + w] = tl[i, :h, :w]
                        assert not tl[i, h:, :].any() and not tl[i, :, w:].any()
                assert np.array_equal(vt.bits(frame), vt.bits(full)), (W, H, world)
                assert total == n_full, (W, H, world)


# ---- flavour 18: march_kernel with its slot arithmetic from LDS tables (csrc/vr_kernels.h: make_cell_lut) ---------------------------
@pytest.mark.parametrize("variant", [capi.LIGHT, capi.BASIC, capi.LIGHT_INSHADER])
def test_slot_tables_in_the_one_lane_kernel(variant):
    """Flavour 18 = flavour 6 with make_cell() reading the clamp-to-edge texel pairs' slot terms from per-axis tables in the
    workgroup's LDS: frames bit-equal to the oracle, per-packet records equal to flavour 6's; volumes whose sides are not multiples
    of the brick edge and differ per axis (the tables' lengths and offsets), cameras that see the volume's faces and edges (the
    clamped pairs at t = -1 and t = n - 1), clip boxes, jitter, one step; the x-fastest layout has no tables (runs as 6); several
    frames per launch; other shaders run as 6."""
    W, H = 200, 120
    rng = np.random.default_rng(5)
    for dims in ((24, 24, 24), (37, 18, 29), (7, 50, 13)):
        nx, ny, nz = dims
        raw = (rng.integers(0, 3000, size=(nz, ny, nx)) * (rng.random((nz, ny, nx)) > 0.6)).astype(np.uint16)
        vol = ob.normalize_data(hr.raw_to_vec4(raw))
        if variant != capi.BASIC:
            vol = ob.precompute_gradient(vol)
        tf = zero_prefix_tf(64, 6, top=0.7)
        step, count = hr.stepping_params(nx, ny, nz)
        with capi.Context(W, H, 0) as ctx:
            for kw in CAMERAS[:9]:
                args = dict(steps_count=count, step_size=step)
                args.update(kw)
                u = hr.make_uniforms(W, H, **args)
                recs = {}
                for fl in (6, 18):
                    ctx.set_kernel_flavour(fl)
                    check(ctx, variant, u, [vol], [tf], W, H)
                    assert ctx.last_kernel_flavour() == fl
                    recs[fl] = ctx.block_trace().astype(np.uint64)
                assert np.array_equal(recs[6][:, :3], recs[18][:, :3]), (dims, kw)
            ctx.set_volume_layout(3)   # the reference's x-fastest arrays: no bricked copy to build tables for
            ctx.set_kernel_flavour(18)
            check(ctx, variant, hr.make_uniforms(W, H, steps_count=count, step_size=step), [vol], [tf], W, H)
            assert ctx.last_kernel_flavour() == 6
            ctx.set_volume_layout(0)
    # several frames per launch, and a shader with two volumes (falls back)
    n = 24
    step, count = hr.stepping_params(n, n, n)
    us = _batch_uniforms(136, 100, count, step)
    vols, tfs = vt.scene(variant, n=n)
    with capi.Context(136, 100, 0) as ctx:
        ctx.set_kernel_flavour(6)
        refs = [vt.gpu_render(ctx, variant, u, vols, tfs) for u in us]
        others = [capi.Context(136, 100, 0) for _ in range(4)]
        try:
            ctx.set_kernel_flavour(18)
            ctx.render_batch_async(variant, [vt.to_capi_uniforms(u) for u in us], [o.frame_device_ptr() for o in others], ctx.stream(0))
            assert ctx.counters()[0] == refs[3][2] and ctx.last_kernel_flavour() == 18
            ctx.resize(136, 100)
            for o, ref in zip(others, refs):
                got, _, _ = o.download()
                assert np.array_equal(vt.bits(got), vt.bits(ref[0]))
        finally:
            for o in others:
                o.close()
        v2, t2 = vt.scene(capi.MULTI_CTRT, n=n)
        ctx.set_kernel_flavour(18)
        check(ctx, capi.MULTI_CTRT, us[0], v2, t2, 136, 100)
        assert ctx.last_kernel_flavour() == 6


def test_measured_kernel_choice_with_interleaved_shapes():
    """Two launch shapes taking turns on one context (a rank's share and another rank's: what a one-process driver of several
    ranks does): each has its own trial, measured on its OWN launches although the other shape's lie in between in the context's
    record ring; both settle with a cost for every candidate, and every frame is the reference's."""
    n, W, H = 32, 512, 384
    vols, tfs = vt.scene(capi.LIGHT, n=n)
    step, count = hr.stepping_params(n, n, n)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step, distance=1.0)
    with capi.Context(W, H, 0) as ctx:
        ctx.set_kernel_flavour(6)
        vt.gpu_render(ctx, capi.LIGHT, u, vols, tfs)
        refs = []
        for r in (0, 1):
            ctx.render_tiles(capi.LIGHT, r, 2)
            refs.append(ctx.download_tiles(ctx.tile_count(r, 2)))
        ctx.set_kernel_flavour(0)
        seen = [set(), set()]
        for k in range(60):
            for r in (0, 1):
                ctx.render_tiles(capi.LIGHT, r, 2)
                tl, ns = ctx.download_tiles(ctx.tile_count(r, 2))
                assert ns == refs[r][1] and np.array_equal(vt.bits(tl), vt.bits(refs[r][0])), (k, r)
                seen[r].add(ctx.last_kernel_flavour())
                if k == 59:
                    cands, ms, chosen = ctx.kernel_choice()
                    assert chosen >= 0 and len(cands) >= 2 and all(m > 0.0 for m in ms), (r, cands, ms, chosen)
                    assert ms[chosen] <= min(ms) * 1.03
        assert all(len(s) >= 2 for s in seen)


def blob_volume(n, lo, hi, lit):
    """Density 0 everywhere but a box of voxels [lo, hi) per axis (0.2 .. 0.9, varying), with its gradient for the lit shader:
    the active bricks' box is small, off-centre, and may touch the volume's edge.  n: voxels per axis, or (nx, ny, nz)."""
    nx, ny, nz = (n, n, n) if np.isscalar(n) else n
    a = np.zeros((nz, ny, nx), dtype=f32)
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    inside = ((x >= lo[0]) & (x < hi[0]) & (y >= lo[1]) & (y < hi[1]) & (z >= lo[2]) & (z < hi[2]))
    a[inside] = (0.2 + 0.7 * ((x + 2 * y + 3 * z) % 17) / 16.0).astype(f32)[inside]
    v = np.zeros((nz, ny, nx, 4), dtype=f32)
    v[..., 3] = a
    if lit:
        v = ob.precompute_gradient(np.ascontiguousarray(v), False)  # (the oracle's PreComputeGradient: VolumeFile.cpp:196-257)
    return np.ascontiguousarray(v)


BLOBS = [((30, 8, 20), (44, 20, 28)),   # small, off-centre
         ((0, 0, 0), (9, 7, 5)),        # in the volume's first corner (edge bricks: positions beyond them map to them)
         ((39, 41, 43), (48, 48, 48)),  # in its last corner
         ((0, 20, 0), (48, 23, 48)),    # a slab across the whole volume
         ((10, 10, 10), (10, 10, 10))]  # nothing at all: no active brick, every ray passes the volume without a sample


@pytest.mark.parametrize("blob", range(len(BLOBS)))
@pytest.mark.parametrize("variant", [capi.LIGHT, capi.BASIC])
def test_box_of_the_active_bricks(blob, variant):
    """The approach loops (march_p2_kernel, march_packet) take a ray through whatever lies outside the box of the active bricks
    without a look-up, and march_p2_kernel leaves its pipelined loop behind it: frames and counts against the oracle, for boxes that
    are small, touch the volume's edges, span it, or do not exist, from outside and from inside the volume, with clip planes,
    variable steps and jitter; the per-packet records of the kernels agree with each other."""
    n, W, H = 48, 160, 96
    lo, hi = BLOBS[blob]
    step, count = hr.stepping_params(n, n, n)
    vols = [blob_volume(n, lo, hi, variant == capi.LIGHT)]
    tfs = [zero_prefix_tf(64, 3, top=0.5)]
    cams = CAMERAS + [dict(yaw=0.3, pitch=0.2, distance=0.15), dict(yaw=-1.2, pitch=-0.9, distance=0.3), dict(yaw=1.5707963, pitch=0.0, distance=1.0)]
    with capi.Context(W, H, 0) as ctx:
        for kw in cams:
            args = dict(steps_count=count, step_size=step)
            args.update(kw)
            u = hr.make_uniforms(W, H, **args)
            recs = {}
            for fl in (1, 6, 17, 12):
                ctx.set_kernel_flavour(fl)
                check(ctx, variant, u, vols, tfs, W, H)
                recs[fl] = ctx.block_trace().astype(np.uint64)
            assert ctx.last_kernel_flavour() == 12
            for fl in (17, 12):
                assert np.array_equal(recs[6][:, :3], recs[fl][:, :3]), (kw, fl)
            assert np.array_equal(recs[6][:, :2], recs[1][:, :2]), kw  # (1 fetches every in-box sample: its third word differs)


def test_box_of_the_active_bricks_in_a_volume_with_three_different_sides():
    """The same with nx != ny != nz (the box is kept in uvw: one scale per axis, bricks that hang over the volume's far faces) and
    a blob that is not aligned with the bricks."""
    dims, W, H = (40, 22, 57), 160, 96
    step, count = hr.stepping_params(*dims)
    tfs = [zero_prefix_tf(64, 3, top=0.5)]
    for lo, hi in (((21, 3, 30), (38, 9, 49)), ((0, 17, 50), (5, 22, 57))):
        vols = [blob_volume(dims, lo, hi, True)]
        with capi.Context(W, H, 0) as ctx:
            for kw in CAMERAS[:7] + [dict(yaw=0.3, pitch=0.2, distance=0.15)]:
                args = dict(steps_count=count, step_size=step)
                args.update(kw)
                u = hr.make_uniforms(W, H, **args)
                recs = {}
                for fl in (6, 17):
                    ctx.set_kernel_flavour(fl)
                    check(ctx, capi.LIGHT, u, vols, tfs, W, H)
                    assert ctx.last_kernel_flavour() == fl
                    recs[fl] = ctx.block_trace().astype(np.uint64)
                assert np.array_equal(recs[6][:, :3], recs[17][:, :3]), kw
