"""The C++ multi-GPU frame loop (libvr_mgpu.so: tile render -> ncclGather -> un-permute, two frames in flight) on the
one-GPU box: a world of one exercises every call RCCL and the stream ordering need; the assembled frame must equal
vr_render bit for bit.  Both ways of driving it: one process per GPU (vr_mgpu_create) and one process owning the
contexts (vr_mgpu_create_local).  N > 1 is unmeasured on hardware (the driver's SCALE run is the only 8-GPU run)."""
import numpy as np
import pytest

import host_ref as hr
import vrtest as vt
from volumerendering_amd import capi, mgpu

pytestmark = pytest.mark.gpu


def scene(ctx, W, H, variant):
    vols, tfs = vt.scene(variant, n=24)
    step, count = hr.stepping_params(24, 24, 24)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step)
    for i, v in enumerate(vols):
        ctx.volume_upload(i, v)
    for i, t in enumerate(tfs):
        ctx.tf_upload(i, t[0], t[1])
    ctx.set_uniforms(vt.to_capi_uniforms(u))


@pytest.mark.parametrize("variant", [capi.LIGHT, capi.VOLUME_MASK])
def test_world_of_one_per_process_mode(variant):
    W, H = 200, 150   # ragged: 4 x 3 tiles, the last row / column partly outside the viewport
    with capi.Context(W, H, 0) as ctx:
        scene(ctx, W, H, variant)
        ctx.render(variant)
        ref, _, n_ref = ctx.download()
        cov = ctx.covered_pixels()
        with mgpu.MultiGpu(ctx.h, 0, 1, mgpu.unique_id()) as m:
            assert m.world() == 1 and m.local_ranks() == 1 and "ncclGather" in m.backend()
            slots = [m.frame_async(variant) for _ in range(5)]   # pipelined two deep
            assert slots == [0, 1, 0, 1, 0]
            m.wait()
            for which in (0, 1):
                got = m.download(which, W, H)
                assert np.array_equal(vt.bits(got), vt.bits(ref)), which
            (comp, covered, fetched), mx = m.reduce(3.5)
            assert comp == n_ref and covered == cov and 0 < fetched <= comp and mx == 3.5
            assert m.frame_device_ptr(0) and m.frame_device_ptr(1) and m.frame_device_ptr(0) != m.frame_device_ptr(1)


def test_world_of_one_local_mode_owns_its_context():
    W, H = 130, 70
    with mgpu.MultiGpu.local(W, H, [0]) as m:
        ctx = m.context(0)
        scene(ctx, W, H, capi.BASIC)
        ctx.render(capi.BASIC)
        ref, _, n_ref = ctx.download()
        m.frame_async(capi.BASIC)
        got = m.download(0, W, H)
        assert np.array_equal(vt.bits(got), vt.bits(ref))
        (comp, _, _), _ = m.reduce()
        assert comp == n_ref


def test_bad_arguments_fail_with_a_message():
    with pytest.raises(capi.VrError):
        mgpu.MultiGpu.local(64, 64, [999])
    with capi.Context(64, 64, 0) as ctx:
        with pytest.raises(capi.VrError):
            mgpu.MultiGpu(ctx.h, 3, 2, mgpu.unique_id())
