"""Device-side data preparation (vr_volume_upload_raw*, vr_volume_normalize, vr_volume_precompute_gradient) against
the oracle's restatement of VolumeFile::NormalizeData / PreComputeGradient: bit-exact, in both scene orders."""
import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(64, 48)
    yield c
    c.close()


@pytest.mark.parametrize("shape", [(24, 24, 24), (5, 9, 300), (1, 1, 1), (33, 2, 65)])
def test_normalise_then_gradient_on_device(ctx, shape):
    rng = np.random.default_rng(11)
    raw = rng.integers(0, 4096, size=shape, dtype=np.uint16)
    raw.flat[-1] = 4000
    ctx.volume_upload_raw(0, raw)
    ref = hr.raw_to_vec4(raw)
    assert np.array_equal(ctx.volume_download(0, shape), ref)
    used = ctx.volume_normalize(0)
    assert used == int(raw.max())
    ref = ob.normalize_data(ref)
    assert np.array_equal(vt.bits(ctx.volume_download(0, shape)), vt.bits(ref))
    ctx.volume_precompute_gradient(0)
    ref = ob.precompute_gradient(ref)
    assert np.array_equal(vt.bits(ctx.volume_download(0, shape)), vt.bits(ref))


def test_gradient_normalised_before_normalisation_on_device(ctx):
    """VolumeMaskApp / MultiCTRTApp order, uint32 path for the dose."""
    raw = synth.ct_phantom_raw(32)
    ctx.volume_upload_raw(0, raw)
    ctx.volume_precompute_gradient(0, True)
    ctx.volume_normalize(0, int(raw.max()))
    ref = ob.normalize_data(ob.precompute_gradient(hr.raw_to_vec4(raw), True), int(raw.max()))
    assert np.array_equal(vt.bits(ctx.volume_download(0, raw.shape)), vt.bits(ref))
    dose = synth.dose_raw(24, 20, 12)
    ctx.volume_upload_raw(1, dose)
    used = ctx.volume_normalize(1)
    assert used == int(dose.max())
    assert np.array_equal(vt.bits(ctx.volume_download(1, dose.shape)), vt.bits(ob.normalize_data(hr.raw_to_vec4(dose))))


def test_render_after_device_prep_equals_host_prep(ctx):
    """Same frame whether the volume was prepared on the host (reference path) or on the device; the brick maxima of
    the empty-space test follow the in-place changes."""
    W, H, n = 64, 48, 32
    raw = synth.ct_phantom_raw(n)
    tf = (hr.default_opacity_tf(64), hr.default_color_tf(64))
    step, count = hr.stepping_params(n, n, n)
    u = hr.make_uniforms(W, H, steps_count=count, step_size=step)
    host_vol = ob.precompute_gradient(ob.normalize_data(hr.raw_to_vec4(raw)))
    ref, n_ref, _ = ob.render(ob.LIGHT, u, [host_vol], [tf], W, H, nthreads=8)
    ctx.volume_upload_raw(0, raw)
    ctx.volume_normalize(0)
    ctx.volume_precompute_gradient(0)
    ctx.tf_upload(0, *tf)
    ctx.set_uniforms(vt.to_capi_uniforms(u))
    ctx.render(capi.LIGHT)
    frag, _, ns = ctx.download()
    assert np.array_equal(vt.bits(frag), vt.bits(ref)) and ns == n_ref
    assert ctx.counters()[2] < ns  # skipping is live: the brick maxima were refreshed after normalisation


def test_prep_errors(ctx):
    c = capi.Context(8, 8)
    with pytest.raises(capi.VrError) as e:
        c.volume_normalize(0)
    assert e.value.code == capi.VR_ERR_NOT_READY
    with pytest.raises(capi.VrError):
        c.volume_precompute_gradient(7)
    c.close()
