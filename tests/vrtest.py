"""Shared helpers for the parity tests (harness only)."""
import ctypes as C

import numpy as np

import host_ref as hr
import oracle_binding as ob
from volumerendering_amd import capi

f32 = np.float32


def to_capi_uniforms(u):
    return capi.Uniforms.from_buffer_copy(bytes(u))


def make_volume(kind, n, *, gradient=False, norm01=False, grad_first=False):
    """Reference data-prep order is per scene: normalise->gradient (BasicVolLight) or
    gradient(true)->normalise (VolumeMask / MultiCTRT)."""
    raw = hr.sphere_raw(n) if kind == "sphere" else hr.ct_phantom_raw(n)
    v = hr.raw_to_vec4(raw)
    if grad_first:
        v = ob.precompute_gradient(v, norm01)
        v = ob.normalize_data(v, int(raw.max()))
    else:
        v = ob.normalize_data(v)
        if gradient:
            v = ob.precompute_gradient(v, norm01)
    return v


def dose_volume():
    v = hr.raw_to_vec4(hr.dose_raw(32, 32, 16))
    return ob.normalize_data(v)


def scene(variant, n=16, tf_res=64, thin=False):
    """(volumes, tfs) for a variant, following the slot tables of include/vr.h."""
    o = hr.thin_opacity_tf(tf_res, 0.05) if thin else hr.default_opacity_tf(tf_res)
    tf0 = (o, hr.default_color_tf(tf_res))
    # a second, different TF pair for the RT dose
    o1 = (hr.default_opacity_tf(2 * tf_res) * f32(0.5)).astype(f32)
    c1 = hr.default_color_tf(2 * tf_res).copy()
    c1[:, 1] = f32(0.25)
    tf1 = (o1, c1)
    if variant == capi.BASIC:
        return [make_volume("phantom", n)], [tf0]
    if variant in (capi.LIGHT, capi.LIGHT_INSHADER):
        return [make_volume("phantom", n, gradient=True)], [tf0]
    if variant == capi.VOLUME_MASK:
        return [hr.mask_vec4(n), dose_volume(), make_volume("phantom", n, norm01=True, grad_first=True)], [tf0, tf1]
    if variant == capi.THREE_FILES:
        return [make_volume("phantom", n), dose_volume(), hr.mask_vec4(n)], [tf0, tf1]
    if variant in (capi.MULTI_CTRT, capi.ILLUSTRATIVE):
        return [make_volume("phantom", n, norm01=True, grad_first=True), dose_volume()], [tf0, tf1]
    if variant == capi.TF_CALIB:
        return [make_volume("phantom", n), hr.mask_vec4(n)], [tf0]
    raise ValueError(variant)


def gpu_render(ctx, variant, u, volumes, tfs, present=False):
    for i, v in enumerate(volumes):
        ctx.volume_upload(i, v)
    for i, t in enumerate(tfs):
        ctx.tf_upload(i, t[0], t[1])
    ctx.set_uniforms(to_capi_uniforms(u))
    ctx.render(variant)
    return ctx.download(present=present)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ---- kernel forms of the build under test -----------------------------------------------------------------------------------
# Flavours 2, 3, 4, 5, 9 and volume layout 2 lost every A/B and are compiled only with VR_EXPERIMENTAL_FLAVOURS=1
# (csrc/vr_launch.h); the shipped library rejects them.  Tests take their flavour / layout lists through these filters, so the
# same suite covers both builds.
def experimental() -> bool:
    try:
        return capi.experimental_flavours()
    except Exception:  # noqa: BLE001  (library not built: the GPU tests that would use the list cannot run anyway)
        return False


def flavours(*fl):
    ex = experimental()
    return [f for f in fl if ex or f not in (2, 3, 4, 5, 9, 14)]  # (15, the LDS tiles, is part of the shipped library)


def layouts(*modes):
    ex = experimental()
    return [m for m in modes if ex or m != 2]
