"""Scale-proofing of the two shortcuts that decide whether work is skipped at all (both argued with margins in the
kernel sources; here they are made to matter):

  hit_rectangle (vr_api.hip)  rays outside the projected box rectangle (+ 3 px) skip ray set-up entirely.  Checked at
                              1080p and 4K with the box large on screen, partly off screen, edge-on, and with the camera's
                              near plane within 1e-4 of a face: covered-pixel counts and frames must equal the oracle's.
  n_inside (march kernels)    the first n steps of a ray are provably inside IsInSampleCoords ("0.1 % + 2 steps" short of the
                              far bound), the six compares are skipped there.  Checked with 4 000-step rays (more than
                              twice C5's 1 773 accumulated additions) and clip planes placed exactly on / one and two ulps
                              either side of a position a ray really takes.
"""
import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi

pytestmark = pytest.mark.gpu
f32 = np.float32


def small_scene(n=8, lit=True):
    raw = hr.ct_phantom_raw(16)[4:4 + n, 4:4 + n, 4:4 + n]
    v = ob.normalize_data(hr.raw_to_vec4(raw))
    if lit:
        v = ob.precompute_gradient(v)
    return v, (hr.thin_opacity_tf(32, 0.02), hr.default_color_tf(32))


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160)])
def test_hit_rectangle_margin_at_high_resolution(size):
    W, H = size
    v, tf = small_scene()
    cams = [dict(distance=1.2, yaw=0.6, pitch=0.35),           # the benchmark camera
            dict(distance=0.62, yaw=0.0, pitch=0.0),            # box fills the screen, corners off screen
            dict(distance=0.2601, yaw=0.0, pitch=0.0),          # near plane 1e-4 in front of the z face
            dict(distance=0.26, yaw=0.0, pitch=0.0),            # ... on it
            dict(distance=0.2599, yaw=0.0, pitch=0.0),          # ... 1e-4 inside: the face's middle is clipped away
            dict(distance=0.9, yaw=np.pi / 2, pitch=0.0),       # edge-on: looking along x exactly
            dict(distance=0.9, yaw=np.pi / 4, pitch=float(np.arctan(1 / np.sqrt(2)))),  # along the box diagonal
            dict(distance=0.8, yaw=0.3, pitch=np.pi / 2 - 1e-3),  # almost straight down
            dict(distance=6.0, yaw=2.0, pitch=-0.7)]            # far away: a few hundred pixels
    with capi.Context(W, H, 0) as ctx:
        ctx.volume_upload(0, v)
        ctx.tf_upload(0, *tf)
        for cam in cams:
            u = hr.make_uniforms(W, H, steps_count=14, step_size=1 / 8, **cam)
            ctx.set_uniforms(vt.to_capi_uniforms(u))
            ref, n_ref, cov_ref = ob.render(capi.LIGHT, u, [v], [tf], W, H, nthreads=32)
            for fl in (0, 1):
                ctx.set_kernel_flavour(fl)
                ctx.render(capi.LIGHT)
                frag, _, ns = ctx.download()
                assert ctx.covered_pixels() == cov_ref, (cam, fl)
                assert ns == n_ref and np.array_equal(vt.bits(frag), vt.bits(ref)), (cam, fl)


def ray_positions(u, W, H, px, py, k_max):
    """The f32 positions p_0 .. p_k a ray really takes (repeated rounded additions, as the shader does)."""
    hit, s, e, _ = ob.setup_ray(u, W, H, px, py)
    assert hit
    d = (e - s).astype(f32)
    inv = f32(1) / np.sqrt(f32(f32(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]), dtype=f32)
    step = ((d * inv).astype(f32) * f32(u.step_size)).astype(f32)
    p = s.astype(f32).copy()
    out = [p.copy()]
    for _ in range(k_max):
        p = (p + step).astype(f32)
        out.append(p.copy())
    return np.array(out), step


@pytest.mark.parametrize("variant", [capi.BASIC, capi.LIGHT])
def test_provably_inside_prefix_with_4000_step_rays_and_clip_planes_on_a_sample(variant):
    W, H = 144, 96
    v, tf = small_scene(n=12, lit=(variant == capi.LIGHT))
    steps, step_size = 4000, 1.0 / 2300.0
    base = hr.make_uniforms(W, H, steps_count=steps, step_size=step_size, yaw=0.6, pitch=0.35)
    pos, step = ray_positions(base, W, H, W // 2, H // 2, 3000)
    with capi.Context(W, H, 0) as ctx:
        ctx.volume_upload(0, v)
        ctx.tf_upload(0, *tf)
        cases = [dict()]
        # far clip bound on x exactly at / 1-2 ulps around the position the central ray has after 1500 and 2999 steps
        for k in (1500, 2999):
            x = pos[k][0]
            for ulps in (-2, -1, 0, 1, 2):
                t = x
                for _ in range(abs(ulps)):
                    t = np.nextafter(t, f32(2) if ulps > 0 else f32(-2))
                # IsInSampleCoords uses bmax = 1.0f - clip.y (x travels towards 1 when step.x > 0, towards 0 otherwise)
                if step[0] > 0:
                    clip = (0.0, float(f32(1) - f32(t)))
                else:
                    clip = (float(t), 0.0)
                cases.append(dict(clip_x=clip))
        cases.append(dict(clip_y=(0.3333333, 0.25), clip_z=(0.125, 0.4)))
        cases.append(dict(toggles=(1, 0, 0, 0)))     # variable step: step = ray length / 4000
        for kw in cases:
            u = hr.make_uniforms(W, H, steps_count=steps, step_size=step_size, yaw=0.6, pitch=0.35, **kw)
            ctx.set_uniforms(vt.to_capi_uniforms(u))
            ref, n_ref, cov_ref = ob.render(variant, u, [v], [tf], W, H, nthreads=32)
            for fl in (0, 1, 11):
                ctx.set_kernel_flavour(fl)
                ctx.render(variant)
                frag, _, ns = ctx.download()
                assert ns == n_ref and ctx.covered_pixels() == cov_ref, (kw, fl)
                assert np.array_equal(vt.bits(frag), vt.bits(ref)), (kw, fl)
