"""Seeded random sweep: shader x kernel form x arithmetic mode x volume layout x camera x clips x stepping x table shape
on small volumes, every frame bit-exact against the oracle (counts included).  The point is breadth: combinations nobody
wrote a dedicated test for (the accumulated-rounding bug of the in-box prefix was of that kind)."""
import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi

pytestmark = pytest.mark.gpu
f32 = np.float32


def random_case(rng):
    variant = int(rng.integers(0, 8))
    n = int(rng.choice([5, 9, 16, 23]))
    W, H = int(rng.integers(17, 150)), int(rng.integers(17, 110))
    vols, tfs = vt.scene(variant, n=n, tf_res=int(rng.choice([16, 64, 257])), thin=bool(rng.integers(0, 2)))
    if rng.random() < 0.4:  # a zero prefix of random length on the CT table
        z = int(rng.integers(1, len(tfs[0][0])))
        o = tfs[0][0].copy()
        o[:z] = 0
        tfs[0] = (o, tfs[0][1])
    steps = int(rng.choice([0, 1, 7, int(np.sqrt(3) * n), 3 * n, 900]))
    kw = dict(steps_count=steps, step_size=float(rng.choice([1.0 / n, 0.37 / n, 1.0 / 900])),
              distance=float(rng.choice([0.27, 0.5, 0.8, 1.2, 3.0])), yaw=float(rng.uniform(-3.2, 3.2)),
              pitch=float(rng.uniform(-1.5, 1.5)), toggles=(int(rng.integers(0, 2)), int(rng.integers(0, 2)), 0, 0))
    if rng.random() < 0.5:
        kw.update(clip_x=(float(rng.uniform(0, 0.4)), float(rng.uniform(0, 0.4))), clip_y=(float(rng.uniform(0, 0.3)), 0.0),
                  clip_z=(0.0, float(rng.uniform(0, 0.45))))
    if rng.random() < 0.2:
        kw.update(fragment_mode=int(rng.integers(1, 5)))
    return variant, W, H, vols, tfs, kw


@pytest.mark.parametrize("seed", range(6))
def test_random_combinations_bit_exact(seed):
    rng = np.random.default_rng(1000 + seed)
    with capi.Context(32, 32, 0) as ctx:
        for _ in range(25):
            variant, W, H, vols, tfs, kw = random_case(rng)
            u = hr.make_uniforms(W, H, **kw)
            fused = bool(rng.integers(0, 2))
            flavour = int(rng.choice([0, 0, 1, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17]))
            layout = int(rng.choice([0, 0, 1, 2, 3]))
            if not vt.experimental():  # (the shipped build has no flavours 4 / 5 / 9 / 14 / 15 and no layout 2: csrc/vr_launch.h)
                flavour = {4: 12, 5: 13, 9: 16, 14: 17, 15: 11}.get(flavour, flavour)
                layout = 3 if layout == 2 else layout
            ctx.resize(W, H)
            ctx.set_arithmetic(capi.ARITH_FUSED if fused else capi.ARITH_SEPARATE)
            ctx.set_kernel_flavour(flavour)
            ctx.set_volume_layout(layout)
            frag, _, ns = vt.gpu_render(ctx, variant, u, vols, tfs)
            with ob.arithmetic(ob.FUSED if fused else ob.SEPARATE):
                ref, n_ref, cov_ref = ob.render(variant, u, vols, tfs, W, H, nthreads=8)
            what = (seed, variant, W, H, kw, fused, flavour, layout)
            fin = np.isfinite(ref)
            assert np.array_equal(np.isnan(frag), np.isnan(ref)), what
            assert np.array_equal(vt.bits(frag)[fin], vt.bits(ref)[fin]), what
            assert ns == n_ref and ctx.covered_pixels() == cov_ref, what
