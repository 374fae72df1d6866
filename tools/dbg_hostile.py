import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import host_ref as hr, vrtest as vt, oracle_binding as ob
from volumerendering_amd import capi
f32 = np.float32
W, H, n = 64, 48, 16
rng = np.random.default_rng(3)
v = np.zeros((n, n, n, 4), dtype=f32)
v[4:12, 4:12, 4:12, 3] = rng.random((8, 8, 8), dtype=f32) * f32(0.5)
v[..., :3] = rng.standard_normal((n, n, n, 3)).astype(f32)
v[2, 2, 2, 3] = -0.25
v[13, 3, 3, 3] = np.inf
v[3, 13, 13, 3] = np.nan
v[10, 13, 2, 3] = -np.inf
v[12, 2, 12, 3] = 3.0e38
step, count = hr.stepping_params(n, n, n)
u = hr.make_uniforms(W, H, steps_count=count, step_size=step)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as T
tf = T.zero_prefix_tf(32, 3)
which = sys.argv[1:] or ["all"]
for drop in ["none", "neg", "inf", "nan", "ninf", "big"]:
    vv = v.copy()
    if drop == "neg": vv[2, 2, 2, 3] = 0
    if drop == "inf": vv[13, 3, 3, 3] = 0
    if drop == "nan": vv[3, 13, 13, 3] = 0
    if drop == "ninf": vv[10, 13, 2, 3] = 0
    if drop == "big": vv[12, 2, 12, 3] = 0
    ref, n_ref, _ = ob.render(capi.BASIC, u, [vv], [tf], W, H, nthreads=8)
    with capi.Context(W, H, 0) as ctx:
        for fl in (0, 1, 5):
            ctx.set_kernel_flavour(fl)
            frag, _, ns = vt.gpu_render(ctx, capi.BASIC, u, [vv], [tf])
            ok = T.same(frag, ref)
            fin = np.isfinite(ref)
            bad = (vt.bits(frag) != vt.bits(ref)) & fin
            badnan = np.isnan(frag) != np.isnan(ref)
            print(f"without {drop:5s} flavour {fl}: same {ok} samples {ns} vs {n_ref}; finite mismatches {int(bad.any(axis=2).sum())} nan-pattern mismatches {int(badnan.any(axis=2).sum())}")
            if not ok and fl == 0:
                ys, xs = np.nonzero((bad | badnan).any(axis=2))
                for y, x in list(zip(ys, xs))[:4]:
                    print("   px", x, y, "gpu", frag[y, x], "ref", ref[y, x])
