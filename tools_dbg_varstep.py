import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle"), os.path.join(os.getcwd(), "tests")]
import numpy as np
import host_ref as hr, oracle_binding as ob, vrtest as vt
from volumerendering_amd import capi
import test_scale_gpu as ts
f32 = np.float32
W, H = 144, 96
variant = capi.BASIC
v, tf = ts.small_scene(n=12, lit=False)
steps, step_size = 4000, 1.0 / 2300.0
u = hr.make_uniforms(W, H, steps_count=steps, step_size=step_size, yaw=0.6, pitch=0.35, toggles=(1, 0, 0, 0))
ref, n_ref, cov_ref = ob.render(variant, u, [v], [tf], W, H, nthreads=16)
with capi.Context(W, H, 0) as ctx:
    ctx.volume_upload(0, v); ctx.tf_upload(0, *tf); ctx.set_uniforms(vt.to_capi_uniforms(u))
    for fl in (1, 5, 6, 9, 4, 8, 11, 7, 10, 0):
        ctx.set_kernel_flavour(fl); ctx.render(variant); frag, _, ns = ctx.download()
        bad = np.argwhere((vt.bits(frag) != vt.bits(ref)).any(axis=2))
        print("fl", fl, "resolved", ctx.last_kernel_flavour(), "ns-n_ref", ns - n_ref, "bad px", len(bad), bad[:3].tolist())
    # analyse the first bad pixel of flavour 0
    ctx.set_kernel_flavour(0); ctx.render(variant); frag, _, ns = ctx.download()
    bad = np.argwhere((vt.bits(frag) != vt.bits(ref)).any(axis=2))
    for (py, px) in bad[:4]:
        hit, s, e, _ = ob.setup_ray(u, W, H, int(px), int(py))
        d = (e - s).astype(f32)
        ln = np.sqrt(f32(f32(d[0]*d[0] + d[1]*d[1]) + d[2]*d[2]), dtype=f32)
        inv = f32(1) / ln
        dirv = (d * inv).astype(f32)
        ss = f32(ln / f32(steps))
        st = (dirv * ss).astype(f32)
        p = s.astype(f32).copy(); inside = []
        for i in range(steps):
            inside.append(bool((p >= 0).all() and (p <= 1).all()))
            p = (p + st).astype(f32)
        ins = np.array(inside)
        print("pixel", px, py, "start", s, "end", e, "step", st, "in-box steps", ins.sum(), "first out", np.argmin(ins) if not ins.all() else None,
              "frag", frag[py, px], "ref", ref[py, px])
