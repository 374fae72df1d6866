#!/usr/bin/env python3
"""Experiment: how much of a C3 frame is transfer-function look-up latency?  Same volume, frame and camera with the
default ramp tables at several resolutions (a 256-entry pair is 5 KiB and stays in L1, the scene's 4096-entry pair is
80 KiB and does not)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import host_ref as hr
    from volumerendering_amd import capi, synth
    n, W, H = 512, 1920, 1080
    ctx = capi.Context(W, H)
    ctx.volume_upload_raw(0, synth.ct_phantom_raw_fast(n))
    ctx.volume_normalize(0)
    ctx.volume_precompute_gradient(0)
    step, count = hr.stepping_params(n, n, n)
    ctx.set_uniforms(capi.Uniforms.from_buffer_copy(bytes(hr.make_uniforms(W, H, steps_count=count, step_size=step))))
    for fl in (6, 10):
        ctx.set_kernel_flavour(fl)
        for R in (256, 1024, 4096, 16384):
            o, c = hr.default_opacity_tf(R), hr.default_color_tf(R)
            ctx.tf_upload(0, o, c)
            for _ in range(3):
                ctx.render(capi.LIGHT)
            ctx.reset_kernel_times()
            for _ in range(15):
                ctx.render(capi.LIGHT)
            comp, cov, fetched = ctx.counters()
            print(f"flavour {fl} TF resolution {R:6d}: kernel {float(np.median(ctx.kernel_times())):.4f} ms, fetched {fetched}", flush=True)


if __name__ == "__main__":
    main()
