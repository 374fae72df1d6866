#!/usr/bin/env python3
"""Generates tests/golden/frames.npz: small frames of every shader variant / debug mode / toggle produced by the CPU
ORACLE on deterministic synthetic inputs (the reference itself ships no fixtures and cannot run here: parity
unpinned).  The fixtures pin the oracle against accidental change (CPU test) and give the HIP
path a second, file-based target (GPU test).  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import host_ref as hr  # noqa: E402
import oracle_binding as ob  # noqa: E402
import vrtest as vt  # noqa: E402

W, H, N = 40, 32, 16


def cases():
    """(name, variant, uniforms kwargs, thin)"""
    step, count = hr.stepping_params(N, N, N)
    base = dict(steps_count=count, step_size=float(step))
    out = []
    for v in range(6):
        out.append((f"variant{v}", v, dict(base), False))
    out.append(("light_thin", 1, dict(base, yaw=-0.7, pitch=0.5), True))
    out.append(("light_clip_varstep", 1, dict(base, clip_x=(0.2, 0.1), clip_z=(0.1, 0.0), toggles=(1, 0, 0, 0)), False))
    out.append(("basic_jitter", 0, dict(base, toggles=(0, 1, 0, 0)), False))
    for m in (1, 2, 3, 4):
        out.append((f"mode{m}", 1, dict(base, fragment_mode=m), False))
    # the illustrative shader, rays kept shorter than 1 in texture space (beyond that its pow(0, 0) puts NaNs -- whose
    # payload bits are platform specific -- into the frame; tests/test_parity_gpu.py covers those NaN-aware)
    out.append(("illustrative", 6, dict(base, steps_count=12, yaw=0.9), False))
    # the lit shader with its own ComputeGradient (BasicVolLightApp.wgsl:212 enabled), fixed and variable step
    out.append(("inshader", 7, dict(base), False))
    out.append(("inshader_varstep_thin", 7, dict(base, toggles=(1, 0, 0, 0), yaw=2.1, pitch=-0.4), True))
    return out


def render_case(variant, kw, thin):
    vols, tfs = vt.scene(variant, n=N, thin=thin)
    u = hr.make_uniforms(W, H, **kw)
    frag, n, cov = ob.render(variant, u, vols, tfs, W, H)
    return u, vols, tfs, frag, n, cov


if __name__ == "__main__":
    data = {}
    for name, variant, kw, thin in cases():
        u, _, _, frag, n, cov = render_case(variant, kw, thin)
        data[name + "_frag"] = frag
        data[name + "_counts"] = np.array([n, cov], dtype=np.int64)
        data[name + "_uniforms"] = np.frombuffer(bytes(u), dtype=np.uint8)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames.npz"), **data)
    print("wrote", len(data) // 3, "cases")
