"""examples/render_volume.cpp: the C++-only caller (host classes above the C ABI, no Python in the path).  The CPU test
compiles and links it and checks that it fails loudly without a GPU; the GPU test runs BASELINE config C1 through it and
compares its sample count and frame sum with the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "volumerendering_amd")


def build(tmp_path):
    exe = str(tmp_path / "render_volume")
    subprocess.run(["g++", "-std=c++20", "-O2", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc", "host"),
                    os.path.join(ROOT, "examples", "render_volume.cpp"), "-L", PKG, "-lvr_host", "-lvr_hip",
                    "-Wl,-rpath," + PKG, "-o", exe], check=True)
    return exe


def test_example_builds_and_has_no_cpu_fallback(tmp_path):
    import torch
    exe = build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 2 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_example_renders_c1_like_the_oracle(tmp_path):
    import host_ref as hr
    import oracle_binding as ob
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp_path, check=True)
    words = r.stdout.split()
    samples, total = int(words[words.index("composited") - 1]), float(words[-1])
    # the same scene through the oracle, with the uniforms the C++ Application / Camera classes produce (the example's own):
    # sphere-64 -> NormalizeData, TF 256, 1/64 x 110, camera pitch .35 / yaw .6 / distance 1.2
    from volumerendering_amd import capi, host, synth
    ct = host.VolumeFile.from_raw(synth.sphere_raw(64))
    with host.Application(256, 256, 0) as app:
        app.OnStart(capi.BASIC, [ct])
        app.camera().SetOrbit(0.35, 0.6, 1.2)
        app.OnUpdate()
        u = hr.Uniforms.from_buffer_copy(bytes(app.uniforms()))
        tfs = [(app.scene_opacity_tf(0).table(), app.scene_color_tf(0).table())]
        ref, n_ref, _ = ob.render(ob.BASIC, u, [ct.data()], tfs, 256, 256, nthreads=8)
    assert samples == n_ref
    assert total == pytest.approx(float(ref.astype(np.float64).sum()), rel=1e-12)
    ppm = (tmp_path / "frame.ppm").read_bytes()
    assert ppm.startswith(b"P6\n256 256\n255\n") and len(ppm) == 15 + 256 * 256 * 3
    got = np.frombuffer(ppm[15:], dtype=np.uint8).reshape(256, 256, 3)
    assert np.array_equal(got, ob.present(ref)[..., [2, 1, 0]])
