"""The BASELINE.json configurations C1..C5 as scenes of the C++ host surface (harness glue shared by bench.py and the
full-size parity tests, so that what is measured and what is checked are the same scene).

    C1  sphere-64,        256x256,   BasicVolumeApp   (unlit), 1/64  x 110
    C2  ct-phantom-256,   1024x1024, BasicVolumeApp   (unlit), 1/256 x 443
    C3  ct-phantom-512,   1920x1080, BasicVolLightApp (lit),   1/512 x 886      <- the metric's configuration
    C4  C3 + 512^3 mask + 128x128x64 dose, VolumeMaskApp (three-volume composite)
    C5  ct-phantom-1024 (16 GiB), 3840x2160, lit, 1/1024 x 1773 (image tiles over 8 GPUs)

Stepping is MiniApp::ComputeRecommendedSteppingParams (App/src/miniapps/include/MiniApp.h:46-54); data preparation
runs through the scene's own OnStart order (NormalizeData / PreComputeGradient); camera: perspective 60 deg, distance
1.2, yaw .6, pitch .35.

Transfer functions (`tf`):
    default  the reference's ramps (OpacityTf.cpp:29-45, ColorTf.cpp:27-42): opacity[0] is the only exact zero
    thin     control points (0,0),(R-1,0.002): no ray terminates
    prefix   preset style: control points (0,0),(0.12 (R-1),0),(R-1,1) -- a real zero prefix below soft tissue, the shape
             of the reference's bone presets (air and its noise map to opacity 0)
    zero     experiment: opacity identically 0 (pure traversal)
Air (`air`): "exact0" = the phantom's air is raw 0; "noisy" = raw 0..80 noise outside the body, as a scanner delivers.
"""
from __future__ import annotations

import math
import sys
import time

WORKLOADS = {
    # name: (volume N, W, H, variant)
    "C1": (64, 256, 256, "BASIC"),
    "C2": (256, 1024, 1024, "BASIC"),
    "C3": (512, 1920, 1080, "LIGHT"),
    "C4": (512, 1920, 1080, "VOLUME_MASK"),
    "C5": (1024, 3840, 2160, "LIGHT"),
}
TF_KINDS = ("default", "thin", "prefix", "zero")
AIR_KINDS = ("exact0", "noisy")
CAMERA = (0.35, 0.6, 1.2)  # pitch, yaw, distance

# per-composited-sample algorithmic bytes: the f32 footprint of one trilinear cell
BYTES_PER_SAMPLE = {"BASIC": 32, "LIGHT": 128, "VOLUME_MASK": 288, "THREE_FILES": 64, "MULTI_CTRT": 160, "TF_CALIB": 48,
                    "ILLUSTRATIVE": 160, "LIGHT_INSHADER": 224}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def apply_tf(app, vname, tf_kind):
    """Edits the scene's opacity table(s) through the reference's control-point surface."""
    if tf_kind == "default":
        return
    for which in range(2 if vname == "VOLUME_MASK" else 1):
        otf = app.scene_opacity_tf(which)
        R = otf.GetTextureResolution()
        if tf_kind == "thin":
            otf.SetControlPoint(1, R - 1, 0.002)
        elif tf_kind == "zero":
            otf.SetControlPoint(1, R - 1, 0.0)
        elif tf_kind == "prefix":
            otf.AddControlPoint(round(0.12 * (R - 1)), 0.0)
        else:
            raise ValueError(tf_kind)


def build_scene(app, workload, tf_kind="default", air="exact0", vol_n=0, camera=CAMERA, quiet=False):
    """Generates the synthetic inputs, runs the reference's data-prep order through the C++ host classes and starts
    the scene on `app` (uploads happen here, outside any timed region).  Returns (variant, [VolumeFile...])."""
    from . import capi, host, synth

    n, W, H, vname = WORKLOADS[workload]
    full_n = n
    n = vol_n or n
    variant = capi.VARIANT_NAMES.index(vname)
    t0 = time.time()
    raw = synth.sphere_raw_fast(n) if workload == "C1" else synth.ct_phantom_raw_fast(n, air_noise=(air == "noisy"))
    ct = host.VolumeFile.from_raw(raw)
    del raw
    vols = [ct]
    if vname == "VOLUME_MASK":
        mask = host.VolumeFile.from_vec4(synth.mask_vec4_fast(n), 1)
        dose = host.VolumeFile.from_raw(synth.dose_raw())
        vols = [mask, dose, ct]
    app.OnStart(variant, vols)  # NormalizeData / PreComputeGradient in the scene's own order + uploads
    apply_tf(app, vname, tf_kind)
    cam = app.camera()
    cam.SetOrbit(*camera)
    if vol_n:
        app.set_params(steps_count=int(math.sqrt(3) * full_n), step_size=1.0 / full_n)
    app.OnUpdate()
    if not quiet:
        log(f"[scene] {workload} ({vname}, {n}^3, {W}x{H}, tf={tf_kind}, air={air}) ready in {time.time() - t0:.1f}s")
    return variant, vols


def oracle_inputs(app, vols):
    """(uniforms bytes, volume arrays, TF table pairs) of a started scene, in the form the checker takes."""
    volumes = [v.data() for v in vols]
    tfs = []
    for which in range(2 if len(vols) == 3 else 1):
        tfs.append((app.scene_opacity_tf(which).table(), app.scene_color_tf(which).table()))
    return bytes(app.uniforms()), volumes, tfs
