"""End-to-end through the C++ host surface (Application + MiniApp scenes -> C ABI -> HIP kernels), checked
bit for bit against the oracle fed with the very same uniforms, voxels and tables the host objects produced."""
import numpy as np
import pytest

import host_ref as hr
import oracle_binding as ob
import vrtest as vt
from volumerendering_amd import capi, host, synth

pytestmark = pytest.mark.gpu
f32 = np.float32


def scene_inputs(variant, n):
    ct = host.VolumeFile.from_raw(synth.ct_phantom_raw(n))
    if variant in (capi.BASIC, capi.LIGHT, capi.LIGHT_INSHADER):
        return [ct]
    dose = host.VolumeFile.from_raw(synth.dose_raw(24, 20, 12))
    mask = host.VolumeFile.from_vec4(synth.mask_vec4(n), 1)
    if variant == capi.VOLUME_MASK:
        return [mask, dose, ct]
    if variant == capi.THREE_FILES:
        return [ct, dose, mask]
    if variant in (capi.MULTI_CTRT, capi.ILLUSTRATIVE):
        return [ct, dose]
    unfilled = synth.mask_vec4(n).copy()
    unfilled[1:-1, 1:-1, 1:-1, 0] *= 0  # keep only a shell, as a contour mask without FILL would
    return [ct, mask, host.VolumeFile.from_vec4(unfilled, 1)]


@pytest.mark.parametrize("variant", range(8))
def test_scene_through_host_surface(variant):
    W, H, n = 112, 72, 20
    vols = scene_inputs(variant, n)
    with host.Application(W, H, 0) as app:
        app.OnStart(variant, vols)
        cam = app.camera()
        cam.SetOrbit(0.35, 0.6, 1.2)
        if variant == capi.THREE_FILES:
            # the reference leaves this scene on the Application defaults (0.01 / 200) with UN-normalised data
            assert app.stepping() == (200, pytest.approx(0.01))
        elif variant in (capi.MULTI_CTRT, capi.TF_CALIB, capi.ILLUSTRATIVE):
            assert app.stepping()[0] == 200  # these scenes never call ComputeRecommendedSteppingParams
        else:
            assert app.stepping()[0] == int(np.sqrt(3) * n)
        app.OnUpdate()
        app.OnRender()
        frag, bgra, samples = app.ReadFrame(present=True)
        u = hr.Uniforms.from_buffer_copy(bytes(app.uniforms()))
        if variant == capi.TF_CALIB:
            volumes = [vols[0].data(), vols[2].data()]
        else:
            volumes = [v.data() for v in vols]
        tfs = [(app.scene_opacity_tf(0).table(), app.scene_color_tf(0).table())]
        if variant in (capi.VOLUME_MASK, capi.THREE_FILES, capi.MULTI_CTRT, capi.ILLUSTRATIVE):
            tfs.append((app.scene_opacity_tf(1).table(), app.scene_color_tf(1).table()))
        ref, n_ref, _ = ob.render(variant, u, volumes, tfs, W, H, nthreads=8)
        fin = np.isfinite(ref)  # (the illustrative shader's pow(0, 0) is NaN on both sides)
        assert fin.all() or variant == capi.ILLUSTRATIVE
        assert np.array_equal(np.isnan(frag), np.isnan(ref))
        assert np.array_equal(vt.bits(frag)[fin], vt.bits(ref)[fin]), float(np.nanmax(np.abs(frag - ref)))
        assert samples == n_ref and samples > 0
        assert np.array_equal(bgra, ob.present(ref))


def test_tf_edit_reuploads_and_resize():
    W, H, n = 96, 64, 16
    with host.Application(W, H, 0) as app:
        app.OnStart(capi.LIGHT, [host.VolumeFile.from_raw(synth.ct_phantom_raw(n))], tf_res=128)
        app.camera().SetOrbit(0.2, -0.8, 1.0)
        app.OnUpdate(); app.OnRender()
        f0, _, _ = app.ReadFrame()
        otf = app.scene_opacity_tf(0)
        otf.AddControlPoint(40, 0.02)      # edit -> m_ShouldUpdate -> OnUpdate re-uploads (OpacityTf.cpp:134-142)
        app.scene_color_tf(0).AddColorControlPoint(64, (1.0, 0.2, 0.1, 1.0))
        app.OnUpdate(); app.OnRender()
        f1, _, n1 = app.ReadFrame()
        assert not np.array_equal(f0, f1)
        u = hr.Uniforms.from_buffer_copy(bytes(app.uniforms()))
        vol = app._keep[0].data()
        ref, n_ref, _ = ob.render(ob.LIGHT, u, [vol], [(otf.table(), app.scene_color_tf(0).table())], W, H, nthreads=8)
        assert np.array_equal(vt.bits(f1), vt.bits(ref)) and n1 == n_ref
        app.OnResize(50, 40)   # like the reference, the camera aspect is not touched by a resize
        app.OnUpdate(); app.OnRender()
        f2, _, n2 = app.ReadFrame()
        u2 = hr.Uniforms.from_buffer_copy(bytes(app.uniforms()))
        assert bytes(u2.proj) == bytes(u.proj)
        ref2, n_ref2, _ = ob.render(ob.LIGHT, u2, [vol], [(otf.table(), app.scene_color_tf(0).table())], 50, 40, nthreads=8)
        assert np.array_equal(vt.bits(f2), vt.bits(ref2)) and n2 == n_ref2


def test_differing_tf_resolutions_in_one_pair():
    """OpacityTF::Load re-resolves only the opacity texture: the pair may have two resolutions."""
    W, H, n = 64, 48, 16
    vols, _ = vt.scene(capi.LIGHT, n=n)
    tf = (hr.default_opacity_tf(64), hr.default_color_tf(200))
    u = hr.make_uniforms(W, H, steps_count=27, step_size=1 / 16)
    with capi.Context(W, H) as ctx:
        frag, _, ns = vt.gpu_render(ctx, capi.LIGHT, u, vols, [tf])
    ref, n_ref, _ = ob.render(ob.LIGHT, u, vols, [tf], W, H, nthreads=8)
    assert np.array_equal(vt.bits(frag), vt.bits(ref)) and ns == n_ref


@pytest.mark.parametrize("variant", [capi.BASIC, capi.LIGHT])
def test_scene_prepared_on_device_equals_host_prepared(variant):
    """MiniApp::SetPrepareOnDevice: NormalizeData / PreComputeGradient run as GPU kernels; same frame."""
    W, H, n = 96, 64, 24
    frames = []
    for on_device in (False, True):
        with host.Application(W, H, 0) as app:
            app.OnStart(variant, [host.VolumeFile.from_raw(synth.ct_phantom_raw(n))], prepare_on_device=on_device)
            app.camera().SetOrbit(0.35, 0.6, 1.2)
            app.OnUpdate(); app.OnRender()
            frag, _, ns = app.ReadFrame()
            frames.append((frag, ns))
    assert np.array_equal(vt.bits(frames[0][0]), vt.bits(frames[1][0])) and frames[0][1] == frames[1][1] > 0


def test_volume_mask_scene_from_dicom_files(tmp_path):
    """The reference's VolumeMaskApp::OnStart from files (VolumeMaskApp.cpp:12-25): CT series, RTDOSE and RTSTRUCT are
    read from (synthetic) DICOM, the mask comes from Create3DMask with the scene's post-processing options, and the
    three-volume composite is rendered by the HIP path -- bit for bit the oracle's frame on the same arrays."""
    import math

    import dicom_writer as dw
    n, nz = 40, 24
    raw = synth.ct_phantom_raw(n)[:nz]  # (z, y, x)
    ctd, dsd = tmp_path / "ct", tmp_path / "dose"
    ctd.mkdir()
    dsd.mkdir()
    for k in range(nz):
        # the reader fills x from Rows and y from Columns and streams the pixels in file order (DicomReader.cpp:181-182)
        dw.write_slice(str(ctd / f"{k:03d}.dcm"), raw[k], rows=n, cols=n, instance=k + 1, position=(0.0, 0.0, 2.0 * k),
                       spacing=(1.0, 1.0), thickness=2.0, largest=int(raw.max()), frame_uid="1.9.9")
    dose = synth.dose_raw(16, 12, 8)
    dw.write_slice(str(dsd / "dose.dcm"), dose, modality="RTDOSE", rows=16, cols=12, frames=8, bits=32, frame_uid="1.9.9")

    def ring(cx, cy, r, z, m=180):
        return [v for i in range(m) for v in (cx + r * math.cos(2 * math.pi * i / m), cy + r * math.sin(2 * math.pi * i / m), z)]

    contours = [[ring(20, 20, 3, 2.0 * k) for k in range(3, 6)],                 # 1: never selected below
                [ring(20, 19, 9, 2.0 * k) for k in range(6, 18)],                # 2 -> mask.r
                [ring(12, 12, 2, 2.0)],                                          # 3
                [[10.0, 26.0, 2.0 * k, 30.0, 27.5, 2.0 * k, 28.0, 33.0, 2.0 * k, 11.0, 31.0, 2.0 * k, 10.0, 26.0, 2.0 * k]
                 for k in range(8, 14)],                                          # 4 -> mask.g (sparse: needs the lines)
                [ring(5, 5, 1, 0.0)]]                                             # 5: the unreachable last one
    dw.write_rtstruct(str(tmp_path / "rs.dcm"), contours, frame_uid="1.9.9")

    ct = host.VolumeFile.from_dicom(str(ctd))
    rt = host.VolumeFile.from_dicom(str(dsd))
    rs = host.StructureFile.read(str(tmp_path / "rs.dcm"))
    SF = host.StructureFile
    mask = rs.create_3d_mask(ct, [2, 4, 0, 0], SF.RECONSTRUCT_BRESENHAM | SF.PROCESS_NON_DUPLICATES | SF.CLOSING | SF.FILL)
    md = mask.data()
    assert md[10, :, :, 0].sum() > 150 and md[10, :, :, 1].sum() > 60 and md[..., 2:].max() == 0

    W, H = 120, 90
    with host.Application(W, H, 0) as app:
        app.OnStart(capi.VOLUME_MASK, [mask, rt, ct])
        app.camera().SetOrbit(0.3, 0.8, 1.3)
        app.OnUpdate()
        app.OnRender()
        frag, _, samples = app.ReadFrame()
        u = hr.Uniforms.from_buffer_copy(bytes(app.uniforms()))
        tfs = [(app.scene_opacity_tf(i).table(), app.scene_color_tf(i).table()) for i in (0, 1)]
        ref, n_ref, _ = ob.render(capi.VOLUME_MASK, u, [mask.data(), rt.data(), ct.data()], tfs, W, H, nthreads=8)
        assert np.array_equal(vt.bits(frag), vt.bits(ref)), float(np.max(np.abs(frag - ref)))
        assert samples == n_ref and samples > 0
        # the structures are visible: where the mask is set the RT table's colour is blended, so the frame differs from
        # one rendered with an empty mask
        empty = host.VolumeFile.from_vec4(np.zeros_like(md), 1)
        app.OnStart(capi.VOLUME_MASK, [empty, rt, ct])
        app.camera().SetOrbit(0.3, 0.8, 1.3)
        app.OnUpdate()
        app.OnRender()
        frag0, _, _ = app.ReadFrame()
        assert not np.array_equal(frag, frag0)


def test_prepare_on_device_uses_the_files_cached_maximum(tmp_path):
    """DICOM input: GetMaxNumber() is LargestPixelValue, which need not be the data maximum.  The on-device preparation
    must divide by it exactly as NormalizeData() does (not by the maximum found in the uploaded voxels)."""
    import dicom_writer as dw
    W, H, n = 96, 64, 16
    raw = synth.ct_phantom_raw(n)
    d = tmp_path / "series"
    d.mkdir()
    for k in range(n):
        dw.write_slice(str(d / f"s{k:03d}.dcm"), raw[k], rows=n, cols=n, instance=k + 1, position=(0, 0, float(k)), largest=4000)
    frames = []
    for on_device in (False, True):
        vf = host.VolumeFile.from_dicom(str(d))
        assert vf.GetMaxNumber() == 4000 and int(raw.max()) != 4000
        with host.Application(W, H, 0) as app:
            app.OnStart(capi.LIGHT, [vf], tf_res=256, prepare_on_device=on_device)
            app.camera().SetOrbit(0.35, 0.6, 1.2)
            app.OnUpdate()
            app.OnRender()
            frag, _, samples = app.ReadFrame()
            frames.append((frag, samples))
            assert vf.GetDataRange() == 4000
            assert vf.IsNormalized() == (not on_device)  # on-device preparation leaves the host voxels as loaded
    assert frames[0][1] == frames[1][1] and frames[0][1] > 0
    assert np.array_equal(vt.bits(frames[0][0]), vt.bits(frames[1][0]))


def test_scene_start_fails_loudly_when_an_upload_fails():
    """Upload / preparation return codes reach the caller of OnStart (an empty volume cannot be uploaded)."""
    with host.Application(32, 32, 0) as app:
        empty = host.VolumeFile.from_vec4(np.zeros((0, 4, 4, 4), dtype=np.float32), 1)
        with pytest.raises(capi.VrError):
            app.OnStart(capi.BASIC, [empty])
