"""ctypes binding of libvr_host.so: the C++ host surface (VolumeFile / OpacityTF / ColorTF / Camera / MiniApp
scenes / Application) that mirrors the reference's classes.  Plumbing for tests and bench.py; the logic lives in
csrc/host/*.cpp."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvr_host.so")
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    capi.load()  # libvr_host.so links against libvr_hip.so
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, f32, f64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float, C.c_double
    sigs = {
        "vrh_volume_from_raw16": (vp, [vp, i32, i32, i32]), "vrh_volume_from_raw32": (vp, [vp, i32, i32, i32]),
        "vrh_volume_from_vec4": (vp, [vp, i32, i32, i32, u64]), "vrh_volume_from_dat": (vp, [C.c_char_p]),
        "vrh_dat_write": (i32, [C.c_char_p, vp, i32, i32, i32]),
        "vrh_volume_from_dicom": (vp, [C.c_char_p, C.c_char_p, i32]), "vrh_dicom_params": (i32, [vp, vp, C.c_char_p, C.c_char_p, i32]),
        "vrh_dicom_transform": (None, [vp, i32, vp, vp]), "vrh_dicom_compare": (i32, [vp, vp, i32]),
        "vrh_dicom_modality": (i32, [C.c_char_p]), "vrh_volume_free": (None, [vp]),
        "vrh_struct_read": (vp, [C.c_char_p]), "vrh_struct_from_contours": (vp, [C.c_char_p, vp, vp, vp, i32]),
        "vrh_struct_free": (None, [vp]), "vrh_struct_contour_count": (i32, [vp]), "vrh_struct_polygon_count": (i32, [vp, i32]),
        "vrh_struct_polygon": (i32, [vp, i32, i32, vp, i32]), "vrh_struct_info": (i32, [vp, C.c_char_p, i32, vp, i32]),
        "vrh_struct_create_mask": (vp, [vp, vp, vp, C.c_uint]),
        "vrh_volume_normalize": (None, [vp, i32]), "vrh_volume_gradient": (None, [vp, i32]),
        "vrh_volume_average_gradient": (None, [vp, i32]), "vrh_volume_data": (vp, [vp]),
        "vrh_volume_max_number": (u64, [vp]), "vrh_volume_data_range": (u64, [vp]),
        "vrh_volume_is_normalized": (i32, [vp]),
        "vrh_volume_index": (i32, [vp, i32, i32, i32]), "vrh_volume_voxel": (None, [vp, i32, i32, i32, vp]),
        "vrh_volume_size": (None, [vp, vp]), "vrh_volume_bbox": (None, [vp, vp]),
        "vrh_set_worker_threads": (None, [C.c_uint]),
        "vrh_otf_create": (vp, [i32]), "vrh_otf_free": (None, [vp]), "vrh_otf_resolution": (i32, [vp]),
        "vrh_otf_data": (vp, [vp]), "vrh_otf_reset": (None, [vp]), "vrh_otf_add_cp": (i32, [vp, f64, f64]),
        "vrh_otf_set_cp": (None, [vp, i32, f64, f64]), "vrh_otf_cp_count": (i32, [vp]), "vrh_otf_cp": (None, [vp, i32, vp]),
        "vrh_otf_set_data_range": (None, [vp, i32]), "vrh_otf_data_range": (i32, [vp]),
        "vrh_otf_save": (i32, [vp, C.c_char_p]), "vrh_otf_load": (None, [vp, C.c_char_p, i32]),
        "vrh_otf_calibrate": (None, [vp, vp, vp, vp]), "vrh_otf_histogram": (None, [vp, vp, vp]),
        "vrh_otf_remap_cp": (None, [vp, f64, f64, i32, i32, vp]),
        "vrh_ctf_create": (vp, [i32]), "vrh_ctf_free": (None, [vp]), "vrh_ctf_resolution": (i32, [vp]),
        "vrh_ctf_data": (vp, [vp]), "vrh_ctf_reset": (None, [vp]), "vrh_ctf_add_cp": (i32, [vp, f64, vp]),
        "vrh_ctf_set_color": (None, [vp, i32, vp]), "vrh_ctf_save": (i32, [vp, C.c_char_p]),
        "vrh_ctf_load": (None, [vp, C.c_char_p]),
        "vrh_camera_create": (vp, [f32, f32, f32, f32]), "vrh_camera_free": (None, [vp]),
        "vrh_camera_set_orbit": (None, [vp, f32, f32, f32]), "vrh_camera_rotate": (None, [vp, f32, f32]),
        "vrh_camera_zoom": (None, [vp, f32]), "vrh_camera_set_position": (None, [vp, f32, f32, f32]),
        "vrh_camera_key": (None, [vp, i32]), "vrh_camera_get": (None, [vp, vp]),
        "vrh_app_create": (vp, [u32, u32, i32]), "vrh_app_free": (None, [vp]), "vrh_app_ok": (i32, [vp]),
        "vrh_app_error": (C.c_char_p, [vp]), "vrh_app_context": (vp, [vp]), "vrh_app_camera": (vp, [vp]),
        "vrh_app_start": (i32, [vp, i32, vp, vp, vp, i32]), "vrh_app_set_prepare_on_device": (None, [vp, i32]), "vrh_app_update": (i32, [vp]), "vrh_app_render": (i32, [vp]),
        "vrh_app_resize": (i32, [vp, u32, u32]), "vrh_app_read_frame": (i32, [vp, vp, vp, C.POINTER(u64)]),
        "vrh_app_set_params": (None, [vp, i32, i32, f32, vp, vp]),
        "vrh_app_get_stepping": (None, [vp, C.POINTER(i32), C.POINTER(f32)]),
        "vrh_app_get_uniforms": (None, [vp, C.POINTER(capi.Uniforms)]),
        "vrh_app_scene_otf": (vp, [vp, i32]), "vrh_app_scene_ctf": (vp, [vp, i32]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class VolumeFile:
    """med::VolumeFile (csrc/host/VolumeFile.h)."""

    def __init__(self, handle):
        self.lib = load()
        self.h = handle

    @classmethod
    def from_raw(cls, raw: np.ndarray):
        """raw: uint16 or uint32 array shaped (nz, ny, nx)."""
        lib = load()
        raw = np.ascontiguousarray(raw)
        nz, ny, nx = raw.shape
        if raw.dtype == np.uint16:
            return cls(lib.vrh_volume_from_raw16(raw.ctypes.data, nx, ny, nz))
        if raw.dtype == np.uint32:
            return cls(lib.vrh_volume_from_raw32(raw.ctypes.data, nx, ny, nz))
        raise TypeError(raw.dtype)

    @classmethod
    def from_vec4(cls, vec4: np.ndarray, max_number: int = 0):
        lib = load()
        v = np.ascontiguousarray(vec4, dtype=np.float32)
        nz, ny, nx = v.shape[:3]
        return cls(lib.vrh_volume_from_vec4(v.ctypes.data, nx, ny, nz, max_number))

    @classmethod
    def from_dat(cls, path: str):
        """med::DatImpl::ReadFile: 6-byte uint16 header (x, y, z) + uint16 voxels."""
        h = load().vrh_volume_from_dat(path.encode())
        if not h:
            raise IOError(f"Check file: {path}")
        return cls(h)

    @classmethod
    def from_dicom(cls, path: str):
        """med::DicomReader::ReadVolumeFile: a directory of .dcm slices or one (multi-frame) .dcm file."""
        err = C.create_string_buffer(512)
        h = load().vrh_volume_from_dicom(path.encode(), err, 512)
        if not h:
            raise IOError(err.value.decode())
        return cls(h)

    def dicom_params(self) -> dict:
        out = (C.c_double * 20)()
        axis, frame = C.create_string_buffer(8), C.create_string_buffer(128)
        if not self.lib.vrh_dicom_params(self.h, out, axis, frame, 128):
            raise ValueError("not a DICOM volume")
        o = list(out)
        return dict(Modality=["UNKNOWN", "CT", "RTSTRUCT", "RTDOSE", "MR", "CONTOURMASK"][int(o[0])], X=int(o[1]), Y=int(o[2]),
                    Z=int(o[3]), BitsStored=int(o[4]), BitsAllocated=int(o[5]), LargestPixelValue=int(o[6]),
                    SmallestPixelValue=int(o[7]), SliceThickness=o[8], ImagePositionPatient=o[9:12],
                    ImageOrientationPatient=o[12:18], PixelSpacing=o[18:20], MainAxis=axis.value.decode(),
                    FrameOfReference=frame.value.decode())

    def dicom_transform(self, which: int, v):
        a = (C.c_float * 3)(*(list(v) + [0.0])[:3])
        o = (C.c_float * 3)()
        self.lib.vrh_dicom_transform(self.h, which, a, o)
        return tuple(o)

    @staticmethod
    def write_dat(path: str, raw: np.ndarray) -> bool:
        raw = np.ascontiguousarray(raw, dtype=np.uint16)
        nz, ny, nx = raw.shape
        return bool(load().vrh_dat_write(path.encode(), raw.ctypes.data, nx, ny, nz))

    def close(self):
        if self.h:
            self.lib.vrh_volume_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def NormalizeData(self, value: int = 0):
        self.lib.vrh_volume_normalize(self.h, value)

    def PreComputeGradient(self, normToZeroOne: bool = False):
        self.lib.vrh_volume_gradient(self.h, int(normToZeroOne))

    def AverageGradient(self, k: int):
        self.lib.vrh_volume_average_gradient(self.h, k)

    def GetSize(self):
        out = (C.c_int * 3)()
        self.lib.vrh_volume_size(self.h, out)
        return tuple(out)

    def GetBBOXSize(self):
        out = (C.c_float * 3)()
        self.lib.vrh_volume_bbox(self.h, out)
        return tuple(out)

    def GetMaxNumber(self) -> int:
        return int(self.lib.vrh_volume_max_number(self.h))

    def GetDataRange(self) -> int:
        return int(self.lib.vrh_volume_data_range(self.h))

    def IsNormalized(self) -> bool:
        return bool(self.lib.vrh_volume_is_normalized(self.h))

    def GetIndexFrom3D(self, x, y, z) -> int:
        return int(self.lib.vrh_volume_index(self.h, x, y, z))

    def GetVoxelData(self, x, y, z):
        out = (C.c_float * 4)()
        self.lib.vrh_volume_voxel(self.h, x, y, z, out)
        return tuple(out)

    def data(self) -> np.ndarray:
        """Zero-copy view of the vec4 voxels, shaped (nz, ny, nx, 4)."""
        nx, ny, nz = self.GetSize()
        ptr = self.lib.vrh_volume_data(self.h)
        buf = (C.c_float * (nx * ny * nz * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float32).reshape(nz, ny, nx, 4)


class StructureFile:
    """med::StructureFileDcm (csrc/host/dicom/StructureFileDcm.h): an RTSTRUCT file's contours + Create3DMask."""
    IGNORE, NEAREST_NEIGHBOUR, RECONSTRUCT_BRESENHAM, CLOSING, FILL, PROCESS_NON_DUPLICATES = 1, 2, 4, 8, 16, 32

    def __init__(self, handle):
        self.lib = load()
        self.h = handle

    @classmethod
    def read(cls, path: str):
        """med::DicomReader::ReadStructFile; None where the reference returns nullptr."""
        h = load().vrh_struct_read(path.encode())
        return cls(h) if h else None

    @classmethod
    def from_contours(cls, contours, frame_of_reference: str = ""):
        """contours[c][k] = flat x y z x y z ... of polygon k of contour c."""
        pts = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.float32).ravel() for c in contours for p in c]
                                                  or [np.zeros(0, np.float32)]), dtype=np.float32)
        sizes = np.asarray([len(np.asarray(p).ravel()) for c in contours for p in c], dtype=np.int32)
        per = np.asarray([len(c) for c in contours], dtype=np.int32)
        return cls(load().vrh_struct_from_contours(frame_of_reference.encode(), pts.ctypes.data, sizes.ctypes.data,
                                                   per.ctypes.data, len(contours)))

    def contours(self):
        out = []
        for c in range(self.lib.vrh_struct_contour_count(self.h)):
            polys = []
            for k in range(self.lib.vrh_struct_polygon_count(self.h, c)):
                n = self.lib.vrh_struct_polygon(self.h, c, k, None, 0)
                a = np.zeros(n, dtype=np.float32)
                self.lib.vrh_struct_polygon(self.h, c, k, a.ctypes.data, n)
                polys.append(a)
            out.append(polys)
        return out

    def info(self) -> dict:
        text = C.create_string_buffer(1 << 16)
        colors = np.zeros((256, 3), dtype=np.float32)
        n = self.lib.vrh_struct_info(self.h, text, 1 << 16, colors.ctypes.data, 256)
        lines = text.value.decode().split("\n")
        rois = [dict(zip(("Number", "Name", "AlgorithmType"), l.split("\t"))) for l in lines[3:] if l]
        for r in rois:
            r["Number"] = int(r["Number"])
        return dict(Label=lines[0], Name=lines[1], FrameOfReference=lines[2], StructureSetROISequence=rois,
                    DisplayColors=colors[:max(n, 0)].copy())

    def create_3d_mask(self, reference: "VolumeFile", contour_ids, post_process: int):
        """Create3DMask(other, contourIDs[4], postProcess) -> VolumeFile (x, y, z, w = the selected contours) or None."""
        ids = (C.c_int * 4)(*(list(contour_ids) + [0, 0, 0, 0])[:4])
        h = self.lib.vrh_struct_create_mask(self.h, reference.h, ids, post_process)
        return VolumeFile(h) if h else None

    def close(self):
        if self.h:
            self.lib.vrh_struct_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _TF:
    def __init__(self, handle, owned, prefix, width):
        self.lib, self.h, self.owned, self.p, self.width = load(), handle, owned, prefix, width

    def _f(self, name):
        return getattr(self.lib, f"vrh_{self.p}_{name}")

    def close(self):
        if self.h and self.owned:
            self._f("free")(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def GetTextureResolution(self) -> int:
        return int(self._f("resolution")(self.h))

    def ResetTF(self):
        self._f("reset")(self.h)

    def table(self) -> np.ndarray:
        r = self.GetTextureResolution()
        buf = (C.c_float * (r * self.width)).from_address(self._f("data")(self.h))
        a = np.frombuffer(buf, dtype=np.float32).copy()
        return a if self.width == 1 else a.reshape(r, 4)

    def Save(self, path: str) -> bool:
        return bool(self._f("save")(self.h, path.encode()))


class OpacityTF(_TF):
    def __init__(self, res: int = 0, handle=None):
        lib = load()
        super().__init__(handle or lib.vrh_otf_create(res), handle is None, "otf", 1)

    def AddControlPoint(self, x, y) -> int:
        return int(self.lib.vrh_otf_add_cp(self.h, x, y))

    def SetControlPoint(self, i, x, y):
        self.lib.vrh_otf_set_cp(self.h, i, x, y)

    def GetControlPoints(self):
        out = []
        for i in range(self.lib.vrh_otf_cp_count(self.h)):
            cp = (C.c_double * 2)()
            self.lib.vrh_otf_cp(self.h, i, cp)
            out.append((cp[0], cp[1]))
        return out

    def SetDataRange(self, r: int):
        self.lib.vrh_otf_set_data_range(self.h, r)

    def GetDataRange(self) -> int:
        return int(self.lib.vrh_otf_data_range(self.h))

    def Load(self, path: str, rescale: bool = False):
        self.lib.vrh_otf_load(self.h, path.encode(), int(rescale))

    def CalibrateOnMask(self, mask: VolumeFile, file: VolumeFile, active=(1, 0, 0, 0)):
        self.lib.vrh_otf_calibrate(self.h, mask.h, file.h, (C.c_int * 4)(*active))

    def ActivateHistogram(self, file: VolumeFile) -> np.ndarray:
        out = np.zeros(self.GetTextureResolution(), dtype=np.float32)
        self.lib.vrh_otf_histogram(self.h, file.h, out.ctypes.data)
        return out

    def RemapCP(self, x, y, data_range, tf_res):
        out = (C.c_double * 2)()
        self.lib.vrh_otf_remap_cp(self.h, x, y, data_range, tf_res, out)
        return out[0], out[1]


class ColorTF(_TF):
    def __init__(self, res: int = 0, handle=None):
        lib = load()
        super().__init__(handle or lib.vrh_ctf_create(res), handle is None, "ctf", 4)

    def AddColorControlPoint(self, x, rgba) -> int:
        return int(self.lib.vrh_ctf_add_cp(self.h, x, (C.c_float * 4)(*rgba)))

    def SetControlColor(self, i, rgba):
        self.lib.vrh_ctf_set_color(self.h, i, (C.c_float * 4)(*rgba))

    def Load(self, path: str):
        self.lib.vrh_ctf_load(self.h, path.encode())


class Camera:
    """med::Camera (csrc/host/Camera.h)."""

    def __init__(self, fov, aspect, near=0.01, far=100.0, handle=None):
        self.lib = load()
        self.owned = handle is None
        self.h = handle or self.lib.vrh_camera_create(fov, aspect, near, far)

    def __del__(self):
        try:
            if self.h and self.owned:
                self.lib.vrh_camera_free(self.h)
        except Exception:
            pass

    def SetOrbit(self, pitch, yaw, distance):
        self.lib.vrh_camera_set_orbit(self.h, pitch, yaw, distance)

    def Rotate(self, dx, dy):
        self.lib.vrh_camera_rotate(self.h, dx, dy)

    def SetZoomDistance(self, delta):
        self.lib.vrh_camera_zoom(self.h, delta)

    def SetPosition(self, x, y, z):
        self.lib.vrh_camera_set_position(self.h, x, y, z)

    def KeyboardEvent(self, key):
        self.lib.vrh_camera_key(self.h, key)

    def get(self):
        """dict(view, proj, view_inv, proj_inv as (4,4) column-major m[col][row]; position; forward)."""
        out = np.zeros(70, dtype=np.float32)
        self.lib.vrh_camera_get(self.h, out.ctypes.data)
        return dict(view=out[0:16].reshape(4, 4), proj=out[16:32].reshape(4, 4), view_inv=out[32:48].reshape(4, 4),
                    proj_inv=out[48:64].reshape(4, 4), position=out[64:67].copy(), forward=out[67:70].copy())


class Application:
    """med::Application (csrc/host/Application.h): camera + uniforms + active scene + render call."""

    def __init__(self, width=1280, height=720, device=0):
        self.lib = load()
        self.h = self.lib.vrh_app_create(width, height, device)
        self.width, self.height = width, height
        if not self.h or not self.lib.vrh_app_ok(self.h):
            msg = (self.lib.vrh_app_error(self.h) or b"").decode() if self.h else "allocation failed"
            if self.h:
                self.lib.vrh_app_free(self.h)
                self.h = None
            raise capi.VrError(capi.VR_ERR_HIP, msg)
        self._keep = []

    def close(self):
        if self.h:
            self.lib.vrh_app_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise capi.VrError(rc, (self.lib.vrh_app_error(self.h) or b"").decode())

    def camera(self) -> Camera:
        return Camera(0, 0, handle=self.lib.vrh_app_camera(self.h))

    def OnStart(self, variant: int, volumes, tf_res: int = 0, prepare_on_device: bool = False):
        self._keep = list(volumes)
        self.lib.vrh_app_set_prepare_on_device(self.h, int(prepare_on_device))
        hs = [v.h if v is not None else None for v in volumes] + [None, None, None]
        self._chk(self.lib.vrh_app_start(self.h, variant, hs[0], hs[1], hs[2], tf_res))

    def set_params(self, fragment_mode=0, steps_count=-1, step_size=-1.0, clips=None, toggles=None):
        c = (C.c_float * 6)(*clips) if clips is not None else None
        t = (C.c_int * 4)(*toggles) if toggles is not None else None
        self.lib.vrh_app_set_params(self.h, fragment_mode, steps_count, step_size, c, t)

    def stepping(self):
        n, s = C.c_int(0), C.c_float(0)
        self.lib.vrh_app_get_stepping(self.h, C.byref(n), C.byref(s))
        return int(n.value), float(s.value)

    def OnUpdate(self):
        self._chk(self.lib.vrh_app_update(self.h))

    def OnRender(self):
        self._chk(self.lib.vrh_app_render(self.h))

    def OnResize(self, w, h):
        self._chk(self.lib.vrh_app_resize(self.h, w, h))
        self.width, self.height = w, h

    def uniforms(self) -> capi.Uniforms:
        u = capi.Uniforms()
        self.lib.vrh_app_get_uniforms(self.h, C.byref(u))
        return u

    def ReadFrame(self, present=False):
        frag = np.empty((self.height, self.width, 4), dtype=np.float32)
        bgra = np.empty((self.height, self.width, 4), dtype=np.uint8) if present else None
        n = C.c_uint64(0)
        self._chk(self.lib.vrh_app_read_frame(self.h, frag.ctypes.data, bgra.ctypes.data if present else None, C.byref(n)))
        return frag, bgra, int(n.value)

    def context(self) -> "RawContext":
        return RawContext(self.lib.vrh_app_context(self.h), self.width, self.height)

    def scene_opacity_tf(self, which=0) -> OpacityTF:
        return OpacityTF(handle=self.lib.vrh_app_scene_otf(self.h, which))

    def scene_color_tf(self, which=0) -> ColorTF:
        return ColorTF(handle=self.lib.vrh_app_scene_ctf(self.h, which))


class RawContext(capi.Context):
    """A capi.Context view over a vr_ctx owned by an Application (never destroys it)."""

    def __init__(self, handle, width, height):  # noqa: super().__init__ intentionally not called
        self.lib = capi.load()
        self.h = C.c_void_p(handle)
        self.width, self.height = width, height

    def close(self):
        self.h = C.c_void_p()
