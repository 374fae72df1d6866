"""ctypes binding of the C ABI in include/vr.h (libvr_hip.so).

Plumbing only: the product is the HIP library.  There is no CPU fallback -- if the shared library is missing
or no HIP device can be opened, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvr_hip.so")

VR_OK = 0
VR_ERR_INVALID_ARG = -1
VR_ERR_HIP = -2
VR_ERR_NOT_READY = -3
VR_ERR_UNSUPPORTED = -4
VR_ERR_OOM = -5

BASIC, LIGHT, VOLUME_MASK, THREE_FILES, MULTI_CTRT, TF_CALIB, ILLUSTRATIVE, LIGHT_INSHADER = range(8)
VARIANT_NAMES = ["BASIC", "LIGHT", "VOLUME_MASK", "THREE_FILES", "MULTI_CTRT", "TF_CALIB", "ILLUSTRATIVE", "LIGHT_INSHADER"]
TILE = 64
ARITH_SEPARATE, ARITH_FUSED = 0, 1

# every symbol include/vr.h declares (tests check that the library exports each of them)
ABI_SYMBOLS = [
    "vr_create", "vr_resize", "vr_destroy", "vr_last_error", "vr_abi_version", "vr_volume_upload",
    "vr_volume_upload_device", "vr_volume_upload_raw16", "vr_volume_upload_raw32", "vr_volume_normalize",
    "vr_volume_precompute_gradient", "vr_volume_download", "vr_tf_upload", "vr_tf_upload_opacity", "vr_tf_upload_color", "vr_set_uniforms", "vr_render", "vr_render_tiles", "vr_tile_count",
    "vr_render_async", "vr_render_tiles_async", "vr_unpack_tiles_async", "vr_download", "vr_download_tiles",
    "vr_render_batch_async", "vr_render_tiles_batch_async", "vr_unpack_tiles_strided_async",
    "vr_last_timing", "vr_kernel_times", "vr_reset_kernel_times", "vr_frame_device_ptr", "vr_last_covered_pixels", "vr_last_counters", "vr_set_kernel_flavour", "vr_last_block_trace", "vr_last_kernel_flavour",
    "vr_set_volume_layout", "vr_volume_layout", "vr_viewport", "vr_set_arithmetic", "vr_present_async", "vr_stream", "vr_hint_frames_in_flight",
    "vr_set_kernel_timing", "vr_present_tiles_async", "vr_last_split_packets", "vr_experimental_flavours", "vr_kernel_choice",
    "vr_present_packed_async", "vr_unpack_tiles_bgra8_async",
]


class Uniforms(C.Structure):
    """struct vr_uniforms (include/vr.h)."""
    _fields_ = [
        ("model", C.c_float * 16), ("view", C.c_float * 16), ("proj", C.c_float * 16),
        ("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16),
        ("camera_pos", C.c_float * 3),
        ("fragment_mode", C.c_int32), ("steps_count", C.c_int32), ("step_size", C.c_float),
        ("clip_x", C.c_float * 2), ("clip_y", C.c_float * 2), ("clip_z", C.c_float * 2),
        ("toggles", C.c_int32 * 4),
        ("light_pos", C.c_float * 4), ("light_ambient", C.c_float * 4), ("light_diffuse", C.c_float * 4),
    ]


def experimental_flavours() -> bool:
    """True if libvr_hip.so was built with -DVR_EXPERIMENTAL_FLAVOURS=1 (flavours 2, 3, 4, 5, 9 and layout 2 compiled in)."""
    return bool(load().vr_experimental_flavours())


class VrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"vr error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load libvr_hip.so; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, u32, u16 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint16
    fp = C.POINTER(C.c_float)
    lib.vr_create.argtypes = [C.POINTER(vp), u32, u32, i32]
    lib.vr_resize.argtypes = [vp, u32, u32]
    lib.vr_destroy.argtypes = [vp]
    lib.vr_destroy.restype = None
    lib.vr_last_error.argtypes = [vp]
    lib.vr_last_error.restype = C.c_char_p
    lib.vr_abi_version.argtypes = []
    lib.vr_volume_upload.argtypes = [vp, i32, vp, u16, u16, u16]
    lib.vr_volume_upload_device.argtypes = [vp, i32, vp, u16, u16, u16]
    lib.vr_volume_upload_raw16.argtypes = [vp, i32, vp, u16, u16, u16]
    lib.vr_volume_upload_raw32.argtypes = [vp, i32, vp, u16, u16, u16]
    lib.vr_volume_normalize.argtypes = [vp, i32, i32, C.POINTER(C.c_int)]
    lib.vr_volume_precompute_gradient.argtypes = [vp, i32, i32]
    lib.vr_volume_download.argtypes = [vp, i32, vp]
    lib.vr_tf_upload.argtypes = [vp, i32, vp, vp, u32]
    lib.vr_tf_upload_opacity.argtypes = [vp, i32, vp, u32]
    lib.vr_tf_upload_color.argtypes = [vp, i32, vp, u32]
    lib.vr_set_uniforms.argtypes = [vp, C.POINTER(Uniforms)]
    lib.vr_render.argtypes = [vp, i32]
    lib.vr_render_tiles.argtypes = [vp, i32, i32, i32]
    lib.vr_tile_count.argtypes = [vp, i32, i32]
    lib.vr_render_async.argtypes = [vp, i32, vp, vp]
    lib.vr_render_tiles_async.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.vr_unpack_tiles_async.argtypes = [vp, vp, i32, vp, vp]
    lib.vr_render_batch_async.argtypes = [vp, i32, i32, C.POINTER(Uniforms), C.POINTER(vp), vp]
    lib.vr_render_tiles_batch_async.argtypes = [vp, i32, i32, i32, i32, C.POINTER(Uniforms), C.POINTER(vp), vp]
    lib.vr_unpack_tiles_strided_async.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.vr_download.argtypes = [vp, vp, vp, C.POINTER(C.c_uint64)]
    lib.vr_download_tiles.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
    lib.vr_last_timing.argtypes = [vp, fp, fp]
    lib.vr_kernel_times.argtypes = [vp, vp, i32]
    lib.vr_reset_kernel_times.argtypes = [vp]
    lib.vr_set_kernel_timing.argtypes = [vp, i32]
    lib.vr_frame_device_ptr.argtypes = [vp]
    lib.vr_frame_device_ptr.restype = vp
    lib.vr_last_covered_pixels.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.vr_last_counters.argtypes = [vp, C.POINTER(C.c_uint64 * 3)]
    lib.vr_last_block_trace.argtypes = [vp, C.c_void_p, C.c_int]
    lib.vr_set_kernel_flavour.argtypes = [vp, i32]
    lib.vr_last_kernel_flavour.argtypes = [vp]
    lib.vr_last_split_packets.argtypes = [vp]
    lib.vr_experimental_flavours.argtypes = []
    lib.vr_set_volume_layout.argtypes = [vp, i32]
    lib.vr_set_arithmetic.argtypes = [vp, i32]
    lib.vr_present_async.argtypes = [vp, vp, vp, vp]
    lib.vr_present_tiles_async.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.vr_hint_frames_in_flight.argtypes = [vp, i32]
    lib.vr_present_packed_async.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.vr_unpack_tiles_bgra8_async.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.vr_kernel_choice.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.vr_stream.argtypes = [vp, i32]
    lib.vr_stream.restype = vp
    lib.vr_volume_layout.argtypes = [vp, i32, C.POINTER(C.c_int)]
    _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


class Context:
    """Thin RAII wrapper over vr_ctx."""

    def __init__(self, width: int, height: int, device_id: int = 0):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.vr_create(C.byref(self.h), width, height, device_id)
        if rc != VR_OK:
            raise VrError(rc, (self.lib.vr_last_error(None) or b"").decode())
        self.width, self.height = width, height

    def _chk(self, rc: int):
        if rc < 0:
            raise VrError(rc, (self.lib.vr_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if self.h:
            self.lib.vr_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def resize(self, w: int, h: int):
        self._chk(self.lib.vr_resize(self.h, w, h))
        self.width, self.height = w, h

    def volume_upload(self, slot: int, vec4: np.ndarray):
        """vec4: float32 array of shape (nz, ny, nx, 4)."""
        v = _f32(vec4)
        assert v.ndim == 4 and v.shape[3] == 4, v.shape
        nz, ny, nx = v.shape[:3]
        self._chk(self.lib.vr_volume_upload(self.h, slot, v.ctypes.data, nx, ny, nz))

    def volume_upload_device(self, slot: int, dptr: int, nx: int, ny: int, nz: int):
        self._chk(self.lib.vr_volume_upload_device(self.h, slot, dptr, nx, ny, nz))

    def volume_upload_raw(self, slot: int, raw: np.ndarray):
        """raw: uint16 or uint32 array (nz, ny, nx); broadcast to vec4 on the device."""
        raw = np.ascontiguousarray(raw)
        nz, ny, nx = raw.shape
        fn = {np.dtype(np.uint16): self.lib.vr_volume_upload_raw16, np.dtype(np.uint32): self.lib.vr_volume_upload_raw32}[raw.dtype]
        self._chk(fn(self.h, slot, raw.ctypes.data, nx, ny, nz))
        self._shapes = getattr(self, "_shapes", {})
        self._shapes[slot] = (nz, ny, nx)

    def volume_normalize(self, slot: int, value: int = 0) -> int:
        used = C.c_int(0)
        self._chk(self.lib.vr_volume_normalize(self.h, slot, value, C.byref(used)))
        return int(used.value)

    def volume_precompute_gradient(self, slot: int, norm_to_zero_one: bool = False):
        self._chk(self.lib.vr_volume_precompute_gradient(self.h, slot, int(norm_to_zero_one)))

    def volume_download(self, slot: int, shape) -> np.ndarray:
        out = np.empty(tuple(shape) + (4,), dtype=np.float32)
        self._chk(self.lib.vr_volume_download(self.h, slot, out.ctypes.data))
        return out

    def tf_upload(self, slot: int, opacity: np.ndarray, color_rgba: np.ndarray):
        o, c = _f32(opacity), _f32(color_rgba)
        if c.size == 4 * o.size:
            self._chk(self.lib.vr_tf_upload(self.h, slot, o.ctypes.data, c.ctypes.data, o.size))
        else:  # the two textures of a pair may differ in resolution
            self._chk(self.lib.vr_tf_upload_opacity(self.h, slot, o.ctypes.data, o.size))
            self._chk(self.lib.vr_tf_upload_color(self.h, slot, c.ctypes.data, c.size // 4))

    def set_uniforms(self, u: Uniforms):
        self._chk(self.lib.vr_set_uniforms(self.h, C.byref(u)))

    def render(self, variant: int):
        self._chk(self.lib.vr_render(self.h, variant))

    def render_tiles(self, variant: int, rank: int, world: int):
        self._chk(self.lib.vr_render_tiles(self.h, variant, rank, world))

    def tile_count(self, rank: int, world: int) -> int:
        return self._chk(self.lib.vr_tile_count(self.h, rank, world))

    def render_async(self, variant: int, d_frame: int = 0, stream: int = 0):
        self._chk(self.lib.vr_render_async(self.h, variant, d_frame, stream))

    def render_tiles_async(self, variant: int, rank: int, world: int, d_tiles: int, stream: int = 0):
        self._chk(self.lib.vr_render_tiles_async(self.h, variant, rank, world, d_tiles, stream))

    def unpack_tiles_async(self, d_gathered: int, world: int, d_frame: int = 0, stream: int = 0):
        self._chk(self.lib.vr_unpack_tiles_async(self.h, d_gathered, world, d_frame, stream))

    def unpack_tiles_strided_async(self, d_gathered: int, world: int, rank_stride_tiles: int, d_frame: int = 0, stream: int = 0):
        self._chk(self.lib.vr_unpack_tiles_strided_async(self.h, d_gathered, world, rank_stride_tiles, d_frame, stream))

    @staticmethod
    def _batch_args(uniforms, buffers):
        n = len(uniforms)
        assert n == len(buffers)
        return n, (Uniforms * n)(*uniforms), (C.c_void_p * n)(*buffers)

    def render_batch_async(self, variant: int, uniforms, d_frames, stream: int = 0):
        """One launch, len(uniforms) frames (1..4) of the bound scene: frame f with uniforms[f] into d_frames[f]."""
        n, us, bufs = self._batch_args(uniforms, d_frames)
        self._chk(self.lib.vr_render_batch_async(self.h, variant, n, us, bufs, stream))

    def render_tiles_batch_async(self, variant: int, rank: int, world: int, uniforms, d_tiles, stream: int = 0):
        n, us, bufs = self._batch_args(uniforms, d_tiles)
        self._chk(self.lib.vr_render_tiles_batch_async(self.h, variant, rank, world, n, us, bufs, stream))

    def download(self, present: bool = False):
        """Returns (frag[H,W,4] float32, bgra8[H,W,4] uint8 or None, composited_samples)."""
        frag = np.empty((self.height, self.width, 4), dtype=np.float32)
        bgra = np.empty((self.height, self.width, 4), dtype=np.uint8) if present else None
        n = C.c_uint64(0)
        self._chk(self.lib.vr_download(self.h, frag.ctypes.data, bgra.ctypes.data if present else None, C.byref(n)))
        return frag, bgra, int(n.value)

    def samples(self) -> int:
        n = C.c_uint64(0)
        self._chk(self.lib.vr_download(self.h, None, None, C.byref(n)))
        return int(n.value)

    def download_tiles(self, n_tiles: int):
        tiles = np.empty((n_tiles, TILE, TILE, 4), dtype=np.float32)
        n = C.c_uint64(0)
        self._chk(self.lib.vr_download_tiles(self.h, tiles.ctypes.data, C.byref(n)))
        return tiles, int(n.value)

    def last_timing(self):
        k, t = C.c_float(0), C.c_float(0)
        self._chk(self.lib.vr_last_timing(self.h, C.byref(k), C.byref(t)))
        return float(k.value), float(t.value)

    def kernel_times(self, capacity: int = 256) -> np.ndarray:
        out = np.zeros(capacity, dtype=np.float32)
        n = self._chk(self.lib.vr_kernel_times(self.h, out.ctypes.data, capacity))
        return out[:n].copy()

    def reset_kernel_times(self):
        self._chk(self.lib.vr_reset_kernel_times(self.h))

    def set_kernel_timing(self, events: bool):
        """vr_kernel_times from HIP events around every launch (True) or from the launches' own records (False, the default)."""
        self._chk(self.lib.vr_set_kernel_timing(self.h, 1 if events else 0))

    def covered_pixels(self) -> int:
        n = C.c_uint64(0)
        self._chk(self.lib.vr_last_covered_pixels(self.h, C.byref(n)))
        return int(n.value)

    def counters(self):
        """(composited samples, covered pixels, samples actually fetched) of the last render."""
        out = (C.c_uint64 * 3)()
        self._chk(self.lib.vr_last_counters(self.h, C.byref(out)))
        return int(out[0]), int(out[1]), int(out[2])

    def block_trace(self):
        """(n, 6) uint64 array, one row per workgroup of the last march launch: composited, covered, fetched,
        start, end (100 MHz device clock), HW_ID | XCC_ID << 32."""
        n = self.lib.vr_last_block_trace(self.h, None, 0)
        if n < 0:
            self._chk(n)
        out = np.zeros((max(n, 0), 6), dtype=np.uint64)
        if n > 0:
            self._chk(min(0, self.lib.vr_last_block_trace(self.h, out.ctypes.data_as(C.c_void_p), n)))
        return out

    def frame_device_ptr(self) -> int:
        return int(self.lib.vr_frame_device_ptr(self.h) or 0)

    def last_split_packets(self) -> int:
        """Flavour 14: packets the last launch marched as two half packets with two lanes per ray."""
        return self._chk(self.lib.vr_last_split_packets(self.h))

    def last_kernel_flavour(self) -> int:
        return self._chk(self.lib.vr_last_kernel_flavour(self.h))

    def kernel_choice(self):
        """(candidate flavours, ms per launch measured for each, index of the one kept or -1) of the default's measured choice."""
        fl, ms, ch = (C.c_int * 6)(), (C.c_float * 6)(), C.c_int(-1)
        n = self.lib.vr_kernel_choice(self.h, fl, ms, C.byref(ch))
        if n < 0:
            self._chk(n)
        return [int(x) for x in fl[:n]], [float(x) for x in ms[:n]], int(ch.value)

    def hint_frames_in_flight(self, frames: int):
        """How many frames the caller keeps in flight on different streams (steers the default kernel choice only)."""
        self._chk(self.lib.vr_hint_frames_in_flight(self.h, frames))

    def stream(self, index: int) -> int:
        """Context-owned stream `index` (0..3) of a set probed to run side by side: use them in turn for frames in flight."""
        s = self.lib.vr_stream(self.h, index)
        if not s:
            raise VrError(VR_ERR_HIP, "vr_stream: no stream available")
        return int(s)

    def present_async(self, d_bgra8: int, d_frame: int = 0, stream: int = 0):
        """BGRA8Unorm present of a device frame into device memory (what a GL / Vulkan interop buffer would be)."""
        self._chk(self.lib.vr_present_async(self.h, d_frame, d_bgra8, stream))

    def present_tiles_async(self, d_gathered: int, world: int, d_bgra8: int, rank_stride_tiles: int = 0, stream: int = 0):
        """vr_present_tiles_async: BGRA8 frame straight from gathered tile-major segments (device pointers)."""
        self._chk(self.lib.vr_present_tiles_async(self.h, d_gathered, world, rank_stride_tiles, d_bgra8, stream))

    def set_arithmetic(self, mode: int):
        """ARITH_SEPARATE (0, default) or ARITH_FUSED (1): per-sample a * b + c with two roundings or one (include/vr.h)."""
        self._chk(self.lib.vr_set_arithmetic(self.h, mode))

    def set_volume_layout(self, mode: int):
        """0 density plane for .a fetches (default), 1 the reference's vec4 voxels only, 2 = 0 + lit gradients on the fly."""
        self._chk(self.lib.vr_set_volume_layout(self.h, mode))

    def volume_layout(self, slot: int) -> int:
        """bit 0 density plane present, bit 1 .rgb verified as central difference of .a, bit 2 last render derived gradients."""
        f = C.c_int(0)
        self._chk(self.lib.vr_volume_layout(self.h, slot, C.byref(f)))
        return int(f.value)

    def set_kernel_flavour(self, flavour: int):
        self._chk(self.lib.vr_set_kernel_flavour(self.h, flavour))
