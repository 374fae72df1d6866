"""ctypes binding of include/vr_mgpu.h (libvr_mgpu.so): the multi-GPU frame loop -- image tiles over the ranks, one
RCCL gather per frame, un-permute on the root.  Plumbing only; the loop itself is C++ (csrc/mgpu/vr_mgpu.cpp)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvr_mgpu.so")
ID_BYTES = 128

ABI_SYMBOLS = ["vr_mgpu_unique_id", "vr_mgpu_create", "vr_mgpu_create_local", "vr_mgpu_destroy", "vr_mgpu_last_error",
               "vr_mgpu_world", "vr_mgpu_local_ranks", "vr_mgpu_context", "vr_mgpu_frame_async", "vr_mgpu_wait",
               "vr_mgpu_frame_device_ptr", "vr_mgpu_download", "vr_mgpu_reduce", "vr_mgpu_backend", "vr_mgpu_frames_async",
               "vr_mgpu_batch_frame_device_ptr", "vr_mgpu_download_batch_frame", "vr_mgpu_comm_count", "vr_mgpu_device",
               "vr_mgpu_set_output", "vr_mgpu_present_device_ptr", "vr_mgpu_download_present", "vr_mgpu_set_stage_timing", "vr_mgpu_set_frames_in_flight",
               "vr_mgpu_stage_times"]

OUT_FRAME, OUT_PRESENT = 1, 2

_lib = None


def bind(path: str) -> C.CDLL:
    """Load a build of csrc/mgpu/vr_mgpu.cpp and declare its prototypes.  load() binds the shipped library; the tests also
    bind a copy linked against their in-process loopback communicator (tests/loopback_comm)."""
    capi.load()
    lib = C.CDLL(path)
    vp, i32, u32 = C.c_void_p, C.c_int, C.c_uint32
    lib.vr_mgpu_unique_id.argtypes = [vp]
    lib.vr_mgpu_create.argtypes = [C.POINTER(vp), vp, i32, i32, vp]
    lib.vr_mgpu_create_local.argtypes = [C.POINTER(vp), u32, u32, C.POINTER(i32), i32]
    lib.vr_mgpu_destroy.argtypes = [vp]
    lib.vr_mgpu_destroy.restype = None
    lib.vr_mgpu_last_error.argtypes = [vp]
    lib.vr_mgpu_last_error.restype = C.c_char_p
    lib.vr_mgpu_world.argtypes = [vp]
    lib.vr_mgpu_local_ranks.argtypes = [vp]
    lib.vr_mgpu_context.argtypes = [vp, i32]
    lib.vr_mgpu_context.restype = vp
    lib.vr_mgpu_frame_async.argtypes = [vp, i32]
    lib.vr_mgpu_wait.argtypes = [vp]
    lib.vr_mgpu_frames_async.argtypes = [vp, i32, i32, C.POINTER(capi.Uniforms)]
    lib.vr_mgpu_batch_frame_device_ptr.argtypes = [vp, i32, i32]
    lib.vr_mgpu_batch_frame_device_ptr.restype = vp
    lib.vr_mgpu_download_batch_frame.argtypes = [vp, i32, i32, vp]
    lib.vr_mgpu_frame_device_ptr.argtypes = [vp, i32]
    lib.vr_mgpu_frame_device_ptr.restype = vp
    lib.vr_mgpu_download.argtypes = [vp, i32, vp]
    lib.vr_mgpu_reduce.argtypes = [vp, C.POINTER(C.c_uint64 * 3), C.c_double, C.POINTER(C.c_double)]
    lib.vr_mgpu_backend.argtypes = [vp]
    lib.vr_mgpu_backend.restype = C.c_char_p
    lib.vr_mgpu_comm_count.argtypes = [vp]
    lib.vr_mgpu_device.argtypes = [vp, i32]
    lib.vr_mgpu_set_output.argtypes = [vp, i32]
    lib.vr_mgpu_present_device_ptr.argtypes = [vp, i32, i32]
    lib.vr_mgpu_present_device_ptr.restype = vp
    lib.vr_mgpu_download_present.argtypes = [vp, i32, i32, vp]
    lib.vr_mgpu_set_stage_timing.argtypes = [vp, i32]
    lib.vr_mgpu_set_frames_in_flight.argtypes = [vp, i32]
    lib.vr_mgpu_stage_times.argtypes = [vp, i32, i32, C.POINTER(C.c_float * 4)]
    return lib


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = bind(LIB_PATH)
    return _lib


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    rc = load().vr_mgpu_unique_id(buf)
    if rc != 0:
        raise capi.VrError(rc, (load().vr_mgpu_last_error(None) or b"").decode())
    return buf.raw


class MultiGpu:
    """One process per GPU: MultiGpu(ctx_handle, rank, world, id).  One process, N GPUs: MultiGpu.local(W, H, devices)."""

    def __init__(self, ctx_handle=None, rank=0, world=1, id128: bytes = b"", _local=None, _lib=None):
        self.lib = _lib if _lib is not None else load()
        self.h = C.c_void_p()
        if _local is not None:
            W, H, devs = _local
            arr = (C.c_int * len(devs))(*devs)
            rc = self.lib.vr_mgpu_create_local(C.byref(self.h), W, H, arr, len(devs))
            self.width, self.height = W, H
        else:
            assert len(id128) == ID_BYTES
            rc = self.lib.vr_mgpu_create(C.byref(self.h), ctx_handle, rank, world, id128)
            self.width = self.height = None
        if rc != 0:
            raise capi.VrError(rc, (self.lib.vr_mgpu_last_error(None) or b"").decode())

    @classmethod
    def local(cls, W, H, devices, _lib=None):
        return cls(_local=(W, H, list(devices)), _lib=_lib)

    def _chk(self, rc):
        if rc < 0:
            raise capi.VrError(rc, (self.lib.vr_mgpu_last_error(self.h) or b"").decode())
        return rc

    def close(self):
        if self.h:
            self.lib.vr_mgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def world(self) -> int:
        return self._chk(self.lib.vr_mgpu_world(self.h))

    def local_ranks(self) -> int:
        return self._chk(self.lib.vr_mgpu_local_ranks(self.h))

    def context(self, local_rank: int) -> "BorrowedContext":
        return BorrowedContext(self.lib.vr_mgpu_context(self.h, local_rank), self.width, self.height)

    def frame_async(self, variant: int) -> int:
        return self._chk(self.lib.vr_mgpu_frame_async(self.h, variant))

    def frames_async(self, variant: int, uniforms) -> int:
        """One launch per rank carrying len(uniforms) frames (1..4); returns the buffer set."""
        n = len(uniforms)
        return self._chk(self.lib.vr_mgpu_frames_async(self.h, variant, n, (capi.Uniforms * n)(*uniforms)))

    def download_batch_frame(self, which: int, frame_in_launch: int, W: int, H: int) -> np.ndarray:
        out = np.empty((H, W, 4), dtype=np.float32)
        self._chk(self.lib.vr_mgpu_download_batch_frame(self.h, which, frame_in_launch, out.ctypes.data))
        return out

    def wait(self):
        self._chk(self.lib.vr_mgpu_wait(self.h))

    def frame_device_ptr(self, which: int) -> int:
        return int(self.lib.vr_mgpu_frame_device_ptr(self.h, which) or 0)

    def download(self, which: int, W: int, H: int) -> np.ndarray:
        out = np.empty((H, W, 4), dtype=np.float32)
        self._chk(self.lib.vr_mgpu_download(self.h, which, out.ctypes.data))
        return out

    def reduce(self, local_value: float = 0.0):
        """(sum over all ranks of the last frame's (composited, covered, fetched), max over ranks of local_value)."""
        c = (C.c_uint64 * 3)()
        mx = C.c_double(0.0)
        self._chk(self.lib.vr_mgpu_reduce(self.h, C.byref(c), local_value, C.byref(mx)))
        return (int(c[0]), int(c[1]), int(c[2])), float(mx.value)

    def backend(self) -> str:
        return (self.lib.vr_mgpu_backend(self.h) or b"").decode()

    def comm_count(self) -> int:
        """ncclCommCount: the ranks RCCL itself sees in the communicator."""
        return self._chk(self.lib.vr_mgpu_comm_count(self.h))

    def device(self, local_rank: int = 0) -> int:
        return self._chk(self.lib.vr_mgpu_device(self.h, local_rank))

    def set_output(self, output: int):
        """OUT_FRAME (assembled float frames), OUT_PRESENT (BGRA8 straight from the gathered tiles), or both."""
        self._chk(self.lib.vr_mgpu_set_output(self.h, output))

    def download_present(self, which: int, frame_in_launch: int, W: int, H: int) -> np.ndarray:
        out = np.empty((H, W, 4), dtype=np.uint8)
        self._chk(self.lib.vr_mgpu_download_present(self.h, which, frame_in_launch, out.ctypes.data))
        return out

    def set_stage_timing(self, enabled: bool):
        self._chk(self.lib.vr_mgpu_set_stage_timing(self.h, 1 if enabled else 0))

    def set_frames_in_flight(self, frames: int):
        """1: one frame at a time on the device (march, gather and output pass of every launch on one stream per rank; the host
        may enqueue ahead); anything else: as many launches in flight as there are buffer sets."""
        self._chk(self.lib.vr_mgpu_set_frames_in_flight(self.h, frames))

    def stage_times(self, local_rank: int, which: int):
        """(march, gather, output, total) ms of the last launch into buffer set `which` on local rank `local_rank`."""
        ms = (C.c_float * 4)()
        self._chk(self.lib.vr_mgpu_stage_times(self.h, local_rank, which, C.byref(ms)))
        return tuple(float(x) for x in ms)


class BorrowedContext(capi.Context):
    """capi.Context view over a vr_ctx owned by the multi-GPU driver (never destroys it)."""

    def __init__(self, handle, width, height):  # noqa: super().__init__ intentionally not called
        self.lib = capi.load()
        self.h = C.c_void_p(handle)
        self.width, self.height = width, height

    def close(self):
        self.h = C.c_void_p()
