"""ctypes binding of the CPU oracle (oracle/libvr_oracle.so).

TEST INFRASTRUCTURE: importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED: the reference ships no fixtures and cannot run here (see vr_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")

BASIC, LIGHT, VOLUME_MASK, THREE_FILES, MULTI_CTRT, TF_CALIB, ILLUSTRATIVE, LIGHT_INSHADER = range(8)


class Volume(C.Structure):
    _fields_ = [("vec4", C.c_void_p), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32)]


class TF(C.Structure):
    _fields_ = [("opacity", C.c_void_p), ("color_rgba", C.c_void_p), ("res", C.c_int32), ("res_color", C.c_int32)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    try:  # make is a no-op when the library is newer than its sources
        build()
    except Exception:
        if not os.path.exists(LIB_PATH):
            raise
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    u64p = C.POINTER(C.c_uint64)
    lib.vro_setup_ray.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp]
    lib.vro_shade_pixel.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, vp, C.POINTER(C.c_int)]
    lib.vro_shade_pixel.restype = C.c_uint32
    lib.vro_render.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, u64p, u64p]
    lib.vro_render_pixels.argtypes = [i32, vp, vp, vp, i32, i32, vp, i32, i32, vp, u64p]
    lib.vro_present.argtypes = [vp, i32, vp]
    lib.vro_present.restype = None
    lib.vro_normalize_data.argtypes = [vp, C.c_int64, i32]
    lib.vro_normalize_data.restype = None
    lib.vro_precompute_gradient.argtypes = [vp, i32, i32, i32, i32]
    lib.vro_precompute_gradient.restype = None
    lib.vro_lerp_float.argtypes = [i32, i32, C.c_float, C.c_float, vp]
    lib.vro_lerp_float.restype = None
    lib.vro_lerp_vec4.argtypes = [i32, i32, vp, vp, vp]
    lib.vro_lerp_vec4.restype = None
    lib.vro_jitter.argtypes = [C.c_float, C.c_float]
    lib.vro_jitter.restype = C.c_float
    lib.vro_pow.argtypes = [C.c_float, C.c_float]
    lib.vro_pow.restype = C.c_float
    lib.vro_set_arithmetic.argtypes = [C.c_int]
    lib.vro_set_arithmetic.restype = None
    lib.vro_get_arithmetic.restype = C.c_int
    _lib = lib
    return lib


def _pack(volumes, tfs):
    """volumes: list of float32 arrays (nz,ny,nx,4); tfs: list of (opacity[R], color[R,4])."""
    keep = []
    vols = (Volume * 3)()
    for i, v in enumerate(volumes):
        if v is None:
            continue
        v = np.ascontiguousarray(v, dtype=np.float32)
        keep.append(v)
        vols[i] = Volume(v.ctypes.data, v.shape[2], v.shape[1], v.shape[0])
    tf_arr = (TF * 2)()
    for i, t in enumerate(tfs):
        if t is None:
            continue
        o = np.ascontiguousarray(t[0], dtype=np.float32)
        c = np.ascontiguousarray(t[1], dtype=np.float32)
        keep += [o, c]
        tf_arr[i] = TF(o.ctypes.data, c.ctypes.data, o.size, c.size // 4)
    return vols, tf_arr, keep


def render(variant, uniforms, volumes, tfs, W, H, y0=0, y1=None, nthreads=1):
    """Returns (frag[H,W,4], composited_samples, covered_pixels). `uniforms` is any ctypes struct with the
    vr_uniforms layout (passed by address)."""
    lib = load()
    vols, tf_arr, keep = _pack(volumes, tfs)
    frag = np.zeros((H, W, 4), dtype=np.float32)
    n, cov = C.c_uint64(0), C.c_uint64(0)
    rc = lib.vro_render(variant, C.addressof(uniforms), C.addressof(vols), C.addressof(tf_arr), W, H, y0,
                        H if y1 is None else y1, nthreads, frag.ctypes.data, C.byref(n), C.byref(cov))
    assert rc == 0
    del keep
    return frag, int(n.value), int(cov.value)


def render_pixels(variant, uniforms, volumes, tfs, W, H, pxy, nthreads=1):
    lib = load()
    vols, tf_arr, keep = _pack(volumes, tfs)
    pxy = np.ascontiguousarray(pxy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((pxy.shape[0], 4), dtype=np.float32)
    n = C.c_uint64(0)
    rc = lib.vro_render_pixels(variant, C.addressof(uniforms), C.addressof(vols), C.addressof(tf_arr), W, H,
                               pxy.ctypes.data, pxy.shape[0], nthreads, out.ctypes.data, C.byref(n))
    assert rc == 0
    del keep
    return out, int(n.value)


def setup_ray(uniforms, W, H, px, py):
    lib = load()
    s, e, w = (np.zeros(3, np.float32) for _ in range(3))
    hit = lib.vro_setup_ray(C.addressof(uniforms), W, H, px, py, s.ctypes.data, e.ctypes.data, w.ctypes.data)
    return bool(hit), s, e, w


def present(frag):
    lib = load()
    f = np.ascontiguousarray(frag, dtype=np.float32)
    n = f.size // 4
    out = np.empty(f.shape[:-1] + (4,), dtype=np.uint8)
    lib.vro_present(f.ctypes.data, n, out.ctypes.data)
    return out


def normalize_data(vec4, normalization_value=0):
    lib = load()
    v = np.ascontiguousarray(vec4, dtype=np.float32).copy()
    lib.vro_normalize_data(v.ctypes.data, v.size // 4, normalization_value)
    return v


def precompute_gradient(vec4, norm_to_zero_one=False):
    lib = load()
    v = np.ascontiguousarray(vec4, dtype=np.float32).copy()
    nz, ny, nx = v.shape[:3]
    lib.vro_precompute_gradient(v.ctypes.data, nx, ny, nz, int(norm_to_zero_one))
    return v


def lerp_float(x0, x1, fx0, fx1):
    lib = load()
    out = np.empty(x1 - x0 + 1, dtype=np.float32)
    lib.vro_lerp_float(x0, x1, fx0, fx1, out.ctypes.data)
    return out


def lerp_vec4(x0, x1, fx0, fx1):
    lib = load()
    a = np.ascontiguousarray(fx0, dtype=np.float32)
    b = np.ascontiguousarray(fx1, dtype=np.float32)
    out = np.empty((x1 - x0 + 1, 4), dtype=np.float32)
    lib.vro_lerp_vec4(x0, x1, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def jitter(x, y):
    return float(load().vro_jitter(x, y))


SEPARATE, FUSED = 0, 1


def set_arithmetic(mode: int):
    """0 = product and sum of every a*b+c rounded separately (default); 1 = the per-sample ones fused (vr_oracle.c header)."""
    load().vro_set_arithmetic(int(mode))


class arithmetic:
    """with ob.arithmetic(ob.FUSED): ...  -- restores the previous mode on exit."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = int(load().vro_get_arithmetic())
        set_arithmetic(self.mode)

    def __exit__(self, *a):
        set_arithmetic(self.prev)


def pow_rep(x, y):
    """The oracle's reproducible WGSL pow (exp2(y * log2(x)) through f64)."""
    return float(load().vro_pow(x, y))
