"""Deterministic synthetic inputs: no DICOM data exists offline, so every test and the
benchmark use these.  numpy definitions for small sizes; `*_fast` variants call the threaded C++ generators of
libvr_host.so (same values, compared in tests/test_synth.py)."""
from __future__ import annotations

import numpy as np

f32 = np.float32


def sphere_raw(n):
    """raw(x,y,z) = round(4095*max(0, 1-|p-c|/(0.45 n))) as uint16, c = (n-1)/2."""
    c = (n - 1) / 2.0
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    r = np.sqrt((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2)
    return np.round(4095.0 * np.maximum(0.0, 1.0 - r / (0.45 * n))).astype(np.uint16)


def xorshift32(x):
    x = x.astype(np.uint32)
    x ^= (x << np.uint32(13))
    x ^= (x >> np.uint32(17))
    x ^= (x << np.uint32(5))
    return x


def ct_phantom_raw(n, z0=0, z1=None, air_noise=False):
    """Nested ellipsoids (air 0 / soft 1000-1100 / bone 2500-3000) + value noise +-40 from
    xorshift32(0x5EED ^ voxel_index); 12-bit uint16.  Returns slices [z0,z1).  air_noise: the air outside the
    body carries the generator's noise as raw 0..80 (scanner-like: no voxel class is an exact constant)."""
    z1 = n if z1 is None else z1
    c = (n - 1) / 2.0
    h = n / 2.0
    zz, yy, xx = np.meshgrid(np.arange(z0, z1), np.arange(n), np.arange(n), indexing="ij")
    qx, qy, qz = (xx - c) / h, (yy - c) / h, (zz - c) / h
    body = (qx / 0.85) ** 2 + (qy / 0.70) ** 2 + (qz / 0.90) ** 2 < 1.0
    outer = (qx / 0.55) ** 2 + (qy / 0.45) ** 2 + (qz / 0.60) ** 2 < 1.0
    inner = (qx / 0.45) ** 2 + (qy / 0.35) ** 2 + (qz / 0.50) ** 2 < 1.0
    val = np.zeros(qx.shape, dtype=np.float64)
    val[body] = (1000.0 + 100.0 * (0.5 + 0.5 * qz))[body]
    shell = outer & ~inner
    val[shell] = (2500.0 + 500.0 * (0.5 + 0.5 * qx))[shell]
    val[inner] = 1040.0
    idx = ((zz.astype(np.uint64) * n + yy.astype(np.uint64)) * n + xx.astype(np.uint64)).astype(np.uint32)
    noise = (xorshift32(np.uint32(0x5EED) ^ idx) % np.uint32(81)).astype(np.int64) - 40
    val = np.where(body, val + noise, (noise + 40.0) if air_noise else 0.0)
    return np.clip(np.round(val), 0, 4095).astype(np.uint16)


def dose_raw(nx=128, ny=128, nz=64):
    """uint32 dose grid: sum of two Gaussians (BitsAllocated = 32 path, DicomReader.cpp:242-249)."""
    z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    qx, qy, qz = x / (nx - 1.0), y / (ny - 1.0), z / (nz - 1.0)
    g1 = np.exp(-(((qx - 0.40) / 0.18) ** 2 + ((qy - 0.50) / 0.20) ** 2 + ((qz - 0.50) / 0.25) ** 2))
    g2 = np.exp(-(((qx - 0.65) / 0.12) ** 2 + ((qy - 0.45) / 0.15) ** 2 + ((qz - 0.55) / 0.20) ** 2))
    return np.round(60000.0 * (g1 + 0.7 * g2)).astype(np.uint32)


def mask_vec4(n):
    """n^3 vec4 mask: r = ellipsoid A, g = ellipsoid B, values exactly 0/1 (StructureFileDcm.cpp:93,172)."""
    c = (n - 1) / 2.0
    h = n / 2.0
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    qx, qy, qz = (x - c) / h, (y - c) / h, (z - c) / h
    a = ((qx - 0.15) / 0.25) ** 2 + ((qy + 0.05) / 0.20) ** 2 + (qz / 0.30) ** 2 < 1.0
    b = ((qx + 0.25) / 0.15) ** 2 + ((qy - 0.10) / 0.15) ** 2 + ((qz + 0.1) / 0.20) ** 2 < 1.0
    m = np.zeros((n, n, n, 4), dtype=f32)
    m[..., 0] = a
    m[..., 1] = b
    return m


def raw_to_vec4(raw):
    """DicomReader::ReadData broadcast: raw integer -> all four lanes (DicomReader.cpp:239,247)."""
    v = raw.astype(f32)
    return np.repeat(v[..., None], 4, axis=3)


def _hostlib():
    import ctypes as C
    from . import host
    lib = host.load()
    lib.vrh_synth_ct_phantom.argtypes = [C.c_int, C.c_void_p]
    lib.vrh_synth_ct_phantom_air.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.vrh_synth_ct_phantom_air.restype = None
    lib.vrh_synth_sphere.argtypes = [C.c_int, C.c_void_p]
    lib.vrh_synth_mask.argtypes = [C.c_int, C.c_void_p]
    for f in (lib.vrh_synth_ct_phantom, lib.vrh_synth_sphere, lib.vrh_synth_mask):
        f.restype = None
    return lib


def ct_phantom_raw_fast(n, air_noise=False):
    out = np.empty((n, n, n), dtype=np.uint16)
    _hostlib().vrh_synth_ct_phantom_air(n, int(air_noise), out.ctypes.data)
    return out


def sphere_raw_fast(n):
    out = np.empty((n, n, n), dtype=np.uint16)
    _hostlib().vrh_synth_sphere(n, out.ctypes.data)
    return out


def mask_vec4_fast(n):
    out = np.empty((n, n, n, 4), dtype=np.float32)
    _hostlib().vrh_synth_mask(n, out.ctypes.data)
    return out
