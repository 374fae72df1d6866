"""numpy restatement of the reference's HOST code that feeds the hot path (camera block, default TF ramps,
stepping parameters) plus the deterministic synthetic inputs of SURVEY.md section 8d.

TEST INFRASTRUCTURE (same rules as the rest of oracle/).  PARITY UNPINNED: glm is an un-vendored, un-pinned
submodule of the reference (.gitmodules:9-11); its published algorithms (glm::perspective RH/NO, glm::quat from
Euler angles, quat * vec3, mat4_cast, glm::translate) are restated here from the call sites in
App/src/Camera.cpp:92-115,146-177.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

f32 = np.float32


class Uniforms(C.Structure):
    """Same byte layout as vr_uniforms / vro_uniforms."""
    _fields_ = [
        ("model", C.c_float * 16), ("view", C.c_float * 16), ("proj", C.c_float * 16),
        ("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16),
        ("camera_pos", C.c_float * 3),
        ("fragment_mode", C.c_int32), ("steps_count", C.c_int32), ("step_size", C.c_float),
        ("clip_x", C.c_float * 2), ("clip_y", C.c_float * 2), ("clip_z", C.c_float * 2),
        ("toggles", C.c_int32 * 4),
        ("light_pos", C.c_float * 4), ("light_ambient", C.c_float * 4), ("light_diffuse", C.c_float * 4),
    ]


# ----------------------------------------------------------------------------- glm restated (float32)

def quat_from_euler(pitch, yaw, roll=0.0):
    """glm::quat(vec3 eulerAngles) as used by Camera::GetOrientation (Camera.cpp:92-95). Returns (w,x,y,z)."""
    e = np.array([pitch, yaw, roll], dtype=f32) * f32(0.5)
    c, s = np.cos(e).astype(f32), np.sin(e).astype(f32)
    w = c[0] * c[1] * c[2] + s[0] * s[1] * s[2]
    x = s[0] * c[1] * c[2] - c[0] * s[1] * s[2]
    y = c[0] * s[1] * c[2] + s[0] * c[1] * s[2]
    z = c[0] * c[1] * s[2] - s[0] * s[1] * c[2]
    return np.array([w, x, y, z], dtype=f32)


def quat_rotate(q, v):
    """glm operator*(quat, vec3)."""
    qv = q[1:4].astype(f32)
    v = np.asarray(v, dtype=f32)
    uv = np.cross(qv, v).astype(f32)
    uuv = np.cross(qv, uv).astype(f32)
    return (v + ((uv * q[0]) + uuv) * f32(2)).astype(f32)


def normalize(v):
    v = np.asarray(v, dtype=f32)
    return (v / np.sqrt(np.dot(v, v), dtype=f32)).astype(f32)


def mat4_cast(q):
    """glm::toMat4(quat): column-major 4x4 as m[col][row]."""
    w, x, y, z = [f32(t) for t in q]
    m = np.eye(4, dtype=f32)
    qxx, qyy, qzz = x * x, y * y, z * z
    qxz, qxy, qyz = x * z, x * y, y * z
    qwx, qwy, qwz = w * x, w * y, w * z
    m[0][0] = f32(1) - f32(2) * (qyy + qzz)
    m[0][1] = f32(2) * (qxy + qwz)
    m[0][2] = f32(2) * (qxz - qwy)
    m[1][0] = f32(2) * (qxy - qwz)
    m[1][1] = f32(1) - f32(2) * (qxx + qzz)
    m[1][2] = f32(2) * (qyz + qwx)
    m[2][0] = f32(2) * (qxz + qwy)
    m[2][1] = f32(2) * (qyz - qwx)
    m[2][2] = f32(1) - f32(2) * (qxx + qyy)
    return m


def translate(v):
    m = np.eye(4, dtype=f32)
    m[3][0:3] = np.asarray(v, dtype=f32)
    return m


def matmul_cm(a, b):
    """(a*b) for column-major m[col][row] storage."""
    return (b.astype(np.float64) @ a.astype(np.float64)).astype(f32)


def inverse_cm(m):
    return np.linalg.inv(m.astype(np.float64)).astype(f32)


def perspective(fovy, aspect, near, far):
    """glm::perspective, right-handed, depth -1..1 (Camera.cpp:168)."""
    t = f32(math.tan(float(f32(fovy)) / 2.0))
    m = np.zeros((4, 4), dtype=f32)
    m[0][0] = f32(1) / (f32(aspect) * t)
    m[1][1] = f32(1) / t
    m[2][2] = -(f32(far) + f32(near)) / (f32(far) - f32(near))
    m[2][3] = -f32(1)
    m[3][2] = -(f32(2) * f32(far) * f32(near)) / (f32(far) - f32(near))
    return m


class Camera:
    """Orbit camera of App/src/Camera.{h,cpp} (perspective branch)."""

    def __init__(self, fov, aspect, near=0.01, far=100.0):
        self.fov, self.aspect, self.near, self.far = fov, aspect, near, far
        self.pitch, self.yaw, self.distance = 0.0, 0.0, 5.0  # Camera.h:61-64
        self.position = np.zeros(3, dtype=f32)

    def orientation(self):
        return quat_from_euler(self.pitch, self.yaw, 0.0)

    def forward(self):  # Camera.cpp:102-105
        return normalize(quat_rotate(self.orientation(), [0, 0, -1]))

    def get_position(self):  # Camera.cpp:112-115
        return (self.position - self.forward() * f32(self.distance)).astype(f32)

    def matrices(self):
        """(view, proj, view_inv, proj_inv), column-major m[col][row]; Camera.cpp:154-177."""
        q = self.orientation()
        transform = matmul_cm(matmul_cm(translate(-self.forward() * f32(self.distance)), translate(self.position)),
                              mat4_cast(q))
        view = inverse_cm(transform)
        proj = perspective(self.fov, self.aspect, self.near, self.far)
        return view, proj, transform, inverse_cm(proj)


def stepping_params(nx, ny, nz):
    """MiniApp::ComputeRecommendedSteppingParams (MiniApp.h:46-54)."""
    mx = max(nx, ny, nz)
    return f32(1.0) / f32(mx), int(math.sqrt(3) * mx)


def make_uniforms(W, H, *, distance=1.2, yaw=0.6, pitch=0.35, steps_count=200, step_size=0.01, fragment_mode=0,
                  clip_x=(0, 0), clip_y=(0, 0), clip_z=(0, 0), toggles=(0, 0, 0, 0),
                  light_pos=(0.0, 5.0, 0.0, 1.0), light_ambient=(0.1,) * 4, light_diffuse=(1.0,) * 4,
                  fov_deg=60.0):
    cam = Camera(math.radians(fov_deg), W / H)
    cam.distance, cam.yaw, cam.pitch = distance, yaw, pitch
    view, proj, view_inv, proj_inv = cam.matrices()
    u = Uniforms()
    u.model[:] = np.eye(4, dtype=f32).reshape(-1).tolist()
    u.view[:] = view.reshape(-1).tolist()
    u.proj[:] = proj.reshape(-1).tolist()
    u.view_inv[:] = view_inv.reshape(-1).tolist()
    u.proj_inv[:] = proj_inv.reshape(-1).tolist()
    u.camera_pos[:] = cam.get_position().tolist()
    u.fragment_mode, u.steps_count, u.step_size = fragment_mode, steps_count, float(step_size)
    u.clip_x[:] = clip_x
    u.clip_y[:] = clip_y
    u.clip_z[:] = clip_z
    u.toggles[:] = toggles
    u.light_pos[:] = light_pos
    u.light_ambient[:] = light_ambient
    u.light_diffuse[:] = light_diffuse
    return u


# ----------------------------------------------------------------------------- transfer functions

def default_opacity_tf(res):
    """OpacityTF::ResetTF: LinearInterpolation::Generate<float>(0, R-1, 0, 1, 1) (OpacityTf.cpp:29-45)."""
    slope = (f32(1.0) - f32(0.0)) * (f32(1.0) / f32(res - 1))
    i = np.arange(res, dtype=np.int32).astype(f32)
    return (f32(0.0) + slope * i).astype(f32)


def default_color_tf(res):
    """ColorTF::ResetTF: black -> white ramp, alpha 1 (ColorTf.cpp:27-42)."""
    o = default_opacity_tf(res)
    return np.stack([o, o, o, np.ones_like(o)], axis=1).astype(f32)


def thin_opacity_tf(res, top=0.002):
    """Control points (0,0),(R-1,top): no ray terminates."""
    slope = (f32(top) - f32(0.0)) * (f32(1.0) / f32(res - 1))
    i = np.arange(res, dtype=np.int32).astype(f32)
    return (f32(0.0) + slope * i).astype(f32)


# ----------------------------------------------------------------------------- synthetic inputs
# single definition lives in the package (they are inputs, not algorithm); re-exported for the tests
from volumerendering_amd.synth import (ct_phantom_raw, dose_raw, mask_vec4, raw_to_vec4, sphere_raw,  # noqa: E402,F401
                                       xorshift32)
