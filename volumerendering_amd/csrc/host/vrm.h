// vrm.h -- the few vector / matrix / quaternion operations the host surface needs, in the conventions of the
// library the reference uses for them (glm: column-major mat4, right-handed, perspective depth -1..1).
// glm is an un-vendored, un-pinned submodule of the reference (.gitmodules:9-11); these are restatements of
// its published algorithms at the call sites App/src/Camera.cpp:92-115,146-177.
#pragma once
#include <cmath>

namespace vrm {

struct vec2 { float x = 0, y = 0; };
struct dvec2 { double x = 0, y = 0; };
struct ivec4 { int x = 0, y = 0, z = 0, w = 0; };

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / length(a)); }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }

struct vec4 {
    union { float x; float r; };
    union { float y; float g; };
    union { float z; float b; };
    union { float w; float a; };
    vec4() : x(0), y(0), z(0), w(0) {}
    vec4(float X, float Y, float Z, float W) : x(X), y(Y), z(Z), w(W) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};
static_assert(sizeof(vec4) == 16, "vec4 must be 4 packed floats (the RGBA32Float texel)");
inline vec4 operator+(vec4 a, vec4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator-(vec4 a, vec4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline vec4 operator*(vec4 a, float s) { return {a.x * s, a.y * s, a.z * s, a.w * s}; }

// column-major: c[col][row], same memory order as glm::mat4 / the WGSL mat4x4<f32>
struct mat4 {
    float c[4][4];
    explicit mat4(float d = 1.0f)
    {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) c[i][j] = (i == j) ? d : 0.0f;
    }
    const float* data() const { return &c[0][0]; }
};

inline mat4 operator*(const mat4& a, const mat4& b)
{
    mat4 r(0.0f);
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r.c[col][row] = a.c[0][row] * b.c[col][0] + a.c[1][row] * b.c[col][1] + a.c[2][row] * b.c[col][2] +
                            a.c[3][row] * b.c[col][3];
    return r;
}

inline mat4 translate(const mat4& m, vec3 v)
{
    mat4 r = m;
    for (int row = 0; row < 4; ++row)
        r.c[3][row] = m.c[0][row] * v.x + m.c[1][row] * v.y + m.c[2][row] * v.z + m.c[3][row];
    return r;
}

// perspective, right-handed, clip-space depth -1..1 (what glm::perspective gives with default defines)
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar)
{
    const float t = std::tan(fovy / 2.0f);
    mat4 r(0.0f);
    r.c[0][0] = 1.0f / (aspect * t);
    r.c[1][1] = 1.0f / t;
    r.c[2][2] = -(zFar + zNear) / (zFar - zNear);
    r.c[2][3] = -1.0f;
    r.c[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}

inline mat4 ortho(float l, float r_, float b, float t, float zNear, float zFar)
{
    mat4 r(1.0f);
    r.c[0][0] = 2.0f / (r_ - l);
    r.c[1][1] = 2.0f / (t - b);
    r.c[2][2] = -2.0f / (zFar - zNear);
    r.c[3][0] = -(r_ + l) / (r_ - l);
    r.c[3][1] = -(t + b) / (t - b);
    r.c[3][2] = -(zFar + zNear) / (zFar - zNear);
    return r;
}

// general 4x4 inverse by cofactors
inline mat4 inverse(const mat4& m)
{
    const float* a = m.data();
    float inv[16];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    float id = 1.0f / det;
    mat4 r(0.0f);
    for (int i = 0; i < 16; ++i) (&r.c[0][0])[i] = inv[i] * id;
    return r;
}

struct quat {
    float w = 1, x = 0, y = 0, z = 0;
};

// quaternion from Euler angles (pitch about x, yaw about y, roll about z)
inline quat quat_from_euler(vec3 e)
{
    vec3 c(std::cos(e.x * 0.5f), std::cos(e.y * 0.5f), std::cos(e.z * 0.5f));
    vec3 s(std::sin(e.x * 0.5f), std::sin(e.y * 0.5f), std::sin(e.z * 0.5f));
    quat q;
    q.w = c.x * c.y * c.z + s.x * s.y * s.z;
    q.x = s.x * c.y * c.z - c.x * s.y * s.z;
    q.y = c.x * s.y * c.z + s.x * c.y * s.z;
    q.z = c.x * c.y * s.z - s.x * s.y * c.z;
    return q;
}

inline vec3 rotate(quat q, vec3 v)
{
    vec3 qv(q.x, q.y, q.z);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}

inline mat4 to_mat4(quat q)
{
    mat4 r(1.0f);
    float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
    float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
    float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
    r.c[0][0] = 1.0f - 2.0f * (qyy + qzz);
    r.c[0][1] = 2.0f * (qxy + qwz);
    r.c[0][2] = 2.0f * (qxz - qwy);
    r.c[1][0] = 2.0f * (qxy - qwz);
    r.c[1][1] = 1.0f - 2.0f * (qxx + qzz);
    r.c[1][2] = 2.0f * (qyz + qwx);
    r.c[2][0] = 2.0f * (qxz + qwy);
    r.c[2][1] = 2.0f * (qyz - qwx);
    r.c[2][2] = 1.0f - 2.0f * (qxx + qyy);
    return r;
}

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
template <typename T>
inline T clamp(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace vrm
