// vr_kernels.h -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): the front-to-back compositing loop
// of the reference's WGSL fragment shaders, one ray per lane, one 8x8 pixel packet per 64-wide wavefront
// (march_kernel; vr_dp.h holds the two / four lanes per ray form, vr_wtb.h the LDS wave-tile experiment), plus the
// auxiliary kernels (brick records and distance field, counters, present, tile unpack, data preparation).
//
// What each piece replaces (paths below the reference root):
//   setup_ray()      rayCoords.wgsl:19-34 + vertex stage BasicVolumeApp.wgsl:44-57 + rasteriser, proxy box
//                    Application.h:147-156, cull state Application.cpp:589-590,602-604
//   tex3_*()         textureSample(texture_3d, samplerLin/NN, p): Sampler.cpp:9-24 (clamp-to-edge, single mip)
//   tf_*()           textureSample(texture_1d, samplerLin, d)
//   march_kernel<V>  fs_main of BasicVolumeApp.wgsl:113-188, BasicVolLightApp.wgsl:151-237,
//                    VolumeMaskApp.wgsl:128-217, ThreeFilesApp.wgsl:170-272, MultiCTRTApp.wgsl:163-259,
//                    TFCalibrationApp.wgsl:114-197, MutliCTRTIllustrative.wgsl:227-313
//   present_kernel   output merge PipelineBuilder.cpp:142-147 over fullscreen.wgsl:33-41 white, BGRA8Unorm
//
// Arithmetic contract (DESIGN.md "Normative arithmetic"): IEEE f32, no FMA contraction (-ffp-contract=off),
// correctly rounded divide and sqrt (hipcc default for HIP), dot = (x*x' + y*y') + z*z',
// normalize(v) = v * (1/sqrt(dot(v,v))), max(x,0) = x > 0 ? x : 0, lerp(a,b,t) = a + (b-a)*t.
// The loop is exact with respect to the WGSL: iterations that cannot blend (outside IsInSampleCoords, or after
// the opacity cut-off) are skipped, never approximated; p and the world position still advance by repeated
// addition, one step at a time, because start + i*step is not bit-equal to the reference's accumulation.
#pragma once
#include "vr_device.h"

// The kernels are compiled once per arithmetic mode, each in a translation unit and a namespace of its own:
//   VR_KNS = vr,  VR_FUSED = 0   a * b + c rounds the product and the sum separately (the default normative arithmetic)
//   VR_KNS = vrf, VR_FUSED = 1   the per-sample expressions of that shape are single fused multiply-adds (vr_fused.hip)
#ifndef VR_KNS
#define VR_KNS vr
#endif
#ifndef VR_FUSED
#define VR_FUSED 0
#endif
// Experiment knob (build with -DVR_LIGHT_WPE=6): ask the compiler for six waves per SIMD for the lit one-lane kernel (80
// VGPRs instead of 88-90) at the price of a few spilled registers.
#ifndef VR_LIGHT_WPE
#define VR_LIGHT_WPE 1
#endif
#define VR_LIGHT_WAVES_PER_EU(V, OTF) (((V) == 1 && !(OTF)) ? VR_LIGHT_WPE : 1)

namespace VR_KNS {
using namespace vr;

enum Variant : int { V_BASIC = 0, V_LIGHT = 1, V_VOLUME_MASK = 2, V_THREE_FILES = 3, V_MULTI_CTRT = 4, V_TF_CALIB = 5,
                     V_ILLUSTRATIVE = 6, V_LIGHT_INSHADER = 7 };

struct f3 {
    float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
// a * b + c of the PER-SAMPLE expressions: two roundings (VR_FUSED 0, the default normative arithmetic) or one
// (VR_FUSED 1).  Ray placement never goes through this: it is separately rounded in both modes (dot3s / normalize3s).
__device__ __forceinline__ float mad(float a, float b, float c)
{
#if VR_FUSED
    return __builtin_fmaf(a, b, c);
#else
    return a * b + c;
#endif
}
__device__ __forceinline__ float dot3s(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float length3s(f3 a) { return sqrtf(dot3s(a, a)); }
#if VR_FUSED
__device__ __forceinline__ float dot3(f3 a, f3 b) { return mad(a.z, b.z, mad(a.y, b.y, a.x * b.x)); }
#else
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
#endif
__device__ __forceinline__ float length3(f3 a) { return sqrtf(dot3(a, a)); }
// 1.0f / sqrtf(x), both operations correctly rounded (the normative normalize = v * (1 / sqrt(dot))).  For x in
// [2^-60, 2^60) the compiler's IEEE expansions need none of their scaling and special-case steps; what is left is
// written out here (13 instructions instead of 30).  Everything else -- 0, denormals, huge, inf, NaN, negative --
// takes the generic operators.
__device__ __forceinline__ float inv_sqrt_exact(float x)
{
    if (__float_as_uint(x) - 0x21800000u < 0x3c000000u) {
        // v_sqrt_f32 is within 1 ulp: choose among y - 1 ulp, y, y + 1 ulp by the sign of the exact residuals
        float y = __builtin_amdgcn_sqrtf(x);
        const float yd = __int_as_float(__float_as_int(y) - 1), yu = __int_as_float(__float_as_int(y) + 1);
        const float rd = fmaf(-yd, y, x), ru = fmaf(-yu, y, x);
        y = (rd <= 0.0f) ? yd : y;
        y = (ru > 0.0f) ? yu : y;
        // 1 / y: reciprocal estimate, one Newton step, then two residual corrections of the quotient
        float r = __builtin_amdgcn_rcpf(y);
        r = fmaf(fmaf(-y, r, 1.0f), r, r);
        float q = r;
        q = fmaf(fmaf(-y, q, 1.0f), r, q);
        return fmaf(fmaf(-y, q, 1.0f), r, q);
    }
    return 1.0f / sqrtf(x);
}
__device__ __forceinline__ f3 normalize3(f3 a)
{
    float inv = inv_sqrt_exact(dot3(a, a));
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ f3 normalize3s(f3 a)
{
    float inv = inv_sqrt_exact(dot3s(a, a));
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
// The wavefront's lane mask of a condition.  (HIP's __ballot takes an int: the condition becomes 0 / 1 in a register and is
// compared again -- v_cndmask + v_cmp and a scalar wait on the vector compare -- where the builtin takes the condition's own mask.)
__device__ __forceinline__ unsigned long long vr_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
__device__ __forceinline__ float max0(float x) { return (x > 0.0f) ? x : 0.0f; }
__device__ __forceinline__ float lerpf(float a, float b, float t) { return mad(b - a, t, a); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
// Texel pair of a clamp-to-edge linear fetch from the floored coordinate x0 (any float, incl. +-inf / NaN):
// the f32 -> i32 conversion saturates (NaN -> 0, v_cvt_i32_f32), and the "+1" is applied after limiting the
// value from above so that it can never overflow.  i0 = clamp(t, 0, n-1), i1 = clamp(t+1, 0, n-1).
__device__ __forceinline__ void texel_pair(float x0, int n, int& i0, int& i1)
{
    int t = min((int)x0, n - 1);
    i0 = max(t, 0);
    i1 = min(max(t + 1, 0), n - 1);
}

// column-major mat4 * (x,y,z,1), summed left to right
__device__ __forceinline__ void mat4_mul_point(const float* m, float x, float y, float z, float w, float out[4])
{
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

// ------------------------------------------------------------------------------------------------ jitter
// BasicVolumeApp.wgsl:107-110.  sin() is evaluated in f64 (quadrant reduction + Taylor) so that the value is
// reproducible; once per pixel, only when toggles[1] == 1.
__device__ inline double sin_d(double x)
{
    double n = rint(x * 0.63661977236758138);
    double r = (x - n * 1.5707963267948966) - n * 6.123233995736766e-17;
    long long q = ((long long)n) & 3;
    double r2 = r * r;
    double s = r2 * (1.0 / 6227020800.0);
    s = r2 * (s + (-1.0 / 39916800.0));
    s = r2 * (s + (1.0 / 362880.0));
    s = r2 * (s + (-1.0 / 5040.0));
    s = r2 * (s + (1.0 / 120.0));
    s = r2 * (s + (-1.0 / 6.0));
    s = r * (s + 1.0);
    double c = r2 * (-1.0 / 87178291200.0);
    c = r2 * (c + (1.0 / 479001600.0));
    c = r2 * (c + (-1.0 / 3628800.0));
    c = r2 * (c + (1.0 / 40320.0));
    c = r2 * (c + (-1.0 / 720.0));
    c = r2 * (c + (1.0 / 24.0));
    c = r2 * (c + (-0.5));
    c = c + 1.0;
    return q == 0 ? s : (q == 1 ? c : (q == 2 ? -s : -c));
}
__device__ inline float jitter(float x, float y)
{
    float d = x * 12.9898f + y * 78.233f;
    float s = (float)sin_d((double)d);
    float v = s * 43758.5453f;
    return v - floorf(v);
}

// WGSL pow(x, y) = exp2(y * log2(x)) through f64 with a fixed operation sequence (oracle: vro_pow, same sequence):
// log2 by exponent split + atanh series, exp2 by rounding split + Taylor series.  Only the illustrative shader uses it.
__device__ inline double log2_d(double x)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)b);
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double t = s2 * (1.0 / 23.0);
    t = s2 * (t + (1.0 / 21.0));
    t = s2 * (t + (1.0 / 19.0));
    t = s2 * (t + (1.0 / 17.0));
    t = s2 * (t + (1.0 / 15.0));
    t = s2 * (t + (1.0 / 13.0));
    t = s2 * (t + (1.0 / 11.0));
    t = s2 * (t + (1.0 / 9.0));
    t = s2 * (t + (1.0 / 7.0));
    t = s2 * (t + (1.0 / 5.0));
    t = s2 * (t + (1.0 / 3.0));
    double ln = (2.0 * s) * (t + 1.0);
    return (double)e + ln * 1.4426950408889634;
}
__device__ inline double exp2_d(double t)
{
    double n = rint(t);
    double z = (t - n) * 0.6931471805599453;
    double r = z * (1.0 / 87178291200.0);
    r = z * (r + (1.0 / 6227020800.0));
    r = z * (r + (1.0 / 479001600.0));
    r = z * (r + (1.0 / 39916800.0));
    r = z * (r + (1.0 / 3628800.0));
    r = z * (r + (1.0 / 362880.0));
    r = z * (r + (1.0 / 40320.0));
    r = z * (r + (1.0 / 5040.0));
    r = z * (r + (1.0 / 720.0));
    r = z * (r + (1.0 / 120.0));
    r = z * (r + (1.0 / 24.0));
    r = z * (r + (1.0 / 6.0));
    r = z * (r + 0.5);
    r = z * (r + 1.0);
    r = r + 1.0;
    int k = (int)n;
    if (k < -1022) return 0.0;
    return r * __longlong_as_double((long long)((unsigned long long)(k + 1023) << 52));
}
__device__ inline float pow_rep(float x, float y)
{
    double lx;
    if (x != x || x < 0.0f) lx = __longlong_as_double(0x7ff8000000000000LL);
    else if (x == 0.0f) lx = -__longlong_as_double(0x7ff0000000000000LL);
    else if (x == __int_as_float(0x7f800000)) lx = __longlong_as_double(0x7ff0000000000000LL);
    else lx = log2_d((double)x);
    double t = (double)y * lx;
    double r;
    if (t != t) r = __longlong_as_double(0x7ff8000000000000LL);
    else if (t >= 1024.0) r = __longlong_as_double(0x7ff0000000000000LL);
    else if (t <= -1100.0) r = 0.0;
    else r = exp2_d(t);
    return (float)r;
}

// ------------------------------------------------------------------------------------------------ sampling
// The workgroup's dynamic LDS (persistent kernels: the merged transfer function of slot 0 first, vr_pw.h / vr_p2.h; flavour 18: the
// slot tables of volume 0, below).
extern __shared__ float4 vr_lds_tf[];
// Indexed buffer loads (buffer_load ... idxen: the VGPR holds a RECORD index, the descriptor its stride): not a clang builtin yet,
// so the intrinsics by their LLVM names -- the compiler tracks them like the raw form.  Same rate as a raw load of the byte offset
// (tools/ubench/struct_buffer.hip) without the shift; index x stride wraps at 32 bits, so for buffers below 4 GiB only.
typedef unsigned vr_u4i __attribute__((ext_vector_type(4)));
extern "C" __device__ vr_u4i vr_struct_load_b128(__amdgpu_buffer_rsrc_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.ptr.buffer.load.v4i32");
extern "C" __device__ unsigned vr_struct_load_b32(__amdgpu_buffer_rsrc_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.ptr.buffer.load.i32");

struct Cell {  // the 2x2x2 texel cell of one linear 3-D fetch
    unsigned o000, o100, o010, o110, o001, o101, o011, o111;  // voxel indices
    float fx, fy, fz;
};

// One axis of texel_pair(): i0 = clamp(t, 0, n-1) and whether i1 = clamp(t+1, 0, n-1) is the NEXT texel (it is
// unless the pair is clamped at either edge: t < 0 or t >= n-1).
__device__ __forceinline__ void texel_step(float x0, int n, int& i0, bool& next)
{
    const int t = min((int)x0, n - 1);  // saturating conversion, NaN -> 0
    i0 = max(t, 0);
    next = (unsigned)t < (unsigned)(n - 1);
}
// Slot tables in LDS (DevVolume::lut): all threads of the workgroup fill them, the caller puts a barrier behind.
__device__ __forceinline__ void lut_fill(const DevVolume& v)
{
    unsigned* tab = reinterpret_cast<unsigned*>(vr_lds_tf);
    for (int e = (int)threadIdx.x; e < v.nx + 2; e += (int)blockDim.x) {
        const unsigned i = (unsigned)min(max(e - 1, 0), v.nx - 1);
        tab[e] = (i >> kVbS) * kVbN + (i & kVbM);
    }
    tab += v.nx + 2;
    for (int e = (int)threadIdx.x; e < v.ny + 2; e += (int)blockDim.x) {
        const unsigned j = (unsigned)min(max(e - 1, 0), v.ny - 1);
        tab[e] = (j >> kVbS) * v.brick_row + ((j & kVbM) << kVbS);
    }
    tab += v.ny + 2;
    for (int e = (int)threadIdx.x; e < v.nz + 2; e += (int)blockDim.x) {
        const unsigned k = (unsigned)min(max(e - 1, 0), v.nz - 1);
        tab[e] = (k >> kVbS) * v.brick_slab + ((k & kVbM) << (2u * kVbS));
    }
}
// make_cell through the tables: the clamp to [-1, n - 1] selects the same texel pair as texel_pair() for ANY value of the floored
// coordinate (saturated conversions, NaN -> 0), so the eight slots are make_cell's.
__device__ __forceinline__ Cell make_cell_lut(const DevVolume& v, f3 p)
{
    const float x = mad(p.x, (float)v.nx, -0.5f), y = mad(p.y, (float)v.ny, -0.5f), z = mad(p.z, (float)v.nz, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    Cell c;
    c.fx = x - x0;
    c.fy = y - y0;
    c.fz = z - z0;
    const int tx = min(max((int)x0, -1), v.nx - 1), ty = min(max((int)y0, -1), v.ny - 1), tz = min(max((int)z0, -1), v.nz - 1);
    const unsigned* tab = reinterpret_cast<const unsigned*>(vr_lds_tf);
    const unsigned* tx_ = tab + (tx + 1);
    const unsigned* ty_ = tab + (v.nx + 2) + (ty + 1);
    const unsigned* tz_ = tab + (v.nx + 2) + (v.ny + 2) + (tz + 1);
    const unsigned ax0 = tx_[0], ax1 = tx_[1], ay0 = ty_[0], ay1 = ty_[1], az0 = tz_[0], az1 = tz_[1];
    const unsigned r00 = ay0 + az0, r10 = ay1 + az0, r01 = ay0 + az1, r11 = ay1 + az1;
    c.o000 = r00 + ax0; c.o100 = r00 + ax1;
    c.o010 = r10 + ax0; c.o110 = r10 + ax1;
    c.o001 = r01 + ax0; c.o101 = r01 + ax1;
    c.o011 = r11 + ax0; c.o111 = r11 + ax1;
    return c;
}
__device__ __forceinline__ Cell make_cell(const DevVolume& v, f3 p)
{
    if (v.lut) return make_cell_lut(v, p);  // (wave-uniform: flavour 18)
    float x = mad(p.x, (float)v.nx, -0.5f);
    float y = mad(p.y, (float)v.ny, -0.5f);
    float z = mad(p.z, (float)v.nz, -0.5f);
    float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    Cell c;
    c.fx = x - x0;
    c.fy = y - y0;
    c.fz = z - z0;
    // the same eight voxel indices as texel_pair() on each axis gives, built from one base index and three
    // strides that are 0 where the pair is clamped
    const unsigned row = (unsigned)v.nx, slab = (unsigned)v.nx * (unsigned)v.ny;
    const int tx = (int)x0, ty = (int)y0, tz = (int)z0;  // saturating conversions, NaN -> 0
    const bool inner = (unsigned)tx < (unsigned)(v.nx - 1) && (unsigned)ty < (unsigned)(v.ny - 1) && (unsigned)tz < (unsigned)(v.nz - 1);
    if (v.bricked) {  // (wave-uniform) bricks of 4 x 4 x 4: the index is a sum of one term per axis (DevVolume)
        int i0 = tx, j0 = ty, k0 = tz, i1 = tx + 1, j1 = ty + 1, k1 = tz + 1;
        if (vr_ballot(!inner) != 0) {  // some ray's cell touches a face: the clamp-to-edge texel pairs
            texel_pair(x0, v.nx, i0, i1);
            texel_pair(y0, v.ny, j0, j1);
            texel_pair(z0, v.nz, k0, k1);
        }
        const unsigned ax0 = ((unsigned)i0 >> kVbS) * kVbN + ((unsigned)i0 & kVbM), ax1 = ((unsigned)i1 >> kVbS) * kVbN + ((unsigned)i1 & kVbM);
        const unsigned ay0 = ((unsigned)j0 >> kVbS) * v.brick_row + (((unsigned)j0 & kVbM) << kVbS);
        const unsigned ay1 = ((unsigned)j1 >> kVbS) * v.brick_row + (((unsigned)j1 & kVbM) << kVbS);
        const unsigned az0 = ((unsigned)k0 >> kVbS) * v.brick_slab + (((unsigned)k0 & kVbM) << (2u * kVbS));
        const unsigned az1 = ((unsigned)k1 >> kVbS) * v.brick_slab + (((unsigned)k1 & kVbM) << (2u * kVbS));
        const unsigned r00 = ay0 + az0, r10 = ay1 + az0, r01 = ay0 + az1, r11 = ay1 + az1;
        c.o000 = r00 + ax0; c.o100 = r00 + ax1;
        c.o010 = r10 + ax0; c.o110 = r10 + ax1;
        c.o001 = r01 + ax0; c.o101 = r01 + ax1;
        c.o011 = r11 + ax0; c.o111 = r11 + ax1;
        return c;
    }
    if (vr_ballot(!inner) == 0) {
        // every ray of the packet that samples now has its cell strictly inside the volume: nothing to clamp
        c.o000 = ((unsigned)tz * (unsigned)v.ny + (unsigned)ty) * row + (unsigned)tx;
        c.o100 = c.o000 + 1u;
        c.o010 = c.o000 + row;
        c.o110 = c.o010 + 1u;
        c.o001 = c.o000 + slab;
        c.o101 = c.o001 + 1u;
        c.o011 = c.o001 + row;
        c.o111 = c.o011 + 1u;
        return c;
    }
    int i0, j0, k0;
    bool sx, sy, sz;
    texel_step(x0, v.nx, i0, sx);
    texel_step(y0, v.ny, j0, sy);
    texel_step(z0, v.nz, k0, sz);
    const unsigned dx = sx ? 1u : 0u, dy = sy ? row : 0u, dz = sz ? slab : 0u;
    c.o000 = ((unsigned)k0 * (unsigned)v.ny + (unsigned)j0) * row + (unsigned)i0;
    c.o100 = c.o000 + dx;
    c.o010 = c.o000 + dy;
    c.o110 = c.o010 + dx;
    c.o001 = c.o000 + dz;
    c.o101 = c.o001 + dx;
    c.o011 = c.o001 + dy;
    c.o111 = c.o011 + dx;
    return c;
}

// OFF32: every bound volume is < 4 GiB, so byte offsets fit 32 bits and the loads use SGPR-base + VGPR-offset
// addressing; otherwise 64-bit addresses.
template <bool OFF32>
__device__ __forceinline__ float4 load_vec4(const float4* base, unsigned idx)
{
    if constexpr (OFF32) {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + (idx << 4));
    } else {
        return base[(size_t)idx];
    }
}
// The 16-byte voxel `idx` of a volume below 4 GiB as ONE buffer_load_dwordx4 (raw buffer, byte offset in a VGPR, the volume's
// size as the descriptor's range: an offset beyond it would return zeros instead of faulting).  An intrinsic, so the compiler
// cannot narrow it: written as a plain float4 load, the lit shader's two uses of a corner -- (z, w) before the per-step
// vote, (x, y) behind it -- were split into two dwordx2 loads per corner in the persistent kernel (16 vector-memory
// instructions per sample instead of 8, and the (x, y) halves sunk behind the vote).  -DVR_BUFFER_LOADS=0: plain loads.
#ifndef VR_BUFFER_LOADS
#define VR_BUFFER_LOADS 1
#endif
typedef unsigned vr_u4 __attribute__((ext_vector_type(4)));
typedef float vr_f4 __attribute__((ext_vector_type(4)));
template <bool OFF32>
__device__ __forceinline__ float4 load_voxel(const DevVolume& v, unsigned idx)
{
#if VR_BUFFER_LOADS
    if constexpr (OFF32) {
        // (indexed: records of 16 B, the voxel's index in the VGPR -- no shift per corner; round 4)
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(v.data), 16, (int)(v.data_bytes >> 4), 0x00020000);
        const vr_f4 f = __builtin_bit_cast(vr_f4, vr_struct_load_b128(rsrc, (int)idx, 0, 0, 0));
        return make_float4(f.x, f.y, f.z, f.w);
    }
#endif
    return load_vec4<OFF32>(v.data, idx);
}
// density of voxel idx: from the scalar plane (a_shift 2) or from the .a lane of the vec4 voxels (a_shift 4, base + 12)
template <bool OFF32>
__device__ __forceinline__ float load_a(const DevVolume& v, unsigned idx)
{
    if constexpr (OFF32) {
        return *reinterpret_cast<const float*>(v.a_base + (idx << v.a_shift));
    } else {
        return *reinterpret_cast<const float*>(v.a_base + ((size_t)idx << v.a_shift));
    }
}
// 4 / 2 consecutive densities of the plane starting at voxel idx (dword-aligned, not 16 / 8-byte aligned)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
template <bool OFF32>
__device__ __forceinline__ f4u load_d4(const float* base, unsigned idx)
{
    if constexpr (OFF32) return *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(base) + (idx << 2));
    else return *reinterpret_cast<const f4u*>(base + (size_t)idx);
}
template <bool OFF32>
__device__ __forceinline__ float load_d1(const float* base, unsigned idx)
{
    if constexpr (OFF32) return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + (idx << 2));
    else return base[(size_t)idx];
}
template <bool OFF32>
__device__ __forceinline__ f2u load_d2(const float* base, unsigned idx)
{
    if constexpr (OFF32) return *reinterpret_cast<const f2u*>(reinterpret_cast<const char*>(base) + (idx << 2));
    else return *reinterpret_cast<const f2u*>(base + (size_t)idx);
}

__device__ __forceinline__ float tri(float v000, float v100, float v010, float v110, float v001, float v101,
                                     float v011, float v111, float fx, float fy, float fz)
{
    float c00 = lerpf(v000, v100, fx);
    float c10 = lerpf(v010, v110, fx);
    float c01 = lerpf(v001, v101, fx);
    float c11 = lerpf(v011, v111, fx);
    float c0 = lerpf(c00, c10, fy);
    float c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz);
}

// The eight corners of one linear 3-D fetch as loaded (possibly still in flight) + the interpolation weights.
// Fetch and interpolation are separate so that the march loop can issue the NEXT step's loads before it waits for
// this step's transfer-function texels.
struct Fetch4 {
    float4 a, b, d, e, f, g, h, i;  // corners 000 100 010 110 001 101 011 111
};
struct Fetch1 {
    float a, b, d, e, f, g, h, i;
};
// (The three interpolation weights travel as separate scalars: as adjacent struct members the vectoriser merges
// their stores and the struct then survives as memory -- the back end parks it in LDS.)
template <bool OFF32>
__device__ __forceinline__ void fetch_rgba(const DevVolume& v, f3 p, Fetch4& q, float& fx, float& fy, float& fz)
{
    Cell c = make_cell(v, p);
    q.a = load_voxel<OFF32>(v, c.o000); q.b = load_voxel<OFF32>(v, c.o100);
    q.d = load_voxel<OFF32>(v, c.o010); q.e = load_voxel<OFF32>(v, c.o110);
    q.f = load_voxel<OFF32>(v, c.o001); q.g = load_voxel<OFF32>(v, c.o101);
    q.h = load_voxel<OFF32>(v, c.o011); q.i = load_voxel<OFF32>(v, c.o111);
    fx = c.fx;
    fy = c.fy;
    fz = c.fz;
}
template <bool OFF32>
__device__ __forceinline__ void fetch_a(const DevVolume& v, f3 p, Fetch1& q, float& fx, float& fy, float& fz)
{
    Cell c = make_cell(v, p);
    q.a = load_a<OFF32>(v, c.o000); q.b = load_a<OFF32>(v, c.o100);
    q.d = load_a<OFF32>(v, c.o010); q.e = load_a<OFF32>(v, c.o110);
    q.f = load_a<OFF32>(v, c.o001); q.g = load_a<OFF32>(v, c.o101);
    q.h = load_a<OFF32>(v, c.o011); q.i = load_a<OFF32>(v, c.o111);
    fx = c.fx;
    fy = c.fy;
    fz = c.fz;
}

// ---- gradients on the fly (volumes whose .rgb is verifiably PreComputeGradient(false) of their .a) ---------------------
// VolumeFile::PreComputeGradient (VolumeFile.cpp:196-257): g = (-(p - m)) * 0.5 per axis from the +-1 neighbours' densities,
// a neighbour outside the grid counting as 0.  The 8 corners of a trilinear cell need 32 distinct densities: the four
// x-rows of the cell extended by one voxel either side (4 x 16 B) and the 2-voxel pieces of the rows above / below / in
// front / behind (8 x 8 B) -- 128 B per sample, as many as the eight vec4 voxels, but out of a volume a quarter the size.
// The same subtraction, negation and halving the preparation pass performs, so the corners come out bit-identical to the
// stored voxels and everything downstream (interpolation, shading) is unchanged.
__device__ __forceinline__ float grad_cd(float p, float m) { return (-(p - m)) * 0.5f; }
// one corner the slow way (cells that touch the volume's faces): clamped texel, out-of-grid neighbours are 0
template <bool OFF32>
__device__ __forceinline__ float4 corner_otf(const DevVolume& v, int i, int j, int k)
{
    const unsigned row = (unsigned)v.nx, slab = (unsigned)v.nx * (unsigned)v.ny;
    const unsigned c = ((unsigned)k * (unsigned)v.ny + (unsigned)j) * row + (unsigned)i;
    const float d = load_d1<OFF32>(v.dens, c);
    const float mx = i > 0 ? load_d1<OFF32>(v.dens, c - 1u) : 0.0f, px = i + 1 < v.nx ? load_d1<OFF32>(v.dens, c + 1u) : 0.0f;
    const float my = j > 0 ? load_d1<OFF32>(v.dens, c - row) : 0.0f, py = j + 1 < v.ny ? load_d1<OFF32>(v.dens, c + row) : 0.0f;
    const float mz = k > 0 ? load_d1<OFF32>(v.dens, c - slab) : 0.0f, pz = k + 1 < v.nz ? load_d1<OFF32>(v.dens, c + slab) : 0.0f;
    return make_float4(grad_cd(px, mx), grad_cd(py, my), grad_cd(pz, mz), d);
}
template <bool OFF32>
__device__ __forceinline__ void fetch_rgba_otf(const DevVolume& v, f3 p, Fetch4& q, float& fx, float& fy, float& fz)
{
    const float x = mad(p.x, (float)v.nx, -0.5f), y = mad(p.y, (float)v.ny, -0.5f), z = mad(p.z, (float)v.nz, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    fx = x - x0;
    fy = y - y0;
    fz = z - z0;
    const int tx = (int)x0, ty = (int)y0, tz = (int)z0;  // saturating conversions, NaN -> 0
    // the cell and its one-voxel apron strictly inside the grid: 1 <= t <= n - 3 on every axis
    const bool inner = (unsigned)(tx - 1) < (unsigned)(v.nx - 3) && (unsigned)(ty - 1) < (unsigned)(v.ny - 3) &&
                       (unsigned)(tz - 1) < (unsigned)(v.nz - 3) && v.nx >= 4 && v.ny >= 4 && v.nz >= 4;
    if (vr_ballot(!inner) == 0) {
        const unsigned row = (unsigned)v.nx, slab = (unsigned)v.nx * (unsigned)v.ny;
        const unsigned b = ((unsigned)tz * (unsigned)v.ny + (unsigned)ty) * row + (unsigned)tx - 1u;  // (x0-1, y0, z0)
        const f4u r00 = load_d4<OFF32>(v.dens, b), r10 = load_d4<OFF32>(v.dens, b + row);
        const f4u r01 = load_d4<OFF32>(v.dens, b + slab), r11 = load_d4<OFF32>(v.dens, b + slab + row);
        const unsigned b1 = b + 1u;
        const f2u ym0 = load_d2<OFF32>(v.dens, b1 - row), ym1 = load_d2<OFF32>(v.dens, b1 - row + slab);
        const f2u yp0 = load_d2<OFF32>(v.dens, b1 + 2u * row), yp1 = load_d2<OFF32>(v.dens, b1 + 2u * row + slab);
        const f2u zm0 = load_d2<OFF32>(v.dens, b1 - slab), zm1 = load_d2<OFF32>(v.dens, b1 - slab + row);
        const f2u zp0 = load_d2<OFF32>(v.dens, b1 + 2u * slab), zp1 = load_d2<OFF32>(v.dens, b1 + 2u * slab + row);
        q.a = make_float4(grad_cd(r00.z, r00.x), grad_cd(r10.y, ym0.x), grad_cd(r01.y, zm0.x), r00.y);  // 000
        q.b = make_float4(grad_cd(r00.w, r00.y), grad_cd(r10.z, ym0.y), grad_cd(r01.z, zm0.y), r00.z);  // 100
        q.d = make_float4(grad_cd(r10.z, r10.x), grad_cd(yp0.x, r00.y), grad_cd(r11.y, zm1.x), r10.y);  // 010
        q.e = make_float4(grad_cd(r10.w, r10.y), grad_cd(yp0.y, r00.z), grad_cd(r11.z, zm1.y), r10.z);  // 110
        q.f = make_float4(grad_cd(r01.z, r01.x), grad_cd(r11.y, ym1.x), grad_cd(zp0.x, r00.y), r01.y);  // 001
        q.g = make_float4(grad_cd(r01.w, r01.y), grad_cd(r11.z, ym1.y), grad_cd(zp0.y, r00.z), r01.z);  // 101
        q.h = make_float4(grad_cd(r11.z, r11.x), grad_cd(yp1.x, r01.y), grad_cd(zp1.x, r10.y), r11.y);  // 011
        q.i = make_float4(grad_cd(r11.w, r11.y), grad_cd(yp1.y, r01.z), grad_cd(zp1.y, r10.z), r11.z);  // 111
        return;
    }
    int i0, i1, j0, j1, k0, k1;
    texel_pair(x0, v.nx, i0, i1);
    texel_pair(y0, v.ny, j0, j1);
    texel_pair(z0, v.nz, k0, k1);
    q.a = corner_otf<OFF32>(v, i0, j0, k0); q.b = corner_otf<OFF32>(v, i1, j0, k0);
    q.d = corner_otf<OFF32>(v, i0, j1, k0); q.e = corner_otf<OFF32>(v, i1, j1, k0);
    q.f = corner_otf<OFF32>(v, i0, j0, k1); q.g = corner_otf<OFF32>(v, i1, j0, k1);
    q.h = corner_otf<OFF32>(v, i0, j1, k1); q.i = corner_otf<OFF32>(v, i1, j1, k1);
}
// Two channels at a time (packed f32 on the register halves the 16-byte loads deliver: no shuffling).  Per lane
// and per channel the operations and their order are those of tri(): a + (b - a) * t, separately rounded.
typedef float v2f __attribute__((ext_vector_type(2)));
// a * b + c on a register pair (v_pk_fma_f32 when fused)
__device__ __forceinline__ v2f mad2(v2f a, v2f b, v2f c)
{
#if VR_FUSED
    return __builtin_elementwise_fma(a, b, c);
#else
    return a * b + c;
#endif
}
__device__ __forceinline__ v2f lerp2(v2f a, v2f b, float t) { return mad2(b - a, v2f{t, t}, a); }
// inv_sqrt_exact of two arguments at once: the same operations on each, the fused multiply-adds of the expansion as packed
// instructions on a register pair (8 of the 32 instructions of two separate calls saved: the lit shader normalises the
// gradient and the light direction in every sample, and the kernel sits on the vector-ALU issue limit).  The generic
// operators when either argument is outside [2^-60, 2^60).
// UNI: the choice is made once for the wavefront (the generic operators for every lane when any lane needs them: same bits,
// one scalar branch instead of two exec-mask regions -- for loops that count their scalar instructions, vr_pw.h).
template <bool UNI = false>
__device__ __forceinline__ v2f inv_sqrt_exact2(float a, float b)
{
    bool in_range = (__float_as_uint(a) - 0x21800000u < 0x3c000000u) && (__float_as_uint(b) - 0x21800000u < 0x3c000000u);
    if constexpr (UNI) in_range = vr_ballot(!in_range) == 0;
    if (in_range) {
        const v2f x = v2f{a, b};
        v2f y = v2f{__builtin_amdgcn_sqrtf(a), __builtin_amdgcn_sqrtf(b)};
        const v2f yd = v2f{__int_as_float(__float_as_int(y.x) - 1), __int_as_float(__float_as_int(y.y) - 1)};
        const v2f yu = v2f{__int_as_float(__float_as_int(y.x) + 1), __int_as_float(__float_as_int(y.y) + 1)};
        const v2f rd = __builtin_elementwise_fma(-yd, y, x), ru = __builtin_elementwise_fma(-yu, y, x);
        y.x = (rd.x <= 0.0f) ? yd.x : y.x;
        y.y = (rd.y <= 0.0f) ? yd.y : y.y;
        y.x = (ru.x > 0.0f) ? yu.x : y.x;
        y.y = (ru.y > 0.0f) ? yu.y : y.y;
        const v2f one = v2f{1.0f, 1.0f};
        v2f r = v2f{__builtin_amdgcn_rcpf(y.x), __builtin_amdgcn_rcpf(y.y)};
        r = __builtin_elementwise_fma(__builtin_elementwise_fma(-y, r, one), r, r);
        v2f q = r;
        q = __builtin_elementwise_fma(__builtin_elementwise_fma(-y, q, one), r, q);
        return __builtin_elementwise_fma(__builtin_elementwise_fma(-y, q, one), r, q);
    }
    return v2f{inv_sqrt_exact(a), inv_sqrt_exact(b)};
}
__device__ __forceinline__ v2f tri2(v2f v000, v2f v100, v2f v010, v2f v110, v2f v001, v2f v101, v2f v011, v2f v111,
                                    float fx, float fy, float fz)
{
    v2f c00 = lerp2(v000, v100, fx);
    v2f c10 = lerp2(v010, v110, fx);
    v2f c01 = lerp2(v001, v101, fx);
    v2f c11 = lerp2(v011, v111, fx);
    v2f c0 = lerp2(c00, c10, fy);
    v2f c1 = lerp2(c01, c11, fy);
    return lerp2(c0, c1, fz);
}
__device__ __forceinline__ v2f lo2(const float4& v) { return v2f{v.x, v.y}; }
__device__ __forceinline__ v2f hi2(const float4& v) { return v2f{v.z, v.w}; }
__device__ __forceinline__ v2f interp_zw(const Fetch4& q, float fx, float fy, float fz)  // (gradient z, density)
{
    return tri2(hi2(q.a), hi2(q.b), hi2(q.d), hi2(q.e), hi2(q.f), hi2(q.g), hi2(q.h), hi2(q.i), fx, fy, fz);
}
__device__ __forceinline__ v2f interp_xy(const Fetch4& q, float fx, float fy, float fz)  // (gradient x, gradient y)
{
    return tri2(lo2(q.a), lo2(q.b), lo2(q.d), lo2(q.e), lo2(q.f), lo2(q.g), lo2(q.h), lo2(q.i), fx, fy, fz);
}
__device__ __forceinline__ float interp_a(const Fetch1& q, float fx, float fy, float fz)
{
    return tri(q.a, q.b, q.d, q.e, q.f, q.g, q.h, q.i, fx, fy, fz);
}

// textureSample(vol, samplerLin, p) -> all four channels (OTF: the corners' gradients derived from the density plane)
template <bool OFF32, bool OTF = false>
__device__ __forceinline__ float4 tex3_rgba(const DevVolume& v, f3 p)
{
    Fetch4 q;
    float fx, fy, fz;
    if constexpr (OTF) fetch_rgba_otf<OFF32>(v, p, q, fx, fy, fz);
    else fetch_rgba<OFF32>(v, p, q, fx, fy, fz);
    const v2f zw = interp_zw(q, fx, fy, fz), xy = interp_xy(q, fx, fy, fz);
    return make_float4(xy.x, xy.y, zw.x, zw.y);
}

// textureSample(vol, samplerLin, p).a  -- only the density plane of the vec4 voxels is touched
template <bool OFF32>
__device__ __forceinline__ float tex3_a(const DevVolume& v, f3 p)
{
    Fetch1 q;
    float fx, fy, fz;
    fetch_a<OFF32>(v, p, q, fx, fy, fz);
    return interp_a(q, fx, fy, fz);
}

// textureSample(vol, samplerNN, p)  (TFCalibrationApp.wgsl:172)
template <bool OFF32>
__device__ __forceinline__ float4 tex3_nearest(const DevVolume& v, f3 p)
{
    int i = clampi((int)floorf(p.x * (float)v.nx), 0, v.nx - 1);
    int j = clampi((int)floorf(p.y * (float)v.ny), 0, v.ny - 1);
    int k = clampi((int)floorf(p.z * (float)v.nz), 0, v.nz - 1);
    unsigned idx = ((unsigned)k * (unsigned)v.ny + (unsigned)j) * (unsigned)v.nx + (unsigned)i;
    if (v.bricked)
        idx = ((unsigned)i >> kVbS) * kVbN + ((unsigned)i & kVbM) + ((unsigned)j >> kVbS) * v.brick_row + (((unsigned)j & kVbM) << kVbS) +
              ((unsigned)k >> kVbS) * v.brick_slab + (((unsigned)k & kVbM) << (2u * kVbS));
    return load_vec4<OFF32>(v.data, idx);
}

struct TfSample {
    float opacity;
    f3 rgb;
};
struct TfFetch {  // the four texels of one opacity + colour look-up as loaded, + the weights
    float o0, o1, fo, fc;
    float4 c0, c1;
};
__device__ __forceinline__ int padded_texel(float x0, int n)
{
    // texel_pair(x0, n) is (j - 1, j) clamped to [0, n-1] = entries [j], [j+1] of the padded table (DevTF);
    // min before the +1 so that it cannot overflow (the conversion saturates, NaN -> 0)
    return max(min((int)x0, n - 1) + 1, 0);
}
__device__ __forceinline__ TfFetch tf_fetch(const DevTF& tf, float d)
{
    // the opacity and the colour texture are separate 1-D textures, each with its own resolution; when the two are
    // equal (every scene of the reference, until a preset of another size is loaded) index and weight are shared
    float xo = mad(d, (float)tf.res_o, -0.5f);
    float xo0 = floorf(xo);
    TfFetch q;
    q.fo = xo - xo0;
    const int jo = padded_texel(xo0, tf.res_o);
    int jc = jo;
    q.fc = q.fo;
    if (tf.res_c != tf.res_o) {  // wave-uniform
        float xc = mad(d, (float)tf.res_c, -0.5f);
        float xc0 = floorf(xc);
        q.fc = xc - xc0;
        jc = padded_texel(xc0, tf.res_c);
    }
    // tables are far below 4 GiB: SGPR base + 32-bit byte offset
    const char* po = reinterpret_cast<const char*>(tf.opacity) + ((unsigned)jo << 2);
    const char* pc = reinterpret_cast<const char*>(tf.color) + ((unsigned)jc << 4);
    q.o0 = reinterpret_cast<const float*>(po)[0];
    q.o1 = reinterpret_cast<const float*>(po)[1];
    q.c0 = reinterpret_cast<const float4*>(pc)[0];
    q.c1 = reinterpret_cast<const float4*>(pc)[1];
    return q;
}
__device__ __forceinline__ TfSample tf_finish(const TfFetch& q)
{
    TfSample s;
    s.opacity = lerpf(q.o0, q.o1, q.fo);
    s.rgb = mk3(lerpf(q.c0.x, q.c1.x, q.fc), lerpf(q.c0.y, q.c1.y, q.fc), lerpf(q.c0.z, q.c1.z, q.fc));
    return s;
}
__device__ __forceinline__ TfSample tf_lookup(const DevTF& tf, float d) { return tf_finish(tf_fetch(tf, d)); }
// LTF: TF slot 0 read from LDS.  The persistent-wave kernel (vr_pw.h) keeps the padded opacity and colour tables of slot 0,
// merged into one float4 per entry -- (r, g, b, opacity); the colour texture's own .a is never sampled (`.rgb`,
// BasicVolumeApp.wgsl:173-175) -- in its workgroup's LDS, so the second dependent round trip of a step (corners, then the
// texels their density selects) is two ds_read_b128 instead of four L1 look-ups behind the other wavefronts' corner loads.
// The host asks for it only when the two tables have one resolution (every scene of the reference until a preset of
// another size is loaded): one index and one weight, as in tf_fetch.  Same texels, same lerps: the same bits.
__device__ __forceinline__ TfFetch tf_fetch_lds(const DevTF& tf, float d)
{
    const float xo = mad(d, (float)tf.res_o, -0.5f);
    const float xo0 = floorf(xo);
    TfFetch q;
    q.fo = xo - xo0;
    q.fc = q.fo;
    const int j = padded_texel(xo0, tf.res_o);
    q.c0 = vr_lds_tf[j];
    q.c1 = vr_lds_tf[j + 1];
    q.o0 = q.c0.w;
    q.o1 = q.c1.w;
    return q;
}
template <bool LTF>
__device__ __forceinline__ TfFetch tf_fetch0(const MarchParams& P, float d)
{
    if constexpr (LTF) return tf_fetch_lds(P.tf[0], d);
    else return tf_fetch(P.tf[0], d);
}
template <bool LTF>
__device__ __forceinline__ TfSample tf_lookup0(const MarchParams& P, float d) { return tf_finish(tf_fetch0<LTF>(P, d)); }
// Pins the point where the texels are first needed: arithmetic on them cannot be moved above this statement, so
// loads issued before it (the next step's corners) are in flight while the wave waits for the texels.
__device__ __forceinline__ void tf_pin(TfFetch& q)
{
    asm volatile("" : "+v"(q.o0), "+v"(q.o1), "+v"(q.c0.x), "+v"(q.c0.y), "+v"(q.c0.z), "+v"(q.c1.x), "+v"(q.c1.y), "+v"(q.c1.z));
}

// ------------------------------------------------------------------------------------------------ ray set-up
struct Ray {
    f3 start, end, world0;
    bool hit;
};

__device__ __forceinline__ f3 unproject(const MarchParams& P, float nx, float ny, float nz)
{
    float v[4], w[4];
    mat4_mul_point(P.proj_inv, nx, ny, nz, 1.0f, v);
    float vx = v[0] / v[3], vy = v[1] / v[3], vz = v[2] / v[3];
    mat4_mul_point(P.view_inv, vx, vy, vz, 1.0f, w);
    return mk3(w[0], w[1], w[2]);
}

__device__ inline Ray setup_ray(const MarchParams& P, int px, int py)
{
    const float bmin[3] = {-0.5f, -0.5f, -0.25f};
    const float bmax[3] = {0.5f, 0.5f, 0.25f};
    Ray ray;
    ray.hit = false;
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float ndcx = (2.0f * fx) / (float)P.W - 1.0f;
    float ndcy = 1.0f - (2.0f * fy) / (float)P.H;
    f3 O = unproject(P, ndcx, ndcy, 0.0f);
    f3 F = unproject(P, ndcx, ndcy, 1.0f);
    f3 D = mk3(F.x - O.x, F.y - O.y, F.z - O.z);
    float seg = length3s(D);
    f3 dn = normalize3s(D);
    float o[3] = {O.x, O.y, O.z}, d[3] = {dn.x, dn.y, dn.z};
    float t0 = -INFINITY, t1 = INFINITY;
    int a0 = -1, a1 = -1;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (d[a] != 0.0f) {
            float inv = 1.0f / d[a];
            float ta = (bmin[a] - o[a]) * inv;
            float tb = (bmax[a] - o[a]) * inv;
            float tn = ta < tb ? ta : tb;
            float tf = ta < tb ? tb : ta;
            if (tn > t0) { t0 = tn; a0 = a; }
            if (tf < t1) { t1 = tf; a1 = a; }
        } else if (o[a] < bmin[a] || o[a] > bmax[a]) {
            miss = true;
        }
    }
    if (miss) return ray;
    if (!(t0 < t1)) return ray;
    if (!(t0 >= 0.0f && t0 <= seg)) return ray;
    if (a0 < 0 || a1 < 0) return ray;
    float P0[3], P1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        P0[a] = o[a] + d[a] * t0;
        P1[a] = o[a] + d[a] * t1;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (a == a0) P0[a] = (d[a] > 0.0f) ? bmin[a] : bmax[a];
        if (a == a1) P1[a] = (d[a] > 0.0f) ? bmax[a] : bmin[a];
    }
    ray.start = mk3(P0[0] + 0.5f, P0[1] + 0.5f, 0.5f - 2.0f * P0[2]);
    ray.end = mk3(P1[0] + 0.5f, P1[1] + 0.5f, 0.5f - 2.0f * P1[2]);
    ray.world0 = mk3(P0[0], P0[1], P0[2]);
    ray.hit = true;
    return ray;
}

// ------------------------------------------------------------------------------------------------ shading / blend
__device__ __forceinline__ f3 shade(f3 N, f3 w, f3 lpos, f3 dif, f3 amb, float kD, float kA)
{
    f3 L = normalize3(mk3(lpos.x - w.x, lpos.y - w.y, lpos.z - w.z));
    float m = max0(dot3(N, L));
    return mk3(mad(dif.x * m, kD, amb.x * kA), mad(dif.y * m, kD, amb.y * kA), mad(dif.z * m, kD, amb.z * kA));
}

__device__ __forceinline__ void blend(f3 rgb, float a, float4& dst)  // FrontToBackBlend
{
    float sr = rgb.x * a, sg = rgb.y * a, sb = rgb.z * a;
    float om = 1.0f - dst.w;
    dst.x = mad(om, sr, dst.x);
    dst.y = mad(om, sg, dst.y);
    dst.z = mad(om, sb, dst.z);
    dst.w = mad(om, a, dst.w);
}

template <int V>
__device__ __forceinline__ bool can_blend(float a)  // the shader's opacity cut-off
{
    if constexpr (V == V_BASIC || V == V_MULTI_CTRT || V == V_TF_CALIB || V == V_ILLUSTRATIVE)
        return a <= 0.95f;  // (V_LIGHT_INSHADER is BasicVolLightApp.wgsl: dst.a < 1.0, :220)
    else
        return a < 1.0f;
}

// What one sample contributes before the blend: colour and opacity of the position p (w = world position), for a
// position that passed IsInSampleCoords and the cut-off.
struct Src {
    f3 rgb;
    float a;
};
template <bool FMED>
__device__ __forceinline__ int brick_of(const MarchParams& P, f3 p);
__device__ __forceinline__ float2 brick_record(const MarchParams& P, int bid);

// Three-volume composite: the mask enters the shader through one comparison (any of r, g, b > 0), and the dose only through
// the samples that pass it.  Where every mask voxel the sample's brick can touch is <= 0 -- the brick record the skipping test
// keeps anyway (kernels with skipping only: P.bricks) -- the interpolated channels are <= 0 too: for a packet all of whose
// rays are in such bricks (most of the body) neither the eight 16-byte mask corners nor the dose are fetched (mask = 0 fails
// the comparison like the real value; *any_masked = false tells src_volume_mask to leave the dose's table out as well).
template <bool OFF32>
__device__ __forceinline__ void fetch_mask_and_dose(const MarchParams& P, f3 p, float4& mask, float& rt, bool& any_masked);

// The arithmetic of three shaders behind their fetches (sample_src fetches and calls these; sample_and_blend puts its vote
// between the two).
template <bool OFF32, bool LTF = false>
__device__ __forceinline__ Src src_inshader(const MarchParams& P, f3 p, f3 w, float ss, float density)
{
    // BasicVolLightApp.wgsl:209-222 with :212 enabled; ComputeGradient :239-253.  dirs[k] * step = (step, 0, 0) ...:
    // the products with 0 and the additions of the resulting zeros are kept (they are the shader's operations)
    Src o;
    TfSample t = tf_lookup0<LTF>(P, density);
    const float d1 = 1.0f * ss, d0 = 0.0f * ss;
    const float rx = tex3_a<OFF32>(P.vol[0], mk3(p.x + d1, p.y + d0, p.z + d0)) - tex3_a<OFF32>(P.vol[0], mk3(p.x - d1, p.y - d0, p.z - d0));
    const float ry = tex3_a<OFF32>(P.vol[0], mk3(p.x + d0, p.y + d1, p.z + d0)) - tex3_a<OFF32>(P.vol[0], mk3(p.x - d0, p.y - d1, p.z - d0));
    const float rz = tex3_a<OFF32>(P.vol[0], mk3(p.x + d0, p.y + d0, p.z + d1)) - tex3_a<OFF32>(P.vol[0], mk3(p.x - d0, p.y - d0, p.z - d1));
    const float l = length3(mk3(rx, ry, rz));
    f3 g = mk3(0.0f, 0.0f, 0.0f);
    if (l != 0.0f) g = mk3((-rx) / l, (-ry) / l, (-rz) / l);  // (NaN length: the division gives NaN, as in the shader)
    f3 N = normalize3(g);  // normalize(vec3(0)) = NaN -> max(NaN, 0) = 0: ambient only
    f3 s = shade(N, w, mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]),
                 mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                 mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 2.5f, 0.5f);
    o.rgb = mk3(t.rgb.x * s.x, t.rgb.y * s.y, t.rgb.z * s.z);
    o.a = t.opacity;
    return o;
}
// (with_rt = false, wave-uniform: the caller knows that no ray of the packet can be masked -- mask is then 0 -- and has
// not fetched the dose; its table look-up, whose result only a masked sample uses, is left out)
template <bool LTF = false>
__device__ __forceinline__ Src src_volume_mask(const MarchParams& P, f3 w, float4 mask, float rt, float4 ct, bool with_rt = true)
{
    Src o;
    TfSample trt;
    trt.rgb = mk3(0.0f, 0.0f, 0.0f);
    trt.opacity = 0.0f;
    if (with_rt) trt = tf_lookup(P.tf[1], rt);
    TfSample tct = tf_lookup0<LTF>(P, ct.w);
    f3 N = normalize3(mk3(ct.x, ct.y, ct.z));
    f3 s = shade(N, w, mk3(0.0f, -5.0f, 0.0f), mk3(0.96f, 0.76f, 0.67f), mk3(1.0f, 1.0f, 1.0f), 1.5f, 0.5f);
    o.rgb = mk3(tct.rgb.x * s.x, tct.rgb.y * s.y, tct.rgb.z * s.z);
    o.a = tct.opacity;
    if (mask.x > 0.0f || mask.y > 0.0f || mask.z > 0.0f) {
        o.a = trt.opacity;
        o.rgb = trt.rgb;
    }
    return o;
}
template <bool LTF = false>
__device__ __forceinline__ Src src_three_files(const MarchParams& P, float ct, float rt)
{
    Src o;
    TfSample tct = tf_lookup0<LTF>(P, ct);
    TfSample trt = tf_lookup(P.tf[1], rt);
    float om = 1.0f - trt.opacity;
    o.rgb = mk3(mad(trt.rgb.x, trt.opacity, tct.rgb.x * om), mad(trt.rgb.y, trt.opacity, tct.rgb.y * om),
                mad(trt.rgb.z, trt.opacity, tct.rgb.z * om));
    o.a = tct.opacity;
    return o;
}

template <bool OFF32>
__device__ __forceinline__ void fetch_mask_and_dose(const MarchParams& P, f3 p, float4& mask, float& rt, bool& any_masked)
{
    mask = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    rt = 0.0f;
    any_masked = true;
    if (P.bricks != nullptr && P.use_rgb && P.zskip_prefix >= -1)  // (wave-uniform)
        any_masked = vr_ballot(!(brick_record(P, brick_of<OFF32>(P, p)).y <= 0.0f)) != 0;
    if (any_masked) {
        mask = tex3_rgba<OFF32>(P.vol[0], p);
        rt = tex3_a<OFF32>(P.vol[1], p);
    }
}

// (start = the ray's first position and dst_a = the opacity accumulated so far are read by the illustrative shader only,
// ss = the ray's step size after the variable-step override by the in-shader gradient only)
template <int V, bool OFF32, bool OTF = false, bool LTF = false>
__device__ __forceinline__ Src sample_src(const MarchParams& P, f3 p, f3 w, f3 start = f3{0.0f, 0.0f, 0.0f}, float dst_a = 0.0f,
                                          float ss = 0.0f)
{
    Src o;
    if constexpr (V == V_BASIC) {
        float density = tex3_a<OFF32>(P.vol[0], p);
        TfSample t = tf_lookup0<LTF>(P, density);
        o.rgb = t.rgb;
        o.a = t.opacity;
    } else if constexpr (V == V_LIGHT) {
        float4 v = tex3_rgba<OFF32, OTF>(P.vol[0], p);
        TfSample t = tf_lookup0<LTF>(P, v.w);
        f3 N = normalize3(mk3(v.x, v.y, v.z));
        f3 s = shade(N, w, mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]),
                     mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                     mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 2.5f, 0.5f);
        o.rgb = mk3(t.rgb.x * s.x, t.rgb.y * s.y, t.rgb.z * s.z);
        o.a = t.opacity;
    } else if constexpr (V == V_LIGHT_INSHADER) {
        o = src_inshader<OFF32, LTF>(P, p, w, ss, tex3_a<OFF32>(P.vol[0], p));
    } else if constexpr (V == V_VOLUME_MASK) {
        const float4 ct = tex3_rgba<OFF32>(P.vol[2], p);
        float4 mask;
        float rt;
        bool any_masked;
        fetch_mask_and_dose<OFF32>(P, p, mask, rt, any_masked);
        o = src_volume_mask<LTF>(P, w, mask, rt, ct, any_masked);
    } else if constexpr (V == V_THREE_FILES) {
        const float ct = tex3_a<OFF32>(P.vol[0], p);
        const float rt = tex3_a<OFF32>(P.vol[1], p);
        o = src_three_files<LTF>(P, ct, rt);
    } else if constexpr (V == V_MULTI_CTRT) {
        float4 ct = tex3_rgba<OFF32>(P.vol[0], p);
        float rt = tex3_a<OFF32>(P.vol[1], p);
        TfSample tct = tf_lookup0<LTF>(P, ct.w);
        TfSample trt = tf_lookup(P.tf[1], rt);
        float om = 1.0f - trt.opacity;
        f3 col = mk3(mad(trt.rgb.x, trt.opacity, tct.rgb.x * om), mad(trt.rgb.y, trt.opacity, tct.rgb.y * om),
                     mad(trt.rgb.z, trt.opacity, tct.rgb.z * om));
        f3 g = mk3(ct.x, ct.y, ct.z);
        f3 N = normalize3(g);
        f3 s = shade(N, w, mk3(0.0f, -5.0f, 0.0f), mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                     mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 3.5f, 0.5f);
        o.rgb = mk3(col.x * s.x, col.y * s.y, col.z * s.z);
        o.a = tct.opacity * length3(g);
    } else if constexpr (V == V_ILLUSTRATIVE) {  // MutliCTRTIllustrative.wgsl:271-310
        float4 ct = tex3_rgba<OFF32>(P.vol[0], p);
        float rt = tex3_a<OFF32>(P.vol[1], p);
        TfSample tct = tf_lookup0<LTF>(P, ct.w);
        TfSample trt = tf_lookup(P.tf[1], rt);
        float om = 1.0f - trt.opacity;
        f3 col = mk3(mad(trt.rgb.x, trt.opacity, tct.rgb.x * om), mad(trt.rgb.y, trt.opacity, tct.rgb.y * om),
                     mad(trt.rgb.z, trt.opacity, tct.rgb.z * om));
        f3 g = mk3(ct.x, ct.y, ct.z);
        f3 N = normalize3(g);
        f3 s3 = shade(N, w, mk3(0.0f, -5.0f, 0.0f), mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                      mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 3.5f, 0.5f);
        o.rgb = mk3(col.x * s3.x, col.y * s3.y, col.z * s3.z);
        // IllustrativeContextPreservingOpacity :158-186 (distance in texture space)
        f3 L = normalize3(mk3(0.0f - w.x, -5.0f - w.y, 0.0f - w.z));
        f3 Vv = normalize3(mk3(P.camera_pos[0] - w.x, P.camera_pos[1] - w.y, P.camera_pos[2] - w.z));
        f3 H = normalize3(mk3(Vv.x + L.x, Vv.y + L.y, Vv.z + L.z));
        float s = (0.5f + 2.5f * length3(mk3(L.x * g.x, L.y * g.y, L.z * g.z))) +
                  1.0f * pow_rep(length3(mk3(H.x * g.x, H.y * g.y, H.z * g.z)), 1.0f);
        float dist = length3(mk3(p.x - start.x, p.y - start.y, p.z - start.z));
        if (dist > 1.0f) dist = 1.0f;
        float inner = pow_rep(((5.0f * s) * (1.0f - dist)) * (1.0f - dst_a), 0.8f);
        o.a = tct.opacity * pow_rep(length3(g), inner);
    } else {  // V_TF_CALIB
        float density = tex3_a<OFF32>(P.vol[0], p);
        float4 mask = tex3_nearest<OFF32>(P.vol[1], p);
        TfSample t = tf_lookup0<LTF>(P, density);
        if (mask.x > 0.0f) {
            t.rgb = mk3(1.0f, 1.0f, 0.0f);
            t.opacity = 0.1f;
        }
        o.rgb = t.rgb;
        o.a = t.opacity;
    }
    return o;
}
// True if the opacity table yields exactly 0 for density d: both texels of the look-up lie in the table's zero prefix
// (the index is tf_fetch's own; a NaN or infinite density gives a NaN opacity and is not "zero").
__device__ __forceinline__ bool opacity_is_zero(const MarchParams& P, float d)
{
    const int jo = padded_texel(floorf(mad(d, (float)P.tf[0].res_o, -0.5f)), P.tf[0].res_o);
    return (d - d == 0.0f) && jo <= P.zskip_prefix;  // finite: an infinite density has a NaN weight, hence a NaN opacity
}

// The lit shader from the interpolated voxel to the blend (BasicVolLightApp.wgsl:216-223), on (x, y) / (r, g) register pairs
// so that the packed instructions need no shuffling; per component the operations and their order are those of
// normalize3 / shade / blend.  zw = (gradient z, density), gxy = (gradient x, gradient y), tq = the table texels of `density`.
// (lpos / dif / amb, kD / kA: the scene's light -- the uniforms of BasicVolLightApp, the constants of VolumeMaskApp.wgsl:116-123;
// MASKED: a lane whose mask sample says so takes the dose table's sample `trt` unshaded instead, VolumeMaskApp.wgsl:201-205)
template <bool UNI = false, bool MASKED = false>
__device__ __forceinline__ void shade_blend_packed(f3 lpos, f3 dif, f3 amb, float kD, float kA, f3 w, v2f zw, v2f gxy, const TfFetch& tq, float4& dst,
                                                   bool masked = false, f3 trt_rgb = f3{0.0f, 0.0f, 0.0f}, float trt_a = 0.0f)
{
    v2f Lxy = v2f{lpos.x - w.x, lpos.y - w.y};
    float Lz = lpos.z - w.z;
#if VR_FUSED
    const v2f inv = inv_sqrt_exact2<UNI>(mad(zw.x, zw.x, mad(gxy.y, gxy.y, gxy.x * gxy.x)), mad(Lz, Lz, mad(Lxy.y, Lxy.y, Lxy.x * Lxy.x)));
#else
    const v2f g2 = gxy * gxy, l2 = Lxy * Lxy;
    const v2f inv = inv_sqrt_exact2<UNI>((g2.x + g2.y) + zw.x * zw.x, (l2.x + l2.y) + Lz * Lz);
#endif
    const float inv_g = inv.x, inv_l = inv.y;  // (the two normalisations' 1 / length, computed side by side)
    const v2f Nxy = gxy * inv_g;
    const float Nz = zw.x * inv_g;
    Lxy = Lxy * inv_l;
    Lz = Lz * inv_l;
#if VR_FUSED
    const float m = max0(mad(Nz, Lz, mad(Nxy.y, Lxy.y, Nxy.x * Lxy.x)));
#else
    const v2f nl = Nxy * Lxy;
    const float m = max0((nl.x + nl.y) + Nz * Lz);
#endif
    const v2f sh_rg = mad2(v2f{dif.x, dif.y} * m, v2f{kD, kD}, v2f{amb.x, amb.y} * kA);
    const float sh_b = mad(dif.z * m, kD, amb.z * kA);
    float opacity = lerpf(tq.o0, tq.o1, tq.fo);
    const v2f c_rg = lerp2(v2f{tq.c0.x, tq.c0.y}, v2f{tq.c1.x, tq.c1.y}, tq.fc);
    const float c_b = lerpf(tq.c0.z, tq.c1.z, tq.fc);
    v2f rgb_rg = c_rg * sh_rg;
    float rgb_b = c_b * sh_b;
    if constexpr (MASKED) {
        rgb_rg = masked ? v2f{trt_rgb.x, trt_rgb.y} : rgb_rg;
        rgb_b = masked ? trt_rgb.z : rgb_b;
        opacity = masked ? trt_a : opacity;
    }
    const v2f src_rg = rgb_rg * opacity;  // FrontToBackBlend: (rgb * a, a)
    const float src_b = rgb_b * opacity;
    const float om = 1.0f - dst.w;
    const v2f d_rg = mad2(src_rg, v2f{om, om}, v2f{dst.x, dst.y});
    dst.x = d_rg.x;
    dst.y = d_rg.y;
    dst.z = mad(om, src_b, dst.z);
    dst.w = mad(om, opacity, dst.w);
}
template <bool UNI = false>
__device__ __forceinline__ void light_shade_blend(const MarchParams& P, f3 w, v2f zw, v2f gxy, const TfFetch& tq, float4& dst)
{
    shade_blend_packed<UNI>(mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]), mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                            mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 2.5f, 0.5f, w, zw, gxy, tq, dst);
}

// ZSKIP (the host has verified what exact empty-space skipping needs: finite colour table and light, SKIP kernels only):
// when the opacity of EVERY ray of the packet is exactly 0 at this step, the blend is the identity for all of them (rgb
// finite, rgb * 0 = 0, dst + (1 - dst.a) * 0 = dst) and the table texels, the gradient, the shade and the blend are not
// computed -- the cells of an active brick that lie in air, before the rays reach the body.  One vote per step.
template <int V, bool OFF32, bool OTF = false, bool ZSKIP = false, bool LTF = false>
__device__ __forceinline__ void sample_and_blend(const MarchParams& P, f3 p, f3 w, float4& dst, f3 start, float ss)
{
    if constexpr (V == V_BASIC && ZSKIP) {
        const float density = tex3_a<OFF32>(P.vol[0], p);
        if (vr_ballot(!opacity_is_zero(P, density)) == 0) return;
        const TfSample t = tf_lookup0<LTF>(P, density);
        blend(t.rgb, t.opacity, dst);
    } else if constexpr (V == V_LIGHT) {
        // sample_src<V_LIGHT> + blend, written on (x, y) / (r, g) register pairs from the interpolation to the blend so
        // that the packed instructions need no shuffling; per component the operations and their order are those of
        // normalize3 / shade / blend.
        Fetch4 q;
        float fx, fy, fz;
        if constexpr (OTF) fetch_rgba_otf<OFF32>(P.vol[0], p, q, fx, fy, fz);
        else fetch_rgba<OFF32>(P.vol[0], p, q, fx, fy, fz);
        const v2f zw = interp_zw(q, fx, fy, fz);  // (gradient z, density)
        if constexpr (ZSKIP) {
            if (vr_ballot(!opacity_is_zero(P, zw.y)) == 0) return;
        }
        const TfFetch tq = tf_fetch0<LTF>(P, zw.y);
        const v2f gxy = interp_xy(q, fx, fy, fz);
        light_shade_blend(P, w, zw, gxy, tq, dst);
    } else {
        if constexpr (ZSKIP && (V == V_VOLUME_MASK || V == V_THREE_FILES || V == V_LIGHT_INSHADER)) {
            // the same vote for the other shaders whose opacity is the CT table's alone: all the step's fetches are issued as
            // sample_src issues them, so a packet that does sample waits no longer than before; a packet in air saves the
            // table look-ups and the arithmetic (and, for the in-shader gradient, its six further density fetches)
            if constexpr (V == V_VOLUME_MASK) {
                const float4 ct = tex3_rgba<OFF32>(P.vol[2], p);
                float4 mask;
                float rt;
                bool any_masked;
                fetch_mask_and_dose<OFF32>(P, p, mask, rt, any_masked);
                const bool inert = !(mask.x > 0.0f || mask.y > 0.0f || mask.z > 0.0f) && opacity_is_zero(P, ct.w);
                if (vr_ballot(!inert) == 0) return;
                const Src s = src_volume_mask<LTF>(P, w, mask, rt, ct, any_masked);
                blend(s.rgb, s.a, dst);
            } else if constexpr (V == V_THREE_FILES) {
                const float ct = tex3_a<OFF32>(P.vol[0], p);
                const float rt = tex3_a<OFF32>(P.vol[1], p);
                if (vr_ballot(!opacity_is_zero(P, ct)) == 0) return;
                const Src s = src_three_files<LTF>(P, ct, rt);
                blend(s.rgb, s.a, dst);
            } else {
                const float density = tex3_a<OFF32>(P.vol[0], p);
                if (vr_ballot(!opacity_is_zero(P, density)) == 0) return;
                const Src s = src_inshader<OFF32, LTF>(P, p, w, ss, density);
                blend(s.rgb, s.a, dst);
            }
            return;
        }
        const Src s = sample_src<V, OFF32, OTF, LTF>(P, p, w, start, dst.w, ss);
        blend(s.rgb, s.a, dst);
    }
}

// ------------------------------------------------------------------------------------------------ work mapping
// The launch walks 64x64 screen tiles (the multi-GPU ownership granule); a workgroup is a 16x16 pixel block
// of a tile, a wavefront an 8x8 packet of it.  blockIdx is de-interleaved over the 8 XCDs so that each XCD's
// L2 sees a contiguous run of neighbouring blocks (blocks b and b+8 share an XCD; performance only).
struct PixelSlot {
    int px, py;      // screen pixel
    int out_index;   // where dst goes (frame index, or packed-tile index)
    bool active;     // inside the viewport and inside the launch
    bool in_launch;  // the block's tile exists (the grid is padded to 8 x 16 blocks)
};

// group g of 8 workgroups -> (its frame, its group index within the frame) for a launch of n frames (MarchBatch)
__device__ __forceinline__ unsigned batch_group(unsigned g, unsigned n)
{
    return n == 1 ? g : (n == 2 ? g >> 1 : (n == 3 ? g / 3u : g >> 2));
}

// the logical block this workgroup works on (wave-uniform; P.order is a permutation of the frame's block indices)
__device__ __forceinline__ int logical_block(const MarchParams& P)
{
    const unsigned b = (batch_group(blockIdx.x >> 3, P.batch_n) << 3) | (blockIdx.x & 7u);  // index within the frame
    if (P.order == nullptr) return (int)b;
    return (int)__builtin_amdgcn_readfirstlane((int)P.order[b]);
}

// The parameters of the frame this workgroup belongs to (wave-uniform).  BATCH = false (launches of ONE frame) addresses
// frame[0] statically: the compiler then loads the kernel arguments once, up front, as it does for a plain by-value
// argument -- behind a computed address it re-loads them inside the march loop instead, which costs a frame that waits for
// its longest ray chains 5 % (C3 one frame at a time: 0.58 -> 0.61 ms).
template <bool BATCH>
__device__ __forceinline__ const MarchParams& frame_params(const MarchBatch& B)
{
    if constexpr (!BATCH) return B.frame[0];
    const unsigned g = blockIdx.x >> 3, n = B.n_frames;
    return B.frame[g - batch_group(g, n) * n];
}

// (lb = logical block, wpb = wavefronts per block, wib = this wavefront's index in its block)
__device__ __forceinline__ PixelSlot map_pixel_at(const MarchParams& P, int lb, int wpb, int wib)
{
    PixelSlot s;
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD; speed only, never
    // correctness).  The 16 blocks of one 64x64 tile stay on ONE XCD (their rays traverse neighbouring voxels:
    // shared L2 lines), while consecutive tiles go to different XCDs so that every XCD gets an even share of the
    // heavy (volume-covered) and the empty parts of the screen.
    // A block is 1 or 4 wavefronts (blockDim.x 64 / 256); a tile is 64 packets of 8x8 pixels either way, packet
    // pk = 4 * (16x16 sub-block) + quadrant.
    const int xcd = lb & 7, q = lb >> 3;
    const int bpt = 64 / wpb;  // blocks per tile
    int n = xcd + 8 * (q / bpt);  // ordinal of the owned tile this block works on
    int pk = (q % bpt) * wpb + wib;
    if (P.xcd_mode == 2 && wpb == 1) {
        // an even sample of the screen for every XCD at the grain of 16x16 sub-blocks: an XCD gets two whole sub-blocks (2 x 4
        // packets) of every tile, so the four packets of a sub-block -- neighbouring rays, neighbouring voxels -- share an L2.
        // Same frame time as the finest grain below, a fifth to a quarter less fabric traffic (C3 1.43 -> 1.16 GB, noisy air
        // 5.2 -> 4.3 GB: gpurun_out/r5i); whole tiles per XCD (mode 0) halve the traffic and cost 9 % (the XCDs' work differs).
        n = lb / 64;
        const int l = lb % 64, hi = l >> 3;
        pk = (((l & 7) | ((hi & 1) << 3)) << 2) | (hi >> 1);
    } else if (P.xcd_mode != 0) {  // consecutive blocks of a tile on consecutive XCDs: every XCD gets an even sample of the screen
        n = lb / bpt;
        pk = (lb % bpt) * wpb + wib;
    }
    const int sub = pk >> 2;
    const bool in_launch = n < P.n_tiles;
    const int t = P.rank + n * P.world;
    const int ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
    const int wave = pk & 3, lane = threadIdx.x & 63;
    const int lx = ((wave & 1) << 3) + (lane & 7);
    const int ly = ((wave >> 1) << 3) + (lane >> 3);
    const int tpx = ((sub & 3) << 4) + lx, tpy = ((sub >> 2) << 4) + ly;  // pixel inside the tile
    s.px = tx * kTile + tpx;
    s.py = ty * kTile + tpy;
    s.in_launch = in_launch;
    s.active = in_launch && (s.px < P.W) && (s.py < P.H) && (P.only_tile < 0 || P.only_tile == n);
    s.out_index = P.packed ? (n * (kTile * kTile) + tpy * kTile + tpx) : (s.py * P.W + s.px);
    return s;
}
__device__ __forceinline__ PixelSlot map_pixel(const MarchParams& P)
{
    return map_pixel_at(P, logical_block(P), (int)(blockDim.x >> 6), (int)(threadIdx.x >> 6));
}

// Exact empty-space test for the brick (kBrickCells^3 cells) that contains the base cell of p (vol[0]).
// A brick whose maximum density bm (over every voxel its cells can touch) maps into the zero prefix of the
// opacity table yields opacity == 0 exactly for every sample inside it, and blending (rgb*0, 0) leaves dst
// bit-identical (rgb is finite: the host checks tables and light).  Interpolated densities can exceed bm by a
// few ulps, which moves the table index by at most one: hence the +2 (floor + 1 neighbour + 1 margin).
// bm <= 0 (all-zero / negative cells) only ever addresses opacity[0].  NaN bm fails every comparison -> active.
template <bool FMED = true>
__device__ __forceinline__ int brick_of(const MarchParams& P, f3 p)
{
    // brick coordinate of the base cell: clamp(floor(p*n - 0.5), 0, n-1) >> kBrickShift.  Scaling by 1/c (c = kBrickCells, a
    // power of two) is exact and commutes with f32 rounding, and floor(floor(x)/c) == floor(x/c), so this is the same
    // integer as clamp(floor(p*(n/c) - 1/(2c)), 0, (n-1) >> kBrickShift).
    int bx, by, bz;
    if constexpr (FMED) {
        // The clamp is taken in float (one v_med3_f32; a NaN comes out as 0 either way) and truncation of the clamped,
        // non-negative value is its floor: four instructions per axis.
        // (mad: the same rounding(s) as the cell's own coordinate p * n - 0.5 -- scaling by 1/8 commutes with either)
        bx = (int)__builtin_amdgcn_fmed3f(mad(p.x, P.bsx, -kBrickHalf), 0.0f, (float)(P.bnx - 1));
        by = (int)__builtin_amdgcn_fmed3f(mad(p.y, P.bsy, -kBrickHalf), 0.0f, (float)(P.bny - 1));
        bz = (int)__builtin_amdgcn_fmed3f(mad(p.z, P.bsz, -kBrickHalf), 0.0f, (float)(P.bnz - 1));
    } else {
        // (integer clamps: the 64-bit-address kernels sit at the register limit of five waves per SIMD and the float
        // form costs them four more)
        bx = clampi((int)floorf(mad(p.x, P.bsx, -kBrickHalf)), 0, P.bnx - 1);
        by = clampi((int)floorf(mad(p.y, P.bsy, -kBrickHalf)), 0, P.bny - 1);
        bz = clampi((int)floorf(mad(p.z, P.bsz, -kBrickHalf)), 0, P.bnz - 1);
    }
    return __mul24(__mul24(bz, P.bny) + by, P.bnx) + bx;  // < 2^24 bricks per axis pair: 24-bit multiplies are exact
}
// distance-field byte of brick `bid` (SGPR base + 32-bit offset addressing)
__device__ __forceinline__ unsigned dist_at(const MarchParams& P, int bid)
{
    return *(reinterpret_cast<const unsigned char*>(P.brick_dist) + (unsigned)bid);
}
// record of brick `bid`: the table is far below 4 GiB, so SGPR base + 32-bit byte offset addressing
__device__ __forceinline__ float2 brick_record(const MarchParams& P, int bid)
{
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(P.bricks) + ((unsigned)bid << 3));
}
__device__ __forceinline__ bool brick_inert(const MarchParams& P, float2 rec)
{
    // VOLUME_MASK: a mask sample with r, g or b > 0 switches to the RT table; if every mask voxel the brick can touch
    // has max(r,g,b) <= 0 the interpolated channels are <= 0 too, so the CT opacity is the one that is blended
    if (P.use_rgb && !(rec.y <= 0.0f)) return false;
    if (rec.x <= 0.0f) return P.tf_zero_prefix >= 0;
    return floorf(rec.x * (float)P.tf[0].res_o - 0.5f) + 2.0f <= (float)P.tf_zero_prefix;
}

// Per-block sums of (composited samples, covered pixels, fetched samples): plain stores, no atomics.  (One atomic
// triple per wavefront on three shared addresses -- ~37 k device-scope atomics per 1080p frame -- serialised at the
// memory side and cost 0.3 ms per frame.)  sum_block_counts_kernel adds the blocks up.  Every thread of the block
// must call this (it contains a barrier).
__device__ __forceinline__ void store_block_counts(const MarchParams& P, unsigned blends, unsigned covered, unsigned fetched,
                                                   unsigned long long t_start)
{
    __shared__ unsigned long long part[4][3];
    unsigned long long packed_cnt = ((unsigned long long)covered << 40) | (unsigned long long)blends;
    unsigned long long fetched_cnt = fetched;
    unsigned crit = fetched;  // most samples any one ray of the wavefront fetched ~ its chain of dependent iterations
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        packed_cnt += __shfl_down(packed_cnt, off, 64);
        fetched_cnt += __shfl_down(fetched_cnt, off, 64);
        crit = max(crit, (unsigned)__shfl_down((int)crit, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = packed_cnt;
        part[threadIdx.x >> 6][1] = fetched_cnt;
        part[threadIdx.x >> 6][2] = crit;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long pc = 0, fc = 0, cr = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
            pc += part[i][0];
            fc += part[i][1];
            cr = part[i][2] > cr ? part[i][2] : cr;
        }
        unsigned long long* o = P.block_counts + (size_t)logical_block(P) * kBlockRecord;  // records are per LOGICAL block
        o[0] = pc & ((1ull << 40) - 1);
        o[1] = pc >> 40;
        o[2] = fc;
        // block trace (vr_last_block_trace): 100 MHz clock at entry / exit, HW_ID | XCC_ID << 32
        o[3] = t_start;
        o[4] = wall_clock64();
        // HW_ID | XCC_ID << 32 | longest per-ray sample chain of the workgroup << 40
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
               ((unsigned long long)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0x7fu) << 32) | (cr << 40);  // (bit 39: kRecSplit)
    }
}

// store_block_counts for a block of ONE wavefront that is not a workgroup (persistent wavefronts, vr_pw.h; mixed lanes per
// ray, vr_mixed.h): no barrier, no LDS.  `rec` = the record's index; split = the packet's rays were marched by two
// wavefronts (vr_mixed.h): this is the first half's record, the second half's is at rec + (logical blocks of the launch)
// (kRecSplit in word 5 tells the readers -- sum_block_counts_kernel, order_blocks_kernel, vr_last_block_trace -- to add it).
__device__ __forceinline__ void store_wave_counts(const MarchParams& P, int rec, unsigned blends, unsigned covered, unsigned fetched,
                                                  unsigned long long t_start, bool split = false)
{
    unsigned long long packed_cnt = ((unsigned long long)covered << 40) | (unsigned long long)blends;
    unsigned long long fetched_cnt = fetched;
    unsigned crit = fetched;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        packed_cnt += __shfl_down(packed_cnt, off, 64);
        fetched_cnt += __shfl_down(fetched_cnt, off, 64);
        crit = max(crit, (unsigned)__shfl_down((int)crit, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = P.block_counts + (size_t)rec * kBlockRecord;
        o[0] = packed_cnt & ((1ull << 40) - 1);
        o[1] = packed_cnt >> 40;
        o[2] = fetched_cnt;
        o[3] = t_start;
        o[4] = wall_clock64();
        o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
               ((unsigned long long)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)) & 0x7fu) << 32) |
               (split ? kRecSplit : 0ull) | ((unsigned long long)crit << 40);
    }
}

// ---- exact empty-space leaping -----------------------------------------------------------------------------
// p advances by one ROUNDED addition of `s` per step.  While p keeps its sign and binary exponent its ulp U is
// constant, p = n*U and s = (k + f)*U with |f| <= 1/2, so every addition moves p by the same whole number of ulps
// (k, or from the second addition on k or k+1 in the exact-tie case f = 1/2, where round-to-even makes the result
// even and the increment constant afterwards).  Hence, with x1 = fl(x + s) and x2 = fl(x1 + s),
//      bits(x after m additions) = bits(x1) + (m - 1) * (bits(x2) - bits(x1)),        m >= 1,
// provided x, x1, x2 and the result share sign and exponent (the sequence is monotone, so the end points suffice).
// That makes a jump over m steps O(1) and bit-identical to m single steps.  Returns false when the condition fails
// (the caller then takes a single ordinary step).  leap_plan also returns how many steps fit before the binade edge.
struct LeapCoord {
    int b0, b1, d;  // bits of x and of x1 = fl(x+s); ulps per step from the second addition on
    float mmax;     // largest m (as float, conservative) for which the result keeps x's sign and exponent
};
__device__ __forceinline__ LeapCoord leap_plan(float x, float s)
{
    const float x1 = x + s, x2 = x1 + s;
    LeapCoord c;
    c.b0 = __float_as_int(x);
    c.b1 = __float_as_int(x1);
    const int b2 = __float_as_int(x2);
    c.d = b2 - c.b1;
    // room, in ulps, between x1 and the edge of its binade in the direction the MAGNITUDE moves
    const int mag1 = c.b1 & 0x7FFFFFFF, mag2 = b2 & 0x7FFFFFFF;
    const int dm = mag2 - mag1;
    const int room = dm > 0 ? (0x7FFFFF - (mag1 & 0x7FFFFF)) : (mag1 & 0x7FFFFF);
    const bool same = (((c.b0 ^ c.b1) | (c.b0 ^ b2)) & (int)0xFF800000) == 0;  // x, x1, x2 in one binade
    // m - 1 further increments of |dm| ulps must fit into `room`
    c.mmax = !same ? 0.0f : (dm == 0 ? 1.0e9f : 1.0f + (float)room * (__builtin_amdgcn_rcpf((float)abs(dm)) * 0.999f));
    return c;
}
__device__ __forceinline__ int leap_apply(const LeapCoord& c, int m, float& out)
{
    const int r = c.b1 + (m - 1) * c.d;
    out = __int_as_float(r);
    return (((c.b0 ^ r) & (int)0xFF800000) == 0) ? 1 : 0;  // final guard: sign and exponent unchanged
}

// Number of steps (>= 0) a ray at p certainly stays inside the cube of bricks within Chebyshev distance D-1 of its
// own brick -- all of them inert by construction of the distance field.  Same conservative span arithmetic as the
// brick definition: base-cell index floor(p*n - 0.5) in [8*lo, 8*(hi+1)), shrunk by 0.01 cell; edge bricks also own
// the clamped cells beyond the volume (no bound there: the caller keeps the leap inside the clip range).
__device__ __forceinline__ float leap_axis(float p, float s, float bs, int nb, int k, float inv_n)
{
    if (s == 0.0f) return 1.0e9f;
    const int b = clampi((int)floorf(mad(p, bs, -kBrickHalf)), 0, nb - 1);
    float room;
    if (s > 0.0f) {
        const int hi = b + k;  // last inert brick index in the direction of travel
        if (hi >= nb - 1) return 1.0e9f;
        room = ((float)((hi + 1) << kBrickShift) + 0.49f) * inv_n - p;
    } else {
        const int lo = b - k;
        if (lo <= 0) return 1.0e9f;
        room = p - ((float)(lo << kBrickShift) + 0.51f) * inv_n;
    }
    return room * (__builtin_amdgcn_rcpf(fabsf(s)) * 0.999f);  // approximate reciprocal, scaled down: never too large
}


// Number of steps a ray certainly stays inside IsInSampleCoords: positions p_0 .. p_{n-1} of the accumulation
// p_{k+1} = fl(p_k + step) are inside [b0, b1] on every axis.
//  - behind the ray nothing can happen: rounding is monotone, so a coordinate never moves against the sign of its step;
//  - ahead, p_k <= p_0 + k * (|step| + e): each rounded addition errs by at most half an ulp of its result, i.e. by at most
//    2^-24 * M with M = the largest magnitude in play (e below is 2^-23 * M: twice that).  For an axis the ray hardly moves
//    along (|step| of the order of e) the ACCUMULATED rounding, not the step, decides when the bound is crossed -- a margin
//    counted in steps alone is wrong there (found by tests/test_scale_gpu.py with 4 000-step rays).
// The quotient is taken 0.1 % + 2 steps short to cover its own evaluation in f32.  NaN / negative -> 0.
__device__ __forceinline__ float steps_inside_axis(float p, float s, float b0, float b1)
{
    const float e = fmaxf(fmaxf(fabsf(b0), fabsf(b1)), fabsf(p)) * 1.1920929e-7f;
    if (s > 0.0f) return (b1 - p) / (s + e);
    if (s < 0.0f) return (p - b0) / (e - s);
    return 3.0e38f;  // fl(p + 0) == p: the coordinate never changes
}
__device__ __forceinline__ int steps_inside(f3 p, f3 step, float bx0, float by0, float bz0, float bx1, float by1, float bz1)
{
    const bool in0 = p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
    const float f = fminf(fminf(steps_inside_axis(p.x, step.x, bx0, bx1), steps_inside_axis(p.y, step.y, by0, by1)),
                          fminf(steps_inside_axis(p.z, step.z, bz0, bz1), 1.0e6f)) * 0.999f - 2.0f;
    return (in0 && f > 0.0f) ? (int)f : 0;
}

#ifndef VR_APPROACH
#define VR_APPROACH 1
#endif
constexpr bool kApproach = VR_APPROACH != 0;  // the approach loop in front of the march loops (-DVR_APPROACH=0: A/B builds)
constexpr int kApproachMax = 1024;            // identity steps one iteration of it may take
// Steps [k0, k1] outside of which the ray's positions certainly lie outside the box [lo, hi] (uvw; the caller's box carries its own
// margin of a whole brick, far above what the rounded additions of n_steps steps can drift -- 2^-23 per step -- and what this
// quotient's evaluation in f32 can be off by: 0.1 % + 2 steps are added on top).  A ray that misses the box: k0 = INT_MAX, k1 = -1.
// NaN anywhere -> no step is excluded.
__device__ __forceinline__ void steps_near_box(f3 p, f3 s, const float* box, int n_steps, int& k0, int& k1)
{
    const float drift = (float)n_steps * 4.8e-7f;
    float t0 = 0.0f, t1 = (float)n_steps;
    bool miss = false;
    const float pp[3] = {p.x, p.y, p.z}, ss[3] = {s.x, s.y, s.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float lo = box[a] - drift, hi = box[3 + a] + drift;
        if (ss[a] == 0.0f) {
            miss = miss || pp[a] < lo || pp[a] > hi;
        } else {
            const float inv = __builtin_amdgcn_rcpf(ss[a]);  // (1 ulp: the margins below are a thousand times that)
            const float ta = (lo - pp[a]) * inv, tb = (hi - pp[a]) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));  // (fminf / fmaxf drop a NaN operand)
            t1 = fminf(t1, fmaxf(ta, tb));
        }
    }
    miss = miss || t0 > t1 + 4.0f;
    k0 = miss ? 0x7fffffff : max((int)fminf(t0 * 0.999f, 2.0e9f) - 2, 0);
    k1 = miss ? -1 : (int)fminf(t1 * 1.001f + 3.0f, 2.0e9f);
}

// OTF (V_LIGHT only): the corners' gradients are derived from the density plane (fetch_rgba_otf) instead of read from
// the vec4 voxels; the host asks for it when the volume's .rgb is verified to be PreComputeGradient(false) of its .a.
// The whole march of one ray (ray set-up, per-pixel prologue, the loop): what a lane does for its pixel `slot`.  Shared by
// march_kernel (one packet per wavefront of the grid) and march_pw_kernel (vr_pw.h: persistent wavefronts that take packet
// after packet from a queue; LTF = transfer-function slot 0 read from the workgroup's LDS).
template <int V, bool OFF32, bool SKIP, int LEAP, bool OTF, bool LTF>
__device__ __forceinline__ void march_packet(const MarchParams& P, const PixelSlot& slot, float4& dst, unsigned& blends,
                                             unsigned& covered, unsigned& fetched)
{
    if (slot.active && slot.px >= P.rect[0] && slot.px <= P.rect[2] && slot.py >= P.rect[1] && slot.py <= P.rect[3]) {
        Ray ray = setup_ray(P, slot.px, slot.py);
        if (ray.hit) {
            covered = 1;
            f3 diff = mk3(ray.end.x - ray.start.x, ray.end.y - ray.start.y, ray.end.z - ray.start.z);
            f3 dir = normalize3s(diff);
            float ray_len = length3s(diff);
            if (P.fragment_mode == 1) {
                dst = make_float4(fabsf(dir.x), fabsf(dir.y), fabsf(dir.z), 1.0f);
            } else if (P.fragment_mode == 2) {
                dst = make_float4(ray.start.x, ray.start.y, ray.start.z, 1.0f);
            } else if (P.fragment_mode == 3) {
                dst = make_float4(ray.end.x, ray.end.y, ray.end.z, 1.0f);
            } else if (P.fragment_mode == 4) {
                dst = make_float4(0.5f * (ray.world0.x / 1.0f) + 0.5f, -0.5f * (ray.world0.y / 1.0f) + 0.5f, 0.0f, 1.0f);
            } else {
                float step_size = P.step_size;
                f3 wstep = mk3(0.0f, 0.0f, 0.0f);
                if constexpr (V == V_LIGHT || V == V_LIGHT_INSHADER) {  // CalculateWorldStep before the override
                    wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.5f));
                    wstep.z = wstep.z * (-1.0f);
                }
                if (P.toggle_varstep == 1) step_size = ray_len / (float)P.steps_count;
                f3 p = ray.start;
                if (P.toggle_jitter == 1) {
                    float j = jitter((float)slot.px + 0.5f, (float)slot.py + 0.5f);
                    p = mk3(p.x + (dir.x * step_size) * j, p.y + (dir.y * step_size) * j, p.z + (dir.z * step_size) * j);
                }
                f3 step = mk3(dir.x * step_size, dir.y * step_size, dir.z * step_size);
                if constexpr (V == V_MULTI_CTRT || V == V_ILLUSTRATIVE) {  // CalculateWorldStep after the override
                    wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.7f));
                    wstep.z = wstep.z * (-1.0f);
                }
                if constexpr (V == V_VOLUME_MASK || V == V_THREE_FILES) wstep = step;
                f3 w = ray.world0;
                const float bx0 = P.bmin[0], by0 = P.bmin[1], bz0 = P.bmin[2];
                const float bx1 = P.bmax[0], by1 = P.bmax[1], bz1 = P.bmax[2];
                // Steps [0, n_inside) are certainly inside IsInSampleCoords (steps_inside): the six compares are skipped there.
                const int n_inside = steps_inside(p, step, bx0, by0, bz0, bx1, by1, bz1);
                // ---- the march loop ---------------------------------------------------------------------------------
                // Empty-space test: one byte per brick (distance field), looked up for the positions AHEAD of the ray
                // (the same rounded additions the advance performs, so they are the positions later iterations really
                // have).  Loads complete in order, so WHERE a byte is waited for decides what else is waited for:
                //   D   byte of p            always arrived
                //   Dn  byte of p + step     arrived for rays that sampled last iteration (`have`)
                //   Dq  byte of p + 2 step   (of p + step at the loop head for the other rays): requested one
                //                            iteration earlier, possibly still in flight, only ever read behind a wait
                // Rays that did not sample last iteration take their byte in a block of their own at the loop head; a
                // wavefront whose rays are all inside tissue skips that block, and the eight corner loads it has issued
                // for the next step (kPipe) stay in flight from the shading of one step to the interpolation of the next.
                constexpr bool kRun = SKIP && (LEAP >= 2);                              // wave-uniform runs of identity steps
                constexpr bool kPipe = (V == V_LIGHT || V == V_BASIC) && LEAP != 3 && LEAP != 1 && !OTF;  // corner prefetch
                unsigned D = 0, Dn = 0, Dq = 0, Dn2 = 0;
                bool have = false;   // the previous iteration sampled: corners of p requested (F4 / F1), Dn arrived
                bool stale = false;  // ... and the one before did, this one did not: Dq was not requested
                Fetch4 F4;
                Fetch1 F1;
                float wfx = 0.0f, wfy = 0.0f, wfz = 0.0f;
                const int lim = min(n_inside, P.steps_count);  // runs stay inside the provably-in-box prefix
                int i0 = 0;  // the step the loop below starts at
                if constexpr (kRun && kApproach) {
                    // THE APPROACH (march_p2_kernel's, vr_p2.h, with votes instead of a wave minimum: the lanes without a ray are
                    // switched off here): until a ray of the packet stands in an active brick, one byte per ray and the identity
                    // steps it allows -- the largest power of two every ray allows -- as plain rounded additions; rays outside the
                    // box of the active bricks (abox) ask for nothing.  The packets that never meet an active brick end here; the
                    // world position follows only when a ray may still sample.  Every exit is wave-uniform.
                    const float vmax = fmaxf(fmaxf(fabsf(step.x) * P.bsx, fabsf(step.y) * P.bsy), fabsf(step.z) * P.bsz);
                    const float lc = 0.999f / vmax;
                    int k0, k1;
                    steps_near_box(p, step, P.abox, P.steps_count, k0, k1);
                    int skipped = 0;
                    while (vr_ballot(i0 >= lim) == 0) {
                        const bool near = i0 >= k0 && i0 <= k1;
                        int safe = kApproachMax;
                        if (vr_ballot(near) != 0) {
                            const unsigned d = dist_at(P, brick_of<OFF32>(P, p));
                            if (vr_ballot(near && d == 0u) != 0) break;
                            const int sf = 1 + max((int)fminf(((float)d - (1.0f + kBrickHalf)) * lc, (float)kApproachMax), 0);
                            safe = near ? sf : safe;
                        }
                        if (i0 < k0) safe = min(safe, k0 - i0);
                        safe = min(safe, lim - i0);  // (>= 1)
                        int sw = 1;
                        while (sw < kApproachMax && vr_ballot(safe < 2 * sw) == 0) sw *= 2;
                        for (int k = 0; k < sw; ++k) p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                        i0 += sw;
                        blends += (unsigned)sw;
                        skipped += sw;
                    }
                    if constexpr (V != V_BASIC && V != V_TF_CALIB) {
                        if (skipped != 0 && vr_ballot(i0 <= k1) != 0)
                            for (int k = 0; k < skipped; ++k) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                }
                if constexpr (SKIP) {
                    D = dist_at(P, brick_of<OFF32>(P, p));
                    Dq = dist_at(P, brick_of<OFF32>(P, mk3(p.x + step.x, p.y + step.y, p.z + step.z)));
                    asm volatile("" : "+v"(D));  // wait for D here; Dq stays in flight
                }
                const int prio_q1 = P.steps_count >> 2, prio_q2 = P.steps_count >> 1, prio_q3 = prio_q1 + prio_q2;
                // kRun: steps a ray at distance-field value D can take while it certainly stays within D-1 bricks of
                // its brick on every axis = (D - 1 - 1/16) / (largest per-step move in brick units), 0.1 % short
                float leap_c = 0.0f;
                if constexpr (kRun) {
                    const float vmax = fmaxf(fmaxf(fabsf(step.x) * P.bsx, fabsf(step.y) * P.bsy), fabsf(step.z) * P.bsz);
                    leap_c = 0.999f / vmax;  // vmax 0 -> inf (capped below), NaN -> n_inside is 0 and nothing leaps
                }
                for (int i = i0; i < P.steps_count;) {
                    f3 pn = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                    const f3 pq = mk3(pn.x + step.x, pn.y + step.y, pn.z + step.z);
                    if constexpr (SKIP) {
                        if (!have) {
                            if constexpr (kPipe) {
                                if (stale) Dq = dist_at(P, brick_of<OFF32>(P, pn));  // first iteration after leaving tissue
                                stale = false;
                            }
                            // requested an iteration ago.  (The copy is spelled out so that it happens HERE and the new
                            // load can go into Dq's own register: a compiler-placed copy behind the load would wait.)
                            asm volatile("v_mov_b32 %0, %1" : "=v"(Dn) : "v"(Dq));
                            __builtin_amdgcn_sched_barrier(0);
                            Dq = dist_at(P, brick_of<OFF32>(P, pq));
                        }
                    }
                    if constexpr (kRun) {
                        // Wave-uniform run of identity steps: when EVERY ray of the packet that is still marching sits
                        // at least 4 safe steps inside inert bricks, all of them take the same number of plain rounded
                        // additions back to back (nothing else per step), then look their brick up again.  The rays
                        // of a packet stay at one step index, so their samples keep sharing cache lines.
                        int m = 0;
                        const bool far = vr_ballot(D < 2) == 0;  // (one vote decides for a packet that is sampling)
                        if (far) m = min((int)fminf(((float)D - (1.0f + kBrickHalf)) * leap_c, 64.0f), lim - i - 1);
                        if (far && vr_ballot(m < 4) == 0) {
                            int mw = 4;
                            if (vr_ballot(m < 8) == 0) {
                                mw = 8;
                                if (vr_ballot(m < 16) == 0) {
                                    mw = 16;
                                    if (vr_ballot(m < 32) == 0) mw = vr_ballot(m < 64) == 0 ? 64 : 32;
                                }
                            }
                            for (int k = 0; k < mw; k += 4) {  // mw is a multiple of 4: one branch per four steps
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                                    if constexpr (V != V_BASIC && V != V_TF_CALIB)
                                        w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                                }
                            }
                            i += mw;
                            blends += (unsigned)mw;
                            D = dist_at(P, brick_of<OFF32>(P, p));
                            Dq = dist_at(P, brick_of<OFF32>(P, mk3(p.x + step.x, p.y + step.y, p.z + step.z)));
                            asm volatile("" : "+v"(D));  // (and everything older); Dq stays in flight
                            have = false;
                            stale = false;
                            continue;
                        }
                    }
                    bool inb = true;
                    if (i >= n_inside)
                        inb = p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
                    int adv = 1;  // steps this iteration advances by
                    if (inb) {
                        if (!SKIP || D == 0) {
                            if (P.prio_mode == 1 && (__builtin_amdgcn_readfirstlane(i) & 7) == 0) {
                                // longest-remaining-path-first: the frame is done when its longest ray is, so wavefronts
                                // whose rays still have far to go get the issue slots first (4 levels, by quarters of
                                // stepsCount; re-evaluated every 8th step; speed only)
                                const int r = lim - i;
                                if (vr_ballot(r > prio_q3) != 0) __builtin_amdgcn_s_setprio(3);
                                else if (vr_ballot(r > prio_q2) != 0) __builtin_amdgcn_s_setprio(2);
                                else if (vr_ballot(r > prio_q1) != 0) __builtin_amdgcn_s_setprio(1);
                                else __builtin_amdgcn_s_setprio(0);
                            }
                            if constexpr (kPipe) {
                                // Software pipeline over the steps of a ray: the corners of THIS step were requested one
                                // iteration ago (or are requested now, on entering tissue).  The next step's corners are
                                // requested after this step's table texels, so waiting for the texels leaves those eight
                                // loads in flight; the look-ahead byte is taken before them for the same reason.
                                unsigned R = 0;
                                if constexpr (SKIP) R = dist_at(P, brick_of<OFF32>(P, pq));
                                TfFetch tq = {};
                                v2f zw = v2f{0.0f, 0.0f}, gxy = zw;
                                bool all_zero = false;  // the per-step vote of sample_and_blend (ZSKIP)
                                if constexpr (V == V_LIGHT) {
                                    if (!have) fetch_rgba<OFF32>(P.vol[0], p, F4, wfx, wfy, wfz);
                                    zw = interp_zw(F4, wfx, wfy, wfz);
                                    if constexpr (SKIP) all_zero = vr_ballot(!opacity_is_zero(P, zw.y)) == 0;
                                    if (!all_zero) {
                                        tq = tf_fetch0<LTF>(P, zw.y);
                                        gxy = interp_xy(F4, wfx, wfy, wfz);
                                    }
                                } else {
                                    if (!have) fetch_a<OFF32>(P.vol[0], p, F1, wfx, wfy, wfz);
                                    zw.y = interp_a(F1, wfx, wfy, wfz);
                                    if constexpr (SKIP) all_zero = vr_ballot(!opacity_is_zero(P, zw.y)) == 0;
                                    if (!all_zero) tq = tf_fetch0<LTF>(P, zw.y);
                                }
                                if constexpr (SKIP) {
                                    asm volatile("" : "+v"(R));
                                    Dn2 = R;
                                }
                                __builtin_amdgcn_sched_barrier(0);  // the old corners are dead here: same registers
                                if constexpr (V == V_LIGHT)
                                    fetch_rgba<OFF32>(P.vol[0], pn, F4, wfx, wfy, wfz);
                                else
                                    fetch_a<OFF32>(P.vol[0], pn, F1, wfx, wfy, wfz);
                                __builtin_amdgcn_sched_barrier(0);
                                have = true;
                                if (!all_zero) {
                                    if constexpr (!LTF) tf_pin(tq);  // (texels from LDS are waited for with their own counter)
                                    if constexpr (V == V_LIGHT) {
                                        light_shade_blend(P, w, zw, gxy, tq, dst);
                                    } else {
                                        const TfSample t = tf_finish(tq);
                                        blend(t.rgb, t.opacity, dst);
                                    }
                                }
                            } else {
                                sample_and_blend<V, OFF32, OTF, SKIP, LTF>(P, p, w, dst, ray.start, step_size);
                            }
                            ++fetched;
                            ++blends;
                            if (!can_blend<V>(dst.w)) break;  // cut-off reached: no later iteration can blend
                        } else {
                            // identity blend(s): the reference executes them, nothing changes and nothing is fetched
                            if constexpr (kPipe) {
                                stale = have;  // leaving tissue: back to the loop-head scheme
                                have = false;
                            }
                            if constexpr (LEAP == 1) {
                                if (D >= 2 && i + 3 < lim) {
                                    const DevVolume& v = P.vol[P.skip_vol];
                                    const int k = (int)D - 1;
                                    float mf = fminf(fminf(leap_axis(p.x, step.x, P.bsx, P.bnx, k, 1.0f / (float)v.nx),
                                                           leap_axis(p.y, step.y, P.bsy, P.bny, k, 1.0f / (float)v.ny)),
                                                     fminf(leap_axis(p.z, step.z, P.bsz, P.bnz, k, 1.0f / (float)v.nz), 255.0f));
                                    // every coordinate also bounds the leap by its room to the edge of its binade
                                    const LeapCoord cx = leap_plan(p.x, step.x), cy = leap_plan(p.y, step.y),
                                                    cz = leap_plan(p.z, step.z);
                                    mf = fminf(fminf(mf - 1.0f, cx.mmax), fminf(cy.mmax, cz.mmax));
                                    LeapCoord wx = cx, wy = cy, wz = cz;
                                    if constexpr (V != V_BASIC && V != V_TF_CALIB) {
                                        wx = leap_plan(w.x, wstep.x);
                                        wy = leap_plan(w.y, wstep.y);
                                        wz = leap_plan(w.z, wstep.z);
                                        mf = fminf(fminf(mf, wx.mmax), fminf(wy.mmax, wz.mmax));
                                    }
                                    const int m = min((int)mf, lim - i - 1);
                                    if (m >= 3) {
                                        f3 q, wq = w;
                                        int ok = leap_apply(cx, m, q.x) & leap_apply(cy, m, q.y) & leap_apply(cz, m, q.z);
                                        if constexpr (V != V_BASIC && V != V_TF_CALIB)
                                            ok = ok & leap_apply(wx, m, wq.x) & leap_apply(wy, m, wq.y) & leap_apply(wz, m, wq.z);
                                        if (ok) {  // all m steps are in-box identity blends: count them and land
                                            adv = m;
                                            pn = q;
                                            if constexpr (V != V_BASIC && V != V_TF_CALIB) w = wq;
                                            Dn = dist_at(P, brick_of<OFF32>(P, pn));
                                            Dq = dist_at(P, brick_of<OFF32>(P, mk3(pn.x + step.x, pn.y + step.y, pn.z + step.z)));
                                            asm volatile("" : "+v"(Dn));
                                        }
                                    }
                                }
                            }
                            blends += (unsigned)adv;
                        }
                    } else {
                        if constexpr (kPipe) {
                            stale = have;
                            have = false;
                        }
                        // p moves monotonically per component: once past the far bound it never returns
                        bool gone = (step.x >= 0.0f && p.x > bx1) || (step.x <= 0.0f && p.x < bx0) ||
                                    (step.y >= 0.0f && p.y > by1) || (step.y <= 0.0f && p.y < by0) ||
                                    (step.z >= 0.0f && p.z > bz1) || (step.z <= 0.0f && p.z < bz0);
                        if (gone) break;
                    }
                    p = pn;
                    D = Dn;
                    if constexpr (kPipe) {
                        if (have) Dn = Dn2;
                    }
                    if (adv == 1) {
                        if constexpr (V != V_BASIC && V != V_TF_CALIB) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                    i += adv;
                }
            }
        }
    }
}

template <int V, bool OFF32, bool SKIP, int LEAP, bool OTF = false, bool BATCH = false>
__global__ __launch_bounds__(256, VR_LIGHT_WAVES_PER_EU(V, OTF)) void march_kernel(const MarchBatch B)
{
    const MarchParams& P = frame_params<BATCH>(B);
    const unsigned long long t_start = wall_clock64();
    PixelSlot slot = map_pixel(P);
    float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned blends = 0, covered = 0, fetched = 0;
    // flavour 18: the workgroup's slot tables (DevVolume::lut, make_cell_lut) -- only a workgroup one of whose pixels can hit the
    // box fills them (three quarters of a frame's packets cannot: they end before they would read a table)
    if (B.frame[0].vol[0].lut) {
        const bool may_hit = slot.active && slot.px >= P.rect[0] && slot.px <= P.rect[2] && slot.py >= P.rect[1] && slot.py <= P.rect[3];
        if (__syncthreads_or(may_hit ? 1 : 0)) {
            lut_fill(B.frame[0].vol[0]);
            __syncthreads();
        }
    }

    march_packet<V, OFF32, SKIP, LEAP, OTF, false>(P, slot, dst, blends, covered, fetched);

    // packed-tile launches write every slot of an owned tile (pixels outside the viewport = 0)
    if (slot.active || (P.packed && slot.in_launch)) P.out[slot.out_index] = dst;

    store_block_counts(P, blends, covered, fetched, t_start);
}

#if !VR_FUSED  // auxiliary kernels (no multiply-adds of the per-sample kind): compiled once, in namespace vr
// One wavefront per brick of c = kBrickCells cells: maximum of .a over the voxels [c b, min(c b + c, n-1)]^3 (NaN if any
// voxel is NaN or infinite).
__global__ __launch_bounds__(64) void brick_max_kernel(const float4* __restrict__ vol, int nx, int ny, int nz, int bnx,
                                                       int bny, float2* __restrict__ out)
{
    const int b = blockIdx.x;
    const int bx = b % bnx, by = (b / bnx) % bny, bz = b / (bnx * bny);
    const int x0 = bx << kBrickShift, y0 = by << kBrickShift, z0 = bz << kBrickShift;
    const int ex = min(kBrickCells + 1, nx - x0), ey = min(kBrickCells + 1, ny - y0), ez = min(kBrickCells + 1, nz - z0);
    float m = -INFINITY, mc = -INFINITY;
    bool has_nan = false, has_nan_c = false;
    for (int t = threadIdx.x; t < ex * ey * ez; t += 64) {
        int lx = t % ex, ly = (t / ex) % ey, lz = t / (ex * ey);
        float4 v = vol[((size_t)(z0 + lz) * ny + (y0 + ly)) * nx + (x0 + lx)];
        // NaN or +-inf: samples next to such a voxel interpolate to NaN (inf * 0, inf - inf) whatever the others hold, and a
        // NaN density has a NaN opacity: the brick must stay active (a -inf would otherwise just lose the maximum)
        if (!(v.w - v.w == 0.0f)) has_nan = true;
        else if (v.w > m) m = v.w;
        if (v.x != v.x || v.y != v.y || v.z != v.z) has_nan_c = true;
        else mc = fmaxf(mc, fmaxf(v.x, fmaxf(v.y, v.z)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m = fmaxf(m, __shfl_down(m, off, 64));
        mc = fmaxf(mc, __shfl_down(mc, off, 64));
        // (every lane takes part in every shuffle: behind a short-circuiting `||` a lane that has already seen a NaN would
        // sit the shuffle out, and the lane reading from it would get nothing -- the flag of a NaN found by any lane but
        // lane 0 was lost that way until round 2, and a brick holding nothing else but that voxel was skipped)
        const int other = __shfl_down((int)has_nan, off, 64), other_c = __shfl_down((int)has_nan_c, off, 64);
        has_nan = has_nan || other != 0;
        has_nan_c = has_nan_c || other_c != 0;
    }
    if (threadIdx.x == 0) out[b] = make_float2(has_nan ? NAN : m, has_nan_c ? NAN : mc);
}

// ------------------------------------------------------------------------------------------------ aux kernels
__device__ __forceinline__ unsigned unorm8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (unsigned)floorf(v * 255.0f + 0.5f);
}

// Output merge over the white background, BGRA8Unorm (PipelineBuilder.cpp:142-147, fullscreen.wgsl:33-41).
__global__ void present_kernel(const float4* __restrict__ frag, uint32_t* __restrict__ bgra, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = frag[i];
    float a = s.w;
    float r = s.x * a + 1.0f * (1.0f - a);
    float g = s.y * a + 1.0f * (1.0f - a);
    float b = s.z * a + 1.0f * (1.0f - a);
    float oa = s.w * a + 1.0f * (1.0f - a);
    bgra[i] = unorm8(b) | (unorm8(g) << 8) | (unorm8(r) << 16) | (unorm8(oa) << 24);
}

// present_kernel reading the frame THROUGH the tile permutation: the root of the multi-GPU gather presents straight from the
// gathered, tile-major segments (gathered[r][n][64*64]), no assembled float frame in between (16 B read + 4 B written per
// pixel instead of an un-permute pass of 16 + 16 and a present pass of 16 + 4).  Same arithmetic as present_kernel.
__global__ void present_tiles_kernel(const float4* __restrict__ gathered, uint32_t* __restrict__ bgra, int W, int H, int tiles_x,
                                     int world, int tiles_per_rank_max)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int t = (y / kTile) * tiles_x + (x / kTile);
    int r = t % world, n = t / world;
    size_t src = ((size_t)r * tiles_per_rank_max + n) * (kTile * kTile) + (y % kTile) * kTile + (x % kTile);
    float4 s = gathered[src];
    float a = s.w;
    float rr = s.x * a + 1.0f * (1.0f - a);
    float g = s.y * a + 1.0f * (1.0f - a);
    float b = s.z * a + 1.0f * (1.0f - a);
    float oa = s.w * a + 1.0f * (1.0f - a);
    bgra[(size_t)y * W + x] = unorm8(b) | (unorm8(g) << 8) | (unorm8(rr) << 16) | (unorm8(oa) << 24);
}

// Root side of the image-tile gather: gathered[r][n][64*64] -> frame[y*W+x]
__global__ void unpack_tiles_kernel(const float4* __restrict__ gathered, float4* __restrict__ frame, int W, int H,
                                    int tiles_x, int world, int tiles_per_rank_max)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int t = (y / kTile) * tiles_x + (x / kTile);
    int r = t % world, n = t / world;
    size_t src = ((size_t)r * tiles_per_rank_max + n) * (kTile * kTile) + (y % kTile) * kTile + (x % kTile);
    frame[(size_t)y * W + x] = gathered[src];
}

// ... and the same un-permute for tiles that were presented where they were rendered (4 B per pixel: the multi-GPU loop gathers
// BGRA8 tiles instead of float tiles when only the presented frame is wanted)
__global__ void unpack_tiles_u32_kernel(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ frame, int W, int H, int tiles_x,
                                        int world, int tiles_per_rank_max)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    int t = (y / kTile) * tiles_x + (x / kTile);
    int r = t % world, n = t / world;
    size_t src = ((size_t)r * tiles_per_rank_max + n) * (kTile * kTile) + (y % kTile) * kTile + (x % kTile);
    frame[(size_t)y * W + x] = gathered[src];
}

// One block: adds the per-block counts of the last march launch up (out[0..2]).
__global__ __launch_bounds__(256) void sum_block_counts_kernel(const unsigned long long* __restrict__ in, int n_blocks,
                                                               unsigned long long* __restrict__ out)
{
    __shared__ unsigned long long part[4][3];
    unsigned long long a = 0, b = 0, f = 0;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        a += in[(size_t)i * kBlockRecord];
        b += in[(size_t)i * kBlockRecord + 1];
        f += in[(size_t)i * kBlockRecord + 2];
        if (in[(size_t)i * kBlockRecord + 5] & kRecSplit) {  // the packet's second half (vr_mixed.h)
            const size_t h = (size_t)(i + n_blocks) * kBlockRecord;
            a += in[h];
            b += in[h + 1];
            f += in[h + 2];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        f += __shfl_down(f, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = a;
        part[threadIdx.x >> 6][1] = b;
        part[threadIdx.x >> 6][2] = f;
    }
    __syncthreads();
    if (threadIdx.x < 3) out[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

// One wavefront that stays busy for `ticks` of the 100 MHz clock (bounded): vr_stream()'s probe for streams that really
// run side by side (HIP maps streams onto a few hardware queues; two streams on one queue serialise).
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks, unsigned* __restrict__ sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < 4000000u) {
        __builtin_amdgcn_s_sleep(8);
        ++n;
    }
    if (sink && threadIdx.x == 0 && n == 0xFFFFFFFFu) *sink = n;  // (never true: keeps the loop observable)
}

// One workgroup: order[] = the logical blocks of the launch whose records are `in`, sorted by the longest per-ray sample
// chain of the block (record word 5, bits 40..), longest first, SEPARATELY within each residue class of the block index
// modulo 8: position b of the order holds a block lb with lb % 8 == b % 8.  Workgroups are dispatched round-robin over the
// 8 XCDs, so a block stays on the XCD its index maps to (what map_pixel's xcd_mode relies on) and every XCD runs its own
// blocks longest-processing-time-first.  Eight counting sorts over 128 key buckets each (n_blocks % 8 == 0).
constexpr int kOrderMaxBlocks = 192 * 1024;  // launches with more blocks keep the index order (C5, one wavefront per block: 130 560)
// It also reports how long the launch took: last workgroup end - first workgroup start of the records it reads (100 MHz
// ticks, to host-visible memory; vr_kernel_times).  Timing events around every launch cost the frame's stream 11 us.
// pw_heads: the queue heads of the persistent-wavefront launch whose records these are (vr_pw.h) -- cleared here, behind that
// launch, for the next launch that uses the slot.
__global__ __launch_bounds__(1024) void order_blocks_kernel(const unsigned long long* __restrict__ in, int n_blocks,
                                                            unsigned* __restrict__ order, unsigned* __restrict__ longest_chain,
                                                            unsigned long long* __restrict__ span_ticks,
                                                            unsigned* __restrict__ pw_heads,
                                                            unsigned long long* __restrict__ end_tick = nullptr)
{
    if (pw_heads != nullptr && threadIdx.x < 8) pw_heads[threadIdx.x * 64] = 0u;
    __shared__ unsigned hist[1024];  // [class 0..7][bucket 0..127]
    __shared__ unsigned base[1024];
    __shared__ unsigned chain_max;
    __shared__ unsigned long long t_first, t_last;
    const int t = threadIdx.x;
    hist[t] = 0;
    if (t == 0) {
        chain_max = 0;
        t_first = ~0ull;
        t_last = 0ull;
    }
    unsigned my_chain = 0;
    unsigned long long my_first = ~0ull, my_last = 0ull;
    __syncthreads();
    // every record is read ONCE (a launch that recycles the record buffer under this kernel can then only change the
    // order, never make it something other than a permutation); bucket 0 = the longest chains (32 steps per bucket)
    unsigned short bk[kOrderMaxBlocks / 1024];  // (private memory: the loops are not unrolled)
    for (int k = 0; k * 1024 < n_blocks; ++k) {
        const int b = t + k * 1024;
        if (b < n_blocks) {
            const unsigned long long w5 = in[(size_t)b * kBlockRecord + 5];
            unsigned long long crit = w5 >> 40;
            unsigned long long b0 = in[(size_t)b * kBlockRecord + 3], b1 = in[(size_t)b * kBlockRecord + 4];
            if (w5 & kRecSplit) {  // marched by two wavefronts (vr_mixed.h): the second half's record
                const size_t h = (size_t)(b + n_blocks) * kBlockRecord;
                const unsigned long long c2 = in[h + 5] >> 40, s2 = in[h + 3], e2 = in[h + 4];
                crit = c2 > crit ? c2 : crit;
                b0 = s2 < b0 ? s2 : b0;
                b1 = e2 > b1 ? e2 : b1;
            }
            my_first = b0 < my_first ? b0 : my_first;
            my_last = b1 > my_last ? b1 : my_last;
            my_chain = crit > my_chain ? (unsigned)crit : my_chain;
            const unsigned q = (unsigned)(crit >> 5);
            bk[k] = (unsigned short)(((unsigned)b & 7u) * 128u + (127u - (q > 127u ? 127u : q)));
            atomicAdd(&hist[bk[k]], 1u);
        }
    }
    __syncthreads();
    {   // exclusive prefix sum inside each class: classes are 128 consecutive entries = two wavefronts
        const unsigned mine = hist[t];
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, off, 64);
            if ((t & 63) >= off) incl += up;
        }
        __shared__ unsigned wave_total[16];
        if ((t & 63) == 63) wave_total[t >> 6] = incl;
        __syncthreads();
        const unsigned before = ((t >> 6) & 1) ? wave_total[(t >> 6) - 1] : 0u;  // the class's first wavefront
        base[t] = before + incl - mine;
    }
    __syncthreads();
    // the launch's longest ray chain + 1 (0 = not written yet), to host-visible memory: the host's choice of lanes per ray for
    // a later launch of the same shape reads it without synchronising (vr_api.hip, enqueue_render)
    if (longest_chain) {
        atomicMax(&chain_max, my_chain);
        __syncthreads();
        if (t == 0) *longest_chain = chain_max + 1u;
    }
    if (span_ticks) {
        // (a launch of several frames: the first frame's records stand for all of them -- the frames' workgroups are
        // interleaved, so its first start and last end are the launch's to within a few workgroup times)
        atomicMin(&t_first, my_first);
        atomicMax(&t_last, my_last);
        __syncthreads();
        if (t == 0) {
            if (end_tick) *end_tick = t_last | 1ull;  // (the last workgroup's end on the device clock; before the span: the host reads the span first)
            __threadfence_system();
            *span_ticks = t_last >= t_first ? t_last - t_first + 1ull : 1ull;  // (0 = not written yet)
        }
    }
    // scatter with one cursor per (class, bucket): the i-th block of class x goes to position 8 * i + x
    for (int k = 0; k * 1024 < n_blocks; ++k) {
        const int b = t + k * 1024;
        if (b < n_blocks) order[8u * atomicAdd(&base[bk[k]], 1u) + ((unsigned)b & 7u)] = (unsigned)b;
    }
}

// ---- brick distance field (rebuilt when the volume or the opacity table changes) --------------------------------
// pass 0: 0 for active bricks, 255 ("not reached yet") for inert ones
__global__ void brick_active_kernel(const float2* __restrict__ rec, unsigned char* __restrict__ dist, int n, int use_rgb,
                                    int zero_prefix, int res_o)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 r = rec[i];
    bool inert;
    if (use_rgb && !(r.y <= 0.0f)) inert = false;
    else if (r.x <= 0.0f) inert = zero_prefix >= 0;
    else inert = floorf(r.x * (float)res_o - 0.5f) + 2.0f <= (float)zero_prefix;
    dist[i] = inert ? 255 : 0;
}
// pass k = 1 .. kDistMax-1: an unreached brick with a neighbour (26-neighbourhood) at distance k-1 is at distance k.
// Neighbours written in the same pass carry k, never k-1, so the in-place update is race-free in effect.
__global__ void brick_dist_pass_kernel(unsigned char* __restrict__ dist, int bnx, int bny, int bnz, int k)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bnx * bny * bnz || dist[i] != 255) return;
    const int x = i % bnx, y = (i / bnx) % bny, z = i / (bnx * bny);
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int X = x + dx, Y = y + dy, Z = z + dz;
                if (X < 0 || Y < 0 || Z < 0 || X >= bnx || Y >= bny || Z >= bnz) continue;
                if (dist[(Z * bny + Y) * bnx + X] == (unsigned char)(k - 1)) {
                    dist[i] = (unsigned char)k;
                    return;
                }
            }
}
// final pass: bricks never reached are at least kDistMax away
__global__ void brick_dist_cap_kernel(unsigned char* __restrict__ dist, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && dist[i] == 255) dist[i] = (unsigned char)kDistMax;
}

// how many bricks are active (distance 0): what share of the volume a frame can be expected to sample (the default kernel
// choice of vr_api.hip: a volume with next to no inert bricks is marched by the persistent kernel)
__global__ void count_active_bricks_kernel(const unsigned char* __restrict__ dist, int n, unsigned* __restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long m = vr_ballot(i < n && dist[i] == 0);
    if ((threadIdx.x & 63) == 0 && m != 0) atomicAdd(out, (unsigned)__popcll(m));
}

// the box of the active bricks (brick coordinates, inclusive; out[0..2] start at INT_MAX, out[3..5] at -1): a ray whose positions stay
// outside it -- one brick of margin -- samples nothing, whatever the distance field says about the bricks on its way (march_p2_kernel's
// approach loop)
__global__ void active_brick_box_kernel(const unsigned char* __restrict__ dist, int bnx, int bny, int bnz, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    if (i < bnx * bny * bnz && dist[i] == 0) {
        lo[0] = hi[0] = i % bnx;
        lo[1] = hi[1] = (i / bnx) % bny;
        lo[2] = hi[2] = i / (bnx * bny);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], __shfl_down(lo[a], off, 64));
            hi[a] = max(hi[a], __shfl_down(hi[a], off, 64));
        }
    if ((threadIdx.x & 63) == 0 && hi[0] >= 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(out + a, lo[a]);
            atomicMax(out + 3 + a, hi[a]);
        }
    }
}

// VOLUME_MASK looks at two volumes on one grid: record = (max density of the CT, max(r,g,b) of the mask)
__global__ void merge_bricks_kernel(const float2* __restrict__ density_vol, const float2* __restrict__ mask_vol,
                                    float2* __restrict__ out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = make_float2(density_vol[i].x, mask_vol[i].y);
}

// ---- bricked copy of a volume (DevVolume::bricked; rebuilt with the brick records after every upload or in-place change) -----
// One thread per slot of the bricked arrays: slot s = brick * 64 + (lz * 16 + ly * 4 + lx) for the default 4 x 4 x 4 bricks;
// voxels beyond the volume's faces (the last brick of an axis whose size is not a multiple of the brick edge) are zero and
// never addressed.
__global__ void rebrick_kernel(const float4* __restrict__ lin, float4* __restrict__ bvol, float* __restrict__ bdens, int nx, int ny,
                               int nz, unsigned nbx, unsigned nby, size_t n_slots)
{
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; s < n_slots; s += stride) {
        const unsigned l = (unsigned)(s & (kVbN - 1u));
        const size_t b = s >> (3u * kVbS);
        const unsigned bx = (unsigned)(b % nbx), by = (unsigned)((b / nbx) % nby), bz = (unsigned)(b / ((size_t)nbx * nby));
        const int x = (int)((bx << kVbS) + (l & kVbM)), y = (int)((by << kVbS) + ((l >> kVbS) & kVbM)), z = (int)((bz << kVbS) + (l >> (2u * kVbS)));
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (x < nx && y < ny && z < nz) v = lin[((size_t)z * ny + y) * nx + x];
        bvol[s] = v;
        bdens[s] = v.w;
    }
}

// ---- density plane / derived-gradient check (run with the brick records after every upload or in-place change) -------
__global__ void extract_density_kernel(const float4* __restrict__ vol, float* __restrict__ dens, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dens[i] = vol[i].w;
}
// *mismatch != 0 afterwards unless EVERY voxel's .rgb has exactly the bits PreComputeGradient(false) computes from the
// .a plane ((-(p - m)) * 0.5 per axis, neighbours outside the grid = 0; NaNs never match).
__global__ void verify_gradient_kernel(const float4* __restrict__ vol, const float* __restrict__ dens, int nx, int ny, int nz,
                                       unsigned* __restrict__ mismatch)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    bool bad = false;
    if (x < nx) {
        const size_t c = ((size_t)z * ny + y) * nx + x;
        const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
        const float mx = x > 0 ? dens[c - 1] : 0.0f, px = x + 1 < nx ? dens[c + 1] : 0.0f;
        const float my = y > 0 ? dens[c - sy] : 0.0f, py = y + 1 < ny ? dens[c + sy] : 0.0f;
        const float mz = z > 0 ? dens[c - sz] : 0.0f, pz = z + 1 < nz ? dens[c + sz] : 0.0f;
        const float4 v = vol[c];
        bad = __float_as_uint(v.x) != __float_as_uint(grad_cd(px, mx)) || __float_as_uint(v.y) != __float_as_uint(grad_cd(py, my)) ||
              __float_as_uint(v.z) != __float_as_uint(grad_cd(pz, mz)) || v.x != v.x || v.y != v.y || v.z != v.z;
    }
    if (vr_ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(mismatch, 1u);
}

// ------------------------------------------------------------------------------------------------ data preparation
//  the reference's single-threaded CPU passes over the voxels, as HBM-bound streaming kernels.

template <typename T>
__global__ void broadcast_raw_kernel(const T* __restrict__ raw, float4* __restrict__ vol, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = (float)raw[i];
        vol[i] = make_float4(v, v, v, v);
    }
}

// out[0] = max over voxels of component `comp` (as f32 bits; values are >= 0 in every use: raw data, magnitudes)
__global__ void max_component_kernel(const float4* __restrict__ vol, size_t n, int comp, unsigned* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float m = 0.0f;
    for (; i < n; i += stride) {
        float4 v = vol[i];
        float c = comp == 0 ? v.x : (comp == 1 ? v.y : (comp == 2 ? v.z : v.w));
        if (m < c) m = c;  // std::max_element semantics of GetMaxNumber: NaNs never win
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));  // non-negative floats order like their bits
}

__global__ void normalize_kernel(float4* __restrict__ vol, size_t n, int value)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float d = (float)value;
    for (; i < n; i += stride) {
        float4 v = vol[i];
        v.w = v.w / d;
        vol[i] = v;
    }
}

// One thread per voxel, x fastest (coalesced); the density is read from the untouched .w lanes, so the pass can
// run in place.  max_mag (f32 bits) accumulates the largest |gradient| when requested.
__global__ void gradient_kernel(float4* __restrict__ vol, int nx, int ny, int nz, int want_max, unsigned* __restrict__ max_mag)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y, z = blockIdx.z;
    float mag = 0.0f;
    if (x < nx) {
        const size_t row = ((size_t)z * ny + y) * nx;
        const size_t c = row + x;
        const size_t sy = (size_t)nx, sz = (size_t)nx * ny;
        float mx = x > 0 ? vol[c - 1].w : 0.0f, px = x + 1 < nx ? vol[c + 1].w : 0.0f;
        float my = y > 0 ? vol[c - sy].w : 0.0f, py = y + 1 < ny ? vol[c + sy].w : 0.0f;
        float mz = z > 0 ? vol[c - sz].w : 0.0f, pz = z + 1 < nz ? vol[c + sz].w : 0.0f;
        float tx = (-(px - mx)) * 0.5f, ty = (-(py - my)) * 0.5f, tz = (-(pz - mz)) * 0.5f;
        if (want_max) mag = sqrtf((tx * tx + ty * ty) + tz * tz);
        float* o = reinterpret_cast<float*>(vol + c);
        o[0] = tx;
        o[1] = ty;
        o[2] = tz;
    }
    if (want_max) {
        if (!(mag > 0.0f)) mag = 0.0f;  // `if (mag > maxGradMag)`: NaN never wins
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mag = fmaxf(mag, __shfl_down(mag, off, 64));
        if ((threadIdx.x & 63) == 0 && mag > 0.0f) atomicMax(max_mag, __float_as_uint(mag));
    }
}

__global__ void scale_gradient_kernel(float4* __restrict__ vol, size_t n, const unsigned* __restrict__ max_mag)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const float d = __uint_as_float(*max_mag);
    for (; i < n; i += stride) {
        float4 v = vol[i];
        v.x = v.x / d;
        v.y = v.y / d;
        v.z = v.z / d;
        vol[i] = v;
    }
}

#endif  // !VR_FUSED

}  // namespace VR_KNS
