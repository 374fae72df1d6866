/*
 * vr_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See vr_oracle.h.
 *
 * PARITY UNPINNED (no reference fixtures exist; see header).  Scalar C99, IEEE f32, compiled with
 * -O2 -ffp-contract=off -fno-fast-math so that every `a*b+c` below is a rounded multiply followed
 * by a rounded add.  Citations are paths below the reference root (/root/reference).
 *
 * Normative choices where WGSL leaves precision to the implementation (documented in DESIGN.md):
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   length(v)     = sqrtf(dot(v,v))                      (IEEE sqrt)
 *   normalize(v)  = v * (1.0f / length(v))               (IEEE divide; HLSL lowers normalize to x*rsqrt)
 *   max(x, 0.0)   = (x > 0) ? x : 0                      (NaN -> 0, as on the reference's D3D12 back end)
 *   lerp(a,b,t)   = a + (b - a) * t                      (linear filtering, WebGPU spec formula)
 *   sin(x)        = double-precision quadrant reduction + Taylor polynomial, rounded to f32
 */
#include "vr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ arithmetic modes
 * WGSL leaves it to the implementation whether `a * b + c` is evaluated with one rounding (fused multiply-add) or two;
 * the reference's back end (Tint -> HLSL -> D3D12 driver) emits `mad` for these expressions.  Mode 0 (default,
 * "separate") rounds the product and the sum separately everywhere.  Mode 1 ("fused") evaluates the PER-SAMPLE
 * expressions of that shape with fmaf -- texture coordinates p * N - 0.5, every linear-filter lerp a + (b - a) * t,
 * dot products (x*x' first, then fma, fma), light.diffuse * m * kD + light.ambient * kA, the CT / RT colour mix and
 * FrontToBackBlend's (1 - dst.a) * src + dst.  Everything that places the ray (matrix products, slab test, SetupRay,
 * step vectors, jitter, the repeated additions p += step) is the same in both modes, so both march the same positions. */
static int g_fused = 0;
void vro_set_arithmetic(int fused) { g_fused = fused ? 1 : 0; }
int vro_get_arithmetic(void) { return g_fused; }
static float mad(float a, float b, float c) { return g_fused ? fmaf(a, b, c) : a * b + c; }

/* ------------------------------------------------------------------ small vector helpers */

/* ray placement (both modes): separately rounded */
static float dot3s(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static float length3s(const float a[3]) { return sqrtf(dot3s(a, a)); }
static void normalize3s(const float a[3], float out[3])
{
    float inv = 1.0f / length3s(a);
    out[0] = a[0] * inv;
    out[1] = a[1] * inv;
    out[2] = a[2] * inv;
}
/* per sample (mode dependent) */
static float dot3(const float a[3], const float b[3]) { return mad(a[2], b[2], mad(a[1], b[1], a[0] * b[0])); }
static float length3(const float a[3]) { return sqrtf(dot3(a, a)); }
static void normalize3(const float a[3], float out[3])
{
    float inv = 1.0f / length3(a);
    out[0] = a[0] * inv;
    out[1] = a[1] * inv;
    out[2] = a[2] * inv;
}
static float max0(float x) { return (x > 0.0f) ? x : 0.0f; }
static float lerpf(float a, float b, float t) { return mad(b - a, t, a); }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* f32 -> i32 conversion is SATURATING with NaN -> 0 (WGSL clamps the value to the target range; C leaves the
 * out-of-range case undefined).  Only reachable with non-finite voxel data. */
static int f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int)x;
}
/* texel pair of a clamp-to-edge linear fetch: i0 = clamp(t, 0, n-1), i1 = clamp(t+1, 0, n-1), t = f2i(floor(x)) */
static void texel_pair(float x0, int n, int* i0, int* i1)
{
    int t = f2i(x0);
    if (t > n - 1) t = n - 1; /* limit from above first: t + 1 cannot overflow */
    *i0 = clampi(t, 0, n - 1);
    *i1 = clampi(t + 1, 0, n - 1);
}

/* column-major mat4 * vec4, summed left to right */
static void mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r)
        out[r] = ((m[0 + r] * v[0] + m[4 + r] * v[1]) + m[8 + r] * v[2]) + m[12 + r] * v[3];
}

/* ------------------------------------------------------------------ deterministic sin (jitter) */

static double sin_d(double x)
{
    double n = rint(x * 0.63661977236758138);                                 /* 2/pi */
    double r = (x - n * 1.5707963267948966) - n * 6.123233995736766e-17;       /* pi/2 hi, lo */
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
    switch (q) {
    case 0: return s;
    case 1: return c;
    case 2: return -s;
    default: return -c;
    }
}

/* BasicVolumeApp.wgsl:107-110  fract(sin(dot(co, vec2(12.9898,78.233))) * 43758.5453) */
float vro_jitter(float x, float y)
{
    float d = x * 12.9898f + y * 78.233f;
    float s = (float)sin_d((double)d);
    float v = s * 43758.5453f;
    return v - floorf(v);
}

/* ------------------------------------------------------------------ deterministic pow (illustrative shader) */

static double log2_d(double x) /* x > 0, finite, normal (every positive float is a normal double) */
{
    uint64_t b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &b, 8); /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0); /* log(m) = 2 atanh(s), |s| <= 0.1716 */
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
    return (double)e + ln * 1.4426950408889634; /* 1 / ln 2 */
}

static double exp2_d(double t) /* finite t in (-1100, 1024) */
{
    double n = rint(t);
    double z = (t - n) * 0.6931471805599453; /* ln 2; |z| <= 0.3466 */
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
    if (k < -1022) return 0.0; /* far below the smallest float */
    uint64_t b = (uint64_t)(k + 1023) << 52;
    double p2;
    memcpy(&p2, &b, 8);
    return r * p2;
}

float vro_pow(float x, float y)
{
    double lx;
    if (x != x || x < 0.0f) lx = NAN;
    else if (x == 0.0f) lx = -INFINITY;
    else if (x == INFINITY) lx = INFINITY;
    else lx = log2_d((double)x);
    double t = (double)y * lx; /* 0 * inf -> NaN, as exp2(y * log2(x)) gives */
    double r;
    if (t != t) r = NAN;
    else if (t >= 1024.0) r = INFINITY;
    else if (t <= -1100.0) r = 0.0;
    else r = exp2_d(t);
    return (float)r;
}

/* ------------------------------------------------------------------ samplers (Sampler.cpp:9-24) */

/* 3-D linear, clamp-to-edge, single mip (Texture.cpp:48); texel (i,j,k) at vec4[(k*ny + j)*nx + i]
 * (VolumeFile.cpp:306). */
static void tex3_linear(const vro_volume* v, const float p[3], float out[4])
{
    float x = mad(p[0], (float)v->nx, -0.5f);
    float y = mad(p[1], (float)v->ny, -0.5f);
    float z = mad(p[2], (float)v->nz, -0.5f);
    float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    float fx = x - x0, fy = y - y0, fz = z - z0;
    int i0, i1, j0, j1, k0, k1;
    texel_pair(x0, v->nx, &i0, &i1);
    texel_pair(y0, v->ny, &j0, &j1);
    texel_pair(z0, v->nz, &k0, &k1);
    const float* d = v->vec4;
    int64_t nx = v->nx, ny = v->ny;
    const float* v000 = d + 4 * ((k0 * ny + j0) * nx + i0);
    const float* v100 = d + 4 * ((k0 * ny + j0) * nx + i1);
    const float* v010 = d + 4 * ((k0 * ny + j1) * nx + i0);
    const float* v110 = d + 4 * ((k0 * ny + j1) * nx + i1);
    const float* v001 = d + 4 * ((k1 * ny + j0) * nx + i0);
    const float* v101 = d + 4 * ((k1 * ny + j0) * nx + i1);
    const float* v011 = d + 4 * ((k1 * ny + j1) * nx + i0);
    const float* v111 = d + 4 * ((k1 * ny + j1) * nx + i1);
    for (int c = 0; c < 4; ++c) {
        float c00 = lerpf(v000[c], v100[c], fx);
        float c10 = lerpf(v010[c], v110[c], fx);
        float c01 = lerpf(v001[c], v101[c], fx);
        float c11 = lerpf(v011[c], v111[c], fx);
        float c0 = lerpf(c00, c10, fy);
        float c1 = lerpf(c01, c11, fy);
        out[c] = lerpf(c0, c1, fz);
    }
}

/* 3-D nearest (TFCalibrationApp.wgsl:172, samplerNN) */
static void tex3_nearest(const vro_volume* v, const float p[3], float out[4])
{
    int i = clampi(f2i(floorf(p[0] * (float)v->nx)), 0, v->nx - 1);
    int j = clampi(f2i(floorf(p[1] * (float)v->ny)), 0, v->ny - 1);
    int k = clampi(f2i(floorf(p[2] * (float)v->nz)), 0, v->nz - 1);
    const float* t = v->vec4 + 4 * (((int64_t)k * v->ny + j) * v->nx + i);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2]; out[3] = t[3];
}

/* 1-D linear lookups of the transfer-function textures at coordinate d */
static void tf_coords(int res, float d, int* i0, int* i1, float* f)
{
    float x = mad(d, (float)res, -0.5f);
    float x0 = floorf(x);
    *f = x - x0;
    texel_pair(x0, res, i0, i1);
}
static float tf_opacity(const vro_tf* tf, float d) /* textureSample(tfOpacity, samplerLin, d).r */
{
    int i0, i1; float f;
    tf_coords(tf->res, d, &i0, &i1, &f);
    return lerpf(tf->opacity[i0], tf->opacity[i1], f);
}
static void tf_color(const vro_tf* tf, float d, float rgb[3]) /* textureSample(tfColor, samplerLin, d).rgb */
{
    int i0, i1; float f;
    tf_coords(tf->res_color, d, &i0, &i1, &f);
    for (int c = 0; c < 3; ++c) rgb[c] = lerpf(tf->color_rgba[4 * i0 + c], tf->color_rgba[4 * i1 + c], f);
}

/* ------------------------------------------------------------------ ray set-up
 * Restates what the reference obtains from fixed-function hardware:
 *   vertex stage   BasicVolumeApp.wgsl:44-57 / rayCoords.wgsl:19-28 (position = P*V*M*v, uvw attribute)
 *   proxy box      Application.h:147-156   x,y in [-.5,.5], z in [-.25,.25]; uvw = (x+.5, y+.5, .5-2z)
 *   front faces    Application.cpp:589-590 (CCW, cull back)  -> entry point, `in.textureCoord`, `in.worldCoord`
 *   back faces     Application.cpp:602-604 (cull front) into texRayEnd -> exit point (rayCoords.wgsl:30-34)
 *   clip volume    WebGPU: 0 <= z_clip <= w_clip
 * For the pixel centre (px+.5, py+.5) the perspective-correct interpolation of uvw over a face equals the
 * intersection of the pixel's view ray with that face.  The view ray is the world-space segment between the
 * un-projected z_ndc = 0 and z_ndc = 1 points, so the near/far clip and "camera inside the box" cases need no
 * special handling: a fragment exists iff the entry point lies on that segment.  On the face that is hit the
 * corresponding uvw coordinate is a vertex constant and is therefore exact (0 or 1).
 */
static void unproject(const vro_uniforms* u, float nx, float ny, float nz, float world[3])
{
    float ndc[4] = { nx, ny, nz, 1.0f };
    float v[4], vw[4], w[4];
    mat4_mul_vec4(u->proj_inv, ndc, v);
    vw[0] = v[0] / v[3];
    vw[1] = v[1] / v[3];
    vw[2] = v[2] / v[3];
    vw[3] = 1.0f;
    mat4_mul_vec4(u->view_inv, vw, w);
    world[0] = w[0]; world[1] = w[1]; world[2] = w[2];
}

int vro_setup_ray(const vro_uniforms* u, int W, int H, int px, int py, float start[3], float end[3], float world0[3])
{
    static const float bmin[3] = { -0.5f, -0.5f, -0.25f };
    static const float bmax[3] = { 0.5f, 0.5f, 0.25f };
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float ndcx = (2.0f * fx) / (float)W - 1.0f;
    float ndcy = 1.0f - (2.0f * fy) / (float)H;
    float O[3], F[3], D[3], d[3];
    unproject(u, ndcx, ndcy, 0.0f, O);
    unproject(u, ndcx, ndcy, 1.0f, F);
    D[0] = F[0] - O[0]; D[1] = F[1] - O[1]; D[2] = F[2] - O[2];
    float seg = length3s(D);
    normalize3s(D, d);

    float t0 = -INFINITY, t1 = INFINITY;
    int a0 = -1, a1 = -1;
    for (int a = 0; a < 3; ++a) {
        if (d[a] != 0.0f) {
            float inv = 1.0f / d[a];
            float ta = (bmin[a] - O[a]) * inv;
            float tb = (bmax[a] - O[a]) * inv;
            float tn = ta < tb ? ta : tb;
            float tf = ta < tb ? tb : ta;
            if (tn > t0) { t0 = tn; a0 = a; }
            if (tf < t1) { t1 = tf; a1 = a; }
        } else if (O[a] < bmin[a] || O[a] > bmax[a]) {
            return 0;
        }
    }
    if (!(t0 < t1)) return 0;            /* the ray misses the box (or only grazes it) */
    if (!(t0 >= 0.0f && t0 <= seg)) return 0; /* entry in front of the near plane / camera inside: front face clipped */
    if (a0 < 0 || a1 < 0) return 0;

    float P0[3], P1[3];
    for (int a = 0; a < 3; ++a) {
        P0[a] = O[a] + d[a] * t0;
        P1[a] = O[a] + d[a] * t1;
    }
    P0[a0] = (d[a0] > 0.0f) ? bmin[a0] : bmax[a0];
    P1[a1] = (d[a1] > 0.0f) ? bmax[a1] : bmin[a1];

    start[0] = P0[0] + 0.5f; start[1] = P0[1] + 0.5f; start[2] = 0.5f - 2.0f * P0[2];
    end[0] = P1[0] + 0.5f;   end[1] = P1[1] + 0.5f;   end[2] = 0.5f - 2.0f * P1[2];
    world0[0] = P0[0]; world0[1] = P0[1]; world0[2] = P0[2]; /* model == identity, Application.cpp:489-492 */
    return 1;
}

/* ------------------------------------------------------------------ per-sample helpers */

/* IsInSampleCoords  BasicVolumeApp.wgsl:71-79 */
static int in_sample_coords(const vro_uniforms* u, const float p[3])
{
    float bminx = 0.0f + u->clip_x[0], bminy = 0.0f + u->clip_y[0], bminz = 0.0f + u->clip_z[0];
    float bmaxx = 1.0f - u->clip_x[1], bmaxy = 1.0f - u->clip_y[1], bmaxz = 1.0f - u->clip_z[1];
    return p[0] >= bminx && p[0] <= bmaxx && p[1] >= bminy && p[1] <= bmaxy && p[2] >= bminz && p[2] <= bmaxz;
}

/* FrontToBackBlend  BasicVolumeApp.wgsl:98-104 */
static void front_to_back_blend(const float rgb[3], float a, float dst[4])
{
    float sr = rgb[0] * a, sg = rgb[1] * a, sb = rgb[2] * a, sa = a;
    float om = 1.0f - dst[3];
    dst[0] = mad(om, sr, dst[0]);
    dst[1] = mad(om, sg, dst[1]);
    dst[2] = mad(om, sb, dst[2]);
    dst[3] = mad(om, sa, dst[3]);
}

/* light.diffuse * max(dot(N,L),0) * kD + light.ambient * kA, L = normalize(lightPos - w) */
static void shade(const float N[3], const float w[3], const float lpos[3], const float diffuse[3],
                  const float ambient[3], float kD, float kA, float out[3])
{
    float lv[3] = { lpos[0] - w[0], lpos[1] - w[1], lpos[2] - w[2] };
    float L[3];
    normalize3(lv, L);
    float m = max0(dot3(N, L));
    for (int c = 0; c < 3; ++c) out[c] = mad(diffuse[c] * m, kD, ambient[c] * kA);
}

/* ------------------------------------------------------------------ fs_main */

uint32_t vro_shade_pixel(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H,
                         int px, int py, float out[4], int* covered)
{
    float start[3], end[3], wc[3];
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (covered) *covered = 0;
    if (!vro_setup_ray(u, W, H, px, py, start, end, wc)) return 0;
    if (covered) *covered = 1;

    /* fs_main prologue, e.g. BasicVolLightApp.wgsl:155-181 */
    float texC[2] = { 0.5f * (wc[0] / 1.0f) + 0.5f, -0.5f * (wc[1] / 1.0f) + 0.5f }; /* worldCoord.w == 1 */
    /* SetupRay  BasicVolumeApp.wgsl:86-96 */
    float diff[3] = { end[0] - start[0], end[1] - start[1], end[2] - start[2] };
    float dir[3];
    normalize3s(diff, dir);
    float ray_len = length3s(diff);

    switch (u->fragment_mode) { /* BasicVolumeApp.wgsl:128-143 */
    case 1: out[0] = fabsf(dir[0]); out[1] = fabsf(dir[1]); out[2] = fabsf(dir[2]); out[3] = 1.0f; return 0;
    case 2: out[0] = start[0]; out[1] = start[1]; out[2] = start[2]; out[3] = 1.0f; return 0;
    case 3: out[0] = end[0]; out[1] = end[1]; out[2] = end[2]; out[3] = 1.0f; return 0;
    case 4: out[0] = texC[0]; out[1] = texC[1]; out[2] = 0.0f; out[3] = 1.0f; return 0;
    default: break;
    }

    float step_size = u->step_size;
    float world_step[3] = { 0.0f, 0.0f, 0.0f };
    if (variant == VRO_LIGHT || variant == VRO_LIGHT_INSHADER) {
        /* CalculateWorldStep BEFORE the variable-step override: BasicVolLightApp.wgsl:184-185, 78-84 */
        world_step[0] = dir[0] * (step_size * 1.0f);
        world_step[1] = dir[1] * (step_size * 1.0f);
        world_step[2] = dir[2] * (step_size * 0.5f);
        world_step[2] = world_step[2] * (-1.0f);
    }
    if (u->toggles[0] == 1) step_size = ray_len / (float)u->steps_count; /* GetStepSize, :62-65 */

    float p[3] = { start[0], start[1], start[2] };
    if (u->toggles[1] == 1) { /* :156-160, in.position.xy = pixel centre */
        float j = vro_jitter((float)px + 0.5f, (float)py + 0.5f);
        for (int c = 0; c < 3; ++c) p[c] = p[c] + (dir[c] * step_size) * j;
    }
    float step[3] = { dir[0] * step_size, dir[1] * step_size, dir[2] * step_size };
    if (variant == VRO_MULTI_CTRT || variant == VRO_ILLUSTRATIVE) {
        /* CalculateWorldStep AFTER the override: MultiCTRTApp.wgsl:213-214, 154-160 (MutliCTRTIllustrative.wgsl:262-263) */
        world_step[0] = dir[0] * (step_size * 1.0f);
        world_step[1] = dir[1] * (step_size * 1.0f);
        world_step[2] = dir[2] * (step_size * 0.7f);
        world_step[2] = world_step[2] * (-1.0f);
    }
    if (variant == VRO_VOLUME_MASK || variant == VRO_THREE_FILES) {
        /* "world" position advanced by the uvw step: VolumeMaskApp.wgsl:213, ThreeFilesApp.wgsl:268 */
        world_step[0] = step[0]; world_step[1] = step[1]; world_step[2] = step[2];
    }

    float dst[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    uint32_t blends = 0;
    const float lpos[3] = { u->light_pos[0], u->light_pos[1], u->light_pos[2] };
    const float ldif[3] = { u->light_diffuse[0], u->light_diffuse[1], u->light_diffuse[2] };
    const float lamb[3] = { u->light_ambient[0], u->light_ambient[1], u->light_ambient[2] };

    for (int i = 0; i < u->steps_count; ++i) {
        switch (variant) {
        case VRO_BASIC: { /* BasicVolumeApp.wgsl:167-185 */
            float v[4], rgb[3];
            tex3_linear(&vols[0], p, v);
            float density = v[3];
            float opacity = tf_opacity(&tfs[0], density);
            tf_color(&tfs[0], density, rgb);
            if (in_sample_coords(u, p) && dst[3] <= 0.95f) { front_to_back_blend(rgb, opacity, dst); ++blends; }
        } break;
        case VRO_LIGHT: { /* BasicVolLightApp.wgsl:207-234 */
            float v[4], rgb[3];
            tex3_linear(&vols[0], p, v);
            float density = v[3];
            float opacity = tf_opacity(&tfs[0], density);
            tf_color(&tfs[0], density, rgb);
            if (in_sample_coords(u, p) && dst[3] < 1.0f) {
                float N[3], s[3];
                normalize3(v, N);
                shade(N, wc, lpos, ldif, lamb, 2.5f, 0.5f, s); /* BlinnPhong :137-147 */
                rgb[0] *= s[0]; rgb[1] *= s[1]; rgb[2] *= s[2];
                front_to_back_blend(rgb, opacity, dst);
                ++blends;
            }
        } break;
        case VRO_LIGHT_INSHADER: { /* BasicVolLightApp.wgsl:207-234 with :212 enabled: gradient = ComputeGradient(...) */
            float v[4], rgb[3];
            tex3_linear(&vols[0], p, v);
            float density = v[3];
            /* ComputeGradient(currentPosition, stepSize, textMain)  :239-253; dirs[k] * step = (step,0,0) ... */
            float r[3], g[3];
            for (int a = 0; a < 3; ++a) {
                float dv[3] = { ((a == 0) ? 1.0f : 0.0f) * step_size, ((a == 1) ? 1.0f : 0.0f) * step_size,
                                ((a == 2) ? 1.0f : 0.0f) * step_size };
                float pp[3] = { p[0] + dv[0], p[1] + dv[1], p[2] + dv[2] };
                float pm[3] = { p[0] - dv[0], p[1] - dv[1], p[2] - dv[2] };
                float sp[4], sm[4];
                tex3_linear(&vols[0], pp, sp);
                tex3_linear(&vols[0], pm, sm);
                r[a] = sp[3] - sm[3];
            }
            float l = length3(r);
            if (l == 0.0f) { g[0] = g[1] = g[2] = 0.0f; }
            else { g[0] = (-r[0]) / l; g[1] = (-r[1]) / l; g[2] = (-r[2]) / l; }
            float opacity = tf_opacity(&tfs[0], density);
            tf_color(&tfs[0], density, rgb);
            if (in_sample_coords(u, p) && dst[3] < 1.0f) {
                float N[3], s[3];
                normalize3(g, N); /* normalize(vec3(0)) = NaN -> max(NaN, 0) = 0: ambient only */
                shade(N, wc, lpos, ldif, lamb, 2.5f, 0.5f, s);
                rgb[0] *= s[0]; rgb[1] *= s[1]; rgb[2] *= s[2];
                front_to_back_blend(rgb, opacity, dst);
                ++blends;
            }
        } break;
        case VRO_VOLUME_MASK: { /* VolumeMaskApp.wgsl:182-214; vols: 0 mask, 1 RT, 2 CT; tfs: 0 CT, 1 RT */
            float mask[4], rt[4], ct[4], crt[3], cct[3];
            tex3_linear(&vols[0], p, mask);
            tex3_linear(&vols[1], p, rt);
            tex3_linear(&vols[2], p, ct);
            float rt_s = rt[3], ct_s = ct[3];
            float o_rt = tf_opacity(&tfs[1], rt_s);
            tf_color(&tfs[1], rt_s, crt);
            float o_ct = tf_opacity(&tfs[0], ct_s);
            tf_color(&tfs[0], ct_s, cct);
            if (in_sample_coords(u, p) && dst[3] < 1.0f) {
                static const float lp[3] = { 0.0f, -5.0f, 0.0f };
                static const float dif[3] = { 0.96f, 0.76f, 0.67f };
                static const float amb[3] = { 1.0f, 1.0f, 1.0f };
                float N[3], s[3], col[3];
                normalize3(ct, N);
                shade(N, wc, lp, dif, amb, 1.5f, 0.5f, s); /* BlinnPhong :116-123 */
                col[0] = cct[0] * s[0]; col[1] = cct[1] * s[1]; col[2] = cct[2] * s[2];
                float opacity = o_ct;
                if (mask[0] > 0.0f || mask[1] > 0.0f || mask[2] > 0.0f) {
                    opacity = o_rt;
                    col[0] = crt[0]; col[1] = crt[1]; col[2] = crt[2];
                }
                front_to_back_blend(col, opacity, dst);
                ++blends;
            }
        } break;
        case VRO_THREE_FILES: { /* ThreeFilesApp.wgsl:224-269; vols: 0 CT, 1 RT; tfs: 0 CT, 1 RT */
            float ct[4], rt[4], cct[3], crt[3], col[3];
            tex3_linear(&vols[0], p, ct);
            tex3_linear(&vols[1], p, rt);
            float o_ct = tf_opacity(&tfs[0], ct[3]);
            tf_color(&tfs[0], ct[3], cct);
            float o_rt = tf_opacity(&tfs[1], rt[3]);
            tf_color(&tfs[1], rt[3], crt);
            if (in_sample_coords(u, p) && dst[3] < 1.0f) {
                for (int c = 0; c < 3; ++c) col[c] = mad(crt[c], o_rt, cct[c] * (1.0f - o_rt));
                front_to_back_blend(col, o_ct, dst);
                ++blends;
            }
        } break;
        case VRO_MULTI_CTRT: { /* MultiCTRTApp.wgsl:219-256 */
            float ct[4], rt[4], cct[3], crt[3], col[3];
            tex3_linear(&vols[0], p, ct);
            tex3_linear(&vols[1], p, rt);
            float o_ct = tf_opacity(&tfs[0], ct[3]);
            tf_color(&tfs[0], ct[3], cct);
            float o_rt = tf_opacity(&tfs[1], rt[3]);
            tf_color(&tfs[1], rt[3], crt);
            if (in_sample_coords(u, p) && dst[3] <= 0.95f) {
                static const float lp[3] = { 0.0f, -5.0f, 0.0f };
                float N[3], s[3];
                for (int c = 0; c < 3; ++c) col[c] = mad(crt[c], o_rt, cct[c] * (1.0f - o_rt));
                normalize3(ct, N);
                shade(N, wc, lp, ldif, lamb, 3.5f, 0.5f, s); /* BlinnPhong :129-139 */
                col[0] *= s[0]; col[1] *= s[1]; col[2] *= s[2];
                float opacity = o_ct * length3(ct); /* GradinetMagnitudeOpacityModulation :148-151 */
                front_to_back_blend(col, opacity, dst);
                ++blends;
            }
        } break;
        case VRO_ILLUSTRATIVE: { /* MutliCTRTIllustrative.wgsl:271-310; vols / tfs as MULTI_CTRT */
            float ct[4], rt[4], cct[3], crt[3], col[3];
            tex3_linear(&vols[0], p, ct);
            tex3_linear(&vols[1], p, rt);
            float o_ct = tf_opacity(&tfs[0], ct[3]);
            tf_color(&tfs[0], ct[3], cct);
            float o_rt = tf_opacity(&tfs[1], rt[3]);
            tf_color(&tfs[1], rt[3], crt);
            if (in_sample_coords(u, p) && dst[3] <= 0.95f) {
                static const float lp[3] = { 0.0f, -5.0f, 0.0f };
                float N[3], s3[3];
                for (int c = 0; c < 3; ++c) col[c] = mad(crt[c], o_rt, cct[c] * (1.0f - o_rt));
                normalize3(ct, N);
                shade(N, wc, lp, ldif, lamb, 3.5f, 0.5f, s3); /* BlinnPhong :132-143 */
                col[0] *= s3[0]; col[1] *= s3[1]; col[2] *= s3[2];
                /* IllustrativeContextPreservingOpacity :158-186 (the texture-space distance option, :182) */
                float Lv[3] = { lp[0] - wc[0], lp[1] - wc[1], lp[2] - wc[2] }, L[3];
                normalize3(Lv, L);
                float Vv[3] = { u->camera_pos[0] - wc[0], u->camera_pos[1] - wc[1], u->camera_pos[2] - wc[2] }, V[3];
                normalize3(Vv, V);
                float Hv[3] = { V[0] + L[0], V[1] + L[1], V[2] + L[2] }, Hn[3];
                normalize3(Hv, Hn);
                float Lg[3] = { L[0] * ct[0], L[1] * ct[1], L[2] * ct[2] };
                float Hg[3] = { Hn[0] * ct[0], Hn[1] * ct[1], Hn[2] * ct[2] };
                float s = (0.5f + 2.5f * length3(Lg)) + 1.0f * vro_pow(length3(Hg), 1.0f);
                float dv[3] = { p[0] - start[0], p[1] - start[1], p[2] - start[2] };
                float dist = length3(dv);
                if (dist > 1.0f) dist = 1.0f;
                float inner = vro_pow(((5.0f * s) * (1.0f - dist)) * (1.0f - dst[3]), 0.8f);
                float opacity = o_ct * vro_pow(length3(ct), inner);
                front_to_back_blend(col, opacity, dst);
                ++blends;
            }
        } break;
        case VRO_TF_CALIB: { /* TFCalibrationApp.wgsl:168-194; vols: 0 CT, 1 mask */
            float v[4], mask[4], rgb[3];
            tex3_linear(&vols[0], p, v);
            tex3_nearest(&vols[1], p, mask);
            float density = v[3];
            float opacity = tf_opacity(&tfs[0], density);
            tf_color(&tfs[0], density, rgb);
            if (mask[0] > 0.0f) { rgb[0] = 1.0f; rgb[1] = 1.0f; rgb[2] = 0.0f; opacity = 0.1f; }
            if (in_sample_coords(u, p) && dst[3] <= 0.95f) { front_to_back_blend(rgb, opacity, dst); ++blends; }
        } break;
        default: return 0;
        }
        p[0] = p[0] + step[0]; p[1] = p[1] + step[1]; p[2] = p[2] + step[2];
        wc[0] = wc[0] + world_step[0]; wc[1] = wc[1] + world_step[1]; wc[2] = wc[2] + world_step[2];
    }
    out[0] = dst[0]; out[1] = dst[1]; out[2] = dst[2]; out[3] = dst[3];
    return blends;
}

/* ------------------------------------------------------------------ frame drivers */

typedef struct job {
    int variant; const vro_uniforms* u; const vro_volume* vols; const vro_tf* tfs; int W, H, y0, y1;
    float* frag; const int32_t* pxy; int n; float* out;
    int next; uint64_t samples, covered; pthread_mutex_t mu;
} job;

static void* row_worker(void* arg)
{
    job* j = (job*)arg;
    uint64_t samples = 0, covered = 0;
    for (;;) {
        int y = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (y >= j->y1) break;
        for (int x = 0; x < j->W; ++x) {
            int cov;
            samples += vro_shade_pixel(j->variant, j->u, j->vols, j->tfs, j->W, j->H, x, y,
                                       j->frag + 4 * ((size_t)y * j->W + x), &cov);
            covered += (uint64_t)cov;
        }
    }
    pthread_mutex_lock(&j->mu);
    j->samples += samples;
    j->covered += covered;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static void* pixel_worker(void* arg)
{
    job* j = (job*)arg;
    uint64_t samples = 0;
    for (;;) {
        int i0 = __atomic_fetch_add(&j->next, 64, __ATOMIC_RELAXED);
        if (i0 >= j->n) break;
        int i1 = i0 + 64 < j->n ? i0 + 64 : j->n;
        for (int i = i0; i < i1; ++i)
            samples += vro_shade_pixel(j->variant, j->u, j->vols, j->tfs, j->W, j->H, j->pxy[2 * i], j->pxy[2 * i + 1],
                                       j->out + 4 * (size_t)i, NULL);
    }
    pthread_mutex_lock(&j->mu);
    j->samples += samples;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

static void run_job(job* j, void* (*fn)(void*), int nthreads)
{
    pthread_mutex_init(&j->mu, NULL);
    if (nthreads <= 1) {
        fn(j);
    } else {
        pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, fn, j);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j->mu);
}

int vro_render(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H, int y0,
               int y1, int nthreads, float* frag, uint64_t* samples, uint64_t* covered)
{
    if (!u || !vols || !tfs || !frag || W <= 0 || H <= 0 || y0 < 0 || y1 > H) return -1;
    job j;
    memset(&j, 0, sizeof j);
    j.variant = variant; j.u = u; j.vols = vols; j.tfs = tfs; j.W = W; j.H = H; j.y0 = y0; j.y1 = y1;
    j.frag = frag; j.next = y0;
    run_job(&j, row_worker, nthreads);
    if (samples) *samples = j.samples;
    if (covered) *covered = j.covered;
    return 0;
}

int vro_render_pixels(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H,
                      const int32_t* pxy, int n, int nthreads, float* out, uint64_t* samples)
{
    if (!u || !vols || !tfs || !pxy || !out || n < 0) return -1;
    job j;
    memset(&j, 0, sizeof j);
    j.variant = variant; j.u = u; j.vols = vols; j.tfs = tfs; j.W = W; j.H = H;
    j.pxy = pxy; j.n = n; j.out = out; j.next = 0;
    run_job(&j, pixel_worker, nthreads);
    if (samples) *samples = j.samples;
    return 0;
}

/* ------------------------------------------------------------------ output merge / present
 * Blend state SrcAlpha / OneMinusSrcAlpha (colour), Src / OneMinusSrc (alpha)  PipelineBuilder.cpp:142-147,
 * destination = white opaque background quad (fullscreen.wgsl:33-41), target BGRA8Unorm
 * (NativeGraphicsContext.cpp:118).  Pixels without a fragment keep the background. */
static uint8_t unorm8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)floorf(v * 255.0f + 0.5f);
}

void vro_present(const float* frag, int n_pixels, uint8_t* bgra8)
{
    for (int i = 0; i < n_pixels; ++i) {
        const float* s = frag + 4 * (size_t)i;
        float a = s[3];
        float r = s[0] * a + 1.0f * (1.0f - a);
        float g = s[1] * a + 1.0f * (1.0f - a);
        float b = s[2] * a + 1.0f * (1.0f - a);
        float oa = s[3] * a + 1.0f * (1.0f - a);
        bgra8[4 * i + 0] = unorm8(b);
        bgra8[4 * i + 1] = unorm8(g);
        bgra8[4 * i + 2] = unorm8(r);
        bgra8[4 * i + 3] = unorm8(oa);
    }
}
