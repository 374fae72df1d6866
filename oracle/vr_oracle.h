/*
 * vr_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C, scalar, one-pixel-at-a-time restatement of the reference's hot path
 * (the WGSL fs_main loops + the rasteriser set-up they depend on).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (libvr_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer vectors, and
 * it cannot be built or run here (Dawn/Tint/glm/dcm submodules are empty, the native layer is
 * Win32 + D3D12 only).  The oracle is therefore pinned only by the
 * analytic known-answer tests in tests/test_oracle_kat.py, not by reference outputs.
 */
#ifndef VR_ORACLE_H_
#define VR_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same byte layout as vr_uniforms in include/vr.h (declared independently on purpose). */
typedef struct vro_uniforms {
    float model[16], view[16], proj[16], view_inv[16], proj_inv[16];
    float camera_pos[3];
    int32_t fragment_mode, steps_count;
    float step_size;
    float clip_x[2], clip_y[2], clip_z[2];
    int32_t toggles[4];
    float light_pos[4], light_ambient[4], light_diffuse[4];
} vro_uniforms;

typedef struct vro_volume { const float* vec4; int32_t nx, ny, nz; } vro_volume; /* x fastest, 4 floats/voxel */
typedef struct vro_tf { const float* opacity; const float* color_rgba; int32_t res; int32_t res_color; } vro_tf; /* res = opacity table */

enum { VRO_BASIC = 0, VRO_LIGHT = 1, VRO_VOLUME_MASK = 2, VRO_THREE_FILES = 3, VRO_MULTI_CTRT = 4, VRO_TF_CALIB = 5,
       VRO_ILLUSTRATIVE = 6, /* MutliCTRTIllustrative.wgsl (compiled but never attached by the reference) */
       VRO_LIGHT_INSHADER = 7 /* BasicVolLightApp.wgsl with line 212 enabled: gradient = ComputeGradient(...) :239-253 */ };

/* Ray set-up for one pixel (restates rayCoords.wgsl + the vertex stage + rasteriser).
 * Returns 1 when the pixel has a fragment; start/end are uvw, world0 the world-space entry. */
int vro_setup_ray(const vro_uniforms* u, int W, int H, int px, int py, float start[3], float end[3], float world0[3]);

/* Shades ONE pixel. out[4] = fs_main return value (0 when no fragment). Returns blends executed. */
uint32_t vro_shade_pixel(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H,
                         int px, int py, float out[4], int* covered);

/* Renders rows [y0,y1) of the W x H frame into frag (full W*H*4 buffer, rows outside untouched).
 * nthreads <= 1: scalar single thread.  Returns 0, fills *samples (blends) and *covered. */
int vro_render(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H, int y0,
               int y1, int nthreads, float* frag, uint64_t* samples, uint64_t* covered);

/* Renders only the listed pixels (bounded CPU-baseline sample). out = n*4 floats. */
int vro_render_pixels(int variant, const vro_uniforms* u, const vro_volume* vols, const vro_tf* tfs, int W, int H,
                      const int32_t* pxy, int n, int nthreads, float* out, uint64_t* samples);

/* Output merge + present (PipelineBuilder.cpp:142-147 over fullscreen.wgsl white, BGRA8Unorm). */
void vro_present(const float* frag, int n_pixels, uint8_t* bgra8);

/* Data preparation restated (VolumeFile.cpp). vec4 = n voxels * 4 floats, in place. */
void vro_normalize_data(float* vec4, int64_t n, int normalization_value);
void vro_precompute_gradient(float* vec4, int nx, int ny, int nz, int norm_to_zero_one);

/* LinearInterpolation::Generate restated: writes (x1 - x0 + 1) values. */
void vro_lerp_float(int x0, int x1, float fx0, float fx1, float* out);
void vro_lerp_vec4(int x0, int x1, const float fx0[4], const float fx1[4], float* out);

/* jitter() helper exposed for tests. */
/* Arithmetic mode (see vr_oracle.c): 0 = every a*b+c rounds product and sum separately (default), 1 = the per-sample
 * expressions of that shape are fused multiply-adds.  Process-wide; set it before rendering. */
void vro_set_arithmetic(int fused);
int vro_get_arithmetic(void);

float vro_jitter(float x, float y);
/* WGSL pow(x, y) = exp2(y * log2(x)) (what the reference's HLSL back end emits), evaluated through f64 with a fixed
 * operation sequence so that the kernel can reproduce it bit for bit; x < 0 -> NaN, pow(0, 0) -> NaN. */
float vro_pow(float x, float y);

#ifdef __cplusplus
}
#endif
#endif
