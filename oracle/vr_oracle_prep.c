/*
 * vr_oracle_prep.c -- CPU ORACLE (TEST INFRASTRUCTURE): restatement of the reference's CPU data
 * preparation that feeds the hot path (App/src/file/VolumeFile.cpp, App/src/tf/LinearInterpolation.h).
 * PARITY UNPINNED (no reference fixtures).  Plain C99, -ffp-contract=off.
 */
#include "vr_oracle.h"

#include <math.h>
#include <stdlib.h>

/* VolumeFile::NormalizeData  VolumeFile.cpp:165-184.  normalization_value == 0 -> GetMaxNumber(), which is the
 * maximum of component [0] truncated to an integer (VolumeFile.cpp:53-60, called from the ctor :9). */
void vro_normalize_data(float* vec4, int64_t n, int normalization_value)
{
    if (normalization_value == 0) {
        float mx = 0.0f;
        int have = 0;
        for (int64_t i = 0; i < n; ++i) {
            float v = vec4[4 * i + 0];
            if (!have || mx < v) { mx = v; have = 1; }
        }
        normalization_value = (int)(size_t)mx;
    }
    for (int64_t i = 0; i < n; ++i) vec4[4 * i + 3] /= (float)normalization_value;
}

static float voxel_a(const float* vec4, int nx, int ny, int nz, int x, int y, int z)
{
    /* GetIndexFrom3D + GetVoxelData: out of range -> vec4(0)  VolumeFile.cpp:287-317 */
    if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) return 0.0f;
    return vec4[4 * (((int64_t)z * ny + y) * nx + x) + 3];
}

/* VolumeFile::PreComputeGradient  VolumeFile.cpp:196-257 */
void vro_precompute_gradient(float* vec4, int nx, int ny, int nz, int norm_to_zero_one)
{
    float max_mag = 0.0f;
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                int64_t c = ((int64_t)z * ny + y) * nx + x;
                float mx = voxel_a(vec4, nx, ny, nz, x - 1, y, z), px = voxel_a(vec4, nx, ny, nz, x + 1, y, z);
                float my = voxel_a(vec4, nx, ny, nz, x, y - 1, z), py = voxel_a(vec4, nx, ny, nz, x, y + 1, z);
                float mz = voxel_a(vec4, nx, ny, nz, x, y, z - 1), pz = voxel_a(vec4, nx, ny, nz, x, y, z + 1);
                float tx = (-(px - mx)) * 0.5f, ty = (-(py - my)) * 0.5f, tz = (-(pz - mz)) * 0.5f;
                if (norm_to_zero_one) {
                    float mag = sqrtf((tx * tx + ty * ty) + tz * tz);
                    if (mag > max_mag) max_mag = mag;
                }
                vec4[4 * c + 0] = tx;
                vec4[4 * c + 1] = ty;
                vec4[4 * c + 2] = tz;
            }
    if (norm_to_zero_one) {
        int64_t n = (int64_t)nx * ny * nz;
        for (int64_t i = 0; i < n; ++i) {
            vec4[4 * i + 0] /= max_mag;
            vec4[4 * i + 1] /= max_mag;
            vec4[4 * i + 2] /= max_mag;
        }
    }
}

/* LinearInterpolation::Generate<float,int>  LinearInterpolation.h:10-33 (step == 1) */
void vro_lerp_float(int x0, int x1, float fx0, float fx1, float* out)
{
    float slope = (fx1 - fx0) * (1.0f / (float)(x1 - x0));
    for (int i = x0; i < x1 + 1; ++i) out[i - x0] = fx0 + slope * (float)(i - x0);
}

/* LinearInterpolation::Generate<glm::vec4,int>: rgb interpolated, alpha forced to 1 (:20-26) */
void vro_lerp_vec4(int x0, int x1, const float fx0[4], const float fx1[4], float* out)
{
    float inv = 1.0f / (float)(x1 - x0);
    float sr = (fx1[0] - fx0[0]) * inv, sg = (fx1[1] - fx0[1]) * inv, sb = (fx1[2] - fx0[2]) * inv;
    for (int i = x0; i < x1 + 1; ++i) {
        float* o = out + 4 * (i - x0);
        o[0] = fx0[0] + sr * (float)(i - x0);
        o[1] = fx0[1] + sg * (float)(i - x0);
        o[2] = fx0[2] + sb * (float)(i - x0);
        o[3] = 1.0f;
    }
}
