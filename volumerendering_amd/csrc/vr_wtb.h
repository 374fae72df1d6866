// vr_wtb.h -- "wave tile" variant of the lit march kernel (BasicVolLightApp.wgsl:151-237): the voxels are staged
// through LDS (kernel flavours 2 and 3, A/B only: measured slower than the direct gather).
// One 64-lane wavefront = one 8x8 pixel packet with a private LDS region.  All rays of the packet sit at the same
// step index, so the cells they touch over the next TWO steps form a compact patch; the wavefront computes the
// bounding box of those cells (wave reduction), loads it into its LDS tile with coalesced row reads (every 64-byte
// line is requested once per tile instead of once per lane that needs it), and the 8-corner gathers of both steps
// then read LDS.  Arithmetic is the same as in vr_kernels.h (bit-identical output); a box that does not fit the
// tile, or a pair of steps in which no lane samples, takes the plain global-memory path for that pair.
#pragma once
#include "vr_kernels.h"

namespace VR_KNS {
using namespace vr;

constexpr int kWtbCap = 448;  // float4 voxels per tile: 7 KiB of LDS per wavefront (22 wavefronts per CU by LDS)

struct TileBox {
    int x0, y0, z0;  // lowest voxel index per axis
    int bx, bxy;     // row length, slice size (in voxels)
};

// Wave-wide component-wise minimum of two packed 16-bit values (v_pk_min_u16): the six bounds of the voxel box
// travel as three such pairs, the upper bounds complemented so that one kind of reduction serves both.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_min_u16(unsigned a, unsigned b)
{
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned wave_pk_min_u16(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = pk_min_u16(v, (unsigned)__shfl_xor((int)v, off, 64));
    return v;
}

struct CellIdx {
    int i0, i1, j0, j1, k0, k1;
    float fx, fy, fz;
};
__device__ __forceinline__ CellIdx cell_idx(const DevVolume& v, f3 p)
{
    float x = p.x * (float)v.nx - 0.5f;
    float y = p.y * (float)v.ny - 0.5f;
    float z = p.z * (float)v.nz - 0.5f;
    float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    CellIdx c;
    c.fx = x - x0;
    c.fy = y - y0;
    c.fz = z - z0;
    texel_pair(x0, v.nx, c.i0, c.i1);
    texel_pair(y0, v.ny, c.j0, c.j1);
    texel_pair(z0, v.nz, c.k0, c.k1);
    return c;
}

__device__ __forceinline__ float4 tri4(float4 a, float4 b, float4 d, float4 e, float4 f, float4 g, float4 h, float4 i,
                                       const CellIdx& c)
{
    float4 r;
    r.x = tri(a.x, b.x, d.x, e.x, f.x, g.x, h.x, i.x, c.fx, c.fy, c.fz);
    r.y = tri(a.y, b.y, d.y, e.y, f.y, g.y, h.y, i.y, c.fx, c.fy, c.fz);
    r.z = tri(a.z, b.z, d.z, e.z, f.z, g.z, h.z, i.z, c.fx, c.fy, c.fz);
    r.w = tri(a.w, b.w, d.w, e.w, f.w, g.w, h.w, i.w, c.fx, c.fy, c.fz);
    return r;
}

// textureSample(vol, samplerLin, p) with the 8 texels taken from the LDS tile
__device__ __forceinline__ float4 tex3_rgba_tile(const float4* tile, const TileBox& tb, const CellIdx& c)
{
    const int ox0 = c.i0 - tb.x0, ox1 = c.i1 - tb.x0;
    const int oy0 = (c.j0 - tb.y0) * tb.bx, oy1 = (c.j1 - tb.y0) * tb.bx;
    const int oz0 = (c.k0 - tb.z0) * tb.bxy, oz1 = (c.k1 - tb.z0) * tb.bxy;
    float4 a = tile[oz0 + oy0 + ox0], b = tile[oz0 + oy0 + ox1];
    float4 d = tile[oz0 + oy1 + ox0], e = tile[oz0 + oy1 + ox1];
    float4 f = tile[oz1 + oy0 + ox0], g = tile[oz1 + oy0 + ox1];
    float4 h = tile[oz1 + oy1 + ox0], i = tile[oz1 + oy1 + ox1];
    return tri4(a, b, d, e, f, g, h, i, c);
}

template <bool OFF32>
__device__ __forceinline__ float4 tex3_rgba_global(const DevVolume& v, const CellIdx& c)
{
    const unsigned r00 = ((unsigned)c.k0 * (unsigned)v.ny + (unsigned)c.j0) * (unsigned)v.nx;
    const unsigned r10 = ((unsigned)c.k0 * (unsigned)v.ny + (unsigned)c.j1) * (unsigned)v.nx;
    const unsigned r01 = ((unsigned)c.k1 * (unsigned)v.ny + (unsigned)c.j0) * (unsigned)v.nx;
    const unsigned r11 = ((unsigned)c.k1 * (unsigned)v.ny + (unsigned)c.j1) * (unsigned)v.nx;
    float4 a = load_vec4<OFF32>(v.data, r00 + c.i0), b = load_vec4<OFF32>(v.data, r00 + c.i1);
    float4 d = load_vec4<OFF32>(v.data, r10 + c.i0), e = load_vec4<OFF32>(v.data, r10 + c.i1);
    float4 f = load_vec4<OFF32>(v.data, r01 + c.i0), g = load_vec4<OFF32>(v.data, r01 + c.i1);
    float4 h = load_vec4<OFF32>(v.data, r11 + c.i0), i = load_vec4<OFF32>(v.data, r11 + c.i1);
    return tri4(a, b, d, e, f, g, h, i, c);
}

// everything of the lit loop body that follows the volume fetch (BasicVolLightApp.wgsl:214-228)
__device__ __forceinline__ void light_shade_blend(const MarchParams& P, float4 v, f3 w, float4& dst)
{
    TfSample t = tf_lookup(P.tf[0], v.w);
    f3 N = normalize3(mk3(v.x, v.y, v.z));
    f3 s = shade(N, w, mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]), mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                 mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 2.5f, 0.5f);
    blend(mk3(t.rgb.x * s.x, t.rgb.y * s.y, t.rgb.z * s.z), t.opacity, dst);
}

// Orders a wavefront's LDS writes before its following LDS reads of the SAME wave-private region (and the reads
// of one pair of steps before the next tile load overwrites them).  A wavefront's LDS instructions execute in
// issue order, so no s_barrier is needed; this only stops the compiler from moving memory operations across it.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <bool OFF32, bool SKIP>
__global__ __launch_bounds__(256) void march_wtb_light_kernel(const MarchBatch B)
{
    const MarchParams& P = frame_params<false>(B);
    // one tile per wavefront; the four wavefronts of the block never touch each other's region
    __shared__ float4 tiles[4 * kWtbCap];
    float4* const tile = tiles + (threadIdx.x >> 6) * kWtbCap;

    // work mapping: identical to march_kernel (16x16-pixel blocks, 8x8 packet per wavefront)
    const unsigned long long t_start = wall_clock64();
    const PixelSlot slot = map_pixel(P);
    const int lane = threadIdx.x & 63;
    const int px = slot.px, py = slot.py;
    const bool in_launch = slot.in_launch;
    const bool active = slot.active;
    const int out_index = slot.out_index;

    float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned blends = 0, covered = 0, fetched = 0;
    bool alive = false;
    f3 p = mk3(0.0f, 0.0f, 0.0f), w = p, step = p, wstep = p;
    if (active) {
        Ray ray = setup_ray(P, px, py);
        if (ray.hit) {
            covered = 1;
            alive = true;
            f3 diff = mk3(ray.end.x - ray.start.x, ray.end.y - ray.start.y, ray.end.z - ray.start.z);
            f3 dir = normalize3(diff);
            float ray_len = length3(diff);
            float step_size = P.step_size;
            wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.5f));
            wstep.z = wstep.z * (-1.0f);
            if (P.toggle_varstep == 1) step_size = ray_len / (float)P.steps_count;
            p = ray.start;
            if (P.toggle_jitter == 1) {
                float j = jitter((float)px + 0.5f, (float)py + 0.5f);
                p = mk3(p.x + (dir.x * step_size) * j, p.y + (dir.y * step_size) * j, p.z + (dir.z * step_size) * j);
            }
            step = mk3(dir.x * step_size, dir.y * step_size, dir.z * step_size);
            w = ray.world0;
        }
    }

    const float bx0 = P.bmin[0], by0 = P.bmin[1], bz0 = P.bmin[2];
    const float bx1 = P.bmax[0], by1 = P.bmax[1], bz1 = P.bmax[2];
    const DevVolume& vol = P.vol[0];
    int cur_brick = -1;
    bool cur_inert = false;
    int i = 0;  // step index: the same for every alive lane of the wavefront

    auto in_box = [&](f3 q) { return q.x >= bx0 && q.x <= bx1 && q.y >= by0 && q.y <= by1 && q.z >= bz0 && q.z <= bz1; };
    auto is_inert = [&](f3 q) {
        if constexpr (SKIP) {
            int bid = brick_of(P, q);
            if (bid != cur_brick) {
                cur_brick = bid;
                cur_inert = dist_at(P, bid) != 0;
            }
            return cur_inert;
        } else {
            return false;
        }
    };

    for (;;) {
        if (!__any(alive)) break;
        // ---- plan the next two steps (A at p, B at p + step; B optimistically assumes the lane survives A)
        const f3 pA = p;
        const f3 pB = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
        const bool inA = alive && i < P.steps_count && in_box(pA);
        const bool inB = alive && i + 1 < P.steps_count && in_box(pB);
        const bool sA = inA && !is_inert(pA);
        const bool sB = inB && !is_inert(pB);
        const bool any_sample = __any(sA || sB);
        bool use_tile = false;
        TileBox tb = TileBox{0, 0, 0, 0, 0};
        CellIdx cA, cB;
        if (any_sample) {  // wave-uniform: rounds in which the whole packet is in air / retired plan nothing
            cA = cell_idx(vol, pA);
            cB = cell_idx(vol, pB);
        }
        if (any_sample && vol.nx <= 65535 && vol.ny <= 65535 && vol.nz <= 65535) {
            unsigned lx = 0xFFFFu, ly = 0xFFFFu, lz = 0xFFFFu, hx = 0u, hy = 0u, hz = 0u;  // neutral for lanes that sample nothing
            if (sA) { lx = cA.i0; hx = cA.i1; ly = cA.j0; hy = cA.j1; lz = cA.k0; hz = cA.k1; }
            if (sB) {
                lx = min(lx, (unsigned)cB.i0); hx = max(hx, (unsigned)cB.i1);
                ly = min(ly, (unsigned)cB.j0); hy = max(hy, (unsigned)cB.j1);
                lz = min(lz, (unsigned)cB.k0); hz = max(hz, (unsigned)cB.k1);
            }
            const unsigned r0 = wave_pk_min_u16(lx | (ly << 16));
            const unsigned r1 = wave_pk_min_u16(lz | ((0xFFFFu - hx) << 16));
            const unsigned r2 = wave_pk_min_u16((0xFFFFu - hy) | ((0xFFFFu - hz) << 16));
            const int x0 = (int)(r0 & 0xFFFFu), y0 = (int)(r0 >> 16), z0 = (int)(r1 & 0xFFFFu);
            const int x1 = (int)(0xFFFFu - (r1 >> 16)), y1 = (int)(0xFFFFu - (r2 & 0xFFFFu)), z1 = (int)(0xFFFFu - (r2 >> 16));
            const int bx = x1 - x0 + 1, by = y1 - y0 + 1, bz = z1 - z0 + 1;
            const int nvox = bx * by * bz;
            if (nvox <= kWtbCap) {
                use_tile = true;
                tb = TileBox{x0, y0, z0, bx, bx * by};
                const float inv_bx = 1.0f / (float)bx, inv_by = 1.0f / (float)by;
                wave_lds_fence();  // the previous pair's reads are done before the tile is overwritten
                for (int base = 0; base < nvox; base += 64) {
                    const int idx = base + lane;
                    if (idx < nvox) {
                        // idx -> (tx, ty, tz): exact for these ranges ((idx + .5)/b is never within rounding of an integer)
                        const int row = (int)(((float)idx + 0.5f) * inv_bx);
                        const int txi = idx - row * bx;
                        const int tz = (int)(((float)row + 0.5f) * inv_by);
                        const int tyi = row - tz * by;
                        const unsigned g = ((unsigned)(z0 + tz) * (unsigned)vol.ny + (unsigned)(y0 + tyi)) * (unsigned)vol.nx +
                                           (unsigned)(x0 + txi);
                        tile[idx] = load_vec4<OFF32>(vol.data, g);
                    }
                }
                wave_lds_fence();
            }
        }
        // ---- step A, then step B
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool in_s = (s == 0) ? inA : inB;
            const bool sample_s = (s == 0) ? sA : sB;
            const CellIdx& c = (s == 0) ? cA : cB;
            if (alive) {
                if (i >= P.steps_count) {
                    alive = false;
                } else if (in_s) {
                    if (sample_s) {
                        float4 v = use_tile ? tex3_rgba_tile(tile, tb, c) : tex3_rgba_global<OFF32>(vol, c);
                        light_shade_blend(P, v, w, dst);
                        ++fetched;
                    }
                    ++blends;
                    if (!can_blend<V_LIGHT>(dst.w)) alive = false;
                } else {
                    const bool gone = (step.x >= 0.0f && p.x > bx1) || (step.x <= 0.0f && p.x < bx0) ||
                                      (step.y >= 0.0f && p.y > by1) || (step.y <= 0.0f && p.y < by0) ||
                                      (step.z >= 0.0f && p.z > bz1) || (step.z <= 0.0f && p.z < bz0);
                    if (gone) alive = false;
                }
                if (alive) {
                    p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                    w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                }
            }
            ++i;  // every alive lane advanced by one step; dead lanes never look at i again
        }
    }

    if (active || (P.packed && in_launch)) P.out[out_index] = dst;

    store_block_counts(P, blends, covered, fetched, t_start);
}

}  // namespace VR_KNS
