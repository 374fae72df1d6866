// vr_lt.h -- "LDS tiles": the lit march kernel with the voxels of the next few steps staged in LDS by LDS-DMA.
//
// BASELINE.json's north star asks for "the volume bricked into LDS tiles for coalesced HBM reads with wavefront-wide ray
// packets".  The bricks are the HBM layout (DevVolume::bricked); this is the LDS half.  A wavefront is an 8x8 pixel packet
// whose rays sit at one step index; the cells its sampling rays touch over the next kLtSteps steps form a compact box of
// voxels.  The wavefront computes that box (two wave reductions on packed 16-bit bounds), fetches it ONCE into its private
// 7.5 KiB LDS tile with global_load_lds_dwordx4 -- every lane names the address of one voxel of the box, the data goes from
// the memory pipeline straight into LDS: no VGPRs, no ds_write, no address arithmetic on the LDS side -- and the 8-corner
// gathers of those steps are ds_read_b128.  What that changes against the direct gather of march_kernel:
//   * the round trip to L2 / HBM is paid once per tile instead of once per step (a step's own memory wait is an LDS read);
//   * a voxel goes through the texture addressers once per tile instead of once per lane and corner that needs it
//     (a 7 x 7 x 2 patch is requested ~5 times over by the eight corner loads of one step alone);
//   * the corner loads leave the vmcnt queue, so nothing else in the loop waits behind them.
// A box that does not fit (rays fanning out, a packet straddling the silhouette) takes the direct gather for that step, and a
// step whose sampling cells are not all inside the current tile builds a new one.  Arithmetic, positions, blend order and
// counts are those of march_kernel's lit path (light_shade_blend, the per-step zero-opacity vote): bit-identical frames.
//
// Loop form: every lane of the wavefront stays in the loop until the whole packet is done (finished rays are predicated
// off), because the tile fill needs all 64 lanes; otherwise the skipping logic is march_kernel's (distance-field byte of
// the next position, wave-uniform runs through inert bricks).
// Reference loop: BasicVolLightApp.wgsl:207-234.  Flavour 15 (vr_set_kernel_flavour), lit shader, one frame per launch.
#pragma once
#include "vr_kernels.h"
#include "vr_wtb.h"

namespace VR_KNS {
using namespace vr;

constexpr int kLtCap = 480;   // float4 voxels per tile: 7.5 KiB of LDS per wavefront (one wavefront per workgroup, 20 per CU)
constexpr int kLtSteps = 4;   // steps a tile is planned for

// index of voxel (x, y, z) in the volume's gather array (bricked or the reference's x-fastest order)
__device__ __forceinline__ unsigned vox_index(const DevVolume& v, unsigned x, unsigned y, unsigned z)
{
    if (v.bricked)
        return (x >> kVbS) * kVbN + (x & kVbM) + (y >> kVbS) * v.brick_row + ((y & kVbM) << kVbS) + (z >> kVbS) * v.brick_slab +
               ((z & kVbM) << (2u * kVbS));
    return (z * (unsigned)v.ny + y) * (unsigned)v.nx + x;
}

// clamp-to-edge texel pairs of the cell of p (make_cell's arithmetic: the same rounding of p * n - 0.5)
struct LtCell {
    int i0, i1, j0, j1, k0, k1;
    float fx, fy, fz;
};
__device__ __forceinline__ LtCell lt_cell(const DevVolume& v, f3 p)
{
    const float x = mad(p.x, (float)v.nx, -0.5f), y = mad(p.y, (float)v.ny, -0.5f), z = mad(p.z, (float)v.nz, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    LtCell c;
    c.fx = x - x0;
    c.fy = y - y0;
    c.fz = z - z0;
    texel_pair(x0, v.nx, c.i0, c.i1);
    texel_pair(y0, v.ny, c.j0, c.j1);
    texel_pair(z0, v.nz, c.k0, c.k1);
    return c;
}

template <bool OFF32, bool SKIP>
__global__ __launch_bounds__(256) void march_lt_kernel(const MarchBatch B)
{
    const MarchParams& P = B.frame[0];
    __shared__ float4 tile[kLtCap];  // one wavefront per workgroup: the tile is the wavefront's own
    const unsigned long long t_start = wall_clock64();
    const int lb = logical_block(P);
    const PixelSlot slot = map_pixel_at(P, lb, 1, 0);
    const int lane = threadIdx.x & 63;
    const DevVolume& vol = P.vol[0];

    float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned blends = 0, covered = 0, fetched = 0;
    bool alive = false;
    f3 p = mk3(0.0f, 0.0f, 0.0f), w = p, step = p, wstep = p;
    int n_inside = 0;
    const float bx0 = P.bmin[0], by0 = P.bmin[1], bz0 = P.bmin[2];
    const float bx1 = P.bmax[0], by1 = P.bmax[1], bz1 = P.bmax[2];

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
                // the per-pixel prologue of march_kernel's lit shader, unchanged
                float step_size = P.step_size;
                wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.5f));
                wstep.z = wstep.z * (-1.0f);
                if (P.toggle_varstep == 1) step_size = ray_len / (float)P.steps_count;
                p = ray.start;
                if (P.toggle_jitter == 1) {
                    float jt = jitter((float)slot.px + 0.5f, (float)slot.py + 0.5f);
                    p = mk3(p.x + (dir.x * step_size) * jt, p.y + (dir.y * step_size) * jt, p.z + (dir.z * step_size) * jt);
                }
                step = mk3(dir.x * step_size, dir.y * step_size, dir.z * step_size);
                w = ray.world0;
                n_inside = steps_inside(p, step, bx0, by0, bz0, bx1, by1, bz1);
                alive = true;
            }
        }
    }

    unsigned D = 0;  // distance-field byte of p: 0 = sample, k >= 1 = identity, and so is everything within k-1 bricks
    if constexpr (SKIP) D = dist_at(P, brick_of<OFF32>(P, p));
    const int lim = min(n_inside, P.steps_count);
    float leap_c = 0.0f;
    if constexpr (SKIP) {
        const float vmax = fmaxf(fmaxf(fabsf(step.x) * P.bsx, fabsf(step.y) * P.bsy), fabsf(step.z) * P.bsz);
        leap_c = 0.999f / vmax;  // see march_kernel
    }
    // the tile: box [tx0, tx1] x [ty0, ty1] x [tz0, tz1] of voxels, row length tbx, slice size tbxy (wave-uniform)
    bool tile_ok = false;
    int tile_built = -kLtSteps;  // step index of the last attempt to build a tile
    int tx0 = 0, ty0 = 0, tz0 = 0, tx1 = -1, ty1 = -1, tz1 = -1, tbx = 0, tbxy = 0;
    const bool small_enough = vol.nx <= 65535 && vol.ny <= 65535 && vol.nz <= 65535;

    int i = 0;  // step index: the same for every lane
    while (i < P.steps_count && vr_ballot(alive) != 0) {
        const f3 pn = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
        unsigned Dn = 0;
        if constexpr (SKIP) Dn = dist_at(P, brick_of<OFF32>(P, pn));
        if constexpr (SKIP) {
            // wave-uniform run of identity steps (march_kernel): every ray that is still marching has at least 4 safe steps
            int m = 1 << 30;
            if (alive) m = (D >= 2) ? min((int)fminf(((float)D - (1.0f + kBrickHalf)) * leap_c, 64.0f), lim - i - 1) : 0;
            if (vr_ballot(m < 4) == 0) {
                int mw = 4;
                if (vr_ballot(m < 8) == 0) {
                    mw = 8;
                    if (vr_ballot(m < 16) == 0) {
                        mw = 16;
                        if (vr_ballot(m < 32) == 0) mw = vr_ballot(m < 64) == 0 ? 64 : 32;
                    }
                }
                for (int k = 0; k < mw; k += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                        w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                }
                i += mw;
                blends += alive ? (unsigned)mw : 0u;
                D = dist_at(P, brick_of<OFF32>(P, p));
                continue;
            }
        }
        bool inb = alive, gone = false;
        if (i >= n_inside) {
            inb = alive && p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
            gone = alive && !inb &&
                   ((step.x >= 0.0f && p.x > bx1) || (step.x <= 0.0f && p.x < bx0) || (step.y >= 0.0f && p.y > by1) ||
                    (step.y <= 0.0f && p.y < by0) || (step.z >= 0.0f && p.z > bz1) || (step.z <= 0.0f && p.z < bz0));
        }
        const bool real = inb && (!SKIP || D == 0);
        if (vr_ballot(real) != 0) {  // (wave-uniform: every lane of the wavefront is here)
            const LtCell c = lt_cell(vol, p);
            bool use_tile = false;
            if (small_enough) {
                const bool inside = !real || (tile_ok && c.i0 >= tx0 && c.i1 <= tx1 && c.j0 >= ty0 && c.j1 <= ty1 && c.k0 >= tz0 && c.k1 <= tz1);
                const bool all_inside = vr_ballot(!inside) == 0;
                // a tile is built at most once per kLtSteps steps: a ray that starts sampling outside the current tile (the
                // packet straddles the body's silhouette) sends THIS step through the direct gather instead of forcing a new
                // tile on everybody, and a box that did not fit is not tried again before kLtSteps steps have passed
                if (!all_inside && i - tile_built >= kLtSteps) {
                    tile_built = i;
                    // a new tile: the cells of the rays that sample now, at this and the next kLtSteps - 1 positions
                    unsigned lx = 0xFFFFu, ly = 0xFFFFu, lz = 0xFFFFu, hx = 0u, hy = 0u, hz = 0u;  // neutral for the other lanes
                    if (real) {
                        lx = (unsigned)c.i0; hx = (unsigned)c.i1; ly = (unsigned)c.j0; hy = (unsigned)c.j1; lz = (unsigned)c.k0; hz = (unsigned)c.k1;
                        f3 q = p;
#pragma unroll
                        for (int k = 1; k < kLtSteps; ++k) {
                            q = mk3(q.x + step.x, q.y + step.y, q.z + step.z);
                            const LtCell cq = lt_cell(vol, q);
                            lx = min(lx, (unsigned)cq.i0); hx = max(hx, (unsigned)cq.i1);
                            ly = min(ly, (unsigned)cq.j0); hy = max(hy, (unsigned)cq.j1);
                            lz = min(lz, (unsigned)cq.k0); hz = max(hz, (unsigned)cq.k1);
                        }
                    }
                    const unsigned r0 = wave_pk_min_u16(lx | (ly << 16));
                    const unsigned r1 = wave_pk_min_u16(lz | ((0xFFFFu - hx) << 16));
                    const unsigned r2 = wave_pk_min_u16((0xFFFFu - hy) | ((0xFFFFu - hz) << 16));
                    const int nx0 = (int)(r0 & 0xFFFFu), ny0 = (int)(r0 >> 16), nz0 = (int)(r1 & 0xFFFFu);
                    const int nx1 = (int)(0xFFFFu - (r1 >> 16)), ny1 = (int)(0xFFFFu - (r2 & 0xFFFFu)), nz1 = (int)(0xFFFFu - (r2 >> 16));
                    int bx = nx1 - nx0 + 1, by = ny1 - ny0 + 1, bz = nz1 - nz0 + 1;
                    int nvox = bx * by * bz;
                    tile_ok = false;
                    tx0 = nx0; ty0 = ny0; tz0 = nz0; tx1 = nx1; ty1 = ny1; tz1 = nz1;
                    if (nvox <= kLtCap) {
                        tile_ok = true;
                        tbx = bx;
                        tbxy = bx * by;
                        const float inv_bx = 1.0f / (float)bx, inv_by = 1.0f / (float)by;
                        wave_lds_fence();  // the previous tile's reads are done before it is overwritten
                        for (int base = 0; base < nvox; base += 64) {
                            const int idx = base + lane;
                            if (idx < nvox) {
                                // idx -> (tx, ty, tz): exact for these ranges ((idx + .5) / b is never within rounding of an integer)
                                const int row = (int)(((float)idx + 0.5f) * inv_bx);
                                const int txi = idx - row * bx;
                                const int tz = (int)(((float)row + 0.5f) * inv_by);
                                const int tyi = row - tz * by;
                                const unsigned g = vox_index(vol, (unsigned)(tx0 + txi), (unsigned)(ty0 + tyi), (unsigned)(tz0 + tz));
                                // LDS-DMA: this lane's 16 bytes land at tile[base + lane]
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vol.data + (size_t)g),
                                                                 (__attribute__((address_space(3))) void*)(tile + base), 16, 0, 0);
                            }
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        wave_lds_fence();
                    }
                }
                use_tile = tile_ok && (all_inside || tile_built == i);
            }
            if (real) {
                Fetch4 q;
                if (use_tile) {
                    const int o = ((c.k0 - tz0) * tbxy + (c.j0 - ty0) * tbx) + (c.i0 - tx0);
                    const int dx = c.i1 - c.i0, dy = (c.j1 - c.j0) * tbx, dz = (c.k1 - c.k0) * tbxy;
                    q.a = tile[o]; q.b = tile[o + dx];
                    q.d = tile[o + dy]; q.e = tile[o + dy + dx];
                    q.f = tile[o + dz]; q.g = tile[o + dz + dx];
                    q.h = tile[o + dz + dy]; q.i = tile[o + dz + dy + dx];
                } else {
                    const unsigned o000 = vox_index(vol, (unsigned)c.i0, (unsigned)c.j0, (unsigned)c.k0);
                    q.a = load_voxel<OFF32>(vol, o000);
                    q.b = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i1, (unsigned)c.j0, (unsigned)c.k0));
                    q.d = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i0, (unsigned)c.j1, (unsigned)c.k0));
                    q.e = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i1, (unsigned)c.j1, (unsigned)c.k0));
                    q.f = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i0, (unsigned)c.j0, (unsigned)c.k1));
                    q.g = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i1, (unsigned)c.j0, (unsigned)c.k1));
                    q.h = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i0, (unsigned)c.j1, (unsigned)c.k1));
                    q.i = load_voxel<OFF32>(vol, vox_index(vol, (unsigned)c.i1, (unsigned)c.j1, (unsigned)c.k1));
                }
                const v2f zw = interp_zw(q, c.fx, c.fy, c.fz);
                bool all_zero = false;
                if constexpr (SKIP) all_zero = vr_ballot(!opacity_is_zero(P, zw.y)) == 0;
                if (!all_zero) {
                    const TfFetch tq = tf_fetch(P.tf[0], zw.y);
                    const v2f gxy = interp_xy(q, c.fx, c.fy, c.fz);
                    light_shade_blend(P, w, zw, gxy, tq, dst);
                }
                ++fetched;
            }
        }
        if (inb) {
            ++blends;
            if (real && !can_blend<V_LIGHT>(dst.w)) alive = false;  // cut-off reached: no later step can blend
        } else if (gone) {
            alive = false;
        }
        if (alive) {
            p = pn;
            w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
        }
        D = Dn;
        ++i;
    }

    if (slot.active || (P.packed && slot.in_launch)) P.out[slot.out_index] = dst;
    store_wave_counts(P, lb, blends, covered, fetched, t_start, false);
}

}  // namespace VR_KNS
