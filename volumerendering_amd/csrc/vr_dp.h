// Depth-parallel march kernel: FOUR lanes per ray.
//
// The expensive part of a step -- eight corner loads, interpolation, table look-ups, shading -- does not depend on
// the accumulated colour; only FrontToBackBlend and the opacity cut-off are sequential.  So the four lanes of a quad
// take four CONSECUTIVE steps of one ray, sample them at the same time, and the quad then blends the four results
// in step order (quad-broadcast DPP, every lane keeps an identical copy of dst).  Per sample nothing changes -- the
// same positions (each lane runs the ray's chain of rounded additions itself, four additions per round), the same
// arithmetic, the same blend order, the same sample counts -- but a ray's chain of dependent work is a quarter as long and
// a wavefront is 16 rays (a 4x4 pixel packet) instead of 64.  That is what the frame time hangs on: with one lane per
// ray the frame waits for the wavefronts with the longest rays (per-workgroup trace, tools/block_trace.py: longest
// workgroup 0.9 ms of a 0.93 ms frame while the average busy workgroup takes 0.42 ms), and a GPU that owns an eighth of
// the tiles is hardly faster than one that owns all of them.  Loop overhead (look-ahead, run test, box test) is also
// paid once per round of four steps instead of once per step.
//
// Reference loop: BasicVolumeApp.wgsl:167-185 and the five other shaders (see vr_kernels.h, march_kernel).
#pragma once
#include "vr_kernels.h"

namespace VR_KNS {
using namespace vr;

// The value depth slot J of the lane's own ray holds (K lanes per ray, K = 2 or 4, rays aligned to quads).
template <int K, int J>
__device__ __forceinline__ int slot_bcast(int v)
{
    // quad_perm:[J,J,J,J] for K = 4, [J,J,2+J,2+J] for K = 2
    constexpr int ctrl = (K == 4) ? (J | (J << 2) | (J << 4) | (J << 6)) : (J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6));
    return __builtin_amdgcn_mov_dpp(v, ctrl, 0xf, 0xf, true);
}
template <int K, int J>
__device__ __forceinline__ float slot_bcast(float v)
{
    return __int_as_float(slot_bcast<K, J>(__float_as_int(v)));
}

// K = 4: 8x8-pixel workgroups (64 per 64x64 tile), a wavefront = a 4x4 pixel packet; K = 2: 16x8-pixel workgroups
// (32 per tile), a wavefront = an 8x4 packet.  ray = lane / K, depth slot = lane % K.  Consecutive workgroups of a
// tile run on consecutive XCDs, so every XCD gets an even sample of the screen.
template <int K>
__device__ __forceinline__ PixelSlot map_pixel_dp(const MarchParams& P)
{
    PixelSlot s;
    // A tile is 256 (K = 4) or 128 (K = 2) wavefront packets; a workgroup is 1 or 4 of them (blockDim.x 64 / 256).  Packets
    // are numbered through the launch, four consecutive ones forming the 8x8 (K = 4) / 16x8 (K = 2) pixel group the
    // 256-thread workgroup used to be, so the pixel <-> lane mapping does not depend on the workgroup size.
    constexpr int kPpt = (K == 4) ? 256 : 128;  // packets per tile
    const int wpb = (int)(blockDim.x >> 6);
    const int pk = logical_block(P) * wpb + (int)(threadIdx.x >> 6);
    const int n = pk / kPpt;                   // ordinal of the owned tile this packet belongs to
    const int w = pk % kPpt;
    const int sub = w >> 2, wave = w & 3;
    const bool in_launch = n < P.n_tiles;
    const int t = P.rank + n * P.world;
    const int ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
    const int ray = (int)(threadIdx.x & 63) / K;
    int tpx, tpy;  // pixel inside the tile
    if constexpr (K == 4) {
        tpx = ((sub & 7) << 3) + ((wave & 1) << 2) + (ray & 3);
        tpy = ((sub >> 3) << 3) + ((wave >> 1) << 2) + (ray >> 2);
    } else {
        tpx = ((sub & 3) << 4) + ((wave & 1) << 3) + (ray & 7);
        tpy = ((sub >> 2) << 3) + ((wave >> 1) << 2) + (ray >> 3);
    }
    s.px = tx * kTile + tpx;
    s.py = ty * kTile + tpy;
    s.in_launch = in_launch;
    s.active = in_launch && (s.px < P.W) && (s.py < P.H) && (P.only_tile < 0 || P.only_tile == n);
    s.out_index = P.packed ? (n * (kTile * kTile) + tpy * kTile + tpx) : (s.py * P.W + s.px);
    return s;
}

// One step of the ordered blend: the lanes of a ray take the result of depth slot J.  dst is kept as two register
// pairs (xy, zw) so that the blend is two packed multiplies and two packed adds.
template <int V, int K, int J>
__device__ __forceinline__ void dp_blend_slot(int flags, v2f s_rg, v2f s_ba, v2f& dxy, v2f& dzw, bool& alive, unsigned& blends,
                                              unsigned& fetched)
{
    const int f = slot_bcast<K, J>(flags);
    const v2f rg = v2f{slot_bcast<K, J>(s_rg.x), slot_bcast<K, J>(s_rg.y)};
    const v2f ba = v2f{slot_bcast<K, J>(s_ba.x), slot_bcast<K, J>(s_ba.y)};
    const bool counted = alive && (f & 1);   // in the sample box: the reference executes the blend
    const bool real = counted && (f & 2);    // ... and it is not a provable identity
    blends += counted ? 1u : 0u;
    fetched += real ? 1u : 0u;
    const float om = 1.0f - dzw.y;           // FrontToBackBlend, src already (rgb * a, a)
    const v2f nxy = mad2(rg, v2f{om, om}, dxy), nzw = mad2(ba, v2f{om, om}, dzw);
    dxy = real ? nxy : dxy;
    dzw = real ? nzw : dzw;
    const bool cut = real && !can_blend<V>(dzw.y);        // cut-off reached: no later step can blend
    const bool left = alive && !(f & 1) && (f & 4);      // past the far side of the sample box
    alive = alive && !cut && !left;
}

// The whole depth-parallel march of the rays of one wavefront: lane = ray * K + depth slot, `slot` = the ray's pixel.  Shared by
// march_dp_kernel and the mixed kernel (vr_mixed.h: two lanes per ray for the packets with the longest chains only).  Depth
// slot 0 of every ray holds (like the others) the ray's result and counts on return.
template <int V, bool OFF32, bool SKIP, int K, bool PIPE>
__device__ __forceinline__ void march_dp_body(const MarchParams& P, const PixelSlot& slot, float4& dst, unsigned& blends,
                                              unsigned& covered, unsigned& fetched)
{
    const int j = threadIdx.x & (K - 1);  // depth slot
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
                // the per-pixel prologue of march_kernel, unchanged
                float step_size = P.step_size;
                if constexpr (V == V_LIGHT) {  // CalculateWorldStep before the override
                    wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.5f));
                    wstep.z = wstep.z * (-1.0f);
                }
                if (P.toggle_varstep == 1) step_size = ray_len / (float)P.steps_count;
                p = ray.start;
                if (P.toggle_jitter == 1) {
                    float jt = jitter((float)slot.px + 0.5f, (float)slot.py + 0.5f);
                    p = mk3(p.x + (dir.x * step_size) * jt, p.y + (dir.y * step_size) * jt, p.z + (dir.z * step_size) * jt);
                }
                step = mk3(dir.x * step_size, dir.y * step_size, dir.z * step_size);
                if constexpr (V == V_MULTI_CTRT) {  // CalculateWorldStep after the override
                    wstep = mk3(dir.x * (step_size * 1.0f), dir.y * (step_size * 1.0f), dir.z * (step_size * 0.7f));
                    wstep.z = wstep.z * (-1.0f);
                }
                if constexpr (V == V_VOLUME_MASK || V == V_THREE_FILES) wstep = step;
                w = ray.world0;
                n_inside = steps_inside(p, step, bx0, by0, bz0, bx1, by1, bz1);
                alive = true;
            }
        }
    }

    // ---- the march: every lane of the wavefront runs the loop (rays that are done are predicated off), so the
    // quad broadcasts and the wavefront votes below always see all their lanes
    constexpr bool kW = (V != V_BASIC && V != V_TF_CALIB);  // the shader uses the world position
    // (The approach loop of march_p2_kernel / march_packet -- no look-ups outside the box of the active bricks, identity steps as plain
    // additions until a ray stands in an active brick -- was built here too, bit-exact, and is no gain for these kernels: a rank's quarter
    // of C3 with four lanes per ray 0.251 -> 0.285 ms, its eighth 0.193 -> 0.201, with two lanes level (tools/experiments/r5c.sh).)
    // depth slot j starts j steps down the ray: the same rounded additions the one-lane loop performs
#pragma unroll
    for (int k = 0; k < K - 1; ++k) {
        if (k < j) {
            p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
            if constexpr (kW) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
        }
    }
    unsigned D = 0;  // distance-field byte of p: 0 = sample, k >= 1 = identity, and so is everything within k-1 bricks
    if constexpr (SKIP) {
        D = dist_at(P, brick_of(P, p));
        if constexpr (PIPE) asm volatile("" : "+v"(D));
    }
    const int lim = min(n_inside, P.steps_count);  // runs stay inside the provably-in-box prefix
    // steps every marching ray of the wavefront is certainly inside the box for (wave-uniform): before that step the box test
    // below is skipped by a scalar branch instead of being evaluated into an empty lane mask every round (vr_pw.h)
    int n_in_w;
    {
        int v = alive ? n_inside : 0x7fffffff;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
        n_in_w = __builtin_amdgcn_readfirstlane(v);
    }
    float leap_c = 0.0f;
    if constexpr (SKIP) {
        const float vmax = fmaxf(fmaxf(fabsf(step.x) * P.bsx, fabsf(step.y) * P.bsy), fabsf(step.z) * P.bsz);
        leap_c = 0.999f / vmax;  // see march_kernel
    }

    v2f dxy = v2f{0.0f, 0.0f}, dzw = dxy;  // dst of the marching rays (fragment modes 1-4 keep theirs in `dst`)
    bool have = false;  // this slot sampled last round: the corners of its current position are requested (F4)
    Fetch4 F4;
    float wfx = 0.0f, wfy = 0.0f, wfz = 0.0f;
    int base = 0;  // step index of depth slot 0 (wave-uniform)
    while (base < P.steps_count && vr_ballot(alive) != 0) {
        const int my = base + j;
        // next round: K more rounded additions; its distance-field byte is requested now, used at the bottom
        f3 pn = p, wn = w;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            pn = mk3(pn.x + step.x, pn.y + step.y, pn.z + step.z);
            if constexpr (kW) wn = mk3(wn.x + wstep.x, wn.y + wstep.y, wn.z + wstep.z);
        }
        unsigned Dn = 0;
        if constexpr (SKIP) Dn = dist_at(P, brick_of(P, pn));

        if constexpr (SKIP) {
            // wave-uniform run of identity steps (march_kernel): every slot of every ray that is still marching
            // has at least 4 safe steps -> all of them advance by the same count; the ray passes steps
            // [base, base + mw), all inside the inert neighbourhood of slot 0 and inside the box
            int m = 1 << 30;
            if (alive) m = (D >= 2) ? min((int)fminf(((float)D - (1.0f + kBrickHalf)) * leap_c, 64.0f), lim - my - 1) : 0;
            if (vr_ballot(m < 4) == 0) {
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
                        if constexpr (kW) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                }
                base += mw;
                blends += alive ? (unsigned)mw : 0u;
                D = dist_at(P, brick_of(P, p));
                if constexpr (PIPE) asm volatile("" : "+v"(D));  // no path reaches the loop head with a load pending
                have = false;
                continue;
            }
        }

        // this slot's step
        const bool valid = alive && my < P.steps_count;
        bool inb = valid, gone = false;
        if (base + K > n_in_w && my >= n_inside) {
            inb = valid && p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
            // p moves monotonically per component: once past the far bound it never returns
            gone = valid && !inb &&
                   ((step.x >= 0.0f && p.x > bx1) || (step.x <= 0.0f && p.x < bx0) || (step.y >= 0.0f && p.y > by1) ||
                    (step.y <= 0.0f && p.y < by0) || (step.z >= 0.0f && p.z > bz1) || (step.z <= 0.0f && p.z < bz0));
        }
        const bool real = inb && (!SKIP || D == 0);
        v2f s_rg = v2f{0.0f, 0.0f}, s_ba = s_rg;
        // (PIPE: the look-ahead byte is taken here, on the straight-line path: a wait placed behind the divergent
        // sampling block could not tell the paths apart and would wait for the corner loads issued inside it)
        if constexpr (PIPE && SKIP) asm volatile("" : "+v"(Dn));
        if (real) {
            if constexpr (PIPE && V == V_LIGHT) {
                // The corners of this position were requested at the end of the previous round (or are requested now,
                // on entering tissue); the next round's are requested after this round's table texels, so the wait for
                // the texels leaves them in flight behind the shading, the ordered blend and the next loop head: a
                // launch small enough to need this kernel waits on memory latency, not on issue slots.
                if (!have) fetch_rgba<OFF32>(P.vol[0], p, F4, wfx, wfy, wfz);
                const v2f zw = interp_zw(F4, wfx, wfy, wfz);
                TfFetch tq = tf_fetch(P.tf[0], zw.y);
                const v2f xy = interp_xy(F4, wfx, wfy, wfz);
                const f3 grad = mk3(xy.x, xy.y, zw.x);
                __builtin_amdgcn_sched_barrier(0);  // the old corners are dead here: same registers
                fetch_rgba<OFF32>(P.vol[0], pn, F4, wfx, wfy, wfz);
                __builtin_amdgcn_sched_barrier(0);
                tf_pin(tq);
                const TfSample t = tf_finish(tq);
                const f3 N = normalize3(grad);
                const f3 sh = shade(N, w, mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]),
                                    mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                                    mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]), 2.5f, 0.5f);
                const f3 col = mk3(t.rgb.x * sh.x, t.rgb.y * sh.y, t.rgb.z * sh.z);
                s_rg = v2f{col.x * t.opacity, col.y * t.opacity};
                s_ba = v2f{col.z * t.opacity, t.opacity};
            } else {
                const Src s = sample_src<V, OFF32>(P, p, w);
                s_rg = v2f{s.rgb.x * s.a, s.rgb.y * s.a};
                s_ba = v2f{s.rgb.z * s.a, s.a};
            }
        }
        have = real;
        const int flags = (inb ? 1 : 0) | (real ? 2 : 0) | (gone ? 4 : 0);
        // the K results, in step order
        dp_blend_slot<V, K, 0>(flags, s_rg, s_ba, dxy, dzw, alive, blends, fetched);
        dp_blend_slot<V, K, 1>(flags, s_rg, s_ba, dxy, dzw, alive, blends, fetched);
        if constexpr (K == 4) {
            dp_blend_slot<V, K, 2>(flags, s_rg, s_ba, dxy, dzw, alive, blends, fetched);
            dp_blend_slot<V, K, 3>(flags, s_rg, s_ba, dxy, dzw, alive, blends, fetched);
        }

        p = pn;
        w = wn;
        D = Dn;
        base += K;
    }

    if (P.fragment_mode < 1 || P.fragment_mode > 4) dst = make_float4(dxy.x, dxy.y, dzw.x, dzw.y);
}

template <int V, bool OFF32, bool SKIP, int K, bool PIPE, bool BATCH = false>
__global__ __launch_bounds__(256) void march_dp_kernel(const MarchBatch B)
{
    const MarchParams& P = frame_params<BATCH>(B);
    const unsigned long long t_start = wall_clock64();
    const PixelSlot slot = map_pixel_dp<K>(P);
    const int j = threadIdx.x & (K - 1);  // depth slot
    float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned blends = 0, covered = 0, fetched = 0;
    march_dp_body<V, OFF32, SKIP, K, PIPE>(P, slot, dst, blends, covered, fetched);
    // slot 0 of every ray holds (like the others) the ray's result and counts
    if (j == 0 && (slot.active || (P.packed && slot.in_launch))) P.out[slot.out_index] = dst;
    store_block_counts(P, j == 0 ? blends : 0u, j == 0 ? covered : 0u, j == 0 ? fetched : 0u, t_start);
}

}  // namespace VR_KNS
