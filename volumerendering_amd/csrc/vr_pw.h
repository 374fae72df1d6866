// vr_pw.h -- persistent-wavefront march kernel: the workgroups stay on their compute units for the whole launch and their
// wavefronts take 8x8 pixel packets from a queue, one after the other, instead of one packet per dispatched workgroup.
//
// What that buys (CDNA4: 160 KiB of LDS per CU, 8 XCDs with an L2 each):
//  * the transfer function of slot 0 lives in LDS.  The reference's lit scene has 4096-entry tables
//    (App/src/miniapps/BasicVolLightApp.cpp:29-30): 16 KiB of opacity + 64 KiB of colour, more than a CU's 32 KiB L1, fetched
//    per sample behind the other wavefronts' corner loads -- the second of the two dependent memory round trips of a step
//    (BasicVolLightApp.wgsl:210-216: the volume sample, then the two table samples its density selects).  A workgroup of 16
//    wavefronts copies the padded tables into LDS once, merged to one float4 (r, g, b, opacity) per entry (64 KiB + 32 B), and
//    every look-up is two ds_read_b128: 40 of the 168 bytes per sample leave the texture-addresser path and the
//    dependent chain of a step loses an L1 / L2 round trip.
//  * the launch order of DESIGN 4.6 (longest ray chains first) becomes a real queue: a wavefront that finishes a packet
//    takes the next one of its class at once (one returning atomic add on one of eight heads, MI355X_MICROARCH.md "dequeue":
//    0.3 us idle, 1.1-1.3 us under load; 4 080 per head and C3 frame), no workgroup is created or retired in between, and
//    nothing waits for the slowest wavefront of a workgroup.
//
// Queue: the logical blocks (8x8 packets, 64 per 64x64 tile; map_pixel_at with one wavefront per block) are split into the
// 8 classes of their index modulo 8, as march_kernel's workgroups are -- a workgroup's class is blockIdx.x % 8, the blocks
// b and b + 8 share an XCD under the observed round-robin placement (speed only, never correctness), and MarchParams::order
// holds a class's blocks at positions 8 i + class, longest chain first.  heads[class] counts the items handed out beyond the
// first one of every wavefront (which is static: no atomic at the start of the launch, and the longest packets are dealt
// over the CUs instead of landing on one).  Every wavefront leaves the loop as soon as the index it draws is past the end
// of its class: a wrong head value can repeat or drop packets (tests would see it) but never hang a wavefront.
// The heads are zero at launch: order_blocks_kernel, which runs behind every ordered launch, clears them for the next user
// of the slot; launches without a sort behind them are preceded by a memset (vr_api.hip).
//
// Arithmetic, positions, blend order and counts are march_packet's (vr_kernels.h): bit-identical frames and records.
#pragma once
#include "vr_kernels.h"
#include <type_traits>

namespace VR_KNS {
using namespace vr;

constexpr int kPwThreads = 1024;      // one workgroup per CU: 16 wavefronts, 4 per SIMD (<= 128 VGPRs each)
constexpr int kPwHeadStride = 64;     // the eight heads are 256 B apart (a memory channel each)

// PIPE (lit and unlit shader): the next step's eight corner loads are issued before this step's shading (march_packet's
// software-pipelined loop form): with the table texels coming from LDS the corner fetch is the one memory round trip left
// in a step, and it then overlaps the arithmetic of the step before.
template <int V, bool OFF32, bool SKIP, bool LTF, bool PIPE>
__global__ __launch_bounds__(kPwThreads) void march_pw_kernel(const MarchBatch B, const PwQueue Q)
{
    const MarchParams& P = B.frame[0];
    if constexpr (LTF) {
        const int n = P.tf[0].res_o + 2;  // (res_c == res_o: the host's condition for LTF)
        for (int j = (int)threadIdx.x; j < n; j += kPwThreads) {
            float4 c = P.tf[0].color[j];
            c.w = P.tf[0].opacity[j];
            vr_lds_tf[j] = c;
        }
        __syncthreads();  // the only barrier: from here on the wavefronts are independent of each other
    }
    const unsigned cls = blockIdx.x & 7u;
    const unsigned groups = (gridDim.x - cls + 7u) >> 3;  // workgroups of this class
    const unsigned wib = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wpb = kPwThreads / 64;
    const unsigned n_c = Q.n_items >> 3;                  // items per class (n_items is a multiple of 8)
    // first item: static, wavefront k of every workgroup before wavefront k + 1 of any (the order is longest first)
    unsigned idx = wib * groups + (blockIdx.x >> 3);
    // A wavefront whose class has run dry goes on with the next class's queue (Q.steal; a class is an XCD's share of the
    // packets: with whole tiles per class -- xcd_mode 0: the 64 packets of a tile share an L2 -- the classes' work differs by a
    // factor of 1.5, which stealing evens out at the end of the launch).  Every item is still handed out exactly once: item
    // index = the head's old value + the number of that class's static items.
    unsigned cur = cls, tried = 0;
    for (;;) {
        if (idx >= n_c) {
            if (!Q.steal || ++tried >= 8u) break;
            cur = (cur + 1u) & 7u;
            const unsigned groups_o = (gridDim.x - cur + 7u) >> 3;
            unsigned r = 0;
            if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cur * kPwHeadStride, 1u);
            idx = groups_o * wpb + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
            continue;
        }
        tried = 0;
        const unsigned pos = (idx << 3) | cur;
        int lb = (int)pos;
        if (P.order != nullptr) lb = __builtin_amdgcn_readfirstlane((int)P.order[pos]);
        const unsigned long long t_start = wall_clock64();
        const PixelSlot slot = map_pixel_at(P, lb, 1, 0);
        float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned blends = 0, covered = 0, fetched = 0;
        march_packet<V, OFF32, SKIP, PIPE ? 2 : (SKIP ? 3 : 0), false, LTF>(P, slot, dst, blends, covered, fetched);
        if (slot.active || (P.packed && slot.in_launch)) P.out[slot.out_index] = dst;
        store_wave_counts(P, lb, blends, covered, fetched, t_start);
        unsigned r = 0;
        if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cur * kPwHeadStride, 1u);
        const unsigned groups_cur = (gridDim.x - cur + 7u) >> 3;
        idx = groups_cur * wpb + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    }
}

// ---- two steps ahead (flavours 16 and 17) ------------------------------------------------------------------------
// Everything in a step but the blend is independent of the step before (positions are known in advance), so the remedy for a
// frame made of latencies is a deeper software pipeline: two corner buffers (A = even steps, B = odd steps, the loop unrolled by
// two so that which registers hold which step is static), the corners of step i + 2 requested as soon as step i's have been
// interpolated, table texels from LDS.  16 = every ray samples from its entry into the box to its cut-off or exit (the host
// picks it for volumes with nothing to skip); 17 = with empty-space skipping decided ahead of the loads (SKIP, below).
// Speculative loads of positions a ray never reaches read clamped, valid voxels and are dropped.  The loads are raw buffer
// loads the compiler tracks (it places the s_waitcnt); kP2Threads keeps the register budget wide enough (168 VGPRs) that the
// allocator has no reason to move a buffer while its loads are in flight.  Arithmetic per sample: light_shade_blend / blend.
constexpr int kP2Threads = 768;  // at most 12 wavefronts per CU, 3 per SIMD (163 - 168 VGPRs: two corner buffers are 64 of them)

// Branch-free cell of a BRICKED volume: clamp-to-edge texel pairs on every axis, separable index (make_cell's arithmetic
// without its wave-uniform fast paths: a branch between address arithmetic and loads defeats the wait-count pass).
__device__ __forceinline__ Cell make_cell_bricked(const DevVolume& v, f3 p)
{
    const float x = mad(p.x, (float)v.nx, -0.5f), y = mad(p.y, (float)v.ny, -0.5f), z = mad(p.z, (float)v.nz, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    Cell c;
    c.fx = x - x0;
    c.fy = y - y0;
    c.fz = z - z0;
    int i0, i1, j0, j1, k0, k1;
    texel_pair(x0, v.nx, i0, i1);
    texel_pair(y0, v.ny, j0, j1);
    texel_pair(z0, v.nz, k0, k1);
    const unsigned ax0 = ((unsigned)i0 >> kVbS) * kVbN + ((unsigned)i0 & kVbM), ax1 = ((unsigned)i1 >> kVbS) * kVbN + ((unsigned)i1 & kVbM);
    // (24-bit multiplies -- full rate, the 32-bit one is a quarter -- : the host keeps brick_row and brick_slab below 2^24)
    const unsigned ay0 = __umul24((unsigned)j0 >> kVbS, v.brick_row) + (((unsigned)j0 & kVbM) << kVbS);
    const unsigned ay1 = __umul24((unsigned)j1 >> kVbS, v.brick_row) + (((unsigned)j1 & kVbM) << kVbS);
    const unsigned az0 = __umul24((unsigned)k0 >> kVbS, v.brick_slab) + (((unsigned)k0 & kVbM) << (2u * kVbS));
    const unsigned az1 = __umul24((unsigned)k1 >> kVbS, v.brick_slab) + (((unsigned)k1 & kVbM) << (2u * kVbS));
    const unsigned r00 = ay0 + az0, r10 = ay1 + az0, r01 = ay0 + az1, r11 = ay1 + az1;
    c.o000 = r00 + ax0; c.o100 = r00 + ax1;
    c.o010 = r10 + ax0; c.o110 = r10 + ax1;
    c.o001 = r01 + ax0; c.o101 = r01 + ax1;
    c.o011 = r11 + ax0; c.o111 = r11 + ax1;
    return c;
}

// requests the eight corners of position q into X, returns the interpolation weights.  MASKED: the lanes that say `idle`
// request nothing; and the distance-field byte of q's brick is asked for just ahead of the corners (every lane) -- the
// skipping's bricks ARE the layout's bricks (brick_of(q) is the base cell's brick: see its comment), so the byte's index is the
// base corner's slot without its six intra-brick bits, for one shift instead of brick_of's twelve instructions.
template <int V, bool MASKED = false, typename T>
__device__ __forceinline__ void p2_request(const MarchParams& P, const DevVolume& vol, __amdgpu_buffer_rsrc_t rsrc, f3 q, T (&X)[8], float& fx,
                                           float& fy, float& fz, bool idle, unsigned& dbyte)
{
    constexpr unsigned kShift = (V == V_LIGHT) ? 4u : 2u;  // bytes per element
    const Cell c = make_cell_bricked(vol, q);
    if constexpr (MASKED) {
        if constexpr (kBrickShift == (int)kVbS) dbyte = dist_at(P, (int)(c.o000 >> (3u * kVbS)));
        else dbyte = dist_at(P, brick_of<true>(P, q));
    }
    fx = c.fx;
    fy = c.fy;
    fz = c.fz;
    unsigned o[8] = {c.o000 << kShift, c.o100 << kShift, c.o010 << kShift, c.o110 << kShift,
                     c.o001 << kShift, c.o101 << kShift, c.o011 << kShift, c.o111 << kShift};
    // The idle lanes are switched off for the eight loads by hand: the compiler does not see a branch (so it keeps no execz jump
    // and the wait counts stay exact), the texture addresser does not see the lanes.  Nothing but the loads runs in between:
    // the offsets are pinned into registers first, and the scheduler is fenced on both sides.
    unsigned long long exec_saved = 0;
    if constexpr (MASKED) {
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
        const unsigned long long keep = vr_ballot(!idle);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1" : "=&s"(exec_saved) : "s"(keep) : "scc");
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if constexpr (V == V_LIGHT) X[k] = __builtin_bit_cast(vr_f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)o[k], 0, 0));
        else X[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o[k], 0, 0));
    }
    if constexpr (MASKED) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 exec, %0" : : "s"(exec_saved));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The per-step vote of sample_and_blend (opacity_is_zero for every sampling ray) as ONE compare: the table index that decides,
// with "not finite" and "does not sample" folded into its value.
__device__ __forceinline__ bool p2_vote(const MarchParams& P, bool sampled, float d)
{
    int j = padded_texel(floorf(mad(d, (float)P.tf[0].res_o, -0.5f)), P.tf[0].res_o);
    j = (d - d == 0.0f) ? j : 0x7fffffff;          // an infinite density has a NaN weight, hence a NaN opacity: never "zero"
    j = sampled ? j : (int)0x80000000;             // a ray that does not sample never asks for the shading
    asm volatile("" : "+v"(j));                    // (kept as a value: the compiler would turn the compare back into mask logic)
    return vr_ballot(j > P.zskip_prefix) != 0;
}

// SKIP: empty-space skipping on top of it.  One distance-field byte per ray rides along with each corner buffer: the byte of
// the exact position whose corners are in flight, asked for just ahead of them and read a trip later.  It says whether the
// step blends (an inert brick: the identity, march_packet's test), whether the two steps after it need their corners at all
// (the ray's safe steps in inert bricks reach them: its lanes are switched off for those loads), and how many steps after it every ray of the
// packet can skip: then the REQUESTS jump (4 .. 64 rounded additions, the identity steps of march_packet's runs) while the two
// steps already in flight are still being consumed -- nothing in flight is thrown away and no latency is exposed.  A step in
// which no ray blends interpolates nothing; the per-step vote (every opacity zero for certain: no texels, no gradient, no
// shading) is march_packet's.
template <int V, bool SKIP>
__global__ __launch_bounds__(kP2Threads) void march_p2_kernel(const MarchBatch B, const PwQueue Q)
{
    static_assert(V == V_LIGHT || V == V_BASIC, "lit / unlit shader");
    const MarchParams& P = B.frame[0];
    {
        const int n = P.tf[0].res_o + 2;  // (res_c == res_o: the host's condition)
        for (int j = (int)threadIdx.x; j < n; j += (int)blockDim.x) {
            float4 c = P.tf[0].color[j];
            c.w = P.tf[0].opacity[j];
            vr_lds_tf[j] = c;
        }
        __syncthreads();
    }
    const DevVolume& vol = P.vol[0];
    // the gather source: the bricked vec4 voxels (lit) or the bricked density plane (unlit), as a raw buffer (< 4 GiB)
    const __amdgpu_buffer_rsrc_t rsrc =
        (V == V_LIGHT) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(vol.data), 0, (int)vol.data_bytes, 0x00020000)
                       : __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vol.a_base), 0, (int)(vol.data_bytes >> 2), 0x00020000);
    const unsigned cls = blockIdx.x & 7u;
    const unsigned groups = (gridDim.x - cls + 7u) >> 3;
    const unsigned wib = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wpb = blockDim.x >> 6;
    const unsigned n_c = Q.n_items >> 3;
    unsigned idx = wib * groups + (blockIdx.x >> 3);
    const float bx0 = P.bmin[0], by0 = P.bmin[1], bz0 = P.bmin[2];
    const float bx1 = P.bmax[0], by1 = P.bmax[1], bz1 = P.bmax[2];
    typedef typename std::conditional<V == V_LIGHT, vr_f4, float>::type Elem;
    unsigned cur = cls, tried = 0;  // (stealing between classes: march_pw_kernel's)
    for (;;) {
        if (idx >= n_c) {
            if (!Q.steal || ++tried >= 8u) break;
            cur = (cur + 1u) & 7u;
            const unsigned groups_o = (gridDim.x - cur + 7u) >> 3;
            unsigned r = 0;
            if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cur * kPwHeadStride, 1u);
            idx = groups_o * wpb + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
            continue;
        }
        tried = 0;
        const unsigned pos = (idx << 3) | cur;
        int lb = (int)pos;
        if (P.order != nullptr) lb = __builtin_amdgcn_readfirstlane((int)P.order[pos]);
        const unsigned long long t_start = wall_clock64();
        const PixelSlot slot = map_pixel_at(P, lb, 1, 0);
        float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned blends = 0, covered = 0, fetched = 0;
#if VR_P2_DEBUG
        unsigned dbg_trips = 0, dbg_sampled = 0, dbg_shaded = 0, dbg_jumps = 0;
#endif
        bool alive = false;
        f3 p = mk3(0.0f, 0.0f, 0.0f), w = p, step = p, wstep = p;
        int n_inside = 0;
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
                    w = ray.world0;
                    n_inside = steps_inside(p, step, bx0, by0, bz0, bx1, by1, bz1);
                    alive = P.steps_count > 0;
                }
            }
        }
        if (vr_ballot(alive) != 0) {  // (wave-uniform: from here on every lane executes every statement)
            // A wavefront issues its instructions in order, one at a time: about 5 cycles a vector instruction, 8 a scalar one, 29
            // a compare whose mask a scalar instruction combines (tools/ubench/valu_issue.hip, 3 wavefronts per SIMD) -- with the
            // loads two steps ahead the loop's own instruction stream is the step's latency, and mask logic is its dearest part.
            // Hence: no per-step bookkeeping that a trip (two steps) can do once, wave-uniform choices wherever the result is the
            // same, and no box test at all in this loop: it runs while every marching ray is provably inside the box and in time
            // (a wave-minimum of the rays' own counts says how long); the last steps of a packet -- rays leave the box a few
            // steps apart -- are taken by a plain loop behind it.
            //
            // Two corner buffers: A = even steps, Bq = odd steps of a trip, the loop unrolled by two so that which registers hold
            // which step is static; a buffer is written by requests inside the loop only (no prologue that loads them: the
            // values entering the loop and the values coming round the back edge would be different registers, and the copies on
            // the back edge need the data -- every trip ended in s_waitcnt vmcnt(0)).  pA / pB are the positions whose corners
            // are in flight into A / Bq, each one rounded addition of `step` after the other: the positions the shader's loop
            // has at those steps, exactly.
            Elem A[8], Bq[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (V == V_LIGHT) {
                    A[k] = vr_f4{0.0f, 0.0f, 0.0f, 0.0f};
                    Bq[k] = vr_f4{0.0f, 0.0f, 0.0f, 0.0f};
                } else {
                    A[k] = 0.0f;
                    Bq[k] = 0.0f;
                }
            }
            float afx = 0.0f, afy = 0.0f, afz = 0.0f, bfx = 0.0f, bfy = 0.0f, bfz = 0.0f;
            f3 pA = p, pB = p;
            int i = 0;  // the step pA is at (wave-uniform)
            // steps every marching ray of the packet is certainly in time and inside the box for (wave-uniform): before that
            // step no ray needs the box test
            int n_in_w;
            {
                int v = alive ? min(n_inside, P.steps_count) : 0x7fffffff;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
                n_in_w = __builtin_amdgcn_readfirstlane(v);
            }
            unsigned DA = 0, DB = 0;  // the distance-field bytes of the bricks of pA and pB (0 = active, n = n bricks from an active one)
            float leap_c = 0.0f;
            int lim = 0;
            if constexpr (SKIP) {
                // steps a ray at distance-field value D can take while it certainly stays within D-1 bricks of its brick on every
                // axis (march_packet's kRun); jumps stay inside the provably-in-box prefix of the ray
                const float vmax = fmaxf(fmaxf(fabsf(step.x) * P.bsx, fabsf(step.y) * P.bsy), fabsf(step.z) * P.bsz);
                leap_c = 0.999f / vmax;
                lim = min(n_inside, P.steps_count);
            }
            bool idle_a = false, idle_b = false;      // this trip: step A / B blends nothing
            bool idle_ra = false, idle_rb = false;    // ... requests nothing
            int mw = 0;                               // identity steps skipped between this trip's two steps and the next trip's
            // consumes the corners in X (of position pX, step ix), then requests into X the corners of pY + step [+ mw steps], which
            // becomes pX
            auto one_step = [&](Elem (&X)[8], float& xfx, float& xfy, float& xfz, f3& pX, const f3& pY, int jump, bool idle_con, bool idle_rq, unsigned& DX) {
                const bool inb = alive;  // (in time and inside the box: the loop's condition)
                const bool sampled = SKIP ? (inb && !idle_con) : inb;
                v2f zw = v2f{0.0f, 0.0f}, gxy;
                TfFetch tq;
                // (wave-uniform) a step in which no ray samples interpolates nothing
                bool shaded = !SKIP || vr_ballot(!idle_con) != 0;  // (idle_con covers the rays that had finished when the trip began)
#if VR_P2_DEBUG
                if (shaded) ++dbg_sampled;
#endif
                if (!shaded) {
                } else if constexpr (V == V_LIGHT) {
                    Fetch4 q;
                    q.a = make_float4(X[0].x, X[0].y, X[0].z, X[0].w); q.b = make_float4(X[1].x, X[1].y, X[1].z, X[1].w);
                    q.d = make_float4(X[2].x, X[2].y, X[2].z, X[2].w); q.e = make_float4(X[3].x, X[3].y, X[3].z, X[3].w);
                    q.f = make_float4(X[4].x, X[4].y, X[4].z, X[4].w); q.g = make_float4(X[5].x, X[5].y, X[5].z, X[5].w);
                    q.h = make_float4(X[6].x, X[6].y, X[6].z, X[6].w); q.i = make_float4(X[7].x, X[7].y, X[7].z, X[7].w);
                    zw = interp_zw(q, xfx, xfy, xfz);
                    // (the per-step vote of sample_and_blend: when every ray's opacity is zero for certain, the texels, the
                    // gradient and the shading are left out -- the blend would be the identity)
                    if constexpr (SKIP) shaded = p2_vote(P, sampled, zw.y);
                    if (shaded) {
                        tq = tf_fetch_lds(P.tf[0], zw.y);
                        gxy = interp_xy(q, xfx, xfy, xfz);
                    }
                } else {
                    Fetch1 q;
                    q.a = X[0]; q.b = X[1]; q.d = X[2]; q.e = X[3]; q.f = X[4]; q.g = X[5]; q.h = X[6]; q.i = X[7];
                    zw.y = interp_a(q, xfx, xfy, xfz);
                    if constexpr (SKIP) shaded = p2_vote(P, sampled, zw.y);
                    if (shaded) tq = tf_fetch_lds(P.tf[0], zw.y);
                }
                // the position of the next request into X
                pX = mk3(pY.x + step.x, pY.y + step.y, pY.z + step.z);
                for (int k = 0; k < jump; ++k) pX = mk3(pX.x + step.x, pX.y + step.y, pX.z + step.z);
                // Everything that reads the old corners must be COMPUTED here, before their registers are loaded again: left alone,
                // the compiler sinks the gradient's interpolation into the `if (sampled)` below (its only user), the old corners
                // then live across the new loads, the new loads get other registers, and the copies that bring them back at the
                // loop's back edge need the data (s_waitcnt vmcnt(0) every trip).
                if constexpr (V == V_LIGHT) asm volatile("" : "+v"(zw.x), "+v"(zw.y), "+v"(gxy.x), "+v"(gxy.y));
                else asm volatile("" : "+v"(zw.y));
                __builtin_amdgcn_sched_barrier(0);  // the old corners are dead here: the new ones may land in their registers
                // (with the byte of the position requested: the next trip decides with it)
                p2_request<V, SKIP>(P, vol, rsrc, pX, X, xfx, xfy, xfz, SKIP && (idle_rq || !alive), DX);
                __builtin_amdgcn_sched_barrier(0);
#if VR_P2_DEBUG
                if (shaded) ++dbg_shaded;
#endif
                if (sampled) {
                    if (shaded) {
                        if constexpr (V == V_LIGHT) {
                            light_shade_blend<true>(P, w, zw, gxy, tq, dst);
                        } else {
                            const TfSample t = tf_finish(tq);
                            blend(t.rgb, t.opacity, dst);
                        }
                    }
                    ++fetched;
                }
                if (inb) ++blends;
                // cut-off reached: no later step can blend (dst.w changes in a sampled step only: the test is the loop's own)
                alive = alive && can_blend<V>(dst.w);
                if constexpr (V == V_LIGHT) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
            };
            bool start = true;  // (wave-uniform) nothing is in flight yet
#if VR_P2_DEBUG
            dbg_trips = dbg_sampled = dbg_shaded = dbg_jumps = 0;
#endif
            while (i + 2 <= n_in_w && vr_ballot(alive) != 0) {
                if (start) {
                    // the bytes of pA and pB and, without waiting for them, the corners of steps 0 and 1 of every ray
                    pB = mk3(pA.x + step.x, pA.y + step.y, pA.z + step.z);
                    p2_request<V, SKIP>(P, vol, rsrc, pA, A, afx, afy, afz, SKIP && !alive, DA);
                    p2_request<V, SKIP>(P, vol, rsrc, pB, Bq, bfx, bfy, bfz, SKIP && !alive, DB);
                    start = false;
                }
                mw = 0;
                if constexpr (SKIP) {
                    // The bytes are those of the rays' exact positions: a step in an inert brick (byte >= 1) is the identity and
                    // blends nothing (march_packet's test); a trip in which no ray blends interpolates nothing.  Decided AHEAD of the
                    // loads, from the number of steps after pB a ray certainly spends in inert bricks: a ray requests nothing for
                    // a position it reaches within them (its lanes are switched off for the loads); and when every marching ray
                    // has at least four such steps the requests skip them -- the next trip's positions are 4 .. 64 rounded
                    // additions further on (the identity steps of march_packet's runs), with nothing in flight thrown away and
                    // no latency exposed.
                    // (A finished ray is folded into the VALUES -- byte 255, any number of safe steps -- so that every vote below is
                    // the lane mask of ONE compare: a vote on `alive && x < k` costs a mask AND, a v_cndmask and a second compare.)
                    unsigned da = alive ? DA : 255u, db = alive ? DB : 255u;
                    asm volatile("" : "+v"(da), "+v"(db));  // (kept as values: the compiler would turn `da >= 1` back into mask logic)
                    idle_a = da >= 1u;
                    idle_b = db >= 1u;
                    // steps after pB the ray certainly spends in inert bricks (march_packet's run length; < 0 at an active brick)
                    int m = min((int)fminf(((float)db - (1.0f + kBrickHalf)) * leap_c, 64.0f), lim - (i + 1) - 1);
                    m = alive ? m : 64;
                    if (vr_ballot(m < 4) == 0) {
                        mw = 4;
                        if (vr_ballot(m < 8) == 0) {
                            mw = 8;
                            if (vr_ballot(m < 16) == 0) {
                                mw = 16;
                                if (vr_ballot(m < 32) == 0) mw = vr_ballot(m < 64) == 0 ? 64 : 32;
                            }
                        }
                    }
                    idle_ra = m >= mw + 1;  // the positions requested now are steps mw + 1 and mw + 2 after pB
                    idle_rb = m >= mw + 2;
                }
#if VR_P2_DEBUG
                ++dbg_trips;
                if (mw > 0) ++dbg_jumps;
#endif
                one_step(A, afx, afy, afz, pA, pB, mw, idle_a, idle_ra, DA);
                one_step(Bq, bfx, bfy, bfz, pB, pA, 0, idle_b, idle_rb, DB);
                if (mw > 0) {
                    if constexpr (V == V_LIGHT) {
                        for (int k = 0; k < mw; ++k) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                    if (alive) blends += (unsigned)mw;
                }
                i += 2 + mw;
            }
            // The last steps of the packet (pA is the exact position of step i, w its world position): the shader's loop as it
            // stands -- box test, identity steps by the distance-field byte, the far-bound exit -- with no loads ahead.
            if (!start) p = pA;
            for (; i < P.steps_count && vr_ballot(alive) != 0; ++i) {
                if (alive) {
                    bool inb = true;
                    if (i >= n_inside) inb = p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
                    if (inb) {
                        bool sampled = true;
                        if constexpr (SKIP) sampled = dist_at(P, brick_of<true>(P, p)) == 0u;
                        if (sampled) {
                            sample_and_blend<V, true, false, SKIP, true>(P, p, w, dst, mk3(0.0f, 0.0f, 0.0f), 0.0f);
                            ++fetched;
                        }
                        ++blends;
                        if (!can_blend<V>(dst.w)) alive = false;  // cut-off reached: no later step can blend
                    } else {
                        // p moves monotonically per component: once past the far bound it never returns
                        const bool gone = (step.x >= 0.0f && p.x > bx1) || (step.x <= 0.0f && p.x < bx0) || (step.y >= 0.0f && p.y > by1) ||
                                          (step.y <= 0.0f && p.y < by0) || (step.z >= 0.0f && p.z > bz1) || (step.z <= 0.0f && p.z < bz0);
                        if (gone) alive = false;
                    }
                }
                p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                if constexpr (V == V_LIGHT) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
            }
        }
        if (slot.active || (P.packed && slot.in_launch)) P.out[slot.out_index] = dst;
        store_wave_counts(P, lb, blends, covered, fetched, t_start);
#if VR_P2_DEBUG
        if ((threadIdx.x & 63) == 0)  // (debug build: the `fetched` word carries the loop's own counters instead)
            P.block_counts[(size_t)lb * kBlockRecord + 2] = (unsigned long long)(dbg_trips & 0xfffu) | ((unsigned long long)(dbg_sampled & 0xfffu) << 12) |
                                                           ((unsigned long long)(dbg_shaded & 0xfffu) << 24) | ((unsigned long long)(dbg_jumps & 0xfffu) << 36);
#endif
        unsigned r = 0;
        if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cur * kPwHeadStride, 1u);
        const unsigned groups_cur = (gridDim.x - cur + 7u) >> 3;
        idx = groups_cur * wpb + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    }
}

}  // namespace VR_KNS
