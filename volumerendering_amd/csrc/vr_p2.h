// vr_p2.h -- march_p2_kernel: persistent wavefronts (vr_pw.h's queue) whose corner loads run TWO steps ahead of the blend
// (flavours 16 / 17).  Round 4 took it from "lit / unlit shader, below 4 GiB, one frame per launch" to every
// BASELINE configuration and every way a caller drives it:
//   * the gather is an INDEXED buffer load (buffer_load_dwordx4 ... idxen, stride 16 / 4 in the descriptor): the VGPR holds
//     the voxel's SLOT in the bricked copy, not a byte offset -- no shift per corner (tools/ubench/struct_buffer.hip: same
//     rate as the raw form);
//   * the slot arithmetic of a trilinear cell comes from a per-axis TABLE IN LDS: entry t + 1 of axis a holds the slot
//     terms of the clamp-to-edge texel pair (clamp(t), clamp(t + 1)) -- texel_pair() and make_cell_bricked()'s shifts, masks
//     and 24-bit multiplies (about 45 vector instructions per request) become one clamp, one address and one ds_read_b64 per
//     axis; the wavefront issues its instructions in order, so the instruction count of a step IS its latency;
//   * WIN: volumes of 4 GiB and more (BASELINE config 5: 1024^3 = 16 GiB of vec4 voxels).  index x stride wraps at 32 bits in
//     the hardware (measured: struct_buffer.hip), so the descriptor's base moves instead: a wave-uniform WINDOW of whole
//     z-slabs of bricks (a packet's rays are a few voxels apart at any step; brick-linear order is z-major), re-centred when a
//     requesting ray leaves it -- a handful of times in a packet's life -- and the lanes keep 32-bit slots relative to it;
//   * BATCH: launches of several frames (vr_render_batch_async, a rank's share of four frames): the queue hands out
//     (frame, packet) items, the frames interleaved so that the long packets of every frame start first;
//   * V_VOLUME_MASK (BASELINE config 4): the CT volume is pipelined, mask and dose are fetched on demand behind the per-brick
//     mask record, which rides along with the distance-field byte.
//   * THE APPROACH: in front of the pipelined loop a packet walks its identity steps without asking for anything ahead -- one
//     distance-field byte per ray and the wave's minimum of the steps they allow, as plain rounded additions -- until a ray stands in
//     an active brick; rays outside the uvw box of the active bricks (MarchParams::abox, steps_near_box) ask for nothing at all, and
//     behind that box the pipelined loop is left.  The packets that cross the volume without ever meeting an active brick (C3: 64 %
//     of the packets that cross it, 22 % of the frame's wavefront time when their jumps were trips of the pipelined loop) end
//     there: C3 one frame at a time 0.522 -> 0.476 ms.
// Arithmetic, positions, blend order and counts are march_packet's (vr_kernels.h): bit-identical frames and records.
#pragma once
#include "vr_pw.h"

namespace VR_KNS {
using namespace vr;

#ifndef VR_P2_JUMP_MAX
#define VR_P2_JUMP_MAX 1024
#endif
constexpr int kP2JumpMax = VR_P2_JUMP_MAX;  // identity steps one trip may skip (each three rounded additions per ray, six with a world position)
#ifndef VR_P2_EXIT
#define VR_P2_EXIT 1
#endif
constexpr bool kP2Exit = kApproach && VR_P2_EXIT != 0;  // ... and the same knowledge behind the box of the active bricks (A/B: -DVR_P2_EXIT=0)
constexpr bool kP2Approach = kApproach;  // the approach loop in front of the pipelined loop (-DVR_APPROACH=0: A/B builds)
constexpr int kP2Threads = 768;  // at most 12 wavefronts per CU, 3 per SIMD (two corner buffers are 64 of ~168 VGPRs)
constexpr int kP2ThreadsUnlit = 1024;  // the unlit shader's buffers are 4-byte densities (101 VGPRs): 4 wavefronts per SIMD fit

// (indexed buffer loads -- vr_struct_load_b128 / _b32 -- are declared in vr_kernels.h)

// Where the workgroup's LDS holds what (dynamic LDS, vr_lds_tf): the merged transfer function of slot 0, then the three axis
// tables.  Byte offsets of entry t = -1 of each axis; wave-uniform.
struct P2Lds {
    unsigned off_x, off_y, off_z;
    int mx, my, mz;       // n - 1 per axis
    unsigned win_last;    // WIN: largest z term, relative to the window's first slab, whose cell still lies inside the window
    unsigned win_slabs;   // WIN: z-slabs of bricks a window holds
    unsigned win_slots;   // WIN: records a window holds (< 4 GiB / stride; smaller in the tests)
};

__device__ __forceinline__ uint2 lds_pair(unsigned byte_off)
{
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(vr_lds_tf) + byte_off);
}

// The workgroup fills the tables: entry t + 1 (t = -1 .. n - 1) = slot terms of texel_pair(t): (clamp(t, 0, n-1), clamp(t+1, 0, n-1)).
__device__ __forceinline__ void p2_fill_tables(const DevVolume& v, const P2Lds& L)
{
    char* lds = reinterpret_cast<char*>(vr_lds_tf);
    for (int e = (int)threadIdx.x; e <= v.nx; e += (int)blockDim.x) {
        const unsigned i0 = (unsigned)max(e - 1, 0), i1 = (unsigned)min(e, v.nx - 1);
        *reinterpret_cast<uint2*>(lds + L.off_x + (unsigned)e * 8u) =
            make_uint2((i0 >> kVbS) * kVbN + (i0 & kVbM), (i1 >> kVbS) * kVbN + (i1 & kVbM));
    }
    for (int e = (int)threadIdx.x; e <= v.ny; e += (int)blockDim.x) {
        const unsigned j0 = (unsigned)max(e - 1, 0), j1 = (unsigned)min(e, v.ny - 1);
        *reinterpret_cast<uint2*>(lds + L.off_y + (unsigned)e * 8u) =
            make_uint2((j0 >> kVbS) * v.brick_row + ((j0 & kVbM) << kVbS), (j1 >> kVbS) * v.brick_row + ((j1 & kVbM) << kVbS));
    }
    for (int e = (int)threadIdx.x; e <= v.nz; e += (int)blockDim.x) {
        const unsigned k0 = (unsigned)max(e - 1, 0), k1 = (unsigned)min(e, v.nz - 1);
        *reinterpret_cast<uint2*>(lds + L.off_z + (unsigned)e * 8u) =
            make_uint2((k0 >> kVbS) * v.brick_slab + ((k0 & kVbM) << (2u * kVbS)), (k1 >> kVbS) * v.brick_slab + ((k1 & kVbM) << (2u * kVbS)));
    }
}

// What the gather reads through (wave-uniform): the records from `ptr` on (the volume's first; WIN: the window's), `records` of
// them, and -- WIN -- the window's first slot (a multiple of brick_slab).  The descriptor is built from it where the loads are,
// behind v_readfirstlane: under register pressure the compiler parks these in VGPRs, and a descriptor it cannot prove uniform
// costs a waterfall loop per load.
struct P2Win {
    unsigned long long ptr;
    unsigned records;
    unsigned base;
};

template <int V>
__device__ __forceinline__ void p2_set_window(P2Win& win, const DevVolume& vol, unsigned first_slot, unsigned n_slots)
{
    // (records of 16 B: the vec4 voxels; of 4 B: the density plane)
    const char* b = (V == V_BASIC) ? vol.a_base + (size_t)first_slot * 4u : reinterpret_cast<const char*>(vol.data) + (size_t)first_slot * 16u;
    win.ptr = (unsigned long long)reinterpret_cast<size_t>(b);
    win.records = n_slots;
    win.base = first_slot;
}
template <int V>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p2_descriptor(const P2Win& win)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)win.ptr), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(win.ptr >> 32));
    void* p = reinterpret_cast<void*>((size_t)(((unsigned long long)hi << 32) | lo));
    return __builtin_amdgcn_make_buffer_rsrc(p, V == V_BASIC ? 4 : 16, __builtin_amdgcn_readfirstlane((int)win.records), 0x00020000);
}
template <int V>
constexpr unsigned p2_window_slots() { return V == V_BASIC ? 0x3fffffffu : 0x0fffffffu; }  // records x stride < 4 GiB

// Requests the eight corners of position q into X, returns the interpolation weights.  MASKED: only the lanes of `keep`
// request anything, and the distance-field byte of q's brick (VOLUME_MASK: and the brick's mask record) is asked for just
// ahead of the corners, by every lane -- the skipping's bricks ARE the layout's bricks (brick_of(q) is the base cell's brick:
// see its comment), so the byte's index is the base corner's slot without its six intra-brick bits.
// WIN: `need` = the lanes whose corners will be consumed (marching, not idle); when one of them leaves the window it is
// re-centred on them (no memory instruction in that block); a packet that does not fit a window at all (never seen: the
// rays of a packet are voxels apart) ends the pipelined loop -- n_in_w = 0 -- and the plain loop behind it finishes the packet.
// (in two parts, so that the tables' LDS latency runs under the interpolation of the step being consumed: p2_address -- the cell's
// coordinates, weights and the three table reads, anywhere before -- and p2_issue -- slots, window, byte, loads -- once the old
// corners are dead)
struct P2Addr {
    uint2 ex, ey, ez;   // slot terms of the texel pairs per axis
    float fx, fy, fz;   // interpolation weights
    int tz;             // base cell's z (the window's test)
};
__device__ __forceinline__ P2Addr p2_address(const DevVolume& vol, const P2Lds& L, f3 q)
{
    P2Addr a;
    const float x = mad(q.x, (float)vol.nx, -0.5f), y = mad(q.y, (float)vol.ny, -0.5f), z = mad(q.z, (float)vol.nz, -0.5f);
    const float x0 = floorf(x), y0 = floorf(y), z0 = floorf(z);
    a.fx = x - x0;
    a.fy = y - y0;
    a.fz = z - z0;
    // (saturating conversions, NaN -> 0; the clamp to [-1, n-1] selects the same texel pair as texel_pair() for ANY value)
    const int tx = min(max((int)x0, -1), L.mx), ty = min(max((int)y0, -1), L.my);
    a.tz = min(max((int)z0, -1), L.mz);
    a.ex = lds_pair(L.off_x + ((unsigned)(tx + 1) << 3));
    a.ey = lds_pair(L.off_y + ((unsigned)(ty + 1) << 3));
    a.ez = lds_pair(L.off_z + ((unsigned)(a.tz + 1) << 3));
    return a;
}
template <int V, bool MASKED, bool WIN, typename T>
__device__ __forceinline__ void p2_issue(const MarchParams& P0, const DevVolume& vol, const P2Lds& L, P2Win& win, int& n_in_w, const P2Addr& a, T (&X)[8],
                                         float& fx, float& fy, float& fz, unsigned long long keep, bool need, unsigned& dbyte, float& mrec)
{
    fx = a.fx;
    fy = a.fy;
    fz = a.fz;
    const uint2 ex = a.ex, ey = a.ey, ez = a.ez;
    const int tz = a.tz;
    unsigned az0 = ez.x, az1 = ez.y;
    if constexpr (WIN) {
        az0 -= win.base;
        az1 -= win.base;
        unsigned chk = need ? az0 : 0u;  // (a slab in front of the window wraps to a huge value)
        asm volatile("" : "+v"(chk));   // (kept as a value: one compare for the vote)
        const unsigned long long out = vr_ballot(chk > L.win_last);
        if (out != 0) {
            // centre the window on the first ray that left it (a packet's rays are a slab or two apart at any step); a packet
            // whose requesting rays do not fit it then -- never seen outside the tests' three-slab windows -- leaves the pipelined
            // loop after this trip (n_in_w = 0) and the plain loop behind it takes the rest
            const int kb = __builtin_amdgcn_readlane(max(tz, 0) >> kVbS, (int)__builtin_ctzll(out));  // that ray's z-slab of bricks
            const int span = (int)L.win_slabs - 2;  // the base corner's slab may be the window's first .. last but one
            const unsigned first = (unsigned)max(0, kb - span / 2) * vol.brick_slab, total = vol.brick_slab * (((unsigned)vol.nz + kVbM) >> kVbS);
            p2_set_window<V>(win, vol, first, min(total - first, L.win_slots));
            az0 = ez.x - win.base;
            az1 = ez.y - win.base;
            unsigned chk2 = need ? az0 : 0u;
            asm volatile("" : "+v"(chk2));
            if (vr_ballot(chk2 > L.win_last) != 0) n_in_w = 0;
        }
    }
    const unsigned r00 = ey.x + az0, r10 = ey.y + az0, r01 = ey.x + az1, r11 = ey.y + az1;
    unsigned o[8] = {r00 + ex.x, r00 + ex.y, r10 + ex.x, r10 + ex.y, r01 + ex.x, r01 + ex.y, r11 + ex.x, r11 + ex.y};
    if constexpr (MASKED) {
        static_assert(kBrickShift == (int)kVbS, "the skipping's bricks are the layout's bricks");
        const unsigned bid = (WIN ? o[0] + win.base : o[0]) >> (3u * kVbS);
        dbyte = dist_at(P0, (int)bid);
        if constexpr (V == V_VOLUME_MASK) mrec = brick_record(P0, (int)bid).y;
    }
    // The idle lanes are switched off for the eight loads by hand: the compiler does not see a branch (so it keeps no execz jump
    // and the wait counts stay exact), the texture addresser does not see the lanes.  Nothing but the loads runs in between:
    // the slots are pinned into registers first, and the scheduler is fenced on both sides.
    const __amdgpu_buffer_rsrc_t rsrc = p2_descriptor<V>(win);
    unsigned long long exec_saved = 0;
    if constexpr (MASKED) {
        asm volatile("" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1" : "=&s"(exec_saved) : "s"(keep) : "scc");
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if constexpr (V == V_BASIC) X[k] = __builtin_bit_cast(float, vr_struct_load_b32(rsrc, (int)o[k], 0, 0, 0));
        else X[k] = __builtin_bit_cast(vr_f4, vr_struct_load_b128(rsrc, (int)o[k], 0, 0, 0));
    }
    if constexpr (MASKED) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_mov_b64 exec, %0" : : "s"(exec_saved));
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int V, bool MASKED, bool WIN, typename T>
__device__ __forceinline__ void p2_request(const MarchParams& P0, const DevVolume& vol, const P2Lds& L, P2Win& win, int& n_in_w, f3 q, T (&X)[8], float& fx,
                                           float& fy, float& fz, unsigned long long keep, bool need, unsigned& dbyte, float& mrec)
{
    const P2Addr a = p2_address(vol, L, q);
    p2_issue<V, MASKED, WIN>(P0, vol, L, win, n_in_w, a, X, fx, fy, fz, keep, need, dbyte, mrec);
}

// The smallest value of a full wavefront (every lane active): four v_min_i32 through the DPP cross-bar (neighbour, pair, half row,
// row: every lane of a row of 16 then holds the row's minimum), the four rows' values read as scalars.  No LDS round trip
// (__shfl_xor goes through ds_bpermute: six dependent LDS latencies), nine instructions.
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm:[1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm:[2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// The per-step vote of sample_and_blend (opacity_is_zero for every sampling ray) as ONE compare: the table index that decides,
// with "not finite", "does not sample" and (VOLUME_MASK) "masked: the dose's table decides" folded into its value.
__device__ __forceinline__ bool p2_vote(const MarchParams& P0, bool sampled, float d, bool masked = false)
{
    int j = padded_texel(floorf(mad(d, (float)P0.tf[0].res_o, -0.5f)), P0.tf[0].res_o);
    j = (d - d == 0.0f) ? j : 0x7fffffff;          // an infinite density has a NaN weight, hence a NaN opacity: never "zero"
    j = masked ? 0x7fffffff : j;
    j = sampled ? j : (int)0x80000000;             // a ray that does not sample never asks for the shading
    asm volatile("" : "+v"(j));                    // (kept as a value: the compiler would turn the compare back into mask logic)
    return vr_ballot(j > P0.zskip_prefix) != 0;
}

// SKIP: empty-space skipping on top of it.  One distance-field byte per ray rides along with each corner buffer: the byte of
// the exact position whose corners are in flight, asked for just ahead of them and read a trip later.  It says whether the
// step blends (an inert brick: the identity, march_packet's test), whether the two steps after it need their corners at all
// (the ray's safe steps in inert bricks reach them: its lanes are switched off for those loads), and how many steps after it
// every ray of the packet can skip: then the REQUESTS jump (4 .. 64 rounded additions, the identity steps of march_packet's
// runs) while the two steps already in flight are still being consumed -- nothing in flight is thrown away and no latency is
// exposed.  A step in which no ray blends interpolates nothing; the per-step vote (every opacity zero for certain: no texels,
// no gradient, no shading) is march_packet's.
template <int V, bool SKIP, bool WIN, bool BATCH>
__global__ __launch_bounds__(V == V_BASIC ? kP2ThreadsUnlit : kP2Threads) void march_p2_kernel(const MarchBatch B, const PwQueue Q)
{
    static_assert(V == V_LIGHT || V == V_BASIC || (V == V_VOLUME_MASK && SKIP), "lit / unlit shader; the three-volume composite with its brick records");
    constexpr int kSrc = (V == V_VOLUME_MASK) ? 2 : 0;  // the volume that is pipelined: the CT of the composite (VolumeMaskApp.wgsl:187)
    constexpr bool kLit = V != V_BASIC;                 // 16-byte voxels (gradient + density) / 4-byte densities
    // (the one-frame >= 4 GiB kernel is at the register limit: with the approach loop in front its pipelined loop reloads two register
    // pairs from scratch per step -- C5 3.39 -> 3.45 ms, with the exit test 3.59, a rank's half of C5 1.82 -> 2.00; its several-frames
    // form and every other form gain: tools/experiments/r4x.sh, r4z.sh)
    // BATCH names the launch (several frames / one); kMulti is the code: the several-frames form -- a frame's parameters read where they
    // are used, through a wave-uniform index -- also for one frame per launch (vr_launch.h: VR_P2_ALL_BATCH / VR_P2_WIN_BATCH; the
    // one-frame launches keep a kernel name of their own in the profiles)
    constexpr bool kMulti = BATCH || (WIN ? VR_P2_WIN_BATCH != 0 : VR_P2_ALL_BATCH != 0);
    constexpr bool kApproachHere = SKIP && kP2Approach && !(WIN && !kMulti);
    constexpr bool kExit = kApproachHere && kP2Exit;
    const MarchParams& P0 = B.frame[0];                 // what every frame of the launch shares: volumes, tables, brick records
    const DevVolume& vol = P0.vol[kSrc];
    P2Lds L;
    {
        const unsigned tf_bytes = (unsigned)(P0.tf[0].res_o + 2) * 16u;  // (res_c == res_o: the host's condition)
        L.off_x = tf_bytes;
        L.off_y = L.off_x + (unsigned)(vol.nx + 1) * 8u;
        L.off_z = L.off_y + (unsigned)(vol.ny + 1) * 8u;
        L.mx = vol.nx - 1;
        L.my = vol.ny - 1;
        L.mz = vol.nz - 1;
        L.win_slots = WIN ? (Q.p2_window ? Q.p2_window : p2_window_slots<V>()) : 0u;
        L.win_slabs = WIN ? L.win_slots / vol.brick_slab : 0u;
        L.win_last = WIN ? (L.win_slabs - 2u) * vol.brick_slab + (kVbM << (2u * kVbS)) : 0u;
        // (the table's loads of six entries per thread in flight together: the launch waits for one memory latency here, not for
        // one per 768 entries -- this copy is in front of every packet of the launch)
        const int n = P0.tf[0].res_o + 2;
        constexpr int kU = 6;
        for (int base = (int)threadIdx.x; base < n; base += kU * (int)blockDim.x) {
            float4 c[kU];
            float o[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int j = min(base + u * (int)blockDim.x, n - 1);
                c[u] = P0.tf[0].color[j];
                o[u] = P0.tf[0].opacity[j];
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int j = base + u * (int)blockDim.x;
                if (j < n) vr_lds_tf[j] = make_float4(c[u].x, c[u].y, c[u].z, o[u]);
            }
        }
        p2_fill_tables(vol, L);
        __syncthreads();  // the only barrier: from here on the wavefronts are independent of each other
    }
    const unsigned total_slots = vol.brick_slab * (((unsigned)vol.nz + kVbM) >> kVbS);
    const unsigned cls = blockIdx.x & 7u;
    const unsigned groups = (gridDim.x - cls + 7u) >> 3;
    const unsigned wib = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wpb = blockDim.x >> 6;
    const unsigned nf = kMulti ? B.n_frames : 1u;
    const unsigned n_c = (Q.n_items >> 3) * nf;  // items per class: every frame's packets of the class, the frames interleaved
    // first item: static (wavefront k of every workgroup before wavefront k + 1 of any: the longest packets are dealt over the CUs, no
    // atomic at the start of the launch) -- or, Q.dynamic (launches in flight: workgroups start when others retire, and the
    // first to start should take the longest packets left), from the class's head like every later one
    unsigned idx = wib * groups + (blockIdx.x >> 3);
    if (Q.dynamic) {
        unsigned r = 0;
        if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cls * kPwHeadStride, 1u);
        idx = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    }
    typedef typename std::conditional<kLit, vr_f4, float>::type Elem;
    const unsigned cur = cls;
    for (;;) {
        if (idx >= n_c) break;  // (the class's queue has run dry: the wavefront leaves)
        // item idx of the class = packet idx / nf (in the launch order) of frame idx % nf
        unsigned item = idx, frame = 0;
        if constexpr (kMulti) {
            item = batch_group(idx, nf);
            frame = idx - item * nf;
        }
        // (the frame's parameters through a wave-uniform index, said so explicitly: scalar loads; behind an index the compiler takes
        // for divergent they become vector loads -- in the middle of the pipelined loop, whose wait counts they then drain)
        const MarchParams& P = B.frame[kMulti ? (unsigned)__builtin_amdgcn_readfirstlane((int)frame) : 0u];
        // what the step loop reads of the frame, once per packet
        const f3 light_pos = mk3(P.light_pos[0], P.light_pos[1], P.light_pos[2]), light_dif = mk3(P.light_dif[0], P.light_dif[1], P.light_dif[2]),
                 light_amb = mk3(P.light_amb[0], P.light_amb[1], P.light_amb[2]);
        const int steps_count = P.steps_count;
        const unsigned pos = (item << 3) | cur;
        int lb = (int)pos;
        if (P0.order != nullptr) lb = __builtin_amdgcn_readfirstlane((int)P0.order[pos]);
        const unsigned long long t_start = wall_clock64();
        const PixelSlot slot = map_pixel_at(P0, lb, 1, 0);  // (rank, tiles, viewport: the launch's, the same for every frame of it)
        const float bx0 = P.bmin[0], by0 = P.bmin[1], bz0 = P.bmin[2];
        const float bx1 = P.bmax[0], by1 = P.bmax[1], bz1 = P.bmax[2];
        float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        unsigned blends = 0, covered = 0, fetched = 0;
#if VR_P2_DEBUG
        unsigned dbg_trips = 0, dbg_sampled = 0, dbg_shaded = 0, dbg_jumps = 0;
        unsigned long long dbg_wait_corners = 0, dbg_wait_bytes = 0, dbg_loop = 0;  // (VR_P2_DEBUG=2: shader-clock cycles)
        unsigned long long dbg_tail = 0;
        const unsigned long long dbg_pkt0 = __builtin_readcyclecounter();
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
                    if (P.toggle_varstep == 1) step_size = ray_len / (float)steps_count;
                    p = ray.start;
                    if (P.toggle_jitter == 1) {
                        float jt = jitter((float)slot.px + 0.5f, (float)slot.py + 0.5f);
                        p = mk3(p.x + (dir.x * step_size) * jt, p.y + (dir.y * step_size) * jt, p.z + (dir.z * step_size) * jt);
                    }
                    step = mk3(dir.x * step_size, dir.y * step_size, dir.z * step_size);
                    if constexpr (V == V_VOLUME_MASK) wstep = step;  // the "world" position advances by the uvw step (VolumeMaskApp.wgsl:213)
                    w = ray.world0;
                    n_inside = steps_inside(p, step, bx0, by0, bz0, bx1, by1, bz1);
                    alive = steps_count > 0;
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
                if constexpr (kLit) {
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
            int n_in_w = wave_min_i32(alive ? min(n_inside, steps_count) : 0x7fffffff);
            // the gather source: the bricked vec4 voxels (lit) or the bricked density plane (unlit) as records of an indexed buffer
            P2Win win;
            p2_set_window<V>(win, vol, 0u, WIN ? min(total_slots, L.win_slots) : total_slots);
            unsigned DA = 0, DB = 0;  // the distance-field bytes of the bricks of pA and pB (0 = active, n = n bricks from an active one)
            float MA = 0.0f, MB = 0.0f;  // VOLUME_MASK: the mask records (largest mask channel) of those bricks
            float leap_c = 0.0f;
            int lim = 0;
            if constexpr (SKIP) {
                // steps a ray at distance-field value D can take while it certainly stays within D-1 bricks of its brick on every
                // axis (march_packet's kRun); jumps stay inside the provably-in-box prefix of the ray
                const float vmax = fmaxf(fmaxf(fabsf(step.x) * P0.bsx, fabsf(step.y) * P0.bsy), fabsf(step.z) * P0.bsz);
                leap_c = 0.999f / vmax;
                lim = min(n_inside, steps_count);
            }
            bool idle_a = false, idle_b = false;      // this trip: step A / B blends nothing
            bool idle_ra = false, idle_rb = false;    // ... requests nothing
            int mw = 0;                               // identity steps skipped between this trip's two steps and the next trip's
            // consumes the corners in X (of position pX, step ix), then requests into X the corners of pY + step [+ mw steps], which
            // becomes pX
            auto one_step = [&](Elem (&X)[8], float& xfx, float& xfy, float& xfz, f3& pX, const f3& pY, int jump, bool idle_con, bool idle_rq, unsigned& DX,
                                float& MX) {
                const bool inb = alive;  // (in time and inside the box: the loop's condition)
                const bool sampled = SKIP ? (inb && !idle_con) : inb;
                // the position of the next request into X, and its cell's table reads: in flight under this step's interpolation
                f3 pN = mk3(pY.x + step.x, pY.y + step.y, pY.z + step.z);
                for (int k = 0; k < jump; ++k) pN = mk3(pN.x + step.x, pN.y + step.y, pN.z + step.z);
                const P2Addr ad = p2_address(vol, L, pN);
                v2f zw = v2f{0.0f, 0.0f}, gxy = zw;
                TfFetch tq;
                float4 mask = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // VOLUME_MASK: the interpolated mask and dose of pX
                float rt = 0.0f;
                bool any_masked = false, masked = false;
                // (wave-uniform) a step in which no ray samples interpolates nothing
                bool shaded = !SKIP || vr_ballot(!idle_con) != 0;  // (idle_con covers the rays that had finished when the trip began)
#if VR_P2_DEBUG
                if (shaded) ++dbg_sampled;
#endif
                if (shaded) {
                    // ONE wait for the buffer: its corners arrived long ago (a -DVR_P2_DEBUG=2 build clocks this wait: 0.2 - 0.6 % of a
                    // packet's cycles, gpurun_out/r4k), so the four partial waits the compiler places (one per pair of corners as the
                    // interpolation reaches it) buy nothing and cost three scalar instructions
#if VR_P2_DEBUG >= 2
                    const unsigned long long c0 = __builtin_readcyclecounter();
#endif
                    if constexpr (kLit) asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]));
                    else asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]));
#if VR_P2_DEBUG >= 2
                    dbg_wait_corners += __builtin_readcyclecounter() - c0;
#endif
                }
                if (!shaded) {
                } else if constexpr (kLit) {
                    Fetch4 q;
                    q.a = make_float4(X[0].x, X[0].y, X[0].z, X[0].w); q.b = make_float4(X[1].x, X[1].y, X[1].z, X[1].w);
                    q.d = make_float4(X[2].x, X[2].y, X[2].z, X[2].w); q.e = make_float4(X[3].x, X[3].y, X[3].z, X[3].w);
                    q.f = make_float4(X[4].x, X[4].y, X[4].z, X[4].w); q.g = make_float4(X[5].x, X[5].y, X[5].z, X[5].w);
                    q.h = make_float4(X[6].x, X[6].y, X[6].z, X[6].w); q.i = make_float4(X[7].x, X[7].y, X[7].z, X[7].w);
                    zw = interp_zw(q, xfx, xfy, xfz);
                    if constexpr (V == V_VOLUME_MASK) {
                        // Mask and dose on demand (fetch_mask_and_dose's rule: a brick whose mask record is <= 0 interpolates to
                        // <= 0, which fails the shader's comparison like a mask of 0).  The record came with the corners; the
                        // fetch itself is a block of its own whose loads are waited for INSIDE it (the values are pinned there),
                        // so that the two corner buffers' wait counts stay exact behind it.  The corners in X are still needed
                        // (the gradient): they are interpolated first.
                        gxy = interp_xy(q, xfx, xfy, xfz);
                        asm volatile("" : "+v"(zw.x), "+v"(zw.y), "+v"(gxy.x), "+v"(gxy.y));
                        any_masked = vr_ballot(sampled && !(MX <= 0.0f)) != 0;
                        if (any_masked) {
                            mask = tex3_rgba<!WIN>(P0.vol[0], pX);
                            rt = tex3_a<!WIN>(P0.vol[1], pX);
                            asm volatile("" : "+v"(mask.x), "+v"(mask.y), "+v"(mask.z), "+v"(rt));
                            masked = mask.x > 0.0f || mask.y > 0.0f || mask.z > 0.0f;
                        }
                        shaded = p2_vote(P0, sampled, zw.y, masked);
                        if (shaded) tq = tf_fetch_lds(P0.tf[0], zw.y);
                    } else {
                        // (the per-step vote of sample_and_blend: when every ray's opacity is zero for certain, the texels, the
                        // gradient and the shading are left out -- the blend would be the identity)
                        if constexpr (SKIP) shaded = p2_vote(P0, sampled, zw.y);
                        if (shaded) {
                            tq = tf_fetch_lds(P0.tf[0], zw.y);
                            gxy = interp_xy(q, xfx, xfy, xfz);
                        }
                    }
                } else {
                    Fetch1 q;
                    q.a = X[0]; q.b = X[1]; q.d = X[2]; q.e = X[3]; q.f = X[4]; q.g = X[5]; q.h = X[6]; q.i = X[7];
                    zw.y = interp_a(q, xfx, xfy, xfz);
                    if constexpr (SKIP) shaded = p2_vote(P0, sampled, zw.y);
                    if (shaded) tq = tf_fetch_lds(P0.tf[0], zw.y);
                }
                pX = pN;  // (VOLUME_MASK has fetched its mask at the consumed position above)
                // Everything that reads the old corners must be COMPUTED here, before their registers are loaded again: left alone,
                // the compiler sinks the gradient's interpolation into the `if (sampled)` below (its only user), the old corners
                // then live across the new loads, the new loads get other registers, and the copies that bring them back at the
                // loop's back edge need the data (s_waitcnt vmcnt(0) every trip).
                if constexpr (kLit) asm volatile("" : "+v"(zw.x), "+v"(zw.y), "+v"(gxy.x), "+v"(gxy.y));
                else asm volatile("" : "+v"(zw.y));
                __builtin_amdgcn_sched_barrier(0);  // the old corners are dead here: the new ones may land in their registers
                // (with the byte of the position requested: the next trip decides with it)
                // (the lanes that request: marching and not idle -- as the lane mask itself, taken before the slots are pinned)
                const bool rq = alive && !(SKIP && idle_rq);
                p2_issue<V, SKIP, WIN>(P0, vol, L, win, n_in_w, ad, X, xfx, xfy, xfz, vr_ballot(rq), rq, DX, MX);
                __builtin_amdgcn_sched_barrier(0);
#if VR_P2_DEBUG
                if (shaded) ++dbg_shaded;
#endif
                if (sampled) {
                    if (shaded) {
                        if constexpr (V == V_LIGHT) {
                            shade_blend_packed<true>(light_pos, light_dif, light_amb, 2.5f, 0.5f, w, zw, gxy, tq, dst);
                        } else if constexpr (V == V_VOLUME_MASK) {
                            // src_volume_mask + blend with the CT texels from LDS (VolumeMaskApp.wgsl:185-213)
                            TfSample trt;
                            trt.rgb = mk3(0.0f, 0.0f, 0.0f);
                            trt.opacity = 0.0f;
                            if (any_masked) {
                                TfFetch rq = tf_fetch(P0.tf[1], rt);
                                tf_pin(rq);  // (waited for here, inside the block)
                                trt = tf_finish(rq);
                            }
                            // (the packed form of normalize3 / shade / blend: per component the same operations in the same order)
                            shade_blend_packed<true, true>(mk3(0.0f, -5.0f, 0.0f), mk3(0.96f, 0.76f, 0.67f), mk3(1.0f, 1.0f, 1.0f), 1.5f, 0.5f, w, zw, gxy, tq,
                                                           dst, masked, trt.rgb, trt.opacity);
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
                if constexpr (kLit) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
            };
            int k_last = 0x7fffffff;  // no step of the ray behind this one can lie in an active brick (steps_near_box)
            if constexpr (kApproachHere) {
                // THE APPROACH: until a ray of the packet stands in an active brick nothing is requested ahead -- a byte per ray, the
                // identity steps it allows (march_packet's run length, and one for the position itself), the wave's minimum of them
                // as plain rounded additions.  A packet that never meets an active brick (C3: 7 775 of the 12 214 packets whose rays
                // cross the box, 22 % of the frame's wavefront time when they went through the pipelined loop's trips) ends here.
                // Rays outside the box of the active bricks (abox; steps_near_box) ask for nothing at all: before it they are safe up
                // to it, behind it to the end.  The world position follows when a ray may still sample (it is only read by the shading).
                int k0, k1;
                steps_near_box(pA, step, P0.abox, steps_count, k0, k1);
                if constexpr (kExit) k_last = k1;
                int skipped = 0;  // (wave-uniform) steps taken here
                while (i < n_in_w && vr_ballot(alive) != 0) {
                    const bool near = alive && i >= k0 && i <= k1;
                    int safe = kP2JumpMax;  // identity steps from step i on
                    if (vr_ballot(near) != 0) {
                        const unsigned d = dist_at(P0, brick_of<true>(P0, pA));
                        if (vr_ballot(near && d == 0u) != 0) break;
                        const int sf = 1 + max((int)fminf(((float)d - (1.0f + kBrickHalf)) * leap_c, (float)kP2JumpMax), 0);
                        safe = near ? sf : safe;
                    }
                    if (alive && i < k0) safe = min(safe, k0 - i);
                    safe = alive ? min(safe, lim - i) : kP2JumpMax;
                    const int sw = wave_min_i32(safe);  // (>= 1: every marching ray has lim > i, an inert brick, or steps to go to its k0)
                    for (int k = 0; k < sw; ++k) pA = mk3(pA.x + step.x, pA.y + step.y, pA.z + step.z);
                    if (alive) blends += (unsigned)sw;
                    i += sw;
                    skipped += sw;
                }
                if constexpr (kLit) {
                    if (skipped != 0 && vr_ballot(alive && i <= k1) != 0)
                        for (int k = 0; k < skipped; ++k) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                }
                p = pA;
                pB = pA;
            }
            bool start = true;  // (wave-uniform) nothing is in flight yet
#if VR_P2_DEBUG
            dbg_trips = dbg_sampled = dbg_shaded = dbg_jumps = 0;
            dbg_loop = __builtin_readcyclecounter();
#endif
            while (i + 2 <= n_in_w && vr_ballot(alive) != 0) {
                if constexpr (kExit) {
                    // (no marching ray can meet an active brick any more: what is left are identity steps -- below)
                    if (vr_ballot(alive && i <= k_last) == 0) break;
                }
                if (start) {
                    // the bytes of pA and pB and, without waiting for them, the corners of steps 0 and 1 of every ray
                    pB = mk3(pA.x + step.x, pA.y + step.y, pA.z + step.z);
                    p2_request<V, SKIP, WIN>(P0, vol, L, win, n_in_w, pA, A, afx, afy, afz, vr_ballot(alive), alive, DA, MA);
                    p2_request<V, SKIP, WIN>(P0, vol, L, win, n_in_w, pB, Bq, bfx, bfy, bfz, vr_ballot(alive), alive, DB, MB);
                    start = false;
                    if constexpr (WIN) {
                        if (n_in_w == 0) break;  // (the packet's first cells do not fit one window: the plain loop takes all of it)
                    }
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
#if VR_P2_DEBUG >= 2
                    {
                        const unsigned long long c0 = __builtin_readcyclecounter();
                        asm volatile("" : "+v"(DA), "+v"(DB));
                        dbg_wait_bytes += __builtin_readcyclecounter() - c0;
                    }
#endif
                    unsigned da = alive ? DA : 255u, db = alive ? DB : 255u;
                    asm volatile("" : "+v"(da), "+v"(db));  // (kept as values: the compiler would turn `da >= 1` back into mask logic)
                    idle_a = da >= 1u;
                    idle_b = db >= 1u;
                    // steps after pB the ray certainly spends in inert bricks (march_packet's run length; < 0 at an active brick)
                    int m = min((int)fminf(((float)db - (1.0f + kBrickHalf)) * leap_c, (float)kP2JumpMax), lim - (i + 1) - 1);
                    m = alive ? m : kP2JumpMax;
                    // (Round 3 jumped by 4, 8 .. 64 steps, decided by up to five votes: a gap of one to three steps -- the packet's rays
                    // reach an active brick a few steps apart -- was walked as idle steps, each the price of a request.  A jump costs
                    // three additions per step: every marching ray's safe steps, however few, are jumped, by exactly their minimum.)
                    if (vr_ballot(m < 1) == 0) mw = min(wave_min_i32(m), kP2JumpMax);
                    idle_ra = m >= mw + 1;  // the positions requested now are steps mw + 1 and mw + 2 after pB
                    idle_rb = m >= mw + 2;
                }
#if VR_P2_DEBUG
                ++dbg_trips;
                if (mw > 0) ++dbg_jumps;
#endif
                one_step(A, afx, afy, afz, pA, pB, mw, idle_a, idle_ra, DA, MA);
                one_step(Bq, bfx, bfy, bfz, pB, pA, 0, idle_b, idle_rb, DB, MB);
                if (mw > 0) {
                    if constexpr (kLit) {
                        for (int k = 0; k < mw; ++k) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
                    }
                    if (alive) blends += (unsigned)mw;
                }
                i += 2 + mw;
            }
#if VR_P2_DEBUG
            dbg_loop = __builtin_readcyclecounter() - dbg_loop;
            dbg_tail = __builtin_readcyclecounter();
#endif
            // The last steps of the packet (pA is the exact position of step i, w its world position): the shader's loop as it
            // stands -- box test, identity steps by the distance-field byte, the far-bound exit -- with no loads ahead.
            // (It also takes ALL the steps of a packet on the box's silhouette once its shortest ray may leave: 13 % of the sampling
            // packets' cycles on C3, 3 % of the longest quarter's (-DVR_P2_DEBUG=3, gpurun_out/r4l).  Going back into the pipelined
            // loop when the short rays have retired was built and measured: bit-exact, and SLOWER -- C3 0.525 -> 0.537 ms, C4 0.575
            // -> 0.600 -- the values that then live across both loops cost spills, in the >= 4 GiB kernels between the two exec
            // writes of p2_issue, where tools/check_exec_regions.py caught them.  Removed.
            // PARKING the rays that reach their own count (position, world position and step to LDS; the pipelined loop goes on
            // with the others, this loop takes every parked ray from its own step) was built too: bit-exact, no spill in the
            // pipelined loop, and slower -- C3 0.553 -> 0.587 ms, C4 0.592 -> 0.636 (tools/experiments/r4u.sh).  A step of this
            // loop costs about what a pipelined step costs (13 % of the cycles for about as many of the steps): keeping the
            // pipelined loop going with ever fewer rays saves nothing, and the parked rays' steps here come on top.)
            if (!start) p = pA;
            if constexpr (kExit) {
                // behind the box of the active bricks: the steps up to the packet's provably-in-box prefix are plain rounded additions
                // (the world position is not read again), and the loop below asks for no byte
                if (i < n_in_w && vr_ballot(alive) != 0 && vr_ballot(alive && i <= k_last) == 0) {
                    const int sw = n_in_w - i;
                    for (int k = 0; k < sw; ++k) p = mk3(p.x + step.x, p.y + step.y, p.z + step.z);
                    if (alive) blends += (unsigned)sw;
                    i += sw;
                }
            }
            for (; i < steps_count && vr_ballot(alive) != 0; ++i) {
                if (alive) {
                    bool inb = true;
                    if (i >= n_inside) inb = p.x >= bx0 && p.x <= bx1 && p.y >= by0 && p.y <= by1 && p.z >= bz0 && p.z <= bz1;
                    if (inb) {
                        bool sampled = true;
                        if constexpr (SKIP) sampled = i <= k_last && dist_at(P0, brick_of<true>(P0, p)) == 0u;
                        if (sampled) {
                            sample_and_blend<V, !WIN, false, SKIP, true>(P, p, w, dst, mk3(0.0f, 0.0f, 0.0f), 0.0f);
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
                if constexpr (kLit) w = mk3(w.x + wstep.x, w.y + wstep.y, w.z + wstep.z);
            }
        }
#if VR_P2_DEBUG
        if (dbg_tail) dbg_tail = __builtin_readcyclecounter() - dbg_tail;
#endif
        if (slot.active || (P0.packed && slot.in_launch)) P.out[slot.out_index] = dst;
        store_wave_counts(P, lb, blends, covered, fetched, t_start);
#if VR_P2_DEBUG
        if ((threadIdx.x & 63) == 0)  // (debug build: the `fetched` word carries the loop's own counters instead)
            P.block_counts[(size_t)lb * kBlockRecord + 2] = (unsigned long long)(dbg_trips & 0xfffu) | ((unsigned long long)(dbg_sampled & 0xfffu) << 12) |
                                                           ((unsigned long long)(dbg_shaded & 0xfffu) << 24) | ((unsigned long long)(dbg_jumps & 0xfffu) << 36);
#if VR_P2_DEBUG >= 2
        if ((threadIdx.x & 63) == 0)  // (... and the `covered` word the cycles spent waiting for corners / bytes and in the pipelined loop, / 64)
#if VR_P2_DEBUG >= 3   // (3: the packet's whole time and its last steps' instead of the two waits)
            P.block_counts[(size_t)lb * kBlockRecord + 1] = ((dbg_tail >> 6) & 0xfffffull) | ((((__builtin_readcyclecounter() - dbg_pkt0) >> 6) & 0xfffffull) << 20) |
                                                           (((dbg_loop >> 6) & 0xffffffull) << 40);
#else
            P.block_counts[(size_t)lb * kBlockRecord + 1] = ((dbg_wait_corners >> 6) & 0xfffffull) | (((dbg_wait_bytes >> 6) & 0xfffffull) << 20) |
                                                           (((dbg_loop >> 6) & 0xffffffull) << 40);
#endif
#endif
#endif
        unsigned r = 0;
        if ((threadIdx.x & 63u) == 0) r = atomicAdd(Q.heads + cur * kPwHeadStride, 1u);
        idx = (Q.dynamic ? 0u : groups * wpb) + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    }
}

}  // namespace VR_KNS
