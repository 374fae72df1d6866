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
    // (A wavefront whose class has run dry leaves.  Round 3 let it go on with the next class's queue -- 2-4 % slower, the classes
    // are even when a tile's sub-blocks are dealt over them -- and whole tiles per class with stealing -- less fabric traffic, 9 %
    // slower: both knobs were removed in round 4, their measurements are in HISTORY.md.)
    const unsigned cur = cls;
    for (;;) {
        if (idx >= n_c) break;
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
        idx = groups * wpb + (unsigned)__builtin_amdgcn_readfirstlane((int)r);
    }
}

// (two steps ahead -- flavours 16 / 17, march_p2_kernel: vr_p2.h)

}  // namespace VR_KNS
