// vr_mixed.h -- lanes per ray chosen PER PACKET inside one launch.
//
// One frame at a time ends with its longest chains of dependent samples: the 8x8 packets whose rays sample ~360 times take
// the whole frame time (1.6 us per step while the machine is full) and the machine drains behind them -- 4 439 of the 32 640
// packets of the C3 frame fetch anything at all, about one per wavefront slot, so there is no queue to balance, only chains to
// shorten.  The depth-parallel kernel (vr_dp.h) halves a chain by giving a ray two lanes, at 1.3x the wavefront time per sample:
// for every packet that loses (0.72 ms a frame), for the long ones alone it wins.  Here a launch takes its packets from an ITEM
// list built behind an earlier launch of the same shape (build_items_kernel, on the sort's stream): a packet whose longest chain
// was at least `threshold` samples appears as two items -- its upper and lower 8x4 pixels, each marched by one wavefront with
// two lanes per ray (march_dp_body<K = 2>) -- every other packet as one item marched with one lane per ray (march_packet), in the
// longest-first order of DESIGN 4.6.  Results are bit-identical either way (both loops are exact restatements of the shader's
// loop); the choice is a heuristic of the previous frames and can be stale without harm.
//
// Reference loop: BasicVolLightApp.wgsl:207-234 and siblings; caller: Application::OnRender (App/src/Application.cpp:121-239).
#pragma once
#include "vr_kernels.h"
#include "vr_dp.h"

namespace VR_KNS {
using namespace vr;

// item = logical block | kind << 30
constexpr unsigned kItemOne = 0u;    // the whole 8x8 packet, one lane per ray
constexpr unsigned kItemPad = 1u;    // nothing (the classes' lists are padded to one length)
constexpr unsigned kItemHalf0 = 2u;  // rows 0-3 of the packet, two lanes per ray
constexpr unsigned kItemHalf1 = 3u;  // rows 4-7

// the pixel of lane's ray when the wavefront marches half `half` (0 / 1) of packet lb with two lanes per ray
__device__ __forceinline__ PixelSlot map_pixel_half(const MarchParams& P, int lb, int half)
{
    PixelSlot s = map_pixel_at(P, lb, 1, 0);  // lane 0's pixel = the packet's corner (lane & 7 = 0, lane >> 3 = 0)
    const int lane = threadIdx.x & 63;
    const int ray = lane >> 1;
    // map_pixel_at placed this lane at (lane & 7, lane >> 3) of the packet: move it to its ray's pixel
    const int dx = (ray & 7) - (lane & 7), dy = ((ray >> 3) + 4 * half) - (lane >> 3);
    s.px += dx;
    s.py += dy;
    s.active = s.in_launch && (s.px < P.W) && (s.py < P.H);
    if (P.only_tile >= 0) s.active = false;  // (experiment knob of the one-lane kernels: not supported here)
    s.out_index += P.packed ? (dy * kTile + dx) : (dy * P.W + dx);
    return s;
}

template <int V, bool OFF32, bool SKIP>
__global__ __launch_bounds__(256) void march_mixed_kernel(const MarchBatch B, const unsigned* __restrict__ items, int n_logical)
{
    const MarchParams& P = B.frame[0];
    const unsigned item = (unsigned)__builtin_amdgcn_readfirstlane((int)items[blockIdx.x]);
    const unsigned kind = item >> 30;
    const int lb = (int)(item & 0x3fffffffu);
    if (kind == kItemPad) return;
    const unsigned long long t_start = wall_clock64();
    float4 dst = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    unsigned blends = 0, covered = 0, fetched = 0;
    if (kind == kItemOne) {
        const PixelSlot slot = map_pixel_at(P, lb, 1, 0);
        march_packet<V, OFF32, SKIP, SKIP ? 3 : 0, false, false>(P, slot, dst, blends, covered, fetched);
        if (slot.active || (P.packed && slot.in_launch)) P.out[slot.out_index] = dst;
        store_wave_counts(P, lb, blends, covered, fetched, t_start, false);
    } else {
        const int half = (int)(kind & 1u);
        const PixelSlot slot = map_pixel_half(P, lb, half);
        march_dp_body<V, OFF32, SKIP, 2, false>(P, slot, dst, blends, covered, fetched);
        const bool first = (threadIdx.x & 1u) == 0;  // depth slot 0 holds the ray's result and counts
        if (first && (slot.active || (P.packed && slot.in_launch))) P.out[slot.out_index] = dst;
        // the first half's record carries the flag; the second half's lies n_logical records further on
        store_wave_counts(P, half ? lb + n_logical : lb, first ? blends : 0u, first ? covered : 0u, first ? fetched : 0u, t_start,
                          half == 0);
    }
}

#if !VR_FUSED
// One workgroup, behind order_blocks_kernel on the sort's stream: the item list of a later launch from this launch's records
// and their longest-first order.  Per class c of the block index modulo 8 (a class stays on the XCD its index maps to, as in
// order_blocks_kernel) the list is the class's blocks in `order`, a block whose longest chain is >= threshold as two items;
// item i of class c sits at items[8 i + c]; the lists are padded to the longest class with kItemPad.  threshold = pct % of
// the launch's longest chain, at least min_chain samples (64 by default: below that a chain is too short to matter).  *n_positions (pinned host
// memory) = 8 x the padded length, i.e. the grid of the launch that uses the list; written last.
__global__ __launch_bounds__(1024) void build_items_kernel(const unsigned long long* __restrict__ rec, int n_blocks,
                                                           const unsigned* __restrict__ order, unsigned pct, unsigned min_chain,
                                                           unsigned* __restrict__ items, unsigned* __restrict__ n_positions,
                                                           unsigned* __restrict__ n_split_out)
{
    __shared__ unsigned s_max, s_len[8], s_wave[16];
    const int t = threadIdx.x, cls = t >> 7, l = t & 127;
    const int n_c = n_blocks >> 3, per = (n_c + 127) / 128;
    const int lo = min(l * per, n_c), hi = min(lo + per, n_c);
    auto chain_of = [&](int b) -> unsigned {
        const unsigned long long w5 = rec[(size_t)b * kBlockRecord + 5];
        unsigned c = (unsigned)(w5 >> 40);
        if (w5 & kRecSplit) {
            const unsigned c2 = (unsigned)(rec[(size_t)(b + n_blocks) * kBlockRecord + 5] >> 40);
            c = c2 > c ? c2 : c;
        }
        return c;
    };
    if (t == 0) s_max = 0;
    __syncthreads();
    unsigned mx = 0;
    for (int i = lo; i < hi; ++i) mx = max(mx, chain_of((int)order[8 * i + cls]));
    atomicMax(&s_max, mx);
    __syncthreads();
    const unsigned thr = max(min_chain, (s_max * pct + 99u) / 100u);
    unsigned cnt = 0;
    for (int i = lo; i < hi; ++i) cnt += chain_of((int)order[8 * i + cls]) >= thr ? 2u : 1u;
    // exclusive scan of cnt over the 128 threads of the class (two wavefronts)
    unsigned incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, off, 64);
        if ((t & 63) >= off) incl += up;
    }
    if ((t & 63) == 63) s_wave[t >> 6] = incl;
    __syncthreads();
    const unsigned before = ((t >> 6) & 1) ? s_wave[(t >> 6) - 1] : 0u;
    unsigned pos = before + incl - cnt;
    if (l == 127) s_len[cls] = before + incl;
    for (int i = lo; i < hi; ++i) {
        const unsigned b = order[8 * i + cls];
        if (chain_of((int)b) >= thr) {
            items[8u * pos + (unsigned)cls] = b | (kItemHalf0 << 30);
            items[8u * (pos + 1u) + (unsigned)cls] = b | (kItemHalf1 << 30);
            pos += 2u;
        } else {
            items[8u * pos + (unsigned)cls] = b | (kItemOne << 30);
            pos += 1u;
        }
    }
    __syncthreads();
    unsigned longest = 0, total = 0;
    for (int c = 0; c < 8; ++c) {
        longest = max(longest, s_len[c]);
        total += s_len[c];
    }
    for (unsigned p = s_len[cls] + (unsigned)l; p < longest; p += 128u) items[8u * p + (unsigned)cls] = kItemPad << 30;
    __syncthreads();
    if (t == 0) {
        if (n_split_out) *n_split_out = total - (unsigned)n_blocks;
        __threadfence_system();
        *n_positions = 8u * longest;
    }
}
#endif

}  // namespace VR_KNS
