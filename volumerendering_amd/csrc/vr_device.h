// vr_device.h -- device-side data structures shared by the kernels and the C-ABI implementation.
// gfx950 (MI355X) only.  All arithmetic IEEE f32, compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vr {

constexpr int kTile = 64;        // multi-GPU ownership granule (pixels)
constexpr int kBlockEdge = 16;   // one 256-thread workgroup = 16x16 pixels = four 8x8 wave packets
constexpr int kBlockRecord = 6;  // u64 words per block in MarchParams::block_counts
constexpr int kBlocksPerTile = (kTile / kBlockEdge) * (kTile / kBlockEdge);
constexpr unsigned long long kRecSplit = 1ull << 39;  // record word 5: the packet was marched as two half packets (vr_mixed.h)
// Empty-space bricks of 4 x 4 x 4 base cells (a brick's cells touch 5 x 5 x 5 voxels).  Round 1 used 8-cell bricks; 4-cell
// ones leave 6.5 % fewer samples of C3 (39 % of C2) inside active bricks for a distance field 8 times the size (2 MB for
// 512^3, 16 MB for 1024^3: one byte per brick) and are faster on every configuration (C3 0.573 -> 0.562 ms one frame at a
// time, 0.435 -> 0.423 batched; C2 0.119 -> 0.103 / 0.072 -> 0.065; C5 3.31 -> 3.28); 2-cell bricks fetch less still and
// are slower again (the look-ups), and their index outgrows a launch at 1024^3.  -DVR_BRICK_SHIFT=3 rebuilds round 1's.
#ifndef VR_BRICK_SHIFT
#define VR_BRICK_SHIFT 2
#endif
constexpr int kBrickShift = VR_BRICK_SHIFT;
constexpr int kBrickCells = 1 << kBrickShift;
constexpr float kBrickInv = 1.0f / (float)kBrickCells;    // exact
constexpr float kBrickHalf = 0.5f / (float)kBrickCells;   // half a cell in brick units (the -0.5 of the cell coordinate)
#ifndef VR_DIST_MAX
#define VR_DIST_MAX 128
#endif
constexpr int kDistMax = VR_DIST_MAX;  // cap of the brick distance field (one dilation pass per value when the field is rebuilt; < 255)
// Storage bricks of the bricked volume copy (DevVolume::bricked): 2^S voxels per axis, S = 2 (4 x 4 x 4 = 1 KiB of vec4 voxels)
// by default; -DVR_VOX_BRICK_SHIFT=1 / 3 rebuilds with 2^3- / 8^3-voxel bricks for A/B (tools/run_r3l.sh).
#ifndef VR_VOX_BRICK_SHIFT
#define VR_VOX_BRICK_SHIFT 2
#endif
constexpr unsigned kVbS = VR_VOX_BRICK_SHIFT;          // log2 of the brick edge
constexpr unsigned kVbM = (1u << kVbS) - 1u;           // mask of the in-brick coordinate
constexpr unsigned kVbN = 1u << (3u * kVbS);           // voxels per brick

struct DevVolume {
    const float4* data;  // reference layout: x fastest, (k*ny + j)*nx + i   (VolumeFile.cpp:306)
    // Scalar density plane: the .a of every voxel, same order, 4 B per voxel (built at upload, rebuilt after every in-place
    // change).  Every fetch that consumes .a alone reads it instead of the 16-byte voxels -- a quarter of the footprint in
    // L2 / Infinity Cache / HBM and four times the voxels per cache line.  a_base / a_shift address either form without a
    // branch: byte offset of voxel idx's density = idx << a_shift from a_base (the plane, or data + 12 bytes when the plane
    // is switched off for A/B measurements).
    const float* dens;
    const char* a_base;
    int a_shift;
    int nx, ny, nz;
    // Bricked layout (default, vr_set_volume_layout(0)): `data` and the density plane behind `a_base` hold the voxels
    // in bricks of 4 x 4 x 4, brick after brick (x fastest), the 64 voxels of a brick in x-fastest order: voxel (x, y, z) lives at
    //     (x >> 2) * 64 + (x & 3)  +  (y >> 2) * brick_row + (y & 3) * 4  +  (z >> 2) * brick_slab + (z & 3) * 16
    // (written with kVbS / kVbM / kVbN in the code: the brick edge is a build-time constant)
    // -- a sum of one term per axis, so the eight corners of a cell are sums of two terms per axis.  A 1 KiB brick is eight
    // 128-byte lines of 4 x 2 x 1 voxels: the 7 x 7 x 2 voxel patch a packet's corner load touches spans ~20 lines instead of
    // the ~30 of the reference's x-fastest rows, and the lines a ray needs next lie in the same or the neighbouring brick
    // whatever direction it travels in (with x-fastest rows a ray along z changes its 4 MiB slice every step).  bricked == 0:
    // the reference's order (VolumeFile.cpp:306), idx = (z * ny + y) * nx + x.
    int bricked;
    unsigned brick_row, brick_slab;  // voxels per row of bricks (ceil(nx / 4) * 64) and per slab of bricks (* ceil(ny / 4))
    unsigned data_bytes;             // size of `data` (the range of the kernels' buffer loads; volumes below 4 GiB)
    // 1: the workgroup's dynamic LDS holds this (bricked) volume's per-axis SLOT TABLES -- entry e (0 .. n + 1) of an axis = the slot
    // term of texel clamp(e - 1, 0, n - 1), the three axes one after the other -- and make_cell() reads the clamp-to-edge texel
    // pair of a coordinate t (-1 .. n - 1) as the entries t + 1, t + 2 with one ds_read2_b32 instead of computing clamps, shifts,
    // masks and multiplies (flavour 18: march_kernel fills the tables per workgroup; 0 elsewhere)
    int lut;
};

// Both tables are stored with their first and their last texel repeated once at either end: table[k] is at [k+1].
// The clamp-to-edge texel pair (i0, i1) of a linear fetch is then ALWAYS the adjacent pair [j], [j+1] with
// j = clamp(floor(x) + 1, 0, R): one index, no second clamp, and the second texel sits at a fixed offset.
struct DevTF {
    const float* opacity;  // R32Float[res_o + 2]     (OpacityTf.cpp:25-26)
    const float4* color;   // RGBA32Float[res_c + 2]  (ColorTf.cpp:23-24)
    int res_o, res_c;
};

// Kernel argument block (passed by value, lives in SGPRs / kernarg segment).
struct MarchParams {
    float proj_inv[16];
    float view_inv[16];
    int W, H;
    int fragment_mode;
    int steps_count;
    float step_size;
    float bmin[3], bmax[3];  // IsInSampleCoords bounds: 0.0f + clip?.x, 1.0f - clip?.y
    int toggle_varstep, toggle_jitter;
    float light_pos[3], light_amb[3], light_dif[3];
    float camera_pos[3];     // cameraPosition uniform (illustrative shader only)
    DevVolume vol[3];
    DevTF tf[2];
    // work decomposition: the launch walks the 64x64 tiles t = rank + n*world, n = 0..n_tiles-1
    int rank, world, tiles_x, tiles_y, n_tiles;
    int packed;              // 0: write frame[y*W+x]; 1: write packed tiles
    int prio_mode;           // 1: wavefronts with long remaining ray paths raise their issue priority (s_setprio)
    int xcd_mode;            // 0: the blocks of a tile share an XCD, 1: they are dealt over the XCDs
    int only_tile;           // experiment (VR_EXP_ONLY_TILE): >= 0 -> rays of every other tile ordinal do not march
    int rect[4];             // x0, y0, x1, y1 (inclusive): no ray outside this pixel rectangle can hit the box
    int n_blocks;            // logical blocks = n_tiles * kBlocksPerTile (grid is padded to a multiple of 8)
    // exact empty-space skipping (BASIC / LIGHT / THREE_FILES): per-brick maximum density of vol[0] over the
    // (c+1)^3 voxels a brick of c^3 base cells can touch (c = kBrickCells), and the length of the opacity table's zero prefix
    const float2* bricks;    // nullptr = disabled; per brick: x = max of vol[skip_vol].a, y = max(r,g,b) of the mask
                             // (y is filled from vol[0]'s bricks for VOLUME_MASK and is 0 otherwise)
    int use_rgb;             // VOLUME_MASK: a brick is inert only if its mask record y <= 0
    const unsigned char* brick_dist;  // per brick: 0 = active; k >= 1 = inert and every brick within Chebyshev
                                      // distance k-1 is inert too (capped); rebuilt when the volume / opacity table change
    int skip_vol;            // which volume carries the density that drives the opacity (0, or 2 for VOLUME_MASK)
    float abox[6];           // uvw box (lo xyz, hi xyz) around the ACTIVE bricks of brick_dist, one brick of margin: outside it nothing is sampled
    int bnx, bny, bnz;       // bricks per axis
    float bsx, bsy, bsz;     // n / kBrickCells per axis of vol[skip_vol] (exact in f32)
    int tf_zero_prefix;      // largest Z with opacity[0..Z] == 0 exactly (-1: none)
    int zskip_prefix;        // the same for the per-step vote of sample_and_blend (-1 with VR_EXP_NO_ZSKIP: never skips)
    // Launch order of the logical blocks: workgroup blockIdx.x works on logical block order[blockIdx.x] (nullptr =
    // identity).  The host sorts the blocks of the previous frame by their longest ray chain, longest first, so that the
    // long blocks start at once and the short ones fill the machine at the end (speed only: a permutation of the blocks).
    const unsigned* order;
    float4* out;
    unsigned long long* block_counts;  // [blocks of this frame][kBlockRecord]: composited, covered, fetched, t0, t1, hw id
    unsigned batch_n;        // frames the launch carries (1 .. kBatchMax): see MarchBatch
};

// One launch may carry up to kBatchMax frames of the same scene and shape (different uniforms, output and record buffers).
// A rank's share of a frame on N GPUs, or a small frame, is a launch too short to fill the machine; several of them in one
// launch do, without depending on how many streams the runtime really runs side by side.  The frames are
// interleaved in groups of 8 workgroups: group g = blockIdx.x / 8 belongs to frame g % n_frames and is that frame's group
// g / n_frames -- so a workgroup keeps the XCD residue of its index within the frame, and the longest-first launch order
// (MarchParams::order) holds across the whole launch: the long workgroups of EVERY frame start first.  (Frame after frame,
// the last frame's long ray chains would start when the others' blocks have all been dispatched.)  Passed by value:
// 4 x ~0.6 KB of the 4 KB kernarg segment.
constexpr int kBatchMax = 4;
struct MarchBatch {
    MarchParams frame[kBatchMax];
    unsigned n_frames;
};

// Work queue of the persistent-wavefront kernel (vr_pw.h): eight heads, one per class of the workgroup index modulo 8,
// zero at launch; heads[c * 64] counts the items of class c handed out beyond every wavefront's first.
struct PwQueue {
    unsigned* heads;
    unsigned n_items;  // logical blocks of the launch (a multiple of 8)
    unsigned dynamic;    // march_p2_kernel: 1 = a wavefront's first item comes from the heads as well (no static deal)
    unsigned p2_window;  // march_p2_kernel<.., WIN>: records per gather window when not 0 (tests: a small window on a small volume)
};

// What enqueue_render decided about one march launch; handed to launch_march of the arithmetic mode's translation unit
// (vr_launch.h: namespace vr = separately rounded multiply-adds, namespace vrf = fused).
struct LaunchDesc {
    int variant;      // vr_variant
    bool off32;       // every bound volume < 4 GiB: 32-bit byte offsets
    int leap_mode;    // LEAP template argument of march_kernel
    int dp;           // lanes per ray of march_dp_kernel (2 / 4), 0 = march_kernel
    bool dp_pipe;     // ... with the next round's corner loads software-pipelined
    bool wtb;         // LDS wave-tile kernel (lit shader, separate arithmetic only)
    bool otf;         // lit shader: corner gradients derived from the density plane
    bool lt;          // LDS tiles filled by LDS-DMA (vr_lt.h; lit shader)
    bool pw;          // persistent wavefronts (vr_pw.h): grid = workgroups of 1024 threads, the packets come from `queue`
    bool pw_ltf;      // ... with TF slot 0 in LDS (lds_bytes of dynamic LDS)
    bool pw_pipe;     // ... with the next step's corner loads software-pipelined (lit / unlit shader)
    bool pw_p2;       // ... the no-skip form with the corner loads two steps ahead (march_p2_kernel; TF slot 0 in LDS, bricked copy)
    bool pw_p2_skip;  // ... ... with skipping by whole wavefronts (march_p2_kernel<V, true>)
    bool pw_p2_win;   // ... ... a bound volume of 4 GiB or more: the gather window moves (march_p2_kernel<.., WIN>)
    unsigned lds_bytes;
    PwQueue queue;
    const unsigned* mixed_items;  // lanes per ray chosen per packet (vr_mixed.h): the item list, grid = its positions
    int n_logical;                // ... and the logical blocks of the launch (where the second halves' records start)
    dim3 grid, block;
};

}  // namespace vr
