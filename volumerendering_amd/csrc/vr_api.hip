// vr_api.hip -- implementation of the C ABI declared in include/vr.h on top of the gfx950 kernels.
// No CPU fallback exists behind this ABI (and nothing under oracle/ is referenced): without a usable HIP
// device vr_create fails with VR_ERR_HIP.
#include "../../include/vr.h"
#include "vr_launch.h"

// the same dispatch over the kernels compiled with fused multiply-adds (vr_fused.hip)
namespace vrf {
void launch_march(const vr::LaunchDesc& L, hipStream_t s, const vr::MarchBatch& B);
}

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace vr;

namespace {

thread_local std::string g_create_error;

struct Timing {
    hipEvent_t ev_begin = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_end = nullptr;
    bool valid = false;
};

constexpr int kRing = 256;
struct KernelRing {  // one (start, stop) event pair per render call, reused round-robin
    hipEvent_t k0[kRing] = {}, k1[kRing] = {};
    long long head = 0;  // total launches recorded since the last reset
};

}  // namespace

constexpr int kInFlight = 8;  // launches that may be in flight at a time (record buffers used in turn; twice the streams, so that
                               // a caller with four frames in flight never blocks on its oldest launch)
constexpr int kStreams = 4;   // vr_stream(): streams for frames in flight
constexpr int kOrderRing = 16;  // launch-order buffers: written behind launch k, read by launches k+3 .. k+6 only (see enqueue_render)

struct vr_ctx {
    int device = 0;
    uint32_t W = 0, H = 0;
    hipStream_t stream = nullptr;
    DevVolume vol[VR_MAX_VOLUMES] = {};
    size_t vol_bytes[VR_MAX_VOLUMES] = {};
    float2* vol_bricks[VR_MAX_VOLUMES] = {};  // per brick: (max density, max(r,g,b)) -- empty-space skipping
    float* vol_dens[VR_MAX_VOLUMES] = {};     // scalar density plane of each slot (DevVolume::dens)
    size_t vol_dens_cap[VR_MAX_VOLUMES] = {};  // in voxels
    float4* vol_bricked[VR_MAX_VOLUMES] = {};  // the voxels again in 4 x 4 x 4 bricks (DevVolume::bricked), what the march kernels gather from
    float* vol_bdens[VR_MAX_VOLUMES] = {};     // ... and their density plane in the same order
    size_t vol_bricked_cap[VR_MAX_VOLUMES] = {};  // in slots (bricks x 64)
    bool vol_grad_derived[VR_MAX_VOLUMES] = {};  // .rgb verified to be PreComputeGradient(false) of .a, bit for bit
    int arith = VR_ARITH_SEPARATE;             // vr_set_arithmetic
    int layout_mode = 0;                       // vr_set_volume_layout: 0 density plane for .a fetches, 1 vec4 voxels only,
                                               // 2 plane + lit gradients derived on the fly
    float2* merged_bricks = nullptr;           // VOLUME_MASK: (CT density max, mask rgb max), rebuilt when stale
    bool merged_stale = true;
    unsigned char* brick_dist = nullptr;       // distance field over the records in use; key below says for what
    size_t dist_cap = 0;
    const void* dist_records = nullptr;
    unsigned long long dist_epoch = ~0ull;     // volume-change counter the field was built at
    int dist_z = -2, dist_res = 0, dist_rgb = -1;
    unsigned long long brick_epoch = 0;        // bumped whenever any brick table changes
    int tf_zero_prefix[VR_MAX_TFS] = {-1, -1};  // zero prefix of each opacity table, -1 if none / not finite
    bool tf_color_finite[VR_MAX_TFS] = {false, false};
    bool tf_opacity_finite[VR_MAX_TFS] = {false, false};
    DevTF tf[VR_MAX_TFS] = {};
    float* tf_opacity[VR_MAX_TFS] = {};
    float4* tf_color[VR_MAX_TFS] = {};
    vr_uniforms u = {};
    bool have_uniforms = false;
    float4* d_frame = nullptr;
    float4* d_tiles = nullptr;
    size_t tiles_cap = 0;       // in float4
    int last_tiles = 0;         // tiles rendered by the last vr_render_tiles
    uint32_t* d_present = nullptr;
    unsigned long long* d_counters = nullptr;  // [3] composited, covered, fetched
    // per-workgroup records (store_block_counts), kInFlight buffers used in turn so that several frames can be in
    // flight on different streams (the next ones fill the machine while the first one's long rays drain)
    unsigned long long* d_block_counts[kInFlight] = {};
    size_t block_counts_cap[kInFlight] = {};   // in blocks
    hipEvent_t slot_done[kInFlight] = {};      // recorded behind the launch that last used the slot (any stream)
    bool slot_used[kInFlight] = {};
    unsigned* d_pw_heads = nullptr;            // queue heads of the persistent-wavefront kernel: kInFlight x 8 heads, 256 B apart
    bool pw_heads_dirty[kInFlight] = {};       // the slot's last persistent launch had no sort behind it to clear its heads
    // Longest-first launch order (MarchParams::order): behind every march launch one small kernel sorts that launch's
    // blocks by their longest ray chain; a later launch of the same shape takes its blocks in that order.
    struct OrderSlot {
        unsigned* buf = nullptr;
        size_t cap = 0;
        hipStream_t stream = nullptr;
        hipEvent_t sorted = nullptr;
        unsigned long long key = 0, seq = 0;
        unsigned long long scene_key = 0;  // what the launch rendered, whatever kernel form it took (the chain length's key)
        bool valid = false;
        unsigned* items = nullptr;  // mixed lanes per ray (vr_mixed.h): the item list built behind that launch's sort
        size_t items_cap = 0;
        bool has_items = false;
    } order_ring[kOrderRing];
    unsigned* h_items = nullptr;  // pinned, one word per ring slot: grid of a launch that takes that slot's item list (0 = not built yet)
    unsigned* h_split = nullptr;  // pinned: how many packets that list splits in two
    int split_pct = 75;           // a packet is split when its longest chain reaches this share of the launch's longest (VR_EXP_SPLIT_PCT)
    int split_min = 64;           // ... and at least this many samples (VR_EXP_SPLIT_MIN)
    unsigned last_split = 0;      // packets the last mixed launch marched with two lanes per ray
    unsigned long long* h_span = nullptr;  // pinned, kRing words: duration of launch q in 100 MHz ticks + 1, from its records (0 = not known)
    bool ring_events[kRing] = {};          // launch q was timed with the events k0 / k1 instead (no sort behind it)
    unsigned long long* h_end = nullptr;   // pinned, kRing words: end of launch q's last workgroup on the 100 MHz device clock, | 1 (0 = not known)
    unsigned* h_chain = nullptr;  // pinned, one word per ring slot: longest ray chain + 1 of that launch (0 = not known yet)
    // Measured kernel choice (flavour 0): every kernel form is bit-identical, so the context tries the eligible ones on
    // the caller's own frames and keeps the fastest by the launches' own records -- per "what is launched of what".
    struct Tune {
        unsigned long long key = 0;   // shader, share, viewport, frames per launch, frames in flight, scene epoch, arithmetic, layout (0 = free)
        unsigned long long shape = 0; // ... the same without the scene's epochs: a new scene starts from what the last one of this shape kept
        int n = 0, cand[6] = {};      // the eligible flavours; cand[0] = the prior's pick (what runs while nothing is known)
        int cur = 0, issued = 0;      // candidate on trial, launches it has had
        int per = 3, settle = 4;      // launches per candidate; launches before the trial starts (no launch order exists yet)
        long long launch[6][16] = {}; // ring.head of every trial launch of every candidate (other shapes' launches may lie in between)
        int choice = -1;              // index into cand of the kernel kept (-1 = trial running)
        unsigned chain_ref = 0;       // longest ray chain + 1 when it was chosen: the trial re-opens when that has moved by a quarter
        float cost[6] = {};           // ms per launch measured (0 = no data)
        unsigned long long used = 0;  // (least recently used slot is recycled)
    } tune[8];
    unsigned long long tune_clock = 0;
    unsigned long long tf_epoch = 0;  // bumped by every table upload
    int tune_mode = 1;                // VR_EXP_TUNE=0: the prior alone (round 3's thresholds)
    int frames_in_flight = 1;                 // vr_hint_frames_in_flight: frames the caller keeps in flight on different streams
    unsigned long long order_seq = 0;
    hipStream_t flight[kStreams] = {};  // vr_stream(): streams probed to run side by side (created on first use)
    int n_flight = 0;
    hipStream_t order_stream = nullptr;  // the sorts run here, behind their launch's event: never on a frame's critical path
    int order_mode = 1;  // 0 = launch the blocks in index order (VR_EXP_ORDER=0)
    // VR_EXP_HOST_ORDER_WAIT=1: one frame at a time, a launch waits for the two sorts it depends on (the launch order it reads, the sort
    // that read its record slot last) on the HOST, before it is enqueued, instead of on its stream (enqueue_render).  Off by default.
    int host_order_wait = 0;
    int cnt_buf = 0;                           // the buffer the last launch wrote
    bool cnt_pending = false;                  // block counts of the last launch not summed / copied yet
    int cnt_blocks = 0;
    bool event_timing = false;                 // vr_set_kernel_timing(VR_TIMING_EVENTS): time every launch with HIP events
    bool zskip = true;                         // per-step zero-opacity vote (VR_EXP_NO_ZSKIP=1 switches it off for A/B)
    size_t cnt_offset = 0;                     // ... and where in that buffer the records of its last frame start (u64 words)
    unsigned long long* h_counters = nullptr;  // pinned [3]
    Timing tm;
    KernelRing ring;
    int flavour = 0;
    int waves_per_block = 1;  // 1 (default: the launch order of section 4.6 works at wavefront granularity) or 4 (VR_EXP_WAVES_PER_BLOCK)
    int only_tile = -1;       // experiment knob VR_EXP_ONLY_TILE
    int prio_mode = 0;        // VR_EXP_PRIO=1: wave priority by remaining ray path (+2-3 % for one frame at a time,
                              // -2 % with frames in flight, where nothing waits for the long rays)
    int n_cus = 256;          // compute units of the device
    int default_flavour = 0;  // what flavour 0 resolves to (experiment knob VR_EXP_FLAVOUR)
    int last_flavour = 0;     // the flavour the last launch resolved to
    bool last_otf = false;    // ... and whether it derived the gradients from the density plane
    int xcd_mode = 2;         // deal a tile's 16x16 sub-blocks over the XCDs (VR_EXP_XCD=1: its packets one by one; 0: one XCD per tile)
    bool pw_ltf = true;       // persistent wavefronts keep TF slot 0 in LDS (VR_EXP_PW_LTF=0: from L1, for A/B)
    unsigned p2_threads = 0;  // flavours 16 / 17: threads per workgroup when not 768 (VR_EXP_P2_THREADS: fewer wavefronts per CU)
    unsigned p2_wgs = 0;      // flavours 16 / 17: workgroups per CU when not chosen by the launch (VR_EXP_P2_WGS)
    int p2_dynq = 0;          // flavours 16 / 17: every item from the queue heads, the wavefronts' first ones too (VR_EXP_P2_DYNQ=1; measured with
                              // launches in flight, where workgroups start as others retire: 0.522 against 0.502 ms per C3 frame, so off)
    unsigned p2_window = 0;   // flavours 16 / 17: records per gather window (VR_EXP_P2_WINDOW: the moving window of volumes >= 4 GiB, forced
                              // onto small volumes by the tests; 0 = what the hardware reaches, just below 4 GiB)
    double active_fraction = 1.0;  // share of bricks that are not inert, of the distance field in use
    float abox[6] = {-3.0e38f, -3.0e38f, -3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};  // uvw box around the active bricks of that field (MarchParams::abox)
    int pw_policy = 1;        // the default (flavour 0) may pick the persistent kernel (VR_EXP_PW_POLICY=0: never)
    std::string err;
};

namespace {

int fail(vr_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define VR_HIP(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return fail((c), e__ == hipErrorOutOfMemory ? VR_ERR_OOM : VR_ERR_HIP,                    \
                        std::string(#call) + " (vr_api.hip:" + std::to_string(__LINE__) + "): " + hipGetErrorString(e__));                          \
    } while (0)

int refresh_bricks(vr_ctx* c, int slot);

int tiles_x_of(const vr_ctx* c) { return (int)((c->W + kTile - 1) / kTile); }
int tiles_y_of(const vr_ctx* c) { return (int)((c->H + kTile - 1) / kTile); }

int tile_count(const vr_ctx* c, int rank, int world)
{
    int total = tiles_x_of(c) * tiles_y_of(c);
    if (rank >= total) return 0;
    return (total - rank + world - 1) / world;
}

bool is_identity(const float* m)
{
    for (int i = 0; i < 16; ++i)
        if (m[i] != ((i % 5 == 0) ? 1.0f : 0.0f)) return false;
    return true;
}

// volumes / TF pairs each variant samples (vr.h slot tables)
void variant_needs(int variant, int* nvol, int* ntf)
{
    switch (variant) {
    case VR_VARIANT_BASIC:
    case VR_VARIANT_LIGHT:
    case VR_VARIANT_LIGHT_INSHADER: *nvol = 1; *ntf = 1; break;
    case VR_VARIANT_VOLUME_MASK: *nvol = 3; *ntf = 2; break;
    case VR_VARIANT_THREE_FILES: *nvol = 2; *ntf = 2; break;  // the mask (slot 2) is bound but never sampled
    case VR_VARIANT_MULTI_CTRT: *nvol = 2; *ntf = 2; break;
    case VR_VARIANT_ILLUSTRATIVE: *nvol = 2; *ntf = 2; break;
    default: *nvol = 2; *ntf = 1; break;  // TF_CALIB
    }
}

int alloc_frame(vr_ctx* c)
{
    VR_HIP(c, hipSetDevice(c->device));
    if (c->d_frame) (void)hipFree(c->d_frame);
    if (c->d_present) (void)hipFree(c->d_present);
    c->d_frame = nullptr;
    c->d_present = nullptr;
    size_t n = (size_t)c->W * c->H;
    VR_HIP(c, hipMalloc(&c->d_frame, n * sizeof(float4)));
    VR_HIP(c, hipMalloc(&c->d_present, n * sizeof(uint32_t)));
    VR_HIP(c, hipMemsetAsync(c->d_frame, 0, n * sizeof(float4), c->stream));
    return VR_OK;
}

// Inverse of a column-major 4x4 in double precision (cofactors); false if singular / not finite.
bool invert4(const float* m, double* o)
{
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    const double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (!(det - det == 0.0) || det == 0.0) return false;
    for (int i = 0; i < 16; ++i) {
        o[i] = inv[i] / det;
        if (!(o[i] - o[i] == 0.0)) return false;
    }
    return true;
}

// Pixel rectangle outside which no ray can hit the box [-.5,.5]^2 x [-.25,.25]: the rays are defined by proj_inv and
// view_inv (setup_ray), so the box corners are projected with the inverses of exactly those.  With every corner in
// front of the eye the box projects inside the hull of its corners; 3 pixels of margin dwarf the rounding.  Anything
// doubtful (singular matrices, a corner at or behind the eye plane, non-finite numbers) -> the whole frame.
void hit_rectangle(const vr_uniforms& u, int W, int H, int rect[4])
{
    rect[0] = 0;
    rect[1] = 0;
    rect[2] = W - 1;
    rect[3] = H - 1;
    double proj[16], view[16];
    if (!invert4(u.proj_inv, proj) || !invert4(u.view_inv, view)) return;
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
    for (int k = 0; k < 8; ++k) {
        const double wp[4] = {(k & 1) ? 0.5 : -0.5, (k & 2) ? 0.5 : -0.5, (k & 4) ? 0.25 : -0.25, 1.0};
        double e[4], cl[4];
        for (int r = 0; r < 4; ++r) e[r] = view[r] * wp[0] + view[4 + r] * wp[1] + view[8 + r] * wp[2] + view[12 + r] * wp[3];
        for (int r = 0; r < 4; ++r) cl[r] = proj[r] * e[0] + proj[4 + r] * e[1] + proj[8 + r] * e[2] + proj[12 + r] * e[3];
        if (!(cl[3] > 1e-9)) return;
        const double px = (cl[0] / cl[3] + 1.0) * 0.5 * W, py = (1.0 - cl[1] / cl[3]) * 0.5 * H;
        if (!(px - px == 0.0) || !(py - py == 0.0)) return;
        x0 = px < x0 ? px : x0;
        x1 = px > x1 ? px : x1;
        y0 = py < y0 ? py : y0;
        y1 = py > y1 ? py : y1;
    }
    auto clampd = [](double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); };
    rect[0] = (int)clampd(x0 - 3.0, 0.0, (double)W);
    rect[1] = (int)clampd(y0 - 3.0, 0.0, (double)H);
    rect[2] = (int)clampd(x1 + 3.0, -1.0, (double)(W - 1));
    rect[3] = (int)clampd(y1 + 3.0, -1.0, (double)(H - 1));
}

// finite and of moderate size: products of a colour, a light term and a shading factor stay finite, so "x * 0 == 0"
// holds for everything a provably-zero opacity is multiplied with
bool all_finite(const float* v, int n)
{
    for (int i = 0; i < n; ++i)
        if (!(v[i] - v[i] == 0.0f) || !(v[i] <= 1.0e15f && v[i] >= -1.0e15f)) return false;
    return true;
}

// the fields of a launch's parameters that come from the uniforms of ONE frame
void fill_frame_params(MarchParams& P, const vr_uniforms& u)
{
    std::memcpy(P.proj_inv, u.proj_inv, sizeof P.proj_inv);
    std::memcpy(P.view_inv, u.view_inv, sizeof P.view_inv);
    hit_rectangle(u, P.W, P.H, P.rect);
    P.fragment_mode = u.fragment_mode;
    P.steps_count = u.steps_count;
    P.step_size = u.step_size;
    // IsInSampleCoords bounds, BasicVolumeApp.wgsl:73-74 (same f32 expressions as the shader)
    P.bmin[0] = 0.0f + u.clip_x[0]; P.bmin[1] = 0.0f + u.clip_y[0]; P.bmin[2] = 0.0f + u.clip_z[0];
    P.bmax[0] = 1.0f - u.clip_x[1]; P.bmax[1] = 1.0f - u.clip_y[1]; P.bmax[2] = 1.0f - u.clip_z[1];
    P.toggle_varstep = u.toggles[0];
    P.toggle_jitter = u.toggles[1];
    for (int i = 0; i < 3; ++i) {
        P.light_pos[i] = u.light_pos[i];
        P.light_amb[i] = u.light_ambient[i];
        P.light_dif[i] = u.light_diffuse[i];
        P.camera_pos[i] = u.camera_pos[i];
    }
}

// The measured kernel choice (flavour 0).  `cand[0 .. n)` are the flavours that may run this launch (cand[0] = the prior's pick); returns
// the one to launch now.  A trial gives every candidate `per` launches in turn -- after `settle` launches of the prior, so that a
// launch order exists (the trial then measures what the steady state runs) -- and reads the launches' durations from
// the pinned words their sorts fill (no synchronisation: a trial is evaluated when its last word has arrived; until then the
// prior runs).  One launch at a time: the shortest first-start-to-last-end span of a candidate's launches but its first.  Launches
// in flight: the mean interval between the ends of its consecutive launches that ran beside launches of the same candidate only
// (3 x in_flight + 2 launches per turn, the first and the last in_flight of them not used).  The trial re-opens when the scene, the tables,
// the launch shape or the frames-in-flight hint change (the key) and when the longest ray chain has moved by a quarter.
int tune_pick(vr_ctx* c, unsigned long long key, unsigned long long shape, const int* cand, int n, unsigned chain_now, bool measurable)
{
    if (n <= 1) return cand[0];
    vr_ctx::Tune* t = nullptr;
    for (auto& e : c->tune)
        if (e.key == key) t = &e;
    const int in_flight = c->frames_in_flight;
    auto reset = [&](vr_ctx::Tune& e, int first) {
        e.key = key;
        e.shape = shape;
        e.n = 0;
        e.cand[e.n++] = first;
        for (int i = 0; i < n; ++i)
            if (cand[i] != first && e.n < 6) e.cand[e.n++] = cand[i];
        e.cur = 0;
        e.issued = 0;
        e.per = in_flight > 1 ? 3 * in_flight + 2 : 3;  // (<= 14: kStreams is 4)
        e.settle = in_flight + 3;
        e.choice = -1;
        e.chain_ref = 0;
        for (int i = 0; i < 6; ++i) {
            e.cost[i] = 0.0f;
            for (int q = 0; q < 16; ++q) e.launch[i][q] = -1;
        }
    };
    if (!t) {
        // a new scene (or table, or arithmetic) of a shape that has been measured before: what that trial kept runs first, if it is
        // still eligible -- a host that edits a table frame after frame keeps its kernel while every new trial settles
        int first = cand[0];
        unsigned long long newest = 0;
        for (const auto& e : c->tune)
            if (e.key != 0 && e.shape == shape && e.choice >= 0 && e.used > newest)
                for (int i = 0; i < n; ++i)
                    if (cand[i] == e.cand[e.choice]) {
                        first = cand[i];
                        newest = e.used;
                    }
        t = &c->tune[0];
        for (auto& e : c->tune)
            if (e.used < t->used) t = &e;
        reset(*t, first);
    } else {
        // the eligible set may have changed under the same key (a flavour knob, a table that fits LDS no more)
        bool same = t->n == n;
        for (int i = 0; i < n && same; ++i) {
            bool found = false;
            for (int j = 0; j < t->n; ++j) found = found || t->cand[j] == cand[i];
            same = found;
        }
        if (!same) reset(*t, cand[0]);
    }
    t->used = ++c->tune_clock;
    if (t->choice >= 0) {
        if (chain_now != 0 && t->chain_ref != 0) {
            const unsigned lo = t->chain_ref - t->chain_ref / 4, hi = t->chain_ref + t->chain_ref / 4;
            if (chain_now < lo || chain_now > hi) reset(*t, t->cand[t->choice]);  // (the kernel kept so far runs while the new trial settles)
        }
        if (t->choice >= 0) return t->cand[t->choice];
    }
    if (!measurable) return t->cand[0];
    if (t->settle > 0) {
        --t->settle;
        return t->cand[0];
    }
    if (t->cur < t->n) {
        const int f = t->cand[t->cur];
        t->launch[t->cur][t->issued] = c->ring.head;  // (the ring slot this launch will record itself in)
        if (++t->issued == t->per) {
            ++t->cur;
            t->issued = 0;
        }
        return f;
    }
    // every candidate has had its turn: are the records in?
    const long long last = t->launch[t->n - 1][t->per - 1];
    // (a launch of the trial was never measured -- timed with events, or not ordered -- or so many launches of other shapes ran in
    // between that the trial's first ring slots are about to be written again: keep the prior)
    if (c->ring.head > last + 64 || c->ring.head - t->launch[0][0] >= kRing) {
        t->choice = 0;
        t->chain_ref = chain_now;
        return t->cand[0];
    }
    for (int i = 0; i < t->n; ++i)
        for (int q = 0; q < t->per; ++q)
            if (*(volatile unsigned long long*)&c->h_span[t->launch[i][q] % kRing] == 0) return t->cand[0];
    int best = 0;
    for (int i = 0; i < t->n; ++i) {
        double ticks;
        if (in_flight > 1) {
            // (its first `in_flight` launches ran beside the candidate before it, its last ones beside the next: the ends of the
            // launches in between are `in_flight + 2` intervals apart that are this candidate's alone)
            const unsigned long long e0 = *(volatile unsigned long long*)&c->h_end[t->launch[i][in_flight] % kRing];
            const unsigned long long e1 = *(volatile unsigned long long*)&c->h_end[t->launch[i][t->per - in_flight] % kRing];
            ticks = e1 > e0 ? (double)(e1 - e0) / (double)(t->per - 2 * in_flight) : 1.0e18;
        } else {
            ticks = 1.0e18;
            for (int q = 1; q < t->per; ++q) {
                const double v = (double)*(volatile unsigned long long*)&c->h_span[t->launch[i][q] % kRing];
                ticks = v < ticks ? v : ticks;
            }
        }
        t->cost[i] = (float)(ticks * 1.0e-5);  // 100 MHz ticks -> ms
        // (another kernel must be 2 % faster than the prior's to replace it: the spans of equal kernels differ by about that much)
        // (... with launches in flight by 5 %: a candidate's interior launches still run beside its neighbours' tails -- a trial that
        // measured march_kernel at 0.407 ms per C3 frame pipelined against 0.418 kept it, and it then ran at 0.467: gpurun_out/s2p)
        if (i > 0 && t->cost[i] < t->cost[best] * (best == 0 ? (in_flight > 1 ? 0.95f : 0.98f) : 1.0f)) best = i;
    }
    t->choice = best;
    t->chain_ref = chain_now;
    return t->cand[best];
}

// Enqueue one launch on `s`: ONE frame with the context's uniforms into `out` (nullptr -> ctx-owned buffer), or, with
// batch_u / batch_out, n_frames (2 .. kBatchMax) frames of the same scene, each with its own uniforms and output buffer.
int enqueue_render(vr_ctx* c, int variant, int rank, int world, bool packed, float4* out, hipStream_t s, bool frame_events,
                   int n_frames = 1, const vr_uniforms* batch_u = nullptr, void* const* batch_out = nullptr)
{
    if (variant < 0 || variant >= VR_VARIANT_COUNT) return fail(c, VR_ERR_INVALID_ARG, "vr_render: bad variant");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, VR_ERR_INVALID_ARG, "vr_render: bad rank/world");
    if (n_frames < 1 || n_frames > kBatchMax) return fail(c, VR_ERR_INVALID_ARG, "vr_render: 1 .. 4 frames per launch");
    if (batch_u) {
        if (!batch_out) return fail(c, VR_ERR_INVALID_ARG, "vr_render: a batch needs its output buffers");
        for (int f = 0; f < n_frames; ++f) {
            if (!batch_out[f]) return fail(c, VR_ERR_INVALID_ARG, "vr_render: output buffer " + std::to_string(f) + " of the batch is NULL");
            if (batch_u[f].steps_count < 0) return fail(c, VR_ERR_INVALID_ARG, "vr_render: negative steps_count");
            if (!is_identity(batch_u[f].model))  // (as vr_set_uniforms)
                return fail(c, VR_ERR_UNSUPPORTED, "vr_render: model matrix must be the identity (App/src/Application.cpp:489-492)");
        }
        out = (float4*)batch_out[0];
    } else {
        if (n_frames != 1) return fail(c, VR_ERR_INVALID_ARG, "vr_render: several frames per launch need their uniforms");
        if (!c->have_uniforms) return fail(c, VR_ERR_NOT_READY, "vr_render: vr_set_uniforms has not been called");
    }
    const vr_uniforms& u0 = batch_u ? batch_u[0] : c->u;
    int nvol, ntf;
    variant_needs(variant, &nvol, &ntf);
    bool off32 = true;
    for (int i = 0; i < nvol; ++i) {
        if (!c->vol[i].data) return fail(c, VR_ERR_NOT_READY, "vr_render: volume slot " + std::to_string(i) + " is empty");
        if (c->vol_bytes[i] > 0xFFFFFFFFull) off32 = false;
    }
    for (int i = 0; i < ntf; ++i)
        if (!c->tf[i].opacity || !c->tf[i].color)
            return fail(c, VR_ERR_NOT_READY, "vr_render: TF slot " + std::to_string(i) + " is empty");
    if (u0.steps_count < 0) return fail(c, VR_ERR_INVALID_ARG, "vr_render: negative steps_count");
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();  // a stale error of somebody else's call must not be reported as a failed launch below

    MarchParams P;
    std::memset(&P, 0, sizeof P);
    P.W = (int)c->W;
    P.H = (int)c->H;
    fill_frame_params(P, u0);
    for (int i = 0; i < VR_MAX_VOLUMES; ++i) {
        P.vol[i] = c->vol[i];
        const bool plane = c->layout_mode != 1 && c->vol_dens[i] && c->vol[i].data;
        P.vol[i].dens = plane ? c->vol_dens[i] : nullptr;
        P.vol[i].a_base = plane ? reinterpret_cast<const char*>(c->vol_dens[i]) : reinterpret_cast<const char*>(c->vol[i].data) + 12;
        P.vol[i].a_shift = plane ? 2 : 4;
        P.vol[i].bricked = 0;
        P.vol[i].brick_row = P.vol[i].brick_slab = 0;
        const size_t lin_bytes = c->vol_bytes[i];
        P.vol[i].data_bytes = lin_bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)lin_bytes;
    }
    // the bricked copies (layout 0) are what the gathers read; decided below, once the kernel form is known (the LDS wave-tile
    // flavours and the on-the-fly gradients address the reference's x-fastest order)
    auto use_bricked = [&]() {
        for (int i = 0; i < VR_MAX_VOLUMES; ++i) {
            if (!c->vol[i].data || !c->vol_bricked[i] || !c->vol_bdens[i]) continue;
            const unsigned nbx = ((unsigned)c->vol[i].nx + kVbM) >> kVbS, nby = ((unsigned)c->vol[i].ny + kVbM) >> kVbS, nbz = ((unsigned)c->vol[i].nz + kVbM) >> kVbS;
            const size_t slots = (size_t)nbx * nby * nbz * kVbN;
            if (slots > 0xFFFFFFFFull) continue;  // (indices are 32 bits)
            P.vol[i].data = c->vol_bricked[i];
            P.vol[i].a_base = reinterpret_cast<const char*>(c->vol_bdens[i]);
            P.vol[i].a_shift = 2;
            P.vol[i].bricked = 1;
            P.vol[i].brick_row = nbx * kVbN;
            P.vol[i].brick_slab = nbx * nby * kVbN;
            P.vol[i].data_bytes = slots * 16 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)(slots * 16);
        }
    };
    for (int i = 0; i < VR_MAX_TFS; ++i) P.tf[i] = c->tf[i];
    P.rank = rank;
    P.world = world;
    P.tiles_x = tiles_x_of(c);
    P.tiles_y = tiles_y_of(c);
    P.n_tiles = tile_count(c, rank, world);
    P.packed = packed ? 1 : 0;
    P.n_blocks = P.n_tiles * kBlocksPerTile;
    P.only_tile = c->only_tile;
    P.prio_mode = c->prio_mode;
    P.xcd_mode = c->xcd_mode;
    // exact empty-space skipping: only for the shaders whose opacity is the CT table value alone, only when a
    // zero-opacity sample is provably the identity (finite colour table and light), and unless flavour 1 asks
    // for the plain kernel
    const bool skip_variant = variant == VR_VARIANT_BASIC || variant == VR_VARIANT_LIGHT ||
                              variant == VR_VARIANT_THREE_FILES || variant == VR_VARIANT_VOLUME_MASK ||
                              variant == VR_VARIANT_LIGHT_INSHADER;
    const int sv = (variant == VR_VARIANT_VOLUME_MASK) ? 2 : 0;  // the volume whose density drives tf[0]'s opacity
    int fl = c->flavour == 0 ? c->default_flavour : c->flavour;
    unsigned chain_known = 0;  // longest ray chain + 1 of the most recent launch of this scene shape whose sort has reported (0: none)
    if (fl == 0) {
        // Default: pick the lanes per ray from what will be on the machine.  With many rays per hardware lane the machine is
        // throughput-bound and one lane per ray does the least work; with few (a small frame, or one GPU's share of the
        // tiles) the frame waits for its longest rays, whose chains of dependent samples the depth-parallel kernel cuts to a
        // half or a quarter (vr_dp.h).  Two things refine the round-1 rule (thresholds measured on C3 at 1 / 2 / 4 / 8 ranks):
        //  * frames in flight: when the caller keeps several frames in flight on different streams (it says so with
        //    vr_hint_frames_in_flight; asking the events instead flushes the runtime's command batches and costs more than it
        //    tells) the other launches fill the machine as well, so the rays per lane count once per frame in flight (a
        //    rank's half of C3, two frames pipelined: 0.34 ms with one lane, 0.42 with two);
        //  * how long the chains really are: the longest ray chain of an earlier launch of this shape (written to pinned
        //    memory by the launch-order sort; read here without synchronising, 0 = not known).  Chains too short to matter --
        //    under 128 samples, 0.2 ms (C2: 102) -- leave nothing for the depth-parallel kernels to cut (C2: 0.133 / 0.091 ms per frame
        //    with one lane, 0.153 / 0.123 with two), unless the launch is too small to fill the machine at all.
        const long long px = (long long)tile_count(c, rank, world) * kTile * kTile;
        const int in_flight = c->frames_in_flight;  // the caller's hint (vr_hint_frames_in_flight)
        const double rays_per_lane = (double)px * in_flight * n_frames / ((double)c->n_cus * 4.0 * 5.0 * 64.0);
        unsigned chain = 0;  // longest chain + 1 of the most recent launch of this scene shape whose sort has reported
        if (c->h_chain) {
            const unsigned long long skey = ((unsigned long long)variant << 16) ^ ((unsigned long long)world << 8) ^ (unsigned long long)rank ^
                                            (packed ? 1ull << 63 : 0ull) ^ ((unsigned long long)c->W << 40) ^ ((unsigned long long)c->H << 24);
            unsigned long long best_seq = 0;
            for (int i = 0; i < kOrderRing; ++i) {
                const unsigned v = *(volatile unsigned*)&c->h_chain[i];
                if (v != 0 && c->order_ring[i].scene_key == skey && c->order_ring[i].seq + 1 > best_seq) {
                    best_seq = c->order_ring[i].seq + 1;
                    chain = v;
                }
            }
        }
        chain_known = chain;
        const bool short_chains = chain != 0 && chain - 1 < 128;
        // (two lanes per ray from 2 rays per lane on, four below: re-measured on the bricked layout -- a rank's quarter of C3
        // (1.6 rays per lane), one frame at a time: 0.274 ms with two lanes, 0.203 with four; a rank's half (3.2): 0.362 / 0.377;
        // a quarter with two launches in flight counts 3.2 and keeps two lanes: 0.190 / 0.217 per frame)
        fl = (rays_per_lane >= 4.5 || (short_chains && rays_per_lane >= 1.2)) ? 6 : (rays_per_lane >= 2.0 ? 11 : 10);
    }
    // persistent wavefronts (12, 13; vr_pw.h) exist for launches of one frame
    if ((fl == 12 || fl == 13) && n_frames != 1) fl = 6;
    // two steps ahead (16, 17; march_p2_kernel, vr_p2.h): lit / unlit shader and the three-volume composite (with its brick records:
    // checked once can_skip is known); TF slot 0 (one resolution for both tables) and the three axis tables in LDS; the bricked copy
    // with 32-bit slots, rows and slabs of bricks below 2^24 slots; a volume of 4 GiB or more through a moving window of at least
    // four z-slabs of bricks.  Launches of several frames and launches in flight included.  Else 13 / 12 (one frame) or 6.
    const int p2_vol = variant == VR_VARIANT_VOLUME_MASK ? 2 : 0;
    bool p2_ok = (variant == VR_VARIANT_LIGHT || variant == VR_VARIANT_BASIC || variant == VR_VARIANT_VOLUME_MASK) && c->pw_ltf &&
                 c->tf[0].res_o == c->tf[0].res_c && c->tf[0].res_o + 2 <= 8192 && c->layout_mode == 0 && c->vol_bricked[p2_vol] && c->vol_bdens[p2_vol];
    unsigned p2_lds = 0;
    if (p2_ok) {
        const size_t nbx = ((unsigned)c->vol[p2_vol].nx + kVbM) >> kVbS, nby = ((unsigned)c->vol[p2_vol].ny + kVbM) >> kVbS, nbz = ((unsigned)c->vol[p2_vol].nz + kVbM) >> kVbS;
        const size_t slab = nbx * nby * kVbN, window = variant == VR_VARIANT_BASIC ? 0x3fffffffull : 0x0fffffffull;
        const size_t lds = (size_t)(c->tf[0].res_o + 2) * 16 + ((size_t)c->vol[p2_vol].nx + c->vol[p2_vol].ny + c->vol[p2_vol].nz + 3) * 8;
        p2_ok = slab * nbz <= 0xFFFFFFFFull && slab < (1u << 24) && (c->p2_window ? c->p2_window / slab >= 3 : window / slab >= 4) && lds <= 160u * 1024u;
        p2_lds = (unsigned)lds;
    }
    if ((fl == 16 || fl == 17) && !p2_ok) fl = n_frames != 1 ? 6 : (fl == 16 ? 13 : 12);
    // 18: march_kernel with the slot tables of volume 0 in its workgroup's LDS (make_cell_lut): the shaders that sample ONE volume, the
    // bricked copy with 32-bit slots (bricks of rows and slabs below 2^24 slots as above); else 6
    bool lut_ok = (variant == VR_VARIANT_LIGHT || variant == VR_VARIANT_BASIC || variant == VR_VARIANT_LIGHT_INSHADER) && c->layout_mode == 0 &&
                  c->vol_bricked[0] && c->vol_bdens[0];
    unsigned lut_lds = 0;
    if (lut_ok) {
        const size_t nbx = ((unsigned)c->vol[0].nx + kVbM) >> kVbS, nby = ((unsigned)c->vol[0].ny + kVbM) >> kVbS, nbz = ((unsigned)c->vol[0].nz + kVbM) >> kVbS;
        lut_lds = (unsigned)(((size_t)c->vol[0].nx + c->vol[0].ny + c->vol[0].nz + 6) * 4);
        lut_ok = nbx * nby * nbz * kVbN <= 0xFFFFFFFFull && lut_lds <= 32u * 1024u;
    }
    if (fl == 18 && !lut_ok) fl = 6;
    if (fl == 16 && variant == VR_VARIANT_VOLUME_MASK) fl = 17;  // (the composite's form is the skipping one: its mask records)
    // LDS tiles (15; vr_lt.h): the lit shader, launches of one frame
    if (fl == 15 && (n_frames != 1 || variant != VR_VARIANT_LIGHT)) fl = 6;
    // mixed lanes per ray (14; vr_mixed.h): launches of one frame, shaders that have a depth-parallel form
    if (fl == 14 && (n_frames != 1 || variant == VR_VARIANT_ILLUSTRATIVE || variant == VR_VARIANT_LIGHT_INSHADER)) fl = 6;
    // the illustrative shader's opacity reads the accumulated alpha: its steps cannot be sampled side by side
    if (variant == VR_VARIANT_ILLUSTRATIVE && (fl == 7 || fl == 8 || fl == 10 || fl == 11)) fl = 6;
    // the in-shader gradient variant (seven density fetches per sample) exists as the one-lane kernel only
    if (variant == VR_VARIANT_LIGHT_INSHADER && fl != 1 && fl != 4 && fl != 5 && fl != 12 && fl != 13 && fl != 18) fl = 6;
    c->last_flavour = fl;
    bool can_skip = skip_variant && fl != 1 && fl != 2 && c->vol_bricks[sv] && c->tf_zero_prefix[0] >= 0 &&
                    c->tf_color_finite[0];
    for (int f = 0; f < n_frames; ++f) can_skip = can_skip && all_finite(batch_u ? batch_u[f].light_pos : c->u.light_pos, 12);
    // the kernels index bricks with 24-bit multiplies and 32-bit byte offsets
    can_skip = can_skip && ((c->vol[sv].nx + kBrickCells - 1) >> kBrickShift) * (long long)((c->vol[sv].ny + kBrickCells - 1) >> kBrickShift) < (1 << 23);
    if (variant == VR_VARIANT_THREE_FILES) can_skip = can_skip && c->tf_color_finite[1] && c->tf_opacity_finite[1];
    if (variant == VR_VARIANT_VOLUME_MASK)  // mask and CT must share one grid so that one brick index serves both
        can_skip = can_skip && c->vol_bricks[0] && c->vol[0].nx == c->vol[2].nx && c->vol[0].ny == c->vol[2].ny &&
                   c->vol[0].nz == c->vol[2].nz;
    if ((fl == 16 || fl == 17) && variant == VR_VARIANT_VOLUME_MASK && !can_skip) {  // (no brick records: no on-demand mask fetch)
        fl = n_frames != 1 ? 6 : 12;
        c->last_flavour = fl;
    }
    if (can_skip) {
        P.skip_vol = sv;
        P.bnx = (c->vol[sv].nx + kBrickCells - 1) >> kBrickShift;
        P.bny = (c->vol[sv].ny + kBrickCells - 1) >> kBrickShift;
        P.bnz = (c->vol[sv].nz + kBrickCells - 1) >> kBrickShift;
        P.bsx = (float)c->vol[sv].nx * kBrickInv;
        P.bsy = (float)c->vol[sv].ny * kBrickInv;
        P.bsz = (float)c->vol[sv].nz * kBrickInv;
        P.tf_zero_prefix = c->tf_zero_prefix[0];
        P.zskip_prefix = c->zskip ? P.tf_zero_prefix : -2;  // (-2: no vote, and mask and dose are always fetched)
        P.bricks = c->vol_bricks[sv];
        P.use_rgb = 0;
        if (variant == VR_VARIANT_VOLUME_MASK) {
            const int nb = P.bnx * P.bny * P.bnz;
            if (c->merged_stale || !c->merged_bricks) {
                if (c->merged_bricks) (void)hipFree(c->merged_bricks);
                c->merged_bricks = nullptr;
                VR_HIP(c, hipMalloc(&c->merged_bricks, (size_t)nb * sizeof(float2)));
                hipLaunchKernelGGL(merge_bricks_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, c->vol_bricks[2],
                                   c->vol_bricks[0], c->merged_bricks, nb);
                VR_HIP(c, hipGetLastError());
                c->merged_stale = false;
            }
            P.bricks = c->merged_bricks;
            P.use_rgb = 1;
        }
        // distance field over the inert bricks (Chebyshev distance to the nearest active brick), rebuilt when the
        // records, the zero prefix or the table resolution changed since it was last built
        const int nb = P.bnx * P.bny * P.bnz;
        if (c->dist_records != (const void*)P.bricks || c->dist_epoch != c->brick_epoch || c->dist_z != P.tf_zero_prefix ||
            c->dist_res != c->tf[0].res_o || c->dist_rgb != P.use_rgb || !c->brick_dist) {
            // (rare: an input changed.  Frames may be in flight on other streams and read the field: drain them first,
            // and finish the rebuild before any other stream's launch can follow)
            VR_HIP(c, hipDeviceSynchronize());
            if ((size_t)nb > c->dist_cap) {
                if (c->brick_dist) (void)hipFree(c->brick_dist);
                c->brick_dist = nullptr;
                c->dist_cap = 0;
                VR_HIP(c, hipMalloc(&c->brick_dist, (size_t)nb));
                c->dist_cap = (size_t)nb;
            }
            const dim3 g((unsigned)((nb + 255) / 256)), b(256);
            hipLaunchKernelGGL(brick_active_kernel, g, b, 0, s, P.bricks, c->brick_dist, nb, P.use_rgb, P.tf_zero_prefix,
                               c->tf[0].res_o);
            for (int k = 1; k < kDistMax; ++k)
                hipLaunchKernelGGL(brick_dist_pass_kernel, g, b, 0, s, c->brick_dist, P.bnx, P.bny, P.bnz, k);
            hipLaunchKernelGGL(brick_dist_cap_kernel, g, b, 0, s, c->brick_dist, nb);
            VR_HIP(c, hipGetLastError());
            {   // share of active bricks (steers the default kernel choice below)
                unsigned* d_cnt = reinterpret_cast<unsigned*>(c->d_counters);
                VR_HIP(c, hipMemsetAsync(d_cnt, 0, sizeof(unsigned), s));
                hipLaunchKernelGGL(count_active_bricks_kernel, g, b, 0, s, c->brick_dist, nb, d_cnt);
                VR_HIP(c, hipGetLastError());
                unsigned cnt = 0;
                VR_HIP(c, hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, s));
                VR_HIP(c, hipStreamSynchronize(s));
                c->active_fraction = nb > 0 ? (double)cnt / (double)nb : 1.0;
            }
            {   // the box of the active bricks, in uvw with one brick of margin (MarchParams::abox): brick b of axis a holds the
                // positions with p * bs - kBrickHalf in [b, b + 1), the first and the last brick those beyond them as well
                int* d_box = reinterpret_cast<int*>(c->d_counters);  // (6 ints: the counters' 24 bytes)
                int box[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1, -1, -1};
                VR_HIP(c, hipMemcpyAsync(d_box, box, sizeof box, hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(active_brick_box_kernel, g, b, 0, s, c->brick_dist, P.bnx, P.bny, P.bnz, d_box);
                VR_HIP(c, hipGetLastError());
                VR_HIP(c, hipMemcpyAsync(box, d_box, sizeof box, hipMemcpyDeviceToHost, s));
                VR_HIP(c, hipStreamSynchronize(s));
                const double bs[3] = {(double)P.bsx, (double)P.bsy, (double)P.bsz};
                for (int a = 0; a < 3; ++a) {
                    if (box[3 + a] < 0) {  // (no active brick: every ray misses)
                        c->abox[a] = 3.0e38f;
                        c->abox[3 + a] = -3.0e38f;
                    } else {
                        c->abox[a] = (float)(((double)box[a] - 1.0 + (double)kBrickHalf) / bs[a]);
                        c->abox[3 + a] = (float)(((double)box[3 + a] + 2.0 + (double)kBrickHalf) / bs[a]);
                    }
                }
            }
            VR_HIP(c, hipStreamSynchronize(s));
            c->dist_records = (const void*)P.bricks;
            c->dist_epoch = c->brick_epoch;
            c->dist_z = P.tf_zero_prefix;
            c->dist_res = c->tf[0].res_o;
            c->dist_rgb = P.use_rgb;
        }
        P.brick_dist = c->brick_dist;
        for (int a = 0; a < 6; ++a) P.abox[a] = c->abox[a];
    }

    // Default choice, second part -- THE PRIOR: what runs before anything has been measured.  Whole frames of the lit / unlit shader
    // and of the composite, one launch at a time: the kernel with the corner loads two steps ahead (vr_p2.h) -- 17, or 16 where next to
    // nothing can be skipped (noisy air under the default ramp 2.95 -> 1.97 ms; C3 0.65 -> 0.51; C4 0.72 -> 0.57) -- unless an earlier
    // launch of this shape says its chains are short (C2, longest chain 102: a packet is too short for the pipeline's fill and a
    // dequeue, 0.111 -> 0.161).  The same with launches in flight and several frames per launch since the approach loop (C3 0.417 / 0.382
    // ms per frame against march_kernel's 0.464 / 0.445; C5 level); shares of a frame: the first part's choice.
    const bool p2_variant = variant == VR_VARIANT_LIGHT || variant == VR_VARIANT_BASIC || (variant == VR_VARIANT_VOLUME_MASK && can_skip);
    const long long px_all = (long long)tile_count(c, rank, world) * kTile * kTile;
    const bool whole_frame = (double)px_all / ((double)c->n_cus * 4.0 * 5.0 * 64.0) >= 4.5;
    const bool nothing_to_skip = !can_skip || c->active_fraction >= 0.9;
    const bool auto_choice = c->flavour == 0 && c->default_flavour == 0;
    if (auto_choice && c->pw_policy && fl == 6 && whole_frame && p2_ok && p2_variant) {
        const bool short_chains = chain_known != 0 && chain_known - 1 < 128;
        if (nothing_to_skip && variant != VR_VARIANT_VOLUME_MASK) fl = 16;
        else if (!short_chains) fl = 17;
    }
    // ... and THE MEASURED CHOICE (tune_pick): the eligible forms take turns on the caller's own frames, the fastest by the launches'
    // own records stays.  Candidates: the prior; the two-steps-ahead kernel; the one-lane kernel; the depth-parallel kernel (launches
    // that leave the machine part empty) or the persistent kernel without the pipeline (the longest chains).
    if (auto_choice && c->tune_mode && c->pw_policy) {
        int cand[6], n = 0;
        auto add = [&](int f) {
            for (int i = 0; i < n; ++i)
                if (cand[i] == f) return;
            if (n < 6) cand[n++] = f;
        };
        add(fl);
        if (p2_ok && p2_variant) add((nothing_to_skip && variant != VR_VARIANT_VOLUME_MASK) || !can_skip ? 16 : 17);
        add(6);
        if (lut_ok && lut_lds <= 8u * 1024u) add(18);  // (the one-lane kernel with its slot arithmetic from LDS tables; larger tables cost it wavefronts per CU: C5 4.2 vs 3.4 ms)
        const bool dp_variant = variant != VR_VARIANT_ILLUSTRATIVE && variant != VR_VARIANT_LIGHT_INSHADER;
        if (!whole_frame && dp_variant) add((double)px_all * c->frames_in_flight * n_frames / ((double)c->n_cus * 4.0 * 5.0 * 64.0) >= 2.0 ? 11 : 10);
        else if (n_frames == 1 && (variant == VR_VARIANT_LIGHT || variant == VR_VARIANT_BASIC)) add(12);
        const unsigned long long shape = 0x9E3779B97F4A7C15ull * (((unsigned long long)variant << 56) ^ ((unsigned long long)world << 48) ^ ((unsigned long long)rank << 40) ^
                                                                 ((unsigned long long)c->W << 24) ^ ((unsigned long long)c->H << 8) ^ (packed ? 0x80ull : 0ull) ^
                                                                 ((unsigned long long)n_frames << 4) ^ (unsigned long long)c->frames_in_flight) | 1ull;
        const unsigned long long key = (shape ^ (c->brick_epoch * 0xD6E8FEB86659FD93ull) ^ (c->tf_epoch << 20) ^ ((unsigned long long)c->arith << 1) ^
                                        ((unsigned long long)c->layout_mode << 2)) | 1ull;
        const bool measurable = c->order_mode == 1 && c->h_span && c->h_end && !c->event_timing;
        fl = tune_pick(c, key, shape, cand, n, chain_known, measurable);
    }
    c->last_flavour = fl;

    if (c->layout_mode == 0 && fl != 2 && fl != 3) use_bricked();
    if (fl == 18 && P.vol[0].bricked) P.vol[0].lut = 1;  // (march_kernel fills the tables; every fetch of volume 0 goes through them)
    for (int i = 0; i < nvol; ++i)  // (a bricked copy is padded to whole bricks: a volume just below 4 GiB may cross the line)
        if (P.vol[i].bricked && (size_t)P.vol[i].brick_slab * (((unsigned)P.vol[i].nz + kVbM) >> kVbS) * 16 > 0xFFFFFFFFull) off32 = false;

    if (packed && !out) {
        size_t need = (size_t)P.n_tiles * kTile * kTile;
        if (need > c->tiles_cap) {
            if (c->d_tiles) (void)hipFree(c->d_tiles);
            c->d_tiles = nullptr;
            c->tiles_cap = 0;
            VR_HIP(c, hipMalloc(&c->d_tiles, (need ? need : 1) * sizeof(float4)));
            c->tiles_cap = need;
        }
        out = c->d_tiles;
    } else if (!out) {
        out = c->d_frame;
    }
    P.out = out;
    c->last_tiles = packed ? P.n_tiles : 0;

    if (frame_events) VR_HIP(c, hipEventRecord(c->tm.ev_begin, s));
    if (P.n_blocks > 0) {
        // flavours 2/3: LDS wave tiles (without / with skipping), lit shader only
        // (launches of several frames exist for the loop forms the default flavours use: plain, runs, depth-parallel)
        const bool wtb = (fl == 2 || fl == 3) && variant == VR_VARIANT_LIGHT && P.fragment_mode == 0 && c->arith == VR_ARITH_SEPARATE &&
                         n_frames == 1;
        const int leap_mode = n_frames > 1 ? (fl == 5 ? 0 : 3) : (fl == 4 ? 1 : (fl == 5 ? 0 : (fl == 9 ? 2 : 3)));
        const int dp = (fl == 7 || fl == 10) ? 4 : ((fl == 8 || fl == 11) ? 2 : 0);
        // gradients on the fly (one-lane kernel, lit shader): the volume's .rgb is verified to be the central difference of
        // its .a, so the eight corners are derived from the density plane -- same bits, a quarter of the footprint
        const bool otf = variant == VR_VARIANT_LIGHT && c->layout_mode == 2 && c->vol_grad_derived[0] && P.vol[0].dens != nullptr &&
                         !wtb && dp == 0;
        c->last_otf = otf;
        const bool dp_pipe = fl == 10 || fl == 11;  // ... with the next round's corner loads software-pipelined  // lanes per ray (vr_dp.h): 64 / 32 workgroups per tile
        // one wavefront per workgroup (launch order at wavefront granularity) -- except for the depth-parallel kernels on
        // large launches, where 4x the workgroups cost more at dispatch than the finer order gains (C2: 32 768 workgroups of
        // a 0.12 ms frame)
        const bool pw = fl == 12 || fl == 13 || fl == 16 || fl == 17;
        int wpb = wtb ? 4 : ((pw || fl == 15) ? 1 : c->waves_per_block);
        if (dp && P.n_tiles * (dp == 4 ? 256 : 128) > 16384) wpb = 4;
        dim3 block((unsigned)(64 * wpb));
        dim3 grid((unsigned)(dp ? P.n_tiles * (dp == 4 ? 256 : 128) / wpb : (P.n_tiles + 7) / 8 * 8 * (64 / wpb)));  // see map_pixel / map_pixel_dp
        if (n_frames > 1 && grid.x % 8u != 0) return fail(c, VR_ERR_INVALID_ARG, "vr_render: launch shape cannot carry several frames");
        // (record slot and order slot are both derived from order_seq, which advances only once a launch has really been
        // enqueued: a failed enqueue cannot shift one against the other)
        const int cb = (int)(c->order_seq % (unsigned long long)kInFlight);
        // the slot's previous launch (kInFlight launches ago, possibly on another stream) must have finished before its
        // record buffer is written again or re-allocated: this is what bounds the launches in flight to kInFlight
        if (c->slot_used[cb]) VR_HIP(c, hipEventSynchronize(c->slot_done[cb]));
        // A stream's wait for another stream's event costs the stream 5 us per launch even when the event completed long ago
        // (tools/ubench/stream_gap.hip): the two waits below are 10 of the 17 us between two march kernels of a one-at-a-time loop.
        // Waiting on the HOST instead (VR_EXP_HOST_ORDER_WAIT=1, callers with vr_hint_frames_in_flight <= 1) buys them back -- C3 0.4685 ->
        // 0.461 ms per frame, C2 0.113 -> 0.106, C1 0.042 -> 0.035; with the ONE wait a launch has left (below) 0.467 -> 0.461 -- but
        // leaves the host two or three launches ahead of the device instead of eight, and a host thread that wakes up a few
        // milliseconds late then idles the device (two of sixteen legs on a shared box: C1 0.040 -> 0.37 ms, C4 0.51 -> 0.64); taking
        // the complete order of eight launches ago keeps the queue deep and loses more to the stale order than the waits cost (+17 us
        // of span per C3 frame).  tools/experiments/s2o.sh, s2p.sh.  Off by default.
        const bool host_wait = c->host_order_wait != 0 && c->frames_in_flight <= 1;
        // ... and the sort that read those records.  Every sort runs on the one order stream, in the order of the launches: when this
        // launch waits for a YOUNGER sort anyway -- the one whose launch order it takes, below -- that wait covers this one, and a wait
        // for another stream's event less is 5 us less between two march kernels.  Waited for at once only when the record buffer is
        // re-allocated (the memset behind the allocation writes it).
        const vr_ctx::OrderSlot* slot_sort = nullptr;
        if (c->order_seq >= (unsigned long long)kInFlight) {
            const vr_ctx::OrderSlot& po = c->order_ring[(c->order_seq - kInFlight) % kOrderRing];
            if (po.valid && po.seq + kInFlight == c->order_seq) slot_sort = &po;
        }
        auto wait_slot_sort = [&]() -> hipError_t {
            if (!slot_sort) return hipSuccess;
            const hipError_t e = host_wait ? hipEventSynchronize(slot_sort->sorted) : hipStreamWaitEvent(s, slot_sort->sorted, 0);
            slot_sort = nullptr;
            return e;
        };
        // every frame of the launch has its own records; twice the space for one frame: a packet marched as two half packets
        // (vr_mixed.h) leaves its second half's record grid.x records further on
        const size_t n_records = (size_t)grid.x * (size_t)(n_frames > 1 ? n_frames : 2);
        if (n_records > c->block_counts_cap[cb]) {
            VR_HIP(c, wait_slot_sort());
            if (c->d_block_counts[cb]) (void)hipFree(c->d_block_counts[cb]);
            c->d_block_counts[cb] = nullptr;
            c->block_counts_cap[cb] = 0;
            VR_HIP(c, hipMalloc(&c->d_block_counts[cb], n_records * kBlockRecord * sizeof(unsigned long long)));
            VR_HIP(c, hipMemsetAsync(c->d_block_counts[cb], 0, n_records * kBlockRecord * sizeof(unsigned long long), s));
            c->block_counts_cap[cb] = n_records;
        }
        P.block_counts = c->d_block_counts[cb];
        c->cnt_buf = cb;
        // launch order: the most recent sort of a launch of the same shape that is three or four launches old (two or three
        // more than the frames the caller says it keeps in flight, if that is more: with short frames -- C2, 0.08 ms -- the
        // sort of the launch that finished one frame time ago is itself only just finishing) -- a younger one may still be waiting
        // for its launch to finish (the sorts run on a side stream behind their launches; waiting for one would put a bubble
        // into this stream, and with four frames in flight it would chain this launch behind the one three before it), an
        // older one's buffer may be recycled under this launch; ordered behind it by its event (long complete by then)
        const unsigned long long okey = ((unsigned long long)grid.x << 32) ^ ((unsigned long long)block.x << 20) ^
                                        ((unsigned long long)variant << 16) ^ ((unsigned long long)world << 8) ^ (unsigned long long)rank ^
                                        ((unsigned long long)(fl == 14 ? 14 : 0) << 44) ^ (packed ? 1ull << 63 : 0ull);
        // (not the flavour: the kernels that march one packet per wavefront -- 6, 12, 13, 16, 17 -- share the logical blocks, so a
        // launch order sorted behind one of them serves the others: the measured choice below tries them in turn on a live scene)
        P.order = nullptr;
        const unsigned* mixed_items = nullptr;  // fl 14: the item list of an earlier launch of this shape, once one exists
        unsigned mixed_grid = 0;
        const bool ordered = c->order_mode == 1 && !wtb && grid.x <= (unsigned)kOrderMaxBlocks && grid.x % 8u == 0;
        if (ordered) {
            const vr_ctx::OrderSlot* best = nullptr;
            const unsigned long long age = (unsigned long long)(c->frames_in_flight + 2 > 3 ? c->frames_in_flight + 2 : 3);
            for (const auto& o : c->order_ring)
                if (o.valid && o.key == okey && o.seq + age + 1 >= c->order_seq && o.seq + age <= c->order_seq && (!best || o.seq > best->seq))
                    best = &o;
            if (best) {
                VR_HIP(c, host_wait ? hipEventSynchronize(best->sorted) : hipStreamWaitEvent(s, best->sorted, 0));
                if (slot_sort && best->seq >= slot_sort->seq) slot_sort = nullptr;  // (covered: the order stream runs its sorts in order)
                P.order = best->buf;
                if (fl == 14 && best->has_items && c->h_items) {
                    const int bi = (int)(best - c->order_ring);
                    const unsigned n_pos = *(volatile unsigned*)&c->h_items[bi];
                    if (n_pos >= grid.x && n_pos <= 2u * grid.x && n_pos % 8u == 0) {
                        mixed_items = best->items;
                        mixed_grid = n_pos;
                        c->last_split = *(volatile unsigned*)&c->h_split[bi];
                    }
                }
            }
        }
        VR_HIP(c, wait_slot_sort());
        if (!mixed_items) c->last_split = 0;
        const int slot = (int)(c->ring.head % kRing);
        if (frame_events) VR_HIP(c, hipEventRecord(c->tm.ev_k0, s));
        // launches with a sort behind them are timed from their own records (order_blocks_kernel); events only otherwise
        const bool time_with_events = !(ordered && c->h_span) || c->event_timing;
        c->ring_events[slot] = time_with_events;
        if (c->h_span) c->h_span[slot] = 0;
        if (c->h_end) c->h_end[slot] = 0;
        if (time_with_events) VR_HIP(c, hipEventRecord(c->ring.k0[slot], s));
        {
            LaunchDesc L;
            L.variant = variant;
            L.off32 = off32;
            L.leap_mode = leap_mode;
            L.dp = dp;
            L.dp_pipe = dp_pipe;
            L.wtb = wtb;
            L.otf = otf;
            L.lt = fl == 15;
            L.pw = false;
            L.pw_ltf = false;
            L.pw_pipe = false;
            L.pw_p2 = false;
            L.pw_p2_skip = false;
            L.pw_p2_win = false;
            L.lds_bytes = (fl == 18 && P.vol[0].lut) ? lut_lds : 0u;
            L.queue = PwQueue{nullptr, 0u, 0u, 0u};
            L.mixed_items = nullptr;
            L.n_logical = 0;
            L.grid = grid;
            L.block = block;
            // frame f of the launch: every n_frames-th group of 8 workgroups (MarchBatch), its own uniforms, output and
            // records; the launch order (a heuristic of the shape) is shared
            static thread_local MarchBatch B;
            P.batch_n = (unsigned)n_frames;
            B.frame[0] = P;
            for (int f = 1; f < n_frames; ++f) {
                MarchParams& Pf = B.frame[f];
                Pf = P;
                fill_frame_params(Pf, batch_u[f]);
                Pf.out = (float4*)batch_out[f];
                Pf.block_counts = P.block_counts + (size_t)f * grid.x * kBlockRecord;
            }
            B.n_frames = (unsigned)n_frames;
            L.grid = dim3(grid.x * (unsigned)n_frames);
                if (pw) {
                // persistent wavefronts: `grid` stays the number of LOGICAL blocks (records, launch order); the launch itself is one
                // workgroup of 16 wavefronts per CU (fewer when there are fewer packets), TF slot 0 in LDS when its two tables
                // have one resolution and fit beside nothing else (R <= 8190: 128 KiB)
                // (march_p2_kernel: two corner buffers, 3 wavefronts per SIMD at most; with every ray sampling all the time two per
                // SIMD are faster -- the corner data in flight is many times the L1 either way: noisy air 2.13 -> 2.04 ms)
                const bool p2 = fl == 16 || fl == 17;
                // (the unlit shader's two buffers are 4-byte densities, 101 VGPRs: 4 wavefronts per SIMD -- C2 one frame at a time 0.121 ->
                // 0.113 ms, thin table 0.255 -> 0.239, four frames per launch 0.070 -> 0.061: tools/experiments/s2h.sh)
                unsigned pw_threads = fl == 17 ? (variant == VR_VARIANT_BASIC ? 1024u : 768u) : (fl == 16 ? 512u : 1024u);
                // (launches in flight: the same shape.  Two workgroups of 6 wavefronts do not share a CU -- the second one's wavefronts
                // would have to go 1-1-2-2 over the SIMDs where the dispatcher deals 2-2-1-1: measured 0.75 ms per C3 frame, what
                // one such workgroup per CU takes -- and two of 4 run at 8 wavefronts per CU: 0.63 against 0.54; three of 4, the same
                // 12 wavefronts per CU, take 0.79 ms one frame at a time and 0.62 in flight against 0.55 / 0.51: profiles/r04_p2_launch_shapes.txt)
                unsigned wg_per_cu = 1;
                if (p2 && c->p2_threads) pw_threads = c->p2_threads;  // (VR_EXP_P2_THREADS: 64 .. 768; .. 1024 for the unlit shader)
                if (p2 && variant != VR_VARIANT_BASIC && pw_threads > 768u) pw_threads = 768u;
                if (p2 && c->p2_wgs) wg_per_cu = c->p2_wgs;           // (VR_EXP_P2_WGS: workgroups per CU the grid is sized for)
                const unsigned per_wg = pw_threads / 64u;
                const unsigned items = grid.x * (unsigned)n_frames;
                const unsigned wgs = (items + per_wg - 1u) / per_wg;
                L.pw = true;
                L.pw_pipe = fl == 13;
                L.pw_p2 = p2;
                L.pw_p2_skip = fl == 17 && P.brick_dist != nullptr;
                L.pw_p2_win = p2 && (!off32 || c->p2_window != 0);
                L.pw_ltf = c->pw_ltf && c->tf[0].res_o == c->tf[0].res_c && c->tf[0].res_o + 2 <= 8192;
                L.lds_bytes = p2 ? p2_lds : (L.pw_ltf ? (unsigned)(c->tf[0].res_o + 2) * 16u : 0u);
                L.queue.heads = c->d_pw_heads + (size_t)cb * 8 * 64;
                L.queue.n_items = grid.x;
                L.queue.p2_window = c->p2_window;
                L.queue.dynamic = p2 && c->p2_dynq == 1 ? 1u : 0u;
                const unsigned max_wgs = (unsigned)c->n_cus * wg_per_cu;
                L.grid = dim3(wgs < max_wgs ? wgs : max_wgs);
                L.block = dim3(pw_threads);
                if (c->pw_heads_dirty[cb]) VR_HIP(c, hipMemsetAsync(L.queue.heads, 0, 8 * 64 * sizeof(unsigned), s));
                c->pw_heads_dirty[cb] = true;  // (until the sort that clears them behind this launch has really been enqueued: below)
            }
            if (mixed_items) {
                L.mixed_items = mixed_items;
                L.n_logical = (int)grid.x;
                L.grid = dim3(mixed_grid);
                L.block = dim3(64);
            }
            if (c->arith == VR_ARITH_FUSED) vrf::launch_march(L, s, B);
            else vr::launch_march(L, s, B);
        }
        VR_HIP(c, hipGetLastError());
        if (time_with_events) VR_HIP(c, hipEventRecord(c->ring.k1[slot], s));
        if (ordered) {
            vr_ctx::OrderSlot& o = c->order_ring[c->order_seq % kOrderRing];
            o.valid = false;
            if (grid.x > o.cap) {
                if (o.buf) (void)hipFree(o.buf);
                o.buf = nullptr;
                o.cap = 0;
                VR_HIP(c, hipMalloc(&o.buf, (size_t)grid.x * sizeof(unsigned)));
                o.cap = grid.x;
            }
            o.stream = s;
            o.key = okey;
            o.scene_key = ((unsigned long long)variant << 16) ^ ((unsigned long long)world << 8) ^ (unsigned long long)rank ^
                          (packed ? 1ull << 63 : 0ull) ^ ((unsigned long long)c->W << 40) ^ ((unsigned long long)c->H << 24);
            o.seq = c->order_seq;
            if (c->h_chain) c->h_chain[c->order_seq % kOrderRing] = 0;  // not known until this launch's sort has run
        }
        VR_HIP(c, hipEventRecord(c->slot_done[cb], s));
        c->slot_used[cb] = true;
        if (ordered) {
            vr_ctx::OrderSlot& o = c->order_ring[c->order_seq % kOrderRing];
            VR_HIP(c, hipStreamWaitEvent(c->order_stream, c->slot_done[cb], 0));
            hipLaunchKernelGGL(order_blocks_kernel, dim3(1), dim3(1024), 0, c->order_stream, c->d_block_counts[cb], (int)grid.x, o.buf,
                               c->h_chain ? c->h_chain + (c->order_seq % kOrderRing) : (unsigned*)nullptr,
                               (c->h_span && !time_with_events) ? c->h_span + slot : (unsigned long long*)nullptr,
                               pw ? c->d_pw_heads + (size_t)cb * 8 * 64 : (unsigned*)nullptr,
                               (c->h_span && c->h_end && !time_with_events) ? c->h_end + slot : (unsigned long long*)nullptr);
            VR_HIP(c, hipGetLastError());
            if (pw) c->pw_heads_dirty[cb] = false;  // (the sort zeroes the heads behind the launch: the slot's next user finds them clean)
            o.has_items = false;
#if VR_EXPERIMENTAL_FLAVOURS
            if (fl == 14 && c->h_items && c->h_split) {
                const int oi = (int)(c->order_seq % kOrderRing);
                if (2 * (size_t)grid.x > o.items_cap) {
                    if (o.items) (void)hipFree(o.items);
                    o.items = nullptr;
                    o.items_cap = 0;
                    VR_HIP(c, hipMalloc(&o.items, 2 * (size_t)grid.x * sizeof(unsigned)));
                    o.items_cap = 2 * (size_t)grid.x;
                }
                c->h_items[oi] = 0;
                c->h_split[oi] = 0;
                hipLaunchKernelGGL(build_items_kernel, dim3(1), dim3(1024), 0, c->order_stream, c->d_block_counts[cb], (int)grid.x, o.buf,
                                   (unsigned)c->split_pct, (unsigned)c->split_min, o.items, c->h_items + oi, c->h_split + oi);
                VR_HIP(c, hipGetLastError());
                o.has_items = true;
            }
#endif
            VR_HIP(c, hipEventRecord(o.sorted, c->order_stream));
            o.valid = true;
        }
        ++c->order_seq;
        if (frame_events) VR_HIP(c, hipEventRecord(c->tm.ev_k1, s));
        ++c->ring.head;
        c->cnt_blocks = (int)grid.x;
        c->cnt_offset = (size_t)(n_frames - 1) * grid.x * kBlockRecord;  // vr_last_counters: the LAST frame of the launch
    } else {
        c->cnt_blocks = 0;
        c->cnt_offset = 0;
        if (frame_events) {
            VR_HIP(c, hipEventRecord(c->tm.ev_k0, s));
            VR_HIP(c, hipEventRecord(c->tm.ev_k1, s));
        }
    }
    // the per-block counts are summed and copied to the host when somebody asks for them (fetch_counters)
    c->cnt_pending = true;
    if (frame_events) VR_HIP(c, hipEventRecord(c->tm.ev_end, s));
    c->tm.valid = frame_events;
    return VR_OK;
}

// Sums the per-block counts of the last launch into h_counters (blocks until that launch has finished).
int fetch_counters(vr_ctx* c)
{
    if (!c->cnt_pending) return VR_OK;
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    if (c->cnt_blocks > 0) {
        // the launch may have been enqueued on a stream of the caller's that no longer exists: wait for the event recorded
        // behind it (owned by the context; other launches in flight are not waited for), then use the context's own stream
        VR_HIP(c, hipEventSynchronize(c->slot_done[c->cnt_buf]));
        hipLaunchKernelGGL(sum_block_counts_kernel, dim3(1), dim3(256), 0, c->stream, c->d_block_counts[c->cnt_buf] + c->cnt_offset,
                           c->cnt_blocks, c->d_counters);
        VR_HIP(c, hipGetLastError());
        VR_HIP(c, hipMemcpyAsync(c->h_counters, c->d_counters, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                 c->stream));
        VR_HIP(c, hipStreamSynchronize(c->stream));
    } else {
        c->h_counters[0] = c->h_counters[1] = c->h_counters[2] = 0;
    }
    c->cnt_pending = false;
    return VR_OK;
}

// per-brick density / rgb maxima for the exact empty-space test (one pass over the volume; after every change)
int refresh_bricks(vr_ctx* c, int slot)
{
    (void)hipGetLastError();  // (see enqueue_render)
    const DevVolume& v = c->vol[slot];
    if (c->vol_bricks[slot]) (void)hipFree(c->vol_bricks[slot]);
    c->vol_bricks[slot] = nullptr;
    c->merged_stale = true;
    ++c->brick_epoch;
    const int bnx = (v.nx + kBrickCells - 1) >> kBrickShift, bny = (v.ny + kBrickCells - 1) >> kBrickShift, bnz = (v.nz + kBrickCells - 1) >> kBrickShift;
    const size_t nbricks = (size_t)bnx * bny * bnz;
    VR_HIP(c, hipMalloc(&c->vol_bricks[slot], nbricks * sizeof(float2)));
    hipLaunchKernelGGL(brick_max_kernel, dim3((unsigned)nbricks), dim3(64), 0, c->stream, v.data, v.nx, v.ny, v.nz, bnx, bny,
                       c->vol_bricks[slot]);
    VR_HIP(c, hipGetLastError());
    // scalar density plane + "is .rgb the central difference of .a?" (decides whether the lit shader may derive gradients)
    const size_t n = (size_t)v.nx * v.ny * v.nz;
    c->vol_grad_derived[slot] = false;
    if (n > c->vol_dens_cap[slot]) {
        if (c->vol_dens[slot]) (void)hipFree(c->vol_dens[slot]);
        c->vol_dens[slot] = nullptr;
        c->vol_dens_cap[slot] = 0;
        // (+ 4 floats of slack: the 16-byte row pieces of fetch_rgba_otf never start beyond the last voxel, but may end there)
        VR_HIP(c, hipMalloc(&c->vol_dens[slot], (n + 4) * sizeof(float)));
        VR_HIP(c, hipMemsetAsync(c->vol_dens[slot] + n, 0, 4 * sizeof(float), c->stream));
        c->vol_dens_cap[slot] = n;
    }
    hipLaunchKernelGGL(extract_density_kernel, dim3(4096), dim3(256), 0, c->stream, v.data, c->vol_dens[slot], n);
    VR_HIP(c, hipGetLastError());
    unsigned* d_flag = reinterpret_cast<unsigned*>(c->d_counters);
    VR_HIP(c, hipMemsetAsync(d_flag, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(verify_gradient_kernel, dim3((unsigned)((v.nx + 255) / 256), (unsigned)v.ny, (unsigned)v.nz), dim3(256), 0,
                       c->stream, v.data, c->vol_dens[slot], v.nx, v.ny, v.nz, d_flag);
    VR_HIP(c, hipGetLastError());
    unsigned flag = 1;
    VR_HIP(c, hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, c->stream));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    c->vol_grad_derived[slot] = flag == 0;
    c->vol[slot].dens = c->vol_dens[slot];
    {   // the bricked copy the march kernels gather from (DevVolume::bricked)
        const unsigned nbx = ((unsigned)v.nx + kVbM) >> kVbS, nby = ((unsigned)v.ny + kVbM) >> kVbS, nbz = ((unsigned)v.nz + kVbM) >> kVbS;
        const size_t slots = (size_t)nbx * nby * nbz * kVbN;
        if (slots > c->vol_bricked_cap[slot]) {
            if (c->vol_bricked[slot]) (void)hipFree(c->vol_bricked[slot]);
            if (c->vol_bdens[slot]) (void)hipFree(c->vol_bdens[slot]);
            c->vol_bricked[slot] = nullptr;
            c->vol_bdens[slot] = nullptr;
            c->vol_bricked_cap[slot] = 0;
            // (the bricked copies cost 20 B per voxel on top of the reference layout's 16 + 4: when they do not fit, the kernels gather
            // from the x-fastest arrays as with vr_set_volume_layout(3) -- slower, not an error)
            if (hipMalloc(&c->vol_bricked[slot], slots * sizeof(float4)) != hipSuccess) c->vol_bricked[slot] = nullptr;
            if (c->vol_bricked[slot] && hipMalloc(&c->vol_bdens[slot], slots * sizeof(float)) != hipSuccess) {
                (void)hipFree(c->vol_bricked[slot]);
                c->vol_bricked[slot] = nullptr;
                c->vol_bdens[slot] = nullptr;
            }
            (void)hipGetLastError();
            c->vol_bricked_cap[slot] = c->vol_bricked[slot] ? slots : 0;
        }
        if (!c->vol_bricked[slot]) {
            VR_HIP(c, hipStreamSynchronize(c->stream));
            return VR_OK;
        }
        hipLaunchKernelGGL(rebrick_kernel, dim3(8192), dim3(256), 0, c->stream, v.data, c->vol_bricked[slot], c->vol_bdens[slot], v.nx,
                           v.ny, v.nz, nbx, nby, slots);
        VR_HIP(c, hipGetLastError());
        VR_HIP(c, hipStreamSynchronize(c->stream));
    }
    return VR_OK;
}

int check_slot(vr_ctx* c, int slot, const char* who)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (slot < 0 || slot >= VR_MAX_VOLUMES) return fail(c, VR_ERR_INVALID_ARG, std::string(who) + ": bad slot");
    if (!c->vol[slot].data) return fail(c, VR_ERR_NOT_READY, std::string(who) + ": volume slot is empty");
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipDeviceSynchronize());  // asynchronous renders on the caller's streams may still read the slot
    (void)hipGetLastError();
    return VR_OK;
}

template <typename T>
int upload_raw(vr_ctx* c, int slot, const T* raw, uint16_t nx, uint16_t ny, uint16_t nz)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (slot < 0 || slot >= VR_MAX_VOLUMES) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload_raw: bad slot");
    if (!raw) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload_raw: data is NULL");
    if (nx == 0 || ny == 0 || nz == 0) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload_raw: empty volume");
    const size_t n = (size_t)nx * ny * nz;
    if (n > 0xFFFFFFFFull) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload_raw: more than 2^32 voxels");
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipDeviceSynchronize());  // asynchronous renders on the caller's streams may still read the slot
    (void)hipGetLastError();
    const size_t bytes = n * sizeof(float4);
    if (c->vol[slot].data && c->vol_bytes[slot] != bytes) {
        (void)hipFree(const_cast<float4*>(c->vol[slot].data));
        c->vol[slot] = DevVolume{};
        c->vol_bytes[slot] = 0;
    }
    float4* d = const_cast<float4*>(c->vol[slot].data);
    if (!d) VR_HIP(c, hipMalloc(&d, bytes));
    T* d_raw = nullptr;
    hipError_t e = hipMalloc(&d_raw, n * sizeof(T));
    if (e == hipSuccess) e = hipMemcpyAsync(d_raw, raw, n * sizeof(T), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((broadcast_raw_kernel<T>), dim3(2048), dim3(256), 0, c->stream, d_raw, d, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_raw) (void)hipFree(d_raw);
    if (e != hipSuccess) {
        if (!c->vol[slot].data) (void)hipFree(d);
        return fail(c, e == hipErrorOutOfMemory ? VR_ERR_OOM : VR_ERR_HIP,
                    std::string("vr_volume_upload_raw: ") + hipGetErrorString(e));
    }
    c->vol[slot].data = d;
    c->vol[slot].nx = nx;
    c->vol[slot].ny = ny;
    c->vol[slot].nz = nz;
    c->vol_bytes[slot] = bytes;
    return refresh_bricks(c, slot);
}

}  // namespace

extern "C" {

int vr_volume_upload_raw16(vr_ctx* c, int slot, const uint16_t* raw, uint16_t nx, uint16_t ny, uint16_t nz)
{
    return upload_raw(c, slot, raw, nx, ny, nz);
}
int vr_volume_upload_raw32(vr_ctx* c, int slot, const uint32_t* raw, uint16_t nx, uint16_t ny, uint16_t nz)
{
    return upload_raw(c, slot, raw, nx, ny, nz);
}

int vr_volume_normalize(vr_ctx* c, int slot, int normalization_value, int* used_value)
{
    int rc = check_slot(c, slot, "vr_volume_normalize");
    if (rc != VR_OK) return rc;
    float4* d = const_cast<float4*>(c->vol[slot].data);
    const size_t n = (size_t)c->vol[slot].nx * c->vol[slot].ny * c->vol[slot].nz;
    if (normalization_value == 0) {  // GetMaxNumber(): max of component [0], truncated
        unsigned* d_max = reinterpret_cast<unsigned*>(c->d_counters);
        VR_HIP(c, hipMemsetAsync(d_max, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(max_component_kernel, dim3(2048), dim3(256), 0, c->stream, d, n, 0, d_max);
        VR_HIP(c, hipGetLastError());
        unsigned bits = 0;
        VR_HIP(c, hipMemcpyAsync(&bits, d_max, sizeof bits, hipMemcpyDeviceToHost, c->stream));
        VR_HIP(c, hipStreamSynchronize(c->stream));
        float mx;
        std::memcpy(&mx, &bits, sizeof mx);
        normalization_value = (int)(size_t)mx;
    }
    if (used_value) *used_value = normalization_value;
    hipLaunchKernelGGL(normalize_kernel, dim3(2048), dim3(256), 0, c->stream, d, n, normalization_value);
    VR_HIP(c, hipGetLastError());
    return refresh_bricks(c, slot);
}

int vr_volume_precompute_gradient(vr_ctx* c, int slot, int norm_to_zero_one)
{
    int rc = check_slot(c, slot, "vr_volume_precompute_gradient");
    if (rc != VR_OK) return rc;
    float4* d = const_cast<float4*>(c->vol[slot].data);
    const DevVolume& v = c->vol[slot];
    const size_t n = (size_t)v.nx * v.ny * v.nz;
    unsigned* d_max = reinterpret_cast<unsigned*>(c->d_counters);
    VR_HIP(c, hipMemsetAsync(d_max, 0, sizeof(unsigned), c->stream));
    dim3 block(256), grid((unsigned)((v.nx + 255) / 256), (unsigned)v.ny, (unsigned)v.nz);
    hipLaunchKernelGGL(gradient_kernel, grid, block, 0, c->stream, d, v.nx, v.ny, v.nz, norm_to_zero_one ? 1 : 0, d_max);
    VR_HIP(c, hipGetLastError());
    if (norm_to_zero_one) {
        hipLaunchKernelGGL(scale_gradient_kernel, dim3(2048), dim3(256), 0, c->stream, d, n, d_max);
        VR_HIP(c, hipGetLastError());
    }
    return refresh_bricks(c, slot);
}

int vr_volume_download(vr_ctx* c, int slot, float* vec4_voxels)
{
    int rc = check_slot(c, slot, "vr_volume_download");
    if (rc != VR_OK) return rc;
    if (!vec4_voxels) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_download: destination is NULL");
    VR_HIP(c, hipMemcpy(vec4_voxels, c->vol[slot].data, c->vol_bytes[slot], hipMemcpyDeviceToHost));
    return VR_OK;
}

int vr_abi_version(void) { return VR_ABI_VERSION; }

const char* vr_last_error(const vr_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vr_create(vr_ctx** out, uint32_t width, uint32_t height, int device_id)
{
    if (!out) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_create: out is NULL");
    *out = nullptr;
    if (width == 0 || height == 0 || width > 32768 || height > 32768)
        return fail(nullptr, VR_ERR_INVALID_ARG, "vr_create: bad viewport size");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VR_ERR_HIP, std::string("vr_create: no HIP device available (") +
                                             (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                             "); this library has no CPU fallback");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_create: bad device_id");
    vr_ctx* c = new (std::nothrow) vr_ctx();
    if (!c) return fail(nullptr, VR_ERR_OOM, "vr_create: out of host memory");
    c->device = device_id;
    c->W = width;
    c->H = height;
    auto bail = [&](int code) {
        g_create_error = c->err;
        vr_destroy(c);
        return code;
    };
    int rc;
    auto hip_ok = [&](hipError_t he, const char* what) {
        if (he == hipSuccess) return true;
        c->err = std::string(what) + ": " + hipGetErrorString(he);
        return false;
    };
    if (!hip_ok(hipSetDevice(device_id), "hipSetDevice")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) return bail(VR_ERR_HIP);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->n_cus = cus;
    }
    // experiment knobs (A/B measurements; none changes any result)
    if (const char* e = getenv("VR_EXP_WAVES_PER_BLOCK")) c->waves_per_block = (atoi(e) == 4) ? 4 : 1;
    if (const char* e = getenv("VR_EXP_ONLY_TILE")) c->only_tile = atoi(e);
    if (const char* e = getenv("VR_EXP_PRIO")) c->prio_mode = atoi(e);
    if (const char* e = getenv("VR_EXP_FLAVOUR")) {
        const int f = atoi(e);
        if (f >= 0 && f <= 18 && (VR_EXPERIMENTAL_FLAVOURS || !(f == 2 || f == 3 || f == 4 || f == 5 || f == 9 || f == 14))) c->default_flavour = f;
    }
    if (const char* e = getenv("VR_EXP_XCD")) c->xcd_mode = atoi(e);
    if (const char* e = getenv("VR_EXP_PW_LTF")) c->pw_ltf = atoi(e) != 0;
    if (const char* e = getenv("VR_EXP_P2_THREADS")) {
        const int t = atoi(e);
        if (t >= 64 && t <= 1024 && t % 64 == 0) c->p2_threads = (unsigned)t;
    }
    if (const char* e = getenv("VR_EXP_P2_WGS")) c->p2_wgs = (unsigned)(atoi(e) > 0 && atoi(e) <= 8 ? atoi(e) : 0);
    if (const char* e = getenv("VR_EXP_P2_DYNQ")) c->p2_dynq = atoi(e);
    if (const char* e = getenv("VR_EXP_P2_WINDOW")) {  // records per gather window of march_p2_kernel (tests: the moving window on small volumes)
        const long long w = atoll(e);
        if (w > 0 && w <= 0x3fffffffll) c->p2_window = (unsigned)w;
    }
    if (const char* e = getenv("VR_EXP_PW_POLICY")) c->pw_policy = atoi(e);
    if (const char* e = getenv("VR_EXP_TUNE")) c->tune_mode = atoi(e);
    if (!hip_ok(hipMalloc(&c->d_pw_heads, (size_t)kInFlight * 8 * 64 * sizeof(unsigned)), "hipMalloc(queue heads)")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipMemset(c->d_pw_heads, 0, (size_t)kInFlight * 8 * 64 * sizeof(unsigned)), "hipMemset(queue heads)")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipEventCreate(&c->tm.ev_begin), "hipEventCreate")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipEventCreate(&c->tm.ev_k0), "hipEventCreate")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipEventCreate(&c->tm.ev_k1), "hipEventCreate")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipEventCreate(&c->tm.ev_end), "hipEventCreate")) return bail(VR_ERR_HIP);
    for (int i = 0; i < kRing; ++i)
        if (!hip_ok(hipEventCreate(&c->ring.k0[i]), "hipEventCreate") || !hip_ok(hipEventCreate(&c->ring.k1[i]), "hipEventCreate"))
            return bail(VR_ERR_HIP);
    for (int i = 0; i < kInFlight; ++i)
        if (!hip_ok(hipEventCreateWithFlags(&c->slot_done[i], hipEventDisableTiming), "hipEventCreate")) return bail(VR_ERR_HIP);
    for (auto& o : c->order_ring)
        if (!hip_ok(hipEventCreateWithFlags(&o.sorted, hipEventDisableTiming), "hipEventCreate")) return bail(VR_ERR_HIP);
    // The sorts run on a stream of their own, default priority.  The runtime deals streams onto a handful of hardware queues
    // per priority level, and a sort waits (a barrier in its queue) for a launch that is still running, so WHICH streams end up
    // sharing a queue with this one matters: measured on this box, a high- or low-priority sort stream lets a third frame in
    // flight overlap (a rank's eighth of C3: 0.142 -> 0.110 ms per frame, kernels alone) but costs the multi-GPU loop 50 us per
    // frame (0.24 -> 0.29 ms one frame at a time; with a high-priority sort stream its gather stream, high priority too, meets
    // the sorts' barriers), and the full C3 frame gains nothing from a third frame in flight either way (tools/exp_tiles.py,
    // tools/exp_queues).
    if (!hip_ok(hipStreamCreateWithFlags(&c->order_stream, hipStreamNonBlocking), "hipStreamCreate")) return bail(VR_ERR_HIP);
    if (hipHostMalloc((void**)&c->h_span, kRing * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess)
        std::memset(c->h_span, 0, kRing * sizeof(unsigned long long));
    else
        c->h_span = nullptr;  // (every launch is then timed with events)
    if (hipHostMalloc((void**)&c->h_items, 2 * kOrderRing * sizeof(unsigned), hipHostMallocDefault) == hipSuccess) {
        std::memset(c->h_items, 0, 2 * kOrderRing * sizeof(unsigned));
        c->h_split = c->h_items + kOrderRing;
    } else {
        c->h_items = c->h_split = nullptr;  // (flavour 14 then always marches with one lane per ray)
    }
    if (const char* e = getenv("VR_EXP_SPLIT_PCT")) c->split_pct = atoi(e) > 0 ? atoi(e) : 75;
    if (const char* e = getenv("VR_EXP_SPLIT_MIN")) c->split_min = atoi(e) > 0 ? atoi(e) : 64;
    if (hipHostMalloc((void**)&c->h_end, kRing * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess)
        std::memset(c->h_end, 0, kRing * sizeof(unsigned long long));
    else
        c->h_end = nullptr;  // (no measured kernel choice with launches in flight: the prior's pick stays)
    if (hipHostMalloc((void**)&c->h_chain, kOrderRing * sizeof(unsigned), hipHostMallocDefault) == hipSuccess)
        std::memset(c->h_chain, 0, kOrderRing * sizeof(unsigned));
    else
        c->h_chain = nullptr;  // (the choice of lanes per ray then goes by the launch size alone)
    if (const char* e = getenv("VR_EXP_ORDER")) c->order_mode = atoi(e);
    if (const char* e = getenv("VR_EXP_HOST_ORDER_WAIT")) c->host_order_wait = atoi(e) != 0;
    if (const char* e = getenv("VR_EXP_NO_ZSKIP")) c->zskip = atoi(e) == 0;
    if (const char* e = getenv("VR_EXP_EVENT_TIMING")) c->event_timing = atoi(e) != 0;
    if (!hip_ok(hipMalloc(&c->d_counters, 3 * sizeof(unsigned long long)), "hipMalloc(counters)")) return bail(VR_ERR_HIP);
    if (!hip_ok(hipHostMalloc((void**)&c->h_counters, 3 * sizeof(unsigned long long), hipHostMallocDefault),
                "hipHostMalloc"))
        return bail(VR_ERR_HIP);
    c->h_counters[0] = c->h_counters[1] = c->h_counters[2] = 0;
    rc = alloc_frame(c);
    if (rc != VR_OK) return bail(rc);
    *out = c;
    return VR_OK;
}

int vr_resize(vr_ctx* c, uint32_t width, uint32_t height)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (width == 0 || height == 0 || width > 32768 || height > 32768)
        return fail(c, VR_ERR_INVALID_ARG, "vr_resize: bad viewport size");
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();  // frames may be in flight on the caller's streams
    c->W = width;
    c->H = height;
    return alloc_frame(c);
}

void vr_destroy(vr_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();  // renders may be in flight on streams of the caller's
    for (int i = 0; i < VR_MAX_VOLUMES; ++i)
        if (c->vol[i].data) (void)hipFree(const_cast<float4*>(c->vol[i].data));
    for (int i = 0; i < VR_MAX_VOLUMES; ++i)
        if (c->vol_bricks[i]) (void)hipFree(c->vol_bricks[i]);
    for (int i = 0; i < VR_MAX_VOLUMES; ++i) {
        if (c->vol_dens[i]) (void)hipFree(c->vol_dens[i]);
        if (c->vol_bricked[i]) (void)hipFree(c->vol_bricked[i]);
        if (c->vol_bdens[i]) (void)hipFree(c->vol_bdens[i]);
    }
    if (c->merged_bricks) (void)hipFree(c->merged_bricks);
    if (c->brick_dist) (void)hipFree(c->brick_dist);
    for (int i = 0; i < VR_MAX_TFS; ++i) {
        if (c->tf_opacity[i]) (void)hipFree(c->tf_opacity[i]);
        if (c->tf_color[i]) (void)hipFree(c->tf_color[i]);
    }
    if (c->d_frame) (void)hipFree(c->d_frame);
    if (c->d_tiles) (void)hipFree(c->d_tiles);
    if (c->d_present) (void)hipFree(c->d_present);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_pw_heads) (void)hipFree(c->d_pw_heads);
    for (auto* b : c->d_block_counts)
        if (b) (void)hipFree(b);
    if (c->h_counters) (void)hipHostFree(c->h_counters);
    for (int i = 0; i < kRing; ++i) {
        if (c->ring.k0[i]) (void)hipEventDestroy(c->ring.k0[i]);
        if (c->ring.k1[i]) (void)hipEventDestroy(c->ring.k1[i]);
    }
    for (auto e : c->slot_done)
        if (e) (void)hipEventDestroy(e);
    for (auto& o : c->order_ring) {
        if (o.sorted) (void)hipEventDestroy(o.sorted);
        if (o.buf) (void)hipFree(o.buf);
        if (o.items) (void)hipFree(o.items);
    }
    if (c->order_stream) (void)hipStreamDestroy(c->order_stream);
    if (c->h_items) (void)hipHostFree(c->h_items);
    if (c->h_chain) (void)hipHostFree(c->h_chain);
    if (c->h_span) (void)hipHostFree(c->h_span);
    if (c->h_end) (void)hipHostFree(c->h_end);
    for (int k = 0; k < c->n_flight; ++k) (void)hipStreamDestroy(c->flight[k]);
    if (c->tm.ev_begin) (void)hipEventDestroy(c->tm.ev_begin);
    if (c->tm.ev_k0) (void)hipEventDestroy(c->tm.ev_k0);
    if (c->tm.ev_k1) (void)hipEventDestroy(c->tm.ev_k1);
    if (c->tm.ev_end) (void)hipEventDestroy(c->tm.ev_end);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int volume_upload_common(vr_ctx* c, int slot, const void* src, bool src_is_device, uint16_t nx, uint16_t ny,
                                uint16_t nz)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (slot < 0 || slot >= VR_MAX_VOLUMES) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload: bad slot");
    if (!src) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload: data is NULL");
    if (nx == 0 || ny == 0 || nz == 0) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload: empty volume");
    unsigned long long voxels = (unsigned long long)nx * ny * nz;
    if (voxels > 0xFFFFFFFFull) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_upload: more than 2^32 voxels");
    size_t bytes = (size_t)voxels * sizeof(float4);
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipDeviceSynchronize());  // asynchronous renders on the caller's streams may still read the slot
    if (c->vol[slot].data && c->vol_bytes[slot] != bytes) {
        (void)hipFree(const_cast<float4*>(c->vol[slot].data));
        c->vol[slot] = DevVolume{};
        c->vol_bytes[slot] = 0;
    }
    float4* d = const_cast<float4*>(c->vol[slot].data);
    if (!d) VR_HIP(c, hipMalloc(&d, bytes));
    hipError_t e = hipMemcpyAsync(d, src, bytes, src_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        if (!c->vol[slot].data) (void)hipFree(d);
        return fail(c, VR_ERR_HIP, std::string("vr_volume_upload: copy failed: ") + hipGetErrorString(e));
    }
    c->vol[slot].data = d;
    c->vol[slot].nx = nx;
    c->vol[slot].ny = ny;
    c->vol[slot].nz = nz;
    c->vol_bytes[slot] = bytes;
    return refresh_bricks(c, slot);
}

int vr_volume_upload(vr_ctx* c, int slot, const float* vec4_voxels, uint16_t nx, uint16_t ny, uint16_t nz)
{
    return volume_upload_common(c, slot, vec4_voxels, false, nx, ny, nz);
}

int vr_volume_upload_device(vr_ctx* c, int slot, const void* d_vec4_voxels, uint16_t nx, uint16_t ny, uint16_t nz)
{
    return volume_upload_common(c, slot, d_vec4_voxels, true, nx, ny, nz);
}

static int tf_upload_one(vr_ctx* c, int slot, const float* table, uint32_t R, bool is_color)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (slot < 0 || slot >= VR_MAX_TFS) return fail(c, VR_ERR_INVALID_ARG, "vr_tf_upload: bad slot");
    if (!table) return fail(c, VR_ERR_INVALID_ARG, "vr_tf_upload: table is NULL");
    if (R == 0 || R > (1u << 24)) return fail(c, VR_ERR_INVALID_ARG, "vr_tf_upload: bad resolution");
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipDeviceSynchronize());  // asynchronous renders on the caller's streams may still read the table
    ++c->tf_epoch;
    if (is_color) {
        if (c->tf[slot].res_c != (int)R) {
            if (c->tf_color[slot]) (void)hipFree(c->tf_color[slot]);
            c->tf_color[slot] = nullptr;
            c->tf[slot].color = nullptr;
            c->tf[slot].res_c = 0;
            VR_HIP(c, hipMalloc(&c->tf_color[slot], ((size_t)R + 2) * sizeof(float4)));
        }
        // device layout (DevTF): the first and the last texel once more at either end
        VR_HIP(c, hipMemcpyAsync(c->tf_color[slot] + 1, table, R * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        VR_HIP(c, hipMemcpyAsync(c->tf_color[slot], table, sizeof(float4), hipMemcpyHostToDevice, c->stream));
        VR_HIP(c, hipMemcpyAsync(c->tf_color[slot] + R + 1, table + 4 * ((size_t)R - 1), sizeof(float4), hipMemcpyHostToDevice,
                                 c->stream));
        VR_HIP(c, hipStreamSynchronize(c->stream));
        c->tf[slot].color = c->tf_color[slot];
        c->tf[slot].res_c = (int)R;
        c->tf_color_finite[slot] = all_finite(table, (int)(4 * R));
    } else {
        if (c->tf[slot].res_o != (int)R) {
            if (c->tf_opacity[slot]) (void)hipFree(c->tf_opacity[slot]);
            c->tf_opacity[slot] = nullptr;
            c->tf[slot].opacity = nullptr;
            c->tf[slot].res_o = 0;
            VR_HIP(c, hipMalloc(&c->tf_opacity[slot], ((size_t)R + 2) * sizeof(float)));
        }
        VR_HIP(c, hipMemcpyAsync(c->tf_opacity[slot] + 1, table, R * sizeof(float), hipMemcpyHostToDevice, c->stream));
        VR_HIP(c, hipMemcpyAsync(c->tf_opacity[slot], table, sizeof(float), hipMemcpyHostToDevice, c->stream));
        VR_HIP(c, hipMemcpyAsync(c->tf_opacity[slot] + R + 1, table + ((size_t)R - 1), sizeof(float), hipMemcpyHostToDevice,
                                 c->stream));
        VR_HIP(c, hipStreamSynchronize(c->stream));
        c->tf[slot].opacity = c->tf_opacity[slot];
        c->tf[slot].res_o = (int)R;
        int z = -1;
        c->tf_opacity_finite[slot] = all_finite(table, (int)R);
        if (c->tf_opacity_finite[slot])
            while (z + 1 < (int)R && table[z + 1] == 0.0f) ++z;
        c->tf_zero_prefix[slot] = z;
    }
    return VR_OK;
}

int vr_tf_upload_opacity(vr_ctx* c, int slot, const float* opacity, uint32_t R) { return tf_upload_one(c, slot, opacity, R, false); }
int vr_tf_upload_color(vr_ctx* c, int slot, const float* color_rgba, uint32_t R) { return tf_upload_one(c, slot, color_rgba, R, true); }

int vr_tf_upload(vr_ctx* c, int slot, const float* opacity, const float* color_rgba, uint32_t R)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!opacity || !color_rgba) return fail(c, VR_ERR_INVALID_ARG, "vr_tf_upload: table is NULL");
    int rc = tf_upload_one(c, slot, opacity, R, false);
    return rc != VR_OK ? rc : tf_upload_one(c, slot, color_rgba, R, true);
}

int vr_set_uniforms(vr_ctx* c, const vr_uniforms* u)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!u) return fail(c, VR_ERR_INVALID_ARG, "vr_set_uniforms: uniforms is NULL");
    if (!is_identity(u->model))
        return fail(c, VR_ERR_UNSUPPORTED,
                    "vr_set_uniforms: model matrix must be the identity (the reference never uploads another one, "
                    "App/src/Application.cpp:489-492)");
    c->u = *u;
    c->have_uniforms = true;
    return VR_OK;
}

int vr_render(vr_ctx* c, int variant)
{
    if (!c) return VR_ERR_INVALID_ARG;
    int rc = enqueue_render(c, variant, 0, 1, false, nullptr, c->stream, true);
    if (rc != VR_OK) return rc;
    VR_HIP(c, hipStreamSynchronize(c->stream));
    return fetch_counters(c);
}

int vr_tile_count(const vr_ctx* c, int rank, int world)
{
    if (!c || world < 1 || rank < 0 || rank >= world) return VR_ERR_INVALID_ARG;
    return tile_count(c, rank, world);
}

int vr_render_tiles(vr_ctx* c, int variant, int rank, int world)
{
    if (!c) return VR_ERR_INVALID_ARG;
    int rc = enqueue_render(c, variant, rank, world, true, nullptr, c->stream, true);
    if (rc != VR_OK) return rc;
    VR_HIP(c, hipStreamSynchronize(c->stream));
    return fetch_counters(c);
}

int vr_render_async(vr_ctx* c, int variant, void* d_frame, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_render(c, variant, 0, 1, false, (float4*)d_frame, s, false);
}

int vr_render_tiles_async(vr_ctx* c, int variant, int rank, int world, void* d_tiles, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_render(c, variant, rank, world, true, (float4*)d_tiles, s, false);
}

int vr_render_batch_async(vr_ctx* c, int variant, int n_frames, const vr_uniforms* uniforms, void* const* d_frames, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!uniforms || !d_frames) return fail(c, VR_ERR_INVALID_ARG, "vr_render_batch_async: uniforms / buffers are NULL");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_render(c, variant, 0, 1, false, nullptr, s, false, n_frames, uniforms, d_frames);
}

int vr_render_tiles_batch_async(vr_ctx* c, int variant, int rank, int world, int n_frames, const vr_uniforms* uniforms,
                                void* const* d_tiles, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!uniforms || !d_tiles) return fail(c, VR_ERR_INVALID_ARG, "vr_render_tiles_batch_async: uniforms / buffers are NULL");
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return enqueue_render(c, variant, rank, world, true, nullptr, s, false, n_frames, uniforms, d_tiles);
}

int vr_unpack_tiles_strided_async(vr_ctx* c, const void* d_gathered, int world, int rank_stride_tiles, void* d_frame, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!d_gathered || world < 1) return fail(c, VR_ERR_INVALID_ARG, "vr_unpack_tiles_async: bad arguments");
    const int tpr = tile_count(c, 0, world);
    if (rank_stride_tiles < tpr) return fail(c, VR_ERR_INVALID_ARG, "vr_unpack_tiles_strided_async: stride smaller than a segment");
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    float4* frame = d_frame ? (float4*)d_frame : c->d_frame;
    dim3 block(64, 4), grid((c->W + 63) / 64, (c->H + 3) / 4);
    hipLaunchKernelGGL(unpack_tiles_kernel, grid, block, 0, s, (const float4*)d_gathered, frame, (int)c->W, (int)c->H,
                       tiles_x_of(c), world, rank_stride_tiles);
    VR_HIP(c, hipGetLastError());
    return VR_OK;
}

int vr_unpack_tiles_async(vr_ctx* c, const void* d_gathered, int world, void* d_frame, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (world < 1) return fail(c, VR_ERR_INVALID_ARG, "vr_unpack_tiles_async: bad arguments");
    return vr_unpack_tiles_strided_async(c, d_gathered, world, tile_count(c, 0, world), d_frame, stream);
}

int vr_present_async(vr_ctx* c, const void* d_frame, void* d_bgra8, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!d_bgra8) return fail(c, VR_ERR_INVALID_ARG, "vr_present_async: destination is NULL");
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t n = (size_t)c->W * c->H;
    hipLaunchKernelGGL(present_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       d_frame ? (const float4*)d_frame : c->d_frame, (uint32_t*)d_bgra8, (int)n);
    VR_HIP(c, hipGetLastError());
    return VR_OK;
}

int vr_present_tiles_async(vr_ctx* c, const void* d_gathered, int world, int rank_stride_tiles, void* d_bgra8, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!d_gathered || !d_bgra8 || world < 1) return fail(c, VR_ERR_INVALID_ARG, "vr_present_tiles_async: bad arguments");
    const int tpr = tile_count(c, 0, world);
    if (rank_stride_tiles <= 0) rank_stride_tiles = tpr;
    if (rank_stride_tiles < tpr) return fail(c, VR_ERR_INVALID_ARG, "vr_present_tiles_async: stride smaller than a segment");
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    dim3 block(64, 4), grid((c->W + 63) / 64, (c->H + 3) / 4);
    hipLaunchKernelGGL(present_tiles_kernel, grid, block, 0, s, (const float4*)d_gathered, (uint32_t*)d_bgra8, (int)c->W, (int)c->H,
                       tiles_x_of(c), world, rank_stride_tiles);
    VR_HIP(c, hipGetLastError());
    return VR_OK;
}

int vr_present_packed_async(vr_ctx* c, const void* d_tiles_rgba, int n_tiles, void* d_tiles_bgra8, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!d_tiles_rgba || !d_tiles_bgra8 || n_tiles < 0) return fail(c, VR_ERR_INVALID_ARG, "vr_present_packed_async: bad arguments");
    if (n_tiles == 0) return VR_OK;
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t n = (size_t)n_tiles * kTile * kTile;
    if (n > 0x7fffffffull) return fail(c, VR_ERR_INVALID_ARG, "vr_present_packed_async: too many tiles");
    hipLaunchKernelGGL(present_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float4*)d_tiles_rgba, (uint32_t*)d_tiles_bgra8, (int)n);
    VR_HIP(c, hipGetLastError());
    return VR_OK;
}

int vr_unpack_tiles_bgra8_async(vr_ctx* c, const void* d_gathered_bgra8, int world, int rank_stride_tiles, void* d_bgra8, void* stream)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!d_gathered_bgra8 || !d_bgra8 || world < 1) return fail(c, VR_ERR_INVALID_ARG, "vr_unpack_tiles_bgra8_async: bad arguments");
    const int tpr = tile_count(c, 0, world);
    if (rank_stride_tiles <= 0) rank_stride_tiles = tpr;
    if (rank_stride_tiles < tpr) return fail(c, VR_ERR_INVALID_ARG, "vr_unpack_tiles_bgra8_async: stride smaller than a segment");
    VR_HIP(c, hipSetDevice(c->device));
    (void)hipGetLastError();
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    dim3 block(64, 4), grid((c->W + 63) / 64, (c->H + 3) / 4);
    hipLaunchKernelGGL(unpack_tiles_u32_kernel, grid, block, 0, s, (const uint32_t*)d_gathered_bgra8, (uint32_t*)d_bgra8, (int)c->W, (int)c->H,
                       tiles_x_of(c), world, rank_stride_tiles);
    VR_HIP(c, hipGetLastError());
    return VR_OK;
}

int vr_download(vr_ctx* c, float* frag_rgba, uint8_t* present_bgra8, uint64_t* composited_samples)
{
    if (!c) return VR_ERR_INVALID_ARG;
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    size_t n = (size_t)c->W * c->H;
    if (frag_rgba) VR_HIP(c, hipMemcpy(frag_rgba, c->d_frame, n * sizeof(float4), hipMemcpyDeviceToHost));
    if (present_bgra8) {
        (void)hipGetLastError();
        hipLaunchKernelGGL(present_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_frame,
                           c->d_present, (int)n);
        VR_HIP(c, hipGetLastError());
        VR_HIP(c, hipStreamSynchronize(c->stream));
        VR_HIP(c, hipMemcpy(present_bgra8, c->d_present, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    if (composited_samples) {
        int rc = fetch_counters(c);
        if (rc != VR_OK) return rc;
        *composited_samples = c->h_counters[0];
    }
    return VR_OK;
}

int vr_download_tiles(vr_ctx* c, float* tiles_rgba, uint64_t* composited_samples)
{
    if (!c) return VR_ERR_INVALID_ARG;
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    if (tiles_rgba && c->last_tiles > 0)
        VR_HIP(c, hipMemcpy(tiles_rgba, c->d_tiles, (size_t)c->last_tiles * kTile * kTile * sizeof(float4),
                            hipMemcpyDeviceToHost));
    if (composited_samples) {
        int rc = fetch_counters(c);
        if (rc != VR_OK) return rc;
        *composited_samples = c->h_counters[0];
    }
    return VR_OK;
}

int vr_last_timing(vr_ctx* c, float* kernel_ms, float* total_ms)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!c->tm.valid) return fail(c, VR_ERR_NOT_READY, "vr_last_timing: no vr_render / vr_render_tiles since the context was created or an *_async call");
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipEventSynchronize(c->tm.ev_end));
    float k = 0.0f, t = 0.0f;
    VR_HIP(c, hipEventElapsedTime(&k, c->tm.ev_k0, c->tm.ev_k1));
    VR_HIP(c, hipEventElapsedTime(&t, c->tm.ev_begin, c->tm.ev_end));
    if (kernel_ms) *kernel_ms = k;
    if (total_ms) *total_ms = t;
    return VR_OK;
}

int vr_kernel_times(vr_ctx* c, float* out_ms, int capacity)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (!out_ms || capacity < 0) return fail(c, VR_ERR_INVALID_ARG, "vr_kernel_times: bad arguments");
    VR_HIP(c, hipSetDevice(c->device));
    long long have = c->ring.head < kRing ? c->ring.head : kRing;
    int n = (int)(have < capacity ? have : capacity);
    bool synced = false;
    for (int i = 0; i < n; ++i) {
        int slot = (int)((c->ring.head - n + i) % kRing);
        if (!c->ring_events[slot]) {  // from the launch's records, written by the sort that runs behind it
            if (!synced) VR_HIP(c, hipStreamSynchronize(c->order_stream));
            synced = true;
            const unsigned long long ticks = *(volatile unsigned long long*)&c->h_span[slot];
            out_ms[i] = ticks ? (float)((double)(ticks - 1) * 1.0e-5) : 0.0f;
            continue;
        }
        VR_HIP(c, hipEventSynchronize(c->ring.k1[slot]));
        VR_HIP(c, hipEventElapsedTime(&out_ms[i], c->ring.k0[slot], c->ring.k1[slot]));
    }
    return n;
}

int vr_set_kernel_timing(vr_ctx* c, int mode)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (mode != VR_TIMING_RECORDS && mode != VR_TIMING_EVENTS) return fail(c, VR_ERR_INVALID_ARG, "vr_set_kernel_timing: bad mode");
    c->event_timing = mode == VR_TIMING_EVENTS;
    return VR_OK;
}

int vr_reset_kernel_times(vr_ctx* c)
{
    if (!c) return VR_ERR_INVALID_ARG;
    // (sorts of earlier launches still report their launch's duration into the ring: let them finish first)
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipStreamSynchronize(c->order_stream));
    c->ring.head = 0;
    return VR_OK;
}

void* vr_frame_device_ptr(vr_ctx* c) { return c ? (void*)c->d_frame : nullptr; }

int vr_viewport(const vr_ctx* c, uint32_t* width, uint32_t* height, int* device_id)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (width) *width = c->W;
    if (height) *height = c->H;
    if (device_id) *device_id = c->device;
    return VR_OK;
}

int vr_last_covered_pixels(vr_ctx* c, uint64_t* covered)
{
    if (!c || !covered) return VR_ERR_INVALID_ARG;
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    int rc = fetch_counters(c);
    if (rc != VR_OK) return rc;
    *covered = c->h_counters[1];
    return VR_OK;
}

int vr_last_counters(vr_ctx* c, uint64_t out[3])
{
    if (!c || !out) return VR_ERR_INVALID_ARG;
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipStreamSynchronize(c->stream));
    int rc = fetch_counters(c);
    if (rc != VR_OK) return rc;
    out[0] = c->h_counters[0];
    out[1] = c->h_counters[1];
    out[2] = c->h_counters[2];
    return VR_OK;
}

int vr_last_block_trace(vr_ctx* c, uint64_t* out, int capacity)
{
    if (!c || capacity < 0 || (capacity > 0 && !out)) return VR_ERR_INVALID_ARG;
    VR_HIP(c, hipSetDevice(c->device));
    VR_HIP(c, hipDeviceSynchronize());
    const int n = c->cnt_blocks < capacity ? c->cnt_blocks : capacity;
    if (n > 0) {
        const unsigned long long* src = c->d_block_counts[c->cnt_buf] + c->cnt_offset;
        VR_HIP(c, hipMemcpy(out, src, (size_t)n * kBlockRecord * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        // packets marched as two half packets (vr_mixed.h): add the second half's record (cnt_offset == 0 for such launches)
        bool any = false;
        for (int i = 0; i < n && !any; ++i) any = (out[(size_t)i * kBlockRecord + 5] & kRecSplit) != 0;
        if (any && c->cnt_offset == 0) {
            std::vector<unsigned long long> second((size_t)n * kBlockRecord);
            VR_HIP(c, hipMemcpy(second.data(), src + (size_t)c->cnt_blocks * kBlockRecord, second.size() * sizeof(unsigned long long),
                                hipMemcpyDeviceToHost));
            for (int i = 0; i < n; ++i) {
                uint64_t* a = out + (size_t)i * kBlockRecord;
                if (!(a[5] & kRecSplit)) continue;
                const unsigned long long* b = second.data() + (size_t)i * kBlockRecord;
                a[0] += b[0];
                a[1] += b[1];
                a[2] += b[2];
                a[3] = b[3] < a[3] ? b[3] : a[3];
                a[4] = b[4] > a[4] ? b[4] : a[4];
                const unsigned long long ca = a[5] >> 40, cb2 = b[5] >> 40;
                a[5] = (a[5] & ((1ull << 40) - 1)) | ((ca > cb2 ? ca : cb2) << 40);
            }
        }
    }
    return c->cnt_blocks;
}

int vr_last_kernel_flavour(vr_ctx* c)
{
    if (!c) return VR_ERR_INVALID_ARG;
    return c->last_flavour;
}

int vr_experimental_flavours(void) { return VR_EXPERIMENTAL_FLAVOURS ? 1 : 0; }

int vr_last_split_packets(vr_ctx* c)
{
    if (!c) return VR_ERR_INVALID_ARG;
    return (int)c->last_split;
}

// Event-timed span of one 150 us single-wavefront spin on a and, if b is given, a second one on b right behind it.
static float spin_span_ms(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1)
{
    const unsigned long long ticks = 15000;  // 150 us of the 100 MHz clock
    (void)hipStreamSynchronize(a);
    if (b) (void)hipStreamSynchronize(b);
    (void)hipEventRecord(e0, a);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, ticks, (unsigned*)nullptr);
    if (b) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, ticks, (unsigned*)nullptr);
    (void)hipEventRecord(e1, b ? b : a);
    (void)hipStreamSynchronize(a);
    if (b) (void)hipStreamSynchronize(b);
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0f;
    return ms;
}

// true if kernels enqueued on a and b run concurrently: two spins take about as long as one (`one_ms`, measured on this
// box a moment ago -- launch overheads differ between boxes and runs, a fixed limit misjudged them now and then), not twice
static bool streams_overlap(hipStream_t a, hipStream_t b, hipEvent_t e0, hipEvent_t e1, float one_ms)
{
    for (int attempt = 0; attempt < 2; ++attempt) {  // a hiccup (page fault, clock ramp) must not cost a stream
        const float ms = spin_span_ms(a, b, e0, e1);
        if (getenv("VR_DEBUG_STREAMS")) fprintf(stderr, "[vr_stream] pair %p %p: %.3f ms (one spin %.3f ms)\n", (void*)a, (void*)b, ms, one_ms);
        if (ms < 0.0f || one_ms <= 0.0f) return true;  // cannot tell: assume the best
        if (ms < one_ms + 0.075f) return true;
    }
    return false;
}

void* vr_stream(vr_ctx* c, int index)
{
    if (!c || index < 0 || index >= kStreams) return nullptr;
    if (c->n_flight == 0) {
        if (hipSetDevice(c->device) != hipSuccess) return nullptr;
        (void)hipGetLastError();
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return nullptr;
        // candidates are created one by one; one is kept if it overlaps with every stream kept so far (at most 12 tries).
        // Rejected candidates stay alive until the search is over: the runtime hands a stream that is destroyed and created
        // again the very same hardware queue, and the search would try one queue twelve times.
        float one_ms = -1.0f;
        hipStream_t rejected[12];
        int n_rejected = 0;
        for (int tries = 0; tries < 12 && c->n_flight < kStreams; ++tries) {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
            if (c->n_flight == 0) {  // the yardstick: one spin alone (the second measurement: the first one warms up)
                (void)spin_span_ms(s, nullptr, e0, e1);
                one_ms = spin_span_ms(s, nullptr, e0, e1);
            }
            bool ok = true;
            for (int k = 0; k < c->n_flight && ok; ++k) ok = streams_overlap(c->flight[k], s, e0, e1, one_ms);
            // ... and with the stream of the launch-order sorts, whose barriers (a sort waits for its launch) would hold back
            // the launches of a render stream that shares its queue
            if (ok && c->order_stream) ok = streams_overlap(c->order_stream, s, e0, e1, one_ms);
            if (ok) c->flight[c->n_flight++] = s;
            else rejected[n_rejected++] = s;  // shares a hardware queue with a kept one
        }
        for (int k = 0; k < n_rejected; ++k) (void)hipStreamDestroy(rejected[k]);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipGetLastError();
        if (c->n_flight == 0) return nullptr;
    }
    return (void*)c->flight[index % c->n_flight];
}

int vr_hint_frames_in_flight(vr_ctx* c, int frames)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (frames < 1 || frames > kStreams) return fail(c, VR_ERR_INVALID_ARG, "vr_hint_frames_in_flight: 1 .. 4");
    c->frames_in_flight = frames;
    return VR_OK;
}

int vr_set_arithmetic(vr_ctx* c, int mode)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (mode != VR_ARITH_SEPARATE && mode != VR_ARITH_FUSED) return fail(c, VR_ERR_INVALID_ARG, "vr_set_arithmetic: unknown mode");
    c->arith = mode;
    return VR_OK;
}

int vr_set_volume_layout(vr_ctx* c, int mode)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (mode < 0 || mode > 3) return fail(c, VR_ERR_INVALID_ARG, "vr_set_volume_layout: unknown mode");
    if (!VR_EXPERIMENTAL_FLAVOURS && mode == 2)
        return fail(c, VR_ERR_UNSUPPORTED, "vr_set_volume_layout: layout 2 is compiled with -DVR_EXPERIMENTAL_FLAVOURS=1 only");
    c->layout_mode = mode;
    return VR_OK;
}

int vr_volume_layout(vr_ctx* c, int slot, int* flags)
{
    if (!c || !flags) return VR_ERR_INVALID_ARG;
    if (slot < 0 || slot >= VR_MAX_VOLUMES) return fail(c, VR_ERR_INVALID_ARG, "vr_volume_layout: bad slot");
    if (!c->vol[slot].data) return fail(c, VR_ERR_NOT_READY, "vr_volume_layout: volume slot is empty");
    *flags = (c->vol_dens[slot] ? 1 : 0) | (c->vol_grad_derived[slot] ? 2 : 0) | (c->last_otf ? 4 : 0) |
             ((c->vol_bricked[slot] && c->layout_mode == 0) ? 8 : 0);
    return VR_OK;
}

int vr_kernel_choice(vr_ctx* c, int flavours[6], float ms_per_launch[6], int* chosen)
{
    if (!c) return VR_ERR_INVALID_ARG;
    const vr_ctx::Tune* t = nullptr;
    for (const auto& e : c->tune)
        if (e.key != 0 && e.used != 0 && (!t || e.used > t->used)) t = &e;
    if (chosen) *chosen = t ? t->choice : -1;
    if (!t) return 0;
    for (int i = 0; i < 6; ++i) {
        if (flavours) flavours[i] = i < t->n ? t->cand[i] : 0;
        if (ms_per_launch) ms_per_launch[i] = i < t->n ? t->cost[i] : 0.0f;
    }
    return t->n;
}

int vr_set_kernel_flavour(vr_ctx* c, int flavour)
{
    if (!c) return VR_ERR_INVALID_ARG;
    if (flavour < 0 || flavour > 18) return fail(c, VR_ERR_INVALID_ARG, "vr_set_kernel_flavour: unknown flavour");
    if (!VR_EXPERIMENTAL_FLAVOURS && (flavour == 2 || flavour == 3 || flavour == 4 || flavour == 5 || flavour == 9 || flavour == 14))
        return fail(c, VR_ERR_UNSUPPORTED, "vr_set_kernel_flavour: flavours 2, 3, 4, 5, 9 and 14 are compiled with -DVR_EXPERIMENTAL_FLAVOURS=1 only");
    c->flavour = flavour;
    return VR_OK;
}

}  // extern "C"
