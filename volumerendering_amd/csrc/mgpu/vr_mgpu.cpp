// vr_mgpu.cpp -- the multi-GPU frame loop behind include/vr_mgpu.h: image tiles interleaved over the ranks, one RCCL
// gather per frame (every peer has a direct xGMI link to the root: the transfers run side by side), un-permute on the
// root, frames pipelined two deep.  C++ host code on the HIP runtime + RCCL; the rendering itself goes through the C ABI
// of libvr_hip.so (vr_render_tiles_async / vr_unpack_tiles_async).  No reference counterpart: the reference is a
// single-device application (WebgpuLib/src/Base/GraphicsContext.h:35-42).
#include "../../../include/vr_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

constexpr int kTilePx = 64;
constexpr int kSlots = 4;  // buffer sets; vr_mgpu::slots of them are used (launches in flight, default 2, VR_MGPU_SLOTS)
constexpr int kBatch = 4;  // frames one launch may carry (vr_render_tiles_batch_async)

thread_local std::string g_create_error;

struct Rank {
    int device = 0;
    int rank = 0;
    vr_ctx* ctx = nullptr;
    bool own_ctx = false;
    ncclComm_t comm = nullptr;
    hipStream_t s_render[kSlots] = {}, s_comm = nullptr;
    hipEvent_t ev_render[kSlots] = {}, ev_gathered[kSlots] = {};
    bool used[kSlots] = {};
    float* tiles[kSlots] = {};     // this rank's packed tiles (segment padded to the largest rank's), batch_cap frames of them
    float* gathered[kSlots] = {};  // root: world x batch_cap segments
    float* frame[kSlots] = {};     // root: batch_cap assembled W x H frames, back to back
    uint32_t* present[kSlots] = {};  // root, VR_MGPU_OUT_PRESENT: batch_cap presented BGRA8 frames, back to back
    uint32_t* ptiles[kSlots] = {};   // this rank's tiles presented where they were rendered (VR_MGPU_OUT_PRESENT alone: 4 B per pixel to gather)
    uint32_t* pgathered[kSlots] = {};  // root: world x batch_cap such segments
    // stage timeline (vr_mgpu_set_stage_timing): render start, render done, segments gathered, output written
    hipEvent_t tm_t0[kSlots] = {}, tm_r[kSlots] = {}, tm_g[kSlots] = {}, tm_d[kSlots] = {};
    bool tm_valid[kSlots] = {};
    unsigned long long* d_red = nullptr;  // 4 x 8 bytes for vr_mgpu_reduce
};

}  // namespace

struct vr_mgpu {
    int world = 1;
    uint32_t W = 0, H = 0;
    size_t seg_floats = 0;  // floats per rank segment = max tiles per rank * 64 * 64 * 4
    std::vector<Rank> r;    // the ranks this process drives
    int slots = 2;          // launches in flight (buffer sets in use)
    bool one_stream = false;  // vr_mgpu_set_frames_in_flight(1): march, gather and output pass of every launch on ONE stream per rank
    int batch_cap = 1;      // frames per launch the buffer sets are sized for (grown on demand by vr_mgpu_frames_async)
    int exp_share = 1;      // experiment (VR_MGPU_EXP_SHARE=N, world of one only): render and gather only rank 0's share of an
                            // N-rank partition -- the timeline of one rank of an N-GPU run on a one-GPU box; frames are incomplete
    int output = VR_MGPU_OUT_FRAME;  // what the root produces from the gathered segments (vr_mgpu_set_output)
    bool gather_presented = true;    // VR_MGPU_OUT_PRESENT alone: every rank presents its own tiles, BGRA8 tiles are gathered (4 B per
                                     // pixel over xGMI instead of 16); VR_MGPU_GATHER_FLOAT=1: float tiles, presented on the root (A/B)
    bool stage_timing = false;       // timing events around the three stages of every launch (vr_mgpu_set_stage_timing)
    bool failed = false;             // a collective or an allocation failed half-way: the ranks' queues no longer match up,
                                     // every later call returns the stored error
    unsigned long long frame_no = 0;
    unsigned long long enqueued = 0;   // device work items enqueued so far (a failed call that enqueued none leaves the ranks in step)
    std::string err, backend;
};

namespace {

int fail(vr_mgpu* m, int code, const std::string& msg)
{
    if (m) m->err = msg;
    else g_create_error = msg;
    return code;
}

#define MG_HIP(m, call)                                                                                                \
    do {                                                                                                               \
        hipError_t e__ = (call);                                                                                       \
        if (e__ != hipSuccess)                                                                                         \
            return fail((m), e__ == hipErrorOutOfMemory ? VR_ERR_OOM : VR_ERR_HIP,                                     \
                        std::string(#call) + " (vr_mgpu.cpp:" + std::to_string(__LINE__) + "): " + hipGetErrorString(e__)); \
    } while (0)
#define MG_NCCL(m, call)                                                                                               \
    do {                                                                                                               \
        ncclResult_t r__ = (call);                                                                                     \
        if (r__ != ncclSuccess)                                                                                        \
            return fail((m), VR_ERR_HIP, std::string(#call) + " (vr_mgpu.cpp:" + std::to_string(__LINE__) + "): " +     \
                                             ncclGetErrorString(r__));                                                 \
    } while (0)
#define MG_VR(m, rk, call)                                                                                             \
    do {                                                                                                               \
        int rc__ = (call);                                                                                             \
        if (rc__ != VR_OK) return fail((m), rc__, std::string(#call) + ": " + vr_last_error((rk).ctx));                \
    } while (0)

// (re)allocates the buffer sets of one rank for m->batch_cap frames per launch; the rank must be idle
int alloc_buffers(vr_mgpu* m, Rank& k)
{
    MG_HIP(m, hipSetDevice(k.device));
    const int gather_world = m->exp_share > 1 ? m->exp_share : m->world;
    const size_t seg = (m->seg_floats ? m->seg_floats : 4) * (size_t)m->batch_cap;
    const size_t frame_floats = (size_t)m->W * m->H * 4 * (size_t)m->batch_cap;
    for (int b = 0; b < m->slots; ++b) {
        if (k.tiles[b]) (void)hipFree(k.tiles[b]);
        if (k.gathered[b]) (void)hipFree(k.gathered[b]);
        if (k.frame[b]) (void)hipFree(k.frame[b]);
        if (k.present[b]) (void)hipFree(k.present[b]);
        if (k.ptiles[b]) (void)hipFree(k.ptiles[b]);
        if (k.pgathered[b]) (void)hipFree(k.pgathered[b]);
        k.tiles[b] = k.gathered[b] = k.frame[b] = nullptr;
        k.present[b] = k.ptiles[b] = k.pgathered[b] = nullptr;
        k.used[b] = false;
        k.tm_valid[b] = false;
        MG_HIP(m, hipMalloc(&k.tiles[b], seg * sizeof(float)));
        MG_HIP(m, hipMemset(k.tiles[b], 0, seg * sizeof(float)));
        MG_HIP(m, hipMalloc(&k.ptiles[b], seg));  // (one BGRA8 word per pixel = a quarter of the float segment's bytes)
        MG_HIP(m, hipMemset(k.ptiles[b], 0, seg));
        if (k.rank == 0) {
            MG_HIP(m, hipMalloc(&k.pgathered[b], seg * (size_t)gather_world));
            MG_HIP(m, hipMemset(k.pgathered[b], 0, seg * (size_t)gather_world));
            MG_HIP(m, hipMalloc(&k.gathered[b], seg * (size_t)gather_world * sizeof(float)));
            MG_HIP(m, hipMemset(k.gathered[b], 0, seg * (size_t)gather_world * sizeof(float)));
            MG_HIP(m, hipMalloc(&k.frame[b], frame_floats * sizeof(float)));
            MG_HIP(m, hipMemset(k.frame[b], 0, frame_floats * sizeof(float)));
            MG_HIP(m, hipMalloc(&k.present[b], frame_floats));  // (4 bytes per pixel = one float's worth)
            MG_HIP(m, hipMemset(k.present[b], 0, frame_floats));
        }
    }
    MG_HIP(m, hipDeviceSynchronize());
    return VR_OK;
}

// streams, events and buffers of one rank (its device must be current)
int setup_rank(vr_mgpu* m, Rank& k)
{
    MG_HIP(m, hipSetDevice(k.device));
    for (int b = 0; b < m->slots; ++b) {
        // the context's own streams: probed to really run side by side (two arbitrary streams may share a hardware queue)
        k.s_render[b] = (hipStream_t)vr_stream(k.ctx, b);
        if (!k.s_render[b]) return fail(m, VR_ERR_HIP, "vr_stream: no render stream available");
        MG_HIP(m, hipEventCreateWithFlags(&k.ev_render[b], hipEventDisableTiming));
        MG_HIP(m, hipEventCreateWithFlags(&k.ev_gathered[b], hipEventDisableTiming));
        MG_HIP(m, hipEventCreate(&k.tm_t0[b]));
        MG_HIP(m, hipEventCreate(&k.tm_r[b]));
        MG_HIP(m, hipEventCreate(&k.tm_g[b]));
        MG_HIP(m, hipEventCreate(&k.tm_d[b]));
    }
    int rc = alloc_buffers(m, k);
    if (rc != VR_OK) return rc;
    {   // gather + un-permute are short and every frame waits for them: highest priority the device offers
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        MG_HIP(m, hipStreamCreateWithPriority(&k.s_comm, hipStreamNonBlocking, greatest));
    }
    MG_HIP(m, hipMalloc(&k.d_red, 4 * sizeof(unsigned long long)));
    MG_HIP(m, hipDeviceSynchronize());
    // the loop keeps `slots` launches in flight per rank: the context's kernel choice is made for that
    (void)vr_hint_frames_in_flight(k.ctx, m->one_stream ? 1 : (m->slots > 4 ? 4 : m->slots));
    return VR_OK;
}

void read_knobs(vr_mgpu* m)
{
    if (const char* e = getenv("VR_MGPU_SLOTS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= kSlots) m->slots = v;
    }
    if (const char* e = getenv("VR_MGPU_GATHER_FLOAT")) m->gather_presented = atoi(e) == 0;
    if (const char* e = getenv("VR_MGPU_EXP_SHARE")) {
        const int v = atoi(e);
        if (v > 1 && m->world == 1) m->exp_share = v;
    }
}

void describe_backend(vr_mgpu* m)
{
    int v = 0;
    (void)ncclGetVersion(&v);
    m->backend = "RCCL " + std::to_string(v / 10000) + "." + std::to_string((v / 100) % 100) + "." + std::to_string(v % 100) +
                 " ncclGather over xGMI, root = rank 0, " + std::to_string(m->slots) + " frames in flight" +
                 (m->exp_share > 1 ? " [EXPERIMENT: one rank's share of " + std::to_string(m->exp_share) + "]" : "");
}

}  // namespace

extern "C" {

int vr_mgpu_unique_id(void* id128)
{
    if (!id128) return VR_ERR_INVALID_ARG;
    static_assert(sizeof(ncclUniqueId) == VR_MGPU_ID_BYTES, "id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, VR_ERR_HIP, std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(id128, &id, sizeof id);
    return VR_OK;
}

const char* vr_mgpu_last_error(const vr_mgpu* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int vr_mgpu_create(vr_mgpu** out, vr_ctx* ctx, int rank, int world, const void* id128)
{
    if (!out) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_mgpu_create: out is NULL");
    *out = nullptr;
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_mgpu_create: bad arguments");
    vr_mgpu* m = new (std::nothrow) vr_mgpu();
    if (!m) return fail(nullptr, VR_ERR_OOM, "vr_mgpu_create: out of host memory");
    auto bail = [&](int code) {
        g_create_error = m->err;
        vr_mgpu_destroy(m);
        return code;
    };
    m->world = world;
    read_knobs(m);
    int dev = 0;
    if (vr_viewport(ctx, &m->W, &m->H, &dev) != VR_OK) return bail(fail(m, VR_ERR_INVALID_ARG, "vr_mgpu_create: bad context"));
    m->seg_floats = (size_t)vr_tile_count(ctx, 0, m->exp_share > 1 ? m->exp_share : world) * kTilePx * kTilePx * 4;
    m->r.resize(1);
    Rank& k = m->r[0];
    k.device = dev;
    k.rank = rank;
    k.ctx = ctx;
    int rc = setup_rank(m, k);
    if (rc != VR_OK) return bail(rc);
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t nr = ncclCommInitRank(&k.comm, world, id, rank);
    if (nr != ncclSuccess) return bail(fail(m, VR_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(nr)));
    describe_backend(m);
    *out = m;
    return VR_OK;
}

int vr_mgpu_create_local(vr_mgpu** out, uint32_t width, uint32_t height, const int* device_ids, int n_devices)
{
    if (!out) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_mgpu_create_local: out is NULL");
    *out = nullptr;
    if (!device_ids || n_devices < 1) return fail(nullptr, VR_ERR_INVALID_ARG, "vr_mgpu_create_local: bad arguments");
    vr_mgpu* m = new (std::nothrow) vr_mgpu();
    if (!m) return fail(nullptr, VR_ERR_OOM, "vr_mgpu_create_local: out of host memory");
    auto bail = [&](int code) {
        g_create_error = m->err;
        vr_mgpu_destroy(m);
        return code;
    };
    m->world = n_devices;
    read_knobs(m);
    m->exp_share = 1;
    m->W = width;
    m->H = height;
    m->r.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) {
        Rank& k = m->r[(size_t)i];
        k.device = device_ids[i];
        k.rank = i;
        k.own_ctx = true;
        if (vr_create(&k.ctx, width, height, device_ids[i]) != VR_OK) {
            k.ctx = nullptr;
            return bail(fail(m, VR_ERR_HIP, std::string("vr_create on device ") + std::to_string(device_ids[i]) + ": " + vr_last_error(nullptr)));
        }
    }
    m->seg_floats = (size_t)vr_tile_count(m->r[0].ctx, 0, n_devices) * kTilePx * kTilePx * 4;
    for (auto& k : m->r) {
        int rc = setup_rank(m, k);
        if (rc != VR_OK) return bail(rc);
    }
    std::vector<ncclComm_t> comms((size_t)n_devices);
    ncclResult_t nr = ncclCommInitAll(comms.data(), n_devices, device_ids);
    if (nr != ncclSuccess) return bail(fail(m, VR_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr)));
    for (int i = 0; i < n_devices; ++i) m->r[(size_t)i].comm = comms[(size_t)i];
    describe_backend(m);
    *out = m;
    return VR_OK;
}

void vr_mgpu_destroy(vr_mgpu* m)
{
    if (!m) return;
    for (auto& k : m->r) {
        if (!k.ctx) continue;  // never set up (vr_create failed for it): nothing to release, and its device id may be invalid
        (void)hipSetDevice(k.device);
        (void)hipDeviceSynchronize();
        if (k.comm) (void)ncclCommDestroy(k.comm);
        for (int b = 0; b < kSlots; ++b) {
            // (s_render[] belong to the context)
            if (k.ev_render[b]) (void)hipEventDestroy(k.ev_render[b]);
            if (k.ev_gathered[b]) (void)hipEventDestroy(k.ev_gathered[b]);
            for (hipEvent_t e : {k.tm_t0[b], k.tm_r[b], k.tm_g[b], k.tm_d[b]})
                if (e) (void)hipEventDestroy(e);
            if (k.present[b]) (void)hipFree(k.present[b]);
            if (k.ptiles[b]) (void)hipFree(k.ptiles[b]);
            if (k.pgathered[b]) (void)hipFree(k.pgathered[b]);
            if (k.tiles[b]) (void)hipFree(k.tiles[b]);
            if (k.gathered[b]) (void)hipFree(k.gathered[b]);
            if (k.frame[b]) (void)hipFree(k.frame[b]);
        }
        if (k.s_comm) (void)hipStreamDestroy(k.s_comm);
        if (k.d_red) (void)hipFree(k.d_red);
        if (k.own_ctx && k.ctx) vr_destroy(k.ctx);
    }
    delete m;
}

int vr_mgpu_world(const vr_mgpu* m) { return m ? m->world : VR_ERR_INVALID_ARG; }
int vr_mgpu_local_ranks(const vr_mgpu* m) { return m ? (int)m->r.size() : VR_ERR_INVALID_ARG; }
vr_ctx* vr_mgpu_context(vr_mgpu* m, int local_rank)
{
    return (m && local_rank >= 0 && local_rank < (int)m->r.size()) ? m->r[(size_t)local_rank].ctx : nullptr;
}
const char* vr_mgpu_backend(const vr_mgpu* m) { return m ? m->backend.c_str() : ""; }

// One launch per rank: n_frames frames (uniforms == nullptr: ONE frame with the uniforms set on each context).
// A failure after the first rank's work has been enqueued leaves the ranks' streams out of step (some have the collective,
// some do not): the handle is marked failed, an open group is closed, and every later call returns the stored error.
static int enqueue_frames_impl(vr_mgpu* m, int variant, int n_frames, const vr_uniforms* uniforms, bool& group_open)
{
    const int b = (int)(m->frame_no % (unsigned long long)m->slots);
    const int part_world = m->exp_share > 1 ? m->exp_share : m->world;
    const size_t seg = m->seg_floats ? m->seg_floats : 4;
    const int tpr = (int)(seg / ((size_t)kTilePx * kTilePx * 4));
    const bool presented = m->output == VR_MGPU_OUT_PRESENT && m->gather_presented && tpr > 0;
    for (auto& k : m->r)
        if (!k.tiles[b] || !k.ptiles[b] || (k.rank == 0 && !k.pgathered[b]) || (k.rank == 0 && (!k.gathered[b] || !k.frame[b] || !k.present[b])))
            return fail(m, VR_ERR_NOT_READY, "vr_mgpu: buffer set not allocated (an earlier allocation failed)");
    // 1. every local rank renders its tiles into buffer set b, behind the gather that last read that tile buffer (not behind
    //    the output pass that followed it: that one reads the gather buffer and writes the frames, which only the
    //    communication stream touches, in order)
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        // (one frame at a time: the communication stream carries the march as well -- stream order instead of three event waits
        // across streams per frame, and the host need not wait between two frames to keep them apart)
        const hipStream_t s_render = m->one_stream ? k.s_comm : k.s_render[b];
        if (k.used[b] && !m->one_stream) MG_HIP(m, hipStreamWaitEvent(s_render, k.ev_gathered[b], 0));
        if (m->stage_timing) MG_HIP(m, hipEventRecord(k.tm_t0[b], s_render));
        if (uniforms) {
            void* ptrs[kBatch];
            for (int f = 0; f < n_frames; ++f) ptrs[f] = k.tiles[b] + (size_t)f * seg;
            MG_VR(m, k, vr_render_tiles_batch_async(k.ctx, variant, k.rank, part_world, n_frames, uniforms, ptrs, s_render));
        } else {
            MG_VR(m, k, vr_render_tiles_async(k.ctx, variant, k.rank, part_world, k.tiles[b], s_render));
        }
        ++m->enqueued;
        // the presented frame alone is wanted: the output merge runs here, on the rank's own tiles (elementwise: the same bytes as
        // presenting the assembled frame), and the gather below moves BGRA8 words
        if (presented) MG_VR(m, k, vr_present_packed_async(k.ctx, k.tiles[b], n_frames * tpr, k.ptiles[b], s_render));
        if (m->stage_timing) MG_HIP(m, hipEventRecord(k.tm_r[b], s_render));
        if (!m->one_stream) {
            MG_HIP(m, hipEventRecord(k.ev_render[b], s_render));
            MG_HIP(m, hipStreamWaitEvent(k.s_comm, k.ev_render[b], 0));
        }
    }
    // 2. one gather: rank r's n_frames segments land back to back at gathered[b] + r * n_frames * seg on the root.  Every
    //    rank's communication stream carries the launches in the same order, so the collectives match up across ranks.
    if (m->r.size() > 1) {
        MG_NCCL(m, ncclGroupStart());
        group_open = true;
    }
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        if (presented)
            MG_NCCL(m, ncclGather(k.ptiles[b], k.rank == 0 ? k.pgathered[b] : nullptr, seg / 4 * (size_t)n_frames, ncclUint32, 0, k.comm, k.s_comm));
        else
            MG_NCCL(m, ncclGather(k.tiles[b], k.rank == 0 ? k.gathered[b] : nullptr, seg * (size_t)n_frames, ncclFloat, 0, k.comm, k.s_comm));
    }
    if (m->r.size() > 1) {
        group_open = false;
        MG_NCCL(m, ncclGroupEnd());
    }
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        MG_HIP(m, hipEventRecord(k.ev_gathered[b], k.s_comm));
        if (m->stage_timing) MG_HIP(m, hipEventRecord(k.tm_g[b], k.s_comm));
    }
    // 3. the root turns the segments into what its consumer reads: the assembled float frames (un-permute: 16 B read + 16 B
    //    written per pixel) and / or the presented BGRA8 frames straight from the tile-major segments (16 + 4 B per pixel,
    //    the output merge of App/src/renderer/PipelineBuilder.cpp:142-154 reading through the permutation)
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        if (k.rank == 0)
            for (int f = 0; f < n_frames; ++f) {
                if (presented) {
                    MG_VR(m, k, vr_unpack_tiles_bgra8_async(k.ctx, k.pgathered[b] + (size_t)f * (seg / 4), part_world, n_frames * tpr,
                                                            k.present[b] + (size_t)f * m->W * m->H, k.s_comm));
                    continue;
                }
                if (m->output & VR_MGPU_OUT_FRAME)
                    MG_VR(m, k, vr_unpack_tiles_strided_async(k.ctx, k.gathered[b] + (size_t)f * seg, part_world, n_frames * tpr,
                                                              k.frame[b] + (size_t)f * m->W * m->H * 4, k.s_comm));
                if (m->output & VR_MGPU_OUT_PRESENT)
                    MG_VR(m, k, vr_present_tiles_async(k.ctx, k.gathered[b] + (size_t)f * seg, part_world, n_frames * tpr,
                                                       k.present[b] + (size_t)f * m->W * m->H, k.s_comm));
            }
        if (m->stage_timing) {
            MG_HIP(m, hipEventRecord(k.tm_d[b], k.s_comm));
            k.tm_valid[b] = true;
        }
        k.used[b] = true;
    }
    ++m->frame_no;
    return b;
}

static int enqueue_frames(vr_mgpu* m, int variant, int n_frames, const vr_uniforms* uniforms)
{
    if (m->failed) return VR_ERR_HIP;  // (m->err still holds what went wrong)
    bool group_open = false;
    const unsigned long long before = m->enqueued;
    const int rc = enqueue_frames_impl(m, variant, n_frames, uniforms, group_open);
    if (rc < 0) {
        if (group_open) (void)ncclGroupEnd();
        // (an argument error on the first rank, before anything was enqueued, leaves every queue as it was: the handle stays usable)
        if (m->enqueued != before || group_open) m->failed = true;
    }
    return rc;
}

int vr_mgpu_frame_async(vr_mgpu* m, int variant)
{
    if (!m) return VR_ERR_INVALID_ARG;
    return enqueue_frames(m, variant, 1, nullptr);
}

int vr_mgpu_frames_async(vr_mgpu* m, int variant, int n_frames, const vr_uniforms* uniforms)
{
    if (!m) return VR_ERR_INVALID_ARG;
    if (!uniforms || n_frames < 1 || n_frames > kBatch) return fail(m, VR_ERR_INVALID_ARG, "vr_mgpu_frames_async: 1 .. 4 frames with their uniforms");
    if (n_frames > m->batch_cap) {  // first launch of this size: drain, then size the buffer sets for it (collective by construction:
                                    // every rank is given the same n_frames)
        int rc = vr_mgpu_wait(m);
        if (rc != VR_OK) return rc;
        const int old_cap = m->batch_cap;
        m->batch_cap = n_frames;  // (alloc_buffers sizes the sets by it)
        for (auto& k : m->r) {
            rc = alloc_buffers(m, k);
            if (rc != VR_OK) {
                // some ranks have the new size, some have no buffers at all: enqueue_frames refuses null buffer sets, and the
                // handle is poisoned so that no rank renders into a set of the wrong size
                m->batch_cap = old_cap;
                m->failed = true;
                return rc;
            }
        }
    }
    return enqueue_frames(m, variant, n_frames, uniforms);
}

int vr_mgpu_wait(vr_mgpu* m)
{
    if (!m) return VR_ERR_INVALID_ARG;
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        for (int b = 0; b < m->slots; ++b) MG_HIP(m, hipStreamSynchronize(k.s_render[b]));
        MG_HIP(m, hipStreamSynchronize(k.s_comm));
    }
    return VR_OK;
}

void* vr_mgpu_batch_frame_device_ptr(vr_mgpu* m, int which, int frame_in_launch)
{
    if (!m || which < 0 || which >= m->slots || frame_in_launch < 0 || frame_in_launch >= m->batch_cap) return nullptr;
    for (auto& k : m->r)
        if (k.rank == 0) return k.frame[which] + (size_t)frame_in_launch * m->W * m->H * 4;
    return nullptr;
}

void* vr_mgpu_frame_device_ptr(vr_mgpu* m, int which) { return vr_mgpu_batch_frame_device_ptr(m, which, 0); }

int vr_mgpu_download_batch_frame(vr_mgpu* m, int which, int frame_in_launch, float* frag_rgba)
{
    if (!m || !frag_rgba || which < 0 || which >= m->slots || frame_in_launch < 0 || frame_in_launch >= m->batch_cap)
        return VR_ERR_INVALID_ARG;
    int rc = vr_mgpu_wait(m);
    if (rc != VR_OK) return rc;
    for (auto& k : m->r)
        if (k.rank == 0) {
            MG_HIP(m, hipSetDevice(k.device));
            MG_HIP(m, hipMemcpy(frag_rgba, k.frame[which] + (size_t)frame_in_launch * m->W * m->H * 4,
                                (size_t)m->W * m->H * 4 * sizeof(float), hipMemcpyDeviceToHost));
            return VR_OK;
        }
    return fail(m, VR_ERR_NOT_READY, "vr_mgpu_download: this process does not drive the root rank");
}

int vr_mgpu_download(vr_mgpu* m, int which, float* frag_rgba) { return vr_mgpu_download_batch_frame(m, which, 0, frag_rgba); }

int vr_mgpu_reduce(vr_mgpu* m, uint64_t counters_sum[3], double local_value, double* max_value)
{
    if (!m) return VR_ERR_INVALID_ARG;
    if (m->failed) return VR_ERR_HIP;
    int rc = vr_mgpu_wait(m);
    if (rc != VR_OK) return rc;
    for (auto& k : m->r) {
        uint64_t c[3] = {0, 0, 0};
        MG_VR(m, k, vr_last_counters(k.ctx, c));
        unsigned long long h[4] = {c[0], c[1], c[2], 0};
        std::memcpy(&h[3], &local_value, sizeof(double));
        MG_HIP(m, hipSetDevice(k.device));
        MG_HIP(m, hipMemcpy(k.d_red, h, sizeof h, hipMemcpyHostToDevice));
    }
    {
        // (a failure between group start and end must still close the group, and leaves the ranks out of step)
        auto body = [&]() -> int {
            for (auto& k : m->r) {
                MG_HIP(m, hipSetDevice(k.device));
                MG_NCCL(m, ncclAllReduce(k.d_red, k.d_red, 3, ncclUint64, ncclSum, k.comm, k.s_comm));
                MG_NCCL(m, ncclAllReduce(k.d_red + 3, k.d_red + 3, 1, ncclFloat64, ncclMax, k.comm, k.s_comm));
            }
            return VR_OK;
        };
        if (m->r.size() > 1) MG_NCCL(m, ncclGroupStart());
        const int brc = body();
        if (m->r.size() > 1) {
            const ncclResult_t er = ncclGroupEnd();
            if (brc == VR_OK && er != ncclSuccess) {
                m->failed = true;
                return fail(m, VR_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(er));
            }
        }
        if (brc != VR_OK) {
            m->failed = true;
            return brc;
        }
    }
    Rank& k0 = m->r[0];
    MG_HIP(m, hipSetDevice(k0.device));
    MG_HIP(m, hipStreamSynchronize(k0.s_comm));
    unsigned long long h[4];
    MG_HIP(m, hipMemcpy(h, k0.d_red, sizeof h, hipMemcpyDeviceToHost));
    for (auto& k : m->r) {
        MG_HIP(m, hipSetDevice(k.device));
        MG_HIP(m, hipStreamSynchronize(k.s_comm));
    }
    if (counters_sum) {
        counters_sum[0] = h[0];
        counters_sum[1] = h[1];
        counters_sum[2] = h[2];
    }
    if (max_value) std::memcpy(max_value, &h[3], sizeof(double));
    return VR_OK;
}

int vr_mgpu_comm_count(const vr_mgpu* m)
{
    if (!m || m->r.empty() || !m->r[0].comm) return VR_ERR_INVALID_ARG;
    int n = 0;
    if (ncclCommCount(m->r[0].comm, &n) != ncclSuccess) return VR_ERR_HIP;
    return n;
}

int vr_mgpu_device(const vr_mgpu* m, int local_rank)
{
    if (!m || local_rank < 0 || local_rank >= (int)m->r.size()) return VR_ERR_INVALID_ARG;
    return m->r[(size_t)local_rank].device;
}

int vr_mgpu_set_output(vr_mgpu* m, int output)
{
    if (!m) return VR_ERR_INVALID_ARG;
    if ((output & ~(VR_MGPU_OUT_FRAME | VR_MGPU_OUT_PRESENT)) != 0 || output == 0)
        return fail(m, VR_ERR_INVALID_ARG, "vr_mgpu_set_output: VR_MGPU_OUT_FRAME and / or VR_MGPU_OUT_PRESENT");
    m->output = output;
    return VR_OK;
}

int vr_mgpu_set_stage_timing(vr_mgpu* m, int enabled)
{
    if (!m) return VR_ERR_INVALID_ARG;
    int rc = vr_mgpu_wait(m);
    if (rc != VR_OK) return rc;
    m->stage_timing = enabled != 0;
    for (auto& k : m->r)
        for (int b = 0; b < kSlots; ++b) k.tm_valid[b] = false;
    return VR_OK;
}

int vr_mgpu_set_frames_in_flight(vr_mgpu* m, int frames)
{
    if (!m || frames < 0 || frames > kSlots) return VR_ERR_INVALID_ARG;
    int rc = vr_mgpu_wait(m);
    if (rc != VR_OK) return rc;
    m->one_stream = frames == 1;
    for (auto& k : m->r) (void)vr_hint_frames_in_flight(k.ctx, m->one_stream ? 1 : (m->slots > 4 ? 4 : m->slots));
    return VR_OK;
}

int vr_mgpu_stage_times(vr_mgpu* m, int local_rank, int which, float ms[4])
{
    if (!m || !ms || local_rank < 0 || local_rank >= (int)m->r.size() || which < 0 || which >= m->slots) return VR_ERR_INVALID_ARG;
    Rank& k = m->r[(size_t)local_rank];
    if (!k.tm_valid[which]) return fail(m, VR_ERR_NOT_READY, "vr_mgpu_stage_times: no timed launch in this buffer set");
    MG_HIP(m, hipSetDevice(k.device));
    MG_HIP(m, hipEventSynchronize(k.tm_d[which]));
    MG_HIP(m, hipEventElapsedTime(&ms[0], k.tm_t0[which], k.tm_r[which]));
    MG_HIP(m, hipEventElapsedTime(&ms[1], k.tm_r[which], k.tm_g[which]));
    MG_HIP(m, hipEventElapsedTime(&ms[2], k.tm_g[which], k.tm_d[which]));
    MG_HIP(m, hipEventElapsedTime(&ms[3], k.tm_t0[which], k.tm_d[which]));
    return VR_OK;
}

void* vr_mgpu_present_device_ptr(vr_mgpu* m, int which, int frame_in_launch)
{
    if (!m || which < 0 || which >= m->slots || frame_in_launch < 0 || frame_in_launch >= m->batch_cap) return nullptr;
    for (auto& k : m->r)
        if (k.rank == 0) return k.present[which] + (size_t)frame_in_launch * m->W * m->H;
    return nullptr;
}

int vr_mgpu_download_present(vr_mgpu* m, int which, int frame_in_launch, uint8_t* bgra8)
{
    if (!m || !bgra8 || which < 0 || which >= m->slots || frame_in_launch < 0 || frame_in_launch >= m->batch_cap)
        return VR_ERR_INVALID_ARG;
    int rc = vr_mgpu_wait(m);
    if (rc != VR_OK) return rc;
    for (auto& k : m->r)
        if (k.rank == 0) {
            MG_HIP(m, hipSetDevice(k.device));
            MG_HIP(m, hipMemcpy(bgra8, k.present[which] + (size_t)frame_in_launch * m->W * m->H, (size_t)m->W * m->H * 4,
                                hipMemcpyDeviceToHost));
            return VR_OK;
        }
    return fail(m, VR_ERR_NOT_READY, "vr_mgpu_download_present: this process does not drive the root rank");
}

}  // extern "C"
