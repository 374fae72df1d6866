/*
 * vr_mgpu.h -- C ABI of the multi-GPU frame loop (libvr_mgpu.so): image-tile partition of one frame over the GPUs of
 * one node, one RCCL gather over xGMI per frame, un-permute on the root.
 *
 * What it replaces: nothing in the reference -- gutiKristian/VolumeRendering renders on one device
 * (WebgpuLib/src/Base/GraphicsContext.h:35-42, static singletons); the caller is still Application::OnRender
 * (App/src/Application.cpp:121-239), once per frame.  The partition is the one BASELINE.json's north_star names:
 * the frame is cut into 64x64 tiles, tile t = ty * tiles_x + tx is owned by rank t % world (interleaved, so every
 * rank gets an even sample of the screen), the volume and the tables are replicated, the uniforms are identical on
 * every rank.  Each rank renders its tiles packed (vr_render_tiles_async, include/vr.h), ncclGather
 * (/opt/rocm/include/rccl/rccl.h) delivers every rank's segment to the root -- every peer owns a direct xGMI link to
 * the root, so the transfers run side by side and are not ring-bound -- and vr_unpack_tiles_async scatters them into
 * the frame.  No other collective exists on the path.
 *
 * Two ways to drive it, same frame loop behind both:
 *   one process per GPU   vr_mgpu_unique_id on rank 0, the 128 bytes carried to the other ranks by the launcher
 *                         (MPI, a torch.distributed store, a file ...), then vr_mgpu_create on every rank with the
 *                         rank's own vr_ctx.  This is what bench.py --gpus N uses under torch.distributed.run.
 *   one process, N GPUs   vr_mgpu_create_local: the driver owns N contexts (one per device), ncclCommInitAll, and
 *                         issues the per-device calls of a frame inside one ncclGroupStart / ncclGroupEnd.  Scene
 *                         resources are uploaded to every context (vr_mgpu_context(m, i)).
 *
 * Frames are pipelined: frame k+1 renders while frame k's tiles travel and are un-permuted (tile / gather / frame
 * buffer sets used in turn; two by default, 1..4 with VR_MGPU_SLOTS).  A rank's share of a frame is a short,
 * latency-bound launch; what fills the GPUs is several frames per launch (vr_mgpu_frames_async below), not more
 * launches in flight: the runtime runs two or three of a process's streams side by side, no more.
 * vr_mgpu_frame_async never blocks the host on the GPU beyond the C ABI's own bound of four launches in flight;
 * vr_mgpu_wait drains the pipeline.  A caller that really keeps frames in flight says so on every rank's context
 * (vr_hint_frames_in_flight(ctx, 2)): a rank's share is a small launch, and the default choice of lanes per ray depends
 * on whether other frames fill the machine beside it.
 *
 * Conventions as in vr.h: plain C, 0 = ok, negative = vr_status, message via vr_mgpu_last_error; no exception
 * crosses the boundary; one thread at a time per handle.
 */
#ifndef VR_MGPU_H_
#define VR_MGPU_H_

#include <stdint.h>

#include "vr.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VR_MGPU_ID_BYTES 128 /* = NCCL_UNIQUE_ID_BYTES */

typedef struct vr_mgpu vr_mgpu;

/* ncclGetUniqueId: call on ONE rank, distribute the bytes to all ranks out of band. */
int vr_mgpu_unique_id(void* id128);

/* One process per GPU.  `ctx` is this rank's context (created on this rank's device; the driver borrows it and never
 * destroys it); rank / world as in vr_render_tiles; id128 from vr_mgpu_unique_id.  Collective: every rank must call
 * it (ncclCommInitRank).                                                                                            */
int vr_mgpu_create(vr_mgpu** out, vr_ctx* ctx, int rank, int world, const void* id128);

/* One process, n_devices GPUs: creates one W x H context per device (owned by the driver) and the communicators
 * (ncclCommInitAll).  Local rank i renders on device_ids[i]; rank 0 is the root.                                    */
int vr_mgpu_create_local(vr_mgpu** out, uint32_t width, uint32_t height, const int* device_ids, int n_devices);

void vr_mgpu_destroy(vr_mgpu* m);
const char* vr_mgpu_last_error(const vr_mgpu* m);

int vr_mgpu_world(const vr_mgpu* m);        /* ranks in the partition                               */
int vr_mgpu_local_ranks(const vr_mgpu* m);  /* ranks driven by this process (1, or n_devices)      */
vr_ctx* vr_mgpu_context(vr_mgpu* m, int local_rank);

/* Enqueue one frame: every local rank renders its tiles, the gather, the root's un-permute.  Uses the uniforms,
 * volumes and tables currently set on each context.  Returns the buffer set (0 / 1) the frame will land in.        */
int vr_mgpu_frame_async(vr_mgpu* m, int variant);

/* Several frames per launch (vr_render_tiles_batch_async, include/vr.h): every local rank marches its tiles of n_frames
 * (1..4) frames in ONE launch -- frame f with uniforms[f], the same array on every rank; the contexts' own uniforms are not
 * used -- one gather carries all n_frames segments of a rank, the root un-permutes n_frames frames.  A rank's share of a
 * frame is a launch too short to fill a GPU (an eighth of the 1080p frame keeps 9 % of the wavefront slots busy): this is
 * the throughput form of the loop, at the price of n_frames frames of delay.  The first call with a larger n_frames drains
 * the pipeline and re-sizes the buffer sets.  Returns the buffer set; frame f of the launch is then found with
 * vr_mgpu_batch_frame_device_ptr / vr_mgpu_download_batch_frame (f = 0 is what vr_mgpu_frame_device_ptr /
 * vr_mgpu_download return).  Collective: every rank must issue the same sequence of calls.                          */
int vr_mgpu_frames_async(vr_mgpu* m, int variant, int n_frames, const vr_uniforms* uniforms);
void* vr_mgpu_batch_frame_device_ptr(vr_mgpu* m, int which, int frame_in_launch);
int vr_mgpu_download_batch_frame(vr_mgpu* m, int which, int frame_in_launch, float* frag_rgba);

/* Block until every enqueued frame is complete (on the root: assembled in its frame buffer).                      */
int vr_mgpu_wait(vr_mgpu* m);

/* Root only (NULL elsewhere): device pointer of assembled frame buffer `which` (W*H*4 floats).                    */
void* vr_mgpu_frame_device_ptr(vr_mgpu* m, int which);

/* Root only: waits, then copies frame buffer `which` to the host (W*H*4 floats).                                  */
int vr_mgpu_download(vr_mgpu* m, int which, float* frag_rgba);

/* Sum over ALL ranks of the last frame's counters (composited samples, covered pixels, fetched samples) and the
 * maximum over all ranks of `local_value` (e.g. a wall time): two tiny ncclAllReduce calls, blocking.  Collective. */
int vr_mgpu_reduce(vr_mgpu* m, uint64_t counters_sum[3], double local_value, double* max_value);

/* Which transport the gather uses, for reports: "RCCL <version> ncclGather".                                      */
const char* vr_mgpu_backend(const vr_mgpu* m);

/* Ranks RCCL itself sees in the communicator (ncclCommCount, /opt/rocm/include/rccl/rccl.h), and the HIP device local rank
 * `local_rank` renders on: what a report should print beside `world`, so that a run describes the partition it really had.  */
int vr_mgpu_comm_count(const vr_mgpu* m);
int vr_mgpu_device(const vr_mgpu* m, int local_rank);

/* What the root produces from the gathered segments of every launch (default VR_MGPU_OUT_FRAME):
 *   VR_MGPU_OUT_FRAME    the assembled W x H float frames (vr_unpack_tiles_strided_async: one un-permute pass per frame,
 *                        16 B read + 16 B written per pixel) -- vr_mgpu_frame_device_ptr / vr_mgpu_download
 *   VR_MGPU_OUT_PRESENT  the presented BGRA8Unorm frames (the output merge of App/src/renderer/PipelineBuilder.cpp:142-154
 *                        over the white background) -- vr_mgpu_present_device_ptr / vr_mgpu_download_present.  ALONE (a viewer
 *                        that only shows the frame): every rank presents its own tiles where it rendered them
 *                        (vr_present_packed_async; the merge is per pixel, so the bytes are those of presenting the assembled
 *                        frame), the gather moves BGRA8 words -- 4 bytes per pixel over xGMI instead of 16: 8.3 MB instead of
 *                        33 MB per 4K frame onto the root -- and the root un-permutes them (vr_unpack_tiles_bgra8_async).
 *                        VR_MGPU_GATHER_FLOAT=1 in the environment restores the float gather for A/B.  Together with
 *                        VR_MGPU_OUT_FRAME: float tiles are gathered and the root presents straight from the tile-major
 *                        segments (vr_present_tiles_async).
 * Both bits may be set.  Takes effect with the next launch; same value on every rank.                              */
#define VR_MGPU_OUT_FRAME 1
#define VR_MGPU_OUT_PRESENT 2
int vr_mgpu_set_output(vr_mgpu* m, int output);
void* vr_mgpu_present_device_ptr(vr_mgpu* m, int which, int frame_in_launch);
int vr_mgpu_download_present(vr_mgpu* m, int which, int frame_in_launch, uint8_t* bgra8);

/* Stage timeline of a rank (instrumentation; off by default: timing events cost each launch's streams a few microseconds).
 * With it on, every launch records four events; vr_mgpu_stage_times waits for the last launch into buffer set `which`
 * and returns, in milliseconds, ms[0] = march (render start .. tiles rendered), ms[1] = gather (tiles rendered .. segments
 * on the root / sent), ms[2] = output (un-permute and / or present on the root; ~0 elsewhere), ms[3] = start .. end.
 * vr_mgpu_set_stage_timing drains the pipeline.                                                                    */
int vr_mgpu_set_stage_timing(vr_mgpu* m, int enabled);

/* frames == 1: ONE FRAME AT A TIME on the device -- the reference's interactive loop, App/src/Application.cpp:332-379: every
 * rank's march, gather and (on the root) output pass go onto one stream, so a launch starts behind the last stage of the launch
 * before it by stream order, and the host may keep enqueueing ahead instead of waiting between two frames.  Any other value: the
 * default -- as many launches in flight as there are buffer sets (march on a stream per buffer set, gather and output on the
 * communication stream).  Drains the pipeline.                                                                              */
int vr_mgpu_set_frames_in_flight(vr_mgpu* m, int frames);
int vr_mgpu_stage_times(vr_mgpu* m, int local_rank, int which, float ms[4]);

#ifdef __cplusplus
}
#endif
#endif /* VR_MGPU_H_ */
