/*
 * vr.h -- C ABI of the MI355X-native volume ray-marcher (libvr_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of gutiKristian/VolumeRendering:
 * the per-pixel front-to-back compositing loop that the reference runs as WGSL
 * fragment shaders (App/shaders/{...}.wgsl `fs_main`) behind WebGPU/Dawn.  The reference
 * has no FFI of its own; what it has is a bind-group contract between
 * `Application::OnUpdate/OnRender` and the shaders.  Every entry point below cites the
 * reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns VR_OK (0) or a negative
 *     vr_status; the message for the last failure is available via vr_last_error().
 *     No exception crosses this boundary.  (The reference only logs / asserts:
 *     WebgpuLib/src/Platform/Native/NativeGraphicsContext.cpp:102-116.)
 *   - a vr_ctx is bound to ONE HIP device (one process per GPU) and must be used by one
 *     thread at a time (the reference is single threaded, App/src/Application.cpp:332-379).
 *   - host pointers are copied at call time; the caller keeps ownership, exactly like
 *     wgpuQueueWriteTexture / wgpuQueueWriteBuffer (App/src/renderer/Texture.cpp:80,
 *     App/src/renderer/UniformBuffer.cpp:27).
 *   - all matrices are column-major float[16] (glm layout, App/src/Application.cpp:102-105).
 *   - there is NO CPU fallback behind this ABI: if no HIP device is usable vr_create
 *     fails with VR_ERR_HIP.
 */
#ifndef VR_H_
#define VR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_ABI_VERSION 1

typedef enum vr_status {
    VR_OK = 0,
    VR_ERR_INVALID_ARG = -1,  /* null pointer, bad slot / variant / size                    */
    VR_ERR_HIP = -2,          /* a HIP runtime call failed (message has hipGetErrorString)   */
    VR_ERR_NOT_READY = -3,    /* render called before the volumes / TFs the variant needs    */
    VR_ERR_UNSUPPORTED = -4,  /* e.g. non-identity model matrix (see vr_set_uniforms)        */
    VR_ERR_OOM = -5
} vr_status;

/* One value per fragment shader of the reference. */
typedef enum vr_variant {
    VR_VARIANT_BASIC = 0,       /* App/shaders/BasicVolumeApp.wgsl:113-188   unlit, cut-off dst.a <= 0.95 */
    VR_VARIANT_LIGHT = 1,       /* App/shaders/BasicVolLightApp.wgsl:151-237 lit,   cut-off dst.a <  1.0  */
    VR_VARIANT_VOLUME_MASK = 2, /* App/shaders/VolumeMaskApp.wgsl:128-217    mask + RT + CT               */
    VR_VARIANT_THREE_FILES = 3, /* App/shaders/ThreeFilesApp.wgsl:170-272    CT/RT colour mix             */
    VR_VARIANT_MULTI_CTRT = 4,  /* App/shaders/MultiCTRTApp.wgsl:163-259     CT/RT mix + shade + |g| opacity */
    VR_VARIANT_TF_CALIB = 5,    /* App/shaders/TFCalibrationApp.wgsl:114-197 CT + nearest-sampled mask     */
    /* App/shaders/MutliCTRTIllustrative.wgsl:227-313: MULTI_CTRT with the context-preserving opacity
     * opacityCT * pow(|g|, pow(5 s (1 - d) (1 - dst.a), 0.8)); slots as MULTI_CTRT, reads camera_pos.  The reference
     * compiles this module next to MultiCTRTApp.wgsl but never attaches it (MutliCTRTApp.cpp:112-119). */
    VR_VARIANT_ILLUSTRATIVE = 6,
    /* App/shaders/BasicVolLightApp.wgsl with the call the reference keeps commented out at :212 enabled:
     * gradient = ComputeGradient(currentPosition, stepSize, textMain) (:239-253) -- central differences of six extra
     * trilinear density samples at +-stepSize along the uvw axes, negated and normalised (zero length -> 0), instead of
     * the pre-computed .rgb of the voxels.  Slots and light as LIGHT; only the density plane of the volume is read.  */
    VR_VARIANT_LIGHT_INSHADER = 7,
    VR_VARIANT_COUNT = 8
} vr_variant;

#define VR_MAX_VOLUMES 3
#define VR_MAX_TFS 2

/*
 * Volume slots = the order of the texture_3d bindings in the scene's @group(1):
 *   BASIC / LIGHT : 0 = volume                      (BasicVolLightApp.wgsl:50)
 *   VOLUME_MASK   : 0 = mask, 1 = RT dose, 2 = CT   (VolumeMaskApp.wgsl:40-42)
 *   THREE_FILES   : 0 = CT, 1 = RT, 2 = mask (bound, never sampled) (ThreeFilesApp.wgsl:50-52)
 *   MULTI_CTRT    : 0 = CT, 1 = RT                  (MultiCTRTApp.wgsl:50-51)
 *   TF_CALIB      : 0 = CT, 1 = mask                (TFCalibrationApp.wgsl:40-41)
 * TF slots = the order of the (opacity, colour) texture_1d pairs:
 *   single-TF scenes: 0;  two-TF scenes: 0 = CT pair, 1 = RT pair (VolumeMaskApp.wgsl:43-46).
 */

/*
 * The uniform block.  Fields 1:1 with @group(0) of every shader
 * (App/shaders/BasicVolumeApp.wgsl:26-37, filled by App/src/Application.cpp:544-556 and
 * rewritten every frame by Application::OnUpdate, App/src/Application.cpp:96-119), followed
 * by the per-scene `Light` (App/src/renderer/Light.h:9-14 <-> BasicVolLightApp.wgsl:25-33).
 */
typedef struct vr_uniforms {
    float model[16];      /* binding 0 @0   : must be identity (Application.cpp:489-492 never rewrites it) */
    float view[16];       /* binding 0 @64  : Camera::GetViewMatrix              */
    float proj[16];       /* binding 0 @128 : Camera::GetProjectionMatrix        */
    float view_inv[16];   /* binding 0 @192 : Camera::GetInverseViewMatrix       */
    float proj_inv[16];   /* binding 0 @256 : Camera::GetInverseProjectionMatrix */
    float camera_pos[3];  /* binding 1      : Camera::GetPosition                */
    int32_t fragment_mode;/* binding 4      : 0 volume, 1 |dir|, 2 start, 3 end, 4 screen uv (Application.h:191-198) */
    int32_t steps_count;  /* binding 5 */
    float step_size;      /* binding 6 */
    float clip_x[2];      /* binding 7 */
    float clip_y[2];      /* binding 8 */
    float clip_z[2];      /* binding 9 */
    int32_t toggles[4];   /* binding 10: [0] variable step, [1] jitter, [2],[3] unused */
    float light_pos[4];   /* Light.Position */
    float light_ambient[4];
    float light_diffuse[4];
} vr_uniforms;

typedef struct vr_ctx vr_ctx;

/* ---- lifetime ------------------------------------------------------------------------- */

/* Replaces: base::Window + GraphicsContext::Init + Application::Initialize{Uniforms,Textures,
 * BindGroups,RenderPipelines} (App/src/Application.cpp:58-94, 475-615).  width x height is the
 * viewport (reference default 1280x720, Application.h:100-101).  device_id = HIP ordinal.      */
int vr_create(vr_ctx** out, uint32_t width, uint32_t height, int device_id);

/* Replaces: Application::OnResize (Application.cpp:299-323) -- re-creates the per-pixel buffers. */
int vr_resize(vr_ctx* ctx, uint32_t width, uint32_t height);

void vr_destroy(vr_ctx* ctx);

/* Message of the last failing call on this ctx (ctx may be NULL: last vr_create failure). */
const char* vr_last_error(const vr_ctx* ctx);

int vr_abi_version(void);

/* ---- scene resources ------------------------------------------------------------------ */

/* Replaces: Texture::CreateFromData(..., WGPUTextureDimension_3D, size, RGBA32Float, ...)
 * (App/src/renderer/Texture.cpp:34-85) called from every MiniApp::OnStart, e.g.
 * App/src/miniapps/BasicVolLightApp.cpp:39-40.  `vec4_voxels` is VolumeFile::GetVoidPtr():
 * nx*ny*nz glm::vec4, x fastest, index z*ny*nx + y*nx + x (VolumeFile.cpp:306), .rgb = gradient
 * (or the raw value), .a = density.                                                             */
int vr_volume_upload(vr_ctx* ctx, int slot, const float* vec4_voxels, uint16_t nx, uint16_t ny, uint16_t nz);

/* Same, from a DEVICE pointer that already holds the vec4 voxels (no host round trip). */
int vr_volume_upload_device(vr_ctx* ctx, int slot, const void* d_vec4_voxels, uint16_t nx, uint16_t ny, uint16_t nz);

/* ---- data preparation on the device ------------------------------------------------------
 * The reference prepares volumes on one CPU thread at load time; these do the same arithmetic on the GPU,
 * in place on an uploaded slot, bit for bit (tests/test_prep_gpu.py).
 *
 * vr_volume_upload_raw16/32: the readers' broadcast of the raw integer to all four lanes
 *   (App/src/file/dicom/DicomReader.cpp:239,247; App/src/file/dat/DatReader.cpp:42).
 * vr_volume_normalize: VolumeFile::NormalizeData (App/src/file/VolumeFile.cpp:165-184): .a /= value;
 *   value == 0 -> the maximum of component [0] truncated to an integer (GetMaxNumber, :53-60); the value used is
 *   returned through *used_value (may be NULL).
 * vr_volume_precompute_gradient: VolumeFile::PreComputeGradient (VolumeFile.cpp:196-257): .rgb = (-(p - m)) * 0.5
 *   per axis from the +-1 neighbours' .a (0 outside the grid); norm_to_zero_one != 0 divides every component by
 *   the largest gradient magnitude.
 * vr_volume_download: reads a slot back (n voxels * 4 floats).                                               */
int vr_volume_upload_raw16(vr_ctx* ctx, int slot, const uint16_t* raw, uint16_t nx, uint16_t ny, uint16_t nz);
int vr_volume_upload_raw32(vr_ctx* ctx, int slot, const uint32_t* raw, uint16_t nx, uint16_t ny, uint16_t nz);
int vr_volume_normalize(vr_ctx* ctx, int slot, int normalization_value, int* used_value);
int vr_volume_precompute_gradient(vr_ctx* ctx, int slot, int norm_to_zero_one);
int vr_volume_download(vr_ctx* ctx, int slot, float* vec4_voxels);

/* Replaces: OpacityTF / ColorTF texture creation and TransferFunction::UpdateTexture
 * (App/src/tf/OpacityTf.cpp:25-26,134-142; App/src/tf/ColorTf.cpp:23-24).  opacity: R floats
 * (R32Float 1-D), color_rgba: 4R floats (RGBA32Float 1-D).                                       */
int vr_tf_upload(vr_ctx* ctx, int slot, const float* opacity, const float* color_rgba, uint32_t resolution);

/* The two textures of a pair are separate objects in the reference and may differ in resolution (OpacityTF::Load
 * re-resolves only its own, OpacityTf.cpp:298): upload one table of a slot without touching the other.       */
int vr_tf_upload_opacity(vr_ctx* ctx, int slot, const float* opacity, uint32_t resolution);
int vr_tf_upload_color(vr_ctx* ctx, int slot, const float* color_rgba, uint32_t resolution);

/* Replaces: the 12 wgpuQueueWriteBuffer calls of Application::OnUpdate (Application.cpp:96-119)
 * plus the scene's Light uniform (BasicVolLightApp.cpp:42).  Returns VR_ERR_UNSUPPORTED when
 * `model` is not the identity (the reference never uploads anything else).                       */
int vr_set_uniforms(vr_ctx* ctx, const vr_uniforms* u);

/* ---- the hot path --------------------------------------------------------------------- */

/* Replaces: the "ray end" pass + the volume pass of Application::OnRender
 * (Application.cpp:150-220): analytic ray/box set-up (rayCoords.wgsl + rasteriser) and the
 * fs_main compositing loop of the chosen shader, for every pixel.  Synchronous on return.       */
int vr_render(vr_ctx* ctx, int variant);

/* Image-tile partition (multi-GPU, one process per GPU): renders only the 64x64 screen tiles t
 * with (t % world) == rank, t = ty * tiles_x + tx, and stores them packed, tile after tile in
 * increasing t, each tile as 64*64 RGBA32F row-major (pixels outside the viewport = 0).
 * The packed buffer is `vr_tile_count(ctx, rank, world) * 64*64*4` floats.                       */
int vr_render_tiles(vr_ctx* ctx, int variant, int rank, int world);
int vr_tile_count(const vr_ctx* ctx, int rank, int world);

/* Asynchronous forms: enqueue on `stream` (a hipStream_t, NULL = the ctx's own stream) and write
 * to DEVICE memory supplied by the caller; nothing is synchronised.  `d_frame` = W*H*4 floats;
 * `d_tiles` as described above.  These are what a multi-rank host (RCCL gather) drives.
 * Up to FOUR renders may be in flight at a time, each on its own stream and into its own buffer (use
 * them in turn): the next frame fills the machine while the previous one's longest rays drain (two in
 * flight give 1.45x the frame rate of one on C3, three 1.5x).  What is enforced is EIGHT launches: a
 * ninth enqueue blocks the calling thread until the oldest of the eight has finished (each launch owns
 * one of eight record buffers, guarded by an event), so a caller that keeps four frames in flight never
 * waits for its oldest launch inside an enqueue.  vr_volume_upload*, vr_volume_normalize / _gradient,
 * vr_tf_upload*, vr_resize and vr_destroy drain the whole device first, so they are safe to call
 * while asynchronous renders are still in flight on the caller's streams.                          */
int vr_render_async(vr_ctx* ctx, int variant, void* d_frame, void* stream);

/* Streams for frames in flight.  HIP maps streams onto a few hardware queues, and two streams that share a queue run
 * their kernels one after the other -- two frames "in flight" on such a pair gain nothing (measured: 0.60 instead of 0.48 ms
 * per C3 frame).  vr_stream(ctx, i), i = 0..3, returns context-owned streams (hipStream_t) that were probed, on first use,
 * to really run side by side (two 150 us single-wavefront kernels; ~3 ms once); fewer than four may exist, then the index
 * wraps.  Use them in turn for vr_render_async / vr_render_tiles_async; NULL on failure.                              */
void* vr_stream(vr_ctx* ctx, int index);

/* Hint: how many frames the caller keeps in flight on different streams (1 = one at a time, the default; up to 4).  It only
 * steers the default choice of lanes per ray (vr_set_kernel_flavour(0)): frames that overlap fill the machine together, so
 * the throughput-optimal one-lane kernel is preferred earlier than for a frame that has the machine to itself.  Results do
 * not depend on it.                                                                                                    */
int vr_hint_frames_in_flight(vr_ctx* ctx, int frames);
int vr_render_tiles_async(vr_ctx* ctx, int variant, int rank, int world, void* d_tiles, void* stream);

/* Root side of the gather: `d_gathered` holds, for r = 0..world-1, rank r's packed tiles, each
 * rank's segment padded to `tiles_per_rank_max * 64*64*4` floats; scatters them back into the
 * W*H*4 frame `d_frame` (device).                                                                */
int vr_unpack_tiles_async(vr_ctx* ctx, const void* d_gathered, int world, void* d_frame, void* stream);

/* Several frames in ONE launch.  Replaces: nothing in the reference (it renders one frame per OnRender,
 * App/src/Application.cpp:121-239); this is the throughput form for callers that know the next frames' cameras
 * (turntables, offline sequences, and above all a multi-GPU run, where one rank's share of a frame is a launch far too
 * short to fill a GPU: a rank's eighth of the 1080p frame keeps 9 % of the wavefront slots busy).  n_frames = 1 .. 4
 * frames of the SAME scene (the volumes and tables currently bound) are marched by one grid -- frame f with
 * uniforms[f] into d_frames[f] / d_tiles[f] (device, each W*H*4 floats / packed tiles as above).  The context's own
 * uniforms (vr_set_uniforms) are neither used nor changed.  Every frame is bit-identical to what vr_render_async
 * would have produced with the same uniforms; vr_last_counters reports the LAST frame of the launch.  One launch counts
 * as one render in flight.                                                                                        */
int vr_render_batch_async(vr_ctx* ctx, int variant, int n_frames, const vr_uniforms* uniforms, void* const* d_frames, void* stream);
int vr_render_tiles_batch_async(vr_ctx* ctx, int variant, int rank, int world, int n_frames, const vr_uniforms* uniforms,
                                void* const* d_tiles, void* stream);

/* vr_unpack_tiles_async for gathered segments that lie `rank_stride_tiles` tiles apart (>= a segment) instead of back to
 * back: after gathering n frames' segments at once rank r's tiles of frame f start at
 * d_gathered + ((r * n + f) * tiles_per_rank_max) * 64*64*4 floats, so frame f is unpacked from the base of ITS first
 * segment with rank_stride_tiles = n * tiles_per_rank_max.                                                        */
int vr_unpack_tiles_strided_async(vr_ctx* ctx, const void* d_gathered, int world, int rank_stride_tiles, void* d_frame, void* stream);

/* Replaces: reading back the colour attachment.  frag_rgba (W*H*4 floats, may be NULL) receives
 * the fragment shader output `dst` per pixel BEFORE output merge; pixels with no fragment = 0.
 * present_bgra8 (W*H*4 bytes, may be NULL) receives the presented pixel: blend
 * SrcAlpha/OneMinusSrcAlpha over the white background quad, BGRA8Unorm
 * (App/src/renderer/PipelineBuilder.cpp:142-147, App/shaders/fullscreen.wgsl:33-41,
 * NativeGraphicsContext.cpp:118).  composited_samples (may be NULL) = number of loop iterations
 * whose blend executed, summed over the pixels rendered by the last render call.                 */
int vr_download(vr_ctx* ctx, float* frag_rgba, uint8_t* present_bgra8, uint64_t* composited_samples);

/* Presentation without a host round trip.  Replaces: the output merge + swap-chain present of Application::OnRender
 * (App/src/Application.cpp:180-233: blend SrcAlpha/OneMinusSrcAlpha over the white background quad into the BGRA8Unorm
 * swap-chain image) for a caller that shows the frame itself: writes the presented BGRA8Unorm pixels of the device frame
 * `d_frame` (W*H*4 floats; NULL = the ctx-owned frame of vr_render) into DEVICE memory `d_bgra8` (W*H*4 bytes) on `stream`
 * (NULL = the ctx's own) -- e.g. a GL / Vulkan buffer or texture staging buffer imported into HIP
 * (hipGraphicsGLRegisterBuffer + hipGraphicsResourceGetMappedPointer, or hipImportExternalMemory).  Nothing is
 * synchronised; the same arithmetic as vr_download's present_bgra8.                                               */
int vr_present_async(vr_ctx* ctx, const void* d_frame, void* d_bgra8, void* stream);

/* vr_present_async for the root of a multi-GPU gather: presents straight from the gathered, tile-major segments
 * (layout as vr_unpack_tiles_strided_async; rank_stride_tiles <= 0 = segments back to back) into `d_bgra8` (W*H*4 bytes,
 * row-major), so a viewer that only wants the presented frame needs no un-permuted float frame in between.  Replaces the
 * same output merge (App/src/renderer/PipelineBuilder.cpp:142-154); bit-identical to vr_unpack_tiles_async + vr_present_async. */
int vr_present_tiles_async(vr_ctx* ctx, const void* d_gathered, int world, int rank_stride_tiles, void* d_bgra8, void* stream);

/* Presenting where the tiles are rendered (multi-GPU, the presented frame alone is wanted): vr_present_packed_async applies the same
 * output merge to a rank's PACKED tiles (n_tiles x 64 x 64 RGBA32F -> as many BGRA8Unorm words, same order) -- the gather then moves
 * 4 bytes per pixel instead of 16 -- and vr_unpack_tiles_bgra8_async is vr_unpack_tiles_strided_async for such gathered BGRA8 tiles
 * (gathered[r][n][64*64] words -> the W x H frame).  Pixel for pixel the bytes of vr_present_async on the assembled frame.     */
int vr_present_packed_async(vr_ctx* ctx, const void* d_tiles_rgba, int n_tiles, void* d_tiles_bgra8, void* stream);
int vr_unpack_tiles_bgra8_async(vr_ctx* ctx, const void* d_gathered_bgra8, int world, int rank_stride_tiles, void* d_bgra8, void* stream);

/* Packed tiles of the last vr_render_tiles (host copy). */
int vr_download_tiles(vr_ctx* ctx, float* tiles_rgba, uint64_t* composited_samples);

/* ---- instrumentation ------------------------------------------------------------------ */

/* HIP-event time of the last synchronous render's (vr_render / vr_render_tiles) march kernel and of
 * the whole render call, in milliseconds; VR_ERR_NOT_READY after an *_async render (those are timed by
 * vr_kernel_times alone).  Replaces the FPS / frame-time read-out (Application.cpp:339-370). */
int vr_last_timing(vr_ctx* ctx, float* kernel_ms, float* total_ms);

/* Durations (ms) of the march launches of the most recent render calls, oldest first: first workgroup start to last
 * workgroup end on the 100 MHz device clock, taken from the launch's own per-workgroup records by the small kernel that
 * sorts them (two timing events around every launch cost the frame's stream 11 us: 2 % of a 1080p frame, 8 % of a
 * 1024^2 unlit one); launches without a sort behind them (LDS-tile flavours, > 192 K workgroups) are timed with HIP events
 * on their stream.  At most `capacity` (<= 256) values are written, the number written is returned (negative = error).
 * Waits for the launches concerned.  vr_reset_kernel_times() empties the ring.                                       */
int vr_kernel_times(vr_ctx* ctx, float* out_ms, int capacity);

/* How vr_kernel_times measures: VR_TIMING_RECORDS (default, as described above) or VR_TIMING_EVENTS -- a pair of HIP events
 * around every march launch on the stream it is enqueued on, whatever the launch (what bench.py's roofline divides by).  */
#define VR_TIMING_RECORDS 0
#define VR_TIMING_EVENTS 1
int vr_set_kernel_timing(vr_ctx* ctx, int mode);
int vr_reset_kernel_times(vr_ctx* ctx);

/* Viewport size and HIP device ordinal of a context (any pointer may be NULL). */
int vr_viewport(const vr_ctx* ctx, uint32_t* width, uint32_t* height, int* device_id);

/* Device pointer of the ctx-owned frame buffer (W*H*4 floats) written by vr_render. */
void* vr_frame_device_ptr(vr_ctx* ctx);

/* Number of pixels that produced a fragment in the last render (front face hit & not clipped). */
int vr_last_covered_pixels(vr_ctx* ctx, uint64_t* covered);

/* Counters of the last render: out[0] = composited samples (blends the reference's loop executes),
 * out[1] = covered pixels, out[2] = samples whose voxels were actually fetched (= out[0] minus the
 * samples the exact empty-space test proved to be the identity).                                 */
int vr_last_counters(vr_ctx* ctx, uint64_t out[3]);

/* Per-workgroup trace of the last march launch, for load-balance analysis: 6 words per workgroup
 * {composited samples, covered pixels, fetched samples, start, end (100 MHz device clock), HW_ID | XCC_ID << 32},
 * in blockIdx order.  Writes min(capacity, n) records and returns n (the number of workgroups launched).      */
int vr_last_block_trace(vr_ctx* ctx, uint64_t* out, int capacity);

/* Kernel flavour for A/B measurements.  All flavours are bit-identical in output and in the composited / covered counts; the
 * FETCHED count (vr_last_counters out[2]) is the same for all but 1 and 16, which skip nothing and fetch every composited sample
 * -- the default may pick 16 for volumes with next to nothing to skip, so `fetched` can differ between frames of one scene.
 * (2, 3, 4, 5, 9 and 14 exist only in builds with -DVR_EXPERIMENTAL_FLAVOURS=1: vr_experimental_flavours below.)
 *   0  default: a MEASURED choice.  Every form below gives the same bits, so the context tries the eligible ones on the
 *      caller's own frames -- three launches each (frames in flight + 3 with launches in flight), behind a few launches of the
 *      prior's pick so that a launch order exists -- and keeps the fastest by the launches' own records (no synchronisation: the
 *      sort behind a launch writes its duration to pinned memory).  Per "what is launched of what": shader, rank share, viewport,
 *      frames per launch, vr_hint_frames_in_flight, volume / table uploads, arithmetic, layout; the trial re-opens when the
 *      longest ray chain has moved by a quarter.  Candidates: the prior's pick (exact skipping; whole one-at-a-time frames of
 *      the lit / unlit shader and the composite: 17, or 16 with nothing to skip; else lanes per ray from the launch size, the
 *      frames in flight and the chain length of an earlier launch), 17 / 16, 6, 18, and 10 / 11 (launches that leave the machine
 *      part empty) or 12.  vr_kernel_choice reports what was measured.  VR_EXP_TUNE=0 in the environment: the prior alone.
 *   1  one lane per ray, no empty-space skipping (every composited sample is fetched)
 *   2 / 3  LDS wave tiles without / with skipping (lit shader only; others fall back to 1 / 6)
 *   4  skipping + closed-form leaping (f32 accumulation as integer arithmetic on the bit patterns)
 *      (launches of several frames exist for the plain, run and depth-parallel loop forms: there 2 runs as 1 and 3 / 4 / 9 as 6)
 *   5  skipping alone, one step per iteration
 *   6  one lane per ray (forced), 7  four lanes per ray (forced), 8  two lanes per ray (forced)
 *   9  one lane per ray with the next step's corner loads software-pipelined behind the shading
 *   10 / 11  four / two lanes per ray with the next round's corner loads software-pipelined (lit shader; what
 *      the default uses for small launches)
 *   12 / 13  persistent wavefronts (csrc/vr_pw.h): one workgroup of 16 wavefronts per CU, the packets come from a queue
 *      (eight heads, longest chains first), TF slot 0 is read from LDS; 13 also issues the next step's corner loads before
 *      this step's shading (lit / unlit shader).  Launches of one frame; fewer bytes through the texture addressers
 *   16 / 17  persistent wavefronts with the corner loads TWO steps ahead (csrc/vr_p2.h): two corner buffers, the gather an indexed
 *      buffer load of the voxel's slot in the bricked copy, the slot arithmetic of a cell from per-axis tables in LDS beside TF
 *      slot 0.  16 skips nothing (volumes with nothing to skip; 2 wavefronts per SIMD); 17 decides the exact empty-space skipping
 *      ahead of the loads (one distance-field byte per ray rides along with each corner buffer; idle rays' lanes are switched off
 *      for the loads; runs of identity steps become jumps of the requests while the steps in flight are consumed; 3 per SIMD).
 *      Lit / unlit shader and (17; 16 runs as 17) the three-volume composite, whose mask and dose are fetched on demand behind the
 *      per-brick mask record; volumes of 4 GiB and more through a window of z-slabs of bricks that follows the packet; launches of
 *      several frames take (frame, packet) items from the one queue.  Needs one table resolution <= 8190 and the bricked copy
 *      (vr_set_volume_layout(0)); else 13 / 12, or 6 for launches of several frames
 *   18  the one-lane kernel (6) with the slot arithmetic of volume 0's cells from per-axis tables in its workgroup's LDS (the
 *      clamp-to-edge texel pair of a coordinate is one ds_read2_b32; filled per workgroup, by the workgroups that can hit the box);
 *      the shaders that sample one volume (lit, unlit, in-shader gradient) on the bricked copy, launches of any number of frames;
 *      else it runs as 6
 *   14  (experimental build) lanes per ray chosen PER PACKET (csrc/vr_mixed.h): packets whose longest ray chain in an earlier
 *      launch of the same shape reached 75 % of that launch's longest are marched as two half packets with two lanes per ray,
 *      the rest with one; one-frame launches of the shaders that have a depth-parallel form
 *   15  the voxels of a packet's next four steps in an LDS tile filled by LDS-DMA (csrc/vr_lt.h): lit shader, launches of
 *       one frame (other launches run 6); never picked by the default -- slower than 17 / 16 wherever measured            */
int vr_set_kernel_flavour(vr_ctx* ctx, int flavour);

/* What the default's measured choice (flavour 0) knows about the launch shape it was asked for last: the candidates' flavours, the
 * milliseconds per launch measured for each (0 = its trial has not been evaluated yet) and the index of the one kept (-1 = trial
 * running: the prior's pick, flavours[0], runs meanwhile).  Returns the number of candidates (0: nothing launched through the
 * default yet, or the shape has a single eligible form).                                                               */
int vr_kernel_choice(vr_ctx* ctx, int flavours[6], float ms_per_launch[6], int* chosen);

/* 1 if the library was built with -DVR_EXPERIMENTAL_FLAVOURS=1: the kernel forms that lost every A/B -- flavours 2, 3, 4, 5, 9,
 * 14 and volume layout 2 -- are then compiled in; 0 (the shipped build): vr_set_kernel_flavour / vr_set_volume_layout return
 * VR_ERR_UNSUPPORTED for them.                                                                                           */
int vr_experimental_flavours(void);

/* Arithmetic mode.  WGSL leaves it to the implementation whether `a * b + c` is evaluated with one rounding or two
 * (the reference's Tint -> HLSL -> D3D12 back end emits `mad`).
 *   VR_ARITH_SEPARATE (default)  product and sum are rounded separately everywhere: bit-exact with the oracle's default mode
 *   VR_ARITH_FUSED               the per-sample expressions of that shape -- texture coordinates p * N - 0.5, every
 *                                linear-filter lerp a + (b - a) * t, dot products, light.diffuse * m * kD + light.ambient * kA,
 *                                the CT / RT colour mix and FrontToBackBlend -- are single fused multiply-adds: bit-exact with
 *                                the oracle's fused mode, about a fifth fewer vector instructions per sample.  Ray placement
 *                                (matrices, slab test, step vectors, p += step) is identical in both modes.             */
#define VR_ARITH_SEPARATE 0
#define VR_ARITH_FUSED 1
int vr_set_arithmetic(vr_ctx* ctx, int mode);

/* Volume layout in HBM (A/B measurements; frames and counts are bit-identical in every mode).
 *   0  default: the march kernels gather from a BRICKED copy of every slot -- the vec4 voxels and a scalar f32 density plane
 *      (what fetches that consume .a alone read: BasicVolumeApp.wgsl:171, the density / dose fetches of the other shaders)
 *      in bricks of 4 x 4 x 4 voxels, brick after brick: 1 KiB per brick, a 128-byte line = 4 x 2 x 1 voxels, so that the
 *      lines a packet of rays needs next are near the ones it has whatever direction it travels in.  The reference's
 *      x-fastest array (App/src/file/VolumeFile.cpp:287-307: what vr_volume_upload takes, the data-preparation calls work on
 *      and vr_volume_download returns) stays resident beside it; the copy is rebuilt after every upload / in-place change
 *   3  round 2's default: the reference's x-fastest vec4 voxels + an x-fastest density plane
 *   1  the reference's RGBA32F voxels only (16 B / voxel; what round 1 measured)
 *   2  0 + the lit shader derives the eight corner gradients from the plane on the fly when the slot's .rgb is verified,
 *      at upload, to be VolumeFile::PreComputeGradient(false) of its .a bit for bit (VolumeFile.cpp:196-257): a quarter of
 *      the footprint and 0.69x the fabric traffic, but 1.5x the L1 accesses of the row-major plane -- measured slower
 *      (DESIGN.md section 4.5), kept for A/B
 * vr_volume_layout: *flags bit 0 = density plane present, bit 1 = .rgb verified as the central difference of .a,
 * bit 2 = the last render derived its gradients on the fly, bit 3 = the bricked copy is what the gathers read.  */
int vr_set_volume_layout(vr_ctx* ctx, int mode);
int vr_volume_layout(vr_ctx* ctx, int slot, int* flags);

/* The flavour the last render actually ran (what 0 resolved to for that launch), or a negative vr_status. */
int vr_last_kernel_flavour(vr_ctx* ctx);

/* Flavour 14 (lanes per ray chosen per packet): how many 8x8 packets the last launch marched as two half packets with two
 * lanes per ray -- 0 until an item list of an earlier launch of the same shape exists (the fourth launch or so), or when the
 * last launch was of another flavour.  Negative vr_status on error.                                                      */
int vr_last_split_packets(vr_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* VR_H_ */
