// vr_launch.h -- picks the instantiation of the march kernels for one launch (variant x addressing x skipping x loop
// form x lanes per ray).  Included once per arithmetic mode: by vr_api.hip (namespace vr, separately rounded
// multiply-adds) and by vr_fused.hip (namespace vrf, fused multiply-adds); enqueue_render calls the one the context's
// arithmetic mode selects (vr_set_arithmetic).
#pragma once
// march_p2_kernel: the several-frames form also for a launch of ONE frame.  It reads a frame's parameters where it uses them, through
// a wave-uniform index, and keeps fewer of them in registers: no scratch reload in the pipelined loop, which the one-frame form of the
// >= 4 GiB kernel has at the register limit (C5 3.37 -> 2.94 ms, a rank's half of it 1.78 -> 1.57), and 1 % on C3 / C4 below 4 GiB
// (tools/experiments/r5a.sh, r5b.sh).  The template argument BATCH still names the launch -- one frame or several: two kernel names in
// a profile -- the code behind both is the same (vr_p2.h: kMulti).  -DVR_P2_WIN_BATCH=0 / -DVR_P2_ALL_BATCH=0: the one-frame code (A/B).
#ifndef VR_P2_ALL_BATCH
#define VR_P2_ALL_BATCH 1
#endif
#ifndef VR_P2_WIN_BATCH
#define VR_P2_WIN_BATCH 1
#endif

#include "../../include/vr.h"
#include "vr_kernels.h"
#include "vr_dp.h"
#include "vr_pw.h"
#include "vr_p2.h"
// Kernel forms that lost every A/B (HISTORY 4.4, 4.5, 4.10) -- flavours 2 / 3 (register-staged LDS wave tiles), 4
// (closed-form leaping), 5 (skipping without runs), 9 (one lane per ray, pipelined corner loads), 14 (lanes per ray chosen per
// packet) and layout 2 (gradients on the fly) -- are compiled only with -DVR_EXPERIMENTAL_FLAVOURS=1
// (VR_EXPERIMENTAL_FLAVOURS=1 in the environment of build.py): half the march kernel instantiations of the shipped library.
// Without them vr_set_kernel_flavour / vr_set_volume_layout reject those values.
// Flavour 15 (vr_lt.h: the voxels of a packet's next steps in an LDS tile filled by LDS-DMA -- the north star's "volume in LDS tiles")
// is part of the shipped library: slower than the two-steps-ahead kernel everywhere measured, selectable with
// vr_set_kernel_flavour(15) and tested on every box, not a candidate of the measured choice.
#ifndef VR_EXPERIMENTAL_FLAVOURS
#define VR_EXPERIMENTAL_FLAVOURS 0
#endif
#include "vr_lt.h"
#if VR_EXPERIMENTAL_FLAVOURS
#include "vr_mixed.h"
#endif
#if !VR_FUSED && VR_EXPERIMENTAL_FLAVOURS
#include "vr_wtb.h"
#endif

namespace VR_KNS {

template <int V, bool OTF = false>
void launch_variant(bool off32, int leap, dim3 grid, dim3 block, hipStream_t s, const MarchBatch& B, unsigned lds_bytes = 0)
{
    constexpr bool kCanSkip = (V == V_BASIC || V == V_LIGHT || V == V_THREE_FILES || V == V_VOLUME_MASK || V == V_LIGHT_INSHADER);
    // launches that carry several frames (MarchBatch) exist for the loop forms the default flavours use: plain and runs
    const bool batch = B.n_frames > 1;
#define VR_LAUNCH(O, S, L)                                                                                             \
    do {                                                                                                               \
        if constexpr ((L) == 0 || (L) == 3) {                                                                          \
            if (batch) {                                                                                               \
                hipLaunchKernelGGL((march_kernel<V, O, S, L, OTF, true>), grid, block, lds_bytes, s, B);                       \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL((march_kernel<V, O, S, L, OTF, false>), grid, block, lds_bytes, s, B);                              \
    } while (0)
    if constexpr (kCanSkip) {
        if (B.frame[0].brick_dist) {
#if VR_EXPERIMENTAL_FLAVOURS
            if (off32) {
                if (leap == 2) VR_LAUNCH(true, true, 2);
                else if (leap == 3) VR_LAUNCH(true, true, 3);
                else if (leap == 1) VR_LAUNCH(true, true, 1);
                else VR_LAUNCH(true, true, 0);
            } else {
                if (leap == 2) VR_LAUNCH(false, true, 2);
                else if (leap == 3) VR_LAUNCH(false, true, 3);
                else if (leap == 1) VR_LAUNCH(false, true, 1);
                else VR_LAUNCH(false, true, 0);
            }
#else
            (void)leap;  // (the loop form with runs is the only skipping form of the shipped library)
            if (off32) VR_LAUNCH(true, true, 3);
            else VR_LAUNCH(false, true, 3);
#endif
            return;
        }
    }
    if (off32) VR_LAUNCH(true, false, 0);
    else VR_LAUNCH(false, false, 0);
#undef VR_LAUNCH
}

template <int V, int K, bool PIPE>
void launch_dp(bool off32, dim3 grid, dim3 block, hipStream_t s, const MarchBatch& B)
{
    constexpr bool kCanSkip = (V == V_BASIC || V == V_LIGHT || V == V_THREE_FILES || V == V_VOLUME_MASK);
#define VR_LAUNCH_DP(O, S)                                                                                             \
    do {                                                                                                               \
        if (B.n_frames > 1) hipLaunchKernelGGL((march_dp_kernel<V, O, S, K, PIPE, true>), grid, block, 0, s, B);       \
        else hipLaunchKernelGGL((march_dp_kernel<V, O, S, K, PIPE, false>), grid, block, 0, s, B);                     \
    } while (0)
    if constexpr (kCanSkip) {
        if (B.frame[0].brick_dist) {
            if (off32) VR_LAUNCH_DP(true, true);
            else VR_LAUNCH_DP(false, true);
            return;
        }
    }
    if (off32) VR_LAUNCH_DP(true, false);
    else VR_LAUNCH_DP(false, false);
#undef VR_LAUNCH_DP
}

// persistent wavefronts (vr_pw.h); the loop form is march_kernel's default (runs through inert bricks when skipping)
template <int V>
void launch_pw(const LaunchDesc& L, hipStream_t s, const MarchBatch& B)
{
    constexpr bool kCanSkip = (V == V_BASIC || V == V_LIGHT || V == V_THREE_FILES || V == V_VOLUME_MASK || V == V_LIGHT_INSHADER);
    const bool skip = kCanSkip && B.frame[0].brick_dist != nullptr;
#define VR_LAUNCH_PW(O, S, T, PP)                                                                                      \
    do {                                                                                                               \
        auto k = march_pw_kernel<V, O, S, T, PP>;                                                                      \
        if (L.lds_bytes > 48u * 1024u) {                                                                               \
            static unsigned raised = 0;  /* per instantiation: the attribute sticks to the function */                 \
            if (L.lds_bytes > raised) {                                                                                \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,\
                                          (int)L.lds_bytes);                                                           \
                raised = L.lds_bytes;                                                                                  \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL(k, L.grid, L.block, L.lds_bytes, s, B, L.queue);                                            \
    } while (0)
#define VR_LAUNCH_PW_T(O, S)                                                                                           \
    do {                                                                                                               \
        if constexpr (kCanPipe) {                                                                                      \
            if (L.pw_pipe) {                                                                                           \
                if (L.pw_ltf) VR_LAUNCH_PW(O, S, true, true);                                                          \
                else VR_LAUNCH_PW(O, S, false, true);                                                                  \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        if (L.pw_ltf) VR_LAUNCH_PW(O, S, true, false);                                                                 \
        else VR_LAUNCH_PW(O, S, false, false);                                                                         \
    } while (0)
    constexpr bool kCanPipe = (V == V_BASIC || V == V_LIGHT);
    constexpr bool kCanP2 = kCanPipe || V == V_VOLUME_MASK;
    if constexpr (kCanP2) {
        if (L.pw_p2) {  // two steps ahead (vr_p2.h; the host: TF slot 0 and the axis tables fit LDS, the bricked copy is in use)
#define VR_LAUNCH_P2(S, WN, BT)                                                                                        \
    do {                                                                                                               \
        auto k = march_p2_kernel<V, S, WN, BT>;                                                                        \
        if (L.lds_bytes > 48u * 1024u) {                                                                               \
            static unsigned raised = 0;                                                                                \
            if (L.lds_bytes > raised) {                                                                                \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,\
                                          (int)L.lds_bytes);                                                           \
                raised = L.lds_bytes;                                                                                  \
            }                                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL(k, L.grid, L.block, L.lds_bytes, s, B, L.queue);                                            \
    } while (0)
#define VR_LAUNCH_P2_W(S)                                                                                              \
    do {                                                                                                               \
        if (L.pw_p2_win) {                                                                                             \
            if (B.n_frames > 1) VR_LAUNCH_P2(S, true, true);                                                           \
            else VR_LAUNCH_P2(S, true, false);                                                                         \
        } else {                                                                                                       \
            if (B.n_frames > 1) VR_LAUNCH_P2(S, false, true);                                                          \
            else VR_LAUNCH_P2(S, false, false);                                                                        \
        }                                                                                                              \
    } while (0)
            if constexpr (V == V_VOLUME_MASK) {
                VR_LAUNCH_P2_W(true);  // (the host asks for it only with the brick records in place)
            } else {
                if (skip && L.pw_p2_skip) VR_LAUNCH_P2_W(true);
                else VR_LAUNCH_P2_W(false);
            }
#undef VR_LAUNCH_P2_W
#undef VR_LAUNCH_P2
            return;
        }
    }
    if constexpr (kCanSkip) {
        if (skip) {
            if (L.off32) VR_LAUNCH_PW_T(true, true);
            else VR_LAUNCH_PW_T(false, true);
            return;
        }
    }
    if (L.off32) VR_LAUNCH_PW_T(true, false);
    else VR_LAUNCH_PW_T(false, false);
#undef VR_LAUNCH_PW_T
#undef VR_LAUNCH_PW
}

#if VR_EXPERIMENTAL_FLAVOURS
// lanes per ray per packet (vr_mixed.h)
template <int V>
void launch_mixed(const LaunchDesc& L, hipStream_t s, const MarchBatch& B)
{
    constexpr bool kCanSkip = (V == V_BASIC || V == V_LIGHT || V == V_THREE_FILES || V == V_VOLUME_MASK);
    if constexpr (kCanSkip) {
        if (B.frame[0].brick_dist) {
            if (L.off32) hipLaunchKernelGGL((march_mixed_kernel<V, true, true>), L.grid, L.block, 0, s, B, L.mixed_items, L.n_logical);
            else hipLaunchKernelGGL((march_mixed_kernel<V, false, true>), L.grid, L.block, 0, s, B, L.mixed_items, L.n_logical);
            return;
        }
    }
    if (L.off32) hipLaunchKernelGGL((march_mixed_kernel<V, true, false>), L.grid, L.block, 0, s, B, L.mixed_items, L.n_logical);
    else hipLaunchKernelGGL((march_mixed_kernel<V, false, false>), L.grid, L.block, 0, s, B, L.mixed_items, L.n_logical);
}
#endif

void launch_march(const LaunchDesc& L, hipStream_t s, const MarchBatch& B)
{
    const int variant = L.variant;
    if (L.lt) {  // LDS tiles (vr_lt.h): lit shader
        const bool skip = B.frame[0].brick_dist != nullptr;
        if (L.off32) {
            if (skip) hipLaunchKernelGGL((march_lt_kernel<true, true>), L.grid, L.block, 0, s, B);
            else hipLaunchKernelGGL((march_lt_kernel<true, false>), L.grid, L.block, 0, s, B);
        } else {
            if (skip) hipLaunchKernelGGL((march_lt_kernel<false, true>), L.grid, L.block, 0, s, B);
            else hipLaunchKernelGGL((march_lt_kernel<false, false>), L.grid, L.block, 0, s, B);
        }
        return;
    }
#if VR_EXPERIMENTAL_FLAVOURS
    if (L.mixed_items) {
        switch (variant) {
        case VR_VARIANT_BASIC: launch_mixed<V_BASIC>(L, s, B); break;
        case VR_VARIANT_LIGHT: launch_mixed<V_LIGHT>(L, s, B); break;
        case VR_VARIANT_VOLUME_MASK: launch_mixed<V_VOLUME_MASK>(L, s, B); break;
        case VR_VARIANT_THREE_FILES: launch_mixed<V_THREE_FILES>(L, s, B); break;
        case VR_VARIANT_MULTI_CTRT: launch_mixed<V_MULTI_CTRT>(L, s, B); break;
        case VR_VARIANT_TF_CALIB: launch_mixed<V_TF_CALIB>(L, s, B); break;
        default: break;  // (enqueue_render never asks: these shaders have no depth-parallel form)
        }
        return;
    }
#endif
    if (L.pw) {
        switch (variant) {
        case VR_VARIANT_BASIC: launch_pw<V_BASIC>(L, s, B); break;
        case VR_VARIANT_LIGHT: launch_pw<V_LIGHT>(L, s, B); break;
        case VR_VARIANT_VOLUME_MASK: launch_pw<V_VOLUME_MASK>(L, s, B); break;
        case VR_VARIANT_THREE_FILES: launch_pw<V_THREE_FILES>(L, s, B); break;
        case VR_VARIANT_MULTI_CTRT: launch_pw<V_MULTI_CTRT>(L, s, B); break;
        case VR_VARIANT_ILLUSTRATIVE: launch_pw<V_ILLUSTRATIVE>(L, s, B); break;
        case VR_VARIANT_LIGHT_INSHADER: launch_pw<V_LIGHT_INSHADER>(L, s, B); break;
        default: launch_pw<V_TF_CALIB>(L, s, B); break;
        }
        return;
    }
    const bool off32 = L.off32, dp_pipe = L.dp_pipe, otf = L.otf;
    (void)otf;  // (used with VR_EXPERIMENTAL_FLAVOURS only)
    const int leap_mode = L.leap_mode, dp = L.dp;
    const dim3 grid = L.grid, block = L.block;
#if !VR_FUSED && VR_EXPERIMENTAL_FLAVOURS
        if (L.wtb) {
            if (B.frame[0].brick_dist) {
                if (off32) hipLaunchKernelGGL((march_wtb_light_kernel<true, true>), grid, dim3(256), 0, s, B);
                else hipLaunchKernelGGL((march_wtb_light_kernel<false, true>), grid, dim3(256), 0, s, B);
            } else {
                if (off32) hipLaunchKernelGGL((march_wtb_light_kernel<true, false>), grid, dim3(256), 0, s, B);
                else hipLaunchKernelGGL((march_wtb_light_kernel<false, false>), grid, dim3(256), 0, s, B);
            }
        } else
#endif
        if (dp == 4) {
            switch (variant) {
            case VR_VARIANT_BASIC: launch_dp<V_BASIC, 4, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_LIGHT:
                if (dp_pipe) launch_dp<V_LIGHT, 4, true>(off32, grid, block, s, B);
                else launch_dp<V_LIGHT, 4, false>(off32, grid, block, s, B);
                break;
            case VR_VARIANT_VOLUME_MASK: launch_dp<V_VOLUME_MASK, 4, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_THREE_FILES: launch_dp<V_THREE_FILES, 4, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_MULTI_CTRT: launch_dp<V_MULTI_CTRT, 4, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_TF_CALIB: launch_dp<V_TF_CALIB, 4, false>(off32, grid, block, s, B); break;
            default: break;  // (no depth-parallel form of this shader: enqueue_render never asks for one)
            }
        } else if (dp == 2) {
            switch (variant) {
            case VR_VARIANT_BASIC: launch_dp<V_BASIC, 2, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_LIGHT:
                if (dp_pipe) launch_dp<V_LIGHT, 2, true>(off32, grid, block, s, B);
                else launch_dp<V_LIGHT, 2, false>(off32, grid, block, s, B);
                break;
            case VR_VARIANT_VOLUME_MASK: launch_dp<V_VOLUME_MASK, 2, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_THREE_FILES: launch_dp<V_THREE_FILES, 2, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_MULTI_CTRT: launch_dp<V_MULTI_CTRT, 2, false>(off32, grid, block, s, B); break;
            case VR_VARIANT_TF_CALIB: launch_dp<V_TF_CALIB, 2, false>(off32, grid, block, s, B); break;
            default: break;  // (no depth-parallel form of this shader: enqueue_render never asks for one)
            }
        } else
        switch (variant) {
        case VR_VARIANT_BASIC: launch_variant<V_BASIC>(off32, leap_mode, grid, block, s, B, L.lds_bytes); break;
        case VR_VARIANT_LIGHT:
#if VR_EXPERIMENTAL_FLAVOURS
            if (otf) launch_variant<V_LIGHT, true>(off32, leap_mode, grid, block, s, B);
            else
#endif
                launch_variant<V_LIGHT>(off32, leap_mode, grid, block, s, B, L.lds_bytes);
            break;
        case VR_VARIANT_VOLUME_MASK: launch_variant<V_VOLUME_MASK>(off32, leap_mode, grid, block, s, B); break;
        case VR_VARIANT_THREE_FILES: launch_variant<V_THREE_FILES>(off32, leap_mode, grid, block, s, B); break;
        case VR_VARIANT_MULTI_CTRT: launch_variant<V_MULTI_CTRT>(off32, leap_mode, grid, block, s, B); break;
        case VR_VARIANT_ILLUSTRATIVE: launch_variant<V_ILLUSTRATIVE>(off32, leap_mode, grid, block, s, B); break;
        case VR_VARIANT_LIGHT_INSHADER: launch_variant<V_LIGHT_INSHADER>(off32, leap_mode, grid, block, s, B, L.lds_bytes); break;
        default: launch_variant<V_TF_CALIB>(off32, leap_mode, grid, block, s, B); break;
        }
}

}  // namespace VR_KNS
