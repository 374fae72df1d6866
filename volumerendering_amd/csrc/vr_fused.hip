// vr_fused.hip -- the march kernels once more, with the per-sample multiply-adds fused (VR_FUSED = 1, namespace vrf):
// texture coordinates p * N - 0.5, every linear-filter lerp a + (b - a) * t, dot products, the shading sum, the CT / RT
// colour mix and FrontToBackBlend are single v_fma_f32 / v_pk_fma_f32 instructions.  WGSL leaves that choice to the
// implementation and the reference's back end emits `mad`; the CPU checker restates both modes.
// Selected per context with vr_set_arithmetic(ctx, VR_ARITH_FUSED); the default stays the separately rounded form.
#define VR_KNS vrf
#define VR_FUSED 1
#include "vr_launch.h"
