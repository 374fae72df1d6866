// MiniApp -- one scene = one fragment-shader variant + the data it binds.  Mirrors med::MiniApp
// (App/src/miniapps/include/MiniApp.h:36-63).  The WebGPU PipelineBuilder / RenderPassEncoder arguments are
// replaced by the vr_ctx the scene uploads to and renders with; the ImGui tab is dropped.
#pragma once
#include <cmath>
#include <memory>
#include <tuple>

#include "ColorTf.h"
#include "Light.h"
#include "OpacityTf.h"
#include "VolumeFile.h"
#include "vr.h"

namespace med {

class MiniApp {
public:
    virtual ~MiniApp() = default;
    virtual void OnStart(vr_ctx* ctx) = 0;   // data prep + uploads (OnStart(PipelineBuilder&))
    virtual void OnUpdate() = 0;             // TF re-upload when edited (OnUpdate(Timestep))
    virtual int OnRender(vr_ctx* ctx) { return vr_render(ctx, Variant()); }  // binds group 1 + draws
    virtual void OnEnd() {}
    virtual int Variant() const = 0;         // which WGSL shader the scene attaches
    virtual const Light* GetLight() const { return nullptr; }
    // first failure of an upload / device-preparation call made by OnStart (VR_OK if none): Application::OnStart returns it
    int StartStatus() const { return m_Status; }

    void ComputeRecommendedSteppingParams(const VolumeFile& file)
    {
        auto [x, y, z] = file.GetSize();
        int max = std::max<int>(x, std::max<int>(y, z));
        m_StepSize = 1.0f / static_cast<float>(max);
        // the longest ray through a unit cube is sqrt(3): with step 1/max that many steps cover it
        m_StepsCount = static_cast<int>(std::sqrt(3) * max);
    }
    // Data preparation on the GPU instead of the reference's single-threaded CPU loops (same arithmetic, bit for
    // bit): the scene uploads the un-prepared voxels and runs vr_volume_normalize / vr_volume_precompute_gradient
    // in the order its OnStart uses.  The host VolumeFile is left as loaded.
    void SetPrepareOnDevice(bool on) { m_PrepareOnDevice = on; }
    float GetStepSize() const { return m_StepSize; }
    int GetStepsCount() const { return m_StepsCount; }
    std::tuple<float, float, float> GetBBoxSize() const { return m_BBoxSize; }

protected:
    // records the first failing status of OnStart's calls into the ray-marcher
    int Check(int rc)
    {
        if (rc != VR_OK && m_Status == VR_OK) m_Status = rc;
        return rc;
    }
    int Upload(vr_ctx* ctx, int slot, const VolumeFile& f)
    {
        auto [x, y, z] = f.GetSize();
        // vr_volume_upload copies x*y*z vec4 from the pointer: a file whose declared size exceeds its data must not get there
        if (f.GetVecReference().size() != static_cast<size_t>(x) * y * z) return Check(VR_ERR_INVALID_ARG);
        return Check(vr_volume_upload(ctx, slot, static_cast<const float*>(f.GetVoidPtr()), x, y, z));
    }
    int m_Status = VR_OK;
    bool m_PrepareOnDevice = false;
    float m_StepSize = 0.0f;
    int m_StepsCount = 0;
    std::tuple<float, float, float> m_BBoxSize = {0.0f, 0.0f, 0.0f};
};

using VolumePtr = std::shared_ptr<VolumeFile>;

// App/src/miniapps/BasicVolumeApp.cpp:74-93 (DemoBasic): normalise, TF 256, unlit shader.
class BasicVolumeApp : public MiniApp {
public:
    explicit BasicVolumeApp(VolumePtr ct, int tfResolution = 256) : p_Ct(std::move(ct)), m_TfRes(tfResolution) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return VR_VARIANT_BASIC; }
    std::unique_ptr<OpacityTF> p_OpacityTf;
    std::unique_ptr<ColorTF> p_ColorTf;
private:
    VolumePtr p_Ct;
    int m_TfRes;
};

// App/src/miniapps/BasicVolLightApp.cpp:12-51: normalise -> gradient -> AverageGradient(5) (no-op), TF 4096,
// light (0,5,0) / ambient .1 / diffuse 1 (BasicVolLightApp.h:32-37).
class BasicVolLightApp : public MiniApp {
public:
    explicit BasicVolLightApp(VolumePtr ct, int tfResolution = 4096) : p_Ct(std::move(ct)), m_TfRes(tfResolution) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return m_InShaderGradient ? VR_VARIANT_LIGHT_INSHADER : VR_VARIANT_LIGHT; }
    // renders with the shader's own ComputeGradient (BasicVolLightApp.wgsl:239-253), the call the reference keeps
    // commented out at :212, instead of the voxels' pre-computed .rgb
    void SetInShaderGradient(bool on) { m_InShaderGradient = on; }
    const Light* GetLight() const override { return &m_Light1; }
    std::unique_ptr<OpacityTF> p_OpacityTf;
    std::unique_ptr<ColorTF> p_ColorTf;
private:
    VolumePtr p_Ct;
    int m_TfRes;
    bool m_InShaderGradient = false;
    Light m_Light1{vrm::vec4(0.0f, 5.0f, 0.0f, 1.0f), vrm::vec4(0.1f), vrm::vec4(1.0f)};
};

// App/src/miniapps/VolumeMaskApp.cpp:12-65: CT gradient(true) BEFORE normalisation, TF 256 (CT) / 4096 (RT),
// bind order mask, RT, CT.
class VolumeMaskApp : public MiniApp {
public:
    VolumeMaskApp(VolumePtr mask, VolumePtr rt, VolumePtr ct) : p_Mask(std::move(mask)), p_Rt(std::move(rt)), p_Ct(std::move(ct)) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return VR_VARIANT_VOLUME_MASK; }
    std::unique_ptr<OpacityTF> p_OpacityTfCT, p_OpacityTfRT;
    std::unique_ptr<ColorTF> p_ColorTfCT, p_ColorTfRT;
private:
    VolumePtr p_Mask, p_Rt, p_Ct;
};

// App/src/miniapps/ThreeFilesApp.cpp:9-44: no normalisation / gradient at all, TF 256 x4, light (5,5,-5).
class ThreeFilesApp : public MiniApp {
public:
    ThreeFilesApp(VolumePtr ct, VolumePtr rt, VolumePtr mask) : p_Ct(std::move(ct)), p_Rt(std::move(rt)), p_Mask(std::move(mask)) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return VR_VARIANT_THREE_FILES; }
    const Light* GetLight() const override { return &m_Light1; }
    std::unique_ptr<OpacityTF> p_OpacityTfCT, p_OpacityTfRT;
    std::unique_ptr<ColorTF> p_ColorTfCT, p_ColorTfRT;
private:
    VolumePtr p_Ct, p_Rt, p_Mask;
    Light m_Light1{vrm::vec4(5.0f, 5.0f, -5.0f, 1.0f), vrm::vec4(0.1f), vrm::vec4(1.0f)};
};

// App/src/miniapps/MutliCTRTApp.cpp:12-69: CT gradient(true) -> normalise both, TF 1024 x4, light (5,5,-5).
class MultiCTRTApp : public MiniApp {
public:
    MultiCTRTApp(VolumePtr ct, VolumePtr rt) : p_Ct(std::move(ct)), p_Rt(std::move(rt)) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return m_Illustrative ? VR_VARIANT_ILLUSTRATIVE : VR_VARIANT_MULTI_CTRT; }
    // renders with MutliCTRTIllustrative.wgsl, the module the reference's IntializePipeline compiles next to
    // MultiCTRTApp.wgsl but never attaches (MutliCTRTApp.cpp:112-119)
    void SetIllustrative(bool on) { m_Illustrative = on; }
    const Light* GetLight() const override { return &m_Light1; }
    std::unique_ptr<OpacityTF> p_OpacityTfCT, p_OpacityTfRT;
    std::unique_ptr<ColorTF> p_ColorTfCT, p_ColorTfRT;
private:
    VolumePtr p_Ct, p_Rt;
    bool m_Illustrative = false;
    Light m_Light1{vrm::vec4(5.0f, 5.0f, -5.0f, 1.0f), vrm::vec4(0.1f), vrm::vec4(1.0f)};
};

// App/src/miniapps/TFCalibrationApp.cpp:10-42: TF resolution = CT max value, opacity TF calibrated on the
// filled mask BEFORE normalisation; the un-filled mask is what the shader samples (nearest).
class TFCalibrationApp : public MiniApp {
public:
    TFCalibrationApp(VolumePtr ct, VolumePtr maskFilled, VolumePtr maskNoFill)
        : p_Ct(std::move(ct)), p_MaskFilled(std::move(maskFilled)), p_MaskNoFill(std::move(maskNoFill)) {}
    void OnStart(vr_ctx* ctx) override;
    void OnUpdate() override;
    int Variant() const override { return VR_VARIANT_TF_CALIB; }
    std::unique_ptr<OpacityTF> p_OpacityTfCT;
    std::unique_ptr<ColorTF> p_ColorTfCT;
private:
    VolumePtr p_Ct, p_MaskFilled, p_MaskNoFill;
};

}  // namespace med
