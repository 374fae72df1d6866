// Scene set-up in the reference's own order of operations (each OnStart cites its source).
#include "MiniApp.h"

namespace med {

// ---- BasicVolumeApp::DemoBasic (App/src/miniapps/BasicVolumeApp.cpp:74-93) -------------------------------------
void BasicVolumeApp::OnStart(vr_ctx* ctx)
{
    const bool onDevice = m_PrepareOnDevice && !p_Ct->IsNormalized();
    if (onDevice) {
        // the divisor is the file's cached maximum (for DICOM input LargestPixelValue, which need not be the data maximum),
        // exactly what NormalizeData() below would use
        Upload(ctx, 0, *p_Ct);
        int used = 0;
        if (Check(vr_volume_normalize(ctx, 0, static_cast<int>(p_Ct->GetMaxNumber()), &used)) == VR_OK) p_Ct->SetDeviceNormalization(used);
    } else {
        p_Ct->NormalizeData();
    }
    ComputeRecommendedSteppingParams(*p_Ct);
    p_OpacityTf = std::make_unique<OpacityTF>(m_TfRes);
    p_OpacityTf->SetDataRange(static_cast<int>(p_Ct->GetDataRange()));
    p_ColorTf = std::make_unique<ColorTF>(m_TfRes);
    if (!onDevice) Upload(ctx, 0, *p_Ct);
    p_OpacityTf->BindTexture(ctx, 0);
    p_ColorTf->BindTexture(ctx, 0);
    OnUpdate();
}
void BasicVolumeApp::OnUpdate()
{
    p_OpacityTf->UpdateTexture();
    p_ColorTf->UpdateTexture();
}

// ---- BasicVolLightApp (App/src/miniapps/BasicVolLightApp.cpp:12-51) ---------------------------------------------
void BasicVolLightApp::OnStart(vr_ctx* ctx)
{
    const bool onDevice = m_PrepareOnDevice && !p_Ct->IsNormalized() && !p_Ct->HasGradient();
    if (onDevice) {
        Upload(ctx, 0, *p_Ct);
        int used = 0;
        if (Check(vr_volume_normalize(ctx, 0, static_cast<int>(p_Ct->GetMaxNumber()), &used)) == VR_OK) p_Ct->SetDeviceNormalization(used);
        Check(vr_volume_precompute_gradient(ctx, 0, 0));
    } else {
        p_Ct->NormalizeData();
        p_Ct->PreComputeGradient();
        p_Ct->AverageGradient(5);
    }
    ComputeRecommendedSteppingParams(*p_Ct);
    p_OpacityTf = std::make_unique<OpacityTF>(m_TfRes);
    p_ColorTf = std::make_unique<ColorTF>(m_TfRes);
    p_OpacityTf->SetDataRange(static_cast<int>(p_Ct->GetMaxNumber()));
    if (!onDevice) Upload(ctx, 0, *p_Ct);
    p_OpacityTf->BindTexture(ctx, 0);
    p_ColorTf->BindTexture(ctx, 0);
    OnUpdate();
}
void BasicVolLightApp::OnUpdate()
{
    p_OpacityTf->UpdateTexture();
    p_ColorTf->UpdateTexture();
}

// ---- VolumeMaskApp (App/src/miniapps/VolumeMaskApp.cpp:12-65) ---------------------------------------------------
void VolumeMaskApp::OnStart(vr_ctx* ctx)
{
    p_Ct->PreComputeGradient(true);
    p_OpacityTfCT = std::make_unique<OpacityTF>(256);
    p_ColorTfCT = std::make_unique<ColorTF>(256);
    p_OpacityTfRT = std::make_unique<OpacityTF>(4096);
    p_ColorTfRT = std::make_unique<ColorTF>(4096);
    p_Rt->NormalizeData();
    p_Ct->NormalizeData();
    ComputeRecommendedSteppingParams(*p_Ct);
    p_OpacityTfCT->SetDataRange(static_cast<int>(p_Ct->GetDataRange()));
    p_OpacityTfRT->SetDataRange(static_cast<int>(p_Rt->GetDataRange()));
    Upload(ctx, 0, *p_Mask);
    Upload(ctx, 1, *p_Rt);
    Upload(ctx, 2, *p_Ct);
    p_OpacityTfCT->BindTexture(ctx, 0);
    p_ColorTfCT->BindTexture(ctx, 0);
    p_OpacityTfRT->BindTexture(ctx, 1);
    p_ColorTfRT->BindTexture(ctx, 1);
    OnUpdate();
}
void VolumeMaskApp::OnUpdate()
{
    p_OpacityTfCT->UpdateTexture();
    p_OpacityTfRT->UpdateTexture();
    p_ColorTfCT->UpdateTexture();
    p_ColorTfRT->UpdateTexture();
}

// ---- ThreeFilesApp (App/src/miniapps/ThreeFilesApp.cpp:9-44) ----------------------------------------------------
void ThreeFilesApp::OnStart(vr_ctx* ctx)
{
    // the reference neither normalises nor computes gradients here and leaves the Application's default
    // step size / count (0.01 / 200) in place: m_StepSize / m_StepsCount stay 0
    p_OpacityTfCT = std::make_unique<OpacityTF>(256);
    p_OpacityTfRT = std::make_unique<OpacityTF>(256);
    p_ColorTfCT = std::make_unique<ColorTF>(256);
    p_ColorTfRT = std::make_unique<ColorTF>(256);
    Upload(ctx, 0, *p_Ct);
    Upload(ctx, 1, *p_Rt);
    Upload(ctx, 2, *p_Mask);
    p_OpacityTfCT->BindTexture(ctx, 0);
    p_ColorTfCT->BindTexture(ctx, 0);
    p_OpacityTfRT->BindTexture(ctx, 1);
    p_ColorTfRT->BindTexture(ctx, 1);
    OnUpdate();
}
void ThreeFilesApp::OnUpdate()
{
    p_OpacityTfCT->UpdateTexture();
    p_OpacityTfRT->UpdateTexture();
    p_ColorTfCT->UpdateTexture();
    p_ColorTfRT->UpdateTexture();
}

// ---- MultiCTRTApp (App/src/miniapps/MutliCTRTApp.cpp:12-69) -----------------------------------------------------
void MultiCTRTApp::OnStart(vr_ctx* ctx)
{
    p_Ct->PreComputeGradient(true);
    p_Ct->NormalizeData();
    p_Rt->NormalizeData();
    p_OpacityTfCT = std::make_unique<OpacityTF>(1024);
    p_OpacityTfRT = std::make_unique<OpacityTF>(1024);
    p_ColorTfCT = std::make_unique<ColorTF>(1024);
    p_ColorTfRT = std::make_unique<ColorTF>(1024);
    Upload(ctx, 0, *p_Ct);
    Upload(ctx, 1, *p_Rt);
    p_OpacityTfCT->BindTexture(ctx, 0);
    p_ColorTfCT->BindTexture(ctx, 0);
    p_OpacityTfRT->BindTexture(ctx, 1);
    p_ColorTfRT->BindTexture(ctx, 1);
    OnUpdate();
}
void MultiCTRTApp::OnUpdate()
{
    p_OpacityTfCT->UpdateTexture();
    p_OpacityTfRT->UpdateTexture();
    p_ColorTfCT->UpdateTexture();
    p_ColorTfRT->UpdateTexture();
}

// ---- TFCalibrationApp (App/src/miniapps/TFCalibrationApp.cpp:10-42) ---------------------------------------------
void TFCalibrationApp::OnStart(vr_ctx* ctx)
{
    const int res = static_cast<int>(p_Ct->GetMaxNumber());
    p_OpacityTfCT = std::make_unique<OpacityTF>(res);
    p_ColorTfCT = std::make_unique<ColorTF>(res);
    p_OpacityTfCT->CalibrateOnMask(p_MaskFilled, p_Ct, {1, 0, 0, 0});  // must run before normalisation
    p_Ct->NormalizeData();
    Upload(ctx, 0, *p_Ct);
    Upload(ctx, 1, *p_MaskNoFill);
    p_OpacityTfCT->BindTexture(ctx, 0);
    p_ColorTfCT->BindTexture(ctx, 0);
    OnUpdate();
}
void TFCalibrationApp::OnUpdate()
{
    p_OpacityTfCT->UpdateTexture();
    p_ColorTfCT->UpdateTexture();
}

}  // namespace med
