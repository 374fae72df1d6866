#include "Application.h"

#include <cstring>

namespace med {

Application::Application(uint32_t width, uint32_t height, int device)
    : m_Width(width), m_Height(height),
      m_Camera(Camera::CreatePerspective(vrm::radians(60.0f), static_cast<float>(width) / static_cast<float>(height), 0.01f, 100.0f))
{
    if (vr_create(&p_Ctx, width, height, device) != VR_OK) {
        m_Error = vr_last_error(nullptr);
        p_Ctx = nullptr;
    }
}

Application::~Application()
{
    if (p_App) p_App->OnEnd();
    p_App.reset();
    vr_destroy(p_Ctx);
}

int Application::OnStart(std::unique_ptr<MiniApp> app)
{
    p_App = std::move(app);
    if (!p_Ctx || !p_App) return VR_ERR_NOT_READY;
    p_App->SetPrepareOnDevice(m_PrepareOnDevice);
    p_App->OnStart(p_Ctx);
    if (p_App->StartStatus() != VR_OK) {  // an upload or a device-preparation call failed (out of memory, bad size ...)
        m_Error = vr_last_error(p_Ctx);
        if (m_Error.empty()) m_Error = "scene start failed: volume data does not match its declared size";
        return p_App->StartStatus();
    }
    // "Using MiniApp's required step size / count" (Application.cpp:74-84)
    if (p_App->GetStepSize() != 0.0f) m_StepSize = p_App->GetStepSize();
    if (p_App->GetStepsCount() != 0) m_StepsCount = p_App->GetStepsCount();
    return VR_OK;
}

int Application::OnUpdate()
{
    if (!p_Ctx) return VR_ERR_HIP;
    vr_uniforms& u = m_Uniforms;
    std::memset(&u, 0, sizeof u);
    const vrm::mat4 model(1.0f);  // dummy_model, never rewritten (Application.cpp:489-492)
    std::memcpy(u.model, model.data(), sizeof u.model);
    std::memcpy(u.view, m_Camera.GetViewMatrix().data(), sizeof u.view);
    std::memcpy(u.proj, m_Camera.GetProjectionMatrix().data(), sizeof u.proj);
    std::memcpy(u.view_inv, m_Camera.GetInverseViewMatrix().data(), sizeof u.view_inv);
    std::memcpy(u.proj_inv, m_Camera.GetInverseProjectionMatrix().data(), sizeof u.proj_inv);
    const vrm::vec3 pos = m_Camera.GetPosition();
    u.camera_pos[0] = pos.x; u.camera_pos[1] = pos.y; u.camera_pos[2] = pos.z;
    u.fragment_mode = m_FragmentMode;
    u.steps_count = m_StepsCount;
    u.step_size = m_StepSize;
    u.clip_x[0] = m_ClipsX.x; u.clip_x[1] = m_ClipsX.y;
    u.clip_y[0] = m_ClipsY.x; u.clip_y[1] = m_ClipsY.y;
    u.clip_z[0] = m_ClipsZ.x; u.clip_z[1] = m_ClipsZ.y;
    for (int i = 0; i < 4; ++i) u.toggles[i] = m_BToggles[i] ? 1 : 0;
    if (p_App && p_App->GetLight()) {
        const Light* l = p_App->GetLight();
        std::memcpy(u.light_pos, &l->Position.x, sizeof u.light_pos);
        std::memcpy(u.light_ambient, &l->Ambient.x, sizeof u.light_ambient);
        std::memcpy(u.light_diffuse, &l->Diffuse.x, sizeof u.light_diffuse);
    }
    int rc = vr_set_uniforms(p_Ctx, &u);
    if (rc != VR_OK) { m_Error = vr_last_error(p_Ctx); return rc; }
    if (p_App) p_App->OnUpdate();
    return VR_OK;
}

int Application::OnRender()
{
    if (!p_Ctx || !p_App) return VR_ERR_NOT_READY;
    int rc = p_App->OnRender(p_Ctx);
    if (rc != VR_OK) m_Error = vr_last_error(p_Ctx);
    return rc;
}

int Application::OnResize(uint32_t width, uint32_t height)
{
    // like the reference, the camera's aspect ratio is NOT updated on resize (SetAspectRatio is never called,
    // Application.cpp:299-323); callers that want it call GetCamera().SetAspectRatio themselves
    m_Width = width;
    m_Height = height;
    int rc = vr_resize(p_Ctx, width, height);
    if (rc != VR_OK) m_Error = vr_last_error(p_Ctx);
    return rc;
}

int Application::ReadFrame(float* frag_rgba, uint8_t* present_bgra8, uint64_t* samples)
{
    int rc = vr_download(p_Ctx, frag_rgba, present_bgra8, samples);
    if (rc != VR_OK) m_Error = vr_last_error(p_Ctx);
    return rc;
}

}  // namespace med
