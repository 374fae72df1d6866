// Application -- the render-call surface: owns the camera and the per-frame uniforms, hands them to the
// ray-marcher once per frame and asks the active scene to render.  Mirrors med::Application
// (App/src/Application.h / Application.cpp: OnStart :58-94, OnUpdate :96-119, OnRender :121-239,
// OnResize :299-323) without the window, the event queue and the ImGui layer.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "Camera.h"
#include "MiniApp.h"
#include "vr.h"

namespace med {

class Application {
public:
    Application(uint32_t width = 1280, uint32_t height = 720, int device = 0);  // Application.h:100-104
    ~Application();
    Application(const Application&) = delete;
    Application& operator=(const Application&) = delete;

    bool Ok() const { return p_Ctx != nullptr; }
    const std::string& LastError() const { return m_Error; }

    int OnStart(std::unique_ptr<MiniApp> app);   // scene selection + "use the MiniApp's step parameters" (:69-84)
    int OnUpdate();                              // rewrites every uniform (:96-119) and lets the scene re-upload TFs
    int OnRender();                              // ray end pass + volume pass
    int OnFrame() { int rc = OnUpdate(); return rc != VR_OK ? rc : OnRender(); }
    int OnResize(uint32_t width, uint32_t height);

    Camera& GetCamera() { return m_Camera; }
    MiniApp* GetApp() { return p_App.get(); }
    vr_ctx* GetContext() { return p_Ctx; }
    const vr_uniforms& GetUniforms() const { return m_Uniforms; }

    // what the ImGui sliders edit (Application.cpp:243-272)
    int m_FragmentMode = 0;
    int m_StepsCount = 200;
    float m_StepSize = 0.01f;
    vrm::vec2 m_ClipsX{}, m_ClipsY{}, m_ClipsZ{};
    bool m_BToggles[4] = {false, false, false, false};  // (variable step size, jitter, -, -)
    bool m_PrepareOnDevice = false;  // forwarded to the scene in OnStart (MiniApp::SetPrepareOnDevice)

    // read back the fragment output / the presented BGRA8 frame of the last OnRender
    int ReadFrame(float* frag_rgba, uint8_t* present_bgra8 = nullptr, uint64_t* samples = nullptr);

    uint32_t Width() const { return m_Width; }
    uint32_t Height() const { return m_Height; }

private:
    uint32_t m_Width, m_Height;
    Camera m_Camera;
    vr_ctx* p_Ctx = nullptr;
    std::unique_ptr<MiniApp> p_App;
    vr_uniforms m_Uniforms{};
    std::string m_Error;
};

}  // namespace med
