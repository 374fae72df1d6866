// TransferFunction -- common state of the 1-D opacity / colour transfer functions.
// Mirrors med::TransferFunction (App/src/tf/TransferFunction.h:18-87) minus the ImPlot editor (Render()).
// The WebGPU 1-D texture is replaced by a binding to a TF slot of a vr_ctx; UpdateTexture() pushes the table
// through the C ABI when it changed.
#pragma once
#include <string>
#include <vector>

#include "vr.h"
#include "vrm.h"

namespace med {

enum class TFLoadOption { NONE = 0, RESCALE_TO_NEW_RANGE = 1 };

class TransferFunction {
public:
    virtual ~TransferFunction() = default;
    virtual void UpdateTexture() = 0;
    virtual std::string GetType() const = 0;
    virtual bool Save(const std::string& name) = 0;
    virtual void Load(const std::string& name, TFLoadOption option = TFLoadOption::NONE) = 0;
    virtual void ResetTF() = 0;

    // WebGPU default limit maxTextureDimension1D, which TransferFunction::GetMaxTextureResolution returns in
    // the reference (TransferFunction.cpp:35-38).  Kept so that ResolveResolution clamps identically.
    static constexpr int kMaxTextureDimension1D = 8192;
    int GetMaxTextureResolution() const { return kMaxTextureDimension1D; }
    int GetTextureResolution() const { return m_TextureResolution; }
    int GetDataRange() const { return m_DataRange; }
    void SetDataRange(int range) { m_DataRange = range; }

    vrm::dvec2 RemapCP(vrm::dvec2 cp, int dataRange, int tfResolution);                              // TransferFunction.cpp:55-84
    std::vector<vrm::dvec2> RemapCPVector(std::vector<vrm::dvec2> cps, int dataRange, int tfResolution);  // :86-121
    int AddControlPoint(double mouseX, double mouseY, bool updateOnAdd = true);                      // :123-160

    const std::vector<vrm::dvec2>& GetControlPoints() const { return m_ControlPoints; }
    bool ShouldUpdate() const { return m_ShouldUpdate; }

    // replaces GetTexture(): where UpdateTexture() uploads to
    void BindTexture(vr_ctx* ctx, int slot) { p_Ctx = ctx; m_Slot = slot; m_ShouldUpdate = true; }

protected:
    virtual void UpdateYAxis(int cpId) = 0;
    void ResolveResolution(int resolution);  // TransferFunction.cpp:10-27

    vr_ctx* p_Ctx = nullptr;
    int m_Slot = 0;
    std::vector<vrm::dvec2> m_ControlPoints{};
    int m_TextureResolution = 0;
    int m_DataRange = 0;
    bool m_ShouldUpdate = false;
};

}  // namespace med
