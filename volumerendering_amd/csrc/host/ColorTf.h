// ColorTF -- 1-D colour transfer function (RGBA32Float[R]).  Mirrors med::ColorTF
// (App/src/tf/ColorTf.h:12-34, ColorTf.cpp) minus the ImPlot/ImGui colour-picker editor.
#pragma once
#include "TransferFunction.h"

namespace med {

class ColorTF : public TransferFunction {
public:
    explicit ColorTF(int desiredTfResolution);

    std::string GetType() const override { return "color"; }
    void UpdateTexture() override;
    bool Save(const std::string& name) override;   // ColorTf.cpp:181-199
    void Load(const std::string& name, TFLoadOption option = TFLoadOption::NONE) override;  // :201-312
    void ResetTF() override;                        // :27-42

    // editor surface without ImGui: add a coloured control point / recolour / move an existing one
    int AddColorControlPoint(double x, vrm::vec4 color);
    void SetControlColor(int cpId, vrm::vec4 color);
    void SetControlPointX(int cpId, double x);

    const std::vector<vrm::vec4>& GetColors() const { return m_Colors; }
    const std::vector<vrm::vec4>& GetControlColors() const { return m_ControlCol; }

private:
    void UpdateYAxis(int cpId) override;  // :315-348

    std::vector<vrm::vec4> m_Colors{};
    std::vector<vrm::vec4> m_ControlCol{};
};

}  // namespace med
