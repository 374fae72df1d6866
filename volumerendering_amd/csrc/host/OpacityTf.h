// OpacityTF -- 1-D opacity transfer function (R32Float[R]).  Mirrors med::OpacityTF
// (App/src/tf/OpacityTf.h:14-67, OpacityTf.cpp) minus the ImPlot editor.
#pragma once
#include <array>
#include <memory>

#include "TransferFunction.h"
#include "VolumeFile.h"

namespace med {

class OpacityTF : public TransferFunction {
public:
    explicit OpacityTF(int desiredTfResolution);  // 0 = maximal resolution

    void UpdateTexture() override;
    void ActivateHistogram(const VolumeFile& file);  // OpacityTf.cpp:144-179
    std::string GetType() const override { return "opacity"; }
    bool Save(const std::string& name) override;     // :181-198
    void Load(const std::string& name, TFLoadOption option = TFLoadOption::NONE) override;  // :200-314
    void ResetTF() override;                          // :29-45
    void CalibrateOnMask(std::shared_ptr<const VolumeFile> mask, std::shared_ptr<const VolumeFile> file,
                         std::array<int, 4> activeContours);  // :316-487

    // editor surface without ImPlot: what a click / drag on the plot does
    void SetControlPoint(int cpId, double x, double y);  // DragPoint + CheckDragBounds + UpdateYAxis (:74-94)

    const std::vector<float>& GetYPoints() const { return m_YPoints; }
    const std::vector<float>& GetHistogram() const { return m_Histogram; }

private:
    void UpdateYAxis(int cpId) override;  // :489-523

    std::vector<float> m_XPoints{};
    std::vector<float> m_YPoints{};
    std::vector<float> m_Histogram{};
};

}  // namespace med
