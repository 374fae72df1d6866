#include "TransferFunction.h"

#include <algorithm>
#include <cmath>

namespace med {

void TransferFunction::ResolveResolution(int resolution)
{
    m_TextureResolution = resolution;
    const int maxTex1Dsize = GetMaxTextureResolution();
    if (maxTex1Dsize < m_TextureResolution || m_TextureResolution <= 0) m_TextureResolution = maxTex1Dsize;
}

vrm::dvec2 TransferFunction::RemapCP(vrm::dvec2 cp, int dataRange, int tfResolution)
{
    tfResolution -= 1;  // indexed from 0
    const int currentDataRange = GetDataRange();
    if (currentDataRange == 0) return cp;  // cannot be remapped
    const double textureCoordinate = cp.x / tfResolution;           // to [0,1]
    const double oldDensityValue = textureCoordinate * dataRange;   // density the cp sat on in its own dataset
    if (oldDensityValue > currentDataRange) return {-1.0, -1.0};    // outside this dataset's range: clip
    const int newX = static_cast<int>((oldDensityValue / currentDataRange) * tfResolution);
    return {static_cast<double>(newX), cp.y};
}

std::vector<vrm::dvec2> TransferFunction::RemapCPVector(std::vector<vrm::dvec2> cps, int dataRange, int tfResolution)
{
    std::vector<vrm::dvec2> result;
    auto exists = [&result](double x) {
        return std::any_of(result.begin(), result.end(), [x](const vrm::dvec2& p) { return p.x == x; });
    };
    for (auto cp : cps) {
        auto n = RemapCP(cp, dataRange, tfResolution);
        if (n.x == -1 && n.y == -1) continue;  // clipped
        if (exists(n.x)) continue;             // merged with an earlier point: first one wins
        result.push_back(n);
    }
    if (!exists(tfResolution - 1)) result.push_back({static_cast<double>(tfResolution - 1), 1.0});
    return result;
}

int TransferFunction::AddControlPoint(double mouseX, double mouseY, bool updateOnAdd)
{
    constexpr int CONTROL_POINT_EXISTS = -1;
    auto before = [](const vrm::dvec2& p, double x) { return p.x < x; };
    const double x = std::round(mouseX);  // x always snaps to an integer texel
    auto it = std::lower_bound(m_ControlPoints.begin(), m_ControlPoints.end(), x, before);
    if (it == m_ControlPoints.end() || x == it->x) return CONTROL_POINT_EXISTS;
    m_ControlPoints.push_back({x, mouseY});
    std::sort(m_ControlPoints.begin(), m_ControlPoints.end(), [](const vrm::dvec2& a, const vrm::dvec2& b) { return a.x < b.x; });
    it = std::lower_bound(m_ControlPoints.begin(), m_ControlPoints.end(), x, before);
    const int cpId = static_cast<int>(it - m_ControlPoints.begin());
    if (updateOnAdd) UpdateYAxis(cpId);
    return cpId;
}

}  // namespace med
