// ColorTF -- table generation follows App/src/tf/ColorTf.cpp (default black->white ramp :27-42, control-point
// re-lerp :315-348, preset text format :181-312).
#include "ColorTf.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "LinearInterpolation.h"
#include "TfUtils.h"

namespace med {

ColorTF::ColorTF(int desiredTfResolution)
{
    ResolveResolution(desiredTfResolution);
    ResetTF();
}

void ColorTF::ResetTF()
{
    const vrm::vec4 black(0.0f, 0.0f, 0.0f, 1.0f), white(1.0f, 1.0f, 1.0f, 1.0f);
    m_Colors = LinearInterpolation::Generate(0, m_TextureResolution - 1, black, white, 1);
    m_ControlCol = {black, white};
    m_ControlPoints.clear();
    m_ControlPoints.push_back({0.0, 0.5});
    m_ControlPoints.push_back({m_TextureResolution - 1.0, 0.5});
    m_ShouldUpdate = true;
}

void ColorTF::UpdateTexture()
{
    if (!m_ShouldUpdate) return;
    if (p_Ctx) vr_tf_upload_color(p_Ctx, m_Slot, &m_Colors[0].x, static_cast<uint32_t>(m_TextureResolution));
    m_ShouldUpdate = false;
}

bool ColorTF::Save(const std::string& name)
{
    std::ofstream file(name);
    if (!file) return false;
    file << GetType() << "\n"
         << "resolution\n" << GetTextureResolution() << "\n"
         << "data range\n" << GetDataRange() << "\n"
         << "control points number\n" << m_ControlPoints.size() << "\n";
    for (size_t i = 0; i < m_ControlCol.size(); ++i)
        file << m_ControlPoints[i].x << " " << m_ControlPoints[i].y << " " << m_ControlCol[i].r << " " << m_ControlCol[i].g
             << " " << m_ControlCol[i].b << " " << m_ControlCol[i].a << "\n";
    return true;
}

void ColorTF::Load(const std::string& name, TFLoadOption /*option*/)
{
    std::ifstream file(name);
    std::string line;
    auto labeled_int = [&](const char* label, int& out) {
        if (!std::getline(file, line) || line != label) return false;
        if (!std::getline(file, line)) return false;
        char* end = nullptr;
        long v = std::strtol(line.c_str(), &end, 10);
        if (end == line.c_str()) return false;
        out = static_cast<int>(v);
        return true;
    };
    if (!std::getline(file, line) || line != GetType()) return;
    int resolution = 0, dataRange = 0, count = 0;
    if (!labeled_int("resolution", resolution) || !labeled_int("data range", dataRange) ||
        !labeled_int("control points number", count))
        return;
    std::vector<vrm::dvec2> pos;
    std::vector<vrm::vec4> col;
    for (int i = 0; i < count; ++i) {
        if (!std::getline(file, line)) return;
        std::istringstream ls(line);
        double x = 0, y = 0;
        vrm::vec4 c;
        ls >> x >> y >> c.r >> c.g >> c.b >> c.a;
        pos.push_back({x, y});
        col.push_back(c);
    }
    m_ControlCol = std::move(col);
    m_ControlPoints = std::move(pos);
    ResolveResolution(resolution);
    m_Colors.resize(m_TextureResolution);
    m_DataRange = dataRange;
    for (int i = 0; i < static_cast<int>(m_ControlPoints.size()); ++i) UpdateYAxis(i);
    m_ShouldUpdate = true;
}

int ColorTF::AddColorControlPoint(double x, vrm::vec4 color)
{
    // TransferFunction::AddControlPoint keeps the positions sorted; the colour list is kept parallel to it
    const int id = AddControlPoint(x, 0.5, /*updateOnAdd=*/false);
    if (id < 0) return id;
    m_ControlCol.insert(m_ControlCol.begin() + id, color);
    UpdateYAxis(id);
    return id;
}

void ColorTF::SetControlColor(int cpId, vrm::vec4 color)
{
    if (cpId < 0 || cpId >= static_cast<int>(m_ControlCol.size())) return;
    m_ControlCol[cpId] = color;
    UpdateYAxis(cpId);
}

void ColorTF::SetControlPointX(int cpId, double x)
{
    if (cpId < 0 || cpId >= static_cast<int>(m_ControlPoints.size())) return;
    m_ControlPoints[cpId].x = x;
    TfUtils::CheckDragBounds(cpId, m_ControlPoints, m_TextureResolution);
    UpdateYAxis(cpId);
}

void ColorTF::UpdateYAxis(int cpId)
{
    if (cpId < 0 || cpId >= static_cast<int>(m_ControlCol.size())) return;
    auto fill = [&](double cx1, double cx2, vrm::vec4 cy1, vrm::vec4 cy2) {
        const int x0 = static_cast<int>(cx1), x1 = static_cast<int>(cx2);
        const std::vector<vrm::vec4> span = LinearInterpolation::Generate(x0, x1, cy1, cy2, 1);
        for (size_t i = 0; i <= static_cast<size_t>(std::abs(x1 - x0)) && i < span.size(); ++i) m_Colors[i + x0] = span[i];
    };
    if (cpId - 1 >= 0) {
        const int pred = static_cast<int>(m_ControlPoints[cpId - 1].x);
        fill(m_ControlPoints[cpId - 1].x, m_ControlPoints[cpId].x, m_Colors[pred], m_ControlCol[cpId]);
    }
    if (cpId + 1 < static_cast<int>(m_ControlCol.size())) {
        const int succ = static_cast<int>(m_ControlPoints[cpId + 1].x);
        fill(m_ControlPoints[cpId].x, m_ControlPoints[cpId + 1].x, m_ControlCol[cpId], m_Colors[succ]);
    }
    m_ShouldUpdate = true;
}

}  // namespace med
