// OpacityTF -- table generation follows App/src/tf/OpacityTf.cpp (default ramp :29-45, control-point re-lerp
// :489-523, preset text format :181-314, mask calibration :316-487, histogram :144-179).
#include "OpacityTf.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "LinearInterpolation.h"
#include "TfUtils.h"

namespace med {

OpacityTF::OpacityTF(int desiredTfResolution)
{
    ResolveResolution(desiredTfResolution);
    m_XPoints.resize(m_TextureResolution, 0.0f);
    m_YPoints.resize(m_TextureResolution, 0.0f);
    ResetTF();
}

void OpacityTF::ResetTF()
{
    m_ControlPoints.clear();
    m_ControlPoints.push_back({0.0, 0.0});
    m_ControlPoints.push_back({m_TextureResolution - 1.0, 1.0});
    const std::vector<float> ramp = LinearInterpolation::Generate(0, m_TextureResolution - 1, 0.0f, 1.0f, 1);
    for (int i = 0; i < m_TextureResolution; ++i) {
        m_XPoints[i] = static_cast<float>(i);
        m_YPoints[i] = ramp[i];
    }
    m_ShouldUpdate = true;
}

void OpacityTF::UpdateTexture()
{
    if (!m_ShouldUpdate) return;
    if (p_Ctx) vr_tf_upload_opacity(p_Ctx, m_Slot, m_YPoints.data(), static_cast<uint32_t>(m_TextureResolution));
    m_ShouldUpdate = false;
}

void OpacityTF::ActivateHistogram(const VolumeFile& file)
{
    m_Histogram.assign(m_TextureResolution, 0.0f);
    const auto& data = file.GetVecReference();
    auto [xSize, ySize, slices] = file.GetSize();
    const size_t size = static_cast<size_t>(xSize) * ySize * slices;
    // integer division, as in the reference (:153): resolution / data range
    const float factor = static_cast<float>(m_TextureResolution / std::max<size_t>(file.GetDataRange(), 1));
    for (size_t i = 0; i < size; ++i) {
        int value = file.IsNormalized() ? static_cast<int>(data[i].a * m_TextureResolution) : static_cast<int>(data[i].a * factor);
        value = std::clamp(value, 0, m_TextureResolution - 1);
        ++m_Histogram[value];
    }
    const float maxVal = std::log10(static_cast<float>(size));
    for (float& h : m_Histogram)
        if (h != 0.0f) h = std::log10(h) / maxVal;
}

bool OpacityTF::Save(const std::string& name)
{
    std::ofstream file(name);
    if (!file) return false;
    file << GetType() << "\n"
         << "resolution\n" << GetTextureResolution() << "\n"
         << "data range\n" << GetDataRange() << "\n"
         << "control points number\n" << m_ControlPoints.size() << "\n";
    for (const auto& cp : m_ControlPoints) file << cp.x << " " << cp.y << "\n";
    return true;
}

namespace {
// reads "<label>\n<int>\n"; false on a format error
bool read_labeled_int(std::ifstream& f, const char* label, int& out)
{
    std::string line;
    if (!std::getline(f, line) || line != label) return false;
    if (!std::getline(f, line)) return false;
    char* end = nullptr;
    long v = std::strtol(line.c_str(), &end, 10);
    if (end == line.c_str()) return false;
    out = static_cast<int>(v);
    return true;
}
}  // namespace

void OpacityTF::Load(const std::string& name, TFLoadOption option)
{
    std::ifstream file(name);
    std::string line;
    if (!std::getline(file, line) || line != GetType()) return;  // "Invalid transform function format"
    int resolution = 0, dataRange = 0, count = 0;
    if (!read_labeled_int(file, "resolution", resolution)) return;
    if (!read_labeled_int(file, "data range", dataRange)) return;
    if (!read_labeled_int(file, "control points number", count)) return;
    std::vector<vrm::dvec2> cps;
    for (int i = 0; i < count; ++i) {
        if (!std::getline(file, line)) return;  // unexpected end of file: TF not loaded
        std::stringstream ls(line);
        double x = 0, y = 0;
        ls >> x >> y;
        cps.push_back({x, y});
    }
    if (option == TFLoadOption::RESCALE_TO_NEW_RANGE) cps = RemapCPVector(cps, dataRange, resolution);

    ResolveResolution(resolution);
    m_XPoints.resize(m_TextureResolution, 0.0f);
    m_YPoints.resize(m_TextureResolution, 0.0f);
    ResetTF();
    m_ControlPoints = std::move(cps);
    for (int i = 0; i < static_cast<int>(m_ControlPoints.size()); ++i) UpdateYAxis(i);
    m_ShouldUpdate = true;
}

void OpacityTF::CalibrateOnMask(std::shared_ptr<const VolumeFile> mask, std::shared_ptr<const VolumeFile> file,
                                std::array<int, 4> activeContours)
{
    if (!file || !mask) return;
    if (mask->GetSize() != file->GetSize()) return;
    auto [x, y, z] = mask->GetSize();
    const size_t size = static_cast<size_t>(x) * y * z;
    if (size == 0) return;
    const auto& maskData = mask->GetVecReference();
    const auto& fileData = file->GetVecReference();
    size_t maxValue = file->GetMaxNumber();
    if (maxValue == 0) {
        maxValue = file->GetMaxNumber(fileData, 3);
        if (maxValue == 0) return;
    }
    std::vector<int> contours;
    for (int i = 0; i < 4; ++i)
        if (activeContours[i] == 1) contours.push_back(i);
    if (contours.empty()) return;

    // histogram of the raw densities inside the selected contours (must run before NormalizeData)
    std::vector<double> bin(maxValue, 0.0);
    for (size_t i = 0; i < size; ++i)
        for (int c : contours)
            if (maskData[i][c] != 0) {
                size_t value = static_cast<size_t>(static_cast<int>(fileData[i].a));
                if (value < bin.size()) ++bin[value];
            }
    const int maxElem = static_cast<int>(*std::max_element(bin.begin(), bin.end()));
    if (maxElem == 0) return;

    std::vector<vrm::dvec2> cps;
    auto exists = [&cps](int px) { return std::any_of(cps.begin(), cps.end(), [px](const vrm::dvec2& p) { return p.x == px; }); };
    const double THRESHOLD_FOR_POINT = 0.6;
    int first = -1, last = -1;
    // every maximal run of bins whose relative frequency is >= the threshold becomes one or two control points
    for (int i = 0; i < static_cast<int>(maxValue); ++i) {
        const double freq = bin[i] / maxElem;
        if (freq >= THRESHOLD_FOR_POINT) {
            if (first == -1) first = i;
            last = i;
        } else if (first != -1) {
            const int cpFirst = static_cast<int>((static_cast<double>(first) / file->GetMaxNumber()) * GetTextureResolution());
            const int cpLast = static_cast<int>((static_cast<double>(last) / file->GetMaxNumber()) * GetTextureResolution());
            if (!exists(cpFirst)) cps.push_back({static_cast<double>(cpFirst), bin[first] / maxElem});
            if (cpFirst != cpLast) cps.push_back({static_cast<double>(cpLast), bin[last] / maxElem});
            first = -1;
        }
    }
    if (cps.empty() || cps.front().x != 0.0) cps.insert(cps.begin(), vrm::dvec2{0.0, 0.0});
    if (cps.back().x != m_TextureResolution - 1) cps.push_back({static_cast<double>(m_TextureResolution - 1), 0.0});

    m_XPoints.resize(m_TextureResolution, 0.0f);
    m_YPoints.resize(m_TextureResolution, 0.0f);
    m_ControlPoints = std::move(cps);
    for (int i = 0; i < static_cast<int>(m_ControlPoints.size()); ++i) UpdateYAxis(i);
    m_ShouldUpdate = true;
}

void OpacityTF::SetControlPoint(int cpId, double x, double y)
{
    if (cpId < 0 || cpId >= static_cast<int>(m_ControlPoints.size())) return;
    m_ControlPoints[cpId] = {x, y};
    TfUtils::CheckDragBounds(cpId, m_ControlPoints, m_TextureResolution);
    UpdateYAxis(cpId);
}

void OpacityTF::UpdateYAxis(int cpId)
{
    if (cpId < 0 || cpId >= static_cast<int>(m_ControlPoints.size())) return;
    // re-lerps the integer span between two control points and copies it into the table
    auto fill = [&](double cx1, double cx2, float cy1, float cy2) {
        const int x0 = static_cast<int>(cx1), x1 = static_cast<int>(cx2);
        const std::vector<float> span = LinearInterpolation::Generate(x0, x1, cy1, cy2, 1);
        for (size_t i = 0; i <= static_cast<size_t>(std::abs(x1 - x0)) && i < span.size(); ++i) m_YPoints[i + x0] = span[i];
    };
    if (cpId - 1 >= 0) {
        const size_t pred = static_cast<size_t>(m_ControlPoints[cpId - 1].x);
        fill(m_ControlPoints[cpId - 1].x, m_ControlPoints[cpId].x, m_YPoints[pred], static_cast<float>(m_ControlPoints[cpId].y));
    }
    if (cpId + 1 < static_cast<int>(m_ControlPoints.size())) {
        const size_t succ = static_cast<size_t>(m_ControlPoints[cpId + 1].x);
        fill(m_ControlPoints[cpId].x, m_ControlPoints[cpId + 1].x, static_cast<float>(m_ControlPoints[cpId].y), m_YPoints[succ]);
    }
    m_ShouldUpdate = true;
}

}  // namespace med
