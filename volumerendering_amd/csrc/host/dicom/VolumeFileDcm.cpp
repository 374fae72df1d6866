#include "VolumeFileDcm.h"

#include <algorithm>
#include <bit>

namespace med {

VolumeFileDcm::VolumeFileDcm(std::filesystem::path path, Size size, FileDataType type, DicomVolumeParams params,
                             std::vector<vrm::vec4>& data)
    : VolumeFile(std::move(path), size, type, data, params.LargestPixelValue), m_Params(std::move(params))
{
    InitializeTransformMatrices();
    CalcMainAxis();
    m_CustomBitWidth = static_cast<int>(std::bit_width(m_MaxNumber));  // FileSystem::GetMaxUsedBits
}

bool VolumeFileDcm::CompareFrameOfReference(const IDicomFile& other) const
{
    return other.GetBaseParams().FrameOfReference == m_Params.FrameOfReference;
}

bool VolumeFileDcm::CompareOrientation(const VolumeFileDcm& other) const
{
    const auto o = other.GetVolumeParams().ImageOrientationPatient;
    int res = 0;  // integer accumulation, as in the reference
    for (size_t i = 0; i < o.size(); ++i) res += static_cast<int>(o[i] * m_Params.ImageOrientationPatient[i]);
    return res != 0;
}

std::tuple<float, float, float> VolumeFileDcm::GetBBOXSize() const
{
    auto [x, y, z] = GetSize();
    const double xMM = (x + 1) * m_Params.PixelSpacing[0];
    const double yMM = (y + 1) * m_Params.PixelSpacing[1];
    const double zMM = z * m_Params.SliceThickness;
    const double mx = std::max(xMM, std::max(yMM, zMM));
    return {RoundTo2Dec(static_cast<float>(xMM / mx)), RoundTo2Dec(static_cast<float>(yMM / mx)),
            RoundTo2Dec(static_cast<float>(zMM / mx))};
}

namespace {
inline void mul(const vrm::mat4& m, float x, float y, float z, float w, float out[4])
{
    for (int r = 0; r < 4; ++r) out[r] = m.c[0][r] * x + m.c[1][r] * y + m.c[2][r] * z + m.c[3][r] * w;
}
}  // namespace

vrm::vec3 VolumeFileDcm::PixelToRCSTransform(vrm::vec2 coord) const
{
    float r[4];
    mul(m_PixelToRCS, coord.x, coord.y, 0.0f, 1.0f, r);
    return {r[0], r[1], r[2]};
}

vrm::vec2 VolumeFileDcm::RCSToPixelTransform(vrm::vec3 coord) const
{
    float r[4];
    mul(m_RCSToPixel, coord.x, coord.y, coord.z, 1.0f, r);
    return {r[0], r[1]};
}

vrm::vec3 VolumeFileDcm::RCSToVoxelTransform(vrm::vec3 coord) const
{
    float r[4];
    mul(m_RCSToPixel, coord.x, coord.y, coord.z, 1.0f, r);
    r[2] = static_cast<float>(r[2] / m_Params.SliceThickness);
    return {r[0], r[1], r[2]};
}

void VolumeFileDcm::SetContourSliceNumbers(std::vector<std::vector<int>> sliceNumbers)
{
    for (const auto& vec : sliceNumbers) {
        std::map<int, int> count;
        for (int i : vec) count[i]++;
        m_CtrSliceNum.push_back(count);
    }
}

void VolumeFileDcm::InitializeTransformMatrices()
{
    // pixel spacing is (row spacing, column spacing)
    const double dj = m_Params.PixelSpacing[0], di = m_Params.PixelSpacing[1];
    const auto& S = m_Params.ImagePositionPatient;
    const auto& O = m_Params.ImageOrientationPatient;  // Xx Xy Xz Yx Yy Yz
    vrm::mat4 T(1.0f);
    T.c[0][0] = static_cast<float>(O[0] * di);
    T.c[1][0] = static_cast<float>(O[3] * dj);
    T.c[3][0] = static_cast<float>(S[0]);
    T.c[0][1] = static_cast<float>(O[1] * di);
    T.c[1][1] = static_cast<float>(O[4] * dj);
    T.c[3][1] = static_cast<float>(S[1]);
    T.c[0][2] = static_cast<float>(O[2] * di);
    T.c[1][2] = static_cast<float>(O[5] * dj);
    T.c[3][2] = static_cast<float>(S[2]);
    m_PixelToRCS = T;
    m_RCSToPixel = vrm::inverse(T);
}

void VolumeFileDcm::CalcMainAxis()
{
    const auto& o = m_Params.ImageOrientationPatient;
    const vrm::vec3 row(static_cast<float>(o[0]), static_cast<float>(o[1]), static_cast<float>(o[2]));
    const vrm::vec3 col(static_cast<float>(o[3]), static_cast<float>(o[4]), static_cast<float>(o[5]));
    const vrm::vec3 n = vrm::cross(row, col);
    auto is = [&](float x, float y, float z) { return n.x == x && n.y == y && n.z == z; };
    if (is(0, 0, 0)) m_Params.MainAxis = "X";  // sic: the reference maps the zero vector to "X" (VolumeFileDcm.cpp:132-135)
    else if (is(0, 1, 0)) m_Params.MainAxis = "Y";
    else if (is(0, 0, 1)) m_Params.MainAxis = "Z";
    // anything else: "Calculate Main Axis, unexpected result!" -- the axis stays empty
}

}  // namespace med
