// VolumeFile -- data preparation that feeds the hot path.  Arithmetic follows App/src/file/VolumeFile.cpp
// expression by expression (so results are bit-identical to the reference's single-threaded loops); the loops
// themselves are split over z-slabs across host threads, which changes nothing per voxel.
#include "VolumeFile.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <thread>

namespace med {

namespace {
unsigned g_workers = 0;

unsigned worker_count(size_t items)
{
    unsigned n = g_workers ? g_workers : std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::min<size_t>(n, std::max<size_t>(items, 1));
}

// runs fn(begin, end, worker) over [0, n) split into contiguous chunks
void parallel_for(size_t n, const std::function<void(size_t, size_t, unsigned)>& fn)
{
    unsigned w = worker_count(n);
    if (w <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    size_t chunk = (n + w - 1) / w;
    for (unsigned t = 0; t < w; ++t) {
        size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
        if (b < e) th.emplace_back(fn, b, e, t);
    }
    for (auto& t : th) t.join();
}

template <typename T>
VolumeFile from_raw(const T* raw, VolumeFile::Size size, FileDataType type)
{
    auto [x, y, z] = size;
    size_t n = (size_t)x * y * z;
    std::vector<vrm::vec4> data(n);
    T mx = 0;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, raw[i]);
    parallel_for(n, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; ++i) data[i] = vrm::vec4(static_cast<float>(raw[i]));
    });
    return VolumeFile("", size, type, data, static_cast<size_t>(mx));
}
}  // namespace

void VolumeFile::SetWorkerThreads(unsigned n) { g_workers = n; }

VolumeFile::VolumeFile(std::filesystem::path path, Size size, FileDataType type, std::vector<vrm::vec4>& data, size_t maxNumber)
    : m_FileDataType(type), m_Path(std::move(path)), m_Size(size), m_MaxNumber(maxNumber), m_Data(std::move(data))
{
    if (m_MaxNumber == 0) m_MaxNumber = GetMaxNumber(m_Data);
}

VolumeFile VolumeFile::FromRaw(const std::uint16_t* raw, Size size, FileDataType type) { return from_raw(raw, size, type); }
VolumeFile VolumeFile::FromRaw(const std::uint32_t* raw, Size size, FileDataType type) { return from_raw(raw, size, type); }

float VolumeFile::RoundTo2Dec(float number) const
{
    int value = static_cast<int>(number * 100 + .5f);
    return static_cast<float>(value) / 100;
}

size_t VolumeFile::GetMaxNumber(const std::vector<vrm::vec4>& vec, int index) const
{
    if (vec.empty()) return 0;
    float best = vec[0][index];
    for (const auto& v : vec)
        if (best < v[index]) best = v[index];
    return static_cast<size_t>(best);
}

size_t VolumeFile::GetDataRange() const { return m_NormalizationValue != 0 ? (size_t)m_NormalizationValue : GetMaxNumber(); }

std::tuple<float, float, float> VolumeFile::GetBBOXSize() const
{
    auto [x, y, z] = m_Size;
    float mx = std::max(x, std::max(y, z));
    return {RoundTo2Dec(x / mx), RoundTo2Dec(y / mx), RoundTo2Dec(z / mx)};
}

int VolumeFile::GetIndexFrom3D(int x, int y, int z) const
{
    const auto& [width, height, depth] = m_Size;
    if (x < 0 || x >= width || y < 0 || y >= height || z < 0 || z >= depth) return -1;
    return z * height * width + y * width + x;
}

vrm::vec4 VolumeFile::GetVoxelData(int x, int y, int z) const
{
    int index = GetIndexFrom3D(x, y, z);
    return index != -1 ? m_Data[index] : vrm::vec4(0.0f);
}

void VolumeFile::NormalizeData(int normalizationValue)
{
    if (m_IsNormalized) return;
    if (normalizationValue == 0) normalizationValue = static_cast<int>(GetMaxNumber());
    m_NormalizationValue = normalizationValue;
    const int nv = m_NormalizationValue;
    parallel_for(m_Data.size(), [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; ++i) m_Data[i].a /= nv;
    });
    m_IsNormalized = true;
}

void VolumeFile::PreComputeGradient(bool normToZeroOne)
{
    if (m_HasGradient) return;  // "Gradient has been already computed, skipping..."
    const int xS = std::get<0>(m_Size), yS = std::get<1>(m_Size), zS = std::get<2>(m_Size);
    // density plane copy: the loop overwrites .xyz only, but reading .a through a scalar plane keeps the
    // z-slabs independent and is what makes the pass bandwidth- instead of latency-bound
    const size_t n = m_Data.size();
    std::vector<float> dens(n);
    parallel_for(n, [&](size_t b, size_t e, unsigned) {
        for (size_t i = b; i < e; ++i) dens[i] = m_Data[i].a;
    });
    auto A = [&](int x, int y, int z) -> float {
        if (x < 0 || x >= xS || y < 0 || y >= yS || z < 0 || z >= zS) return 0.0f;  // GetVoxelData: vec4(0)
        return dens[((size_t)z * yS + y) * xS + x];
    };
    std::vector<float> slabMax(worker_count((size_t)zS) + 1, 0.0f);
    parallel_for((size_t)zS, [&](size_t zb, size_t ze, unsigned worker) {
        float localMax = 0.0f;
        for (int z = (int)zb; z < (int)ze; ++z)
            for (int y = 0; y < yS; ++y)
                for (int x = 0; x < xS; ++x) {
                    vrm::vec3 p(A(x + 1, y, z), A(x, y + 1, z), A(x, y, z + 1));
                    vrm::vec3 m(A(x - 1, y, z), A(x, y - 1, z), A(x, y, z - 1));
                    vrm::vec3 temp = (-(p - m)) * vrm::vec3(0.5f);  // 1/2h, h = 1
                    if (normToZeroOne) {
                        float mag = vrm::length(temp);
                        if (mag > localMax) localMax = mag;
                    }
                    vrm::vec4& v = m_Data[((size_t)z * yS + y) * xS + x];
                    v.x = temp.x;
                    v.y = temp.y;
                    v.z = temp.z;
                }
        slabMax[worker] = localMax;
    });
    if (normToZeroOne) {
        float maxGradMag = 0.0f;
        for (float v : slabMax)
            if (v > maxGradMag) maxGradMag = v;
        parallel_for(n, [&](size_t b, size_t e, unsigned) {
            for (size_t i = b; i < e; ++i) {
                m_Data[i].x /= maxGradMag;
                m_Data[i].y /= maxGradMag;
                m_Data[i].z /= maxGradMag;
            }
        });
    }
    m_HasGradient = true;
}

void VolumeFile::PreComputeGradientSobel()
{
    if (m_HasGradient) return;
    static const float SX[3][3][3] = {{{-1, 0, 1}, {-2, 0, 2}, {-1, 0, 1}}, {{-2, 0, 2}, {-4, 0, 4}, {-2, 0, 2}}, {{-1, 0, 1}, {-2, 0, 2}, {-1, 0, 1}}};
    static const float SY[3][3][3] = {{{-1, -2, -1}, {0, 0, 0}, {1, 2, 1}}, {{-2, -4, -2}, {0, 0, 0}, {2, 4, 2}}, {{-1, -2, -1}, {0, 0, 0}, {1, 2, 1}}};
    static const float SZ[3][3][3] = {{{-1, -2, -1}, {-2, -4, -2}, {-1, -2, -1}}, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {{1, 2, 1}, {2, 4, 2}, {1, 2, 1}}};
    auto [xS, yS, zS] = m_Size;
    for (int z = 1; z < zS - 1; ++z)
        for (int y = 1; y < yS - 1; ++y)
            for (int x = 1; x < xS - 1; ++x) {
                float gx = 0.0f, gy = 0.0f, gz = 0.0f;
                for (int i = -1; i <= 1; ++i)
                    for (int j = -1; j <= 1; ++j)
                        for (int k = -1; k <= 1; ++k) {
                            float a = GetVoxelData(x + i, y + j, z + k).a;
                            gx += a * SX[k + 1][j + 1][i + 1];
                            gy += a * SY[k + 1][j + 1][i + 1];
                            gz += a * SZ[k + 1][j + 1][i + 1];
                        }
                vrm::vec4& v = m_Data[GetIndexFrom3D(x, y, z)];
                v.x = gx;
                v.y = gy;
                v.z = gz;
            }
    m_HasGradient = true;
}

void VolumeFile::AverageGradient(int /*kernelSize*/)
{
    // The reference accumulates into a local and discards it (VolumeFile.cpp:138-159): gradients stay
    // un-averaged.  Only its side effect is kept: it computes the gradient first when there is none (:123-127).
    if (!m_HasGradient) PreComputeGradient();
}

}  // namespace med
