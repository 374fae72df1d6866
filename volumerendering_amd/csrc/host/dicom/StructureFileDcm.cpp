#include "StructureFileDcm.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <sstream>

namespace med {

namespace {

// One slice-by-slice rasteriser over the mask volume (x fastest, then y, then z; channel = contour).
class MaskRaster {
public:
    MaskRaster(int nx, int ny, int nz) : m_Nx(nx), m_Ny(ny), m_Nz(nz), m_Data(static_cast<size_t>(nx) * ny * nz) {}

    bool Inside(int x, int y, int z) const { return x >= 0 && y >= 0 && z >= 0 && x < m_Nx && y < m_Ny && z < m_Nz; }
    float& At(int x, int y, int z, int channel)
    {
        vrm::vec4& v = m_Data[(static_cast<size_t>(z) * m_Ny + y) * m_Nx + x];
        return channel == 0 ? v.x : (channel == 1 ? v.y : (channel == 2 ? v.z : v.w));
    }
    void Mark(int x, int y, int z, int channel)
    {
        if (Inside(x, y, z)) At(x, y, z, channel) = 1.0f;
    }

    // StructureFileDcm.cpp:321-411 with the reference's only structuring element, 3x3 ones, attached in the middle:
    // dilation sets a voxel whose neighbourhood holds a 1, erosion clears one whose neighbourhood holds a 0; both
    // read a copy of the slice taken before the pass and leave the one-voxel border alone.
    void Morph3x3(int z, int channels, bool erode)
    {
        if (z < 0 || z >= m_Nz) return;
        const size_t base = static_cast<size_t>(z) * m_Ny * m_Nx;
        const std::vector<vrm::vec4> before(m_Data.begin() + base, m_Data.begin() + base + static_cast<size_t>(m_Ny) * m_Nx);
        auto old = [&](int x, int y, int c) {
            const vrm::vec4& v = before[static_cast<size_t>(y) * m_Nx + x];
            return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
        };
        for (int c = 0; c < channels; ++c)
            for (int y = 1; y < m_Ny - 1; ++y)
                for (int x = 1; x < m_Nx - 1; ++x) {
                    bool found = false;  // a 0 (erosion) / a 1 (dilation) under the element
                    for (int dy = -1; dy <= 1 && !found; ++dy)
                        for (int dx = -1; dx <= 1 && !found; ++dx) found = old(x + dx, y + dy, c) == (erode ? 0.0f : 1.0f);
                    if (found) At(x, y, z, c) = erode ? 0.0f : 1.0f;
                }
    }

    // StructureFileDcm.cpp:413-462: scans rows downwards from yStart; in a row, behind a run of fewer than five 1s
    // the first 0 is taken to be inside; behind a wider run (an already filled shape) the scan goes on.
    bool FindSeed(int yStart, int z, int channel, int* sx, int* sy)
    {
        if (z < 0 || z >= m_Nz) return false;
        constexpr int kEdgeWidth = 5;
        for (int y = yStart; y >= 0 && y < m_Ny; ++y) {
            for (int x = 0; x < m_Nx; ++x) {
                if (At(x, y, z, channel) != 1.0f) continue;
                int ones = 0;
                for (; x < m_Nx; ++x) {
                    const float v = At(x, y, z, channel);
                    if (v == 1.0f) {
                        ++ones;
                    } else if (v == 0.0f) {
                        if (ones < kEdgeWidth) {
                            *sx = x;
                            *sy = y;
                            return true;
                        }
                        break;  // passed a filled shape; the row scan resumes behind this 0
                    }
                }
            }
        }
        return false;
    }

    // StructureFileDcm.cpp:464-486: the 4-connected region of 0s around the seed becomes 1 (order does not matter)
    void FloodFill(int sx, int sy, int z, int channel)
    {
        if (z < 0 || z >= m_Nz) return;
        std::vector<std::pair<int, int>> stack{{sx, sy}};
        while (!stack.empty()) {
            auto [x, y] = stack.back();
            stack.pop_back();
            if (x < 0 || y < 0 || x >= m_Nx || y >= m_Ny || At(x, y, z, channel) != 0.0f) continue;
            At(x, y, z, channel) = 1.0f;
            stack.push_back({x - 1, y});
            stack.push_back({x + 1, y});
            stack.push_back({x, y - 1});
            stack.push_back({x, y + 1});
        }
    }

    std::vector<vrm::vec4>& Data() { return m_Data; }

private:
    int m_Nx, m_Ny, m_Nz;
    std::vector<vrm::vec4> m_Data;
};

// StructureFileDcm.cpp:245-319: all voxels of the line start..end (both included), sampled along the longer axis
std::vector<std::pair<int, int>> BresenhamLine(vrm::vec3 start, vrm::vec3 end)
{
    const int dX = static_cast<int>(end.x - start.x), dY = static_cast<int>(end.y - start.y);
    const int aX = std::abs(dX), aY = std::abs(dY);
    const int incX = dX > 0 ? 1 : -1, incY = dY > 0 ? 1 : -1;
    const bool alongX = aX > aY;
    const int steps = alongX ? aX : aY;
    int x = static_cast<int>(start.x), y = static_cast<int>(start.y);
    int d = -steps;
    std::vector<std::pair<int, int>> line;
    line.reserve(static_cast<size_t>(steps) + 1);
    for (int i = 0; i <= steps; ++i) {
        line.emplace_back(x, y);
        d += 2 * (alongX ? aY : aX);
        if (d >= 0) {
            x += incX;
            y += incY;
            d -= 2 * steps;
        } else if (alongX) {
            x += incX;
        } else {
            y += incY;
        }
    }
    return line;
}

vrm::vec3 RoundHalfAway(vrm::vec3 v) { return {std::round(v.x), std::round(v.y), std::round(v.z)}; }

}  // namespace

StructureFileDcm::StructureFileDcm(std::filesystem::path path, DicomStructParams params,
                                   std::vector<std::vector<std::vector<float>>> data)
    : m_Path(std::move(path)), m_Params(std::move(params)), m_Data(std::move(data))
{
}

std::string StructureFileDcm::ListAvailableContours() const
{
    std::stringstream s;
    s << "Contours:\n";
    for (const auto& roi : m_Params.StructureSetROISequence) s << "\t" << roi.Number << " : " << roi.Name << roi.AlgorithmType << "\n";
    return s.str();
}

std::shared_ptr<VolumeFileDcm> StructureFileDcm::Create3DMask(const IDicomFile& other, std::array<int, 4> contourIDs,
                                                              ContourPostProcess postProcess)
{
    unsigned opt = postProcess;
    if ((opt & IGNORE_DUPLICATES) && (opt & ~1u)) opt = IGNORE_DUPLICATES;  // ambiguous: everything is ignored (:51-55)
    if (other.GetModality() != DicomModality::CT || !CompareFrameOfReference(other)) return nullptr;
    const auto* reference = dynamic_cast<const VolumeFileDcm*>(&other);
    if (!reference) return nullptr;

    std::vector<int> contours;  // indices into m_Data, in channel order
    for (int id : contourIDs)
        if (id > 0 && static_cast<size_t>(id) < m_Data.size()) contours.push_back(id - 1);
    std::vector<std::vector<int>> sliceNumbers(contours.size());

    const auto [xSize, ySize, zSize] = reference->GetSize();
    MaskRaster mask(xSize, ySize, zSize);
    const DicomVolumeParams vp = reference->GetVolumeParams();
    // contour point -> voxel: first-slice position and the spacings only (no orientation matrix), :105-110,125
    const vrm::vec3 origin{static_cast<float>(vp.ImagePositionPatient[0]), static_cast<float>(vp.ImagePositionPatient[1]),
                           static_cast<float>(vp.ImagePositionPatient[2])};
    const vrm::vec3 spacing{static_cast<float>(vp.PixelSpacing[0]), static_cast<float>(vp.PixelSpacing[1]),
                            static_cast<float>(vp.SliceThickness)};
    auto toVoxel = [&](vrm::vec3 rcs) {
        const vrm::vec3 d = rcs - origin;
        return vrm::vec3{d.x / spacing.x, d.y / spacing.y, d.z / spacing.z};
    };
    const bool pointLevel = !(opt & IGNORE_DUPLICATES);
    const int channels = static_cast<int>(contours.size());

    for (int l = 0; l < channels; ++l) {
        for (const std::vector<float>& polygon : m_Data[static_cast<size_t>(contours[static_cast<size_t>(l)])]) {
            if (polygon.size() < 3) continue;
            int sliceNumber = -1;
            std::vector<float> rows;  // voxel y of every marked point
            rows.reserve(polygon.size() / 3);
            for (size_t j = 0; j + 3 <= polygon.size(); j += 3) {
                const vrm::vec3 point{polygon[j], polygon[j + 1], polygon[j + 2]};
                vrm::vec3 voxel = toVoxel(point);
                voxel.z = std::fabs(voxel.z);
                voxel = RoundHalfAway(voxel);
                const int vx = static_cast<int>(voxel.x), vy = static_cast<int>(voxel.y), vz = static_cast<int>(voxel.z);
                sliceNumber = vz;
                if (!mask.Inside(vx, vy, vz)) continue;  // "Contour point out of bounds, skipping..."
                const bool duplicate = mask.At(vx, vy, vz, l) == 1.0f;
                if (pointLevel && (duplicate || (opt & PROCESS_NON_DUPLICATES))) {
                    if (opt & NEAREST_NEIGHBOUR) {
                        // the 8-neighbour whose centre is closest to the contour point in patient space (:205-243)
                        float best = std::numeric_limits<float>::max();
                        int bx = vx, by = vy;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                if (dx == 0 && dy == 0) continue;
                                const int nx = vx + dx, ny = vy + dy;
                                if (nx < 0 || nx >= xSize || ny < 0 || ny > ySize) continue;  // (sic: <= ySize, :227)
                                const vrm::vec3 rcs = reference->PixelToRCSTransform({static_cast<float>(nx), static_cast<float>(ny)});
                                const float dist = vrm::length(point - rcs);
                                if (dist < best) {
                                    best = dist;
                                    bx = nx;
                                    by = ny;
                                }
                            }
                        mask.Mark(bx, by, vz, l);
                    }
                    if ((opt & RECONSTRUCT_BRESENHAM) && j + 6 <= polygon.size()) {
                        const vrm::vec3 next = RoundHalfAway(toVoxel({polygon[j + 3], polygon[j + 4], polygon[j + 5]}));
                        const auto line = BresenhamLine(voxel, next);
                        // without the last voxel: it is the next point and must not look like a duplicate then
                        for (size_t k = 0; k + 1 < line.size(); ++k) mask.Mark(line[k].first, line[k].second, vz, l);
                    }
                }
                rows.push_back(voxel.y);
                mask.At(vx, vy, vz, l) = 1.0f;
            }
            sliceNumbers[static_cast<size_t>(l)].push_back(sliceNumber);
            if (opt & CLOSING) {
                mask.Morph3x3(sliceNumber, channels, /*erode=*/false);
                mask.Morph3x3(sliceNumber, channels, /*erode=*/true);
            }
            if ((opt & FILL) && !rows.empty()) {
                std::sort(rows.begin(), rows.end());
                int sx = -1, sy = -1;
                if (mask.FindSeed(static_cast<int>(rows[rows.size() / 2]), sliceNumber, l, &sx, &sy))
                    mask.FloodFill(sx, sy, sliceNumber, l);  // else: "Did not find the seed for fill."
            }
        }
    }

    auto file = std::make_shared<VolumeFileDcm>(m_Path, reference->GetSize(), FileDataType::Float, vp, mask.Data());
    file->SetContourSliceNumbers(sliceNumbers);
    return file;
}

}  // namespace med
