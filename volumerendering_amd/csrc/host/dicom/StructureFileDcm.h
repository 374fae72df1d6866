// StructureFileDcm -- an RTSTRUCT file: the contours of a structure set and the rasteriser that turns up to four of
// them into the RGBA mask volume of the VolumeMask / TFCalibration scenes.  Mirrors med::StructureFileDcm
// (App/src/file/dicom/StructureFileDcm.{h,cpp}); same options, same results, written as a small rasteriser
// (mark / line / morphology / seed / fill) instead of the reference's one long loop.
#pragma once
#include <array>
#include <filesystem>
#include <memory>
#include <vector>

#include "DicomParams.h"
#include "VolumeFileDcm.h"

namespace med {

// StructureFileDcm.h:14-28 (the reference's first enumerator is called IGNORE, a macro on some platforms)
enum ContourPostProcess : unsigned int {
    IGNORE_DUPLICATES = 1 << 0,       // reference: IGNORE -- no post-processing at all
    NEAREST_NEIGHBOUR = 1 << 1,       // a point that lands on a marked voxel also marks the closest of its 8 neighbours
    RECONSTRUCT_BRESENHAM = 1 << 2,   // ... also marks the line to the next contour point
    CLOSING = 1 << 3,                 // 3x3 dilation then erosion of the slice after every contour
    FILL = 1 << 4,                    // flood fill from a seed found on the contour's median row
    PROCESS_NON_DUPLICATES = 1 << 5   // apply the point-level options to every point, not only to duplicates
};
inline ContourPostProcess operator|(ContourPostProcess a, ContourPostProcess b)
{
    return static_cast<ContourPostProcess>(static_cast<unsigned>(a) | static_cast<unsigned>(b));
}

class StructureFileDcm : public IDicomFile {
public:
    // data[contour][polygon] = x0 y0 z0 x1 y1 z1 ... in the patient coordinate system (millimetres)
    StructureFileDcm(std::filesystem::path path, DicomStructParams params, std::vector<std::vector<std::vector<float>>> data);

    DicomStructParams GetStructParams() const { return m_Params; }
    const std::vector<std::vector<std::vector<float>>>& GetContourData() const { return m_Data; }
    std::string ListAvailableContours() const;  // the text the reference logs (StructureFileDcm.cpp:22-31)

    // StructureFileDcm.cpp:49-203.  `other` must be a CT VolumeFileDcm in the same frame of reference (else nullptr).
    // contourIDs are 1-based; 0 is ignored, and -- as in the reference -- so is anything >= the number of contours,
    // i.e. the LAST contour of a file cannot be selected (`id >= m_Data.size()`, :68).  Channel l of the mask is the
    // l-th accepted id.  Points outside the volume are skipped; where the reference would index out of bounds
    // (neighbour row == ySize :227, Bresenham voxels off the slice :156) this implementation skips the voxel.
    std::shared_ptr<VolumeFileDcm> Create3DMask(const IDicomFile& other, std::array<int, 4> contourIDs,
                                                ContourPostProcess postProcess);

    DicomBaseParams GetBaseParams() const override { return m_Params; }
    DicomModality GetModality() const override { return m_Params.Modality; }
    bool CompareFrameOfReference(const IDicomFile& other) const override
    {
        return m_Params.FrameOfReference == other.GetBaseParams().FrameOfReference;
    }

private:
    std::filesystem::path m_Path;
    DicomStructParams m_Params;
    std::vector<std::vector<std::vector<float>>> m_Data;
};

}  // namespace med
