// VolumeFileDcm -- a VolumeFile that came from a DICOM series: keeps the geometry attributes and the pixel <->
// reference-coordinate-system (RCS) transforms.  Mirrors med::VolumeFileDcm (App/src/file/dicom/VolumeFileDcm.{h,cpp}).
#pragma once
#include <map>
#include <vector>

#include "../VolumeFile.h"
#include "DicomParams.h"

namespace med {

class IDicomFile {
public:
    virtual ~IDicomFile() = default;
    virtual DicomBaseParams GetBaseParams() const = 0;
    virtual DicomModality GetModality() const = 0;
    virtual bool CompareFrameOfReference(const IDicomFile& other) const = 0;
};

class VolumeFileDcm : public IDicomFile, public VolumeFile {
public:
    VolumeFileDcm(std::filesystem::path path, Size size, FileDataType type, DicomVolumeParams params, std::vector<vrm::vec4>& data);

    bool CompareOrientation(const VolumeFileDcm& other) const;       // VolumeFileDcm.cpp:34-43
    vrm::vec3 PixelToRCSTransform(vrm::vec2 coord) const;            // :61-65
    vrm::vec2 RCSToPixelTransform(vrm::vec3 coord) const;            // :67-71
    vrm::vec3 RCSToVoxelTransform(vrm::vec3 coord) const;            // :73-78
    DicomBaseParams GetBaseParams() const override { return m_Params; }
    DicomModality GetModality() const override { return m_Params.Modality; }
    bool CompareFrameOfReference(const IDicomFile& other) const override;
    DicomVolumeParams GetVolumeParams() const { return m_Params; }
    std::tuple<float, float, float> GetBBOXSize() const override;    // :50-59 (millimetre extents)
    void SetContourSliceNumbers(std::vector<std::vector<int>> sliceNumbers);

private:
    void InitializeTransformMatrices();  // :94-121
    void CalcMainAxis();                 // :123-150

    DicomVolumeParams m_Params;
    vrm::mat4 m_PixelToRCS{1.0f}, m_RCSToPixel{1.0f};
    std::vector<std::map<int, int>> m_CtrSliceNum;
};

}  // namespace med
