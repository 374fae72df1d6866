// Mirrors App/src/file/dicom/DicomParams.h: the DICOM attributes the volume path keeps.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../vrm.h"

namespace med {

enum class DicomModality { UNKNOWN, CT, RTSTRUCT, RTDOSE, MR, CONTOURMASK };

struct DicomBaseParams {
    explicit DicomBaseParams(DicomModality mod) : Modality(mod) {}
    DicomModality Modality{DicomModality::UNKNOWN};
    std::string FrameOfReference{};
};

// CT, MR, RTDose
struct DicomVolumeParams : public DicomBaseParams {
    DicomVolumeParams() : DicomBaseParams(DicomModality::UNKNOWN) {}
    std::uint16_t X = 0;                                   // filled from (0028,0010) Rows    (DicomReader.cpp:181)
    std::uint16_t Y = 0;                                   // filled from (0028,0011) Columns (DicomReader.cpp:182)
    std::uint16_t Z = 0;                                   // (0028,0008) Number of Frames, or the number of slice files
    std::uint16_t BitsStored = 0;                          // (0028,0101)
    std::uint16_t BitsAllocated = 0;                       // (0028,0100)
    std::int16_t NumberOfFrames = 0;
    std::uint16_t LargestPixelValue = 0;                   // (0028,0107)
    std::uint16_t SmallestPixelValue = 0;                  // (0028,0106)
    double SliceThickness = 0.0;                           // (0018,0050)
    std::array<double, 3> ImagePositionPatient{0.0};       // (0020,0032)
    std::array<double, 6> ImageOrientationPatient{0.0};    // (0020,0037)
    std::array<double, 2> PixelSpacing{0.0};               // (0028,0030) row / column spacing
    std::string MainAxis{};
};

// RTSTRUCT (DicomParams.h:44-66)
struct DicomStructParams : public DicomBaseParams {
    DicomStructParams() : DicomBaseParams(DicomModality::RTSTRUCT) {}
    struct StructureSetROI {      // an item of (3006,0020) Structure Set ROI Sequence
        int Number = 0;           // (3006,0022) ROI Number
        std::string Name{};       // (3006,0026) ROI Name
        std::string AlgorithmType{};  // (3006,0036) ROI Generation Algorithm
    };
    std::string Label{};  // (3006,0002) Structure Set Label
    std::string Name{};   // (3006,0004) Structure Set Name
    std::vector<StructureSetROI> StructureSetROISequence{};
    std::vector<vrm::vec3> DisplayColors{};  // (3006,002A) ROI Display Color, / 255
};

}  // namespace med
