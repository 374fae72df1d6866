// Mirrors App/src/file/dicom/DicomParams.h: the DICOM attributes the volume path keeps.
#pragma once
#include <array>
#include <cstdint>
#include <string>

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

}  // namespace med
