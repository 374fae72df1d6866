// Mirrors App/src/file/FileDataType.h: what kind of data a VolumeFile holds.
#pragma once
namespace med {
enum class FileDataType { Undefined, DicomCT, DicomMR, DicomRTDose, DicomRTStruct, Dat, Synthetic };
}
