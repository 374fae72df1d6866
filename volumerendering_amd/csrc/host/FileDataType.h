// Mirrors App/src/file/FileDataType.h: element type of the source data a VolumeFile was built from.
#pragma once
namespace med {
enum class FileDataType { Undefined, Uint8, Uint16, Uint32, Float, Double };
}
