// DatImpl -- reader of the ".dat" raw volume format: a 6-byte header of three little-endian uint16 dimensions
// (x, y, z) followed by x*y*z little-endian uint16 voxels, x fastest.  Mirrors med::DatImpl
// (App/src/file/dat/DatReader.{h,cpp}).
#pragma once
#include <filesystem>

#include "VolumeFile.h"

namespace med {

class DatImpl {
public:
    // Throws std::runtime_error("Check file") when the file cannot be opened or is truncated (the reference
    // throws the MSVC-only std::exception("Check file"), DatReader.cpp:17-20).
    [[nodiscard]] VolumeFile ReadFile(const std::filesystem::path& name, bool isDir = false);

    // Writes a volume's density (.a of an un-normalised VolumeFile, i.e. the raw integer) in the same format.
    static bool WriteFile(const std::filesystem::path& name, const std::uint16_t* raw, std::uint16_t x, std::uint16_t y,
                          std::uint16_t z);
};

}  // namespace med
