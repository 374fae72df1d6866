// Follows App/src/file/dat/DatReader.cpp:11-46 with one deliberate difference: the reference sizes its vec4
// vector to `res` elements AND back_inserts the converted voxels behind them (:39,42), so the first x*y*z voxels --
// the ones Texture::CreateFromData uploads -- are all zero.  Here the voxels are written in place.
#include "DatReader.h"

#include <fstream>
#include <stdexcept>
#include <vector>

namespace med {

VolumeFile DatImpl::ReadFile(const std::filesystem::path& name, bool /*isDir*/)
{
    std::ifstream file(name, std::ios_base::binary);
    if (file.fail() || !file.is_open()) throw std::runtime_error("Check file");
    constexpr int HEADER_SIZE = 6;
    unsigned char hdr[HEADER_SIZE];
    file.read(reinterpret_cast<char*>(hdr), HEADER_SIZE);
    if (file.gcount() != HEADER_SIZE) throw std::runtime_error("Check file");
    std::uint16_t dims[3];
    for (int i = 0; i < 3; ++i) dims[i] = static_cast<std::uint16_t>(hdr[2 * i] | (hdr[2 * i + 1] << 8));
    const size_t res = static_cast<size_t>(dims[0]) * dims[1] * dims[2];
    if (res == 0) throw std::runtime_error("File is empty");
    std::vector<unsigned char> bytes(res * 2);
    file.read(reinterpret_cast<char*>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
    if (static_cast<size_t>(file.gcount()) != bytes.size()) throw std::runtime_error("Check file");
    std::vector<std::uint16_t> raw(res);
    for (size_t i = 0; i < res; ++i) raw[i] = static_cast<std::uint16_t>(bytes[2 * i] | (bytes[2 * i + 1] << 8));
    return VolumeFile::FromRaw(raw.data(), {dims[0], dims[1], dims[2]}, FileDataType::Uint16);
}

bool DatImpl::WriteFile(const std::filesystem::path& name, const std::uint16_t* raw, std::uint16_t x, std::uint16_t y,
                        std::uint16_t z)
{
    std::ofstream file(name, std::ios_base::binary);
    if (!file) return false;
    const std::uint16_t dims[3] = {x, y, z};
    for (std::uint16_t d : dims) {
        file.put(static_cast<char>(d & 0xFF));
        file.put(static_cast<char>(d >> 8));
    }
    const size_t n = static_cast<size_t>(x) * y * z;
    for (size_t i = 0; i < n; ++i) {
        file.put(static_cast<char>(raw[i] & 0xFF));
        file.put(static_cast<char>(raw[i] >> 8));
    }
    return static_cast<bool>(file);
}

}  // namespace med
