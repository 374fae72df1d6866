// VolumeFile -- host container of one volume in the layout the ray-marcher consumes.
// Mirrors the public surface of med::VolumeFile (App/src/file/VolumeFile.h:19-152): one vec4 per voxel,
// .rgb = (optional) pre-computed gradient, .a = density, x fastest (VolumeFile.cpp:306).
#pragma once
#include <cstdint>
#include <filesystem>
#include <tuple>
#include <vector>

#include "FileDataType.h"
#include "vrm.h"

namespace med {

class VolumeFile {
public:
    using Size = std::tuple<std::uint16_t, std::uint16_t, std::uint16_t>;

    VolumeFile() = default;
    // VolumeFile.cpp:6-19: takes ownership of `data`; maxNumber == 0 -> computed from component [0]
    VolumeFile(std::filesystem::path path, Size size, FileDataType type, std::vector<vrm::vec4>& data, size_t maxNumber = 0);
    virtual ~VolumeFile() = default;

    // Builds the vec4 voxels the way the readers do: the raw integer broadcast to all four lanes
    // (DicomReader.cpp:239,247; DatReader.cpp:42).
    static VolumeFile FromRaw(const std::uint16_t* raw, Size size, FileDataType type = FileDataType::Undefined);
    static VolumeFile FromRaw(const std::uint32_t* raw, Size size, FileDataType type = FileDataType::Undefined);

    void PreComputeGradient(bool normToZeroOne = false);  // VolumeFile.cpp:196-257
    void PreComputeGradientSobel();                       // VolumeFile.cpp:77-117 (never called by the reference)
    void AverageGradient(int kernelSize);                 // VolumeFile.cpp:119-163 (computes and discards: a no-op)
    void NormalizeData(int normalizationValue = 0);       // VolumeFile.cpp:165-184

    // The device copy was normalised by `value` (MiniApp::SetPrepareOnDevice) while these host voxels stay as loaded:
    // GetDataRange() then reports what NormalizeData() would have recorded; IsNormalized() keeps describing the host data.
    void SetDeviceNormalization(int value) { m_NormalizationValue = value; }
    bool IsNormalized() const { return m_IsNormalized; }
    bool HasGradient() const { return m_HasGradient; }
    [[nodiscard]] Size GetSize() const { return m_Size; }
    virtual std::tuple<float, float, float> GetBBOXSize() const;
    [[nodiscard]] FileDataType GetFileType() const { return m_FileDataType; }
    [[nodiscard]] const void* GetVoidPtr() const { return m_Data.data(); }
    [[nodiscard]] const std::vector<vrm::vec4>& GetVecReference() const { return m_Data; }
    [[nodiscard]] size_t GetMaxNumber() const { return m_MaxNumber; }
    [[nodiscard]] size_t GetMaxNumber(const std::vector<vrm::vec4>& vec, int index = 0) const;
    [[nodiscard]] int GetMaxUsedBitDepth() const { return m_CustomBitWidth; }
    [[nodiscard]] size_t GetDataRange() const;
    [[nodiscard]] int GetIndexFrom3D(int x, int y, int z) const;
    [[nodiscard]] vrm::vec4 GetVoxelData(int x, int y, int z) const;

    // Worker threads used by the data-preparation passes (results do not depend on it). 0 = hardware.
    static void SetWorkerThreads(unsigned n);

protected:
    float RoundTo2Dec(float number) const;

    bool m_HasGradient = false;
    bool m_IsNormalized = false;
    FileDataType m_FileDataType = FileDataType::Undefined;
    std::filesystem::path m_Path{};
    Size m_Size{0, 0, 0};
    size_t m_MaxNumber = 0;
    int m_CustomBitWidth = 0;
    int m_NormalizationValue = 0;
    std::vector<vrm::vec4> m_Data{};
};

}  // namespace med
