// DicomReader -- reads a CT / MR / RTDOSE series (a directory of single-frame .dcm slices, or one multi-frame
// file) into a VolumeFileDcm, and an RTSTRUCT file into a StructureFileDcm.  Mirrors med::DicomReader
// (App/src/file/dicom/DicomReader.{h,cpp}) and StructVisitor.h; the `dcm` library is replaced by dicom/DicomFile.
#pragma once
#include <filesystem>
#include <memory>
#include <string>
#include <vector>

#include "DicomFile.h"
#include "StructureFileDcm.h"
#include "VolumeFileDcm.h"

namespace med {

class DicomReader {
public:
    // Throws std::runtime_error on: not a .dcm file, empty directory, unreadable / unsupported file, unknown
    // BitsAllocated (the reference throws MSVC-only std::exception("..."), DicomReader.cpp:52,63,87,313).
    [[nodiscard]] static std::shared_ptr<VolumeFileDcm> ReadVolumeFile(std::filesystem::path name);
    // DicomReader.cpp:100-148: a .dcm file, or a directory that holds exactly one.  nullptr (as the reference, which
    // logs) when there is none / more than one, the file is unreadable or its modality is not RTSTRUCT.
    [[nodiscard]] static std::shared_ptr<StructureFileDcm> ReadStructFile(std::filesystem::path name);
    [[nodiscard]] static DicomModality CheckModality(const std::filesystem::path& name);
    [[nodiscard]] static DicomModality ResolveModality(std::string modality);  // not case sensitive
    [[nodiscard]] static std::string ResolveModality(DicomModality modality);
    [[nodiscard]] static std::vector<std::filesystem::path> SortDicomSlices(const std::vector<std::filesystem::path>& paths);
    [[nodiscard]] static bool IsDicomFile(const std::filesystem::path& path) { return path.extension() == ".dcm"; }

private:
    void ReadDicomVolumeVariables(const dcmlite::DicomFile& f);
    void ReadData(const dcmlite::DicomFile& f);
    void ResolveFileType();

    DicomVolumeParams m_Params;
    FileDataType m_FileDataType = FileDataType::Undefined;
    std::vector<vrm::vec4> m_Data;
};

// "x\\y\\z\\x\\y\\z..." -> floats; unparsable entries are dropped (DicomParseUtil.inl:76-92)
[[nodiscard]] std::vector<float> ParseContours(const std::string& str);

// "1\\0\\0" -> numbers; missing / unparsable entries stay 0 (DicomParseUtil.inl:15-80)
template <typename T, size_t N>
std::array<T, N> ParseStringToNumArr(const std::string& str);

}  // namespace med
