#include "DicomReader.h"

#include <algorithm>
#include <cctype>
#include <sstream>
#include <stdexcept>

namespace med {

using namespace dcmlite;

template <typename T, size_t N>
std::array<T, N> ParseStringToNumArr(const std::string& str)
{
    std::array<T, N> numbers{};
    std::stringstream ss(str);
    std::string temp;
    size_t i = 0;
    while (std::getline(ss, temp, '\\') && i < N) {
        try {
            if constexpr (std::is_same_v<T, int>) numbers[i++] = std::stoi(temp);
            else numbers[i++] = static_cast<T>(std::stod(temp));
        } catch (const std::exception&) {
            ++i;  // "Error parsing string to number": the slot keeps its 0
        }
    }
    return numbers;
}
template std::array<int, 1> ParseStringToNumArr<int, 1>(const std::string&);
template std::array<double, 1> ParseStringToNumArr<double, 1>(const std::string&);
template std::array<double, 2> ParseStringToNumArr<double, 2>(const std::string&);
template std::array<double, 3> ParseStringToNumArr<double, 3>(const std::string&);
template std::array<double, 6> ParseStringToNumArr<double, 6>(const std::string&);

std::shared_ptr<VolumeFileDcm> DicomReader::ReadVolumeFile(std::filesystem::path name)
{
    DicomReader reader;
    const bool isDir = std::filesystem::is_directory(name);
    bool hasSoloFile = false;
    std::vector<std::filesystem::path> paths;
    if (isDir) {
        std::vector<std::filesystem::path> listed;
        for (const auto& e : std::filesystem::directory_iterator(name))
            if (e.path().extension().string() == ".dcm") listed.push_back(e.path());
        std::sort(listed.begin(), listed.end());  // directory order is unspecified; the slice sort below decides
        paths = SortDicomSlices(listed);
        hasSoloFile = paths.size() == 1;
    } else {
        if (!IsDicomFile(name)) throw std::runtime_error("File is not a dicom file!");
        paths.push_back(name);
    }
    const std::size_t numberOfFiles = paths.size();
    if (numberOfFiles == 0) throw std::runtime_error("No dicom files found!");

    bool firstRun = true;
    for (const auto& file : paths) {
        DicomFile f(file.string());
        if (!f.Load()) throw std::runtime_error("Cannot continue, unable to open: " + file.string() + " (" + f.Error() + ")");
        if (firstRun) {
            reader.ReadDicomVolumeVariables(f);  // geometry comes from the first slice only (DicomReader.cpp:74-79)
            reader.m_Params.Modality = ResolveModality(f.GetString(tags::kModality));
            reader.m_Data.reserve(static_cast<size_t>(reader.m_Params.X) * reader.m_Params.Y * (isDir ? numberOfFiles : reader.m_Params.Z));
        }
        firstRun = false;
        reader.ReadData(f);
    }
    if (isDir && !hasSoloFile) reader.m_Params.Z = static_cast<std::uint16_t>(numberOfFiles);
    VolumeFile::Size size{reader.m_Params.X, reader.m_Params.Y, reader.m_Params.Z};
    return std::make_shared<VolumeFileDcm>(name, size, reader.m_FileDataType, reader.m_Params, reader.m_Data);
}

DicomModality DicomReader::CheckModality(const std::filesystem::path& name)
{
    DicomFile f(name.string());
    if (f.Load()) return ResolveModality(f.GetString(tags::kModality));
    return DicomModality::UNKNOWN;
}

DicomModality DicomReader::ResolveModality(std::string modality)
{
    std::transform(modality.begin(), modality.end(), modality.begin(), [](unsigned char c) { return std::toupper(c); });
    if (modality == "CT") return DicomModality::CT;
    if (modality == "MR") return DicomModality::MR;
    if (modality == "RTDOSE") return DicomModality::RTDOSE;
    if (modality == "RTSTRUCT") return DicomModality::RTSTRUCT;
    return DicomModality::UNKNOWN;
}

std::string DicomReader::ResolveModality(DicomModality modality)
{
    switch (modality) {
    case DicomModality::CT: return "CT";
    case DicomModality::RTSTRUCT: return "RTSTRUCT";
    case DicomModality::RTDOSE: return "RTDOSE";
    case DicomModality::MR: return "MR";
    default: return "UNKNOWN";
    }
}

void DicomReader::ReadDicomVolumeVariables(const DicomFile& f)
{
    DicomVolumeParams p;
    p.FrameOfReference = f.GetString(tags::kFrameOfReference);
    p.Modality = ResolveModality(f.GetString(tags::kModality));
    f.GetUint16(tags::kRows, &p.X);      // sic: X <- Rows, Y <- Columns (DicomReader.cpp:181-182)
    f.GetUint16(tags::kColumns, &p.Y);
    std::string str;
    f.GetString(tags::kNumberOfFrames, &str);
    p.Z = static_cast<std::uint16_t>(ParseStringToNumArr<int, 1>(str)[0]);
    if (p.Z == 0) p.Z = 1;  // "usually, the number of frames is 0 when it is one frame"
    f.GetUint16(tags::kBitsStored, &p.BitsStored);
    f.GetUint16(tags::kBitsAllocated, &p.BitsAllocated);
    str.clear();
    f.GetString(tags::kImageOrientationPatient, &str);
    p.ImageOrientationPatient = ParseStringToNumArr<double, 6>(str);
    str.clear();
    f.GetString(tags::kImagePositionPatient, &str);
    p.ImagePositionPatient = ParseStringToNumArr<double, 3>(str);
    str.clear();
    f.GetString(tags::kSliceThickness, &str);
    p.SliceThickness = ParseStringToNumArr<double, 1>(str)[0];
    str.clear();
    f.GetString(tags::kPixelSpacing, &str);
    p.PixelSpacing = ParseStringToNumArr<double, 2>(str);
    f.GetUint16(tags::kLargestPixelValue, &p.LargestPixelValue);
    f.GetUint16(tags::kSmallestPixelValue, &p.SmallestPixelValue);
    m_Params = p;
    ResolveFileType();
}

void DicomReader::ReadData(const DicomFile& f)
{
    // the raw integer is broadcast to all four lanes; no rescale slope / intercept is applied (DicomReader.cpp:230-255)
    switch (m_FileDataType) {
    case FileDataType::Uint16: {
        std::vector<std::uint16_t> vec;
        f.GetUint16Array(tags::kPixelData, &vec);
        for (auto v : vec) m_Data.emplace_back(static_cast<float>(v));
        break;
    }
    case FileDataType::Uint32: {
        std::vector<std::uint32_t> vec;
        f.GetUint32Array(tags::kPixelData, &vec);
        for (auto v : vec) m_Data.emplace_back(static_cast<float>(v));
        break;
    }
    default: break;  // FileDataType::Double: declared, never read by the reference
    }
}

std::vector<std::filesystem::path> DicomReader::SortDicomSlices(const std::vector<std::filesystem::path>& paths)
{
    if (paths.size() < 2) return paths;
    std::vector<std::pair<int, std::filesystem::path>> pairs;
    for (const auto& path : paths) {
        DicomFile f(path.string());
        std::string value;
        if (!f.Load() || !f.GetString(tags::kInstanceNumber, &value)) continue;  // "missing instance number"
        try {
            pairs.emplace_back(std::stoi(value), path);
        } catch (const std::exception&) {
        }
    }
    if (paths.size() != pairs.size()) return paths;  // "default order of path is going to be used"
    std::stable_sort(pairs.begin(), pairs.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<std::filesystem::path> result;
    for (const auto& p : pairs) result.push_back(p.second);
    return result;
}

void DicomReader::ResolveFileType()
{
    switch (m_Params.BitsAllocated) {
    case 16: m_FileDataType = FileDataType::Uint16; break;
    case 32: m_FileDataType = FileDataType::Uint32; break;
    case 64: m_FileDataType = FileDataType::Double; break;
    default: throw std::runtime_error("Unknown type");
    }
}

}  // namespace med
