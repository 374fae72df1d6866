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
template std::array<int, 3> ParseStringToNumArr<int, 3>(const std::string&);
template std::array<double, 1> ParseStringToNumArr<double, 1>(const std::string&);
template std::array<double, 2> ParseStringToNumArr<double, 2>(const std::string&);
template std::array<double, 3> ParseStringToNumArr<double, 3>(const std::string&);
template std::array<double, 6> ParseStringToNumArr<double, 6>(const std::string&);

std::vector<float> ParseContours(const std::string& str)
{
    std::vector<float> res;
    std::stringstream ss(str);
    std::string temp;
    while (std::getline(ss, temp, '\\')) {
        try {
            res.push_back(std::stof(temp));
        } catch (const std::exception&) {  // "Parsing contour data failed!"
        }
    }
    return res;
}

namespace {
// What the reference's StructVisitor collects while the dcm library walks the data set (StructVisitor.h:22-121):
// every item of (3006,0039) ROI Contour Sequence opens a contour, every (3006,0050) Contour Data below it adds a
// polygon; every item of (3006,0020) Structure Set ROI Sequence opens an ROI description.
class StructCollector : public dcmlite::Visitor {
public:
    void BeginItem(dcmlite::Tag sequence, size_t) override
    {
        if (sequence == 0x30060039) ContourData.emplace_back();
        else if (sequence == 0x30060020) Params.StructureSetROISequence.emplace_back();
    }
    void Element(dcmlite::Tag tag, const char* bytes, size_t length) override
    {
        auto text = [&] {
            size_t n = length;
            while (n > 0 && (bytes[n - 1] == ' ' || bytes[n - 1] == '\0')) --n;
            return std::string(bytes, n);
        };
        switch (tag) {
        case 0x30060050:  // Contour Data
            if (!ContourData.empty()) ContourData.back().push_back(ParseContours(text()));
            break;
        case 0x3006002A: {  // ROI Display Color
            const auto d = ParseStringToNumArr<int, 3>(text());
            constexpr float factor = 1 / 255.0f;
            Params.DisplayColors.push_back({d[0] * factor, d[1] * factor, d[2] * factor});
            break;
        }
        case 0x30060022:  // ROI Number
            if (!Params.StructureSetROISequence.empty()) Params.StructureSetROISequence.back().Number = ParseStringToNumArr<int, 1>(text())[0];
            break;
        case 0x30060026:  // ROI Name
            if (!Params.StructureSetROISequence.empty()) Params.StructureSetROISequence.back().Name = text();
            break;
        case 0x30060036:  // ROI Generation Algorithm
            if (!Params.StructureSetROISequence.empty()) Params.StructureSetROISequence.back().AlgorithmType = text();
            break;
        case 0x00200052: Params.FrameOfReference = text(); break;  // wherever it occurs; the last one wins
        case 0x30060004: Params.Name = text(); break;
        case 0x30060002: Params.Label = text(); break;
        default: break;
        }
    }
    DicomStructParams Params;
    std::vector<std::vector<std::vector<float>>> ContourData;
};
}  // namespace

std::shared_ptr<StructureFileDcm> DicomReader::ReadStructFile(std::filesystem::path name)
{
    if (std::filesystem::is_directory(name)) {
        std::vector<std::filesystem::path> files;
        for (const auto& e : std::filesystem::directory_iterator(name))
            if (e.path().extension().string() == ".dcm") files.push_back(e.path());
        if (files.size() != 1) return nullptr;  // "No structure files found!" / "Multiple contour files are not allowed!"
        name = files[0];
    }
    if (!IsDicomFile(name)) return nullptr;
    DicomFile f(name.string());
    if (!f.Load()) return nullptr;
    if (ResolveModality(f.GetString(tags::kModality)) != DicomModality::RTSTRUCT) return nullptr;  // "Not a valid struct file"
    StructCollector visitor;
    if (!f.Walk(visitor)) return nullptr;
    return std::make_shared<StructureFileDcm>(name, visitor.Params, visitor.ContourData);
}

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
        // every file must deliver exactly the pixels the first slice's geometry promises (the reference asserts
        // vec.size() == X * Y * frames in ReadData, DicomReader.cpp:238,246): a slice with other Rows / Columns or a short
        // PixelData element would leave the declared size larger than the data behind it
        const size_t before = reader.m_Data.size();
        reader.ReadData(f);
        const size_t expect = static_cast<size_t>(reader.m_Params.X) * reader.m_Params.Y * ((isDir && !hasSoloFile) ? 1 : reader.m_Params.Z);
        if (reader.m_Data.size() - before != expect)
            throw std::runtime_error("Pixel data of " + file.string() + " holds " + std::to_string(reader.m_Data.size() - before) +
                                     " values, expected " + std::to_string(expect) + " (Rows x Columns x frames of the first slice)");
    }
    if (isDir && !hasSoloFile) {
        if (numberOfFiles > 0xFFFFu) throw std::runtime_error("More than 65535 slices");
        reader.m_Params.Z = static_cast<std::uint16_t>(numberOfFiles);
    }
    if (reader.m_Params.X == 0 || reader.m_Params.Y == 0 || reader.m_Params.Z == 0) throw std::runtime_error("Empty volume (Rows, Columns or frames is 0)");
    if (reader.m_Data.size() != static_cast<size_t>(reader.m_Params.X) * reader.m_Params.Y * reader.m_Params.Z)
        throw std::runtime_error("Pixel data does not match Rows x Columns x slices");
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
