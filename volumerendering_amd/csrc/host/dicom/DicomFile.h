// DicomFile -- minimal DICOM Part 10 reader: what the volume path needs from the `dcm` library the reference
// links (an un-vendored fork, .gitmodules:21-24; call sites DicomReader.cpp:70-72,177-213,237,245).  Supports the
// uncompressed little-endian transfer syntaxes (Implicit VR 1.2.840.10008.1.2, Explicit VR 1.2.840.10008.1.2.1);
// the Get* accessors see the top-level, non-sequence elements (sequences are skipped over), Walk() visits every
// element in file order including those nested in sequences (what the reference's dcm::Visitor does for RTSTRUCT
// files, StructVisitor.h:18-121); encapsulated pixel data is rejected.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace med::dcmlite {

using Tag = std::uint32_t;  // (group << 16) | element, as the reference's dcm::Tag constants
namespace tags {
constexpr Tag kTransferSyntaxUID = 0x00020010;
constexpr Tag kModality = 0x00080060;
constexpr Tag kSliceThickness = 0x00180050;
constexpr Tag kInstanceNumber = 0x00200013;
constexpr Tag kImagePositionPatient = 0x00200032;
constexpr Tag kImageOrientationPatient = 0x00200037;
constexpr Tag kFrameOfReference = 0x00200052;
constexpr Tag kNumberOfFrames = 0x00280008;
constexpr Tag kRows = 0x00280010;
constexpr Tag kColumns = 0x00280011;
constexpr Tag kPixelSpacing = 0x00280030;
constexpr Tag kBitsAllocated = 0x00280100;
constexpr Tag kBitsStored = 0x00280101;
constexpr Tag kSmallestPixelValue = 0x00280106;
constexpr Tag kLargestPixelValue = 0x00280107;
constexpr Tag kPixelData = 0x7FE00010;
}  // namespace tags

// Walk() callbacks, in file order.  Element values are the raw bytes (text VRs: trailing padding still attached).
class Visitor {
public:
    virtual ~Visitor() = default;
    virtual void Element(Tag /*tag*/, const char* /*bytes*/, size_t /*length*/) {}
    virtual void BeginSequence(Tag /*tag*/) {}
    virtual void BeginItem(Tag /*sequence*/, size_t /*index*/) {}
    virtual void EndItem(Tag /*sequence*/, size_t /*index*/) {}
    virtual void EndSequence(Tag /*tag*/) {}
};

class DicomFile {
public:
    explicit DicomFile(std::string path) : m_Path(std::move(path)) {}
    bool Load();  // false: unreadable, not Part 10, unsupported transfer syntax, malformed
    const std::string& Error() const { return m_Error; }

    bool GetString(Tag tag, std::string* value) const;  // trailing spaces / NULs trimmed
    std::string GetString(Tag tag) const
    {
        std::string s;
        GetString(tag, &s);
        return s;
    }
    bool GetUint16(Tag tag, std::uint16_t* value) const;
    bool GetUint16Array(Tag tag, std::vector<std::uint16_t>* values) const;
    bool GetUint32Array(Tag tag, std::vector<std::uint32_t>* values) const;
    // Visits the main data set (after Load()).  In Implicit VR a sequence of DEFINED length cannot be told from a
    // plain element without a dictionary: the RT Structure Set sequences are known, others are reported as elements.
    bool Walk(Visitor& visitor) const;

private:
    struct Element {
        size_t offset = 0;
        std::uint32_t length = 0;
    };
    // depth = nesting level of undefined-length items / sequences; beyond kMaxDepth the file is rejected (a crafted file of
    // nested items must not be able to exhaust the stack)
    static constexpr int kMaxDepth = 64;
    bool ParseDataset(size_t pos, size_t end, bool explicitVr, bool topLevel, bool metaPass, int depth, size_t* stop);
    bool SkipSequence(size_t* pos, size_t end, std::uint32_t length, bool explicitVr, int depth);
    bool WalkDataset(size_t* pos, size_t end, bool inItem, int depth, Visitor& visitor, std::string* err) const;

    std::string m_Path, m_Error;
    std::vector<unsigned char> m_Bytes;
    std::map<Tag, Element> m_Elements;  // top-level, non-sequence elements
    size_t m_DatasetStart = 0;          // first byte behind the file meta group
    bool m_ExplicitVr = true;           // of the main data set
};

}  // namespace med::dcmlite
