#include "DicomFile.h"

#include <cstring>
#include <fstream>

namespace med::dcmlite {

namespace {
constexpr std::uint32_t kUndefined = 0xFFFFFFFFu;
constexpr Tag kItem = 0xFFFEE000, kItemDelim = 0xFFFEE00D, kSeqDelim = 0xFFFEE0DD;

inline std::uint16_t rd16(const unsigned char* p) { return static_cast<std::uint16_t>(p[0] | (p[1] << 8)); }
inline std::uint32_t rd32(const unsigned char* p)
{
    return static_cast<std::uint32_t>(p[0]) | (static_cast<std::uint32_t>(p[1]) << 8) |
           (static_cast<std::uint32_t>(p[2]) << 16) | (static_cast<std::uint32_t>(p[3]) << 24);
}
inline bool long_vr(const unsigned char* vr)
{
    static const char* kLong[] = {"OB", "OD", "OF", "OL", "OV", "OW", "SQ", "UC", "UN", "UR", "UT"};
    for (const char* l : kLong)
        if (vr[0] == l[0] && vr[1] == l[1]) return true;
    return false;
}
}  // namespace

bool DicomFile::Load()
{
    m_Elements.clear();
    std::ifstream f(m_Path, std::ios::binary);
    if (!f) {
        m_Error = "cannot open file";
        return false;
    }
    m_Bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    if (m_Bytes.size() < 132 || std::memcmp(m_Bytes.data() + 128, "DICM", 4) != 0) {
        m_Error = "not a DICOM Part 10 file";
        return false;
    }
    // file meta information (group 0002) is always Explicit VR Little Endian
    size_t pos = 132;
    if (!ParseDataset(pos, m_Bytes.size(), /*explicitVr=*/true, /*topLevel=*/true, /*metaPass=*/true, 0, &pos)) return false;
    if (m_Elements.empty()) {
        m_Error = "no file meta information (group 0002) behind the DICM prefix";
        return false;
    }
    std::string ts = GetString(tags::kTransferSyntaxUID);
    bool explicitVr;
    if (ts == "1.2.840.10008.1.2.1") explicitVr = true;
    else if (ts == "1.2.840.10008.1.2" || ts.empty()) explicitVr = false;
    else {
        m_Error = "unsupported transfer syntax " + ts + " (only uncompressed little endian)";
        return false;
    }
    m_DatasetStart = pos;
    m_ExplicitVr = explicitVr;
    return ParseDataset(pos, m_Bytes.size(), explicitVr, /*topLevel=*/true, /*metaPass=*/false, 0, &pos);
}

namespace {
// sequences of the RT Structure Set module (PS3.3 C.8.8.5 / C.8.8.6) and the ones nested in them
bool known_sequence(Tag t)
{
    switch (t) {
    case 0x30060010: case 0x30060012: case 0x30060014: case 0x30060016: case 0x30060020: case 0x30060030:
    case 0x30060039: case 0x30060040: case 0x30060080: case 0x30060086: case 0x300600A0: case 0x300600B0:
    case 0x00081140: case 0x00081155: case 0x00081115: case 0x0008114A:
        return true;
    default:
        return false;
    }
}
}  // namespace

bool DicomFile::Walk(Visitor& visitor) const
{
    if (m_Bytes.empty() || m_DatasetStart == 0) return false;
    size_t pos = m_DatasetStart;
    std::string err;
    return WalkDataset(&pos, m_Bytes.size(), false, 0, visitor, &err);
}

// Elements from *pos to `end` (or to the item delimiter when inItem and the item has undefined length).
bool DicomFile::WalkDataset(size_t* pos, size_t end, bool inItem, int depth, Visitor& visitor, std::string* err) const
{
    if (depth > kMaxDepth) {
        *err = "sequences nested too deeply";
        return false;
    }
    while (*pos + 8 <= end) {
        const unsigned char* p = m_Bytes.data() + *pos;
        const Tag tag = (static_cast<Tag>(rd16(p)) << 16) | rd16(p + 2);
        if (tag == kItemDelim) {
            *pos += 8;
            return inItem;
        }
        if (tag == kSeqDelim) return false;  // a sequence delimiter is consumed by the sequence loop below
        std::uint32_t length;
        bool isSeq;
        size_t header;
        if (m_ExplicitVr) {
            const unsigned char* vr = p + 4;
            if (long_vr(vr)) {
                if (*pos + 12 > end) return false;
                length = rd32(p + 8);
                header = 12;
            } else {
                length = rd16(p + 6);
                header = 8;
            }
            isSeq = (vr[0] == 'S' && vr[1] == 'Q') || (length == kUndefined && tag != tags::kPixelData);
        } else {
            length = rd32(p + 4);
            header = 8;
            isSeq = (length == kUndefined && tag != tags::kPixelData) || known_sequence(tag);
        }
        *pos += header;
        if (!isSeq) {
            if (length == kUndefined || *pos + length > end) {
                *err = "element runs past the end of the file";
                return false;
            }
            visitor.Element(tag, reinterpret_cast<const char*>(m_Bytes.data() + *pos), length);
            *pos += length;
            continue;
        }
        const size_t seqEnd = length == kUndefined ? end : *pos + length;
        if (seqEnd > end) return false;
        visitor.BeginSequence(tag);
        size_t index = 0;
        bool closed = length != kUndefined;
        while (*pos + 8 <= seqEnd) {
            const unsigned char* q = m_Bytes.data() + *pos;
            const Tag itag = (static_cast<Tag>(rd16(q)) << 16) | rd16(q + 2);
            const std::uint32_t ilen = rd32(q + 4);
            *pos += 8;
            if (itag == kSeqDelim) {
                closed = true;
                break;
            }
            if (itag != kItem) {
                *err = "malformed sequence";
                return false;
            }
            visitor.BeginItem(tag, index);
            if (ilen == kUndefined) {
                if (!WalkDataset(pos, seqEnd, true, depth + 1, visitor, err)) return false;
            } else {
                size_t ipos = *pos;
                const size_t iend = ipos + ilen;
                if (iend > seqEnd) return false;
                if (ipos < iend && !WalkDataset(&ipos, iend, false, depth + 1, visitor, err)) return false;
                *pos = iend;
            }
            visitor.EndItem(tag, index);
            ++index;
        }
        if (!closed) return false;
        visitor.EndSequence(tag);
    }
    return !inItem;  // an item of undefined length must end with its delimiter
}

// Parses elements from pos; at top level it stops after the meta group when asked to switch syntax (group > 0002
// seen while parsing the meta header), otherwise at `end` or at an item delimiter (nested use).
bool DicomFile::ParseDataset(size_t pos, size_t end, bool explicitVr, bool topLevel, bool metaPass, int depth, size_t* stop)
{
    if (depth > kMaxDepth) {
        m_Error = "sequences nested too deeply";
        return false;
    }
    while (pos + 8 <= end) {
        const unsigned char* p = m_Bytes.data() + pos;
        const Tag tag = (static_cast<Tag>(rd16(p)) << 16) | rd16(p + 2);
        if (metaPass && (tag >> 16) != 0x0002) break;  // end of the file meta group
        if (tag == kItemDelim || tag == kSeqDelim) {
            pos += 8;
            *stop = pos;
            return true;
        }
        std::uint32_t length;
        bool isSeq = false;
        size_t header;
        if (explicitVr && (tag >> 16) != 0xFFFE) {
            const unsigned char* vr = p + 4;
            if (long_vr(vr)) {
                if (pos + 12 > end) break;
                length = rd32(p + 8);
                header = 12;
            } else {
                length = rd16(p + 6);
                header = 8;
            }
            isSeq = vr[0] == 'S' && vr[1] == 'Q';
        } else {
            length = rd32(p + 4);
            header = 8;
            isSeq = length == kUndefined && tag != tags::kPixelData;  // implicit VR: only undefined lengths need walking
        }
        pos += header;
        if (isSeq || (length == kUndefined && tag != tags::kPixelData)) {
            if (!SkipSequence(&pos, end, length, explicitVr, depth)) return false;
            continue;
        }
        if (length == kUndefined) {
            m_Error = "encapsulated (compressed) pixel data is not supported";
            return false;
        }
        if (pos + length > end) {
            m_Error = "element runs past the end of the file";
            return false;
        }
        if (topLevel) m_Elements[tag] = Element{pos, length};
        pos += length;
    }
    *stop = pos;
    return true;
}

bool DicomFile::SkipSequence(size_t* pos, size_t end, std::uint32_t length, bool explicitVr, int depth)
{
    if (length != kUndefined) {
        if (*pos + length > end) {
            m_Error = "sequence runs past the end of the file";
            return false;
        }
        *pos += length;
        return true;
    }
    // undefined length: items until the sequence delimiter
    while (*pos + 8 <= end) {
        const unsigned char* p = m_Bytes.data() + *pos;
        const Tag tag = (static_cast<Tag>(rd16(p)) << 16) | rd16(p + 2);
        const std::uint32_t len = rd32(p + 4);
        *pos += 8;
        if (tag == kSeqDelim) return true;
        if (tag != kItem) {
            m_Error = "malformed sequence";
            return false;
        }
        if (len != kUndefined) {
            if (*pos + len > end) {
                m_Error = "item runs past the end of the file";
                return false;
            }
            *pos += len;
        } else {
            size_t stop = *pos;
            if (!ParseDataset(*pos, end, explicitVr, /*topLevel=*/false, /*metaPass=*/false, depth + 1, &stop)) return false;
            *pos = stop;
        }
    }
    m_Error = "unterminated sequence";
    return false;
}

bool DicomFile::GetString(Tag tag, std::string* value) const
{
    auto it = m_Elements.find(tag);
    if (it == m_Elements.end()) return false;
    const char* b = reinterpret_cast<const char*>(m_Bytes.data() + it->second.offset);
    size_t n = it->second.length;
    while (n > 0 && (b[n - 1] == ' ' || b[n - 1] == '\0')) --n;
    value->assign(b, n);
    return true;
}

bool DicomFile::GetUint16(Tag tag, std::uint16_t* value) const
{
    auto it = m_Elements.find(tag);
    if (it == m_Elements.end() || it->second.length < 2) return false;
    *value = rd16(m_Bytes.data() + it->second.offset);
    return true;
}

bool DicomFile::GetUint16Array(Tag tag, std::vector<std::uint16_t>* values) const
{
    auto it = m_Elements.find(tag);
    if (it == m_Elements.end()) return false;
    const size_t n = it->second.length / 2;
    values->resize(n);
    const unsigned char* b = m_Bytes.data() + it->second.offset;
    for (size_t i = 0; i < n; ++i) (*values)[i] = rd16(b + 2 * i);
    return true;
}

bool DicomFile::GetUint32Array(Tag tag, std::vector<std::uint32_t>* values) const
{
    auto it = m_Elements.find(tag);
    if (it == m_Elements.end()) return false;
    const size_t n = it->second.length / 4;
    values->resize(n);
    const unsigned char* b = m_Bytes.data() + it->second.offset;
    for (size_t i = 0; i < n; ++i) (*values)[i] = rd32(b + 4 * i);
    return true;
}

}  // namespace med::dcmlite
