"""Test-side DICOM Part 10 writer (uncompressed little endian, explicit or implicit VR): produces the synthetic
series the DicomReader tests read back.  No DICOM data exists offline (App/assets is git-ignored in the reference)."""
import struct

LONG_VR = {"OB", "OW", "OF", "SQ", "UT", "UN"}


def _pad(b: bytes, ch: bytes) -> bytes:
    return b + ch if len(b) % 2 else b


def element(tag, vr, value, explicit=True, undefined_len=False):
    g, e = tag >> 16, tag & 0xFFFF
    if isinstance(value, str):
        value = _pad(value.encode(), b"\0" if vr == "UI" else b" ")
    head = struct.pack("<HH", g, e)
    ln = 0xFFFFFFFF if undefined_len else len(value)
    if explicit:
        if vr in LONG_VR:
            head += vr.encode() + b"\0\0" + struct.pack("<I", ln)
        else:
            head += vr.encode() + struct.pack("<H", ln)
    else:
        head += struct.pack("<I", ln)
    return head + value


def us(v):
    return struct.pack("<H", v)


def sequence_undefined(tag, items, explicit=True):
    """A sequence of undefined length whose items are of undefined length too (the reader must walk it)."""
    body = b""
    for it in items:
        body += struct.pack("<HHI", 0xFFFE, 0xE000, 0xFFFFFFFF) + it + struct.pack("<HHI", 0xFFFE, 0xE00D, 0)
    body += struct.pack("<HHI", 0xFFFE, 0xE0DD, 0)
    return element(tag, "SQ", body, explicit, undefined_len=True)


def write_slice(path, pixels, *, modality="CT", rows, cols, frames=None, bits=16, instance=None, position=(0.0, 0.0, 0.0),
                orientation=(1, 0, 0, 0, 1, 0), spacing=(1.0, 1.0), thickness=1.0, frame_uid="1.2.3.4", largest=None,
                explicit=True, with_sequence=True):
    ts = "1.2.840.10008.1.2.1" if explicit else "1.2.840.10008.1.2"
    meta = element(0x00020010, "UI", ts, True)
    meta = element(0x00020000, "UL", struct.pack("<I", len(meta)), True) + meta
    ds = b""
    ds += element(0x00080060, "CS", modality, explicit)
    if with_sequence:  # an undefined-length sequence in front of the tags the reader needs, plus a private tag
        inner = element(0x00080100, "SH", "CODE", explicit) + element(0x00090010, "LO", "PRIVATE", explicit)
        ds += sequence_undefined(0x00081140, [inner, inner], explicit)
    ds += element(0x00180050, "DS", repr(float(thickness)), explicit)
    if instance is not None:
        ds += element(0x00200013, "IS", str(instance), explicit)
    ds += element(0x00200032, "DS", "\\".join(repr(float(v)) for v in position), explicit)
    ds += element(0x00200037, "DS", "\\".join(repr(float(v)) for v in orientation), explicit)
    ds += element(0x00200052, "UI", frame_uid, explicit)
    if frames is not None:
        ds += element(0x00280008, "IS", str(frames), explicit)
    ds += element(0x00280010, "US", us(rows), explicit)
    ds += element(0x00280011, "US", us(cols), explicit)
    ds += element(0x00280030, "DS", "\\".join(repr(float(v)) for v in spacing), explicit)
    ds += element(0x00280100, "US", us(bits), explicit)
    ds += element(0x00280101, "US", us(min(bits, 12) if bits == 16 else bits), explicit)
    if largest is not None:
        ds += element(0x00280107, "US", us(largest), explicit)
    ds += element(0x7FE00010, "OW", pixels.tobytes(), explicit)
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + meta + ds)


def sequence(tag, items, explicit=True, undefined=True):
    """A sequence whose own length and item lengths are either all undefined (delimited) or all explicit."""
    if undefined:
        return sequence_undefined(tag, items, explicit)
    body = b"".join(struct.pack("<HHI", 0xFFFE, 0xE000, len(it)) + it for it in items)
    return element(tag, "SQ", body, explicit)


def write_rtstruct(path, contours, *, rois=None, colors=None, frame_uid="1.2.3.4", label="LABEL", name="NAME",
                   modality="RTSTRUCT", explicit=True, undefined=True, fmt=repr):
    """contours[c] = list of polygons, a polygon = flat list x y z x y z ...  Layout as in PS3.3 C.8.8.5 / C.8.8.6:
    Referenced Frame of Reference Sequence, Structure Set ROI Sequence (one item per ROI), ROI Contour Sequence (one item
    per ROI: display colour, Contour Sequence with one item per polygon)."""
    ts = "1.2.840.10008.1.2.1" if explicit else "1.2.840.10008.1.2"
    meta = element(0x00020010, "UI", ts, True)
    meta = element(0x00020000, "UL", struct.pack("<I", len(meta)), True) + meta
    rois = rois or [(i + 1, f"ROI{i + 1}", "MANUAL") for i in range(len(contours))]
    colors = colors or [(255, 10 * i, 0) for i in range(len(contours))]
    ds = element(0x00080060, "CS", modality, explicit)
    ds += element(0x30060002, "SH", label, explicit)
    ds += element(0x30060004, "LO", name, explicit)
    ds += sequence(0x30060010, [element(0x00200052, "UI", frame_uid, explicit)], explicit, undefined)
    ds += sequence(0x30060020, [element(0x30060022, "IS", str(n), explicit) + element(0x30060024, "UI", frame_uid, explicit) +
                                element(0x30060026, "LO", nm, explicit) + element(0x30060036, "CS", alg, explicit)
                                for n, nm, alg in rois], explicit, undefined)
    items = []
    for c, polys in enumerate(contours):
        cs = [element(0x30060042, "CS", "CLOSED_PLANAR", explicit) + element(0x30060046, "IS", str(len(p) // 3), explicit) +
              element(0x30060050, "DS", "\\".join(fmt(float(v)) for v in p), explicit) for p in polys]
        items.append(element(0x3006002A, "IS", "\\".join(str(v) for v in colors[c]), explicit) +
                     sequence(0x30060040, cs, explicit, undefined) + element(0x30060084, "IS", str(rois[c][0]), explicit))
    ds += sequence(0x30060039, items, explicit, undefined)
    with open(path, "wb") as f:
        f.write(b"\0" * 128 + b"DICM" + meta + ds)
