"""CPU restatement of med::StructureFileDcm::Create3DMask (App/src/file/dicom/StructureFileDcm.cpp:49-203 and the
helpers :205-486), in plain Python loops over numpy float32 -- test infrastructure only (tests/test_rtstruct.py
checks the C++ host class against it).  Parity unpinned: the reference ships no RTSTRUCT data or expected masks.

Where the reference would index out of bounds (a neighbour row == ySize, :227; Bresenham voxels off the slice,
:156; morphology / fill on a slice that does not exist; an empty row list, :189) this restatement -- like the C++
class -- skips the access; tests keep away from those inputs except to check that nothing crashes."""
import math

import numpy as np

IGNORE, NEAREST_NEIGHBOUR, RECONSTRUCT_BRESENHAM, CLOSING, FILL, PROCESS_NON_DUPLICATES = 1, 2, 4, 8, 16, 32
f32 = np.float32


def _round(v):  # glm::round / std::round: half away from zero
    v = float(v)
    return f32(math.copysign(math.floor(abs(v) + 0.5), v))


def pixel_to_rcs_matrix(params):
    """VolumeFileDcm::InitializeTransformMatrices (VolumeFileDcm.cpp:94-121): products in double, stored as f32."""
    dj, di = params["PixelSpacing"]
    sx, sy, sz = params["ImagePositionPatient"]
    xx, xy, xz, yx, yy, yz = params["ImageOrientationPatient"]
    t = np.eye(4, dtype=np.float32)  # t[col][row] as glm
    t[0][0], t[1][0], t[3][0] = f32(xx * di), f32(yx * dj), f32(sx)
    t[0][1], t[1][1], t[3][1] = f32(xy * di), f32(yy * dj), f32(sy)
    t[0][2], t[1][2], t[3][2] = f32(xz * di), f32(yz * dj), f32(sz)
    return t


def pixel_to_rcs(t, x, y):
    """m * vec4(x, y, 0, 1) the way glm sums it: (m0*x + m1*y) + (m2*0 + m3*1)."""
    x, y = f32(x), f32(y)
    return [f32(f32(t[0][r] * x) + f32(t[1][r] * y)) + f32(f32(t[2][r] * f32(0)) + f32(t[3][r] * f32(1))) for r in range(3)]


def bresenham(start, end):
    dX, dY = int(f32(end[0]) - f32(start[0])), int(f32(end[1]) - f32(start[1]))
    cx, cy = int(start[0]), int(start[1])
    ix, iy = (1 if dX > 0 else -1), (1 if dY > 0 else -1)
    d = -abs(dX)
    if abs(dX) > abs(dY):
        step = abs(dX)
    else:
        step = abs(dY)
        d = -abs(dY)
    res = []
    for _ in range(step + 1):
        res.append((cx, cy))
        if abs(dX) > abs(dY):
            d += 2 * abs(dY)
            if d >= 0:
                cx += ix
                cy += iy
                d -= 2 * abs(dX)
            else:
                cx += ix
        else:
            d += 2 * abs(dX)
            if d >= 0:
                cx += ix
                cy += iy
                d -= 2 * abs(dY)
            else:
                cy += iy
    return res


def morph(mask, z, channels, erode):
    nz, ny, nx, _ = mask.shape
    if not 0 <= z < nz:
        return
    altered = mask[z].copy()
    for c in range(channels):
        for y in range(1, ny - 1):
            for x in range(1, nx - 1):
                missed = hit = False
                for i in (-1, 0, 1):
                    for j in (-1, 0, 1):
                        if erode and altered[y + i, x + j, c] == 0:
                            missed = True
                            break
                        if not erode and altered[y + i, x + j, c] == 1:
                            hit = True
                            break
                    if missed or hit:
                        break
                if missed:
                    mask[z, y, x, c] = 0
                if hit:
                    mask[z, y, x, c] = 1


def find_seed(mask, y_start, z, c):
    nz, ny, nx, _ = mask.shape
    if not 0 <= z < nz:
        return (-1, -1)
    y = y_start
    while 0 <= y < ny:
        x = 0
        while x < nx:
            if mask[z, y, x, c] == 1:
                ones = 0
                while x < nx:
                    v = mask[z, y, x, c]
                    if v == 1:
                        ones += 1
                    elif v == 0 and ones < 5:
                        return (x, y)
                    elif v == 0 and ones >= 5:
                        break
                    x += 1
            x += 1
        y += 1
    return (-1, -1)


def flood_fill(mask, seed, z, c):
    nz, ny, nx, _ = mask.shape
    if not 0 <= z < nz:
        return
    queue = [seed]
    while queue:
        x, y = queue.pop()
        if 0 <= x < nx and 0 <= y < ny and mask[z, y, x, c] == 0:
            mask[z, y, x, c] = 1
            queue += [(x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)]


def create_3d_mask(contours, ref_params, size, contour_ids, post_process):
    """contours[c][k] = flat float list; ref_params = dict as VolumeFile.dicom_params(); size = (x, y, z).
    Returns (mask[z, y, x, 4] float32, slice_numbers) or None where the reference returns nullptr."""
    opt = post_process
    if (opt & IGNORE) and (opt & ~1):
        opt = IGNORE
    if ref_params["Modality"] != "CT":
        return None
    final = [i - 1 for i in contour_ids if 0 < i < len(contours)]
    nx, ny, nz = size
    mask = np.zeros((nz, ny, nx, 4), dtype=np.float32)
    slice_numbers = [[] for _ in final]
    origin = [f32(v) for v in ref_params["ImagePositionPatient"]]
    spacing = [f32(ref_params["PixelSpacing"][0]), f32(ref_params["PixelSpacing"][1]), f32(ref_params["SliceThickness"])]
    t = pixel_to_rcs_matrix(ref_params)

    def inside(x, y, z):
        return 0 <= x < nx and 0 <= y < ny and 0 <= z < nz

    for l, cid in enumerate(final):
        for poly in contours[cid]:
            poly = [f32(v) for v in poly]
            if len(poly) < 3:
                continue
            slice_number = -1
            ycoords = []
            for j in range(0, len(poly) - 2, 3):
                cp = poly[j:j + 3]
                voxel = [f32(f32(cp[a] - origin[a]) / spacing[a]) for a in range(3)]
                voxel[2] = f32(abs(voxel[2]))
                voxel = [_round(v) for v in voxel]
                vx, vy, vz = int(voxel[0]), int(voxel[1]), int(voxel[2])
                slice_number = vz
                if not inside(vx, vy, vz):
                    continue
                if (mask[vz, vy, vx, l] == 1 or (opt & PROCESS_NON_DUPLICATES)) and not (opt & IGNORE):
                    if opt & NEAREST_NEIGHBOUR:
                        best, sub = np.finfo(np.float32).max, (vx, vy)
                        for i in (-1, 0, 1):
                            for jj in (-1, 0, 1):
                                if i == 0 and jj == 0:
                                    continue
                                nxx, nyy = vx + jj, vy + i
                                if 0 <= nxx < nx and 0 <= nyy <= ny:
                                    rcs = pixel_to_rcs(t, nxx, nyy)
                                    dd = [f32(cp[a] - rcs[a]) for a in range(3)]
                                    dist = f32(np.sqrt(f32(f32(f32(dd[0] * dd[0]) + f32(dd[1] * dd[1])) + f32(dd[2] * dd[2]))))
                                    if dist < best:
                                        best, sub = dist, (nxx, nyy)
                        if inside(sub[0], sub[1], vz):
                            mask[vz, sub[1], sub[0], l] = 1
                    if (opt & RECONSTRUCT_BRESENHAM) and j + 6 <= len(poly):
                        cn = poly[j + 3:j + 6]
                        nxt = [_round(f32(f32(cn[a] - origin[a]) / spacing[a])) for a in range(3)]
                        line = bresenham(voxel, nxt)
                        for (bx, by) in line[:-1]:
                            if inside(bx, by, vz):
                                mask[vz, by, bx, l] = 1
                ycoords.append(voxel[1])
                mask[vz, vy, vx, l] = 1
            slice_numbers[l].append(slice_number)
            if opt & CLOSING:
                morph(mask, slice_number, len(final), False)
                morph(mask, slice_number, len(final), True)
            if (opt & FILL) and ycoords:
                ycoords.sort()
                seed = find_seed(mask, int(ycoords[len(ycoords) // 2]), slice_number, l)
                if seed[0] != -1:
                    flood_fill(mask, seed, slice_number, l)
    return mask, slice_numbers
