"""Image-tile partition bookkeeping (host side of the multi-GPU path; mirrors vr_tile_count / the packed layout of
vr_render_tiles / unpack_tiles_kernel in csrc/).  Tile t = ty * tiles_x + tx is owned by rank t % world; a rank's
tiles are packed in increasing t, each as 64x64 RGBA32F row-major, pixels outside the viewport = 0."""
from __future__ import annotations

import numpy as np

TILE = 64


def tiles_xy(W: int, H: int):
    return (W + TILE - 1) // TILE, (H + TILE - 1) // TILE


def tile_count(W: int, H: int, rank: int, world: int) -> int:
    tx, ty = tiles_xy(W, H)
    total = tx * ty
    return 0 if rank >= total else (total - rank + world - 1) // world


def owned_tiles(W: int, H: int, rank: int, world: int):
    tx, ty = tiles_xy(W, H)
    return list(range(rank, tx * ty, world))


def pack(frame: np.ndarray, rank: int, world: int, pad_to: int | None = None) -> np.ndarray:
    """frame [H,W,4] -> packed [n_tiles(, padded), 64, 64, 4] holding this rank's tiles."""
    H, W = frame.shape[:2]
    tx, _ = tiles_xy(W, H)
    mine = owned_tiles(W, H, rank, world)
    out = np.zeros((pad_to if pad_to is not None else len(mine), TILE, TILE, 4), dtype=np.float32)
    for i, t in enumerate(mine):
        y0, x0 = (t // tx) * TILE, (t % tx) * TILE
        h, w = min(TILE, H - y0), min(TILE, W - x0)
        out[i, :h, :w] = frame[y0:y0 + h, x0:x0 + w]
    return out


def unpack(gathered: np.ndarray, W: int, H: int, world: int) -> np.ndarray:
    """gathered [world, tiles_per_rank_max, 64, 64, 4] -> frame [H,W,4] (what unpack_tiles_kernel does)."""
    tx, _ = tiles_xy(W, H)
    frame = np.zeros((H, W, 4), dtype=np.float32)
    for r in range(world):
        for i, t in enumerate(owned_tiles(W, H, r, world)):
            y0, x0 = (t // tx) * TILE, (t % tx) * TILE
            h, w = min(TILE, H - y0), min(TILE, W - x0)
            frame[y0:y0 + h, x0:x0 + w] = gathered[r, i, :h, :w]
    return frame
