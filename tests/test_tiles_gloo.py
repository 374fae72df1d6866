"""The N > 1 path on CPU: two gloo ranks each produce the tiles they own (with the oracle standing in for the GPU
kernel), rank 0 gathers and un-permutes them exactly as bench.py does with RCCL, and the result must equal the
single-process frame bit for bit.  Also pins the partition bookkeeping against brute force."""
import os
import socket
import sys

import numpy as np
import pytest

from volumerendering_amd import tiles

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("W,H", [(200, 150), (64, 64), (65, 1), (1920, 1080)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_partition_bookkeeping(W, H, world):
    tx, ty = tiles.tiles_xy(W, H)
    seen = []
    for r in range(world):
        mine = tiles.owned_tiles(W, H, r, world)
        assert len(mine) == tiles.tile_count(W, H, r, world)
        assert all(t % world == r for t in mine)
        seen += mine
    assert sorted(seen) == list(range(tx * ty))
    assert tiles.tile_count(W, H, 0, world) == max(tiles.tile_count(W, H, r, world) for r in range(world))
    rng = np.random.default_rng(1)
    frame = rng.random((H, W, 4), dtype=np.float32)
    tpr = tiles.tile_count(W, H, 0, world)
    gathered = np.stack([tiles.pack(frame, r, world, pad_to=tpr) for r in range(world)])
    assert np.array_equal(tiles.unpack(gathered, W, H, world), frame)


def _worker(rank, world, port, W, H, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
    import torch
    import torch.distributed as dist
    import host_ref as hr
    import oracle_binding as ob
    import vrtest as vt
    from volumerendering_amd import capi, tiles as tl

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vols, tfs = vt.scene(capi.LIGHT, n=16)
    u = hr.make_uniforms(W, H, steps_count=27, step_size=1 / 16)
    # every rank renders ONLY the pixels of the tiles it owns
    tx, _ = tl.tiles_xy(W, H)
    mine = tl.owned_tiles(W, H, rank, world)
    frame = np.zeros((H, W, 4), dtype=np.float32)
    samples = 0
    for t in mine:
        y0, x0 = (t // tx) * tl.TILE, (t % tx) * tl.TILE
        ys, xs = np.meshgrid(np.arange(y0, min(y0 + tl.TILE, H)), np.arange(x0, min(x0 + tl.TILE, W)), indexing="ij")
        pxy = np.stack([xs.ravel(), ys.ravel()], axis=1)
        out, n = ob.render_pixels(capi.LIGHT, u, vols, tfs, W, H, pxy)
        frame[ys.ravel(), xs.ravel()] = out
        samples += n
    tpr = tl.tile_count(W, H, 0, world)
    packed = torch.from_numpy(tl.pack(frame, rank, world, pad_to=tpr).reshape(-1))
    gathered = torch.zeros((world, packed.numel()), dtype=torch.float32) if rank == 0 else None
    dist.gather(packed, [gathered[r] for r in range(world)] if rank == 0 else None, dst=0)
    total = torch.tensor([samples], dtype=torch.int64)
    dist.all_reduce(total)
    if rank == 0:
        full = tl.unpack(gathered.numpy().reshape(world, tpr, tl.TILE, tl.TILE, 4), W, H, world)
        np.save(out_path, full)
        np.save(out_path + ".n.npy", total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_the_frame(tmp_path):
    import torch.multiprocessing as mp
    import host_ref as hr
    import oracle_binding as ob
    import vrtest as vt
    from volumerendering_amd import capi

    W, H, world = 200, 150, 2
    out = str(tmp_path / "frame.npy")
    port = free_port()
    mp.spawn(_worker, args=(world, port, W, H, out), nprocs=world, join=True)
    got = np.load(out)
    n_got = int(np.load(out + ".n.npy")[0])
    vols, tfs = vt.scene(capi.LIGHT, n=16)
    u = hr.make_uniforms(W, H, steps_count=27, step_size=1 / 16)
    ref, n_ref, _ = ob.render(capi.LIGHT, u, vols, tfs, W, H, nthreads=4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert n_got == n_ref
