"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the row tiling + gather + re-interleave
(cutrace_amd/tiling.py), with the oracle standing in for the renderer (tests are the only place
the oracle may be used).  The gathered frames on rank 0 must equal the single-process frame."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, frames, steps, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cutrace_amd as ca
    from cutrace_amd.tiling import FrameTiler
    s = ca.HostScene.load("scene/sphere_plane.json")
    s.set_size(w, h)
    tiler = FrameTiler(w, h, frames, rank, world, "cpu")
    total_rays = 0
    for step in range(steps):
        slot = step % tiler.slots
        tiler.begin(slot)
        for f in range(frames):
            r = oracle.oracle_render(s, bounces=2 + f, rows=tiler.frame_rows(f), threads=2)  # frames differ by bounces
            d, c, n = tiler.views(slot, f)
            d.copy_(torch.from_numpy(r["depth"]).reshape(-1))
            c.copy_(torch.from_numpy(r["color"]).reshape(-1))
            n.copy_(torch.from_numpy(r["normal"]).reshape(-1))
            total_rays += r["ray_count"]
        tiler.gather(slot)
    tiler.finish()
    t = torch.tensor([total_rays], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        np.savez(out_path, depth=tiler.final["depth"].numpy(), color=tiler.final["color"].numpy(),
                 normal=tiler.final["normal"].numpy(), rays=int(t[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h", [(2, 40), (3, 37)])
def test_gather_reassembles_frames(ca, tmp_path, world, h):
    w, frames, steps = 48, 2, 3
    out = str(tmp_path / "final.npz")
    mp.spawn(_worker, args=(world, _free_port(), w, h, frames, steps, out), nprocs=world, join=True)
    got = np.load(out)
    s = ca.HostScene.load("scene/sphere_plane.json")
    s.set_size(w, h)
    rays = 0
    for f in range(frames):
        full = oracle.oracle_render(s, bounces=2 + f, threads=4)
        rays += full["ray_count"]
        assert np.array_equal(got["depth"][f].view(np.uint32), full["depth"].view(np.uint32))
        assert np.array_equal(got["color"][f].view(np.uint32), full["color"].view(np.uint32))
        assert np.array_equal(got["normal"][f].view(np.uint32), full["normal"].view(np.uint32))
    assert int(got["rays"]) == rays * steps


def test_partition_covers_every_row_once():
    from cutrace_amd.tiling import part_rows, max_part_rows
    for h in (1, 7, 8, 9, 135, 1080, 4096):
        for n in (1, 2, 4, 8):
            rows = sorted(y for p in range(n) for y in part_rows(h, p, n))
            assert rows == list(range(h))
            assert max_part_rows(h, n) >= (h + n - 1) // n
    # 1080 rows over 8 ranks in 8-row blocks: 135 blocks -> 17 or 16 blocks per rank
    assert [len(part_rows(1080, p, 8)) for p in range(8)] == [136] * 7 + [128]


@pytest.mark.parametrize("h,block_rows,n", [(1080, 8, 8), (4096, 8, 8), (131, 8, 3), (100, 5, 4), (7, 8, 2), (16, 8, 3), (37, 4, 3)])
def test_row_partition_arithmetic(ca, h, block_rows, n):
    """The row map of ctr_render_multi's re-interleave kernel (csrc/ctr_multi.hip: p = (y/B) % n,
    k = (y/B/n)*B + y % B) against the partition ctr_rows defines (host library's ctr_rows_count) and the tiler's."""
    from cutrace_amd.tiling import part_rows
    seen = {}
    for y in range(h):
        blk = y // block_rows
        p, k = blk % n, (blk // n) * block_rows + y % block_rows
        seen.setdefault(p, []).append((k, y))
    total = 0
    for p in range(n):
        rows = part_rows(h, p, n, block_rows)
        assert ca.rows_count(h, (0, h, block_rows, p, n)) == len(rows)
        got = seen.get(p, [])
        assert [y for _, y in got] == rows                 # increasing y
        assert [k for k, _ in got] == list(range(len(rows)))  # compact local rows 0..len-1
        total += len(rows)
    assert total == h
