"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the row tiling + gather + re-interleave
(cutrace_amd/tiling.py), with the oracle standing in for the renderer (tests are the only place
the oracle may be used).  The gathered frames on their root ranks must equal the single-process frames."""
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


def _scene(ca, out_dir, which):
    if which == "c4":  # BASELINE config 5's scene (16 translated bunnies in one room), as bench.py --workload c4 generates it
        from cutrace_amd import scenes
        return ca.HostScene.load(scenes.make_bunny_grid(os.path.join(out_dir, "c4_scene")))
    return ca.HostScene.load("scene/sphere_plane.json")


def _worker(rank, world, port, w, h, frames, steps, roots, out_dir, which="sphere_plane"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cutrace_amd as ca
    from cutrace_amd.tiling import FrameTiler
    s = _scene(ca, os.path.join(out_dir, f"rank{rank}"), which)
    s.set_size(w, h)
    tiler = FrameTiler(w, h, frames, rank, world, "cpu", roots=roots)
    total_rays = 0
    for step in range(steps):
        slot = step % tiler.slots
        tiler.begin(slot)
        for f in range(frames):
            # frames differ by bounces, steps by fudge: a piece left over from another frame or step would show
            r = oracle.oracle_render(s, bounces=2 + f, fudge=1e-3 * (1 + step), rows=tiler.frame_rows(f), threads=2)
            d, c, n = tiler.views(slot, f)
            d.copy_(torch.from_numpy(r["depth"]).reshape(-1))
            c.copy_(torch.from_numpy(r["color"]).reshape(-1))
            n.copy_(torch.from_numpy(r["normal"]).reshape(-1))
            total_rays += r["ray_count"]
        tiler.gather(slot)
    tiler.finish()
    t = torch.tensor([total_rays], dtype=torch.int64)
    dist.all_reduce(t)
    if tiler.final_frames:  # every root saves the frames it assembled (last step)
        np.savez(os.path.join(out_dir, f"final_rank{rank}.npz"), frames=np.array(tiler.final_frames),
                 depth=tiler.final["depth"].numpy(), color=tiler.final["color"].numpy(),
                 normal=tiler.final["normal"].numpy(), rays=int(t[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,frames,roots", [(2, 40, 2, "rotate"), (3, 37, 3, "rotate"), (2, 40, 3, "rotate"),
                                                    (3, 37, 2, "rank0"), (2, 24, 1, "rotate")])
def test_gather_reassembles_frames(ca, tmp_path, world, h, frames, roots):
    """Every frame of a step reaches ONE rank whole: rank f % world with rotating roots (a step of several frames),
    rank 0 otherwise — bitwise the single-process frame, and every frame exactly once."""
    w, steps = 48, 3
    mp.spawn(_worker, args=(world, _free_port(), w, h, frames, steps, roots, str(tmp_path)), nprocs=world, join=True)
    s = ca.HostScene.load("scene/sphere_plane.json")
    s.set_size(w, h)
    seen, rays, total = {}, 0, None
    for rank in range(world):
        path = tmp_path / f"final_rank{rank}.npz"
        if not path.exists():
            continue
        got = np.load(path)
        total = int(got["rays"])
        for i, f in enumerate(got["frames"]):
            assert int(f) not in seen
            seen[int(f)] = rank
            full = oracle.oracle_render(s, bounces=2 + int(f), fudge=1e-3 * steps, threads=4)
            assert np.array_equal(got["depth"][i].view(np.uint32), full["depth"].view(np.uint32)), (rank, f)
            assert np.array_equal(got["color"][i].view(np.uint32), full["color"].view(np.uint32)), (rank, f)
            assert np.array_equal(got["normal"][i].view(np.uint32), full["normal"].view(np.uint32)), (rank, f)
    rotating = roots == "rotate" and frames > 1
    assert seen == {f: (f % world if rotating else 0) for f in range(frames)}
    for step in range(steps):
        for f in range(frames):
            rays += oracle.oracle_render(s, bounces=2 + f, fudge=1e-3 * (1 + step), threads=4)["ray_count"]
    assert total == rays


def test_c4_strong_mode_world_two(ca, tmp_path):
    """bench.py --workload c4 --scaling strong --roots rank0 on two gloo ranks: ONE frame of the 4x4 bunny grid per step,
    interleaved 8-row blocks over the ranks, gathered to rank 0 — bitwise the single-process frame (oracle as renderer,
    small image: the reference's flat walk over 16 000 triangles is slow)."""
    w, h, steps, world = 24, 20, 2, 2
    mp.spawn(_worker, args=(world, _free_port(), w, h, 1, steps, "rank0", str(tmp_path), "c4"), nprocs=world, join=True)
    s = _scene(ca, str(tmp_path / "check"), "c4")
    s.set_size(w, h)
    assert not (tmp_path / "final_rank1.npz").exists()
    got = np.load(tmp_path / "final_rank0.npz")
    assert list(got["frames"]) == [0]
    full = oracle.oracle_render(s, bounces=2, fudge=1e-3 * steps, threads=4)
    for k in ("depth", "color", "normal"):
        assert np.array_equal(got[k][0].view(np.uint32), full[k].view(np.uint32)), k


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


def test_bench_plain_multi_gpu_command_launches_ranks_itself():
    """`python3 bench.py --gpus 2` without a launcher and without RANK / WORLD_SIZE: bench.py starts torch.distributed.run as
    a child (before importing torch) and hands back the child's exit code.  There is no GPU here, so the two ranks must
    both die on bench.py's "needs a GPU" assertion — and that non-zero code must come back (round 3's bench.py raised
    SystemExit("launch with torch.distributed.run ...") at this point instead: VERDICT r03 item 1)."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip("the GPU flavour of this test lives in test_gpu_multi.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert "without a launcher: starting" in r.stderr
    assert "torch.distributed.run" in r.stderr
    assert "bench.py needs a GPU" in r.stderr          # said by the child ranks
    assert "launch with torch.distributed.run" not in r.stderr
    assert r.returncode != 0
