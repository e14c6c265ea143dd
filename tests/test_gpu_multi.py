"""ctr_render_multi (include/cutrace_amd.h): one frame row-tiled over several devices of ONE process, gathered to
device 0 and re-interleaved by a HIP kernel.  The GPU box has one MI355X, so groups of 2..4 list device 0 several
times: every part is then rendered by its own scene handle on its own stream and moved with peer copies instead
of RCCL (RCCL refuses one device twice) — partition, compact buffers, gather targets and the re-interleave kernel
are exactly what an 8-GPU node runs.  Reference boundary: inc/kernel.hpp:86-130."""
import os
import subprocess

import numpy as np
import pytest

from tests.conftest import load_scene
from tests.util import same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    return all(same_bits(a[k], b[k]) for k in ("depth", "color", "normal")) and a["ray_count"] == b["ray_count"] and \
        a["max_depth"] == b["max_depth"]


def test_one_device_group_equals_ctr_render(ca):
    s = load_scene(ca, "bunny", 256, 144)
    want = ca.DeviceScene(s).render(bounces=5)
    m = ca.MultiScene(s, [0])
    assert m.transport == "single"
    got = m.render(bounces=5)
    assert _same(got, want)
    m.close()


@pytest.mark.parametrize("n_dev,w,h,block_rows", [(2, 256, 144, 8), (3, 250, 131, 8), (4, 192, 100, 5), (2, 64, 7, 8), (3, 40, 16, 8)])
def test_group_on_one_gpu_reassembles_bitwise(ca, n_dev, w, h, block_rows):
    s = load_scene(ca, "bunny", w, h)
    want = ca.DeviceScene(s).render(bounces=3)
    m = ca.MultiScene(s, [0] * n_dev)
    assert m.transport == "peer-copy"
    got = m.render(bounces=3, block_rows=block_rows)
    assert _same(got, want)
    assert len(got["kernel_ms_per_device"]) == n_dev
    # a second frame through the same buffers (tile-order feedback active), then another size
    assert _same(m.render(bounces=3, block_rows=block_rows), want)
    m.set_size(w - 8, h - 3)
    s.set_size(w - 8, h - 3)
    assert _same(m.render(bounces=3, block_rows=block_rows), ca.DeviceScene(s).render(bounces=3))
    m.close()


def test_full_size_group_of_four(ca):
    """bunny.json@1920x1080, four parts on the one GPU: bitwise the single-device frame, same ray count."""
    s = load_scene(ca, "bunny")
    want = ca.DeviceScene(s).render(bounces=5)
    m = ca.MultiScene(s, [0, 0, 0, 0])
    got = m.render(bounces=5)
    assert got["ray_count"] == 64278888
    assert _same(got, want)
    m.close()


def test_reinterleave_entry_point(ca):
    """ctr_reinterleave_device on its own (what bench.py's rank 0 calls after the RCCL gather)."""
    import ctypes as C
    import torch
    from cutrace_amd import _lib
    from cutrace_amd.tiling import part_rows
    w, h, n, B = 52, 37, 3, 4
    rng = np.random.default_rng(5)
    full = {k: rng.standard_normal((h, w) + sh).astype(np.float32) for k, sh in (("depth", ()), ("color", (3,)), ("normal", (3,)))}
    parts = (_lib.ReintPart * n)()
    keep = []
    for p in range(n):
        rows = part_rows(h, p, n, B)
        for k, fld in (("depth", "d_depth"), ("color", "d_color3"), ("normal", "d_normal3")):
            t = torch.from_numpy(np.ascontiguousarray(full[k][rows])).cuda()
            keep.append(t)
            setattr(parts[p], fld, t.data_ptr())
    out = {k: torch.zeros(v.shape, dtype=torch.float32, device="cuda") for k, v in full.items()}
    st = _lib.hip_lib().ctr_reinterleave_device(parts, n, B, w, h, out["depth"].data_ptr(), out["color"].data_ptr(),
                                                out["normal"].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    for k in full:
        assert np.array_equal(out[k].cpu().numpy(), full[k])


def test_cli_device_list(ca, tmp_path):
    """`CUTRACE_DEVICE_LIST=0,0 cutrace scene.json`: the multi-device CLI path writes the same three files."""
    from cutrace_amd import build
    cli = build.build_cli()
    env = dict(os.environ, CUTRACE_WIDTH="160", CUTRACE_HEIGHT="90")
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir(); two.mkdir()
    for d, extra in ((one, {}), (two, {"CUTRACE_DEVICE_LIST": "0,0"})):
        os.symlink(os.path.join(ROOT, "scene"), d / "scene")  # mesh paths are relative to the CWD (schema.md:73-74)
        r = subprocess.run([cli, "scene/bunny.json"], cwd=d, env=dict(env, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        assert "Render time was " in r.stdout
    for f in ("frame.jpg", "depth_map.jpg", "normal_map.jpg"):
        assert (one / f).read_bytes() == (two / f).read_bytes(), f


def test_nccl_backend_world_one(ca):
    """The torch.distributed path of bench.py with the nccl (= RCCL) backend, world size 1: process group, tiler,
    device-buffer render and the rank-0 hand-over run on the real backend (a one-GPU box cannot hold more ranks)."""
    import socket
    import torch
    import torch.distributed as dist
    from cutrace_amd.tiling import FrameTiler
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        s = load_scene(ca, "bunny", 128, 72)
        ds = ca.DeviceScene(s)
        tiler = FrameTiler(128, 72, 1, 0, 1, dev)
        buf = tiler.local[0]
        d0, c0, n0, _ = tiler.sec
        esz = buf.element_size()
        ds.render_device(buf.data_ptr() + d0 * esz, buf.data_ptr() + c0 * esz, buf.data_ptr() + n0 * esz,
                         stream=torch.cuda.current_stream().cuda_stream, bounces=5)
        tiler.gather(0)
        tiler.finish()
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)     # one real RCCL collective on this box
        dist.barrier()
        # the primitive the tiler's exchange is made of (grouped ncclSend / ncclRecv through batch_isend_irecv), on the
        # real backend: three pieces sent to and received from the only rank there is, on a side stream like the tiler's
        src = [torch.arange(1000 * (k + 1), dtype=torch.float32, device=dev) + k for k in range(3)]
        dst = [torch.zeros_like(x) for x in src]
        ops = [dist.P2POp(dist.isend, x, 0) for x in src] + [dist.P2POp(dist.irecv, y, 0) for y in dst]
        works = dist.batch_isend_irecv(ops)
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            for wk in works:
                wk.wait()
            got = [y.clone() for y in dst]
        side.synchronize()
        for x, y in zip(src, got):
            assert torch.equal(x, y)
        want = ds.render(bounces=5)
        assert same_bits(tiler.final["depth"][0].cpu().numpy(), want["depth"])
        assert same_bits(tiler.final["color"][0].cpu().numpy(), want["color"])
        assert float(t[0]) == 1.0
    finally:
        dist.destroy_process_group()


def test_rccl_plumbing_self_send_recv(ca):
    """The RCCL code of ctr_render_multi on the one GPU there is: librccl is dlopen'ed, a one-rank communicator
    created, and the rendered frame goes through ONE grouped ncclSend / ncclRecv (rank 0 to rank 0) before it is
    delivered — the same calls, stream order and buffers an N-GPU group uses, minus the second device."""
    s = load_scene(ca, "bunny", 256, 144)
    want = ca.DeviceScene(s).render(bounces=5)
    os.environ["CUTRACE_MULTI_TRANSPORT"] = "rccl-self"
    try:
        m = ca.MultiScene(s, [0])
    finally:
        del os.environ["CUTRACE_MULTI_TRANSPORT"]
    if m.transport != "rccl-self":
        pytest.skip("librccl.so could not be loaded on this box")
    got = m.render(bounces=5)
    assert _same(got, want)
    assert _same(m.render(bounces=5), want)
    m.close()


@pytest.mark.parametrize("n_dev", [1, 2, 4])
def test_page_locked_destination_is_written_by_the_device(ca, n_dev):
    """A page-locked destination (ctr_frame_alloc) is written by device 0's kernels themselves — the render kernel
    for one device, the re-interleave kernel for several — instead of a D2H copy afterwards: same bytes as with
    CTR_VAR_NO_DIRECT and as into pageable memory, twice in a row (the block is reused)."""
    s = load_scene(ca, "bunny", 480, 270)
    want = ca.DeviceScene(s).render(bounces=4)
    m = ca.MultiScene(s, [0] * n_dev)
    for variant in (0, ca.VAR_NO_DIRECT, 0):
        m.set_variant(variant)
        got = m.render(bounces=4, pinned=True)
        assert _same(got, want), (n_dev, variant)
        got["depth"][:] = 0.0     # the next call must rewrite every pixel
        got["color"][:] = 0.0
        got["normal"][:] = 0.0
    m.close()


@pytest.mark.parametrize("n_dev,pinned", [(4, True), (4, False), (2, True), (1, True), (1, False)])
def test_pipelined_submit_wait_equals_synchronous(ca, n_dev, pinned):
    """ctr_multi_submit / ctr_multi_wait (two frames in flight: frame k is gathered, re-interleaved and copied out while
    frame k+1 renders) against ctr_render_multi, bit for bit, on a group that lists device 0 n times.  The frames differ
    (the image size changes the camera rays; bounces alternate), so a slot handed over too early or a stale buffer shows."""
    s = load_scene(ca, "bunny", 320, 180)
    m = ca.MultiScene(s, [0] * n_dev)
    want = [m.render(bounces=b) for b in (2, 4, 3, 5, 1)]
    want = [{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()} for r in want]
    if pinned:
        frames = [m.alloc_frame() for _ in range(3)]
    else:
        frames = [dict(depth=np.empty((180, 320), np.float32), color=np.empty((180, 320, 3), np.float32),
                       normal=np.empty((180, 320, 3), np.float32)) for _ in range(3)]
    bs = (2, 4, 3, 5, 1)
    m.submit(frames[0], bounces=bs[0])
    for k in range(1, len(bs) + 1):
        if k < len(bs):
            m.submit(frames[k % 3], bounces=bs[k])          # second frame in flight
        st = m.wait()                                       # frame k-1
        got = frames[(k - 1) % 3]
        for key in ("depth", "color", "normal"):
            assert same_bits(got[key], want[k - 1][key]), (k - 1, key)
        assert st["ray_count"] == want[k - 1]["ray_count"] and st["max_depth"] == want[k - 1]["max_depth"]
        got["depth"][:] = 0.0
        got["color"][:] = 0.0
    # misuse is refused, not queued: a wait without a frame, a third frame in flight, a synchronous frame in between
    with pytest.raises(RuntimeError):
        m.wait()
    if n_dev > 1:
        m.submit(frames[0], bounces=1)
        m.submit(frames[1], bounces=1)
        with pytest.raises(RuntimeError):
            m.submit(frames[2], bounces=1)
        with pytest.raises(RuntimeError):
            m.render(bounces=1)
        m.wait()
        m.wait()
    assert _same(m.render(bounces=5), want[3])
    if pinned:
        for f in frames:
            m.free_frame(f)
    m.close()


def test_group_grows_then_renders(ca):
    """A group created small and grown by ctr_multi_set_size must regrow EVERY buffer — round 2 kept the rccl-self copy of
    part 0 at its first size and the next receive overran it."""
    for transport in ("peer", "rccl-self"):
        n_dev = 1 if transport == "rccl-self" else 3
        s = load_scene(ca, "bunny", 64, 36)
        os.environ["CUTRACE_MULTI_TRANSPORT"] = transport
        try:
            m = ca.MultiScene(s, [0] * n_dev)
        finally:
            del os.environ["CUTRACE_MULTI_TRANSPORT"]
        if transport == "rccl-self" and m.transport != "rccl-self":
            m.close()
            continue
        m.render(bounces=2)
        for (w, h) in ((256, 144), (640, 360), (200, 100)):
            m.set_size(w, h)
            s.set_size(w, h)
            assert _same(m.render(bounces=3), ca.DeviceScene(s).render(bounces=3)), (transport, w, h)
        m.close()


def test_bench_c4_strong_two_ranks_on_one_gpu(ca):
    """bench.py --workload c4 --scaling strong --roots rank0 --check as the driver would launch it on 2 GPUs, rehearsed on
    the one GPU there is (both ranks on device 0, gloo standing in for RCCL, which refuses one device twice): one frame
    of the 4x4 grid per step, row-tiled over the ranks, gathered to rank 0, every gathered frame bitwise the
    single-process render; the JSON line carries the strong-scaling workload."""
    import json
    import socket
    import sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, CUTRACE_BENCH_SHARE_GPU="1", CUTRACE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "c4", "--scaling", "strong", "--roots", "rank0", "--width", "512", "--height", "512", "--check",
           "--skip-probe", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gathered frame(s) bitwise equal" in r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["frames_per_step"] == 1
    assert "bunny_grid4x4.json@512x512" in line["config"]["workload"] and "gather to rank 0" in line["config"]["workload"]


def test_bench_default_line_two_ranks_on_one_gpu(ca):
    """The driver's multi-GPU command as it stands (weak scaling on bunny.json, rotating roots) rehearsed with two ranks on
    the one GPU (gloo instead of RCCL): it must still end in ONE JSON line, now carrying BASELINE config 5 measured in the
    same run (config.c4_strong: the 4x4 grid @4096x4096, one frame per step row-tiled over both ranks, gathered to rank 0)."""
    import json
    import socket
    import sys
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, CUTRACE_BENCH_SHARE_GPU="1", CUTRACE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--check"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_per_step"] == 2
    c4 = line["config"]["c4_strong"]
    assert c4["n_gpus"] == 2 and c4["rays_per_frame"] == 520093696 and c4["frame_ms"] > 0
    assert line["roofline"]["frac"] is None          # a per-launch fraction is claimed for the one-GPU workload only


def test_bench_plain_command_starts_its_own_ranks(ca):
    """`python3 bench.py --gpus 2 ...` with NO launcher in the command and no RANK / WORLD_SIZE in the environment — the form
    the driver may use — must start the two ranks itself (a child `torch.distributed.run`, before any GPU call), relay rank
    0's ONE JSON line and the child's exit code.  Rehearsed on the one GPU (both ranks on device 0, gloo for RCCL)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CUTRACE_BENCH_SHARE_GPU="1", CUTRACE_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--check",
           "--no-extras", "--skip-probe"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "gathered frame(s) bitwise equal" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_step"] == 2
    d = line["config"]["dist"]
    assert d["world_size"] == 2 and d["backend"] == "gloo" and d["devices"] == [0, 0] and d["shared_gpu_rehearsal"]
    assert "bench.py itself" in d["launched_by"]
    # and a failing child is not swallowed: an unknown scene makes every rank exit non-zero
    bad = subprocess.run(cmd + ["--scene", "scene/no_such_scene.json"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_one_device_mixes_async_and_synchronous_frames(ca):
    """with ONE device and CTR_VAR_NO_DIRECT a page-locked destination takes the asynchronous path
    (render stream + transfer stream) while a pageable one takes the synchronous ctr_render shortcut on the null stream.
    Alternating the two with two frames in flight used to let two launches of ONE scene handle overlap (shared counter
    shards, cost table, dispatch order).  The shortcut now drains the streams first: every frame and every ray count must
    equal the synchronous render, whatever the order."""
    s = load_scene(ca, "bunny", 640, 360)
    m = ca.MultiScene(s, [0])
    bs = (5, 2, 4, 3, 5, 1, 4, 2)
    want = [m.render(bounces=b) for b in bs]
    want = [{k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in r.items()} for r in want]
    m.set_variant(ca.VAR_NO_DIRECT)
    pin = [m.alloc_frame() for _ in range(2)]
    page = [dict(depth=np.empty((360, 640), np.float32), color=np.empty((360, 640, 3), np.float32),
                 normal=np.empty((360, 640, 3), np.float32)) for _ in range(2)]
    dest = [pin[0], page[0], pin[1], page[1], page[0], pin[0], pin[1], page[1]]   # async, sync, async, sync, sync, async ...
    m.submit(dest[0], bounces=bs[0])
    for k in range(1, len(bs) + 1):
        if k < len(bs):
            m.submit(dest[k], bounces=bs[k])
        st = m.wait()
        got = dest[k - 1]
        for key in ("depth", "color", "normal"):
            assert same_bits(got[key], want[k - 1][key]), (k - 1, key)
        assert st["ray_count"] == want[k - 1]["ray_count"] and st["max_depth"] == want[k - 1]["max_depth"], k - 1
        got["depth"][:] = 0.0
    for f in pin:
        m.free_frame(f)
    m.close()


def test_pipelined_submit_wait_over_the_rccl_branch(ca):
    """The RCCL branch of ctr_multi_submit (per-device transfer stream, ev_moved recorded after the grouped ncclSend /
    ncclRecv) with two frames in flight — on the one GPU as a one-rank self send/recv (CUTRACE_MULTI_TRANSPORT=rccl-self)."""
    s = load_scene(ca, "bunny", 320, 180)
    os.environ["CUTRACE_MULTI_TRANSPORT"] = "rccl-self"
    try:
        m = ca.MultiScene(s, [0])
    finally:
        del os.environ["CUTRACE_MULTI_TRANSPORT"]
    if m.transport != "rccl-self":
        m.close()
        pytest.skip("librccl.so could not be loaded on this box")
    bs = (2, 4, 3, 5, 1, 4)
    ref = ca.DeviceScene(s)
    want = [ref.render(bounces=b) for b in bs]
    for pinned in (True, False):
        if pinned:
            frames = [m.alloc_frame() for _ in range(3)]
        else:
            frames = [dict(depth=np.empty((180, 320), np.float32), color=np.empty((180, 320, 3), np.float32),
                           normal=np.empty((180, 320, 3), np.float32)) for _ in range(3)]
        m.submit(frames[0], bounces=bs[0])
        for k in range(1, len(bs) + 1):
            if k < len(bs):
                m.submit(frames[k % 3], bounces=bs[k])
            st = m.wait()
            got = frames[(k - 1) % 3]
            for key in ("depth", "color", "normal"):
                assert same_bits(got[key], want[k - 1][key]), (pinned, k - 1, key)
            assert st["ray_count"] == want[k - 1]["ray_count"]
            got["depth"][:] = 0.0
        if pinned:
            for f in frames:
                m.free_frame(f)
    m.close()


def test_bench_two_steps_in_flight_two_ranks(ca):
    """bench.py --in-flight 2 (consecutive steps alternate between two scene handles on two streams, so the next step's first
    waves fill the tail of the previous one) with two ranks on the one GPU (gloo): four buffer slots per rank, exchanges of
    two steps outstanding — every gathered frame still bitwise the single-process render."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CUTRACE_BENCH_SHARE_GPU="1", CUTRACE_BENCH_BACKEND="gloo")
    for extra in (["--width", "640", "--height", "360"],
                  ["--workload", "c4", "--scaling", "strong", "--roots", "rank0", "--width", "512", "--height", "512"]):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--in-flight", "2",
               "--check", "--no-extras", "--skip-probe"] + extra
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "gathered frame(s) bitwise equal" in r.stderr
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["config"]["steps_in_flight"] == 2
