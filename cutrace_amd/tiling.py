"""Row tiling of a frame over the GPUs of one node + the single gather of the final frame.

North-star shape (BASELINE.json): the image is row-tiled across the ranks, every rank renders
its rows with the full (replicated, < 1 MB) scene, and ONE gather over RCCL/xGMI brings the
three buffers to rank 0, which re-interleaves them into the grid<> layout.

Rows are dealt in interleaved blocks (block b -> rank b % world) because contiguous bands are
badly balanced (rows crossing the mesh cost far more than wall-only rows).

Pipelining: the gather of step k runs on the communication stream (RCCL's own) and the
re-interleave on a side stream while step k+1 is being rendered into the other half of a
double buffer — xGMI is point-to-point (7 links per GPU), so the 7 incoming messages of rank 0
ride 7 different links, but they are still ~1 ms for 58 MB each and must not serialise with
rendering.  `finish()` drains everything (called inside the timed region by bench.py).

The module is renderer-agnostic: the caller fills `views(slot, frame)` with this rank's compact
buffers.  bench.py uses the HIP path (ctr_render_device); the gloo CPU tests use the oracle,
which is the only place the oracle may be used.
"""
import torch
import torch.distributed as dist

BLOCK_ROWS = 8  # one wave tile high: a tile never straddles two row blocks


def part_rows(h, part, n_parts, block_rows=BLOCK_ROWS):
    """Global row indices rendered by `part` (increasing)."""
    return [y for y in range(h) if (y // block_rows) % n_parts == part]


def max_part_rows(h, n_parts, block_rows=BLOCK_ROWS):
    return max(len(part_rows(h, p, n_parts, block_rows)) for p in range(n_parts))


class FrameTiler:
    """Buffers and index tables for rendering `frames` frames of w x h per step on `world` ranks."""

    def __init__(self, w, h, frames, rank, world, device, block_rows=BLOCK_ROWS, slots=2, rotate=True):
        self.w, self.h, self.frames, self.rank, self.world = w, h, frames, rank, world
        self.device = torch.device(device)
        self.block_rows = block_rows
        self.rows = (0, h, block_rows, rank, world) if world > 1 else None
        # frame f of a step is rendered by this rank for part (rank + f*part_stride) % world: over a
        # batch every rank sees every row block, so ranks with costly blocks do not lag
        self.part_stride = 1 if (rotate and world > 1 and frames > 1) else 0
        self.my_rows = part_rows(h, rank, world, block_rows) if world > 1 else list(range(h))
        self.cap = max_part_rows(h, world, block_rows) if world > 1 else h  # padded rows per rank
        self.slots = slots if world > 1 else 1
        f32 = torch.float32
        per = frames * self.cap * w
        self.per = per
        # packed local buffers (one per pipeline slot): [depth | color | normal], 7 floats per pixel
        self.local = [torch.zeros(7 * per, dtype=f32, device=self.device) for _ in range(self.slots)]
        self.sec = (0, per, 4 * per, 7 * per)
        self.gathered = None
        self.final = None
        self.pending = [None] * self.slots      # outstanding gather per slot
        self.side = None                         # side stream for the re-interleave on rank 0
        self.assembled = [None] * self.slots     # event: slot's frames are in `final`
        self.is_cuda = self.device.type == "cuda"
        if rank == 0:
            if world > 1:
                self.gathered = [torch.zeros(world, 7 * per, dtype=f32, device=self.device) for _ in range(self.slots)]
                # position of global row y inside the (world*cap) padded row axis
                perm = torch.empty(frames, h, dtype=torch.int64)
                for f in range(frames):
                    for p in range(world):
                        r = (p - f * self.part_stride) % world  # the rank that rendered part p of frame f
                        for k, y in enumerate(part_rows(h, p, world, block_rows)):
                            perm[f, y] = (f * world + r) * self.cap + k
                self.perm = perm.reshape(-1).to(self.device)
                if self.is_cuda:
                    self.side = torch.cuda.Stream(device=self.device)
            self.final = dict(depth=torch.zeros(frames, h, w, dtype=f32, device=self.device),
                              color=torch.zeros(frames, h, w, 3, dtype=f32, device=self.device),
                              normal=torch.zeros(frames, h, w, 3, dtype=f32, device=self.device))

    def frame_part(self, frame):
        return (self.rank + frame * self.part_stride) % self.world

    def frame_rows(self, frame):
        """ctr_rows tuple for what this rank renders of `frame`."""
        if self.world == 1:
            return None
        return (0, self.h, self.block_rows, self.frame_part(frame), self.world)

    def views(self, slot, frame):
        """Compact (rows_local x w) views of this rank's buffers for one frame of one slot."""
        n = len(part_rows(self.h, self.frame_part(frame), self.world, self.block_rows)) if self.world > 1 else self.h
        w, cap = self.w, self.cap
        d0, c0, n0, _ = self.sec
        buf = self.local[slot]
        d = buf[d0 + frame * cap * w: d0 + frame * cap * w + n * w]
        c = buf[c0 + 3 * frame * cap * w: c0 + 3 * frame * cap * w + 3 * n * w]
        m = buf[n0 + 3 * frame * cap * w: n0 + 3 * frame * cap * w + 3 * n * w]
        return d, c, m

    def begin(self, slot):
        """Call before rendering into `slot`: its previous gather (two steps ago) must be done."""
        if self.world == 1:
            return
        self._wait(slot)

    def _wait(self, slot):
        work = self.pending[slot]
        if work is not None:
            work.wait()           # current stream waits for the collective
            self.pending[slot] = None
        ev = self.assembled[slot]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self.assembled[slot] = None

    def gather(self, slot):
        """The one collective of the path, asynchronous: packed local buffers of `slot` -> rank 0;
        rank 0 then re-interleaves them into the final row-major frames on a side stream."""
        if self.world == 1:
            F, h, w = self.frames, self.h, self.w
            d0, c0, n0, e = self.sec
            buf = self.local[0]
            self.final["depth"] = buf[d0:c0].view(F, h, w)
            self.final["color"] = buf[c0:n0].view(F, h, w, 3)
            self.final["normal"] = buf[n0:e].view(F, h, w, 3)
            return
        if self.rank == 0:
            work = dist.gather(self.local[slot], [self.gathered[slot][i] for i in range(self.world)], dst=0,
                               async_op=True)
            if self.is_cuda:
                with torch.cuda.stream(self.side):
                    work.wait()   # side stream waits for the collective, the render stream does not
                    self.assemble(slot)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                self.assembled[slot] = ev
                self.pending[slot] = None
            else:
                work.wait()
                self.assemble(slot)
        else:
            self.pending[slot] = dist.gather(self.local[slot], None, dst=0, async_op=True)

    def finish(self):
        """Drain every outstanding gather / re-interleave."""
        for slot in range(self.slots):
            self._wait(slot)
        if self.is_cuda:
            torch.cuda.synchronize(self.device)

    def assemble(self, slot):
        if self.is_cuda:
            return self._assemble_hip(slot)
        W, F, cap, w = self.world, self.frames, self.cap, self.w
        d0, c0, n0, e = self.sec
        g = self.gathered[slot]
        h = self.h
        dep = g[:, d0:c0].view(W, F, cap, w).permute(1, 0, 2, 3).reshape(F * W * cap, w)
        col = g[:, c0:n0].view(W, F, cap, w, 3).permute(1, 0, 2, 3, 4).reshape(F * W * cap, w, 3)
        nor = g[:, n0:e].view(W, F, cap, w, 3).permute(1, 0, 2, 3, 4).reshape(F * W * cap, w, 3)
        torch.index_select(dep, 0, self.perm, out=self.final["depth"].view(F * h, w))
        torch.index_select(col, 0, self.perm, out=self.final["color"].view(F * h, w, 3))
        torch.index_select(nor, 0, self.perm, out=self.final["normal"].view(F * h, w, 3))

    def _assemble_hip(self, slot):
        """Rank 0, on the GPU: one launch of the library's re-interleave kernel per frame (ctr_reinterleave_device)
        instead of three torch.index_select passes — each row is read and written once, 16 bytes per lane."""
        from . import _lib
        L = _lib.hip_lib()
        W, F, cap, w, h = self.world, self.frames, self.cap, self.w, self.h
        d0, c0, n0, _ = self.sec
        g = self.gathered[slot]
        base, esz, stride = g.data_ptr(), g.element_size(), g.stride(0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for f in range(F):
            parts = (_lib.ReintPart * W)()
            for p in range(W):
                r = (p - f * self.part_stride) % W  # the rank that rendered part p of frame f
                row0 = base + r * stride * esz
                parts[p].d_depth = row0 + (d0 + f * cap * w) * esz
                parts[p].d_color3 = row0 + (c0 + 3 * f * cap * w) * esz
                parts[p].d_normal3 = row0 + (n0 + 3 * f * cap * w) * esz
            st = L.ctr_reinterleave_device(parts, W, self.block_rows, w, h, self.final["depth"][f].data_ptr(),
                                           self.final["color"][f].data_ptr(), self.final["normal"][f].data_ptr(), stream)
            if st:
                raise RuntimeError("ctr_reinterleave_device failed")
