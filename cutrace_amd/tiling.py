"""Row tiling of a frame over the GPUs of one node + the single gather of each final frame.

North-star shape (BASELINE.json): the image is row-tiled across the ranks, every rank renders
its rows with the full (replicated, < 1 MB) scene, and ONE gather over RCCL/xGMI brings the
three buffers of a frame to its root rank, which re-interleaves them into the grid<> layout.

Rows are dealt in interleaved blocks (block b -> rank b % world) because contiguous bands are
badly balanced (rows crossing the mesh cost far more than wall-only rows).

Which rank is the root.  A single frame is gathered to rank 0.  A step of SEVERAL frames (weak scaling: `world`
frames per step, a camera path) gathers frame f to rank f % world (`roots="rotate"`): xGMI is point-to-point,
7 links per GPU, so with every frame rooted at rank 0 each of rank 0's inbound links would carry a whole
rank's output (58 MB per 1080p frame-equivalent and step) and rank 0 alone would re-interleave all `world`
frames; rotated, every link carries 1/world of that in each direction and every rank re-interleaves one frame.
`roots="rank0"` keeps everything on rank 0.  Either way a frame needs ONE grouped send/recv exchange
(`batch_isend_irecv` = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on RCCL; the same calls on gloo).

Pipelining: the exchange of step k runs on the communication stream (RCCL's own) and the
re-interleave on a side stream while step k+1 is being rendered into the other half of a
double buffer.  `finish()` drains everything (called inside the timed region by bench.py).

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

    def __init__(self, w, h, frames, rank, world, device, block_rows=BLOCK_ROWS, slots=2, rotate=True, roots="rotate"):
        self.w, self.h, self.frames, self.rank, self.world = w, h, frames, rank, world
        self.device = torch.device(device)
        self.block_rows = block_rows
        self.rows = (0, h, block_rows, rank, world) if world > 1 else None
        # frame f of a step is rendered by this rank for part (rank + f*part_stride) % world: over a
        # batch every rank sees every row block, so ranks with costly blocks do not lag
        self.part_stride = 1 if (rotate and world > 1 and frames > 1) else 0
        self.my_rows = part_rows(h, rank, world, block_rows) if world > 1 else list(range(h))
        self.cap = max_part_rows(h, world, block_rows) if world > 1 else h  # padded rows per rank
        self.slots = slots if (world > 1 or slots > 2) else 1   # (slots > 2: asked for explicitly — frames in flight on one GPU)
        self.rotate_roots = roots == "rotate" and world > 1 and frames > 1
        self.final_frames = [f for f in range(frames) if self.root_of(f) == rank]  # frames assembled on this rank
        f32 = torch.float32
        P = self.cap * w                          # padded pixels of one rank's piece of one frame
        self.P = P
        per = frames * P
        self.per = per
        # packed local buffers (one per pipeline slot): [depth | color | normal], 7 floats per pixel
        self.local = [torch.zeros(7 * per, dtype=f32, device=self.device) for _ in range(self.slots)]
        self.sec = (0, per, 4 * per, 7 * per)
        self.gathered = None
        self.final = None
        self.pending = [None] * self.slots      # outstanding exchange per slot (list of works)
        self._ops = [None] * self.slots         # the exchange's send / receive descriptors per slot
        self._parts = {}                        # (slot, my frame) -> pointer table of the re-interleave launch
        self.side = None                         # side stream for the re-interleave
        self.assembled = [None] * self.slots     # event: slot's frames are in `final`
        self.is_cuda = self.device.type == "cuda"
        nf = len(self.final_frames)
        if nf:
            if world > 1:
                # pieces received for my frames: [my frame][source rank][depth P | color 3P | normal 3P]
                self.gathered = [torch.zeros(nf, world, 7 * P, dtype=f32, device=self.device) for _ in range(self.slots)]
                if self.is_cuda:
                    self.side = torch.cuda.Stream(device=self.device)
                else:
                    # position of global row y of my i-th frame inside the (world*cap) padded row axis
                    perm = torch.empty(nf, h, dtype=torch.int64)
                    for i, f in enumerate(self.final_frames):
                        for p in range(world):
                            r = (p - f * self.part_stride) % world  # the rank that rendered part p of frame f
                            for k, y in enumerate(part_rows(h, p, world, block_rows)):
                                perm[i, y] = r * self.cap + k
                    self.perm = perm
            self.final = dict(depth=torch.zeros(nf, h, w, dtype=f32, device=self.device),
                              color=torch.zeros(nf, h, w, 3, dtype=f32, device=self.device),
                              normal=torch.zeros(nf, h, w, 3, dtype=f32, device=self.device))

    def root_of(self, frame):
        """The rank frame `frame` of a step is gathered to."""
        return frame % self.world if self.rotate_roots else 0

    def frame_part(self, frame):
        return (self.rank + frame * self.part_stride) % self.world

    def frame_rows(self, frame):
        """ctr_rows tuple for what this rank renders of `frame`."""
        if self.world == 1:
            return None
        return (0, self.h, self.block_rows, self.frame_part(frame), self.world)

    def pieces(self, slot, frame):
        """This rank's padded (cap x w) pieces of one frame of one slot: depth (P), color (3P), normal (3P) floats."""
        P = self.P
        d0, c0, n0, _ = self.sec
        buf = self.local[slot]
        return (buf[d0 + frame * P: d0 + (frame + 1) * P], buf[c0 + 3 * frame * P: c0 + 3 * (frame + 1) * P],
                buf[n0 + 3 * frame * P: n0 + 3 * (frame + 1) * P])

    def views(self, slot, frame):
        """Compact (rows_local x w) views of this rank's buffers for one frame of one slot."""
        n = len(part_rows(self.h, self.frame_part(frame), self.world, self.block_rows)) if self.world > 1 else self.h
        d, c, m = self.pieces(slot, frame)
        return d[:n * self.w], c[:3 * n * self.w], m[:3 * n * self.w]

    def begin(self, slot):
        """Call before rendering into `slot`: its previous exchange (two steps ago) must be done."""
        if self.world == 1:
            return
        self._wait(slot)

    def _wait(self, slot):
        works = self.pending[slot]
        if works:
            for wk in works:
                wk.wait()         # current stream waits for the exchange
            self.pending[slot] = None
        ev = self.assembled[slot]
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
            self.assembled[slot] = None

    def gather(self, slot):
        """The one exchange of the path, asynchronous: every frame's pieces of `slot` go to the frame's root rank,
        which then re-interleaves them into the final row-major frame on a side stream."""
        if self.world == 1:
            F, h, w = self.frames, self.h, self.w
            d0, c0, n0, e = self.sec
            buf = self.local[slot]
            self.final["depth"] = buf[d0:c0].view(F, h, w)
            self.final["color"] = buf[c0:n0].view(F, h, w, 3)
            self.final["normal"] = buf[n0:e].view(F, h, w, 3)
            return
        ops = self._exchange_ops(slot)
        works = dist.batch_isend_irecv(ops) if ops else []
        if not self.final_frames:
            self.pending[slot] = works
            return
        if self.is_cuda:
            self.side.wait_stream(torch.cuda.current_stream(self.device))  # my own pieces: rendered on the current stream
            with torch.cuda.stream(self.side):
                for wk in works:
                    wk.wait()     # side stream waits for the exchange, the render stream does not
                self.assemble(slot)
                ev = torch.cuda.Event()
                ev.record(self.side)
            self.assembled[slot] = ev
            self.pending[slot] = None
        else:
            for wk in works:
                wk.wait()
            self.assemble(slot)

    def _exchange_ops(self, slot):
        """The sends and receives of one step for `slot` (its buffers never move: built once, reused every step).
        Every rank walks the frames in the same order, so the sends and receives of any pair of ranks match up."""
        if self._ops[slot] is None:
            P = self.P
            ops = []
            for f in range(self.frames):
                root = self.root_of(f)
                if root == self.rank:
                    g = self.gathered[slot][self.final_frames.index(f)]
                    for q in range(self.world):
                        if q != self.rank:
                            ops += [dist.P2POp(dist.irecv, g[q][0:P], q), dist.P2POp(dist.irecv, g[q][P:4 * P], q),
                                    dist.P2POp(dist.irecv, g[q][4 * P:7 * P], q)]
                else:
                    ops += [dist.P2POp(dist.isend, t, root) for t in self.pieces(slot, f)]
            self._ops[slot] = ops
        return self._ops[slot]

    def finish(self):
        """Drain every outstanding exchange / re-interleave."""
        for slot in range(self.slots):
            self._wait(slot)
        if self.is_cuda:
            torch.cuda.synchronize(self.device)

    def _sources(self, slot, i, f):
        """(depth, color, normal) pieces of my i-th frame (global frame f) by SOURCE RANK: my own from the local
        buffer, the others' from what was received."""
        P = self.P
        out = []
        for r in range(self.world):
            if r == self.rank:
                out.append(self.pieces(slot, f))
            else:
                g = self.gathered[slot][i][r]
                out.append((g[0:P], g[P:4 * P], g[4 * P:7 * P]))
        return out

    def assemble(self, slot):
        if self.is_cuda:
            return self._assemble_hip(slot)
        W, cap, w, h = self.world, self.cap, self.w, self.h
        for i, f in enumerate(self.final_frames):
            src = self._sources(slot, i, f)
            dep = torch.cat([s[0] for s in src]).view(W * cap, w)
            col = torch.cat([s[1] for s in src]).view(W * cap, w, 3)
            nor = torch.cat([s[2] for s in src]).view(W * cap, w, 3)
            torch.index_select(dep, 0, self.perm[i], out=self.final["depth"][i])
            torch.index_select(col, 0, self.perm[i], out=self.final["color"][i])
            torch.index_select(nor, 0, self.perm[i], out=self.final["normal"][i])

    def _assemble_hip(self, slot):
        """On the GPU: one launch of the library's re-interleave kernel per frame (ctr_reinterleave_device)
        instead of three torch.index_select passes — each row is read and written once, 16 bytes per lane."""
        from . import _lib
        L = _lib.hip_lib()
        W = self.world
        stream = torch.cuda.current_stream(self.device).cuda_stream
        for i, f in enumerate(self.final_frames):
            parts = self._parts.get((slot, i))
            if parts is None:  # (the buffers never move: the pointer table is built once)
                src = self._sources(slot, i, f)
                parts = (_lib.ReintPart * W)()
                for p in range(W):
                    r = (p - f * self.part_stride) % W  # the rank that rendered part p of frame f
                    parts[p].d_depth, parts[p].d_color3, parts[p].d_normal3 = (t.data_ptr() for t in src[r])
                self._parts[(slot, i)] = parts
            st = L.ctr_reinterleave_device(parts, W, self.block_rows, self.w, self.h, self.final["depth"][i].data_ptr(),
                                           self.final["color"][i].data_ptr(), self.final["normal"][i].data_ptr(), stream)
            if st:
                raise RuntimeError("ctr_reinterleave_device failed")
