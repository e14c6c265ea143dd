"""Row tiling of a frame over the GPUs of one node + the single gather of the final frame.

North-star shape (BASELINE.json): the image is row-tiled across the ranks, every rank renders
its rows with the full (replicated, < 1 MB) scene, and ONE gather over RCCL/xGMI brings the
three buffers to rank 0, which re-interleaves them into the grid<> layout.

Rows are dealt in interleaved blocks (block b -> rank b % world) because contiguous bands are
badly balanced (rows crossing the mesh cost far more than wall-only rows).

The module is renderer-agnostic: `render_rows(frame_index, rows_tuple, out_views)` fills this
rank's compact buffers.  bench.py passes the HIP path (ctr_render_device); the gloo CPU tests
pass the oracle, which is the only place the oracle may be used.
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

    def __init__(self, w, h, frames, rank, world, device, block_rows=BLOCK_ROWS):
        self.w, self.h, self.frames, self.rank, self.world = w, h, frames, rank, world
        self.device = device
        self.block_rows = block_rows
        self.rows = (0, h, block_rows, rank, world) if world > 1 else None
        self.my_rows = part_rows(h, rank, world, block_rows) if world > 1 else list(range(h))
        self.cap = max_part_rows(h, world, block_rows) if world > 1 else h  # padded rows per rank
        f32 = torch.float32
        per = frames * self.cap * w
        # packed local buffer: [depth | color | normal], 7 floats per pixel
        self.local = torch.zeros(7 * per, dtype=f32, device=device)
        self.sec = (0, per, 4 * per, 7 * per)
        self.gathered = None
        self.final = None
        if rank == 0:
            if world > 1:
                self.gathered = torch.zeros(world, 7 * per, dtype=f32, device=device)
                # position of global row y inside the (world*cap) padded row axis
                perm = torch.empty(h, dtype=torch.int64)
                for p in range(world):
                    for k, y in enumerate(part_rows(h, p, world, block_rows)):
                        perm[y] = p * self.cap + k
                self.perm = perm.to(device)
            self.final = dict(depth=torch.zeros(frames, h, w, dtype=f32, device=device),
                              color=torch.zeros(frames, h, w, 3, dtype=f32, device=device),
                              normal=torch.zeros(frames, h, w, 3, dtype=f32, device=device))

    def views(self, frame):
        """Compact (rows_local x w) views of this rank's buffers for one frame."""
        n, w, cap = len(self.my_rows), self.w, self.cap
        d0, c0, n0, _ = self.sec
        d = self.local[d0 + frame * cap * w: d0 + frame * cap * w + n * w]
        c = self.local[c0 + 3 * frame * cap * w: c0 + 3 * frame * cap * w + 3 * n * w]
        m = self.local[n0 + 3 * frame * cap * w: n0 + 3 * frame * cap * w + 3 * n * w]
        return d, c, m

    def gather(self):
        """The one collective of the path: packed local buffers -> rank 0, then re-interleave
        into the final row-major frames (device side)."""
        if self.world == 1:
            F, h, w = self.frames, self.h, self.w
            d0, c0, n0, e = self.sec
            self.final["depth"] = self.local[d0:c0].view(F, h, w)
            self.final["color"] = self.local[c0:n0].view(F, h, w, 3)
            self.final["normal"] = self.local[n0:e].view(F, h, w, 3)
            return
        if self.rank == 0:
            dist.gather(self.local, [self.gathered[i] for i in range(self.world)], dst=0)
            self.assemble()
        else:
            dist.gather(self.local, None, dst=0)

    def assemble(self):
        W, F, cap, w = self.world, self.frames, self.cap, self.w
        d0, c0, n0, e = self.sec
        g = self.gathered
        dep = g[:, d0:c0].view(W, F, cap, w).permute(1, 0, 2, 3).reshape(F, W * cap, w)
        col = g[:, c0:n0].view(W, F, cap, w, 3).permute(1, 0, 2, 3, 4).reshape(F, W * cap, w, 3)
        nor = g[:, n0:e].view(W, F, cap, w, 3).permute(1, 0, 2, 3, 4).reshape(F, W * cap, w, 3)
        torch.index_select(dep, 1, self.perm, out=self.final["depth"])
        torch.index_select(col, 1, self.perm, out=self.final["color"])
        torch.index_select(nor, 1, self.perm, out=self.final["normal"])
