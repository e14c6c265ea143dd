"""C4 strong-scaling tail: per-tile costs of the whole frame (ctr_tile_costs), their distribution, and a
list-scheduling simulation of what each of the 8 parts (interleaved 8-row blocks) can reach on 6144 wave slots —
whole tiles against the dearest p % of the tiles split into two 8x4 halves (each half assumed to cost `half_factor` of the tile)."""
import sys, os, json, tempfile, heapq
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
s = ca.HostScene.load(scenes.make_bunny_grid(d))
ds = ca.DeviceScene(s)
for _ in range(4):
    r = ds.render(bounces=5)
cost = ds.tile_costs().astype(np.float64)   # shader-clock ticks / 64 per tile, row-major tile grid (512 x 512)
ms = r["kernel_ms"]
n = cost.size
tx = 4096 // 8
grid = cost.reshape(-1, tx)
print(json.dumps({"kernel_ms": ms, "tiles": n, "sum": cost.sum(), "mean": cost.mean(), "max_over_mean": cost.max() / cost.mean(),
                  "p50/mean": np.percentile(cost, 50) / cost.mean(), "p90/mean": np.percentile(cost, 90) / cost.mean(),
                  "p99/mean": np.percentile(cost, 99) / cost.mean(), "p99.9/mean": np.percentile(cost, 99.9) / cost.mean()}))
SLOTS = 1024 * 6
# wave-slot time: a tile's cost was measured with ~6 waves sharing its SIMD; ticks_to_ms from the whole frame
ticks_to_ms = ms / (cost.sum() / SLOTS)   # if the slots were perfectly packed the frame would take sum/SLOTS ticks

def lpt(costs, slots=SLOTS):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in sorted(costs, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)

print("whole frame: LPT makespan %.3f ms, perfect packing %.3f ms, dearest tile %.3f ms" % (lpt(cost) * ticks_to_ms, cost.sum() / SLOTS * ticks_to_ms, cost.max() * ticks_to_ms))
for parts in (2, 4, 8):
    for p_split, half in ((0.0, 1.0), (0.02, 0.6), (0.05, 0.6), (0.10, 0.6), (0.05, 0.7), (0.2, 0.6)):
        worst = 0.0
        ideal = 0.0
        for part in range(parts):
            c = grid[part::parts].ravel()   # tile rows of this part (8-row blocks = tile rows)
            if p_split > 0:
                thr = np.percentile(c, 100 * (1 - p_split))
                big = c[c >= thr]
                c = np.concatenate([c[c < thr], big * half, big * half])
            worst = max(worst, lpt(c))
            ideal = max(ideal, c.sum() / SLOTS)
        print("%d parts, dearest %4.1f %% of tiles as two halves (each %.0f %% of the tile): slowest part %.3f ms (perfect packing %.3f) -> efficiency bound %.3f"
              % (parts, 100 * p_split, 100 * half, worst * ticks_to_ms, ideal * ticks_to_ms, ms / (parts * worst * ticks_to_ms)), flush=True)
