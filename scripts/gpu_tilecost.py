"""Measured per-tile costs of a frame (ctr_tile_costs) saved for offline study of the dispatch order."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
for name in ("bunny", "mirror", "sphere_plane"):
    s = ca.HostScene.load(f"scene/{name}.json")
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_NO_PRIOR)
    ds.render()
    c = ds.tile_costs()
    np.save(f"gpurun_out/tilecost_{name}.npy", c)
    w, h = s.size
    tx, ty = (w + 7) // 8, (h + 7) // 8
    g = c.reshape(ty, tx).astype(np.float64)
    print(name, "tiles", c.size, "cost min/mean/max", c.min(), round(c.mean(), 1), c.max(), flush=True)
    # coarse picture: mean cost of 15x15-tile cells, in units of the global mean
    cell = 15
    for y in range(0, ty, cell):
        print(" ".join(f"{g[y:y + cell, x:x + cell].mean() / g.mean():4.1f}" for x in range(0, tx, cell)))
