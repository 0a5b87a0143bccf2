"""How sparse are the particle tiles late in a run? (GPU box) Runs C3 to step N, downloads the fluid cells and counts, per 8^3 tile,
the (y, z) rows of 8 x-cells, the z-slices and the 128-byte lines (4 y-rows of a z-slice) that hold at least one unknown - what a
PCG kernel that skips empty rows / lines of a tile would still have to move."""
import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg_name, n_steps = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("C3", 550)
cfg = scenes.CONFIGS[cfg_name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
for k in range(n_steps):
    s.time_step(min(3.0 * s.cfl(), 0.033))
    if k in (40, n_steps - 1):
        s.hash()
        fc = s.fluid_cells().astype(np.int64)
        nx, ny, nz = cfg["size"]
        x, y, z = fc % nx, (fc // nx) % ny, fc // (nx * ny)
        tile = (x >> 3) + (nx >> 3) * ((y >> 3) + (ny >> 3) * (z >> 3))
        tiles = np.unique(tile)
        rows = np.unique(tile * 64 + (y & 7) + 8 * (z & 7))
        lines = np.unique(tile * 16 + ((y & 7) >> 2) + 2 * (z & 7))
        slices = np.unique(tile * 8 + (z & 7))
        print(f"{cfg_name} step {k}: {len(fc)} unknowns in {len(tiles)} tiles ({len(fc) / len(tiles) / 512:.2f} full); rows with an unknown "
              f"{len(rows) / (64 * len(tiles)):.2f}, 128-byte lines {len(lines) / (16 * len(tiles)):.2f}, z-slices {len(slices) / (8 * len(tiles)):.2f} of all")
s.close()
