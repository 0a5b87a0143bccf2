"""Distribution of the particles per cell at the P2G of a running scene, and what it means for a lane-per-cell walk:
lane efficiency of a wave that owns the 64 cells of one z-slice of a tile = particles / (64 x the slice's fullest cell)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes
cfg = scenes.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
steps = [int(a) for a in sys.argv[2:]] or [1, 30, 60]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
nx, ny, nz = cfg["size"]
done = 0
for target in steps:
    while done < target:
        s.time_step(min(3.0 * s.cfl(), 0.033))
        done += 1
    c = s.cell_counts().reshape(nz, ny, nx)
    occ = c[c > 0]
    hist = np.bincount(np.minimum(occ, 40))
    t = c.reshape(nz // 8, 8, ny // 8, 8, nx // 8, 8).transpose(0, 2, 4, 1, 3, 5)  # [tz, ty, tx, lz, ly, lx]
    sl = t.reshape(-1, 8, 64)  # [tile, slice, cell]
    mx = sl.max(axis=2).astype(np.int64)
    sm = sl.sum(axis=2).astype(np.int64)
    eff = sm.sum() / max(1, (64 * mx).sum())
    tile_sum = sm.sum(axis=1)
    for K in (2, 4, 8):
        lanes = ((sl + K - 1) // K).sum()
        print(f"  chunks of {K}: lane-iterations {lanes * K} for {sm.sum()} particles = {sm.sum() / (lanes * K):.3f}")
    print(f"step {target}: particles {occ.sum()}, occupied cells {occ.size}, mean {occ.mean():.2f}, max {occ.max()}, "
          f"particle tiles {(tile_sum > 0).sum()}, slice efficiency {eff:.3f}, mean of slice max {mx[mx > 0].mean():.2f}")
    print("  histogram of particles per occupied cell (last bin = 40+):", hist.tolist())
