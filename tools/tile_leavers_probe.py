"""How many particles change their 8^3 tile per step? (GPU box) VERDICT r05 item 4: tile-resident particle storage pays only if most
particles stay. Downloads the positions before and after single steps of the moving dam and counts, by particle id, the ones whose
tile (and whose cell) differs.
usage: python tools/tile_leavers_probe.py C4 20 25 300"""
import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes

name = sys.argv[1]
marks = sorted(int(a) for a in sys.argv[2:])
cfg = scenes.CONFIGS[name]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])


def tiles_by_id():
    p = s.download_particles(write_positions=True)
    ids = s.particle_ids()
    c = np.floor(p["pos"]).astype(np.int32)
    np.clip(c, 0, np.asarray(cfg["size"]) - 1, out=c)
    cell = np.empty((len(ids), 3), dtype=np.int32)
    cell[ids] = c
    return cell


k = 0
for m in marks:
    while k < m:
        s.time_step(min(3.0 * s.cfl(), 0.033)); k += 1
    a = tiles_by_id()
    dt = min(3.0 * s.cfl(), 0.033)
    s.time_step(dt); k += 1
    b = tiles_by_id()
    tile_changed = np.any((a >> 3) != (b >> 3), axis=1)
    cell_changed = np.any(a != b, axis=1)
    d = np.abs(b - a).max(axis=1)
    print(f"{name} step {m}: dt {dt:.5f}; particles that changed tile {tile_changed.mean():.3f}, changed cell {cell_changed.mean():.3f}; "
          f"cells moved (max-norm) mean {d.mean():.2f} p99 {np.percentile(d, 99):.0f} max {d.max()}", flush=True)
