"""Late in a run: how many particle tiles hold only unknowns that couple to nothing outside the tile (spray droplets)? (GPU box)
Such a tile's block of the pressure matrix is decoupled from the rest: it could be solved on its own, once, and leave the PCG's tile
list. Counts, per step given, closed tiles, their unknowns, and the singleton unknowns (no fluid neighbour at all).
usage: python tools/closed_tile_probe.py C3 550 [more steps]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import libfluid_amd as lfa
from libfluid_amd import scenes



def analyse(f):
    """f[z, y, x] bool -> (open_tile[tz, ty, tx]: a fluid-fluid coupling crosses one of the tile's faces; nbrs: fluid neighbours per cell)"""
    nz, ny, nx = f.shape
    open_tile = np.zeros((nz >> 3, ny >> 3, nx >> 3), dtype=bool)
    nbrs = np.zeros(f.shape, dtype=np.uint8)
    tiles = lambda a, ax: a  # noqa: E731
    for ax in range(3):
        lo = [slice(None)] * 3; hi = [slice(None)] * 3
        lo[ax] = slice(0, -1); hi[ax] = slice(1, None)
        both = f[tuple(lo)] & f[tuple(hi)]
        nbrs[tuple(lo)] += both; nbrs[tuple(hi)] += both
        # pairs (i, i + 1) with i % 8 == 7 straddle a tile face
        sel = [slice(None)] * 3
        sel[ax] = slice(7, None, 8)
        cross = both[tuple(sel)]          # along ax: one entry per interior tile face
        shp = list(cross.shape)
        other = [a for a in range(3) if a != ax]
        # reduce the two in-face axes to tiles
        r = cross
        for a in other:
            sh = list(r.shape)
            sh[a:a + 1] = [sh[a] >> 3, 8]
            r = r.reshape(sh).any(axis=a + 1)
        n_faces = r.shape[ax]
        a_lo = [slice(None)] * 3; a_hi = [slice(None)] * 3
        a_lo[ax] = slice(0, n_faces); a_hi[ax] = slice(1, n_faces + 1)
        open_tile[tuple(a_lo)] |= r
        open_tile[tuple(a_hi)] |= r
    return open_tile, nbrs


name = sys.argv[1] if len(sys.argv) > 1 else "selftest"
if name == "selftest":
    f = np.zeros((16, 16, 16), dtype=bool)
    f[3, 3, 3] = True                      # singleton in tile (0,0,0)
    f[7, 12, 12] = f[8, 12, 12] = True     # a pair across the z face between tiles (0,1,1) and (1,1,1)
    f[12, 2, 7] = f[12, 2, 6] = True       # a pair inside tile (1,0,0)
    o, n = analyse(f)
    assert o.sum() == 2 and o[0, 1, 1] and o[1, 1, 1], o
    assert n[3, 3, 3] == 0 and n[7, 12, 12] == 1 and n[12, 2, 6] == 1
    print("selftest ok")
    sys.exit(0)

marks = sorted(int(a) for a in sys.argv[2:])
cfg = scenes.CONFIGS[name]
nx, ny, nz = cfg["size"]
s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
s.seed_block(*cfg["block"])
k = 0
for m in marks:
    while k < m:
        s.time_step(min(3.0 * s.cfl(), 0.033)); k += 1
    s.hash()
    fc = s.fluid_cells().astype(np.int64)
    f = np.zeros((nz, ny, nx), dtype=bool)
    f.reshape(-1)[fc] = True
    open_tile, nbrs = analyse(f)
    has = f.reshape(nz >> 3, 8, ny >> 3, 8, nx >> 3, 8).any(axis=(1, 3, 5))
    cnt = f.reshape(nz >> 3, 8, ny >> 3, 8, nx >> 3, 8).sum(axis=(1, 3, 5))
    closed = has & ~open_tile
    single = f & (nbrs == 0)
    print(f"{name} step {m}: {len(fc)} unknowns in {has.sum()} tiles; closed tiles (no fluid-fluid coupling across a tile face) {closed.sum()} "
          f"({closed.sum() / has.sum():.2f}) holding {cnt[closed].sum()} unknowns (max {cnt[closed].max() if closed.any() else 0} per tile); "
          f"singleton unknowns {single.sum()} ({single.sum() / len(fc):.3f}); tiles with <= 8 unknowns {(has & (cnt <= 8)).sum()}, <= 32: {(has & (cnt <= 32)).sum()}, "
          f"<= 128: {(has & (cnt <= 128)).sum()}", flush=True)
