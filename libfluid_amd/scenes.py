"""Synthetic workloads: dam-break blocks seeded 2x2x2 per cell with a counter-based generator.

The generator is the build's own (the reference's seeding depends on pcg32 draw order, SURVEY 8(c) "seeding
hazard"): particle ``i`` of the block, axis ``a`` draws ``u = splitmix64(seed + (3 i + a + 1) * GOLDEN) >> 11``
scaled by 2^-53. ``i = 8 * (block cell index, x fastest) + (sx + 2 sy + 4 sz)``. The device-side twin is
``lfa_seed_block`` (libfluid_amd/csrc/boundary.hip) and produces bit-identical fp64 positions.
"""
import numpy as np

SEED = 0x5EED0001
GOLDEN = 0x9E3779B97F4A7C15
MASK = (1 << 64) - 1

PARTICLE_DTYPE = np.dtype(
    [("pos", "<f8", 3), ("vel", "<f8", 3), ("cx", "<f8", 3), ("cy", "<f8", 3), ("cz", "<f8", 3),
     ("old_pos", "<f8", 3), ("raw", "<u8")]
)
CELL_DTYPE = np.dtype([("vel", "<f8", 3), ("type", "u1"), ("pad", "u1", 7)])

# BASELINE.json configs -> (grid size, seeded block [lo, hi) in cells, method, blending)
PIC, FLIP, APIC = 0, 1, 2
CONFIGS = {
    "C1": dict(size=(64, 64, 64), block=((0, 0, 0), (32, 32, 32)), method=FLIP, blending=1.0),
    "testbed0": dict(size=(50, 50, 50), block=((15, 15, 15), (35, 35, 35)), method=APIC, blending=1.0),
    "C2": dict(size=(128, 128, 128), block=((0, 0, 0), (64, 64, 64)), method=APIC, blending=1.0),
    "C3": dict(size=(256, 256, 256), block=((0, 0, 0), (128, 128, 128)), method=FLIP, blending=0.95),
    "C4": dict(size=(512, 512, 512), block=((0, 0, 0), (128, 256, 256)), method=APIC, blending=1.0),
    "C5": dict(size=(1024, 512, 512), block=((0, 0, 0), (256, 256, 256)), method=APIC, blending=1.0),
}


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def uniform01(counter, seed=SEED):
    """U[0,1) fp64 from a uint64 counter array."""
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (np.asarray(counter, dtype=np.uint64) + np.uint64(1)) * np.uint64(GOLDEN)
    return (splitmix64(x) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def seed_block(lo, hi, cell_size=1.0, offset=(0.0, 0.0, 0.0), seed=SEED, velocity=(0.0, 0.0, 0.0)):
    """Particles for cells [lo, hi): 8 per cell, jittered inside their 2x2x2 sub-cell. Returns a 152-B AoS array."""
    lo = np.asarray(lo, dtype=np.int64)
    hi = np.asarray(hi, dtype=np.int64)
    ext = hi - lo
    ncell = int(ext[0] * ext[1] * ext[2])
    cidx = np.arange(ncell, dtype=np.int64)
    cx = lo[0] + cidx % ext[0]
    cy = lo[1] + (cidx // ext[0]) % ext[1]
    cz = lo[2] + cidx // (ext[0] * ext[1])
    i = (cidx[:, None] * 8 + np.arange(8, dtype=np.int64)[None, :]).reshape(-1)
    sub = np.tile(np.arange(8, dtype=np.int64), ncell)
    cells = np.stack([np.repeat(cx, 8), np.repeat(cy, 8), np.repeat(cz, 8)], axis=1).astype(np.float64)
    subs = np.stack([sub & 1, (sub >> 1) & 1, (sub >> 2) & 1], axis=1).astype(np.float64)
    u = np.stack([uniform01(3 * i.astype(np.uint64) + np.uint64(a), seed) for a in range(3)], axis=1)
    parts = np.zeros(i.shape[0], dtype=PARTICLE_DTYPE)
    off = np.asarray(offset, dtype=np.float64)
    parts["pos"] = off[None, :] + (cells + (subs + u) * 0.5) * cell_size
    parts["old_pos"] = parts["pos"]
    parts["vel"] = np.asarray(velocity, dtype=np.float64)[None, :]
    return parts


def sphere_solid_cells(size, center, radius):
    """Cells whose centre lies inside a sphere (testbed setup 4 style, testbed/main.cpp:167-176) as int32[k,3]."""
    x, y, z = np.meshgrid(np.arange(size[0]), np.arange(size[1]), np.arange(size[2]), indexing="ij")
    d2 = (x + 0.5 - center[0]) ** 2 + (y + 0.5 - center[1]) ** 2 + (z + 0.5 - center[2]) ** 2
    m = d2 < radius * radius
    return np.stack([x[m], y[m], z[m]], axis=1).astype(np.int32)


# ---- triangle meshes for the voxelizer: positions float64[nv,3], indices uint64[3 nt] ----------
def box_mesh(lo, hi):
    """Axis-aligned box, 12 triangles."""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    c = np.array([[(lo, hi)[(i >> d) & 1][d] for d in range(3)] for i in range(8)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    tri = [t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))]
    return c, np.asarray(tri, dtype=np.uint64).reshape(-1)


def icosphere(center, radius, subdivisions=2):
    """Closed triangulated sphere (subdivided icosahedron)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    v = [np.asarray(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        cache, nf = {}, []

        def mid(a, b):
            k = (min(a, b), max(a, b))
            if k not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[k] = len(v) - 1
            return cache[k]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    pos = np.asarray(center, dtype=np.float64)[None, :] + radius * np.asarray(v, dtype=np.float64)
    return pos, np.asarray(f, dtype=np.uint64).reshape(-1)


def rotate_mesh(pos, axis, angle, about):
    """Rigid rotation of mesh vertices (Rodrigues), so triangles cut the grid at generic angles."""
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    p = pos - np.asarray(about, dtype=np.float64)[None, :]
    c, s_ = np.cos(angle), np.sin(angle)
    r = p * c + np.cross(a[None, :], p) * s_ + a[None, :] * (p @ a)[:, None] * (1.0 - c)
    return r + np.asarray(about, dtype=np.float64)[None, :]
