"""Solid-boundary voxelizer.

CPU (`-m "not gpu"`): oracle/voxelizer_oracle.c against the golden vectors the real reference produced
(tests/golden/voxelizer.npz, tests/golden/make_golden_voxelizer.py) and against a live oracle/_ref when it is present.
GPU (`-m gpu`): libfluid_amd/csrc/voxelizer.hip through the C ABI against the same vectors. Everything is bit-exact:
voxel types, grid placement, and the ordered cell lists of the Maya VoxelizerNode (voxelizer_node.cpp:285-343)."""
import os

import numpy as np
import pytest

import libfluid_amd as lfa
from oracle import loader as orc
from tests import voxel_cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxelizer.npz")


@pytest.fixture(scope="module")
def golden():
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


def inputs(g, name):
    return g[f"{name}_pos"], g[f"{name}_idx"], float(g[f"{name}_cs"]), g[f"{name}_off"], g[f"{name}_ref_size"]


def cells_from_types(types, kinds, grid_min=None, ref_size=None):
    """grid3::for_each order (z slowest, x fastest) list of the cells whose type is in `kinds`."""
    z, y, x = np.nonzero(np.isin(types, kinds))
    c = np.stack([x, y, z], axis=1).astype(np.int64)
    if ref_size is not None:
        c = c + np.asarray(grid_min, dtype=np.int64)[None, :]
        c = c[np.all((c >= 0) & (c < np.asarray(ref_size)[None, :]), axis=1)]
    return c.astype(np.int32)


@pytest.mark.parametrize("name", voxel_cases.NAMES)
def test_golden_inputs_are_the_generated_meshes(golden, name):
    pos, idx, cs, off, rs = voxel_cases.make(name)
    gp, gi, gcs, goff, grs = inputs(golden, name)
    assert np.array_equal(pos, gp) and np.array_equal(idx, gi) and cs == gcs
    assert np.array_equal(np.asarray(off), goff) and np.array_equal(np.asarray(rs), grs)


@pytest.mark.parametrize("name", voxel_cases.NAMES)
def test_oracle_matches_reference_golden(golden, name):
    pos, idx, cs, off, rs = inputs(golden, name)
    gmin, goff, types = orc.voxelize(pos, idx, cs, off, kind="oracle")
    assert np.array_equal(gmin, golden[f"{name}_grid_min"])
    assert np.array_equal(goff, golden[f"{name}_grid_off"])
    assert np.array_equal(types, golden[f"{name}_types"])
    # the lists in the fixture are consistent with the types (for_each order, clipping)
    assert np.array_equal(cells_from_types(types, [0], gmin, rs), golden[f"{name}_cells_ref_interior"])
    assert np.array_equal(cells_from_types(types, [0, 2]), golden[f"{name}_cells_all"])


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref is only built where /root/reference exists")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_matches_live_reference_on_random_meshes(seed):
    rng = np.random.default_rng(seed)
    nv = 40
    pos = rng.uniform(1.0, 11.0, size=(nv, 3))
    idx = rng.integers(0, nv, size=3 * 25).astype(np.uint64)
    cs, off = float(rng.choice([0.25, 0.37, 1.0])), rng.uniform(-1.0, 1.0, size=3)
    a, b = orc.voxelize(pos, idx, cs, off, kind="oracle"), orc.voxelize(pos, idx, cs, off, kind="ref")
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", voxel_cases.NAMES)
def test_device_voxelizer_is_bit_exact(golden, name):
    pos, idx, cs, off, rs = inputs(golden, name)
    v = lfa.Voxels.from_mesh(pos, idx, cs, off)
    assert np.array_equal(v.grid_min, golden[f"{name}_grid_min"])
    assert np.array_equal(v.grid_offset, golden[f"{name}_grid_off"])
    assert np.array_equal(v.types(), golden[f"{name}_types"])
    assert np.array_equal(v.cells(True, False, rs), golden[f"{name}_cells_ref_interior"])
    assert np.array_equal(v.cells(True, True, None), golden[f"{name}_cells_all"])
    assert np.array_equal(v.cells(False, True, None), cells_from_types(golden[f"{name}_types"], [2]))
    # 32-bit indices (mesh<double, int, ...> of the Maya node) take the same path
    v32 = lfa.Voxels.from_mesh(pos, idx.astype(np.uint32), cs, off)
    assert np.array_equal(v32.types(), golden[f"{name}_types"])
    v.close()
    v32.close()


@pytest.mark.gpu
def test_staged_calls_equal_the_fused_sequence(golden):
    name = "box_rot"
    pos, idx, cs, off, rs = inputs(golden, name)
    want = golden[f"{name}_types"]
    v = lfa.Voxels.create(want.shape[::-1], golden[f"{name}_grid_off"], cs)
    assert (v.types() == lfa.VOX_INTERIOR).all()           # grid3<cell_type>(size, interior), src/voxelizer.cpp:38
    v.voxelize_triangles(pos, idx[:18])                     # voxelize_triangle is incremental (voxelizer.cpp:76-80)
    v.voxelize_triangles(pos, idx[18:])
    surf = v.types()
    assert np.array_equal(surf == lfa.VOX_SURFACE, want == lfa.VOX_SURFACE) and not (surf == lfa.VOX_EXTERIOR).any()
    v.mark_exterior()
    assert np.array_equal(v.types(), want)
    v.mark_exterior()                                       # idempotent
    assert np.array_equal(v.types(), want)
    # a host may edit the voxels between the calls (public member voxelizer::voxels)
    edited = surf.copy()
    edited[:, :, :] = lfa.VOX_INTERIOR
    edited[3, :, :] = lfa.VOX_SURFACE                       # a wall across the grid: only z < 3 is reachable from the corner
    v.upload(edited)
    v.mark_exterior()
    got = v.types()
    assert (got[:3] == lfa.VOX_EXTERIOR).all() and (got[3] == lfa.VOX_SURFACE).all() and (got[4:] == lfa.VOX_INTERIOR).all()
    v.close()


@pytest.mark.gpu
def test_surface_corner_and_empty_mesh():
    # voxel (0,0,0) on the surface: mark_exterior returns at once (src/voxelizer.cpp:88-90)
    v = lfa.Voxels.create((6, 5, 4), (0.0, 0.0, 0.0), 1.0)
    t = np.zeros((4, 5, 6), dtype=np.uint8)
    t[0, 0, 0] = lfa.VOX_SURFACE
    v.upload(t)
    v.mark_exterior()
    assert np.array_equal(v.types(), t)
    v.close()
    # no triangles: everything is exterior; an empty vertex list gives the 2^3 grid around vec3d() of get_bounding_box
    v = lfa.Voxels.from_mesh(np.zeros((0, 3)), np.zeros(0, dtype=np.uint64), 1.0, (0.0, 0.0, 0.0))
    assert v.size == (2, 2, 2) and (v.types() == lfa.VOX_EXTERIOR).all() and len(v.cells(True, True)) == 0
    v.close()
    with pytest.raises(lfa.LibfluidError):
        lfa.Voxels.from_mesh(np.zeros((3, 3)), np.array([0, 1, 7], dtype=np.uint64))


@pytest.mark.gpu
def test_solid_cells_from_voxels_equal_the_host_list(golden):
    name = "sphere_clip"
    pos, idx, cs, off, rs = inputs(golden, name)
    size = tuple(int(x) for x in rs)
    v = lfa.Voxels.from_mesh(pos, idx, cs, off)
    a = lfa.Sim(size)
    a.set_solid_from_voxels(v, True, True)
    b = lfa.Sim(size)
    b.set_solid_cells(v.cells(True, True, rs))
    ta, tb = a.cells()["type"], b.cells()["type"]
    assert np.array_equal(ta, tb) and (ta == 4).sum() == len(v.cells(True, True, rs)) > 0
    for s in (a, b):
        s.close()
    v.close()


@pytest.mark.gpu
def test_large_grid_flood_fill_reaches_the_fixed_point():
    # 200^3 voxels around a sphere with a sphere-shaped cavity: many relaxation passes across blocks
    from libfluid_amd import scenes
    p1, i1 = scenes.icosphere((50.0, 50.0, 50.0), 45.0, 3)
    p2, i2 = scenes.icosphere((50.0, 50.0, 50.0), 20.0, 2)
    pos, idx = np.concatenate([p1, p2]), np.concatenate([i1, i2 + np.uint64(len(p1))])
    v = lfa.Voxels.from_mesh(pos, idx, 0.5, (0.0, 0.0, 0.0))
    got = v.types()
    _, _, want = orc.voxelize(pos, idx, 0.5, (0.0, 0.0, 0.0), kind="oracle")
    assert got.shape == want.shape and np.array_equal(got, want)
    assert (got == lfa.VOX_INTERIOR).sum() > 100000
    v.close()
