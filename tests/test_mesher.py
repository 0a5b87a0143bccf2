"""Surface mesher.

CPU (`-m "not gpu"`): oracle/mesher_oracle.c against the golden vectors the real reference produced
(tests/golden/mesher.npz, tests/golden/make_golden_mesher.py) and against a live oracle/_ref when it is present.
GPU (`-m gpu`): libfluid_amd/csrc/mesher.hip through the C ABI against the same vectors. Bit-exact throughout: sampled
surface function (NaNs where the reference produces 0/0), vertex positions, vertex order, index lists."""
import os

import numpy as np
import pytest

import libfluid_amd as lfa
from libfluid_amd import scenes
from oracle import loader as orc
from tests import mesher_cases as mc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesher.npz")


@pytest.fixture(scope="module")
def golden():
    with np.load(GOLDEN) as z:
        return {k: z[k] for k in z.files}


def same(a, b):
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def all_cases(g):
    """Yields (values[2,2,2], positions, indices) for the 512 single-cell fixtures."""
    vo = io = 0
    k = 0
    for variant in (None, g["cases_mags"]):
        for case in range(256):
            nv, ni = g["cases_counts"][k]
            yield (mc.single_cell_values(case, None if variant is None else variant[case]),
                   g["cases_pos"][vo:vo + nv], g["cases_idx"][io:io + ni])
            vo, io, k = vo + nv, io + ni, k + 1


def test_all_256_cases_oracle(golden):
    for v, pos, idx in all_cases(golden):
        p, i = orc.mesher_mesh(None, (1, 1, 1), values=v, kind="oracle")
        assert same(p, pos) and same(i, idx)


@pytest.mark.parametrize("seed,size", [(1, (7, 6, 5)), (2, (1, 9, 1)), (3, (12, 1, 3))])
def test_marching_cubes_oracle_on_random_fields(golden, seed, size):
    v = golden[f"field{seed}_values"]
    assert same(v, mc.random_field(seed, size))
    p, i = orc.mesher_mesh(None, size, (0.25, -1.5, 3.0), 0.7, values=v, kind="oracle")
    assert same(p, golden[f"field{seed}_pos"]) and same(i, golden[f"field{seed}_idx"])


@pytest.mark.parametrize("name", mc.PARTICLE_CASES)
def test_sampling_and_mesh_oracle(golden, name):
    p, kw = mc.particle_case(name)
    assert same(p, golden[f"{name}_particles"])
    assert same(orc.mesher_surface(p, kind="oracle", **kw), golden[f"{name}_values"])
    pos, idx = orc.mesher_mesh(p, kind="oracle", **kw)
    assert same(pos, golden[f"{name}_pos"]) and same(idx, golden[f"{name}_idx"])


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref is only built where /root/reference exists")
def test_oracle_matches_live_reference():
    rng = np.random.default_rng(21)
    p = rng.uniform(0.5, 7.5, size=(900, 3))
    kw = dict(size=(9, 8, 10), grid_offset=(-0.1, 0.2, 0.05), cell_size=0.8, particle_extent=1.3, cell_radius=2, r=0.45)
    assert same(orc.mesher_surface(p, kind="oracle", **kw), orc.mesher_surface(p, kind="ref", **kw))
    a, b = orc.mesher_mesh(p, kind="oracle", **kw), orc.mesher_mesh(p, kind="ref", **kw)
    assert same(a[0], b[0]) and same(a[1], b[1])


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_all_256_cases_device(golden):
    m = lfa.Mesher((1, 1, 1))
    for v, pos, idx in all_cases(golden):
        m.set_values(v)
        p, i = m.marching_cubes()
        assert same(p, pos) and same(i, idx)
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,size", [(1, (7, 6, 5)), (2, (1, 9, 1)), (3, (12, 1, 3))])
def test_marching_cubes_device_on_random_fields(golden, seed, size):
    m = lfa.Mesher(size, (0.25, -1.5, 3.0), 0.7)
    m.set_values(golden[f"field{seed}_values"])
    p, i = m.marching_cubes()
    assert same(p, golden[f"field{seed}_pos"]) and same(i, golden[f"field{seed}_idx"])
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", mc.PARTICLE_CASES)
def test_sampling_and_mesh_device(golden, name):
    p, kw = mc.particle_case(name)
    r = kw.pop("r")
    m = lfa.Mesher(**kw)
    m.sample(p, r)
    assert same(m.values(), golden[f"{name}_values"])
    pos, idx = m.marching_cubes()
    assert same(pos, golden[f"{name}_pos"]) and same(idx, golden[f"{name}_idx"])
    # a second sampling on the same handle (fewer particles, then none) starts from a clean hash
    m.sample(p[: len(p) // 2], r)
    assert same(m.values(), orc.mesher_surface(p[: len(p) // 2], kind="oracle", r=r, **kw))
    m.sample(np.zeros((0, 3)), r)
    assert (m.values() == 1.0).all() and len(m.marching_cubes()[1]) == 0
    m.close()


@pytest.mark.gpu
def test_dam_break_surface_at_scale():
    """48^3 simulation cells of fluid meshed on a 2x finer grid (testbed/main.cpp:101-107): 880k particles, 100^3 cells."""
    p = scenes.seed_block((1, 1, 1), (49, 49, 49))["pos"]
    p = p[np.random.default_rng(9).permutation(len(p))]
    kw = dict(size=(100, 100, 100), grid_offset=(0.0, 0.0, 0.0), cell_size=0.5, particle_extent=1.0, cell_radius=3)
    m = lfa.Mesher(**kw)
    pos, idx = m.generate_mesh(p, 0.5)
    want_v = orc.mesher_surface(p, kind="oracle", r=0.5, **kw)
    assert same(m.values(), want_v)
    wp, wi = orc.mesher_mesh(None, kw["size"], kw["grid_offset"], kw["cell_size"], values=want_v, kind="oracle")
    assert same(pos, wp) and same(idx, wi) and len(idx) > 100000
    # closed surface: every edge is shared by exactly two triangles
    tri = idx.reshape(-1, 3).astype(np.int64)
    e = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", mc.PARTICLE_CASES)
@pytest.mark.parametrize("cuts", [(0.5,), (0.25, 0.6), (0.1, 0.2, 0.3, 0.7, 0.9)])
def test_z_windows_reproduce_the_single_grid_mesh(name, cuts, golden):
    """lfa_mesher_create_window (a rank of a slab run meshes its own cell layers): N windows that partition the grid in z, each
    fed only the particles it can see, in an arbitrary order, plus their global indices, give - concatenated in z order, indices
    shifted by the vertex counts of the windows below - the reference's mesh bit for bit (golden vectors of the single grid)."""
    p, kw = mc.particle_case(name)
    kw = dict(kw)
    r = kw.pop("r")
    nz = kw["size"][2]
    bounds = sorted({0, nz} | {max(1, min(nz - 1, int(round(c * nz)))) for c in cuts})
    rng = np.random.default_rng(3)
    pos_parts, idx_parts, below = [], [], 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        m = lfa.Mesher(window=(lo, hi), **kw)
        assert m.own == (lo, hi)
        # the particles this window may see (a slab rank holds its own layers and some ghosts), shuffled, with their input indices
        zc = (p[:, 2] - kw["grid_offset"][2]) / kw["cell_size"]
        see = np.nonzero((zc >= m.z0 - 1.0) & (zc <= m.z0 + m.n_planes + 1.0))[0]
        see = see[rng.permutation(len(see))]
        m.sample(p[see], r, ids=see.astype(np.uint32))
        s_lo = max(lo - 1, 0)
        assert same(m.values()[s_lo - m.z0: hi - m.z0 + 1], golden[f"{name}_values"][s_lo: hi + 1])
        m.marching_cubes()
        m.rebase(below)
        pos, idx = m.download_mesh()
        pos_parts.append(pos); idx_parts.append(idx)
        below += len(pos)
        m.close()
    assert same(np.concatenate(pos_parts), golden[f"{name}_pos"])
    assert np.array_equal(np.concatenate(idx_parts), golden[f"{name}_idx"])
