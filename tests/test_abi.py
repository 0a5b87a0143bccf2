"""CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol include/libfluid_amd.h declares;
without a GPU it refuses to create a handle (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import libfluid_amd as lfa
from libfluid_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "libfluid_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lfa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = lfa.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/libfluid_amd.h but not exported"
    assert set(names) == set(lfa.SIGNATURES), set(names) ^ set(lfa.SIGNATURES)


def test_params_struct_matches_reference_defaults():
    p = lfa.default_params()
    # include/fluid/simulation.h:182-190 and include/fluid/pressure_solver.h:39-42
    assert np.isnan(p.cell_size)
    assert (p.cfl_number, p.blending_factor, p.density, p.boundary_skin_width, p.correction_stiffness) == \
        (3.0, 1.0, 1.0, 0.1, 5.0)
    assert p.velocity_extrapolation_iterations == 1 and p.simulation_method == lfa.APIC
    assert (p.tau, p.sigma, p.tolerance, p.max_iterations) == (0.97, 0.25, 1e-6, 200)


def test_no_cpu_fallback(gpu_available):
    if gpu_available:
        pytest.skip("a GPU is present")
    with pytest.raises(lfa.LibfluidError) as e:
        lfa.Sim((8, 8, 8))
    assert e.value.code == -2  # LFA_E_NO_DEVICE


def test_layouts_are_the_reference_layouts():
    assert scenes.PARTICLE_DTYPE.itemsize == 152 and scenes.CELL_DTYPE.itemsize == 32
    assert scenes.PARTICLE_DTYPE.fields["raw"][1] == 144 and scenes.CELL_DTYPE.fields["type"][1] == 24


def test_seed_block_is_deterministic_and_in_cell():
    a = scenes.seed_block((1, 2, 3), (4, 5, 6))
    b = scenes.seed_block((1, 2, 3), (4, 5, 6))
    assert np.array_equal(a["pos"], b["pos"]) and len(a) == 27 * 8
    cell = np.floor(a["pos"]).astype(int)
    assert cell.min(axis=0).tolist() == [1, 2, 3] and cell.max(axis=0).tolist() == [3, 4, 5]
    # 2x2x2 sub-cell stratification: one particle per octant
    octant = ((a["pos"] - cell) >= 0.5).astype(int) @ np.array([1, 2, 4])
    assert np.array_equal(np.sort(octant.reshape(-1, 8), axis=1), np.tile(np.arange(8), (27, 1)))
