"""Inputs of the surface-mesher tests."""
import numpy as np

from libfluid_amd import scenes

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])


def single_cell_values(case, magnitudes=None):
    """Values at the 8 corners of one cell: corner i inside (negative) iff bit i of `case`. float64[2,2,2] (z,y,x)."""
    v = np.ones((2, 2, 2))
    for i, (x, y, z) in enumerate(CORNERS):
        m = 1.0 if magnitudes is None else magnitudes[i]
        v[z, y, x] = -m if (case >> i) & 1 else m
    return v


def random_field(seed, size):
    """Smooth-ish random function with both signs on a (size+1)^3 point grid."""
    rng = np.random.default_rng(seed)
    return rng.normal(size=(size[2] + 1, size[1] + 1, size[0] + 1)) + 0.15


# name -> (particles float64[n,3], mesher settings)
def particle_case(name):
    if name == "block":  # dam-break block, input order shuffled: the sums depend on the order inside a cell
        p = scenes.seed_block((2, 2, 2), (9, 8, 10))["pos"]
        p = p[np.random.default_rng(5).permutation(len(p))]
        return p, dict(size=(12, 12, 14), grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=2.0, cell_radius=3, r=0.5)
    if name == "fine":  # the testbed's relation between the grids: mesher cells half a simulation cell (main.cpp:101-107)
        p = scenes.seed_block((1, 1, 1), (6, 7, 5))["pos"]
        p = p[np.random.default_rng(6).permutation(len(p))]
        return p, dict(size=(16, 18, 14), grid_offset=(0.3, -0.2, 0.1), cell_size=0.5, particle_extent=1.0, cell_radius=2, r=0.35)
    if name == "edges":  # particles in cells with an index 0 (dropped, src/mesher.cpp:337), outside, and on the max side
        rng = np.random.default_rng(8)
        p = rng.uniform(-1.0, 9.0, size=(600, 3))
        return p, dict(size=(8, 8, 8), grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=1.5, cell_radius=2, r=0.4)
    if name == "sparse":  # isolated particles: has_particles with zero weight gives 0/0 = NaN samples, like the reference
        p = np.array([[3.2, 3.3, 3.4], [7.7, 2.2, 5.5], [7.9, 2.3, 5.6]])
        return p, dict(size=(10, 8, 9), grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=0.5, cell_radius=2, r=0.3)
    raise KeyError(name)


PARTICLE_CASES = ["block", "fine", "edges", "sparse"]
