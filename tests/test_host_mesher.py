"""The C++17 host class fluid_amd::mesher (libfluid_amd/host/mesher.h).

CPU: it compiles with g++ against include/libfluid_amd.h and links to libfluid_amd.so.
GPU: generate_mesh + save_obj against the real reference's mesh (tests/golden/mesher.npz) and the reference's OBJ writer;
meshing the particles resident in a simulation handle equals meshing their downloaded positions."""
import os
import subprocess

import numpy as np
import pytest

import libfluid_amd as lfa
from libfluid_amd import scenes
from oracle import loader as orc
from tests import mesher_cases as mc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_mesher_driver.cpp")
GOLDEN = os.path.join(ROOT, "tests", "golden", "mesher.npz")


def build_driver(tmp_path):
    exe = str(tmp_path / "host_mesher_driver")
    lfa.load_library()
    cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-Wall", "-Wextra", *os.environ.get("LFA_HOST_CXXFLAGS", "").split(), "-o", exe, SRC, "-L" + os.path.dirname(lfa.LIB_PATH),
           "-l:libfluid_amd.so", "-Wl,-rpath," + os.path.dirname(lfa.LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_host_mesher_compiles_and_links(tmp_path):
    build_driver(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["block", "fine"])
def test_host_mesher_matches_reference(tmp_path, name):
    with np.load(GOLDEN) as z:
        want_pos, want_idx = z[f"{name}_pos"], z[f"{name}_idx"]
    p, kw = mc.particle_case(name)
    exe = build_driver(tmp_path)
    p.tofile(tmp_path / "p.bin")
    args = [exe, str(tmp_path / "p.bin"), *(str(x) for x in kw["size"]), *(repr(float(x)) for x in kw["grid_offset"]),
            repr(kw["cell_size"]), repr(kw["particle_extent"]), str(kw["cell_radius"]), repr(kw["r"]),
            str(tmp_path / "m.bin"), str(tmp_path / "m.obj")]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(tmp_path / "m.bin", "rb").read()
    nv, ni = np.frombuffer(raw, dtype=np.uint64, count=2)
    pos = np.frombuffer(raw, dtype=np.float64, count=3 * int(nv), offset=16).reshape(-1, 3)
    idx = np.frombuffer(raw, dtype=np.uint64, count=int(ni), offset=16 + 24 * int(nv))
    assert np.array_equal(pos, want_pos) and np.array_equal(idx, want_idx)
    if orc.have_ref():  # testbed F3: sim_mesh.save_obj (testbed/main.cpp:328-334)
        text, _ = orc.ref_mesh_obj(want_pos, want_idx)
        assert open(tmp_path / "m.obj", "rb").read() == text


@pytest.mark.gpu
def test_meshing_resident_particles_equals_meshing_downloaded_positions():
    size = (24, 24, 24)
    sim = lfa.Sim(size, method=lfa.APIC)
    parts = scenes.seed_block((2, 1, 3), (12, 10, 11))
    parts = parts[np.random.default_rng(4).permutation(len(parts))]
    sim.upload_particles(parts)
    for _ in range(2):
        sim.time_step(0.01)  # particles move, the binned storage order differs from the upload order
    kw = dict(size=(48, 48, 48), grid_offset=(0.0, 0.0, 0.0), cell_size=0.5, particle_extent=1.0, cell_radius=3)
    a, b = lfa.Mesher(**kw), lfa.Mesher(**kw)
    a.sample_sim(sim, 0.5)
    pos = sim.download_particles(write_positions=True)["pos"]
    b.sample(pos, 0.5)
    va, vb = a.values(), b.values()
    assert np.array_equal(va, vb, equal_nan=True) and (va < 0).sum() > 1000
    ma, mb = a.marching_cubes(), b.marching_cubes()
    # vertices next to a NaN sample are NaN, in the reference too
    assert np.array_equal(ma[0], mb[0], equal_nan=True) and np.array_equal(ma[1], mb[1]) and len(ma[1]) > 1000
    for h in (a, b, sim):
        h.close()
