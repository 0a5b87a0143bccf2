"""The C++17 host classes fluid_amd::voxelizer / fluid_amd::obstacle (libfluid_amd/host/voxelizer.h).

CPU: they compile with g++ against include/libfluid_amd.h and link to libfluid_amd.so.
GPU: staged members and the obstacle constructor against the real reference's results (tests/golden/voxelizer.npz)."""
import os
import subprocess

import numpy as np
import pytest

import libfluid_amd as lfa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_voxelizer_driver.cpp")
GOLDEN = os.path.join(ROOT, "tests", "golden", "voxelizer.npz")


def build_driver(tmp_path):
    exe = str(tmp_path / "host_voxelizer_driver")
    lfa.load_library()
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", *os.environ.get("LFA_HOST_CXXFLAGS", "").split(), "-o", exe, SRC, "-L" + os.path.dirname(lfa.LIB_PATH),
           "-l:libfluid_amd.so", "-Wl,-rpath," + os.path.dirname(lfa.LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_host_voxelizer_compiles_and_links(tmp_path):
    build_driver(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sphere_clip", "box_rot", "two_shells"])
def test_host_voxelizer_matches_reference(tmp_path, name):
    with np.load(GOLDEN) as z:
        g = {k: z[k] for k in z.files if k.startswith(name + "_")}
    pos, idx = g[f"{name}_pos"], g[f"{name}_idx"]
    cs, off, rs = float(g[f"{name}_cs"]), g[f"{name}_off"], g[f"{name}_ref_size"]
    exe = build_driver(tmp_path)
    fm, ft, fc = tmp_path / "mesh.bin", tmp_path / "types.bin", tmp_path / "cells.bin"
    with open(fm, "wb") as f:
        np.array([len(pos), len(idx)], dtype=np.uint64).tofile(f)
        pos.astype(np.float64).tofile(f)
        idx.astype(np.uint64).tofile(f)
    r = subprocess.run([exe, str(fm), repr(cs), *(repr(float(x)) for x in off), *(str(int(x)) for x in rs), str(ft), str(fc)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    vals = [int(x) for x in r.stdout.split()]
    gmin, size = vals[:3], vals[3:6]
    want = g[f"{name}_types"]
    assert gmin == list(g[f"{name}_grid_min"]) and size == list(want.shape[::-1])
    # get_overlapping_cell_range (src/voxelizer.cpp:41-57): min = max(-offset, 0), max = clamp(offset + size, 0, ref)
    assert vals[6:9] == [max(-o, 0) for o in gmin]
    assert vals[9:12] == [min(max(o + s, 0), int(r_)) for o, s, r_ in zip(gmin, size, rs)]
    assert np.array_equal(np.fromfile(ft, dtype=np.uint8).reshape(want.shape), want)
    assert np.array_equal(np.fromfile(fc, dtype=np.int32).reshape(-1, 3), g[f"{name}_cells_ref_interior"])
