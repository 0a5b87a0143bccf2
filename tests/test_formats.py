"""On-disk formats: the plain-text point cloud and Wavefront .OBJ writers of the host headers
libfluid_amd/host/point_cloud.h and mesh.h against the real reference's (include/fluid/data_structures/point_cloud.h,
mesh.h), byte for byte. Host code only: everything here runs without a GPU. Golden files under tests/golden/formats/
were written by the reference through oracle/_ref (tests/golden/make_golden_formats.py)."""
import os
import subprocess

import numpy as np
import pytest

from libfluid_amd import scenes
from oracle import loader as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_formats_driver.cpp")
GOLD = os.path.join(ROOT, "tests", "golden", "formats")


def sample_points():
    rng = np.random.default_rng(7)
    pts = rng.normal(size=(64, 3)) * np.array([1.0, 1e-4, 1e5])
    pts[0] = (0.0, -0.0, 1.0)
    pts[1] = (1e-7, 123456789.0, -2.5)
    pts[2] = (1.0 / 3.0, 1e21, 1e-21)
    return pts


def sample_mesh():
    pos, idx = scenes.icosphere((1.0, -2.0, 0.5), 1.75, 1)
    uv = np.stack([np.arctan2(pos[:, 1], pos[:, 0]), pos[:, 2]], axis=1)
    return pos, idx, uv


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fmt") / "host_formats_driver")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-fopenmp", "-Wall", "-Wextra", *os.environ.get("LFA_HOST_CXXFLAGS", "").split(), "-o", out, SRC], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def write_mesh(path, pos, idx, uv):
    with open(path, "wb") as f:
        np.array([len(pos), len(idx)], dtype=np.uint64).tofile(f)
        pos.astype(np.float64).tofile(f)
        idx.astype(np.uint64).tofile(f)
        uv.astype(np.float64).tofile(f)


def test_point_cloud_text_is_byte_identical(exe, tmp_path):
    pts = sample_points()
    pts.tofile(tmp_path / "p.bin")
    subprocess.run([exe, "points", str(tmp_path / "p.bin"), str(tmp_path / "p.txt")], check=True)
    got = open(tmp_path / "p.txt", "rb").read()
    assert got == open(os.path.join(GOLD, "points.txt"), "rb").read()
    if orc.have_ref():
        assert got == orc.ref_points_text(pts)


@pytest.mark.parametrize("count", [0, 5, 64, 10 ** 9])
def test_point_cloud_parse(exe, tmp_path, count):
    text = open(os.path.join(GOLD, "points.txt"), "rb").read() + b"7.5 8.5\nnot a number 1 2 3\n"
    open(tmp_path / "in.txt", "wb").write(text)
    subprocess.run([exe, "parse", str(tmp_path / "in.txt"), str(count), str(tmp_path / "out.bin")], check=True)
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float64).reshape(-1, 3)
    assert len(got) == min(count, 64)  # stops at the incomplete record, like `in >> x >> y >> z` failing
    want = np.array([[float(t) for t in line.split()] for line in text.split(b"\n")[:len(got)]]).reshape(-1, 3)
    assert np.array_equal(got, want)
    if orc.have_ref():
        assert np.array_equal(got, orc.ref_points_parse(text, count))


@pytest.mark.parametrize("mode", range(8))
def test_obj_text_is_byte_identical(exe, tmp_path, mode):
    pos, idx, uv = sample_mesh()
    write_mesh(tmp_path / "m.bin", pos, idx, uv)
    subprocess.run([exe, "obj", str(tmp_path / "m.bin"), str(mode), str(tmp_path / "m.obj"), str(tmp_path / "n.bin")], check=True)
    got = open(tmp_path / "m.obj", "rb").read()
    assert got == open(os.path.join(GOLD, f"mesh_mode{mode}.obj"), "rb").read()
    if orc.have_ref():
        text, nrm = orc.ref_mesh_obj(pos, idx, uv if mode & 2 else None, bool(mode & 1), bool(mode & 4))
        assert got == text
        if mode & 1:  # generate_normals: same sums, same normalisation
            assert np.array_equal(np.fromfile(tmp_path / "n.bin", dtype=np.float64).reshape(-1, 3), nrm)


def test_degenerate_normals_fall_back_to_x_axis(exe, tmp_path):
    pos = np.array([[0, 0, 0], [1, 0, 0], [2, 0, 0], [5, 5, 5]], dtype=np.float64)  # collinear triangle + unused vertex
    idx = np.array([0, 1, 2], dtype=np.uint64)
    write_mesh(tmp_path / "m.bin", pos, idx, np.zeros((4, 2)))
    subprocess.run([exe, "obj", str(tmp_path / "m.bin"), "1", str(tmp_path / "m.obj"), str(tmp_path / "n.bin")], check=True)
    nrm = np.fromfile(tmp_path / "n.bin", dtype=np.float64).reshape(-1, 3)
    assert np.array_equal(nrm, np.tile([1.0, 0.0, 0.0], (4, 1)))
