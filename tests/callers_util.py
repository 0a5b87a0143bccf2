"""Scenarios, inputs and builds of tests/callers/reference_callers.cpp - host code written against the reference's own headers
and types, compiled unchanged against the real reference (CPU) and against this repository's shim headers (device).
Shared by tests/test_ref_callers.py and tests/golden/make_golden_callers.py."""
import os
import subprocess

import numpy as np

from tests import util, voxel_cases

ROOT = util.ROOT
SRC = os.path.join(ROOT, "tests", "callers", "reference_callers.cpp")
SHIM = os.path.join(ROOT, "libfluid_amd", "host", "shim")
SHIM_STANDALONE = os.path.join(ROOT, "libfluid_amd", "host", "shim_standalone")
REFERENCE = os.environ.get("REFERENCE_DIR", "/root/reference")
REF_INCLUDE = os.path.join(REFERENCE, "include")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "callers_ref")                  # the real reference (oracle/Makefile)
DEV_REFTYPES_EXE = os.path.join(ROOT, "oracle", "_ref", "callers_dev_reftypes")  # shim + the reference's headers, built where they exist

# name -> (scenario, argument builder). Sizes: the reference finishes each in about a second.
SCENARIOS = {
    "testbed_scene0": ("testbed", lambda d: ["20", "0", "3"]),     # the testbed's default scene: a box of fluid in the air
    "testbed_scene3": ("testbed", lambda d: ["20", "3", "4"]),     # dam break against the wall: pressure solves from step 1
    "testbed_scene4": ("testbed", lambda d: ["20", "4", "4"]),     # fluid source + voxel sphere (seeded positions are random by design)
    "gridnode_flip": ("gridnode", lambda d: [_particles_file(d, True), "24", "2", "1"]),
    "gridnode_apic": ("gridnode", lambda d: [_particles_file(d, False), "24", "2", "2"]),
    "gridnode_source": ("gridnode", lambda d: ["-", "24", "3", "2"]),
    "mesher": ("mesher", lambda d: [_points_file(d), "28", "0.5", "0.5"]),
    "voxelizer_sphere": ("voxelizer", lambda d: _voxel_args(d, "sphere")),
    "voxelizer_box_rot": ("voxelizer", lambda d: _voxel_args(d, "box_rot")),
    "voxelizer_clip": ("voxelizer", lambda d: _voxel_args(d, "sphere_clip")),
    "points": ("points", lambda d: [_points_file(d)]),
}
OUTPUT = {"testbed": "testbed.bin", "gridnode": "gridnode.bin", "mesher": "mesher.bin", "voxelizer": "voxelizer.bin", "points": None}


def _particles_file(d, identity):
    """The particles a previous GridNode evaluation left behind: a block of fluid on the floor, one cell from the -x and -z walls,
    in the node's world coordinates (cell size 0.5, offset (-1, 0.25, 2)). `identity`: cx.x carries the particle's number
    (PIC / FLIP never touch cx; for APIC it is the affine matrix and stays zero)."""
    parts = util.scenes.seed_block((1, 0, 1), (7, 12, 15), cell_size=0.5, offset=(-1.0, 0.25, 2.0))
    if identity:
        parts["cx"][:, 0] = np.arange(len(parts))
    path = os.path.join(d, "node_particles.bin")
    parts.tofile(path)
    return path


def _points_file(d):
    p = util.scenes.seed_block((1, 1, 1), (5, 6, 4))["pos"]
    p = p[np.random.default_rng(11).permutation(len(p))]
    path = os.path.join(d, "points.bin")
    np.ascontiguousarray(p, dtype=np.float64).tofile(path)
    return path


def _voxel_args(d, case):
    pos, idx, cell, off, ref = voxel_cases.make(case)
    path = os.path.join(d, f"mesh_{case}.bin")
    with open(path, "wb") as f:
        np.array([len(pos), len(idx)], dtype=np.uint64).tofile(f)
        np.ascontiguousarray(pos, dtype=np.float64).tofile(f)
        np.ascontiguousarray(idx, dtype=np.uint64).tofile(f)
    return [path, repr(cell), *map(repr, off), *map(str, ref)]


def read_dump(path):
    """name\\0 u64 count f64[count] records -> dict of float64 arrays."""
    out = {}
    raw = open(path, "rb").read()
    at = 0
    while at < len(raw):
        end = raw.index(b"\0", at)
        name = raw[at:end].decode()
        n = int(np.frombuffer(raw, dtype="<u8", count=1, offset=end + 1)[0])
        out[name] = np.frombuffer(raw, dtype="<f8", count=n, offset=end + 9).copy()
        at = end + 9 + 8 * n
    return out


def run(exe, name, workdir):
    """Runs one scenario; returns (records, stdout, text files)."""
    scenario, args = SCENARIOS[name]
    d = os.path.join(str(workdir), name + "_" + os.path.basename(exe))
    os.makedirs(d, exist_ok=True)
    r = subprocess.run([exe, scenario, d, *args(d)], capture_output=True, text=True)
    assert r.returncode == 0, (name, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    rec = read_dump(os.path.join(d, OUTPUT[scenario])) if OUTPUT[scenario] else {}
    texts = {}
    for fn in ("mesh.obj", "points.txt"):
        p = os.path.join(d, fn)
        if os.path.exists(p):
            texts[fn] = open(p).read()
    return rec, r.stdout, texts


def have_reference():
    return os.path.isdir(REF_INCLUDE)


def lib_flags():
    import libfluid_amd as lfa
    lfa.load_library()
    libdir = os.path.dirname(lfa.LIB_PATH)
    return ["-L" + libdir, "-l:libfluid_amd.so", "-Wl,-rpath," + libdir]


def build_device(out, reference_types, extra=()):
    """The caller program against the shim headers: with the reference's headers behind them (`reference_types`) or with the
    self-contained value types (no libfluid checkout at all). The source file is the same, unchanged."""
    inc = ["-I" + SHIM] + (["-I" + REF_INCLUDE] if reference_types else ["-I" + SHIM_STANDALONE])
    cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-Wall", "-Wextra", "-Werror", *os.environ.get("LFA_HOST_CXXFLAGS", "").split(),
           *inc, *extra, "-o", out, SRC, *lib_flags()]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return out


def build_reference():
    """oracle/_ref/callers_ref: the same source against the real reference (needs /root/reference)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "callers"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(REF_EXE), r.stdout + r.stderr
    return REF_EXE


def match_particles(pos, ref_pos, tol, strays=0.0):
    """Index array m with pos[m[i]] ~ ref_pos[i]: the particle sets are the same, the array orders are not (the reference sorts
    with an unstable std::sort every step, the device keeps its own order). Nearest neighbour, checked to be a bijection.
    `strays`: the fraction of particles allowed to be further than `tol` from their partner - particles pressed into the same
    point of a wall get a RANDOM push in the reference (std::random_device, src/simulation.cpp:573,587): a run of the reference
    does not reproduce such a particle either. Returns (m, ok) then, ok marking the particles within `tol`."""
    from scipy.spatial import cKDTree
    pos = pos.reshape(-1, 3)
    ref_pos = ref_pos.reshape(-1, 3)
    assert pos.shape == ref_pos.shape, (pos.shape, ref_pos.shape)
    dist, m = cKDTree(pos).query(ref_pos)
    ok = dist <= tol
    assert (~ok).sum() <= strays * len(ok), \
        f"{(~ok).sum()} of {len(ok)} particles are further than {tol:.1e} from their reference position (worst {dist.max():.3e})"
    # (particles pressed into the very same point of a wall cannot be told apart by position: they count as strays)
    uniq, first, count = np.unique(m, return_index=True, return_counts=True)
    shared = np.isin(m, uniq[count > 1])
    assert (shared & ok).sum() <= strays * len(ok), f"nearest-neighbour matching is not one-to-one ({(shared & ok).sum()} share a partner)"
    ok &= ~shared
    return (m, ok) if strays else m
