"""TEST INFRASTRUCTURE ONLY: ctypes loader for the CPU oracle (oracle/liboracle.so, prefix ``orc_``) and, when it
has been built in this container, the real reference hot path (oracle/_ref/libref.so, prefix ``ref_``).

Both libraries export the same stage-level entry points, so one wrapper class drives either. Nothing in
``libfluid_amd/`` imports this module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.environ.get("LFA_ORACLE_SO") or os.path.join(HERE, "liboracle.so")  # (LFA_ORACLE_SO: the sanitizer build, `make asan`)
REF_SO = os.path.join(HERE, "_ref", "libref.so")

# Reference layouts: include/fluid/simulation.h:24-34 and include/fluid/mac_grid.h:15-27.
PARTICLE_DTYPE = np.dtype(
    [("pos", "<f8", 3), ("vel", "<f8", 3), ("cx", "<f8", 3), ("cy", "<f8", 3), ("cz", "<f8", 3),
     ("old_pos", "<f8", 3), ("raw", "<u8")]
)
CELL_DTYPE = np.dtype([("vel", "<f8", 3), ("type", "u1"), ("pad", "u1", 7)])
assert PARTICLE_DTYPE.itemsize == 152 and CELL_DTYPE.itemsize == 32

AIR, FLUID, SOLID = 1, 2, 4
PIC, FLIP, APIC = 0, 1, 2


def build(force=False):
    """Compile liboracle.so (always possible: plain C) and oracle/_ref (only where /root/reference exists)."""
    srcs = [os.path.join(HERE, f) for f in ("oracle.c", "voxelizer_oracle.c", "mesher_oracle.c")]
    if os.environ.get("LFA_ORACLE_SO"):
        pass  # an externally built variant: used as it is
    elif force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-s", "-C", HERE, os.path.join(HERE, "liboracle.so")])
    ref_src = os.path.join(HERE, "ref_harness.cpp")
    if os.path.isdir(os.environ.get("REFERENCE_DIR", "/root/reference")):
        if force or not os.path.exists(REF_SO) or os.path.getmtime(REF_SO) < os.path.getmtime(ref_src):
            subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
        # the caller-shaped host program (tests/callers/reference_callers.cpp): against the real reference, and - once
        # libfluid_amd.so exists - against the shim headers in front of the reference's headers (runs on the GPU box)
        root = os.path.dirname(HERE)
        deps = [os.path.join(root, "tests", "callers", "reference_callers.cpp"), os.path.join(HERE, "ref_callers_main.cpp")]
        host = os.path.join(root, "libfluid_amd", "host")
        deps += [os.path.join(d, f) for d, _, fs in os.walk(host) for f in fs]
        lib = os.path.join(root, "libfluid_amd", "libfluid_amd.so")
        outs = [os.path.join(HERE, "_ref", "callers_ref")] + ([os.path.join(HERE, "_ref", "callers_dev_reftypes")] if os.path.exists(lib) else [])
        newest = max(os.path.getmtime(f) for f in deps if os.path.exists(f))
        if force or any(not os.path.exists(o) or os.path.getmtime(o) < newest for o in outs):
            subprocess.check_call(["make", "-s", "-C", HERE, "callers"])


def have_ref():
    return os.path.exists(REF_SO)


_vp, _sz, _dbl, _int = C.c_void_p, C.c_size_t, C.c_double, C.c_int


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _Lib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        sig = {
            "create": (_vp, [_sz, _sz, _sz, _dbl, _vp, _vp, _int, _dbl, _dbl]),
            "destroy": (None, [_vp]),
            "set_extrapolation_iterations": (None, [_vp, _sz]),
            "set_pcg_params": (None, [_vp, _dbl, _dbl, _dbl, _sz]),
            "set_solid_cells": (None, [_vp, _vp, _sz]),
            "set_particles": (None, [_vp, _vp, _sz]),
            "num_particles": (_sz, [_vp]),
            "get_particles": (None, [_vp, _vp]),
            "get_cells": (None, [_vp, _vp]),
            "set_cells": (None, [_vp, _vp]),
            "get_old_cells": (None, [_vp, _vp]),
            "hash": (None, [_vp]),
            "num_fluid_cells": (_sz, [_vp]),
            "get_fluid_cells": (None, [_vp, _vp]),
            "get_space_hash": (None, [_vp, _vp, _vp]),
            "p2g": (None, [_vp]),
            "add_gravity": (None, [_vp, _dbl]),
            "build_system": (None, [_vp, _dbl]),
            "get_abits": (None, [_vp, _vp]),
            "get_b": (None, [_vp, _vp]),
            "get_precon": (None, [_vp, _vp]),
            "apply_precon": (None, [_vp, _vp, _vp]),
            "apply_a": (None, [_vp, _vp, _vp]),
            "solve": (None, [_vp, _dbl, _vp, _vp, _vp]),
            "apply_pressure": (None, [_vp, _dbl, _vp]),
            "extrapolate": (None, [_vp]),
            "g2p": (None, [_vp]),
            "cfl": (_dbl, [_vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(self.lib, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        for name, (res, args) in {
            "hot_step": (None, [_vp, _dbl, _vp, _vp, _vp]),
            "advect": (None, [_vp, _dbl]),
            "detect_collisions": (None, [_vp]),
            "correct_positions": (None, [_vp, _dbl]),
            "time_step": (None, [_vp, _dbl, _vp, _vp]),
            "clear_sources": (None, [_vp]),
            "add_source": (None, [_vp, _vp, _sz, _vp, _sz, _int, _int]),
            "update_sources": (None, [_vp]),
            "update": (_sz, [_vp, _dbl, _vp, _sz]),
        }.items():
            fn = getattr(self.lib, prefix + name, None)
            if fn is not None:
                fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


_libs = {}


def _get(kind):
    if kind not in _libs:
        build()
        _libs[kind] = _Lib(ORACLE_SO, "orc_") if kind == "oracle" else _Lib(REF_SO, "ref_")
    return _libs[kind]


class CpuSim:
    """Stage-level driver for the oracle (kind='oracle') or the real reference (kind='ref')."""

    def __init__(self, size, cell_size=1.0, offset=(0.0, 0.0, 0.0), gravity=(0.0, -981.0, 0.0), method=APIC,
                 blending=1.0, density=1.0, kind="oracle"):
        self.L = _get(kind)
        self.size = tuple(int(s) for s in size)
        self.ncells = self.size[0] * self.size[1] * self.size[2]
        off = np.asarray(offset, dtype=np.float64)
        g = np.asarray(gravity, dtype=np.float64)
        self.h = C.c_void_p(self.L.create(*self.size, float(cell_size), _ptr(off), _ptr(g), int(method),
                                          float(blending), float(density)))
        self._built = False

    def close(self):
        if self.h:
            self.L.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ---------------------------------------------------------------------------------------------
    def set_solid_cells(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        self.L.set_solid_cells(self.h, _ptr(xyz), xyz.shape[0])

    def set_particles(self, parts):
        parts = np.ascontiguousarray(parts, dtype=PARTICLE_DTYPE)
        self.L.set_particles(self.h, _ptr(parts), parts.shape[0])

    def particles(self):
        out = np.empty(self.L.num_particles(self.h), dtype=PARTICLE_DTYPE)
        self.L.get_particles(self.h, _ptr(out))
        return out

    def cells(self):
        out = np.empty(self.ncells, dtype=CELL_DTYPE)
        self.L.get_cells(self.h, _ptr(out))
        return out

    def set_cells(self, cells):
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        assert cells.shape[0] == self.ncells
        self.L.set_cells(self.h, _ptr(cells))

    def old_cells(self):
        out = np.empty(self.ncells, dtype=CELL_DTYPE)
        self.L.get_old_cells(self.h, _ptr(out))
        return out

    def set_extrapolation_iterations(self, n):
        self.L.set_extrapolation_iterations(self.h, int(n))

    def set_pcg_params(self, tau=0.97, sigma=0.25, tol=1e-6, maxit=200):
        self.L.set_pcg_params(self.h, tau, sigma, tol, maxit)

    # -- stages ------------------------------------------------------------------------------------------
    def hash(self):
        self.L.hash(self.h)

    def fluid_cells(self):
        out = np.empty(self.L.num_fluid_cells(self.h), dtype=np.uint64)
        self.L.get_fluid_cells(self.h, _ptr(out))
        return out

    def space_hash(self):
        b = np.empty(self.ncells, dtype=np.uint64)
        c = np.empty(self.ncells, dtype=np.uint64)
        self.L.get_space_hash(self.h, _ptr(b), _ptr(c))
        return b, c

    def p2g(self):
        self.L.p2g(self.h)

    def add_gravity(self, dt):
        self.L.add_gravity(self.h, dt)

    def build_system(self, dt):
        self.L.build_system(self.h, dt)
        self._n = self.L.num_fluid_cells(self.h)

    def abits(self):
        out = np.empty(self._n, dtype=np.uint8)
        self.L.get_abits(self.h, _ptr(out))
        return out

    def b(self):
        out = np.empty(self._n, dtype=np.float64)
        self.L.get_b(self.h, _ptr(out))
        return out

    def precon(self):
        out = np.empty(self._n, dtype=np.float64)
        self.L.get_precon(self.h, _ptr(out))
        return out

    def apply_precon(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        self.L.apply_precon(self.h, _ptr(r), _ptr(z))
        return z

    def apply_a(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros_like(v)
        self.L.apply_a(self.h, _ptr(v), _ptr(out))
        return out

    def solve(self, dt):
        n = self.L.num_fluid_cells(self.h)
        self._n = n
        p = np.zeros(max(n, 1), dtype=np.float64)
        res = C.c_double(0.0)
        it = C.c_uint64(0)
        self.L.solve(self.h, dt, _ptr(p), C.byref(res), C.byref(it))
        return p[:n], res.value, it.value

    def apply_pressure(self, dt, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.L.apply_pressure(self.h, dt, _ptr(p))

    def extrapolate(self):
        self.L.extrapolate(self.h)

    def g2p(self):
        self.L.g2p(self.h)

    def cfl(self):
        return self.L.cfl(self.h)

    # -- fluid sources, update() -------------------------------------------------------------------------------
    def clear_sources(self):
        self.L.clear_sources(self.h)

    def add_source(self, cells, velocity=(0.0, 0.0, 0.0), density_cubic_root=2, active=True, coerce_velocity=False):
        xyz = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        vel = np.asarray(velocity, dtype=np.float64)
        self.L.add_source(self.h, _ptr(xyz), xyz.shape[0], _ptr(vel), int(density_cubic_root), int(active), int(coerce_velocity))

    def update_sources(self):
        """_update_sources + hash_particles (src/simulation.cpp:63-64)."""
        self.L.update_sources(self.h)

    def update(self, dt, cap=4096):
        """simulation::update(dt): returns the lengths of the CFL sub-steps taken."""
        dts = np.zeros(cap, dtype=np.float64)
        n = self.L.update(self.h, float(dt), _ptr(dts), cap)
        return dts[:min(n, cap)]

    def hot_step(self, dt):
        """hash -> p2g -> gravity -> solve -> apply -> extrapolate -> g2p; returns (p, residual, iters)."""
        if self.L.hot_step is not None:
            cap = max(min(self.L.num_particles(self.h), self.ncells), 1)
            p = np.zeros(cap, dtype=np.float64)
            res = C.c_double(0.0)
            it = C.c_uint64(0)
            self.L.hot_step(self.h, dt, _ptr(p), C.byref(res), C.byref(it))
            return p[:self.L.num_fluid_cells(self.h)], res.value, it.value
        self.hash()
        self.p2g()
        self.add_gravity(dt)
        self.build_system(dt)
        p, res, it = self.solve(dt)
        self.apply_pressure(dt, p)
        self.extrapolate()
        self.g2p()
        return p, res, it


# ---- voxelizer ---------------------------------------------------------------------------------
def voxelize(positions, indices, cell_size=1.0, ref_grid_offset=(0.0, 0.0, 0.0), kind="oracle"):
    """Runs get_bounding_box + resize_reposition_grid_constrained + voxelize_mesh_surface + mark_exterior on the oracle
    (kind='oracle') or the real reference (kind='ref'). Returns (grid_min int32[3], grid_offset float64[3],
    types uint8[nz, ny, nx])."""
    L = _get(kind).lib
    pre = "orc_" if kind == "oracle" else "ref_"
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint64).reshape(-1)
    off = np.asarray(ref_grid_offset, dtype=np.float64)
    gmin, size, goff = np.zeros(3, np.int32), np.zeros(3, np.uint64), np.zeros(3, np.float64)
    grid = getattr(L, pre + "vox_grid")
    grid.restype, grid.argtypes = None, [_vp, _sz, _dbl, _vp, _vp, _vp, _vp]
    grid(_ptr(pos), pos.shape[0], float(cell_size), _ptr(off), _ptr(gmin), _ptr(size), _ptr(goff))
    types = np.zeros(int(size[0] * size[1] * size[2]), dtype=np.uint8)
    vox = getattr(L, pre + "voxelize")
    vox.restype, vox.argtypes = None, [_vp, _sz, _vp, _sz, _dbl, _vp, _vp]
    vox(_ptr(pos), pos.shape[0], _ptr(idx), idx.size, float(cell_size), _ptr(off), _ptr(types))
    return gmin, goff, types.reshape(int(size[2]), int(size[1]), int(size[0]))


def ref_voxel_cells(positions, indices, cell_size, ref_grid_offset, include_interior, include_surface, ref_grid_size=None):
    """Cell lists of the Maya VoxelizerNode computed with the real reference's voxelizer and grid3::for_each
    (plugins/maya/nodes/voxelizer_node.cpp:285-343) as int32[k,3]; ref_grid_size=None: voxel-grid coordinates."""
    L = _get("ref").lib
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint64).reshape(-1)
    off = np.asarray(ref_grid_offset, dtype=np.float64)
    rs = None if ref_grid_size is None else np.asarray(ref_grid_size, dtype=np.int64)
    rsp = None if rs is None else _ptr(rs)
    fn = L.ref_voxel_cells
    fn.restype, fn.argtypes = _sz, [_vp, _sz, _vp, _sz, _dbl, _vp, _int, _int, _vp, _vp, _sz]
    args = (_ptr(pos), pos.shape[0], _ptr(idx), idx.size, float(cell_size), _ptr(off), int(include_interior),
            int(include_surface), rsp)
    n = fn(*args, None, 0)
    out = np.zeros((n, 3), dtype=np.int32)
    fn(*args, _ptr(out), n)
    return out


# ---- on-disk formats of the real reference; kind='ref' only -----------------------------------------
def ref_points_text(points):
    L = _get("ref").lib
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    L.ref_points_text.restype, L.ref_points_text.argtypes = _sz, [_vp, _sz, _vp, _sz]
    n = L.ref_points_text(_ptr(pts), pts.shape[0], None, 0)
    buf = C.create_string_buffer(n)
    L.ref_points_text(_ptr(pts), pts.shape[0], buf, n)
    return buf.raw[:n]


def ref_points_parse(text, count):
    L = _get("ref").lib
    L.ref_points_parse.restype, L.ref_points_parse.argtypes = _sz, [C.c_char_p, _sz, _sz, _vp, _sz]
    cap = text.count(b"\n") + 2
    out = np.zeros((cap, 3), dtype=np.float64)
    n = L.ref_points_parse(text, len(text), int(count), _ptr(out), cap)
    return out[:n]


def ref_mesh_obj(positions, indices, uvs=None, normals=False, reverse=False):
    """(obj text, normals float64[nv,3] or None) from mesh::save_obj / generate_normals of the real reference."""
    L = _get("ref").lib
    pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
    idx = np.ascontiguousarray(indices, dtype=np.uint64).reshape(-1)
    uv = None if uvs is None else np.ascontiguousarray(uvs, dtype=np.float64).reshape(-1, 2)
    nrm = np.zeros_like(pos) if normals else None
    L.ref_mesh_obj.restype = _sz
    L.ref_mesh_obj.argtypes = [_vp, _sz, _vp, _sz, _vp, _int, _int, _vp, _sz, _vp]
    args = (_ptr(pos), pos.shape[0], _ptr(idx), idx.size, None if uv is None else _ptr(uv), int(normals), int(reverse))
    n = L.ref_mesh_obj(*args, None, 0, None)
    buf = C.create_string_buffer(n)
    L.ref_mesh_obj(*args, buf, n, None if nrm is None else _ptr(nrm))
    return buf.raw[:n], nrm


# ---- surface mesher ----------------------------------------------------------------------------
def _mesher_args(points, size, grid_offset, cell_size, particle_extent, cell_radius, r):
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    sz = np.asarray(size, dtype=np.uint64)
    off = np.asarray(grid_offset, dtype=np.float64)
    return pts, sz, off, (_ptr(pts), pts.shape[0], _ptr(sz), _ptr(off), float(cell_size), float(particle_extent),
                          int(cell_radius), float(r))


def mesher_surface(points, size, grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=0.5, cell_radius=2, r=0.5,
                   kind="oracle"):
    """mesher::_sample_surface_function: float64[nz+1, ny+1, nx+1] values at the grid points."""
    L = _get(kind).lib
    fn = getattr(L, ("orc_" if kind == "oracle" else "ref_") + "mesher_surface")
    fn.restype, fn.argtypes = None, [_vp, _sz, _vp, _vp, _dbl, _dbl, C.c_uint64, _dbl, _vp]
    pts, sz, off, args = _mesher_args(points, size, grid_offset, cell_size, particle_extent, cell_radius, r)
    out = np.zeros((int(sz[2]) + 1, int(sz[1]) + 1, int(sz[0]) + 1), dtype=np.float64)
    fn(*args, _ptr(out))
    return out


def mesher_mesh(points, size, grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=0.5, cell_radius=2, r=0.5,
                values=None, kind="oracle"):
    """mesher::generate_mesh, or mesher::_marching_cubes on given grid-point values. Returns (positions float64[nv,3],
    indices uint64[ni])."""
    L = _get(kind).lib
    fn = getattr(L, ("orc_" if kind == "oracle" else "ref_") + "mesher_mesh")
    fn.restype = None
    fn.argtypes = [_vp, _sz, _vp, _vp, _dbl, _dbl, C.c_uint64, _dbl, _vp, _vp, _sz, _vp, _sz, _vp]
    pts, sz, off, args = _mesher_args(points if points is not None else np.zeros((0, 3)), size, grid_offset, cell_size,
                                      particle_extent, cell_radius, r)
    vals = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    vp = None if vals is None else _ptr(vals)
    counts = np.zeros(2, dtype=np.uint64)
    fn(*args, vp, None, 0, None, 0, _ptr(counts))
    pos, idx = np.zeros((int(counts[0]), 3), dtype=np.float64), np.zeros(int(counts[1]), dtype=np.uint64)
    fn(*args, vp, _ptr(pos), pos.shape[0], _ptr(idx), idx.size, _ptr(counts))
    return pos, idx
