"""libfluid_amd -- MI355X-native PIC/FLIP/APIC hot path behind libfluid's ``fluid::simulation`` step API.

The product is the C-ABI shared library ``libfluid_amd.so`` (include/libfluid_amd.h, kernels in csrc/) plus the C++17 host
class in host/simulation.h. This module is the thin ctypes binding the tests and bench.py use to call that C ABI; it
adds no computation of its own and has no CPU fallback: loading fails loudly when the HIP library is missing.
"""
import ctypes as C
import os

import numpy as np

from . import scenes  # noqa: F401
from .scenes import CELL_DTYPE, PARTICLE_DTYPE  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfluid_amd.so")

PIC, FLIP_BLEND, APIC = 0, 1, 2
P2G_LDS_BINNED, P2G_GLOBAL_ATOMIC = 0, 1
PRECOND_MIC0_TILED, PRECOND_MIC0_EXACT, PRECOND_MULTILEVEL, PRECOND_MULTIGRID = 0, 1, 2, 3
PCG_F32, PCG_F64 = 0, 1
OK, W_PCG_NOT_CONVERGED = 0, 1
NUM_TIMERS = 10
NUM_STEP_TIMERS = 16
STEP_TIMER_NAMES = ["advect_collide", "bin", "p2g", "p2g_scatter_kernel", "build_system", "pcg_loop", "apply_pressure",
                    "correct_cell_index", "correct_tiled_kernel", "correct_collide", "extrapolate", "g2p", "time_step",
                    "pcg_iterations", "pcg_iteration_mean", "overlapped"]
TIMER_NAMES = ["bin", "p2g", "gravity", "build_system", "pcg_loop", "apply_pressure", "extrapolate", "g2p",
               "p2g_scatter_kernel", "pcg_iteration_mean"]


class Params(C.Structure):
    """lfa_params (include/libfluid_amd.h) == public fields of fluid::simulation / fluid::pressure_solver."""
    _fields_ = [
        ("grid_offset", C.c_double * 3),
        ("gravity", C.c_double * 3),
        ("cell_size", C.c_double),
        ("blending_factor", C.c_double),
        ("density", C.c_double),
        ("boundary_skin_width", C.c_double),
        ("correction_stiffness", C.c_double),
        ("cfl_number", C.c_double),
        ("velocity_extrapolation_iterations", C.c_uint64),
        ("simulation_method", C.c_int32),
        ("tau", C.c_double),
        ("sigma", C.c_double),
        ("tolerance", C.c_double),
        ("max_iterations", C.c_uint64),
        ("p2g_variant", C.c_int32),
        ("precond", C.c_int32),
        ("pcg_dtype", C.c_int32),
        ("apic_unscaled_kernel", C.c_int32),
        ("pcg_fused", C.c_int32),
        ("pcg_warm_start", C.c_int32),
    ]


class LibfluidError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libfluid_amd error {code}: {msg}")
        self.code = code


_lib = None

# name -> (restype, argtypes); also the list the CPU test checks against include/libfluid_amd.h
_vp, _u64, _dbl, _int = C.c_void_p, C.c_uint64, C.c_double, C.c_int
SIGNATURES = {
    "lfa_default_params": (None, [C.POINTER(Params)]),
    "lfa_create": (_int, [C.POINTER(_vp), _u64, _u64, _u64, _int]),
    "lfa_destroy": (None, [_vp]),
    "lfa_last_error": (C.c_char_p, [_vp]),
    "lfa_set_params": (_int, [_vp, C.POINTER(Params)]),
    "lfa_get_params": (_int, [_vp, C.POINTER(Params)]),
    "lfa_synchronize": (_int, [_vp]),
    "lfa_stream": (_vp, [_vp]),
    "lfa_upload_particles": (_int, [_vp, _vp, _u64]),
    "lfa_download_particles": (_int, [_vp, _vp, _u64, _int]),
    "lfa_download_particle_ids": (_int, [_vp, _vp, _u64]),
    "lfa_num_particles": (_u64, [_vp]),
    "lfa_seed_block": (_int, [_vp, _vp, _vp, _u64]),
    "lfa_set_solid_cells": (_int, [_vp, _vp, _u64]),
    "lfa_clear_solid_cells": (_int, [_vp]),
    "lfa_upload_cells": (_int, [_vp, _vp]),
    "lfa_download_cells": (_int, [_vp, _vp]),
    "lfa_download_old_cells": (_int, [_vp, _vp]),
    "lfa_hash_particles": (_int, [_vp]),
    "lfa_num_fluid_cells": (_u64, [_vp]),
    "lfa_download_fluid_cells": (_int, [_vp, _vp, _u64]),
    "lfa_download_cell_counts": (_int, [_vp, _vp]),
    "lfa_p2g": (_int, [_vp]),
    "lfa_add_gravity": (_int, [_vp, _dbl]),
    "lfa_build_system": (_int, [_vp, _dbl]),
    "lfa_download_abits": (_int, [_vp, _vp, _u64]),
    "lfa_download_rhs": (_int, [_vp, _vp, _u64]),
    "lfa_download_precon": (_int, [_vp, _vp, _u64]),
    "lfa_apply_preconditioner": (_int, [_vp, _vp, _vp, _u64]),
    "lfa_apply_a": (_int, [_vp, _vp, _vp, _u64]),
    "lfa_pcg_solve": (_int, [_vp, _dbl, C.POINTER(_dbl), C.POINTER(_u64)]),
    "lfa_get_solver_stats": (_int, [_vp, _vp]),
    "lfa_get_mg_level_tiles": (_int, [_vp, _vp]),
    "lfa_download_pressure": (_int, [_vp, _vp, _u64]),
    "lfa_upload_pressure": (_int, [_vp, _vp, _u64]),
    "lfa_apply_pressure": (_int, [_vp, _dbl]),
    "lfa_extrapolate": (_int, [_vp]),
    "lfa_g2p": (_int, [_vp]),
    "lfa_cfl": (_int, [_vp, C.POINTER(_dbl)]),
    "lfa_step_hot": (_int, [_vp, _dbl, C.POINTER(_dbl), C.POINTER(_u64)]),
    "lfa_enable_timing": (_int, [_vp, _int]),
    "lfa_get_timings": (_int, [_vp, C.POINTER(_dbl * NUM_TIMERS)]),
    "lfa_get_counts": (_int, [_vp, C.POINTER(_u64 * 5)]),
    "lfa_bench_kernel": (_int, [_vp, _int, _int, C.POINTER(_dbl)]),
    "lfa_bench_stream": (_int, [_vp, _u64, _int, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "lfa_bench_stream_variant": (C.c_char_p, []),
    "lfa_pool_trim": (None, []),
    "lfa_pool_stats": (None, [_vp]),
    "lfa_voxels_create": (_int, [C.POINTER(_vp), _vp, _vp, _dbl, _int]),
    "lfa_voxels_destroy": (None, [_vp]),
    "lfa_voxels_last_error": (C.c_char_p, [_vp]),
    "lfa_voxels_info": (_int, [_vp, _vp, _vp, _vp, C.POINTER(_dbl)]),
    "lfa_voxels_upload": (_int, [_vp, _vp]),
    "lfa_voxels_download": (_int, [_vp, _vp]),
    "lfa_voxels_voxelize_triangles": (_int, [_vp, _vp, _u64, _vp, _int, _u64]),
    "lfa_voxels_mark_exterior": (_int, [_vp]),
    "lfa_voxelize_mesh": (_int, [C.POINTER(_vp), _vp, _u64, _vp, _int, _u64, _dbl, _vp, _int]),
    "lfa_voxels_count": (_int, [_vp, _int, _int, _vp, C.POINTER(_u64)]),
    "lfa_voxels_cells": (_int, [_vp, _int, _int, _vp, _vp, _u64, C.POINTER(_u64)]),
    "lfa_set_solid_from_voxels": (_int, [_vp, _vp, _int, _int]),
    "lfa_mesher_create": (_int, [C.POINTER(_vp), _vp, _vp, _dbl, _dbl, _u64, _int]),
    "lfa_mesher_create_window": (_int, [C.POINTER(_vp), _vp, _vp, _dbl, _dbl, _u64, _u64, _u64, _int]),
    "lfa_mesher_window": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    "lfa_mesher_sample_ids": (_int, [_vp, _vp, _vp, _u64, _dbl]),
    "lfa_mesher_rebase": (_int, [_vp, _u64]),
    "lfa_mesher_destroy": (None, [_vp]),
    "lfa_mesher_last_error": (C.c_char_p, [_vp]),
    "lfa_mesher_sample": (_int, [_vp, _vp, _u64, _dbl]),
    "lfa_mesher_sample_sim": (_int, [_vp, _vp, _dbl]),
    "lfa_mesher_download_values": (_int, [_vp, _vp]),
    "lfa_mesher_upload_values": (_int, [_vp, _vp]),
    "lfa_mesher_marching_cubes": (_int, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "lfa_mesher_download_mesh": (_int, [_vp, _vp, _vp]),
    "lfa_clear_sources": (_int, [_vp]),
    "lfa_add_source": (_int, [_vp, _vp, _u64, _vp, _u64, _int, _int]),
    "lfa_update_sources": (_int, [_vp, C.POINTER(_u64)]),
    "lfa_advect_collide": (_int, [_vp, _dbl]),
    "lfa_advect": (_int, [_vp, _dbl]),
    "lfa_correct": (_int, [_vp, _dbl]),
    "lfa_collide": (_int, [_vp]),
    "lfa_correct_collide": (_int, [_vp, _dbl]),
    "lfa_time_step": (_int, [_vp, _dbl, C.POINTER(_dbl), C.POINTER(_u64)]),
    "lfa_get_step_timings": (_int, [_vp, C.POINTER(_dbl * NUM_STEP_TIMERS)]),
    "lfa_set_step_overlap": (_int, [_vp, _int]),
    "lfa_get_correction_stats": (_int, [_vp, C.POINTER(_u64 * 2)]),
    "lfa_get_correction_stats_ex": (_int, [_vp, C.POINTER(_u64 * 3)]),
    "lfa_correct_collide_begin": (_int, [_vp, _dbl]),
    "lfa_correct_collide_end": (_int, [_vp]),
    "lfa_correct_collide_undo": (_int, [_vp]),
    "lfa_dist_unique_id": (_int, [_vp]),
    "lfa_dist_init_rccl": (_int, [_vp, _int, _int, _vp, _vp]),
    "lfa_dist_local_hub_create": (_vp, [_int]),
    "lfa_dist_local_hub_destroy": (None, [_vp]),
    "lfa_dist_init_local": (_int, [_vp, _vp, _int, _vp]),
    "lfa_dist_init_shm": (_int, [_vp, C.c_char_p, _int, _int, _vp]),
    "lfa_dist_get_slab": (_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lfa_dist_abandon": (_int, [_vp]),
}
KERNELS = {"spmv_dot": 0, "axpy_max": 1, "mic_apply_dot": 2, "update_s": 3, "p2g_scatter": 4, "p2g_finalize": 5,
           "g2p": 6, "bin": 7, "mic_fine": 8, "coarse_levels": 9, "pcg_a": 10, "pcg_b": 11,
           "mg_axpy_presmooth": 12, "mg_down0": 13, "mg_coarse": 14, "mg_up0": 15}


def load_library():
    """Loads libfluid_amd.so and binds every entry point of include/libfluid_amd.h. Raises if the library is missing."""
    global _lib
    if _lib is None:
        path = os.environ.get("LFA_LIB_PATH") or LIB_PATH  # (A/B measurements: a variant build, libfluid_amd/build.py build_variant)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the hot path)")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def rccl_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the caller broadcasts it to the other ranks)."""
    buf = np.zeros(128, dtype=np.uint8)
    rc = load_library().lfa_dist_unique_id(_ptr(buf))
    if rc != 0:
        raise LibfluidError(rc, load_library().lfa_last_error(None).decode())
    return buf.tobytes()


def balanced_layer_bounds(ntz, nranks, lo_layer=0, hi_layer=None):
    """Splits the tile layers [lo_layer, hi_layer) that hold fluid evenly over the ranks; the empty layers below/above go
    to the first/last rank. Returns nranks+1 bounds partitioning [0, ntz)."""
    hi_layer = ntz if hi_layer is None else hi_layer
    span = max(hi_layer - lo_layer, nranks)
    hi_layer = min(lo_layer + span, ntz)
    lo_layer = max(hi_layer - span, 0)
    b = [lo_layer + (span * r) // nranks for r in range(nranks + 1)]
    b[0], b[-1] = 0, ntz
    return b


def pool_trim():
    """Releases the process-wide cache of device blocks / streams of destroyed handles (lfa_pool_trim)."""
    load_library().lfa_pool_trim()


def pool_stats():
    arr = (C.c_uint64 * 4)()
    load_library().lfa_pool_stats(C.byref(arr))
    return dict(zip(["cached_bytes", "cached_blocks", "hits", "misses"], list(arr)))


class LocalHub:
    """In-process transport between the handles of several "virtual slabs" (one host thread per handle)."""

    def __init__(self, nranks):
        self.lib = load_library()
        self.h = C.c_void_p(self.lib.lfa_dist_local_hub_create(int(nranks)))

    def close(self):
        if self.h:
            self.lib.lfa_dist_local_hub_destroy(self.h)
            self.h = None


VOX_INTERIOR, VOX_EXTERIOR, VOX_SURFACE = 0, 1, 2


class Voxels:
    """Device voxel grid of the solid-boundary voxelizer (lfa_voxels): mirrors fluid::voxelizer's call sequence."""

    def __init__(self, handle):
        self.lib = load_library()
        self.h = handle
        gmin, size, off, cs = np.zeros(3, np.int32), np.zeros(3, np.uint64), np.zeros(3, np.float64), _dbl()
        self._chk(self.lib.lfa_voxels_info(self.h, _ptr(gmin), _ptr(size), _ptr(off), C.byref(cs)))
        self.grid_min, self.size, self.grid_offset, self.cell_size = gmin, tuple(int(x) for x in size), off, cs.value

    def _chk(self, rc):
        if rc < 0:
            raise LibfluidError(rc, self.lib.lfa_voxels_last_error(self.h).decode())
        return rc

    @classmethod
    def create(cls, size, grid_offset, cell_size=1.0, device=-1):
        lib = load_library()
        h = C.c_void_p()
        sz, off = np.asarray(size, dtype=np.uint64), np.asarray(grid_offset, dtype=np.float64)
        rc = lib.lfa_voxels_create(C.byref(h), _ptr(sz), _ptr(off), float(cell_size), int(device))
        if rc != 0:
            raise LibfluidError(rc, lib.lfa_last_error(None).decode())
        return cls(h)

    @classmethod
    def from_mesh(cls, positions, indices, cell_size=1.0, ref_grid_offset=(0.0, 0.0, 0.0), device=-1):
        """get_bounding_box + resize_reposition_grid_constrained + voxelize_mesh_surface + mark_exterior."""
        lib = load_library()
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        idx = np.ascontiguousarray(indices)
        if idx.dtype not in (np.uint32, np.uint64):
            idx = idx.astype(np.uint64)
        off = np.asarray(ref_grid_offset, dtype=np.float64)
        h = C.c_void_p()
        rc = lib.lfa_voxelize_mesh(C.byref(h), _ptr(pos), pos.shape[0], _ptr(idx), idx.dtype.itemsize, idx.size,
                                   float(cell_size), _ptr(off), int(device))
        if rc != 0:
            raise LibfluidError(rc, lib.lfa_last_error(None).decode())
        return cls(h)

    def voxelize_triangles(self, positions, indices):
        pos = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        idx = np.ascontiguousarray(indices)
        if idx.dtype not in (np.uint32, np.uint64):
            idx = idx.astype(np.uint64)
        self._chk(self.lib.lfa_voxels_voxelize_triangles(self.h, _ptr(pos), pos.shape[0], _ptr(idx), idx.dtype.itemsize,
                                                         idx.size))

    def mark_exterior(self):
        self._chk(self.lib.lfa_voxels_mark_exterior(self.h))

    def upload(self, types):
        t = np.ascontiguousarray(types, dtype=np.uint8).reshape(-1)
        assert t.size == self.size[0] * self.size[1] * self.size[2]
        self._chk(self.lib.lfa_voxels_upload(self.h, _ptr(t)))

    def types(self):
        """uint8[nz, ny, nx] (x fastest, like grid3<cell_type>)."""
        out = np.empty(self.size[0] * self.size[1] * self.size[2], dtype=np.uint8)
        self._chk(self.lib.lfa_voxels_download(self.h, _ptr(out)))
        return out.reshape(self.size[2], self.size[1], self.size[0])

    def cells(self, include_interior=True, include_surface=False, ref_grid_size=None):
        """int32[k,3]: voxel-grid coordinates, or reference-grid coordinates clipped to `ref_grid_size`."""
        ref = None if ref_grid_size is None else np.asarray(ref_grid_size, dtype=np.int64)
        refp = None if ref is None else _ptr(ref)
        n = _u64()
        self._chk(self.lib.lfa_voxels_count(self.h, int(include_interior), int(include_surface), refp, C.byref(n)))
        out = np.empty((n.value, 3), dtype=np.int32)
        self._chk(self.lib.lfa_voxels_cells(self.h, int(include_interior), int(include_surface), refp, _ptr(out), n.value,
                                            C.byref(n)))
        return out

    def close(self):
        if self.h:
            self.lib.lfa_voxels_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mesher:
    """Device surface mesher (lfa_mesher): mirrors fluid::mesher (resize + public fields, generate_mesh)."""

    def __init__(self, size, grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=0.5, cell_radius=2, device=-1,
                 window=None):
        """window = (zlo, zhi): only the cell layers [zlo, zhi) of the grid (lfa_mesher_create_window)."""
        self.lib = load_library()
        self.size = tuple(int(x) for x in size)
        sz, off = np.asarray(self.size, dtype=np.uint64), np.asarray(grid_offset, dtype=np.float64)
        h = C.c_void_p()
        zlo, zhi = (0, self.size[2]) if window is None else window
        rc = self.lib.lfa_mesher_create_window(C.byref(h), _ptr(sz), _ptr(off), float(cell_size), float(particle_extent),
                                               int(cell_radius), int(zlo), int(zhi), int(device))
        if rc != 0:
            raise LibfluidError(rc, self.lib.lfa_last_error(None).decode())
        self.h = h
        z0, npl, lo, hi = _u64(), _u64(), _u64(), _u64()
        self._chk(self.lib.lfa_mesher_window(self.h, C.byref(z0), C.byref(npl), C.byref(lo), C.byref(hi)))
        self.z0, self.n_planes, self.own = z0.value, npl.value, (lo.value, hi.value)

    def _chk(self, rc):
        if rc < 0:
            raise LibfluidError(rc, self.lib.lfa_mesher_last_error(self.h).decode())
        return rc

    def sample(self, points, r, ids=None):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        if ids is None:
            self._chk(self.lib.lfa_mesher_sample(self.h, _ptr(pts), pts.shape[0], float(r)))
        else:
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            self._chk(self.lib.lfa_mesher_sample_ids(self.h, _ptr(pts), _ptr(ids), pts.shape[0], float(r)))

    def rebase(self, vertices_below):
        self._chk(self.lib.lfa_mesher_rebase(self.h, int(vertices_below)))

    def sample_sim(self, sim, r):
        """Samples from the particles resident in a Sim handle (no host copy)."""
        self._chk(self.lib.lfa_mesher_sample_sim(self.h, sim.h, float(r)))

    def values(self):
        """float64[stored planes, ny+1, nx+1] (the whole grid: nz + 1 planes; a window: planes z0 .. z0 + n_planes - 1)."""
        out = np.empty((self.n_planes, self.size[1] + 1, self.size[0] + 1), dtype=np.float64)
        self._chk(self.lib.lfa_mesher_download_values(self.h, _ptr(out)))
        return out

    def set_values(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.size == (self.size[0] + 1) * (self.size[1] + 1) * (self.size[2] + 1)
        self._chk(self.lib.lfa_mesher_upload_values(self.h, _ptr(v)))

    def marching_cubes(self):
        """(positions float64[nv,3], indices uint64[ni])"""
        nv, ni = _u64(), _u64()
        self._chk(self.lib.lfa_mesher_marching_cubes(self.h, C.byref(nv), C.byref(ni)))
        self._counts = (nv.value, ni.value)
        return self.download_mesh()

    def download_mesh(self):
        """The mesh of the last marching_cubes() (indices as lfa_mesher_rebase left them)."""
        nv, ni = self._counts
        pos, idx = np.empty((nv, 3), dtype=np.float64), np.empty(ni, dtype=np.uint64)
        self._chk(self.lib.lfa_mesher_download_mesh(self.h, _ptr(pos), _ptr(idx)))
        return pos, idx

    def generate_mesh(self, points, r):
        self.sample(points, r)
        return self.marching_cubes()

    def close(self):
        if self.h:
            self.lib.lfa_mesher_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_params():
    p = Params()
    load_library().lfa_default_params(C.byref(p))
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Sim:
    """Handle to a device-resident simulation; mirrors the stage-level C ABI one-to-one."""

    def __init__(self, size, cell_size=1.0, offset=(0.0, 0.0, 0.0), gravity=(0.0, -981.0, 0.0), method=APIC,
                 blending=1.0, density=1.0, device=-1, **extra):
        self.lib = load_library()
        self.size = tuple(int(s) for s in size)
        self.ncells = self.size[0] * self.size[1] * self.size[2]
        h = C.c_void_p()
        import time as _time
        t0 = _time.perf_counter()
        rc = self.lib.lfa_create(C.byref(h), *self.size, int(device))
        self.create_ms = 1e3 * (_time.perf_counter() - t0)  # wall time of lfa_create (tests of the handle cache read it)
        if rc != 0:
            raise LibfluidError(rc, self.lib.lfa_last_error(None).decode())
        self.h = h
        p = default_params()
        p.cell_size = float(cell_size)
        for k in range(3):
            p.grid_offset[k] = float(offset[k])
            p.gravity[k] = float(gravity[k])
        p.simulation_method = int(method)
        p.blending_factor = float(blending)
        p.density = float(density)
        self.params = p
        self.set_params(**extra)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lfa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise LibfluidError(rc, self.lib.lfa_last_error(self.h).decode())
        return rc

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        self._chk(self.lib.lfa_set_params(self.h, C.byref(self.params)))

    def synchronize(self):
        self._chk(self.lib.lfa_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.lfa_stream(self.h)

    # -- data ----------------------------------------------------------------------------------------------
    def upload_particles(self, parts):
        parts = np.ascontiguousarray(parts, dtype=PARTICLE_DTYPE)
        self._chk(self.lib.lfa_upload_particles(self.h, _ptr(parts), parts.shape[0]))

    def download_particles(self, into=None, write_positions=False):
        n = self.num_particles
        out = np.zeros(n, dtype=PARTICLE_DTYPE) if into is None else into
        self._chk(self.lib.lfa_download_particles(self.h, _ptr(out), n, 1 if (write_positions or into is None) else 0))
        return out

    def particle_ids(self):
        """Global id of each record of download_particles() (slab decomposition: particles migrate between ranks)."""
        n = self.num_particles
        ids = np.zeros(n, dtype=np.uint32)
        self._chk(self.lib.lfa_download_particle_ids(self.h, _ptr(ids), n))
        return ids

    @property
    def num_particles(self):
        return int(self.lib.lfa_num_particles(self.h))

    def seed_block(self, lo, hi, seed=scenes.SEED):
        lo = np.asarray(lo, dtype=np.int64)
        hi = np.asarray(hi, dtype=np.int64)
        self._chk(self.lib.lfa_seed_block(self.h, _ptr(lo), _ptr(hi), int(seed)))

    def set_solid_cells(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        self._chk(self.lib.lfa_set_solid_cells(self.h, _ptr(xyz), xyz.shape[0]))

    def clear_solid_cells(self):
        self._chk(self.lib.lfa_clear_solid_cells(self.h))

    def set_solid_from_voxels(self, voxels, include_interior=True, include_surface=False):
        """Marks the selected voxels of a Voxels grid solid, device to device (lfa_set_solid_from_voxels)."""
        self._chk(self.lib.lfa_set_solid_from_voxels(self.h, voxels.h, int(include_interior), int(include_surface)))

    def upload_cells(self, cells):
        cells = np.ascontiguousarray(cells, dtype=CELL_DTYPE)
        assert cells.shape[0] == self.ncells
        self._chk(self.lib.lfa_upload_cells(self.h, _ptr(cells)))

    def cells(self):
        out = np.zeros(self.ncells, dtype=CELL_DTYPE)
        self._chk(self.lib.lfa_download_cells(self.h, _ptr(out)))
        return out

    def old_cells(self):
        out = np.zeros(self.ncells, dtype=CELL_DTYPE)
        self._chk(self.lib.lfa_download_old_cells(self.h, _ptr(out)))
        return out

    # -- stages ------------------------------------------------------------------------------------------
    def hash(self):
        self._chk(self.lib.lfa_hash_particles(self.h))

    @property
    def num_fluid_cells(self):
        return int(self.lib.lfa_num_fluid_cells(self.h))

    def fluid_cells(self):
        n = self.num_fluid_cells
        out = np.zeros(n, dtype=np.uint64)
        self._chk(self.lib.lfa_download_fluid_cells(self.h, _ptr(out), n))
        return out

    def cell_counts(self):
        out = np.zeros(self.ncells, dtype=np.uint32)
        self._chk(self.lib.lfa_download_cell_counts(self.h, _ptr(out)))
        return out

    def p2g(self):
        self._chk(self.lib.lfa_p2g(self.h))

    def add_gravity(self, dt):
        self._chk(self.lib.lfa_add_gravity(self.h, float(dt)))

    def build_system(self, dt):
        self._chk(self.lib.lfa_build_system(self.h, float(dt)))

    def _vec(self, fn, dtype=np.float64):
        n = self.num_fluid_cells
        out = np.zeros(n, dtype=dtype)
        self._chk(fn(self.h, _ptr(out), n))
        return out

    def abits(self):
        return self._vec(self.lib.lfa_download_abits, np.uint8)

    def b(self):
        return self._vec(self.lib.lfa_download_rhs)

    def precon(self):
        return self._vec(self.lib.lfa_download_precon)

    def pressure(self):
        return self._vec(self.lib.lfa_download_pressure)

    def upload_pressure(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        self._chk(self.lib.lfa_upload_pressure(self.h, _ptr(p), p.shape[0]))

    def apply_precon(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        self._chk(self.lib.lfa_apply_preconditioner(self.h, _ptr(r), _ptr(z), r.shape[0]))
        return z

    def apply_a(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.zeros_like(v)
        self._chk(self.lib.lfa_apply_a(self.h, _ptr(v), _ptr(out), v.shape[0]))
        return out

    def solve(self, dt):
        """Returns (pressure in the reference's unknown order, residual, iterations, return code)."""
        res, it = C.c_double(0.0), C.c_uint64(0)
        rc = self._chk(self.lib.lfa_pcg_solve(self.h, float(dt), C.byref(res), C.byref(it)))
        return self.pressure(), res.value, it.value, rc

    def apply_pressure(self, dt):
        self._chk(self.lib.lfa_apply_pressure(self.h, float(dt)))

    def extrapolate(self):
        self._chk(self.lib.lfa_extrapolate(self.h))

    def g2p(self):
        self._chk(self.lib.lfa_g2p(self.h))

    def cfl(self):
        out = C.c_double(0.0)
        self._chk(self.lib.lfa_cfl(self.h, C.byref(out)))
        return out.value

    def step_hot(self, dt):
        """One device-resident pass of the hot path; returns (residual, iterations, return code)."""
        res, it = C.c_double(0.0), C.c_uint64(0)
        rc = self._chk(self.lib.lfa_step_hot(self.h, float(dt), C.byref(res), C.byref(it)))
        return res.value, it.value, rc

    # -- particle stages around the hot path, full step ---------------------------------------------------
    def clear_sources(self):
        self._chk(self.lib.lfa_clear_sources(self.h))

    def add_source(self, cells, velocity=(0.0, 0.0, 0.0), density_cubic_root=2, active=True, coerce_velocity=False):
        """fluid::source (data_structures/source.h:12-22): cells as int32[k,3]."""
        xyz = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 3)
        vel = np.asarray(velocity, dtype=np.float64)
        self._chk(self.lib.lfa_add_source(self.h, _ptr(xyz), xyz.shape[0], _ptr(vel), int(density_cubic_root), int(active),
                                          int(coerce_velocity)))

    def update_sources(self):
        """_update_sources + hash_particles (src/simulation.cpp:63-64); returns the number of particles created."""
        n = C.c_uint64(0)
        self._chk(self.lib.lfa_update_sources(self.h, C.byref(n)))
        return n.value

    def advect(self, dt):
        self._chk(self.lib.lfa_advect(self.h, float(dt)))

    def correct(self, dt):
        self._chk(self.lib.lfa_correct(self.h, float(dt)))

    def collide(self):
        self._chk(self.lib.lfa_collide(self.h))

    def advect_collide(self, dt):
        self._chk(self.lib.lfa_advect_collide(self.h, float(dt)))

    def correct_collide(self, dt):
        self._chk(self.lib.lfa_correct_collide(self.h, float(dt)))

    def correct_collide_begin(self, dt):
        """lfa_correct_collide on the second stream; grid-only stages may run beside it until correct_collide_end."""
        self._chk(self.lib.lfa_correct_collide_begin(self.h, float(dt)))

    def correct_collide_end(self):
        self._chk(self.lib.lfa_correct_collide_end(self.h))

    def correct_collide_undo(self):
        self._chk(self.lib.lfa_correct_collide_undo(self.h))

    def time_step(self, dt):
        """Device-resident simulation::time_step(dt); returns (residual, iterations, return code)."""
        res, it = C.c_double(0.0), C.c_uint64(0)
        rc = self._chk(self.lib.lfa_time_step(self.h, float(dt), C.byref(res), C.byref(it)))
        return res.value, it.value, rc

    def step_timings(self):
        arr = (C.c_double * NUM_STEP_TIMERS)()
        self._chk(self.lib.lfa_get_step_timings(self.h, C.byref(arr)))
        return dict(zip(STEP_TIMER_NAMES, list(arr)))

    def correction_stats(self):
        """(half tiles the last position correction handed to its slow fallback, half tiles in all)."""
        arr = (C.c_uint64 * 2)()
        self._chk(self.lib.lfa_get_correction_stats(self.h, C.byref(arr)))
        return int(arr[0]), int(arr[1])

    def correction_stats_ex(self):
        """(half tiles of the last correction's global-gather fallback, half tiles in all, half tiles of the tiled kernel's second,
        one-workgroup-per-CU pass)."""
        arr = (C.c_uint64 * 3)()
        self._chk(self.lib.lfa_get_correction_stats_ex(self.h, C.byref(arr)))
        return int(arr[0]), int(arr[1]), int(arr[2])

    def set_step_overlap(self, on):
        """time_step: position correction on a second stream beside the pressure solve (default on); off = back to back."""
        self._chk(self.lib.lfa_set_step_overlap(self.h, int(bool(on))))

    # -- z-slab decomposition ------------------------------------------------------------------------------
    def init_local_slab(self, hub, rank, layer_bounds):
        b = np.ascontiguousarray(layer_bounds, dtype=np.int32)
        self._chk(self.lib.lfa_dist_init_local(self.h, hub, int(rank), _ptr(b)))

    def init_rccl_slab(self, rank, nranks, unique_id, layer_bounds):
        b = np.ascontiguousarray(layer_bounds, dtype=np.int32)
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        self._chk(self.lib.lfa_dist_init_rccl(self.h, int(rank), int(nranks), _ptr(uid), _ptr(b)))

    def init_shm_slab(self, name, rank, nranks, layer_bounds):
        """one process per rank, messages staged through the POSIX shared-memory segment `name` (lfa_dist_init_shm)"""
        b = np.ascontiguousarray(layer_bounds, dtype=np.int32)
        self._chk(self.lib.lfa_dist_init_shm(self.h, str(name).encode(), int(rank), int(nranks), _ptr(b)))

    def abandon_transport(self):
        """the job gives this handle's transport up (a peer failed): close() will not wait for the peers (lfa_dist_abandon)"""
        self._chk(self.lib.lfa_dist_abandon(self.h))

    def slab(self):
        lo, hi = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.lfa_dist_get_slab(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    # -- measurement ---------------------------------------------------------------------------------------
    def enable_timing(self, on=True):
        self._chk(self.lib.lfa_enable_timing(self.h, 1 if on else 0))

    def timings(self):
        arr = (C.c_double * NUM_TIMERS)()
        self._chk(self.lib.lfa_get_timings(self.h, C.byref(arr)))
        return dict(zip(TIMER_NAMES, list(arr)))

    def bench_kernel(self, name, reps=20):
        out = C.c_double(0.0)
        self._chk(self.lib.lfa_bench_kernel(self.h, KERNELS[name], int(reps), C.byref(out)))
        return out.value

    def bench_stream(self, nbytes=1 << 30, reps=10):
        """(device-copy GB/s, read-only GB/s) measured on this device."""
        c, r = C.c_double(0.0), C.c_double(0.0)
        self._chk(self.lib.lfa_bench_stream(self.h, int(nbytes), int(reps), C.byref(c), C.byref(r)))
        self.stream_variant = self.lib.lfa_bench_stream_variant().decode()
        return c.value, r.value

    def solver_stats(self):
        arr = (C.c_uint64 * 8)()
        self._chk(self.lib.lfa_get_solver_stats(self.h, C.byref(arr)))
        return dict(zip(["launches_per_iteration", "transport_calls_per_iteration", "mg_levels", "mg_first_level_in_coarse_launch",
                         "iterations", "transport_calls_per_solve", "whole_solve_in_one_launch", "device_waits_given_up"], list(arr)))

    def mg_level_tiles(self):
        arr = (C.c_uint64 * 12)()
        self._chk(self.lib.lfa_get_mg_level_tiles(self.h, C.byref(arr)))
        out = list(arr)
        while out and out[-1] == 0:
            out.pop()
        return out

    def counts(self):
        arr = (C.c_uint64 * 5)()
        self._chk(self.lib.lfa_get_counts(self.h, C.byref(arr)))
        return dict(zip(["particles", "unknowns", "particle_tiles", "processed_tiles", "padded_cells"], list(arr)))
