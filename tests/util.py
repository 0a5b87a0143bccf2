"""Shared helpers of the test-suite: scenes, staged CPU runs (oracle or real reference) and comparisons."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from libfluid_amd import scenes  # noqa: E402
from oracle import loader as orc  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIC, FLIP, APIC = 0, 1, 2

# name -> scene description. Small enough that the oracle and the reference finish in well under a second.
CASES = {
    "pic16": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=PIC, blend=1.0, solid=None, vel=3.0),
    "flip16": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=FLIP, blend=0.95, solid=None, vel=3.0),
    "apic16": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=APIC, blend=1.0, solid=None, vel=3.0),
    "apic16_solid": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=APIC, blend=1.0,
                         solid=((11, 3, 4), 3.2), vel=3.0),
    "flip16_solid": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=FLIP, blend=0.95,
                         solid=((11, 3, 4), 3.2), vel=3.0),
    # ragged grid (not a multiple of the 8-cell tile), block in the interior touching the max walls, rest state
    "apic_ragged": dict(size=(21, 13, 18), block=((9, 0, 5), (21, 9, 18)), method=APIC, blend=1.0, solid=None, vel=0.0),
    "pic_ragged": dict(size=(21, 13, 18), block=((9, 0, 5), (21, 9, 18)), method=PIC, blend=1.0,
                       solid=((5, 2, 9), 3.5), vel=2.0),
    # hydrostatic tank: fluid fills the bottom half wall to wall, at rest
    "apic_tank": dict(size=(12, 16, 12), block=((0, 0, 0), (12, 8, 12)), method=APIC, blend=1.0, solid=None, vel=0.0),
    # the parameters hosts actually set (Maya GridNode attributes, plugins/maya/nodes/grid_node.cpp:258-274): cell_size != 1,
    # grid_offset != 0, density != 1. This is where the fp64 key division (simulation.cpp:253), PIC's /h kernel (:313-315),
    # APIC's un-scaled one (:367-369: a NARROWER hat for h > 1, a WIDER one truncated by the 27-cell gather for h < 1),
    # _grad_kernel's /cell_size (:223), a_scale = dt/(rho h^2), coeff = dt/(rho h), the rhs 1/h and the running `+= h` face
    # positions show. Positions are h * (unit-grid positions) + offset; velocities are world units.
    "pic_h05": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=PIC, blend=1.0, solid=None, vel=1.5,
                    h=0.5, off=(0.3, -0.2, 0.1), rho=2.0),
    "flip_h17": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=FLIP, blend=0.95, solid=((11, 3, 4), 3.2),
                     vel=5.0, h=1.7, off=(0.3, -0.2, 0.1), rho=2.0),
    "apic_h05": dict(size=(16, 16, 16), block=((0, 0, 0), (8, 8, 8)), method=APIC, blend=1.0, solid=((11, 3, 4), 3.2),
                     vel=1.5, h=0.5, off=(0.3, -0.2, 0.1), rho=2.0),
    "apic_h17": dict(size=(21, 13, 18), block=((9, 0, 5), (21, 9, 18)), method=APIC, blend=1.0, solid=None, vel=5.0,
                     h=1.7, off=(-3.25, 0.7, 11.0), rho=0.5),
}
DT = 0.01


def case_params(c):
    """(cell_size, grid_offset, density) of a case; the round-1 cases are the testbed's h = 1, offset 0, rho = 1."""
    return float(c.get("h", 1.0)), tuple(c.get("off", (0.0, 0.0, 0.0))), float(c.get("rho", 1.0))


def vel_atol(c):
    """Velocity scale one step injects (|g| dt, world units): the absolute floor for fields whose exact value is ~0."""
    return 1e-5 * 981.0 * DT


def make_case(name):
    c = CASES[name]
    h, off, rho = case_params(c)
    parts = scenes.seed_block(*c["block"], cell_size=h, offset=off)
    if c["vel"]:
        rng = np.random.default_rng(1234)
        n = len(parts)
        parts["vel"] = rng.normal(size=(n, 3)) * c["vel"]
        for f in ("cx", "cy", "cz"):
            parts[f] = rng.normal(size=(n, 3)) * c["vel"] * 0.1
    solid = None
    if c["solid"] is not None:
        solid = scenes.sphere_solid_cells(c["size"], *c["solid"])
    return c, parts, solid


def cpu_sim(c, kind="oracle", solid=None):
    """CpuSim (oracle or real reference) set up with a case's grid, method and physical parameters."""
    h, off, rho = case_params(c)
    s = orc.CpuSim(c["size"], cell_size=h, offset=off, method=c["method"], blending=c["blend"], density=rho, kind=kind)
    if solid is not None:
        s.set_solid_cells(solid)
    return s


def order_by_position(parts):
    """Permutation that sorts particles by position (positions are unique): identifies particles across sorts."""
    p = parts["pos"]
    return np.lexsort((p[:, 2], p[:, 1], p[:, 0]))


def staged_cpu_run(name, kind, steps=2):
    """Runs `steps` passes of the hot path stage by stage on the oracle or the reference and records every stage."""
    c, parts, solid = make_case(name)
    s = cpu_sim(c, kind, solid)
    s.set_particles(parts)
    in_order = order_by_position(parts)
    out = {}
    for st in range(steps):
        s.hash()
        out[f"fluid_cells{st}"] = s.fluid_cells()
        out[f"counts{st}"] = s.space_hash()[1].astype(np.uint32)
        s.p2g()
        cells = s.cells()
        out[f"p2g_vel{st}"] = cells["vel"].copy()
        out[f"p2g_type{st}"] = cells["type"].copy()
        if c["method"] == FLIP:
            out[f"old_vel{st}"] = s.old_cells()["vel"].copy()
        s.add_gravity(DT)
        out[f"grav_vel{st}"] = s.cells()["vel"].copy()
        s.build_system(DT)
        out[f"abits{st}"] = s.abits()
        out[f"b{st}"] = s.b()
        out[f"precon{st}"] = s.precon()
        n = len(out[f"b{st}"])
        probe = np.sin(np.arange(n) * 0.37) + 0.25
        out[f"Mprobe{st}"] = s.apply_precon(probe)
        out[f"Aprobe{st}"] = s.apply_a(probe)
        p, res, it = s.solve(DT)
        out[f"p{st}"] = p
        out[f"residual{st}"] = np.float64(res)
        out[f"iters{st}"] = np.int64(it)
        s.apply_pressure(DT, p)
        out[f"apply_vel{st}"] = s.cells()["vel"].copy()
        s.extrapolate()
        out[f"extrap_vel{st}"] = s.cells()["vel"].copy()
        s.g2p()
        after = s.particles()
        back = np.empty_like(in_order)
        back[in_order] = order_by_position(after)  # after[back[i]] is input particle i
        after = after[back]
        assert np.array_equal(after["pos"], parts["pos"])
        out[f"g2p_vel{st}"] = after["vel"].copy()
        out[f"g2p_c{st}"] = np.concatenate([after["cx"], after["cy"], after["cz"]], axis=1)
        out[f"raw{st}"] = after["raw"].copy()
        out[f"cfl{st}"] = np.float64(s.cfl())
    s.close()
    return out


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_golden(name):
    with np.load(golden_path(name)) as z:
        return {k: z[k] for k in z.files}


def pointwise_rel(a, b, floor):
    """max_i |a_i - b_i| / (|b_i| + floor): the pointwise relative error with an absolute floor (a quantity that is exactly 0
    in the reference - a face between air cells - has no relative error)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float((np.abs(a - b) / (np.abs(b) + floor)).max())


def assert_close(a, b, rel, what, atol=0.0, pw=None):
    """Max-norm bar: max|a - b| <= rel * max|b| + atol. pw = (rel_pw, floor_frac) adds the POINTWISE bar
    |a_i - b_i| <= rel_pw * (|b_i| + floor_frac * max|b|) for every i, so that small entries of a field with a large maximum
    (near-surface pressures under a deep column) are held to a relative error of their own."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.size == 0:
        return
    assert np.isfinite(a).all() and np.isfinite(b).all(), what
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max()
    assert err <= rel * scale + atol, f"{what}: max|diff| {err:.3e} > {rel:.1e} * {scale:.3e}"
    if pw is not None:
        rel_pw, floor_frac = pw
        e = pointwise_rel(a, b, floor_frac * scale + atol)
        assert e <= rel_pw, f"{what}: pointwise relative error {e:.3e} > {rel_pw:.1e} (floor {floor_frac:.1e} * max)"


def assert_same_record(got, want, rel, label):
    """Compares two stage dictionaries: integer fields bit-exact, floating-point fields to `rel` of their max."""
    assert set(got) == set(want), (label, sorted(set(got) ^ set(want)))
    for k in sorted(want):
        w, g = want[k], got[k]
        if np.asarray(w).dtype.kind in "iu":
            assert np.array_equal(np.asarray(g), np.asarray(w)), f"{label}:{k} differs"
        elif k.startswith("residual"):
            # signed max(r) at the converging iteration: a quantity of size <= 1e-6 carrying the rounding of the
            # whole solve (the residual itself is O(10) before the first iteration)
            assert_close(g, w, 0.0, f"{label}:{k}", atol=max(1e-10, 1e6 * rel * 1e-6))
        elif np.isinf(np.asarray(w)).any():
            assert np.array_equal(np.asarray(g), np.asarray(w)), f"{label}:{k} differs"
        else:
            assert_close(g, w, rel, f"{label}:{k}", atol=1e-300)
