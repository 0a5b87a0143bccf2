"""GPU parity tests (-m gpu): every stage of the HIP path, called through the C ABI (libfluid_amd.so), against
 (1) the golden vectors produced by the real reference (tests/golden/*.npz), and
 (2) the CPU oracle (oracle/oracle.c) run here on the same seeded inputs.

Bars (SURVEY.md 8d "parity gates"):
  bit-exact   particle cell keys, per-cell particle counts, _fluid_cells, cell types, A bits, PCG iteration count
              (exact MIC(0) schedule, fp64 vectors)
  fp32 fields face velocities after P2G / gravity / apply / extrapolate and particle v, C after G2P: max-norm relative
              error <= 2e-5 (fp32 storage and arithmetic against the fp64 reference; sums of O(64) terms)
  pressure    max-norm relative error <= 1e-4 (north star), in every preconditioner / dtype combination
"""
import os

import numpy as np
import pytest

import libfluid_amd as lfa
from tests import util

pytestmark = pytest.mark.gpu

VEL_REL = 2e-5
P_REL = 1e-4
# velocity scale one step injects (|g| dt): the absolute floor for fields whose exact value is ~0 (tank at rest)
VEL_ATOL = 1e-5 * 981.0 * util.DT


def make_gpu(name, **extra):
    c, parts, solid = util.make_case(name)
    h, off, rho = util.case_params(c)
    s = lfa.Sim(c["size"], cell_size=h, offset=off, density=rho, method=c["method"], blending=c["blend"], **extra)
    if solid is not None:
        s.set_solid_cells(solid)
    s.upload_particles(parts)
    return c, parts, solid, s


def cells_from(vel, types):
    cells = np.zeros(len(vel), dtype=lfa.CELL_DTYPE)
    cells["vel"] = vel
    cells["type"] = types
    return cells


@pytest.mark.parametrize("name", sorted(util.CASES))
def test_keys_counts_and_fluid_cells_bit_exact(name):
    c, parts, solid, s = make_gpu(name)
    g = util.load_golden(name)
    s.hash()
    assert np.array_equal(s.fluid_cells(), g["fluid_cells0"])
    assert np.array_equal(s.cell_counts(), g["counts0"])
    out = s.download_particles(into=parts.copy())
    assert np.array_equal(out["raw"], g["raw0"])
    assert np.array_equal(out["pos"], parts["pos"])
    # positions reconstructed from the device's (cell, fraction) representation: fp32 fraction of a cell
    rec = s.download_particles()
    assert np.abs(rec["pos"] - parts["pos"]).max() <= 2.0 ** -23 * util.case_params(c)[0] + 1e-13
    s.close()


@pytest.mark.parametrize("variant", [lfa.P2G_LDS_BINNED, lfa.P2G_GLOBAL_ATOMIC])
@pytest.mark.parametrize("name", sorted(util.CASES))
def test_p2g_and_gravity(name, variant):
    c, parts, solid, s = make_gpu(name, p2g_variant=variant)
    g = util.load_golden(name)
    s.hash()
    s.p2g()
    cells = s.cells()
    assert np.array_equal(cells["type"], g["p2g_type0"])
    util.assert_close(cells["vel"], g["p2g_vel0"], VEL_REL, "p2g velocities")
    if c["method"] == util.FLIP:
        util.assert_close(s.old_cells()["vel"], g["old_vel0"], VEL_REL, "flip old grid")
    s.add_gravity(util.DT)
    util.assert_close(s.cells()["vel"], g["grav_vel0"], VEL_REL, "after gravity")
    s.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("name", ["apic16", "apic16_solid", "pic_ragged", "apic_tank", "pic_h05", "flip_h17", "apic_h05", "apic_h17"])
def test_system_matrix_rhs_and_exact_mic(name, dtype):
    """A bits bit-exact; b, MIC(0) factor, M^-1 probe and A probe against the reference on the reference's own grid."""
    c, parts, solid, s = make_gpu(name, precond=lfa.PRECOND_MIC0_EXACT, pcg_dtype=dtype)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    s.build_system(util.DT)
    assert np.array_equal(s.abits(), g["abits0"])
    rel = 1e-6 if dtype == lfa.PCG_F32 else 1e-6  # b is built from fp32 face velocities in both cases
    util.assert_close(s.b(), g["b0"], rel, "rhs")
    util.assert_close(s.precon(), g["precon0"], 1e-6 if dtype == lfa.PCG_F32 else 1e-13, "MIC(0) factor")
    n = len(g["b0"])
    probe = np.sin(np.arange(n) * 0.37) + 0.25
    util.assert_close(s.apply_a(probe), g["Aprobe0"], 1e-6 if dtype == lfa.PCG_F32 else 1e-13, "A probe")
    util.assert_close(s.apply_precon(probe), g["Mprobe0"], 2e-5 if dtype == lfa.PCG_F32 else 1e-12, "M^-1 probe")
    s.close()


@pytest.mark.parametrize("name", sorted(util.CASES))
def test_pcg_exact_schedule_matches_reference_iterations(name):
    """Hyperplane-ordered MIC(0) with fp64 vectors: same recurrence as the reference => same iteration count."""
    c, parts, solid, s = make_gpu(name, precond=lfa.PRECOND_MIC0_EXACT, pcg_dtype=lfa.PCG_F64)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    p, res, it, rc = s.solve(util.DT)
    assert rc == 0
    assert it == int(g["iters0"])
    assert res < 1e-6
    util.assert_close(p, g["p0"], 1e-6, "pressure (exact MIC, f64)")
    s.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("precond", [lfa.PRECOND_MIC0_TILED, lfa.PRECOND_MIC0_EXACT, lfa.PRECOND_MULTILEVEL, lfa.PRECOND_MULTIGRID])
@pytest.mark.parametrize("name", ["apic16_solid", "flip16", "pic_ragged", "apic_tank", "pic_h05", "flip_h17", "apic_h05", "apic_h17"])
def test_pcg_pressure_within_tolerance(name, precond, dtype):
    c, parts, solid, s = make_gpu(name, precond=precond, pcg_dtype=dtype)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    p, res, it, rc = s.solve(util.DT)
    assert rc == 0 and 0 < it <= 200 and res < 1e-6
    util.assert_close(p, g["p0"], P_REL, "pressure")
    s.close()


@pytest.mark.parametrize("name", sorted(util.CASES))
def test_apply_pressure_extrapolate_g2p(name):
    """Each downstream stage from the reference's state of the stage before it."""
    c, parts, solid, s = make_gpu(name)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    s.build_system(util.DT)
    s.upload_pressure(g["p0"])
    s.apply_pressure(util.DT)
    util.assert_close(s.cells()["vel"], g["apply_vel0"], VEL_REL, "after apply_pressure")
    s.upload_cells(cells_from(g["apply_vel0"], g["p2g_type0"]))
    s.extrapolate()
    util.assert_close(s.cells()["vel"], g["extrap_vel0"], VEL_REL, "after extrapolation")
    if c["method"] == util.FLIP:
        # FLIP's blend reads the old grid its own P2G left on the device (src/simulation.cpp:340-344,463-505): run that P2G,
        # then replace the live grid by the reference's extrapolated one - the G2P stage alone against the reference's
        s.hash()
        s.p2g()
        util.assert_close(s.old_cells()["vel"], g["old_vel0"], VEL_REL, "flip old grid")
    s.upload_cells(cells_from(g["extrap_vel0"], g["p2g_type0"]))
    s.g2p()
    out = s.download_particles(into=parts.copy())
    util.assert_close(out["vel"], g["g2p_vel0"], VEL_REL * (2 if c["method"] == util.FLIP else 1), "particle velocity after G2P")
    if c["method"] == util.APIC:
        cc = np.concatenate([out["cx"], out["cy"], out["cz"]], axis=1)
        util.assert_close(cc, g["g2p_c0"], 5e-5, "APIC C after G2P")
    s.close()


@pytest.mark.parametrize("precond,dtype", [(lfa.PRECOND_MIC0_EXACT, lfa.PCG_F64), (lfa.PRECOND_MIC0_TILED, lfa.PCG_F32),
                                           (lfa.PRECOND_MULTILEVEL, lfa.PCG_F32), (lfa.PRECOND_MULTIGRID, lfa.PCG_F32)])
@pytest.mark.parametrize("name", sorted(util.CASES))
def test_two_hot_steps_end_to_end(name, precond, dtype):
    """Two full passes of the hot path, staged exactly like the golden run, against the reference's final state."""
    c, parts, solid, s = make_gpu(name, precond=precond, pcg_dtype=dtype)
    g = util.load_golden(name)
    for st in range(2):
        s.hash()
        assert np.array_equal(s.fluid_cells(), g[f"fluid_cells{st}"])
        s.p2g()
        s.add_gravity(util.DT)
        p, res, it, rc = s.solve(util.DT)
        assert rc == 0
        if precond == lfa.PRECOND_MIC0_EXACT:
            assert abs(it - int(g[f"iters{st}"])) <= 1  # inputs carry fp32 P2G rounding here
        util.assert_close(p, g[f"p{st}"], P_REL, f"pressure step {st}")
        s.apply_pressure(util.DT)
        s.extrapolate()
        util.assert_close(s.cells()["vel"], g[f"extrap_vel{st}"], 1e-4, f"grid step {st}", atol=VEL_ATOL)
        s.g2p()
        out = s.download_particles(into=parts.copy())
        util.assert_close(out["vel"], g[f"g2p_vel{st}"], 1e-4, f"particle velocity step {st}", atol=VEL_ATOL)
        util.assert_close(np.concatenate([out["cx"], out["cy"], out["cz"]], axis=1), g[f"g2p_c{st}"], 2e-4,
                          f"particle C step {st}", atol=VEL_ATOL)
        hh = util.case_params(c)[0]
        vmax_ref = hh / float(g[f"cfl{st}"])  # cfl = cell_size / max|v| (src/simulation.cpp:199-205)
        assert abs(hh / s.cfl() - vmax_ref) <= 1e-4 * vmax_ref + VEL_ATOL
    s.close()


@pytest.mark.parametrize("name", ["apic16_solid", "flip16", "apic_ragged", "pic_h05", "flip_h17", "apic_h05", "apic_h17"])
def test_step_hot_equals_staged_calls_and_oracle(name):
    """lfa_step_hot (fused gravity, no intermediate downloads) against the oracle's hot_step on the same inputs."""
    from oracle import loader as orc
    c, parts, solid, s = make_gpu(name, precond=lfa.PRECOND_MIC0_EXACT, pcg_dtype=lfa.PCG_F64)
    o = util.cpu_sim(c, "oracle", solid)
    o.set_particles(parts)
    for st in range(2):
        res, it, rc = s.step_hot(util.DT)
        po, reso, ito = o.hot_step(util.DT)
        assert rc == 0 and abs(it - ito) <= 1
        util.assert_close(s.pressure(), po, P_REL, "pressure")
        util.assert_close(s.cells()["vel"], o.cells()["vel"], 1e-4, "grid", atol=VEL_ATOL)
    got = s.download_particles(into=parts.copy())
    want = o.particles()
    want = want[util.order_by_position(want)]
    gi = util.order_by_position(got)
    util.assert_close(got["vel"][gi], want["vel"], 1e-4, "particle velocities", atol=VEL_ATOL)
    assert np.array_equal(got["raw"][gi], want["raw"])
    s.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
def test_multilevel_preconditioner_is_symmetric_and_cuts_iterations(dtype):
    """The coarse-space correction must keep M^-1 symmetric positive definite (CG needs it) and pay for itself."""
    name = "apic_ragged"
    c, parts, solid, s = make_gpu(name, precond=lfa.PRECOND_MULTILEVEL, pcg_dtype=dtype)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    s.build_system(util.DT)
    n = len(g["b0"])
    rng = np.random.default_rng(7)
    x, y = rng.normal(size=n), rng.normal(size=n)
    mx, my = s.apply_precon(x), s.apply_precon(y)
    assert abs(y @ mx - x @ my) <= (2e-5 if dtype == lfa.PCG_F32 else 1e-11) * (abs(y @ mx) + np.linalg.norm(x) * np.linalg.norm(my))
    assert x @ mx > 0 and y @ my > 0
    p, res, it_ml, rc = s.solve(util.DT)
    util.assert_close(p, g["p0"], P_REL, "pressure (multilevel)")
    s.set_params(precond=lfa.PRECOND_MIC0_TILED)
    _, _, it_tiled, _ = s.solve(util.DT)
    assert it_ml <= it_tiled
    s.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("name", ["apic_ragged", "apic16_solid", "apic_tank"])
def test_multigrid_preconditioner_is_symmetric_positive_definite(name, dtype):
    """The V-cycle (down: red->black from zero, up: black->red) must be a symmetric positive definite operator."""
    c, parts, solid, s = make_gpu(name, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    g = util.load_golden(name)
    s.hash()
    s.upload_cells(cells_from(g["grav_vel0"], g["p2g_type0"]))
    s.build_system(util.DT)
    n = len(g["b0"])
    rng = np.random.default_rng(11)
    x, y = rng.normal(size=n), rng.normal(size=n)
    mx, my = s.apply_precon(x), s.apply_precon(y)
    assert abs(y @ mx - x @ my) <= (2e-5 if dtype == lfa.PCG_F32 else 1e-11) * (abs(y @ mx) + np.linalg.norm(x) * np.linalg.norm(my))
    assert x @ mx > 0 and y @ my > 0
    p, res, it, rc = s.solve(util.DT)
    assert rc == 0 and res < 1e-6
    util.assert_close(p, g["p0"], P_REL, "pressure (multigrid)")
    s.close()


@pytest.mark.parametrize("size,block", [((8, 8, 8), ((0, 0, 0), (5, 4, 6))), ((7, 5, 3), ((1, 0, 0), (6, 3, 3))),
                                        ((40, 9, 24), ((3, 0, 2), (37, 6, 20)))])
def test_multigrid_on_grids_of_one_or_few_tiles(size, block):
    """A grid that fits one tile has a single level (the two smoothing sweeps alone); a flat one has levels that stop
    coarsening in one direction. Same pressure as the reference's schedule."""
    parts = util.scenes.seed_block(*block)
    rng = np.random.default_rng(3)
    parts["vel"] = rng.normal(size=(len(parts), 3)) * 2.0
    ps = {}
    for precond in (lfa.PRECOND_MIC0_EXACT, lfa.PRECOND_MULTIGRID):
        s = lfa.Sim(size, precond=precond, pcg_dtype=lfa.PCG_F64)
        s.upload_particles(parts)
        res, it, rc = s.step_hot(util.DT)
        assert rc == 0 and res < 1e-6
        ps[precond] = s.pressure()
        s.close()
    util.assert_close(ps[lfa.PRECOND_MULTIGRID], ps[lfa.PRECOND_MIC0_EXACT], 1e-6, "pressure")


def test_multigrid_follows_a_moving_free_surface():
    """40 device-resident time steps of a collapsing column: the particle-tile set changes from step to step (the cached tile
    lists of the hierarchy are rebuilt, vectors of tiles that left the set are cleared), every solve converges in a few
    iterations, and the flow stays the one the MIC(0)-preconditioned solver produces (both solve the same system to the same
    residual, so the centres of mass agree far better than the cell size)."""
    size, block = (64, 40, 24), ((0, 0, 0), (20, 30, 24))
    com = {}
    for precond in (lfa.PRECOND_MULTIGRID, lfa.PRECOND_MULTILEVEL):
        s = lfa.Sim(size, precond=precond)
        s.seed_block(*block)
        its, t = [], 0.0
        for _ in range(40):
            dt = min(3.0 * s.cfl(), 0.004)
            res, it, rc = s.time_step(dt)
            assert rc == 0 and res < 1e-6, (precond, rc, res)
            its.append(it)
            t += dt
        pos = s.download_particles(write_positions=True)["pos"]
        assert np.isfinite(pos).all() and (pos >= 0).all() and (pos <= np.asarray(size)).all()
        com[precond] = (pos.mean(axis=0), max(its), t)
        s.close()
    (ca, ia, ta), (cb, ib, tb) = com[lfa.PRECOND_MULTIGRID], com[lfa.PRECOND_MULTILEVEL]
    assert ia <= 30, ia
    assert abs(ta - tb) <= 1e-3 * tb            # the CFL-limited time steps follow the same velocities
    assert ca[0] > 10.5                          # the column has collapsed sideways (it started at x = 10)
    assert np.abs(ca - cb).max() < 0.05, (ca, cb)


def test_multigrid_iteration_count_barely_grows_with_the_grid():
    """Dam-break blocks of 16^3 ... 64^3 cells at rest: MIC(0) per tile needs several times the iterations when the block
    doubles twice, the V-cycle a handful more; both converge to the same pressure."""
    its, pres = {}, {}
    for n in (16, 32, 64):
        for precond in (lfa.PRECOND_MULTIGRID, lfa.PRECOND_MIC0_TILED):
            s = lfa.Sim((2 * n,) * 3, precond=precond, max_iterations=1000)
            s.seed_block((0, 0, 0), (n, n, n))
            res, it, rc = s.step_hot(0.01)
            assert rc == 0
            its[(n, precond)] = it
            pres[(n, precond)] = s.pressure()
            s.close()
        util.assert_close(pres[(n, lfa.PRECOND_MULTIGRID)], pres[(n, lfa.PRECOND_MIC0_TILED)], P_REL, "pressure")
    mg = [its[(n, lfa.PRECOND_MULTIGRID)] for n in (16, 32, 64)]
    tiled = [its[(n, lfa.PRECOND_MIC0_TILED)] for n in (16, 32, 64)]
    assert max(mg) <= 30 and mg[2] <= mg[0] + 10, (mg, tiled)
    assert tiled[2] >= 2 * mg[2], (mg, tiled)


def test_seed_block_matches_numpy_generator():
    s = lfa.Sim((24, 16, 16))
    s.seed_block((2, 1, 3), (10, 9, 11))
    want = util.scenes.seed_block((2, 1, 3), (10, 9, 11))
    got = s.download_particles()
    assert got.shape == want.shape
    assert np.abs(got["pos"] - want["pos"]).max() <= 2.0 ** -23
    s.hash()
    o = lfa.Sim((24, 16, 16))
    o.upload_particles(want)
    o.hash()
    assert np.array_equal(s.cell_counts(), o.cell_counts())
    assert np.array_equal(s.download_particles()["raw"], o.download_particles()["raw"])
    s.close(); o.close()


def test_particle_on_the_max_face_and_empty_set():
    """Edge cases of SURVEY 8(a1): unclamped index == size (position exactly on the max face) and no particles."""
    from oracle import loader as orc
    size = (8, 8, 8)
    parts = util.scenes.seed_block((5, 5, 5), (8, 8, 8))
    parts["pos"][0] = (8.0, 7.25, 6.5)     # on the +x wall
    parts["pos"][1] = (7.5, 8.0, 8.0)      # on an edge
    parts["pos"][2] = (0.0, 6.5, 7.75)     # on the -x wall
    parts["vel"] = np.random.default_rng(5).normal(size=(len(parts), 3))
    s = lfa.Sim(size, precond=lfa.PRECOND_MIC0_EXACT, pcg_dtype=lfa.PCG_F64)
    o = orc.CpuSim(size)
    s.upload_particles(parts); o.set_particles(parts)
    s.step_hot(util.DT); o.hot_step(util.DT)
    assert np.array_equal(s.fluid_cells(), o.fluid_cells())
    got = s.download_particles(into=parts.copy())
    want = o.particles()
    gi, wi = util.order_by_position(got), util.order_by_position(want)
    assert np.array_equal(got["raw"][gi], want["raw"][wi])
    util.assert_close(got["vel"][gi], want["vel"][wi], 1e-4, "velocities with wall particles")
    util.assert_close(np.concatenate([got["cx"], got["cy"], got["cz"]], axis=1)[gi],
                      np.concatenate([want["cx"], want["cy"], want["cz"]], axis=1)[wi], 2e-4, "C with wall particles")
    s.close()
    e = lfa.Sim(size)
    e.upload_particles(np.zeros(0, dtype=lfa.PARTICLE_DTYPE))
    res, it, rc = e.step_hot(util.DT)
    assert (res, it, rc) == (0.0, 0, 0) and e.num_fluid_cells == 0 and np.isinf(e.cfl())
    e.close()


def test_multiple_extrapolation_iterations():
    from oracle import loader as orc
    c, parts, solid, s = make_gpu("apic16_solid", velocity_extrapolation_iterations=3)
    o = orc.CpuSim(c["size"], method=c["method"])
    o.set_solid_cells(solid); o.set_particles(parts); o.set_extrapolation_iterations(3)
    o.hash(); o.p2g(); o.add_gravity(util.DT)
    s.hash(); s.p2g(); s.add_gravity(util.DT)
    o.build_system(util.DT); o.extrapolate()
    s.extrapolate()
    util.assert_close(s.cells()["vel"], o.cells()["vel"], VEL_REL, "3 extrapolation sweeps")
    s.close()


def test_properties_at_scale():
    """128^3 / 2M particles (BASELINE config 2): size-independent properties instead of an oracle run.
    - hydrostatic column: p = rho |g| depth to 1e-4, velocities stay ~0
    - the projected grid field is divergence free (rhs of a second build ~ 0 relative to the first)
    - binned and global-atomic P2G agree."""
    n = 128
    s = lfa.Sim((n, n, n), precond=lfa.PRECOND_MULTILEVEL, pcg_dtype=lfa.PCG_F32)
    s.seed_block((0, 0, 0), (n, 32, n))  # tank filled wall to wall, 32 cells deep
    res, it, rc = s.step_hot(util.DT)
    assert rc == 0 and res < 1e-6
    fc = s.fluid_cells().astype(np.int64)
    y = (fc // n) % n
    util.assert_close(s.pressure(), 981.0 * (32 - y), P_REL, "hydrostatic pressure at 128^3")
    assert np.abs(s.download_particles()["vel"]).max() < 981.0 * util.DT * 1e-3
    s.close()

    s = lfa.Sim((n, n, n))
    s.seed_block((0, 0, 0), (64, 64, 64))
    s.hash(); s.p2g(); s.add_gravity(util.DT)
    s.build_system(util.DT)
    b0 = np.abs(s.b()).max()
    p, res, it, rc = s.solve(util.DT)
    assert rc == 0
    s.apply_pressure(util.DT)
    s.build_system(util.DT)
    # fp32 pressures of O(6e4) carry ulp(p) ~ 4e-3, i.e. a divergence floor of ~1e-4 |b|; the reference's SIGNED
    # max(r) stopping rule (pressure_solver.cpp:54) does not bound negative residuals either
    assert np.abs(s.b()).max() < 1e-3 * b0
    a = s.cells()["vel"].copy()
    s.close()
    t = lfa.Sim((n, n, n), p2g_variant=lfa.P2G_GLOBAL_ATOMIC)
    t.seed_block((0, 0, 0), (64, 64, 64))
    t.hash(); t.p2g(); t.add_gravity(util.DT)
    t.solve(util.DT); t.apply_pressure(util.DT)
    util.assert_close(t.cells()["vel"], a, 1e-4, "binned vs atomic P2G at 128^3")
    t.close()


def test_properties_at_full_size():
    """BASELINE config 4 sizes (512^3 grid, 67 M particles, 8.4 M unknowns): no oracle run is affordable there, so
    size-independent properties (what moves between host and device stays small: ids, unknown lists, pressures).
    - a wall-to-wall tank 32 cells deep (the same particle and unknown counts as C4) is hydrostatic: p = rho |g| depth
    - the projected field is divergence free
    - the C4 dam break: binning is a permutation of the particle ids, the multigrid PCG converges in a few iterations,
      and a second step from the updated particles works on the same unknown count."""
    n = 512
    s = lfa.Sim((n, n, n), pcg_dtype=lfa.PCG_F32)
    s.seed_block((0, 0, 0), (n, 32, n))
    assert s.counts()["particles"] == 67108864
    s.hash(); s.p2g(); s.add_gravity(util.DT)
    s.build_system(util.DT)
    b0 = np.abs(s.b()).max()
    p, res, it, rc = s.solve(util.DT)
    assert rc == 0 and it <= 30, (rc, it)
    fc = s.fluid_cells().astype(np.int64)
    assert len(fc) == 8388608
    y = (fc // n) % n
    util.assert_close(p, 981.0 * (32 - y), P_REL, "hydrostatic pressure at 512^3")
    s.apply_pressure(util.DT)
    s.build_system(util.DT)
    assert np.abs(s.b()).max() < 1e-3 * b0  # fp32 pressure: divergence floor ~1e-4 |b| (see test_properties_at_scale)
    s.close()

    cfg = util.scenes.CONFIGS["C4"]
    s = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"])
    s.seed_block(*cfg["block"])
    its = []
    for _ in range(2):
        res, it, rc = s.step_hot(0.033)
        assert rc == 0 and res < 1e-6
        its.append(it)
    assert max(its) <= 25, its
    c = s.counts()
    assert c["particles"] == 67108864 and c["unknowns"] == 8388608
    ids = np.sort(s.particle_ids())
    assert ids[0] == 0 and ids[-1] == len(ids) - 1 and np.all(np.diff(ids) == 1), "binning must permute the particles"
    s.close()


def test_pcg_warm_start_converges_to_the_same_pressure_in_fewer_iterations():
    """lfa_params.pcg_warm_start: the second solve of a state that barely changed starts from the first one's pressure."""
    c, parts, solid, s = make_gpu("apic16_solid", pcg_warm_start=1)
    g = util.load_golden("apic16_solid")
    s.hash(); s.p2g(); s.add_gravity(util.DT)
    p0, res0, it0, rc0 = s.solve(util.DT)
    cells = s.cells()
    cells["vel"] *= 1.001  # a slightly different right-hand side
    s.upload_cells(cells)
    p1, res1, it1, rc1 = s.solve(util.DT)
    assert rc0 == 0 and rc1 == 0 and res1 < 1e-6 and 0 < it1 < it0, (it0, it1)
    t = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    t.set_solid_cells(solid)
    t.upload_particles(parts)
    t.hash()
    t.upload_cells(cells)
    pc, _, itc, _ = t.solve(util.DT)
    assert itc >= it0 - 1
    util.assert_close(p1, pc, P_REL, "warm-started pressure")
    s.close(); t.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("size,block", [((136, 72, 104), ((0, 0, 0), (90, 50, 70))), ((64, 48, 40), ((3, 0, 2), (50, 30, 33))),
                                        ((200, 120, 96), ((0, 0, 0), (200, 60, 96)))])
def test_multigrid_single_launch_coarse_levels_are_bitwise_the_launch_per_phase_path(dtype, size, block, monkeypatch):
    """k_mg_coarse runs every phase of the coarse levels (pre-smoothing, residual + restriction, coarsest solve, prolongation +
    post-smoothing) inside one launch as dataflow between workgroups: one tile per workgroup and level, resident in LDS, ready
    flags tagged with the launch number, the level arrays accessed with agent-scope atomics. Same arithmetic per cell as the
    launch-per-phase kernels (LFA_MG_NO_PERSIST=1, the path a handle retreats to): identical iteration counts, bit-identical
    pressures - on a ragged three-level grid, a small one (everything below the finest level inside the launch) and one whose
    level 1 (375 tiles) is the launch's first level."""
    res = []
    # "tagged": the default with fp32 vectors - values travel between the workgroups as {tag, value} words (fp64: the same as
    # "flags"); "flags": level arrays + ready flags (LFA_MG_NO_TAGGED=1); "phase": a launch per phase
    # "tagged" (fp32) also runs the level above the launch's first one inside it, in launch order; "tagged-no-top": without that
    for mode in ("tagged", "tagged-no-top", "flags", "phase"):
        for k in ("LFA_MG_NO_PERSIST", "LFA_MG_NO_TAGGED", "LFA_MG_NO_TOP"):
            monkeypatch.delenv(k, raising=False)
        if mode == "phase":
            monkeypatch.setenv("LFA_MG_NO_PERSIST", "1")
        elif mode == "flags":
            monkeypatch.setenv("LFA_MG_NO_TAGGED", "1")
        elif mode == "tagged-no-top":
            monkeypatch.setenv("LFA_MG_NO_TOP", "1")
        s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
        s.seed_block(*block)
        its = []
        for _ in range(3):
            r, it, rc = s.step_hot(util.DT)
            assert rc == 0
            its.append(it)
        st = s.solver_stats()
        assert (st["mg_first_level_in_coarse_launch"] >= 1) == (mode != "phase") and st["device_waits_given_up"] == 0, st
        res.append((its, s.pressure().copy()))
        s.close()
    for other in res[1:]:
        assert res[0][0] == other[0]
        assert np.array_equal(res[0][1], other[1])


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("scene", ["pool_and_spray", "spray_only", "moving"])
def test_level_1_tiles_without_unknowns_are_left_out_bitwise(scene, dtype, monkeypatch):
    """Round 6: a level-1 tile that holds no unknown of level 1 - the parent of spray, of a film thinner than a coarse cell (a coarse
    cell is AIR as soon as one child is) - is no longer in level 1's active set; levels 2.. are the parents of what is left. Such a
    tile only ever computed zeros, so the V-cycle is the same operator: identical iteration counts and bit-identical pressures
    against LFA_MG_NO_PRUNE=1, fewer tiles on the coarse levels. "spray_only": NO coarse cell is an unknown - the hierarchy is the
    finest level's sweeps alone. "moving": the active set changes from step to step (tiles arrive in and leave level 1 while their
    children keep restricting into them)."""
    size = (96, 64, 80)
    rng = np.random.default_rng(5)
    if scene == "moving":
        parts = util.scenes.seed_block((0, 0, 0), (40, 40, 30))
    else:
        # droplets: single cells with 8 particles each, scattered over the air
        cells = np.unique(rng.integers([0, 30, 0], [96, 64, 80], size=(400, 3)), axis=0)
        sub = np.array([(a, b, c) for a in (0.25, 0.75) for b in (0.25, 0.75) for c in (0.25, 0.75)])
        pos = (cells[:, None, :] + sub[None, :, :]).reshape(-1, 3)
        spray = np.zeros(len(pos), dtype=lfa.PARTICLE_DTYPE)
        spray["pos"] = pos
        spray["vel"] = rng.normal(size=pos.shape) * 20.0
        parts = spray if scene == "spray_only" else np.concatenate([util.scenes.seed_block((0, 0, 0), (96, 20, 80)), spray])
    res = []
    monkeypatch.setenv("LFA_MG_NO_CLOSED", "1")  # (the droplets stay in the PCG here: it is their PARENTS this test is about)
    for prune in (True, False):
        monkeypatch.delenv("LFA_MG_NO_PRUNE", raising=False)
        if not prune:
            monkeypatch.setenv("LFA_MG_NO_PRUNE", "1")
        s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
        s.upload_particles(parts)
        its, ps, tiles = [], [], []
        for k in range(12 if scene == "moving" else 2):
            if scene == "moving":
                r, it, rc = s.time_step(min(3.0 * s.cfl(), 0.033))
            else:
                r, it, rc = s.step_hot(util.DT)
            assert rc == 0
            its.append(it)
            tiles.append(s.mg_level_tiles())
            if scene != "moving" or k >= 10:
                ps.append(s.pressure().copy())
        assert s.solver_stats()["device_waits_given_up"] == 0
        res.append((its, ps, tiles))
        s.close()
    (its_p, ps_p, tiles_p), (its_n, ps_n, tiles_n) = res
    if scene == "moving":  # (full steps: the position correction sums in atomic order - the runs drift apart in the last bits)
        assert all(abs(a - b) <= 1 for a, b in zip(its_p, its_n)), (its_p, its_n)
    else:
        assert its_p == its_n, (its_p, its_n)
        assert all(np.array_equal(a, b) for a, b in zip(ps_p, ps_n))
        # fewer level-1 tiles: the droplets' parents hold no coarse unknown
        l1 = lambda t: t[1] if len(t) > 1 else 0  # (mg_level_tiles drops trailing empty levels)
        assert l1(tiles_p[-1]) < l1(tiles_n[-1]), (tiles_p[-1], tiles_n[-1])
        if scene == "spray_only":
            assert len(tiles_p[-1]) == 1, tiles_p[-1]


def _pool_and_droplets(size, pool_top, n_drops, seed, clusters=True):
    """A pool wall to wall + droplets in the air above it: single cells and small clusters (some of them across a tile face)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = size
    cells = rng.integers([1, pool_top + 3, 1], [nx - 2, ny - 2, nz - 2], size=(n_drops, 3))
    if clusters:
        grow = cells[::3] + rng.integers(0, 2, size=(len(cells[::3]), 3))  # a second cell beside every third droplet
        cells = np.concatenate([cells, grow, np.array([[7, pool_top + 5, 8], [8, pool_top + 5, 8]])])  # one pair across a tile face (x = 7 | 8)
    cells = np.unique(cells, axis=0)
    sub = np.array([(a, b, c) for a in (0.25, 0.75) for b in (0.25, 0.75) for c in (0.25, 0.75)])
    pos = (cells[:, None, :] + sub[None, :, :]).reshape(-1, 3)
    spray = np.zeros(len(pos), dtype=lfa.PARTICLE_DTYPE)
    spray["pos"] = pos
    spray["vel"] = np.repeat(rng.normal(size=(len(cells), 3)) * 30.0, 8, axis=0) + rng.normal(size=pos.shape) * 3.0
    pool = util.scenes.seed_block((0, 0, 0), (nx, pool_top, nz))
    pool["vel"] = rng.normal(size=(len(pool), 3)) * 2.0
    return np.concatenate([pool, spray]), len(cells)


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
def test_closed_tiles_are_solved_on_their_own(dtype, monkeypatch):
    """Round 6: a particle tile whose unknowns couple to nothing outside the tile (spray: at step 550 of the C3 run a third of the
    tiles) is a block of the pressure matrix on its own. It is solved once, exactly (k_mg_solve_closed), and stays out of the tile
    list the PCG iterates over. What must hold: fewer tiles in the iteration, the same pressures as with LFA_MG_NO_CLOSED=1 and as
    the ORACLE's (the reference's PCG solves the droplets with everything else) under the usual bars - a droplet pair ACROSS a tile
    face is not closed and stays in the PCG -, face and particle velocities after the whole hot pass likewise; and the early-out of a
    zero right-hand side still returns p = 0 exactly."""
    from oracle import loader as orc
    size = (40, 40, 32)
    parts, n_drops = _pool_and_droplets(size, 10, 90, 3)
    o = orc.CpuSim(size, method=orc.APIC)
    o.set_particles(parts)
    po, reso, ito = o.hot_step(util.DT)
    oc = o.cells()
    res = []
    for closed in (True, False):
        monkeypatch.delenv("LFA_MG_NO_CLOSED", raising=False)
        if not closed:
            monkeypatch.setenv("LFA_MG_NO_CLOSED", "1")
        s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
        s.upload_particles(parts)
        r, it, rc = s.step_hot(util.DT)
        assert rc == 0 and r < 1e-6
        assert np.array_equal(s.fluid_cells(), o.fluid_cells())
        p = s.pressure()
        util.assert_close(p, po, P_REL, f"pressure, closed tiles {'out' if closed else 'in'}", pw=(1e-3, 1e-4))
        gc = s.cells()
        assert np.array_equal(gc["type"], oc["type"])
        util.assert_close(gc["vel"], oc["vel"], 1e-4, "face velocities", atol=1e-5 * 981.0 * util.DT)
        res.append((it, p, s.mg_level_tiles(), s.counts()["particle_tiles"]))
        s.close()
    (it_c, p_c, tiles_c, nt_c), (it_n, p_n, tiles_n, nt_n) = res
    assert nt_c == nt_n == tiles_n[0] and tiles_c[0] < tiles_n[0] - 20, (tiles_c, tiles_n)  # dozens of droplet tiles left the iteration
    assert abs(it_c - it_n) <= 1, (it_c, it_n)
    # the droplets' own pressures (small against the pool's): the two device runs agree to the solvers' tolerance
    util.assert_close(p_c, p_n, 1e-5, "pressure with / without the closed tiles in the PCG", pw=(2e-3, 1e-6))
    # lfa_apply_preconditioner with closed tiles: their share of M^-1 is their exact inverse - the operator stays symmetric
    # positive definite, and a residual that lives on the droplets alone comes back solved (A z = r there)
    monkeypatch.delenv("LFA_MG_NO_CLOSED", raising=False)
    s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    s.upload_particles(parts)
    s.hash(); s.p2g(); s.add_gravity(util.DT); s.build_system(util.DT)
    fc = s.fluid_cells().astype(np.int64)
    n = len(fc)
    rng = np.random.default_rng(12)
    x, y = rng.normal(size=n), rng.normal(size=n)
    mx, my = s.apply_precon(x), s.apply_precon(y)
    assert abs(y @ mx - x @ my) <= (2e-5 if dtype == lfa.PCG_F32 else 1e-11) * (abs(y @ mx) + np.linalg.norm(x) * np.linalg.norm(my))
    assert x @ mx > 0 and y @ my > 0
    high = (fc // size[0]) % size[1] >= 16  # unknowns in the tile layers above the pool's: droplets
    r = np.where(high, rng.normal(size=n), 0.0)
    z = s.apply_precon(r)
    az = s.apply_a(z)
    closed_rows = high & (np.abs(az - r) < 1e-4 * np.abs(r).max())
    assert closed_rows.sum() > 0.7 * high.sum(), (int(closed_rows.sum()), int(high.sum()))  # (droplets across a tile face are smoothed, not solved)
    s.close()
    # a block in free fall: divergence-free, the reference returns p = 0 without iterating (src/pressure_solver.cpp:33-35)
    s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    drops, _ = _pool_and_droplets(size, 0, 40, 4)
    drops = drops[drops["pos"][:, 1] > 3.0]
    drops["vel"] = (3.0, -20.0, 1.0)
    s.upload_particles(drops)
    r, it, rc = s.step_hot(util.DT)
    assert rc == 0 and it == 0 and not s.pressure().any()
    s.close()
    o.close()


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
def test_nothing_but_closed_tiles_needs_no_iteration(dtype):
    """Spray alone: every particle tile is closed, the PCG has no tile to iterate over - zero iterations, and the droplets' pressures
    (each its own little system, solved by k_mg_solve_closed) are the oracle's."""
    from oracle import loader as orc
    size = (40, 40, 32)
    parts, _ = _pool_and_droplets(size, 0, 60, 8, clusters=False)
    parts = parts[parts["pos"][:, 1] > 2.0]  # (pool_top = 0: no pool)
    o = orc.CpuSim(size, method=orc.APIC)
    o.set_particles(parts)
    po, reso, ito = o.hot_step(util.DT)
    assert ito > 0 and np.abs(po).max() > 0
    s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    s.upload_particles(parts)
    r, it, rc = s.step_hot(util.DT)
    assert rc == 0 and it == 0, (rc, it)
    assert np.array_equal(s.fluid_cells(), o.fluid_cells())
    util.assert_close(s.pressure(), po, P_REL, "droplet pressures", pw=(1e-3, 1e-4))
    util.assert_close(s.cells()["vel"], o.cells()["vel"], 1e-4, "face velocities", atol=1e-5 * 981.0 * util.DT)
    assert s.mg_level_tiles() in ([], [0]) and s.counts()["particle_tiles"] > 20
    s.close(); o.close()


def test_multigrid_single_launch_coarse_levels_repeat_bitwise_over_many_solves():
    """The hand-off between the phases of k_mg_coarse is a race if it is wrong: 40 solves of the same system must all return the
    same bits."""
    s = lfa.Sim((96, 64, 80), precond=lfa.PRECOND_MULTIGRID)
    s.seed_block((0, 0, 0), (60, 40, 50))
    r, it0, rc = s.step_hot(util.DT)
    assert rc == 0
    ref = None
    for _ in range(40):
        p, r, it, rc = s.solve(util.DT)
        assert rc == 0
        if ref is None:
            ref = (it, p)
        assert it == ref[0] and np.array_equal(p, ref[1])
    s.close()


@pytest.mark.parametrize("method", [lfa.FLIP_BLEND, lfa.PIC])
def test_pic_flip_keep_c_in_its_home_array_through_steps_sources_and_a_change_to_apic(method):
    """PIC / FLIP never change C (src/simulation.cpp:336-341,515-556: their transfers neither read nor write cx, cy, cz), the
    hosts' records just carry it. The device parks it in an array indexed by the particle id instead of moving 36 bytes per
    particle with every binning (lfa_sim::c_home). What must hold: a download returns every particle's C unchanged after full
    steps; a coercing source zeroes the C of the particles in its cells and the particles it seeds have C = 0; a change to APIC
    finds the C where APIC reads it: the C matrices its first G2P writes are those of a handle that was given the same particles
    (C included) as an upload."""
    size = (24, 24, 24)
    parts = util.scenes.seed_block((2, 2, 2), (14, 16, 12))
    rng = np.random.default_rng(11)
    for k in ("cx", "cy", "cz"):
        parts[k] = rng.normal(size=(len(parts), 3)).astype(np.float32)  # exactly representable on the device
    cells = np.array([(x, y, z) for x in (3, 4) for y in (3, 4, 5) for z in (3, 4)], dtype=np.int32)
    for _once in (0,):
        s = lfa.Sim(size, method=method, blending=0.95)
        s.upload_particles(parts)
        for _ in range(3):
            res, it, rc = s.time_step(0.004)
            assert rc == 0
        out = s.download_particles(into=parts.copy(), write_positions=True)
        for k in ("cx", "cy", "cz"):
            assert np.array_equal(out[k], parts[k]), k  # downloads come out in upload order (single domain)
        # a coercing source that also seeds
        s.add_source(cells, velocity=(30.0, 0.0, 5.0), density_cubic_root=3, active=True, coerce_velocity=True)
        for _ in range(2):
            res, it, rc = s.time_step(0.004)
            assert rc == 0
        n = s.num_particles
        assert n > len(parts)
        out = s.download_particles(into=np.zeros(n, dtype=lfa.PARTICLE_DTYPE), write_positions=True)
        old, new = out[:len(parts)], out[len(parts):]
        assert np.abs(np.concatenate([new["cx"], new["cy"], new["cz"]], axis=1)).max() == 0.0
        zeroed = np.abs(np.concatenate([old["cx"], old["cy"], old["cz"]], axis=1)).max(axis=1) == 0.0
        assert 0 < zeroed.sum() < len(parts)
        kept = ~zeroed
        assert np.array_equal(old["cx"][kept], parts["cx"][kept])
        # on to APIC: its P2G reads C in particle order. Reference run: a fresh APIC handle given the state as an upload.
        before = s.download_particles(into=np.zeros(n, dtype=lfa.PARTICLE_DTYPE), write_positions=True)
        s.clear_sources()
        s.set_params(simulation_method=lfa.APIC)
        res, it, rc = s.time_step(0.004)
        assert rc == 0
        a = s.download_particles(into=np.zeros(n, dtype=lfa.PARTICLE_DTYPE), write_positions=True)
        s.close()
        t = lfa.Sim(size, method=lfa.APIC, blending=0.95)
        t.upload_particles(before)
        res, it, rc = t.time_step(0.004)
        assert rc == 0
        b = t.download_particles(into=np.zeros(n, dtype=lfa.PARTICLE_DTYPE), write_positions=True)
        t.close()
        for k in ("cx", "cy", "cz"):  # same up to the summation order of the P2G that read them
            assert np.abs(a[k] - b[k]).max() <= 1e-4 * np.abs(b[k]).max(), k
        assert np.abs(a["pos"] - b["pos"]).max() < 1e-3 and np.abs(a["vel"] - b["vel"]).max() < 0.5
