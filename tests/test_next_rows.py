"""SURVEY.md 8(f) rank 1 -- the per-step particle stages around the hot path (advection, collision, position correction)
and the device-resident full time step, to the same bar as the hot path: oracle pinned against the real reference's
outputs (golden vectors), HIP path against the golden vectors through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import libfluid_amd as lfa
from oracle import loader as orc
from tests import util
from tests.test_host_class import FULLSTEP, fullstep_inputs

DT_NEXT = 0.02
DT_CORR = 0.1  # large enough for the springs to push particles into the sphere's skin and the walls


def next_inputs():
    c, parts, solid = fullstep_inputs()
    rng = np.random.default_rng(3)
    parts = parts.copy()
    parts["vel"] = rng.normal(size=(len(parts), 3)) * 80.0  # up to ~7 cells per step: several cells crossed, sphere and walls hit
    return c, parts, solid


def by_id(out):
    ids = np.rint(out["cx"][:, 0]).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(len(out)))
    return out[np.argsort(ids)]


def run_next_cpu(kind):
    """Stage A: advect + collide. Stage B: correct + collide from the seeded (coincidence-free) positions: particles piled
    up on a wall by stage A coincide, and for coincident pairs the reference adds a std::random_device jitter
    (src/simulation.cpp:584-587), so its output after A+B is not reproducible even against itself."""
    c, parts, solid = next_inputs()
    s = orc.CpuSim(c["size"], method=c["method"], blending=c["blend"], kind=kind)
    s.set_solid_cells(solid)
    s.set_particles(parts)
    s.hash()
    s.L.advect(s.h, DT_NEXT)
    s.L.detect_collisions(s.h)
    rec = {"after_advect_collide": by_id(s.particles())["pos"].copy()}
    s.set_particles(parts)
    s.hash()
    s.L.correct_positions(s.h, DT_CORR)
    s.L.detect_collisions(s.h)
    rec["after_correct_collide"] = by_id(s.particles())["pos"].copy()
    s.close()
    return rec


def test_oracle_next_stages_match_golden():
    g = util.load_golden("next_stages")
    got = run_next_cpu("oracle")
    for k in g:
        util.assert_close(got[k], g[k], 1e-13, k)
    # the scene really exercises the collision code: some particles were stopped by the solid sphere / the walls
    c, parts, _ = next_inputs()
    free_flight = np.clip(parts["pos"] + parts["vel"] * DT_NEXT, 0.1, np.array(c["size"]) - 0.1)
    assert (np.abs(g["after_advect_collide"] - free_flight).max(axis=1) > 1e-3).sum() > 20


def test_oracle_full_time_step_matches_golden():
    c, parts, solid = fullstep_inputs()
    g = util.load_golden("fullstep_flip")
    s = orc.CpuSim(c["size"], method=c["method"], blending=c["blend"])
    s.set_solid_cells(solid)
    s.set_particles(parts)
    its = []
    for _ in range(c["steps"]):
        res, it = C.c_double(0), C.c_uint64(0)
        s.L.time_step(s.h, c["dt"], C.byref(res), C.byref(it))
        its.append(it.value)
    out = by_id(s.particles())
    assert its == g["iters"].tolist()
    util.assert_close(out["pos"], g["pos"], 1e-12, "positions")
    util.assert_close(out["vel"], g["vel"], 1e-11, "velocities")


@pytest.mark.gpu
def test_device_split_move_and_collide_stages_against_the_oracle():
    """lfa_advect / lfa_correct / lfa_collide: the two moving stages with their collision handling split off, for hosts that
    install post_advection_callback / post_correction_callback (src/simulation.cpp:50-59,111-117). In between, the device reports
    what such a callback sees in the reference: the moved, not yet collided position and the position of before as old_position
    (oracle: advect / correct_positions without detect_collisions); after lfa_collide: the fused stages' result (golden)."""
    c, parts, solid = next_inputs()
    g = util.load_golden("next_stages")
    o = orc.CpuSim(c["size"], method=c["method"], blending=c["blend"])
    o.set_solid_cells(solid)
    o.set_particles(parts)
    o.hash()
    o.L.advect(o.h, DT_NEXT)
    mid = by_id(o.particles())
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    s.advect(DT_NEXT)
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - mid["pos"]).max() < 2e-5
    assert np.abs(out["old_pos"] - parts["pos"]).max() < 1e-6  # the position of before the move
    assert np.abs(out["pos"] - g["after_advect_collide"]).max() > 1e-3  # (the scene does collide)
    s.collide()
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - g["after_advect_collide"]).max() < 2e-5
    assert np.array_equal(out["pos"], out["old_pos"])
    # correction
    o.set_particles(parts)
    o.hash()
    o.L.correct_positions(o.h, DT_CORR)
    mid = by_id(o.particles())
    o.close()
    s.upload_particles(parts)
    s.hash()
    s.correct(DT_CORR)
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - mid["pos"]).max() < 5e-5
    assert np.abs(out["old_pos"] - parts["pos"]).max() < 1e-6
    s.collide()
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - g["after_correct_collide"]).max() < 5e-5
    # an upload between the move and lfa_collide replaces the particles: the collision pass starts from the uploaded positions
    s.upload_particles(parts)
    s.advect(DT_NEXT)
    s.upload_particles(parts)
    s.collide()
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - parts["pos"]).max() < 0.11  # at most the skin push-out (boundary_skin_width 0.1)
    s.close()


@pytest.mark.gpu
def test_device_advect_collide_and_correct_collide():
    c, parts, solid = next_inputs()
    g = util.load_golden("next_stages")
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    s.advect_collide(DT_NEXT)
    out = s.download_particles(into=parts.copy(), write_positions=True)
    # positions are (cell, fp32 fraction) on the device: 2^-23 of a cell, plus the fp32 velocity in x += v dt
    assert np.abs(out["pos"] - g["after_advect_collide"]).max() < 2e-5
    assert np.array_equal(out["pos"], out["old_pos"])
    s.upload_particles(parts)
    s.hash()
    s.correct_collide(DT_CORR)
    out = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(out["pos"] - g["after_correct_collide"]).max() < 5e-5
    assert np.abs(out["pos"] - parts["pos"]).max() > 1e-2  # the correction did move particles
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precond,dtype", [(lfa.PRECOND_MIC0_EXACT, lfa.PCG_F64), (lfa.PRECOND_MULTILEVEL, lfa.PCG_F32),
                                           (lfa.PRECOND_MULTIGRID, lfa.PCG_F32)])  # the last one is the default configuration
def test_device_time_step_matches_reference(precond, dtype):
    """Three device-resident simulation::time_step(dt) against the real reference's particles (fullstep_flip.npz)."""
    c, parts, solid = fullstep_inputs()
    g = util.load_golden("fullstep_flip")
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"], precond=precond, pcg_dtype=dtype)
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    its = []
    for _ in range(c["steps"]):
        res, it, rc = s.time_step(c["dt"])
        assert rc == 0
        its.append(it)
    out = s.download_particles(into=parts.copy(), write_positions=True)
    if precond == lfa.PRECOND_MIC0_EXACT:
        assert all(abs(a - b) <= 1 for a, b in zip(its, g["iters"].tolist()))
    util.assert_close(out["pos"], g["pos"], 1e-6, "positions after 3 device time steps", atol=3e-4)
    util.assert_close(out["vel"], g["vel"], 3e-4, "velocities after 3 device time steps")
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dense", ["all", "half", "second-pass"])
def test_position_correction_fallback_for_crowded_tiles(dense):
    """Half tiles whose 10 x 10 x 6-cell neighbourhood holds more particles than the LDS of the tiled kernel (5632) are redone by
    the same kernel with a CU's whole LDS to itself (12288), and what exceeds that by the global-gather kernel. 32 particles per
    cell ("all": every half tile), 48 in the lower part of the block only ("half": all three kernels run in one call and the
    later ones have to pick their particles by their keys from BEFORE the first one moved the others in place), or 16
    ("second-pass": nothing is left for the gather kernel). Checked against the oracle's _correct_positions + collisions."""
    size, lo, hi = (24, 24, 24), (2, 2, 2), (18, 16, 18)
    layers = {"all": 4, "half": 1, "second-pass": 2}[dense]
    parts = [util.scenes.seed_block(lo, hi, seed=util.scenes.SEED + 7 * k) for k in range(layers)]
    if dense == "half":
        parts += [util.scenes.seed_block(lo, (hi[0], 8, hi[2]), seed=util.scenes.SEED + 7 * k) for k in range(1, 6)]
    parts = np.concatenate(parts)
    parts["cx"][:, 0] = np.arange(len(parts))  # ids, like fullstep_inputs (by_id reads them back)
    cpu = orc.CpuSim(size, method=orc.PIC)
    cpu.set_particles(parts)
    cpu.hash()
    cpu.L.correct_positions(cpu.h, DT_CORR)
    cpu.L.detect_collisions(cpu.h)
    want = by_id(cpu.particles())["pos"].copy()
    cpu.close()
    s = lfa.Sim(size, method=lfa.PIC)
    s.upload_particles(parts)
    s.hash()
    s.correct_collide(DT_CORR)
    flagged, total, second = s.correction_stats_ex()
    assert (flagged, total) == s.correction_stats()
    if dense == "second-pass":
        assert flagged == 0 and 0 < second < total
    else:  # every kernel ran (also in "all": the sparsely filled half tiles at the block's edge fit the first pass)
        assert 0 < flagged < second < total
    out = s.download_particles(into=parts.copy(), write_positions=True)
    s.close()
    assert np.abs(out["pos"] - parts["pos"]).max() > 1e-2  # the correction did move particles
    assert np.abs(out["pos"] - want).max() < 5e-5


@pytest.mark.gpu
def test_position_correction_with_coincident_particles():
    """Coincident pairs get a random push in the reference (src/simulation.cpp:584-587, std::random_device) and a hashed one
    here: the branch-free pair walk of the LDS-tiled kernel detects them and redoes those particles with the branching walk.
    Every OTHER particle reads old positions only, so it still has to match the oracle; the twins have to come apart."""
    size, lo, hi = (24, 24, 24), (2, 2, 2), (18, 16, 18)
    parts = util.scenes.seed_block(lo, hi)
    rng = np.random.default_rng(11)
    twins = rng.choice(len(parts), 200, replace=False)
    parts = np.concatenate([parts, parts[twins]])
    parts["cx"][:, 0] = np.arange(len(parts))
    n0 = len(parts) - len(twins)
    cpu = orc.CpuSim(size, method=orc.PIC)
    cpu.set_particles(parts)
    cpu.hash()
    cpu.L.correct_positions(cpu.h, DT_CORR)
    cpu.L.detect_collisions(cpu.h)
    want = by_id(cpu.particles())["pos"].copy()
    cpu.close()
    s = lfa.Sim(size, method=lfa.PIC)
    s.upload_particles(parts)
    s.hash()
    s.correct_collide(DT_CORR)
    out = by_id(s.download_particles(into=parts.copy(), write_positions=True))
    s.close()
    single = np.ones(len(parts), bool)
    single[twins] = False
    single[n0:] = False
    assert np.abs(out["pos"][single] - want[single]).max() < 5e-5
    apart = np.abs(out["pos"][twins] - out["pos"][n0:]).max(axis=1)
    assert (apart > 1e-4).all() and np.isfinite(out["pos"]).all()
    # a twin feels its other partners like the oracle's does; the push itself is a unit-box vector times dt * stiffness * re
    assert np.abs(out["pos"][twins] - want[twins]).max() < 2.0 * DT_CORR * 5.0 * 0.7072 * np.sqrt(3.0) + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("method", [lfa.APIC, lfa.FLIP_BLEND, lfa.PIC])
def test_correct_collide_begin_end_undo(method):
    """lfa_correct_collide_begin / _end = lfa_correct_collide on the second stream, with grid-only stages beside it;
    _undo puts back the positions of before it, bit for bit (the correction keeps its inputs), also after a download has
    joined it; entry points that change positions or the binning make _undo refuse."""
    c, parts, solid = fullstep_inputs()
    s = lfa.Sim(c["size"], method=method, blending=c["blend"])
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    s.advect_collide(c["dt"]); s.hash(); s.p2g()
    before = s.download_particles(into=parts.copy(), write_positions=True)
    s.correct_collide_begin(DT_CORR)
    s.add_gravity(c["dt"]); s.solve(c["dt"]); s.apply_pressure(c["dt"])  # grid only: beside the correction
    s.correct_collide_end()
    moved = s.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(moved["pos"] - before["pos"]).max() > 1e-2
    with pytest.raises(lfa.LibfluidError):  # _end .. then a binning: nothing to take back any more
        s.hash()
        s.correct_collide_undo()
    # the same from the same state, serially
    t = lfa.Sim(c["size"], method=method, blending=c["blend"])
    t.set_solid_cells(solid)
    t.upload_particles(parts)
    t.advect_collide(c["dt"]); t.hash(); t.p2g(); t.add_gravity(c["dt"]); t.solve(c["dt"]); t.apply_pressure(c["dt"])
    t.correct_collide(DT_CORR)
    serial = t.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(moved["pos"] - serial["pos"]).max() < 2e-6  # (neighbour sums in the order of an atomically ordered binning)
    # undo: after grid-only stages beside it; and after a download has joined it - then either exact or refused (a download that
    # had to complete a deferred binning has overwritten the arrays the correction keeps its inputs in)
    for joined_first in (False, True):
        t.hash()
        here = t.download_particles(into=parts.copy(), write_positions=True)
        t.hash()  # (a deferred binning again: the download above completed the last one)
        t.correct_collide_begin(DT_CORR)
        if joined_first:
            mid = t.download_particles(into=parts.copy(), write_positions=True)
            assert np.abs(mid["pos"] - here["pos"]).max() > 1e-3
            try:
                t.correct_collide_undo()
            except lfa.LibfluidError:
                continue
        else:
            t.extrapolate()
            t.correct_collide_undo()
        back = t.download_particles(into=parts.copy(), write_positions=True)
        assert np.array_equal(back["pos"], here["pos"]) and np.array_equal(back["vel"], here["vel"])
        t.correct_collide(DT_CORR)  # and the stage still runs from there
    s.close()
    t.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", [lfa.APIC, lfa.FLIP_BLEND])
def test_time_step_with_the_correction_beside_the_solve_equals_the_serial_step(method):
    """lfa_time_step forks the position correction onto a second stream after the P2G and joins before the G2P
    (lfa_set_step_overlap): same kernels on the same data. Not bitwise - the binning orders the particles of a tile with
    atomics, so the correction's neighbour sums are taken in a different order from run to run: two SERIAL runs of two steps
    differ by 1e-6 cells / 4e-4 in velocities of O(30) as well (and after five, when particles piled up in a corner coincide
    and the hashed stand-in for the reference's random push depends on that order, by 0.1 cells for a few of them). A stage
    that read positions before the other stream had written them would be off by the correction's displacements, 1e-2 cells."""
    c, parts, solid = fullstep_inputs()
    outs = []
    for overlap in (True, False):
        s = lfa.Sim(c["size"], method=method, blending=c["blend"])
        s.set_solid_cells(solid)
        s.upload_particles(parts)
        s.set_step_overlap(overlap)
        s.enable_timing(True)
        for _ in range(2):
            _, it, rc = s.time_step(c["dt"])
            assert rc == 0 and it > 0
        assert s.step_timings()["overlapped"] == float(overlap)
        outs.append(s.download_particles(into=parts.copy(), write_positions=True))
        s.close()
    a, b = outs  # (download_particles keeps the upload order)
    assert np.abs(a["pos"] - b["pos"]).max() < 1e-5
    assert np.abs(a["vel"] - b["vel"]).max() < 3e-3
    assert np.abs(a["pos"] - parts["pos"]).max() > 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("method", [lfa.PIC, lfa.FLIP_BLEND, lfa.APIC])
def test_g2p_on_the_order_of_the_last_binning_equals_g2p_after_rebinning(method):
    """lfa_time_step bins once per step: after the position correction the G2P runs on the P2G-time order, and the
    particles whose cell has left their tile take the global-gather kernel. Same stages with and without a second
    binning before the G2P: the same velocities and C vectors per particle (same samples, same arithmetic)."""
    c, parts, solid = fullstep_inputs()
    outs = []
    for rebin in (True, False):
        s = lfa.Sim(c["size"], method=method, blending=0.9)
        s.set_solid_cells(solid)
        s.upload_particles(parts)
        s.hash(); s.p2g(); s.add_gravity(c["dt"])
        s.solve(c["dt"]); s.apply_pressure(c["dt"])
        s.correct_collide(DT_CORR)  # moves particles by up to a cell: some leave their tile
        s.extrapolate()
        if rebin:
            s.hash()
        s.g2p()
        outs.append(s.download_particles(into=parts.copy(), write_positions=True))  # upload order
        s.close()
    a, b = outs
    moved_tile = (np.floor(a["pos"] / 8) != np.floor(parts["pos"] / 8)).any(axis=1).sum()
    assert moved_tile > 50, "the scene is meant to push particles across tile faces"
    # (the correction's spring sums run in the order of an LDS atomic counter, so two runs agree to fp32 rounding only;
    # a particle that missed its transfer would keep its old velocity: O(1) off)
    assert np.abs(a["pos"] - b["pos"]).max() < 1e-5
    for f in ("vel", "cx", "cy", "cz") if method == lfa.APIC else ("vel",):
        scale = np.abs(a[f]).max()
        assert np.abs(a[f] - b[f]).max() < 2e-4 * scale, f
    assert np.abs(a["vel"] - parts["vel"]).max() > 1e-2 * np.abs(a["vel"]).max()  # the transfer did change velocities


@pytest.mark.gpu
@pytest.mark.parametrize("method", [lfa.APIC, lfa.FLIP_BLEND, lfa.PIC])
@pytest.mark.parametrize("seq", ["hash", "hash,hash", "hash,p2g", "hash,p2g,hash", "hash,correct", "hash,advect,hash",
                                 "hash,p2g,correct,hash,p2g", "hash,cfl"])
def test_deferred_binning_never_loses_velocities(seq, method):
    """The binning moves key, t and id (PIC / FLIP: and C); v (APIC: and C) follow lazily - the P2G reads them through the
    source index, the G2P rewrites them. Whatever is called in between, a download returns every particle's own v and C."""
    c, parts, solid = fullstep_inputs()
    parts = parts.copy()
    rng = np.random.default_rng(11)
    for f in ("cx", "cy", "cz"):
        parts[f] = rng.normal(size=(len(parts), 3))
    s = lfa.Sim(c["size"], method=method, blending=0.9)
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    for op in seq.split(","):
        if op == "hash":
            s.hash()
        elif op == "p2g":
            s.p2g()
        elif op == "correct":
            s.correct_collide(1e-3)
        elif op == "advect":
            s.advect_collide(1e-3)
        elif op == "cfl":
            s.cfl()
    out = s.download_particles(into=parts.copy())  # velocities and C only; upload order
    s.close()
    for f in ("vel", "cx", "cy", "cz"):
        assert np.array_equal(out[f].astype(np.float32), parts[f].astype(np.float32)), (seq, f)


@pytest.mark.gpu
@pytest.mark.parametrize("h,off,method", [(0.5, (0.3, -0.2, 0.1), lfa.APIC), (1.7, (-3.25, 0.7, 11.0), lfa.FLIP_BLEND)])
def test_particle_stages_and_full_steps_at_other_cell_sizes_and_offsets(h, off, method):
    """The stages around the hot path at cell_size != 1 and grid_offset != 0 (skin width, correction radius re = h / sqrt 2, the
    world-space clamp and the DDA in grid units all depend on them), each against the oracle from the same state, then three whole
    device time steps against the oracle's."""
    c, parts, solid = fullstep_inputs()
    parts = parts.copy()
    parts["pos"] = parts["pos"] * h + np.asarray(off)
    parts["old_pos"] = parts["pos"]
    rng = np.random.default_rng(4)
    parts["vel"] = rng.normal(size=(len(parts), 3)) * 60.0 * h

    def cpu():
        s = orc.CpuSim(c["size"], cell_size=h, offset=off, method=method, blending=0.95)
        s.set_solid_cells(solid)
        s.set_particles(parts)
        return s

    def gpu():
        s = lfa.Sim(c["size"], cell_size=h, offset=off, method=method, blending=0.95)
        s.set_solid_cells(solid)
        s.upload_particles(parts)
        return s

    def ids_of(p):
        # identity = the (unique) start position is gone after a move: the oracle keeps array order until its next hash, which
        # these stage calls do not run after moving, and the device downloads in upload order
        return p

    o, g = cpu(), gpu()
    o.hash()
    o.L.advect(o.h, DT_NEXT); o.L.detect_collisions(o.h)
    g.advect_collide(DT_NEXT)
    want = o.particles()  # hash() sorted them: match by the id carried in cx (FLIP/PIC) or by sorted coordinates (APIC zeroes nothing here)
    got = g.download_particles(into=parts.copy(), write_positions=True)
    want = want[np.argsort(np.rint(want["cx"][:, 0]).astype(np.int64))]
    assert np.abs(got["pos"] - want["pos"]).max() < 3e-5 * h
    o.close(); g.close()

    o, g = cpu(), gpu()
    o.hash(); g.hash()
    o.L.correct_positions(o.h, DT_CORR); o.L.detect_collisions(o.h)
    g.correct_collide(DT_CORR)
    want = o.particles()
    want = want[np.argsort(np.rint(want["cx"][:, 0]).astype(np.int64))]
    got = g.download_particles(into=parts.copy(), write_positions=True)
    assert np.abs(got["pos"] - want["pos"]).max() < 6e-5 * h
    assert np.abs(got["pos"] - parts["pos"]).max() > 1e-2 * h
    o.close(); g.close()

    o, g = cpu(), gpu()
    slow = parts.copy()
    slow["vel"] *= 0.05
    o.set_particles(slow); g.upload_particles(slow)
    for _ in range(3):
        o.L.time_step(o.h, 0.004, None, None)
        g.time_step(0.004)
    want, got = o.particles(), g.download_particles(into=slow.copy(), write_positions=True)
    if method == lfa.APIC:
        # cx is overwritten: compare the clouds as sorted coordinate sets. At h < 1 the reference's un-scaled APIC hat reaches
        # beyond the 27 cells its gather visits and is cut off there at weights of up to 0.25 (src/simulation.cpp:367-369 with
        # simulation.h:212-223): its own P2G jumps when a particle crosses a cell face, so a particle that the device's fp32
        # position puts on the other side of a face (2^-24 cells) changes the faces around it by O(1) - a handful of particles
        # after three steps (the first two steps agree to 1e-6). The bulk is held to the usual bar, the outliers to a loose one.
        for k in range(3):
            dp = np.abs(np.sort(got["pos"][:, k]) - np.sort(want["pos"][:, k]))
            dv = np.abs(np.sort(got["vel"][:, k]) - np.sort(want["vel"][:, k]))
            assert np.quantile(dp, 0.99) < 3e-4 * h and dp.max() < 0.05 * h, (k, np.quantile(dp, 0.99), dp.max())
            assert np.quantile(dv, 0.99) < 3e-3 * np.abs(want["vel"]).max(), (k, np.quantile(dv, 0.99))
    else:
        want = want[np.argsort(np.rint(want["cx"][:, 0]).astype(np.int64))]
        util.assert_close(got["pos"], want["pos"], 1e-6, "positions", atol=3e-4 * h)
        util.assert_close(got["vel"], want["vel"], 3e-4, "velocities", atol=1e-3 * h)
    o.close(); g.close()
