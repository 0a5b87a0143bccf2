"""The C++17 host class (libfluid_amd/host/simulation.h) that mirrors fluid::simulation.

CPU: it compiles with g++ against include/libfluid_amd.h and links to libfluid_amd.so.
GPU: full `time_step`s through the class - with the testbed's callbacks installed (testbed/main.cpp:101-123), with none, with
callbacks that edit device-resident state, with solid cells edited between steps, at a cell size other than 1 - against the REAL
reference's `simulation::time_step` (golden vector tests/golden/fullstep_flip.npz generated through oracle/_ref), against the
oracle, or against the same sequence of C-ABI calls made from Python."""
import os
import re
import subprocess

import numpy as np
import pytest

import libfluid_amd as lfa
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER_SRC = os.path.join(ROOT, "tests", "host_sim_driver.cpp")

FULLSTEP = dict(size=(20, 20, 20), block=((0, 0, 0), (10, 12, 10)), method=util.FLIP, blend=0.95,
                solid=((14, 3, 5), 3.3), dt=0.01, steps=3)


def fullstep_inputs():
    c = FULLSTEP
    parts = util.scenes.seed_block(*c["block"])
    parts["cx"][:, 0] = np.arange(len(parts))  # PIC/FLIP never touch cx: it carries the particle identity
    solid = util.scenes.sphere_solid_cells(c["size"], *c["solid"])
    return c, parts, solid


def build_driver(tmp_path):
    exe = str(tmp_path / "host_sim_driver")
    lfa.load_library()
    cmd = ["g++", "-std=c++17", "-O2", "-fopenmp", "-Wall", "-Wextra", *os.environ.get("LFA_HOST_CXXFLAGS", "").split(), "-o", exe, DRIVER_SRC,
           "-L" + os.path.dirname(lfa.LIB_PATH), "-l:libfluid_amd.so", "-Wl,-rpath," + os.path.dirname(lfa.LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run_driver(tmp_path, exe, c, parts, solid, mode, steps=None, cell_size=None, method=None):
    fin, fout, fsol = tmp_path / f"in_{mode}.bin", tmp_path / f"out_{mode}.bin", tmp_path / f"solid_{mode}.bin"
    parts.tofile(fin)
    if solid is not None:
        np.ascontiguousarray(solid, dtype=np.int32).tofile(fsol)
    args = [exe, *map(str, c["size"]), str(c["method"] if method is None else method), str(c["blend"]), str(c["dt"]),
            str(c["steps"] if steps is None else steps), str(fin), str(fout), str(fsol) if solid is not None else "-", mode]
    if cell_size is not None:
        args.append(str(cell_size))
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.fromfile(fout, dtype=lfa.PARTICLE_DTYPE)
    ms = float(re.search(r"step_ms ([0-9.]+)", r.stdout).group(1))
    return out, r.stdout, ms


def by_id(out, n):
    ids = np.rint(out["cx"][:, 0]).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(n))
    return out[np.argsort(ids)]


def test_host_class_compiles_and_links(tmp_path):
    build_driver(tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["callbacks", "two", "nocb"])
def test_full_time_steps_match_reference(tmp_path, mode):
    """callbacks: the testbed's three callbacks (stage-by-stage device calls, lazy downloads); two: without the one that reads
    particles(); nocb: one lfa_time_step per step. All three: the reference's particles after three steps."""
    c, parts, solid = fullstep_inputs()
    g = util.load_golden("fullstep_flip")
    exe = build_driver(tmp_path)
    out, stdout, _ = run_driver(tmp_path, exe, c, parts, solid, mode)
    assert len(out) == len(parts)
    out = by_id(out, len(parts))
    # three steps of a dam break: velocities O(30), displacements O(0.5) cells; fp32 device stages vs fp64 reference
    util.assert_close(out["pos"], g["pos"], 1e-6, "positions after 3 full steps", atol=3e-4)
    util.assert_close(out["vel"], g["vel"], 3e-4, "velocities after 3 full steps")
    # raw_cell_index is the cell of the position the class hands back
    cell = np.minimum(np.floor(out["pos"]).astype(np.int64), np.asarray(c["size"]) - 1)
    assert np.array_equal(out["raw"], cell[:, 0] + c["size"][0] * (cell[:, 1] + c["size"][1] * cell[:, 2]))
    if mode != "nocb":
        its = [int(m) for m in re.findall(r"(\d+) iterations", stdout)]
        assert len(its) == 3 and all(0 < i <= 40 for i in its)
        assert stdout.count("time step 0.01") == 3
    if mode == "callbacks":
        vmax = [float(m) for m in re.findall(r"max particle velocity = ([0-9.eE+-]+)", stdout)]
        assert len(vmax) == 3 and abs(vmax[-1] - np.sqrt((g["vel"] ** 2).sum(axis=1).max())) < 1e-2


@pytest.mark.gpu
def test_callbacks_do_not_evict_the_step_from_the_device(tmp_path):
    """A larger scene (64^3, 262 144 particles): with the testbed's first two callbacks a step costs what it costs without any
    (they touch nothing but n pressures); the third one reads particles() and pays for one download per step, nothing else."""
    cfg = dict(FULLSTEP, size=(64, 64, 64), block=((0, 0, 0), (32, 32, 32)), steps=8)
    parts = util.scenes.seed_block(*cfg["block"])
    parts["cx"][:, 0] = np.arange(len(parts))
    exe = build_driver(tmp_path)
    ms = {}
    for mode in ("nocb", "two", "callbacks"):
        best = min(run_driver(tmp_path, exe, cfg, parts, None, mode)[2] for _ in range(2))
        ms[mode] = best
    print("host class step ms:", ms)
    assert ms["two"] <= 1.10 * ms["nocb"] + 0.35, ms      # + the extra launches / event syncs of the staged path (measured +0.2 ms; a
                                                            # timing on a shared box: the margin keeps the check from flaking, a step
                                                            # that left the device costs tens of ms)
    assert ms["callbacks"] <= ms["two"] + 25.0, ms          # one 40 MB download + a max over it on one core


def staged_python_replica(c, parts, solid, steps, edit):
    """The stage sequence of simulation::time_step (src/simulation.cpp:43-125) over the C ABI from Python, with the same edits
    the driver's `edit` callbacks make."""
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    if solid is not None:
        s.set_solid_cells(solid)
    s.upload_particles(parts)
    host = parts.copy()
    for _ in range(steps):
        s.advect_collide(c["dt"]); s.hash(); s.p2g(); s.add_gravity(c["dt"])
        if edit:
            cells = s.cells()
            cells["vel"][1 + c["size"][0] * (1 + c["size"][1] * 1)] = 0.0
            s.upload_cells(cells)
        s.solve(c["dt"]); s.apply_pressure(c["dt"]); s.correct_collide(c["dt"]); s.extrapolate(); s.g2p()
        if edit:
            host = s.download_particles(into=host, write_positions=True)
            host["vel"][0] *= 0.5
            s.upload_particles(host)
    out = s.download_particles(into=host.copy(), write_positions=True)
    s.close()
    return out


@pytest.mark.gpu
def test_edits_through_mutable_references_reach_the_device(tmp_path):
    c, parts, solid = fullstep_inputs()
    exe = build_driver(tmp_path)
    out, _, _ = run_driver(tmp_path, exe, c, parts, solid, "edit")
    want = staged_python_replica(c, parts, solid, c["steps"], True)
    plain = staged_python_replica(c, parts, solid, c["steps"], False)
    out, want, plain = by_id(out, len(parts)), by_id(want, len(parts)), by_id(plain, len(parts))
    util.assert_close(out["vel"], want["vel"], 1e-6, "velocities with edits", atol=1e-6)
    util.assert_close(out["pos"], want["pos"], 1e-7, "positions with edits", atol=1e-6)
    assert np.abs(want["vel"][0] - plain["vel"][0]).max() > 1e-3  # and the edits did change the run


@pytest.mark.gpu
def test_callback_between_p2g_and_correction_sees_uncorrected_particles(tmp_path):
    """With stage callbacks the class starts the position correction on the device's second stream right after the P2G
    (lfa_correct_collide_begin). A callback that asks for particles() before the reference's correction stage
    (post_apply_pressure here) must still see - and may edit - the positions of BEFORE it: the class takes the correction
    back (lfa_correct_collide_undo) and runs it at the reference's place. Against the same stage sequence made serially
    over the C ABI; an inexact take-back would be off by the correction's displacements (1e-2 cells)."""
    c, parts, solid = fullstep_inputs()
    exe = build_driver(tmp_path)
    out, stdout, _ = run_driver(tmp_path, exe, c, parts, solid, "window")
    seen = [float(m) for m in re.findall(r"window_pos_sum ([0-9.eE+-]+)", stdout)]
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    s.set_solid_cells(solid)
    s.upload_particles(parts)
    host, want_seen = parts.copy(), []
    for _ in range(c["steps"]):
        s.advect_collide(c["dt"]); s.hash(); s.p2g(); s.add_gravity(c["dt"]); s.solve(c["dt"]); s.apply_pressure(c["dt"])
        host = s.download_particles(into=host, write_positions=True)
        want_seen.append(float(host["pos"][:, 0].sum()))
        host["vel"][0] *= 0.5
        s.upload_particles(host)
        s.hash()
        s.correct_collide(c["dt"]); s.extrapolate(); s.g2p()
    want = s.download_particles(into=host.copy(), write_positions=True)
    s.close()
    assert len(seen) == c["steps"]
    # 9600 positions of O(10): a correction left in place would move the sum by ~1 (1e-4 cells per particle on average)
    assert np.allclose(seen, want_seen, rtol=0, atol=2e-3), (seen, want_seen)
    out, want = by_id(out, len(parts)), by_id(want, len(parts))
    util.assert_close(out["pos"], want["pos"], 1e-7, "positions", atol=2e-6)
    util.assert_close(out["vel"], want["vel"], 1e-6, "velocities", atol=2e-4)


@pytest.mark.gpu
def test_solid_cells_edited_between_steps(tmp_path):
    """Hosts put solids in and take them out through sim.grid() between steps (testbed scene reset, testbed/main.cpp:125-178;
    grid_node.cpp:330-339). Step 0 runs without the sphere, steps 1 .. n-2 with it, the last one without again."""
    c, parts, solid = fullstep_inputs()
    steps = 4
    exe = build_driver(tmp_path)
    out, _, _ = run_driver(tmp_path, exe, c, parts, solid, "obstacle", steps=steps)
    s = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    s.upload_particles(parts)
    for i in range(steps):
        if i == 1:
            s.set_solid_cells(solid)
        if i == steps - 1:
            s.clear_solid_cells()
        s.time_step(c["dt"])
    want = s.download_particles(into=parts.copy(), write_positions=True)
    s.close()
    out, want = by_id(out, len(parts)), by_id(want, len(parts))
    util.assert_close(out["pos"], want["pos"], 1e-7, "positions", atol=1e-6)
    util.assert_close(out["vel"], want["vel"], 1e-6, "velocities", atol=1e-6)
    # the obstacle mattered: a run that never sees it ends elsewhere
    t = lfa.Sim(c["size"], method=c["method"], blending=c["blend"])
    t.upload_particles(parts)
    for i in range(steps):
        t.time_step(c["dt"])
    free = by_id(t.download_particles(into=parts.copy(), write_positions=True), len(parts))
    t.close()
    assert np.abs(free["pos"] - want["pos"]).max() > 1e-3


@pytest.mark.gpu
def test_default_apic_at_other_cell_sizes_against_the_oracle(tmp_path):
    """The class's defaults (APIC with the reference's un-scaled hat) at cell_size 0.5 and 1.7: two whole steps against the
    oracle's time_step. (Round 1 returned LFA_E_UNSUPPORTED here and silently fell back to an un-projected step.)"""
    from oracle import loader as orc
    exe = build_driver(tmp_path)
    for h in (0.5, 1.7):
        c = dict(FULLSTEP, method=util.APIC, blend=1.0, steps=2, dt=0.004)
        parts = util.scenes.seed_block(*c["block"], cell_size=h)
        rng = np.random.default_rng(2)
        parts["vel"] = rng.normal(size=(len(parts), 3)) * 3.0 * h
        solid = util.scenes.sphere_solid_cells(c["size"], *c["solid"])
        out, _, _ = run_driver(tmp_path, exe, c, parts, solid, "two", cell_size=h)
        o = orc.CpuSim(c["size"], cell_size=h, method=orc.APIC)
        o.set_solid_cells(solid)
        o.set_particles(parts)
        for _ in range(c["steps"]):
            o.L.time_step(o.h, c["dt"], None, None)
        want = o.particles()
        assert len(out) == len(want)
        for k in range(3):  # APIC overwrites every identity-carrying field: compare the clouds as sorted coordinate sets
            util.assert_close(np.sort(out["pos"][:, k]), np.sort(want["pos"][:, k]), 1e-6, f"h={h} positions", atol=3e-4 * h)
            util.assert_close(np.sort(out["vel"][:, k]), np.sort(want["vel"][:, k]), 3e-4, f"h={h} velocities", atol=1e-3 * h)
        o.close()


@pytest.mark.gpu
def test_handle_recreation_is_cheap(tmp_path):
    """The Maya node builds a fresh simulation per evaluation (plugins/maya/nodes/grid_node.cpp:256-274,350-366): 100 x { new
    simulation, resize(64^3), 262 144 particles in, update(1/60), particles() out, destroy }. Device blocks, streams, events and
    the pinned page of the destroyed handle are adopted by the next one (csrc/pool.hip): lfa_create stays under a millisecond, and the results of the last cycle are those of the first."""
    size = (64, 64, 64)
    parts = util.scenes.seed_block((0, 0, 0), (32, 32, 32))
    assert len(parts) == 262144
    c = dict(size=size, method=util.APIC, blend=1.0, dt=1.0 / 60.0, steps=100)
    exe = build_driver(tmp_path)
    out, stdout, _ = run_driver(tmp_path, exe, c, parts, None, "recreate")
    m = re.search(r"cycle_ms ([0-9.]+) resize_ms ([0-9.]+) create_ms ([0-9.]+)", stdout)
    cycle_ms, resize_ms, create_ms = float(m.group(1)), float(m.group(2)), float(m.group(3))
    print(f"handle re-creation at 64^3 / 262 144 particles: {cycle_ms:.3f} ms per cycle, {resize_ms:.3f} ms of it in resize() "
          f"(host grid + space hash + their content hash), {create_ms:.3f} ms of that in lfa_create")
    assert create_ms < 1.0, (cycle_ms, resize_ms, create_ms)
    one, _, _ = run_driver(tmp_path, exe, dict(c, steps=1), parts, None, "recreate")
    assert len(out) == len(parts)
    util.assert_close(np.sort(out["pos"][:, 1]), np.sort(one["pos"][:, 1]), 1e-6, "heights after a re-created handle's update", atol=1e-5)


@pytest.mark.gpu
def test_block_cache_serves_a_second_handle_and_can_be_trimmed():
    lfa.pool_trim()
    s = lfa.Sim((48, 40, 56), method=lfa.APIC)
    s.seed_block((0, 0, 0), (24, 20, 28))
    r1 = s.step_hot(util.DT)
    p1 = s.pressure().copy()
    s.close()
    before = lfa.pool_stats()
    assert before["cached_blocks"] > 20 and before["cached_bytes"] > 0
    s = lfa.Sim((48, 40, 56), method=lfa.APIC)
    assert s.create_ms < 3.0, s.create_ms  # (one sample on a shared box; 0.15 ms as a rule)
    s.seed_block((0, 0, 0), (24, 20, 28))
    r2 = s.step_hot(util.DT)
    after = lfa.pool_stats()
    assert after["misses"] == before["misses"], "a second handle of the same size must not reach the driver's allocator"
    assert r1[1] == r2[1] and np.array_equal(p1, s.pressure())  # blocks come back dirty: every consumer initialises what it reads
    s.close()
    lfa.pool_trim()
    assert lfa.pool_stats()["cached_bytes"] == 0


@pytest.mark.gpu
def test_advection_and_correction_callbacks_sit_where_the_reference_has_them(tmp_path):
    """post_advection_callback / post_correction_callback (src/simulation.cpp:51-59,111-117) run between the move and its collision
    handling: they see particles whose position differs from old_position (the device keeps the positions of before a split
    move), and the step's result is that of the fused stages - the reference's particles after three steps."""
    c, parts, solid = fullstep_inputs()
    g = util.load_golden("fullstep_flip")
    exe = build_driver(tmp_path)
    out, stdout, _ = run_driver(tmp_path, exe, c, parts, solid, "placed")
    m = re.search(r"moved_advect ([0-9.e+-]+) moved_correct ([0-9.e+-]+)", stdout)
    moved_advect, moved_correct = float(m.group(1)), float(m.group(2))
    assert moved_advect > 1.0 and moved_correct > 1e-3, stdout  # summed over 1 200 particles and three steps
    out = by_id(out, len(parts))
    util.assert_close(out["pos"], g["pos"], 1e-6, "positions after 3 steps with the two callbacks placed", atol=3e-4)
    util.assert_close(out["vel"], g["vel"], 3e-4, "velocities after 3 steps with the two callbacks placed")
