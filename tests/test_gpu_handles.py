"""Independent handles on one GPU, and the bound on every device-side wait (-m gpu).

The C ABI promises that different handles are independent (include/libfluid_amd.h) and the reference's hosts hold several
simulations at a time (one fluid::simulation per Maya GridNode, plugins/maya/nodes/grid_node.cpp:256). The V-cycle's coarse levels
run in ONE launch whose workgroups wait for each other (mg.hip: k_mg_coarse): two of those in flight from two handles, each half
resident, would wait for ever - so they are chained across handles (CoGate), and every wait has a ceiling (co_wait) after which
the solve is repeated on the launch-per-phase path instead of hanging the GPU."""
import threading

import numpy as np
import pytest

import libfluid_amd as lfa
from tests import util

pytestmark = pytest.mark.gpu

SIZE, BLOCKS = (64, 64, 64), [((0, 0, 0), (32, 40, 32)), ((8, 0, 8), (40, 36, 44)), ((0, 0, 16), (48, 24, 40)), ((16, 0, 0), (44, 44, 30))]
STEPS = 40


def run_one(block, overlap, out, k):
    try:
        s = lfa.Sim(SIZE)
        s.set_step_overlap(overlap)
        s.seed_block(*block)
        its = []
        for _ in range(STEPS):
            r, it, rc = s.time_step(min(3.0 * s.cfl(), 0.033))
            assert rc == 0, rc
            its.append(it)
        out[k] = (its, s.download_particles(), s.solver_stats())
        s.close()
    except BaseException as e:  # noqa: BLE001 - reported by the main thread
        out[k] = e


@pytest.mark.parametrize("overlap", [1, 0])
def test_four_independent_handles_step_concurrently(overlap):
    """Four simulations, four host threads, one GPU, 40 full time steps each (with and without the position correction on its own
    stream beside the solve): every run completes, no wait is given up, and each ends where the same simulation ends when it has
    the device to itself (full steps are not bit-reproducible - the correction sums in atomic order - so: particle count exact,
    iteration counts within one, positions to the run-to-run noise of a 40-step dam break)."""
    alone = [None] * len(BLOCKS)
    for k, b in enumerate(BLOCKS):
        run_one(b, overlap, alone, k)
        assert not isinstance(alone[k], BaseException), alone[k]
    together = [None] * len(BLOCKS)
    threads = [threading.Thread(target=run_one, args=(b, overlap, together, k)) for k, b in enumerate(BLOCKS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "a handle did not finish: device-side deadlock?"
    for k in range(len(BLOCKS)):
        assert not isinstance(together[k], BaseException), together[k]
        (ia, pa, sa), (ib, pb, sb) = alone[k], together[k]
        assert sb["device_waits_given_up"] == 0 and sa["device_waits_given_up"] == 0
        assert len(pa) == len(pb)
        assert np.isfinite(pb["pos"]).all()
        # Two runs of a 40-step splash drift apart by themselves (the correction's atomically ordered sums seed a chaotic flow:
        # the reference against itself from positions perturbed by 1e-6 cells drifts by half a cell over 120 steps):
        # the bulk - centre of mass, extents - agrees, every particle is where a particle can be
        assert np.abs(pa["pos"].mean(axis=0) - pb["pos"].mean(axis=0)).max() < 1.0, k
        assert np.abs(pa["pos"].max(axis=0) - pb["pos"].max(axis=0)).max() < 4.0, k
        assert pb["pos"].min() >= 0.0 and (pb["pos"] <= np.asarray(SIZE, dtype=float)).all()
        assert abs(np.median(ia) - np.median(ib)) <= 1


@pytest.mark.parametrize("dtype,flags", [(lfa.PCG_F32, False), (lfa.PCG_F32, True), (lfa.PCG_F64, False)])
def test_a_wait_that_is_never_answered_ends_in_a_repeated_solve_not_in_a_hang(dtype, flags, monkeypatch):
    """LFA_MG_CO_FAULT=1: workgroup 0 of k_mg_coarse never raises its first flag. Its neighbours' waits pass the ceiling (50 ms),
    the kernel leaves, the host finds the abort word at its next poll, retires the kernels that wait for this handle and repeats
    the solve on the launch-per-phase path: same iteration count and the same bits as a run that never used k_mg_coarse, one
    given-up wait in the solver statistics, and the handle keeps working."""
    size, block = (136, 72, 104), ((0, 0, 0), (90, 50, 70))
    res = []
    # (fp32: the tagged hand-off - the neighbours poll the words the faulty workgroup never writes - and, `flags`, the ready flags)
    if flags:
        monkeypatch.setenv("LFA_MG_NO_TAGGED", "1")
    else:
        monkeypatch.delenv("LFA_MG_NO_TAGGED", raising=False)
    for fault in (True, False):
        if fault:
            monkeypatch.setenv("LFA_MG_CO_FAULT", "1")
            monkeypatch.delenv("LFA_MG_NO_PERSIST", raising=False)
        else:
            monkeypatch.delenv("LFA_MG_CO_FAULT", raising=False)
            monkeypatch.setenv("LFA_MG_NO_PERSIST", "1")
        s = lfa.Sim(size, precond=lfa.PRECOND_MULTIGRID, pcg_dtype=dtype)
        s.seed_block(*block)
        its = []
        for _ in range(2):
            r, it, rc = s.step_hot(util.DT)
            assert rc == 0
            its.append(it)
        st = s.solver_stats()
        assert st["device_waits_given_up"] == (1 if fault else 0), st
        if fault:
            assert st["mg_first_level_in_coarse_launch"] == 0  # the second step ran launch-per-phase from the start
        res.append((its, s.pressure().copy()))
        s.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1])




def test_a_handle_recovers_from_a_nan_solve_and_from_the_kernel_bench():
    """Entries of the solver's arrays that are no unknowns are assumed to be zero and no kernel rewrites them; a non-finite value
    next to one makes it NaN for good (the smoother computes x + 0 * sum there). A solve that met a NaN, and lfa_bench_kernel's
    repeated launches (whose AXPYs overflow: round 6, `bench.py --config C4 --late 300` failed on the step after them), must not
    leave a handle that fails every later solve: the next system build re-creates the arrays (pcg.hip: pcg_scrub)."""
    size, block = (64, 64, 64), ((0, 0, 0), (32, 40, 32))
    ref = lfa.Sim(size)
    ref.seed_block(*block)
    s = lfa.Sim(size)
    s.seed_block(*block)
    for q in (ref, s):
        for _ in range(12):
            q.time_step(util.DT)
    good = s.download_particles()
    s.params.gravity[1] = float("nan")  # (the fixed-point P2G swallows a NaN particle velocity; gravity reaches every face)
    s.set_params()
    with pytest.raises(lfa.LibfluidError, match="NaN"):
        s.step_hot(util.DT)
    s.params.gravity[1] = -981.0
    s.set_params()
    s.upload_particles(good)
    for name in ("pcg_a", "mg_axpy_presmooth", "mg_down0", "mg_coarse", "mg_up0"):  # bench.py's sequence
        _, it0, rc0 = s.step_hot(util.DT) if name == "pcg_a" else (0, 1, 0)
        assert rc0 == 0 and it0 > 0
        s.bench_kernel(name, 20)
    ref.upload_particles(good)  # (same state on both handles: the reference handle steps from the same particles)
    ref.step_hot(util.DT)
    want = [ref.time_step(util.DT)[1] for _ in range(3)]
    got = [s.time_step(util.DT) for _ in range(3)]
    assert all(rc == 0 for _, _, rc in got)
    assert all(abs(it - w) <= 1 for (_, it, _), w in zip(got, want)), (got, want)
    assert np.isfinite(s.download_particles()["vel"]).all()
    s.close()
    ref.close()
