"""CPU tests: the oracle (oracle/oracle.c) against the golden vectors generated from the real reference, against the
real reference itself when it is buildable here, and known-answer checks of the hot path."""
import numpy as np
import pytest

from tests import util
from oracle import loader as orc

# fp64 restatement vs fp64 reference: the only licence is summation order inside a cell (std::sort is unstable,
# src/simulation.cpp:269), i.e. a few ulp.
REL = 1e-11


@pytest.mark.parametrize("name", sorted(util.CASES))
def test_oracle_matches_golden(name):
    got = util.staged_cpu_run(name, "oracle")
    util.assert_same_record(got, util.load_golden(name), REL, name)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("name", ["apic16_solid", "flip16", "pic_ragged"])
def test_oracle_matches_live_reference(name):
    util.assert_same_record(util.staged_cpu_run(name, "oracle"), util.staged_cpu_run(name, "ref"), REL, name)


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("name", sorted(util.CASES))
def test_golden_files_are_reference_outputs(name):
    """Every committed staged fixture is reproducible from the reference, bit for bit (guards against stale or hand-edited files)."""
    util.assert_same_record(util.staged_cpu_run(name, "ref"), util.load_golden(name), 0.0, name)


def test_hydrostatic_pressure_known_answer():
    """Tank at rest: p = rho |g| depth * dt-scaling drops out -> after one solve p_i = rho*|g|*h*(rows above + 1/2)...
    the discrete answer is p(y) = g*(H - y - 1/2) for cell row y of a column of H fluid cells under an air cell
    (ghost pressure 0 at the air cell centre), independent of x,z."""
    c, parts, _ = util.make_case("apic_tank")
    s = orc.CpuSim(c["size"], method=c["method"])
    s.set_particles(parts)
    p, res, it = s.hot_step(util.DT)
    fc = s.fluid_cells().astype(np.int64)
    y = (fc // c["size"][0]) % c["size"][1]
    H = 8
    want = 981.0 * (H - y)  # free surface: air cell centre is one cell above the top fluid cell centre
    assert it > 0 and res < 1e-6
    np.testing.assert_allclose(p, want, rtol=1e-6)
    # and the projected velocity field is at rest: particles keep zero velocity
    v = s.particles()["vel"]
    assert np.abs(v).max() < 1e-5


def test_p2g_g2p_uniform_translation_identity():
    """Uniform velocity field: APIC P2G followed by G2P returns the same velocity away from walls, C = 0."""
    size = (16, 16, 16)
    parts = util.scenes.seed_block((4, 4, 4), (12, 12, 12))
    parts["vel"] = np.array([1.5, -2.0, 0.75])
    s = orc.CpuSim(size, method=util.APIC, gravity=(0, 0, 0))
    s.set_particles(parts)
    s.hash(); s.p2g(); s.build_system(util.DT); s.extrapolate(); s.g2p()
    out = s.particles()
    inner = np.all((out["pos"] > 5.0) & (out["pos"] < 11.0), axis=1)
    np.testing.assert_allclose(out["vel"][inner], np.broadcast_to([1.5, -2.0, 0.75], out["vel"][inner].shape), rtol=1e-12)
    assert np.abs(np.concatenate([out["cx"], out["cy"], out["cz"]], axis=1)[inner]).max() < 1e-12


def test_divergence_free_after_projection():
    c, parts, solid = util.make_case("apic16_solid")
    s = orc.CpuSim(c["size"], method=c["method"])
    s.set_solid_cells(solid)
    s.set_particles(parts)
    s.hash(); s.p2g(); s.add_gravity(util.DT); s.build_system(util.DT)
    p, res, it = s.solve(util.DT)
    s.apply_pressure(util.DT, p)
    s.build_system(util.DT)
    assert np.abs(s.b()).max() < 1e-4  # rhs of the projected field = its divergence


def test_cfl_is_inf_at_rest():
    s = orc.CpuSim((8, 8, 8))
    s.set_particles(util.scenes.seed_block((0, 0, 0), (2, 2, 2)))
    assert np.isinf(s.cfl())


def test_empty_particle_set():
    s = orc.CpuSim((8, 8, 8))
    s.set_particles(np.zeros(0, dtype=orc.PARTICLE_DTYPE))
    p, res, it = s.hot_step(util.DT)
    assert len(p) == 0 and it == 0 and res == 0.0
