// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// oracle/ref_harness.cpp : C entry points around the *real* lukedan/libfluid hot path.
//
// This file contains no reference code. It is a unity translation unit that #includes the reference's own
// sources where they lie under $REFERENCE_DIR (default /root/reference) and wraps them with `extern "C"`
// stage-level entry points, so that tests can (a) pin oracle/oracle.c against the true reference and
// (b) generate the golden vectors in tests/golden/ (tests/golden/make_golden.py).
// Output goes to oracle/_ref/libref.so (git-ignored). When /root/reference is absent (GPU box) this file is
// simply not built; nothing at run time depends on it.
//
// pcg32: reference include/fluid/simulation.h:11 does `#include <pcg_random.hpp>` from an un-vendored submodule
// (.gitmodules:1-3). The genuine pcg-cpp header ships inside this image's pyarrow wheel
// (.../pyarrow/include/arrow/vendored/pcg/pcg_random.hpp, namespace arrow_vendored); the Makefile puts that
// directory on the include path and the using-declaration below lifts pcg32 to the global namespace. No stand-in
// header is written. The RNG touches seeding and the coincident-particle jitter only (simulation.h:85,101,
// simulation.cpp:141-145,573,587); none of the hot-path arithmetic depends on it, and the harness injects
// particles rather than seeding them.
//
// Private/protected members of fluid::simulation / fluid::pressure_solver are reached with the
// `#define private public` technique after all standard headers have been included.

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <initializer_list>
#include <iostream>
#include <limits>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include <pcg_random.hpp>
using arrow_vendored::pcg32;

#define private public
#define protected public
#include "src/mac_grid.cpp"
#include "src/simulation.cpp"
#include "src/pressure_solver.cpp"
#undef private
#undef protected
// the reference's voxelizer and its obstacle host
#include "src/math/intersection.cpp"
#include "src/voxelizer.cpp"
#include "src/data_structures/obstacle.cpp"
// surface mesher (private members reached like the solver's)
#define private public
#include "src/mesher.cpp"
#undef private
// on-disk formats
#include "src/data_structures/point_cloud.cpp"
#include <sstream>

namespace {
	using fluid::vec3d;
	using fluid::vec3s;
	using sim_t = fluid::simulation;

	struct ref_ctx {
		sim_t sim;
		std::vector<vec3s> fluid_cells;
		std::unique_ptr<fluid::pressure_solver> solver;
		std::vector<double> precon;
	};

	static_assert(sizeof(sim_t::particle) == 152, "particle layout (SURVEY 8: 152-B AoS)");
	static_assert(sizeof(fluid::mac_grid::cell) == 32, "cell layout (SURVEY 8: 32-B AoS)");
}

extern "C" {
	void *ref_create(
		std::size_t nx, std::size_t ny, std::size_t nz, double cell_size,
		const double *offset, const double *gravity, int method, double blending, double density
	) {
		auto *c = new ref_ctx();
		c->sim.resize(vec3s(nx, ny, nz));
		c->sim.cell_size = cell_size;
		c->sim.grid_offset = vec3d(offset[0], offset[1], offset[2]);
		c->sim.gravity = vec3d(gravity[0], gravity[1], gravity[2]);
		c->sim.simulation_method = static_cast<sim_t::method>(method);
		c->sim.blending_factor = blending;
		c->sim.density = density;
		return c;
	}
	void ref_destroy(void *h) {
		delete static_cast<ref_ctx*>(h);
	}
	void ref_set_extrapolation_iterations(void *h, std::size_t n) {
		static_cast<ref_ctx*>(h)->sim.velocity_extrapolation_iterations = n;
	}

	void ref_set_solid_cells(void *h, const int *xyz, std::size_t k) {
		auto &g = static_cast<ref_ctx*>(h)->sim.grid().grid();
		for (std::size_t i = 0; i < k; ++i) {
			g(vec3s(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])).cell_type = fluid::mac_grid::cell::type::solid;
		}
	}

	void ref_set_particles(void *h, const void *aos152, std::size_t n) {
		auto &ps = static_cast<ref_ctx*>(h)->sim.particles();
		ps.resize(n);
		std::memcpy(static_cast<void*>(ps.data()), aos152, n * sizeof(sim_t::particle));
	}
	std::size_t ref_num_particles(void *h) {
		return static_cast<ref_ctx*>(h)->sim.particles().size();
	}
	void ref_get_particles(void *h, void *aos152) {
		auto &ps = static_cast<ref_ctx*>(h)->sim.particles();
		std::memcpy(aos152, static_cast<const void*>(ps.data()), ps.size() * sizeof(sim_t::particle));
	}

	void ref_get_cells(void *h, void *aos32) {
		auto &g = static_cast<ref_ctx*>(h)->sim.grid().grid();
		std::memcpy(aos32, static_cast<const void*>(&g[0]), g.get_array_size(g.get_size()) * 32);
	}
	void ref_set_cells(void *h, const void *aos32) {
		auto &g = static_cast<ref_ctx*>(h)->sim.grid().grid();
		std::memcpy(static_cast<void*>(&g[0]), aos32, g.get_array_size(g.get_size()) * 32);
	}
	void ref_get_old_cells(void *h, void *aos32) {
		auto &g = static_cast<ref_ctx*>(h)->sim._old_grid.grid();
		std::memcpy(aos32, static_cast<const void*>(&g[0]), g.get_array_size(g.get_size()) * 32);
	}

	/// simulation::update_and_hash_particles (src/simulation.cpp:251-264).
	void ref_hash(void *h) {
		static_cast<ref_ctx*>(h)->sim.update_and_hash_particles();
	}
	std::size_t ref_num_fluid_cells(void *h) {
		return static_cast<ref_ctx*>(h)->sim._fluid_cells.size();
	}
	void ref_get_fluid_cells(void *h, std::uint64_t *out) {
		auto &fc = static_cast<ref_ctx*>(h)->sim._fluid_cells;
		for (std::size_t i = 0; i < fc.size(); ++i) {
			out[i] = fc[i];
		}
	}
	/// _space_hash begin/count per cell (src/simulation.cpp:266-291).
	void ref_get_space_hash(void *h, std::uint64_t *begin, std::uint64_t *count) {
		auto &sh = static_cast<ref_ctx*>(h)->sim._space_hash;
		std::size_t n = sh.get_array_size(sh.get_size());
		for (std::size_t i = 0; i < n; ++i) {
			begin[i] = sh[i].begin;
			count[i] = sh[i].count;
		}
	}

	/// simulation::_transfer_to_grid (src/simulation.cpp:400-412).
	void ref_p2g(void *h) {
		static_cast<ref_ctx*>(h)->sim._transfer_to_grid();
	}
	/// gravity loop (src/simulation.cpp:72-78).
	void ref_add_gravity(void *h, double dt) {
		auto &sim = static_cast<ref_ctx*>(h)->sim;
		auto sz = sim.grid().grid().get_size();
		for (std::size_t z = 0; z < sz.z; ++z) {
			for (std::size_t y = 0; y < sz.y; ++y) {
				for (std::size_t x = 0; x < sz.x; ++x) {
					sim.grid().grid()(x, y, z).velocities_posface += sim.gravity * dt;
				}
			}
		}
	}

	/// fluid cell list + pressure_solver ctor + the set-up half of solve() (src/simulation.cpp:83-99,
	/// src/pressure_solver.cpp:20-25).
	void ref_build_system(void *h, double dt) {
		auto *c = static_cast<ref_ctx*>(h);
		c->fluid_cells.clear();
		for (std::size_t raw : c->sim._fluid_cells) {
			c->fluid_cells.emplace_back(c->sim.grid().grid().index_from_raw(raw));
		}
		c->solver = std::make_unique<fluid::pressure_solver>(c->sim, c->fluid_cells);
		auto &s = *c->solver;
		s._compute_fluid_cell_indices();
		s._a_scale = dt / (c->sim.density * c->sim.cell_size * c->sim.cell_size);
		s._compute_a_matrix();
		c->precon = s._compute_preconditioner();
	}
	void ref_set_pcg_params(void *h, double tau, double sigma, double tol, std::size_t maxit) {
		auto &s = *static_cast<ref_ctx*>(h)->solver;
		s.tau = tau;
		s.sigma = sigma;
		s.tolerance = tol;
		s.max_iterations = maxit;
	}
	/// cell_data as one byte: bits 0-2 nonsolid_neighbors, bit 3 fluid_xpos, bit 4 fluid_ypos, bit 5 fluid_zpos.
	void ref_get_abits(void *h, std::uint8_t *out) {
		auto &s = *static_cast<ref_ctx*>(h)->solver;
		for (std::size_t i = 0; i < s._a.size(); ++i) {
			out[i] = static_cast<std::uint8_t>(
				s._a[i].nonsolid_neighbors | (s._a[i].fluid_xpos << 3) | (s._a[i].fluid_ypos << 4) |
				(s._a[i].fluid_zpos << 5)
			);
		}
	}
	void ref_get_b(void *h, double *out) {
		auto b = static_cast<ref_ctx*>(h)->solver->_compute_b_vector();
		std::copy(b.begin(), b.end(), out);
	}
	void ref_get_precon(void *h, double *out) {
		auto &p = static_cast<ref_ctx*>(h)->precon;
		std::copy(p.begin(), p.end(), out);
	}
	void ref_apply_precon(void *h, const double *r, double *z) {
		auto *c = static_cast<ref_ctx*>(h);
		std::size_t n = c->fluid_cells.size();
		std::vector<double> rv(r, r + n), zv(n, 0.0), q(n, 0.0);
		c->solver->_apply_preconditioner(zv, q, c->precon, rv);
		std::copy(zv.begin(), zv.end(), z);
	}
	void ref_apply_a(void *h, const double *v, double *out) {
		auto *c = static_cast<ref_ctx*>(h);
		std::size_t n = c->fluid_cells.size();
		std::vector<double> vv(v, v + n), ov(n, 0.0);
		c->solver->_apply_a(ov, vv);
		std::copy(ov.begin(), ov.end(), out);
	}
	/// pressure_solver::solve (src/pressure_solver.cpp:19-71). Rebuilds the system like the reference does.
	void ref_solve(void *h, double dt, double *p, double *residual, std::uint64_t *iters) {
		auto *c = static_cast<ref_ctx*>(h);
		auto [pv, res, it] = c->solver->solve(dt);
		std::copy(pv.begin(), pv.end(), p);
		*residual = res;
		*iters = it;
	}
	/// pressure_solver::apply_pressure (src/pressure_solver.cpp:73-148).
	void ref_apply_pressure(void *h, double dt, const double *p) {
		auto *c = static_cast<ref_ctx*>(h);
		std::vector<double> pv(p, p + c->fluid_cells.size());
		c->solver->apply_pressure(dt, pv);
	}
	/// simulation::_extrapolate_velocities (src/simulation.cpp:685-754).
	void ref_extrapolate(void *h) {
		auto *c = static_cast<ref_ctx*>(h);
		c->sim._extrapolate_velocities(c->fluid_cells);
	}
	/// simulation::_transfer_from_grid (src/simulation.cpp:548-560).
	void ref_g2p(void *h) {
		static_cast<ref_ctx*>(h)->sim._transfer_from_grid();
	}
	double ref_cfl(void *h) {
		return static_cast<ref_ctx*>(h)->sim.cfl();
	}
	/// Per-step stages outside the hot path, exposed for the "next" rows.
	void ref_advect(void *h, double dt) {
		static_cast<ref_ctx*>(h)->sim._advect_particles(dt);
	}
	void ref_detect_collisions(void *h) {
		auto &sim = static_cast<ref_ctx*>(h)->sim;
		sim._detect_collisions();
		for (auto &p : sim._particles) {
			p.old_position = p.position;
		}
	}
	void ref_correct_positions(void *h, double dt) {
		static_cast<ref_ctx*>(h)->sim._correct_positions(dt);
	}
	/// Fluid sources (include/fluid/data_structures/source.h:12-22), appended to simulation::sources (simulation.h:179).
	void ref_clear_sources(void *h) {
		static_cast<ref_ctx*>(h)->sim.sources.clear();
	}
	void ref_add_source(void *h, const int *xyz, std::size_t k, const double *vel, std::size_t root, int active, int coerce) {
		auto src = std::make_unique<fluid::source>();
		for (std::size_t i = 0; i < k; ++i) {
			src->cells.emplace_back(vec3s(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
		}
		src->velocity = vec3d(vel[0], vel[1], vel[2]);
		src->target_density_cubic_root = root;
		src->active = active != 0;
		src->coerce_velocity = coerce != 0;
		static_cast<ref_ctx*>(h)->sim.sources.emplace_back(std::move(src));
	}
	/// simulation::_update_sources + the hash_particles that follows it in time_step (src/simulation.cpp:63-64,756-765).
	void ref_update_sources(void *h) {
		auto &sim = static_cast<ref_ctx*>(h)->sim;
		sim._update_sources();
		sim.hash_particles();
	}
	/// simulation::update(dt) (src/simulation.cpp:31-41): CFL sub-stepping. Returns the number of time steps taken and
	/// writes up to `cap` of their lengths to `dts` (pre_time_step_callback, include/fluid/simulation.h:153).
	std::size_t ref_update(void *h, double dt, double *dts, std::size_t cap) {
		auto &sim = static_cast<ref_ctx*>(h)->sim;
		std::size_t n = 0;
		sim.pre_time_step_callback = [&](double step) {
			if (dts && n < cap) {
				dts[n] = step;
			}
			++n;
		};
		sim.update(dt);
		sim.pre_time_step_callback = nullptr;
		return n;
	}
	/// Full simulation::time_step(dt) (src/simulation.cpp:43-125); pressure/residual/iters of the step are
	/// captured through post_pressure_solve_callback (include/fluid/simulation.h:166).
	void ref_time_step(void *h, double dt, double *residual, std::uint64_t *iters) {
		auto &sim = static_cast<ref_ctx*>(h)->sim;
		sim.post_pressure_solve_callback = [&](double, std::vector<double>&, double res, std::size_t it) {
			if (residual) {
				*residual = res;
			}
			if (iters) {
				*iters = it;
			}
		};
		sim.time_step(dt);
		sim.post_pressure_solve_callback = nullptr;
	}
}

// ---- voxelizer (src/voxelizer.cpp, src/data_structures/obstacle.cpp) -------------------------------------------------
namespace {
	fluid::obstacle::mesh_t make_mesh(const double *pos, std::size_t nv, const std::uint64_t *idx, std::size_t ni) {
		fluid::obstacle::mesh_t m;
		m.positions.resize(nv);
		for (std::size_t i = 0; i < nv; ++i) m.positions[i] = vec3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
		m.indices.assign(idx, idx + ni);
		return m;
	}
}
extern "C" {
	void ref_vox_grid(
		const double *pos, std::size_t nv, double cs, const double *ref_off, std::int32_t *grid_min, std::uint64_t *size,
		double *grid_off
	) {
		auto m = make_mesh(pos, nv, nullptr, 0);
		auto [rmin, rmax] = fluid::voxelizer::get_bounding_box(m.positions.begin(), m.positions.end());
		fluid::voxelizer vox;
		fluid::vec3i o = vox.resize_reposition_grid_constrained(rmin, rmax, cs, vec3d(ref_off[0], ref_off[1], ref_off[2]));
		vec3s n = vox.voxels.get_size();
		grid_min[0] = o.x; grid_min[1] = o.y; grid_min[2] = o.z;
		size[0] = n.x; size[1] = n.y; size[2] = n.z;
		grid_off[0] = vox.grid_offset.x; grid_off[1] = vox.grid_offset.y; grid_off[2] = vox.grid_offset.z;
	}
	void ref_voxelize(
		const double *pos, std::size_t nv, const std::uint64_t *idx, std::size_t ni, double cs, const double *ref_off,
		std::uint8_t *types
	) {
		auto m = make_mesh(pos, nv, idx, ni);
		auto [rmin, rmax] = fluid::voxelizer::get_bounding_box(m.positions.begin(), m.positions.end());
		fluid::voxelizer vox;
		vox.resize_reposition_grid_constrained(rmin, rmax, cs, vec3d(ref_off[0], ref_off[1], ref_off[2]));
		vox.voxelize_mesh_surface(m);
		vox.mark_exterior();
		vec3s n = vox.voxels.get_size();
		for (std::size_t i = 0; i < n.x * n.y * n.z; ++i) types[i] = static_cast<std::uint8_t>(vox.voxels[i]);
	}
	/// The cell lists of the Maya VoxelizerNode, gathered with the reference's own grid3::for_each: voxel-grid
	/// coordinates of the selected types (plugins/maya/nodes/voxelizer_node.cpp:285-323) or, with ref_size != NULL,
	/// reference-grid coordinates clipped to the reference grid (:325-343). Returns the count, fills at most `cap`
	/// triples. (fluid::obstacle, src/data_structures/obstacle.cpp:20-28, is not used as the golden source: it passes a
	/// max corner in reference-grid coordinates to for_each_in_range_unchecked on the voxel grid, which walks out of
	/// bounds whenever the voxel grid starts at a positive offset; no host of the reference constructs it.)
	std::size_t ref_voxel_cells(
		const double *pos, std::size_t nv, const std::uint64_t *idx, std::size_t ni, double cs, const double *ref_off,
		int include_interior, int include_surface, const std::int64_t *ref_size, std::int32_t *xyz, std::size_t cap
	) {
		auto m = make_mesh(pos, nv, idx, ni);
		auto [rmin, rmax] = fluid::voxelizer::get_bounding_box(m.positions.begin(), m.positions.end());
		fluid::voxelizer vox;
		fluid::vec3i o = vox.resize_reposition_grid_constrained(rmin, rmax, cs, vec3d(ref_off[0], ref_off[1], ref_off[2]));
		vox.voxelize_mesh_surface(m);
		vox.mark_exterior();
		std::size_t n = 0;
		vox.voxels.for_each([&](vec3s p, fluid::voxelizer::cell_type t) {
			bool take = (t == fluid::voxelizer::cell_type::interior && include_interior) ||
				(t == fluid::voxelizer::cell_type::surface && include_surface);
			if (!take) return;
			std::int64_t c[3] = {static_cast<std::int64_t>(p.x), static_cast<std::int64_t>(p.y), static_cast<std::int64_t>(p.z)};
			if (ref_size) {
				c[0] += o.x; c[1] += o.y; c[2] += o.z;
				for (int d = 0; d < 3; ++d) if (c[d] < 0 || c[d] >= ref_size[d]) return;
			}
			if (n < cap) for (int d = 0; d < 3; ++d) xyz[3 * n + d] = static_cast<std::int32_t>(c[d]);
			++n;
		});
		return n;
	}
}

// ---- formats (include/fluid/data_structures/point_cloud.h, mesh.h) ----------------------------------------------------
namespace {
	std::size_t emit(const std::string &s, char *buf, std::size_t cap) {
		if (buf && cap) std::memcpy(buf, s.data(), std::min(cap, s.size()));
		return s.size();
	}
}
extern "C" {
	/// point_cloud::save_to_naive on a default-formatted stream; returns the text length.
	std::size_t ref_points_text(const double *pos, std::size_t n, char *buf, std::size_t cap) {
		std::vector<vec3d> pts(n);
		for (std::size_t i = 0; i < n; ++i) pts[i] = vec3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
		std::ostringstream out;
		fluid::point_cloud::save_to_naive(out, pts.begin(), pts.end());
		return emit(out.str(), buf, cap);
	}
	/// point_cloud::load_from_naive; returns the number of points read (at most `count`), fills at most `cap` of them.
	std::size_t ref_points_parse(const char *text, std::size_t len, std::size_t count, double *pos, std::size_t cap) {
		std::istringstream in(std::string(text, len));
		std::vector<vec3d> pts = fluid::point_cloud::load_from_naive(in, count);
		for (std::size_t i = 0; i < pts.size() && i < cap; ++i) { pos[3 * i] = pts[i].x; pos[3 * i + 1] = pts[i].y; pos[3 * i + 2] = pts[i].z; }
		return pts.size();
	}
	/// mesh::save_obj after optional generate_normals / reverse_face_directions; uvs = u,v pairs or NULL.
	std::size_t ref_mesh_obj(
		const double *pos, std::size_t nv, const std::uint64_t *idx, std::size_t ni, const double *uvs, int normals, int reverse,
		char *buf, std::size_t cap, double *normals_out
	) {
		auto m = make_mesh(pos, nv, idx, ni);
		if (uvs) for (std::size_t i = 0; i < nv; ++i) m.uvs.emplace_back(uvs[2 * i], uvs[2 * i + 1]);
		if (reverse) m.reverse_face_directions();
		if (normals) m.generate_normals();
		if (normals && normals_out)
			for (std::size_t i = 0; i < nv; ++i) { normals_out[3 * i] = m.normals[i].x; normals_out[3 * i + 1] = m.normals[i].y; normals_out[3 * i + 2] = m.normals[i].z; }
		std::ostringstream out;
		m.save_obj(out);
		return emit(out.str(), buf, cap);
	}
}

// ---- surface mesher (src/mesher.cpp) -------------------------------------------------------------------------------------
namespace {
	void setup_mesher(fluid::mesher &m, const std::uint64_t *size, const double *off, double cs, double extent, std::uint64_t radius) {
		m.resize(vec3s(size[0], size[1], size[2]));
		m.grid_offset = vec3d(off[0], off[1], off[2]);
		m.cell_size = cs;
		m.particle_extent = extent;
		m.cell_radius = radius;
	}
	std::vector<vec3d> to_points(const double *pos, std::size_t n) {
		std::vector<vec3d> pts(n);
		for (std::size_t i = 0; i < n; ++i) pts[i] = vec3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
		return pts;
	}
}
extern "C" {
	/// mesher::_sample_surface_function (src/mesher.cpp:333-376): values at the (size+1)^3 grid points, x fastest.
	void ref_mesher_surface(
		const double *pos, std::size_t n, const std::uint64_t *size, const double *off, double cs, double extent,
		std::uint64_t radius, double r, double *values
	) {
		fluid::mesher m;
		setup_mesher(m, size, off, cs, extent, radius);
		std::vector<vec3d> pts = to_points(pos, n);
		m._sample_surface_function(pts, r);
		std::size_t nv = (size[0] + 1) * (size[1] + 1) * (size[2] + 1);
		for (std::size_t i = 0; i < nv; ++i) values[i] = m._surface_function[i];
	}
	/// mesher::generate_mesh (values == NULL) or mesher::_marching_cubes on given grid-point values (src/mesher.cpp:400-515).
	/// counts[0..1] = vertices, indices; fills at most cap_v / cap_i of them.
	void ref_mesher_mesh(
		const double *pos, std::size_t n, const std::uint64_t *size, const double *off, double cs, double extent,
		std::uint64_t radius, double r, const double *values, double *vpos, std::size_t cap_v, std::uint64_t *idx,
		std::size_t cap_i, std::uint64_t *counts
	) {
		fluid::mesher m;
		setup_mesher(m, size, off, cs, extent, radius);
		fluid::mesher::mesh_t res;
		if (values) {
			std::size_t nv = (size[0] + 1) * (size[1] + 1) * (size[2] + 1);
			for (std::size_t i = 0; i < nv; ++i) m._surface_function[i] = values[i];
			res = m._marching_cubes();
		} else {
			std::vector<vec3d> pts = to_points(pos, n);
			res = m.generate_mesh(pts, r);
		}
		counts[0] = res.positions.size();
		counts[1] = res.indices.size();
		for (std::size_t i = 0; i < res.positions.size() && i < cap_v; ++i) {
			vpos[3 * i] = res.positions[i].x; vpos[3 * i + 1] = res.positions[i].y; vpos[3 * i + 2] = res.positions[i].z;
		}
		for (std::size_t i = 0; i < res.indices.size() && i < cap_i; ++i) idx[i] = res.indices[i];
	}
}
