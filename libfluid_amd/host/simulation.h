// libfluid_amd/host/simulation.h -- C++17 host class with the public surface of lukedan/libfluid's `fluid::simulation`
// (include/fluid/simulation.h:21-280), running the per-step hot path on an MI355X through the C ABI in
// include/libfluid_amd.h. Header-only; link with libfluid_amd.so.
//
// Drop-in contract: same member names, argument meaning, callback order and defaults as the reference:
//   resize :57  update :60  time_step :62-64  reset_space_hash :67  update_and_hash_particles :69  hash_particles :71
//   seed_cell/seed_func/seed_box/seed_sphere :76-123  world_position_to_cell_index[_unclamped] :126-128  cfl :131
//   grid :134-140  particles :142-148  the eight std::function callbacks :153-175  public fields :177-190
// `namespace fluid_amd` mirrors `namespace fluid`; a host that wants the device path replaces
// `fluid::simulation` by `fluid_amd::simulation`.
//
// What runs where: without fluid sources and callbacks a whole time_step runs on the device (lfa_time_step) and the
// particles stay there; particles()/grid() download lazily. With a source or any callback set, the stages of SURVEY 8(a)
// run on the device one by one and the stages outside it (advection, collision, position correction, sources:
// src/simulation.cpp:226-249,562-683,756-765) run here on the host exactly where the reference runs them, so that every
// callback sees host-visible state at its point of the step (particles then cross PCIe twice up and once down per step).
// The class never throws on the step path; device errors are kept in last_status()/last_error().
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <random>
#include <string>
#include <utility>
#include <vector>

#include "../../include/libfluid_amd.h"

namespace fluid_amd {
	template <typename T> struct vec3 {
		T x{}, y{}, z{};
		vec3() = default;
		vec3(T a, T b, T c) : x(a), y(b), z(c) {}
		template <typename U> explicit vec3(const vec3<U> &o) : x(static_cast<T>(o.x)), y(static_cast<T>(o.y)), z(static_cast<T>(o.z)) {}
		T &operator[](std::size_t i) { return (&x)[i]; }
		T operator[](std::size_t i) const { return (&x)[i]; }
		vec3 &operator+=(const vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
		vec3 &operator-=(const vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
		friend vec3 operator+(vec3 a, const vec3 &b) { return a += b; }
		friend vec3 operator-(vec3 a, const vec3 &b) { return a -= b; }
		friend vec3 operator*(vec3 a, T s) { a.x *= s; a.y *= s; a.z *= s; return a; }
		friend vec3 operator*(T s, vec3 a) { return a * s; }
		friend vec3 operator/(vec3 a, T s) { a.x /= s; a.y /= s; a.z /= s; return a; }
		T squared_length() const { T r{}; r += x * x; r += y * y; r += z * z; return r; }
	};
	using vec3d = vec3<double>;
	using vec3s = vec3<std::size_t>;
	using vec3i = vec3<int>;
	inline double dot(const vec3d &a, const vec3d &b) { double r = 0.0; r += a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }

	/// Dense x-fastest 3-D array with the indexing surface of fluid::grid3 (include/fluid/data_structures/grid.h:13-246).
	template <typename Cell> class grid3 {
	public:
		grid3() = default;
		explicit grid3(vec3s size, const Cell &c = Cell{}) : _cells(size.x * size.y * size.z, c), _size(size) {}
		Cell &operator()(std::size_t x, std::size_t y, std::size_t z) { return _cells[index_to_raw(vec3s(x, y, z))]; }
		const Cell &operator()(std::size_t x, std::size_t y, std::size_t z) const { return _cells[index_to_raw(vec3s(x, y, z))]; }
		Cell &operator()(vec3s i) { return _cells[index_to_raw(i)]; }
		const Cell &operator()(vec3s i) const { return _cells[index_to_raw(i)]; }
		Cell &operator[](std::size_t raw) { return _cells[raw]; }
		const Cell &operator[](std::size_t raw) const { return _cells[raw]; }
		vec3s get_size() const { return _size; }
		std::size_t get_array_size() const { return _cells.size(); }
		void fill(const Cell &c) { std::fill(_cells.begin(), _cells.end(), c); }
		std::size_t index_to_raw(vec3s i) const { return i.x + _size.x * (i.y + _size.y * i.z); }
		vec3s index_from_raw(std::size_t r) const {
			vec3s v;
			v.x = r % _size.x; r /= _size.x;
			v.y = r % _size.y; r /= _size.y;
			v.z = r;
			return v;
		}
		template <typename Cb> void for_each(Cb &&cb) {
			for (std::size_t z = 0; z < _size.z; ++z)
				for (std::size_t y = 0; y < _size.y; ++y)
					for (std::size_t x = 0; x < _size.x; ++x) cb(vec3s(x, y, z), (*this)(x, y, z));
		}
		Cell *data() { return _cells.data(); }
		const Cell *data() const { return _cells.data(); }
	private:
		std::vector<Cell> _cells;
		vec3s _size;
	};

	/// fluid::mac_grid (include/fluid/mac_grid.h:12-73): 32-byte cells, out-of-range == solid.
	class mac_grid {
	public:
		struct cell {
			enum class type : unsigned char { air = 0x1, fluid = 0x2, solid = 0x4 };
			vec3d velocities_posface;
			type cell_type = type::air;
		};
		mac_grid() = default;
		explicit mac_grid(vec3s n) : _grid(n) {}
		cell *get_cell(vec3s i) {
			vec3s n = _grid.get_size();
			return (i.x >= n.x || i.y >= n.y || i.z >= n.z) ? nullptr : &_grid(i);
		}
		std::pair<cell*, cell::type> get_cell_and_type(vec3s i) {
			if (cell *c = get_cell(i)) return {c, c->cell_type};
			return {nullptr, cell::type::solid};
		}
		grid3<cell> &grid() { return _grid; }
		const grid3<cell> &grid() const { return _grid; }
	private:
		grid3<cell> _grid;
	};
	static_assert(sizeof(mac_grid::cell) == 32, "cell layout must match the reference (32-B AoS)");

	/// fluid::source (include/fluid/data_structures/source.h:12-22).
	class source {
	public:
		std::vector<vec3s> cells;
		vec3d velocity;
		std::size_t target_density_cubic_root = 2;
		bool active = true, coerce_velocity = false;
	};

	/// pcg32 (XSH-RR 64/32, the generator family of the reference's `pcg32 random` member, simulation.h:177), own code.
	class pcg32 {
	public:
		using result_type = std::uint32_t;
		explicit pcg32(std::uint64_t seed = 0xcafef00dd15ea5e5ull) { _state = 0; (*this)(); _state += seed; (*this)(); }
		static constexpr result_type min() { return 0; }
		static constexpr result_type max() { return 0xffffffffu; }
		result_type operator()() {
			std::uint64_t old = _state;
			_state = old * 6364136223846793005ull + 1442695040888963407ull;
			std::uint32_t xs = static_cast<std::uint32_t>(((old >> 18u) ^ old) >> 27u), rot = static_cast<std::uint32_t>(old >> 59u);
			return (xs >> rot) | (xs << ((32u - rot) & 31u));
		}
	private:
		std::uint64_t _state;
	};

	class simulation {
	public:
		struct particle {
			vec3d position, velocity, cx, cy, cz, old_position;
			std::size_t raw_cell_index = 0;
			vec3s compute_cell_index(vec3d off, double h) const { return vec3s((position - off) / h); }
		};
		static_assert(sizeof(std::size_t) == 8, "64-bit host expected");
		enum class method : unsigned char { pic, flip_blend, apic };
		constexpr static bool precise_collision_detection = true;
		constexpr static std::size_t default_seeding_density = 2;

		simulation() = default;
		simulation(const simulation&) = delete;
		simulation &operator=(const simulation&) = delete;
		~simulation() { if (_dev) lfa_destroy(_dev); }

		void resize(vec3s sz) {
			_grid = mac_grid(sz);
			_space_hash = grid3<_cell_particles>(sz);
			if (_dev) { lfa_destroy(_dev); _dev = nullptr; }
			_status = lfa_create(&_dev, sz.x, sz.y, sz.z, device);
			if (_status != LFA_OK) _error = lfa_last_error(nullptr);
			_solids_dirty = true;
		}

		/// simulation::update (src/simulation.cpp:31-41).
		void update(double dt) {
			while (true) {
				double ts = cfl_number * cfl();
				if (ts > dt) { time_step(dt); break; }
				time_step(ts);
				dt -= ts;
			}
		}
		/// simulation::time_step() (src/simulation.cpp:127-129).
		void time_step() { time_step(std::min(cfl_number * cfl(), 0.033)); }
		/// simulation::time_step(dt) (src/simulation.cpp:43-125): same stage and callback order.
		void time_step(double dt);

		void reset_space_hash() { _space_hash.fill(_cell_particles()); _fluid_cells.clear(); }
		void update_and_hash_particles();
		void hash_particles();

		void seed_cell(vec3s cell, vec3d velocity, std::size_t density = default_seeding_density);
		template <typename Func> void seed_func(vec3s start, vec3s size, const Func &pred, vec3d velocity = vec3d(),
		                                        std::size_t density = default_seeding_density);
		void seed_box(vec3d start, vec3d size, vec3d velocity = vec3d(), std::size_t density = default_seeding_density);
		void seed_sphere(vec3d center, double radius, vec3d velocity = vec3d(), std::size_t density = default_seeding_density);

		vec3s world_position_to_cell_index(vec3d pos) const {
			vec3s u = world_position_to_cell_index_unclamped(pos), n = _grid.grid().get_size();
			return vec3s(std::min(u.x, n.x), std::min(u.y, n.y), std::min(u.z, n.z));
		}
		vec3s world_position_to_cell_index_unclamped(vec3d pos) const {
			vec3d g = (pos - grid_offset) / cell_size;
			return vec3s(static_cast<std::size_t>(std::max(g.x, 0.0)), static_cast<std::size_t>(std::max(g.y, 0.0)),
			             static_cast<std::size_t>(std::max(g.z, 0.0)));
		}
		/// simulation::cfl (src/simulation.cpp:199-205); on the device when the particles are resident there.
		double cfl() const {
			if (_dev && !_dev_stale && _host_stale) {
				double out = 0.0;
				if (lfa_cfl(_dev, &out) == LFA_OK) return out;
			}
			const_cast<simulation*>(this)->_sync_host();
			double m = 0.0;
			for (const particle &p : _particles) m = std::max(m, p.velocity.squared_length());
			return cell_size / std::sqrt(m);
		}

		// State lives on the device between steps when nothing forces it to the host (no sources, no callbacks): these
		// accessors synchronise lazily. The non-const particles() hands out a mutable reference ("do not store references",
		// simulation.h:141), so the device copy is considered stale afterwards and is re-uploaded by the next step.
		mac_grid &grid() { _sync_grid(); return _grid; }
		const mac_grid &grid() const { const_cast<simulation*>(this)->_sync_grid(); return _grid; }
		std::vector<particle> &particles() { _sync_host(); _dev_stale = true; return _particles; }
		const std::vector<particle> &particles() const { const_cast<simulation*>(this)->_sync_host(); return _particles; }

		// callbacks, in calling order (include/fluid/simulation.h:150-175)
		std::function<void(double)> pre_time_step_callback, post_advection_callback,
			post_particle_to_grid_transfer_callback, post_gravity_callback;
		std::function<void(double, std::vector<double>&, double, std::size_t)> post_pressure_solve_callback;
		std::function<void(double)> post_apply_pressure_callback, post_correction_callback,
			post_grid_to_particle_transfer_callback;

		pcg32 random;
		std::vector<std::unique_ptr<source>> sources;
		vec3d grid_offset, gravity;
		double cfl_number = 3.0, blending_factor = 1.0, cell_size = std::numeric_limits<double>::quiet_NaN(), density = 1.0,
		       boundary_skin_width = 0.1, correction_stiffness = 5.0;
		std::size_t velocity_extrapolation_iterations = 1;
		method simulation_method = method::apic;

		// -- device-path selectors (not in the reference) and status
		int device = -1;                            ///< HIP device (-1: current); takes effect at resize()
		bool device_resident_steps = true;          ///< run whole steps on the device when no source/callback needs the host
		int p2g_variant = LFA_P2G_LDS_BINNED, precond = LFA_PRECOND_MULTIGRID, pcg_dtype = LFA_PCG_F32;
		double pcg_tau = 0.97, pcg_sigma = 0.25, pcg_tolerance = 1e-6;   ///< pressure_solver.h:39-41
		std::size_t pcg_max_iterations = 200;                             ///< pressure_solver.h:42
		int last_status() const { return _status; }
		const std::string &last_error() const { return _error; }
		lfa_sim *device_handle() { return _dev; }

	private:
		struct _cell_particles { std::size_t begin = 0, count = 0; };
		std::vector<particle> _particles;
		mac_grid _grid;
		grid3<_cell_particles> _space_hash;
		std::vector<std::size_t> _fluid_cells;
		lfa_sim *_dev = nullptr;
		int _status = LFA_OK;
		std::string _error;
		bool _solids_dirty = true;
		bool _host_stale = false;  // the device holds newer particles than _particles
		bool _dev_stale = true;    // _particles may have been edited since the last upload
		bool _grid_stale = false;  // the device holds a newer grid than _grid

		void _sync_host() {
			if (_host_stale && _dev) {
				_ok(lfa_download_particles(_dev, _particles.data(), _particles.size(), LFA_DL_POSITIONS));
				_host_stale = false;
			}
		}
		void _sync_grid() {
			if (_grid_stale && _dev) {
				_ok(lfa_download_cells(_dev, _grid.grid().data()));
				_grid_stale = false;
			}
		}
		bool _any_callback() const {
			return pre_time_step_callback || post_advection_callback || post_particle_to_grid_transfer_callback ||
			       post_gravity_callback || post_pressure_solve_callback || post_apply_pressure_callback ||
			       post_correction_callback || post_grid_to_particle_transfer_callback;
		}

		bool _ok(int rc) {
			if (rc < 0) { _status = rc; _error = _dev ? lfa_last_error(_dev) : lfa_last_error(nullptr); return false; }
			return true;
		}
		bool _push_params();
		bool _push_solids();
		void _pull_grid() { if (_dev) _ok(lfa_download_cells(_dev, _grid.grid().data())); }
		template <typename Cb> void _for_all_nearby_particles(vec3s c, Cb &&cb);
		void _advect_particles(double dt);
		void _correct_positions(double dt);
		void _detect_collisions();
		void _update_sources();
	};
	static_assert(sizeof(simulation::particle) == 152, "particle layout must match the reference (152-B AoS)");

	// ============================================================================================ implementation
	inline bool simulation::_push_params() {
		lfa_params p;
		lfa_default_params(&p);
		for (int k = 0; k < 3; ++k) { p.grid_offset[k] = grid_offset[k]; p.gravity[k] = gravity[k]; }
		p.cell_size = cell_size; p.blending_factor = blending_factor; p.density = density;
		p.boundary_skin_width = boundary_skin_width; p.correction_stiffness = correction_stiffness; p.cfl_number = cfl_number;
		p.velocity_extrapolation_iterations = velocity_extrapolation_iterations;
		p.simulation_method = static_cast<int>(simulation_method);
		p.tau = pcg_tau; p.sigma = pcg_sigma; p.tolerance = pcg_tolerance; p.max_iterations = pcg_max_iterations;
		p.p2g_variant = p2g_variant; p.precond = precond; p.pcg_dtype = pcg_dtype;
		return _ok(lfa_set_params(_dev, &p));
	}
	/// Solid cells are set by the hosts directly on grid() (testbed/main.cpp:167-176, grid_node.cpp:330-339); they are
	/// pushed to the device as the flat int[3k] list the Maya plugin uses.
	inline bool simulation::_push_solids() {
		std::vector<std::int32_t> xyz;
		_grid.grid().for_each([&](vec3s i, mac_grid::cell &c) {
			if (c.cell_type == mac_grid::cell::type::solid) {
				xyz.push_back(static_cast<std::int32_t>(i.x)); xyz.push_back(static_cast<std::int32_t>(i.y));
				xyz.push_back(static_cast<std::int32_t>(i.z));
			}
		});
		if (!_ok(lfa_clear_solid_cells(_dev))) return false;
		return _ok(lfa_set_solid_cells(_dev, xyz.data(), xyz.size() / 3));
	}

	inline void simulation::update_and_hash_particles() {
		_sync_host();
		_dev_stale = true;
		vec3s n = _grid.grid().get_size();
		for (particle &p : _particles) {
			vec3d g = (p.position - grid_offset) / cell_size;
			vec3s i(std::min(static_cast<std::size_t>(std::max(g.x, 0.0)), n.x - 1),
			        std::min(static_cast<std::size_t>(std::max(g.y, 0.0)), n.y - 1),
			        std::min(static_cast<std::size_t>(std::max(g.z, 0.0)), n.z - 1));
			p.raw_cell_index = _grid.grid().index_to_raw(i);
		}
		hash_particles();
	}
	inline void simulation::hash_particles() {
		_sync_host();
		_dev_stale = true;
		reset_space_hash();
		std::sort(_particles.begin(), _particles.end(),
		          [](const particle &a, const particle &b) { return a.raw_cell_index < b.raw_cell_index; });
		for (std::size_t i = 0; i < _particles.size();) {
			std::size_t c = _particles[i].raw_cell_index, j = i;
			while (j < _particles.size() && _particles[j].raw_cell_index == c) ++j;
			_space_hash[c].begin = i;
			_space_hash[c].count = j - i;
			_fluid_cells.push_back(c);
			i = j;
		}
	}

	inline void simulation::seed_cell(vec3s cell, vec3d velocity, std::size_t dens) {
		_sync_host();
		_dev_stale = true;
		std::size_t index = _grid.grid().index_to_raw(cell), num = _space_hash(cell).count, target = dens * dens * dens;
		std::uniform_real_distribution<double> dist(0.0, cell_size);
		vec3d base = grid_offset + vec3d(cell) * cell_size;
		for (; num < target; ++num) {
			particle p;
			double a = dist(random), b = dist(random), c = dist(random);
			p.old_position = p.position = base + vec3d(a, b, c);
			p.velocity = velocity;
			p.raw_cell_index = index;
			_particles.emplace_back(p);
		}
		_space_hash(cell).count = target;
	}
	template <typename Func>
	void simulation::seed_func(vec3s start, vec3s size, const Func &pred, vec3d velocity, std::size_t dens) {
		_sync_host();
		_dev_stale = true;
		double sub = cell_size / static_cast<double>(dens);
		std::uniform_real_distribution<double> dist(0.0, sub);
		vec3s n = _grid.grid().get_size();
		vec3s end(std::min(start.x + size.x, n.x), std::min(start.y + size.y, n.y), std::min(start.z + size.z, n.z));
		for (std::size_t z = start.z; z < end.z; ++z)
			for (std::size_t y = start.y; y < end.y; ++y)
				for (std::size_t x = start.x; x < end.x; ++x) {
					vec3d cell_off = vec3d(vec3s(x, y, z)) * cell_size;
					std::size_t raw = _grid.grid().index_to_raw(vec3s(x, y, z));
					for (std::size_t sx = 0; sx < dens; ++sx)
						for (std::size_t sy = 0; sy < dens; ++sy)
							for (std::size_t sz = 0; sz < dens; ++sz) {
								double a = dist(random), b = dist(random), c = dist(random);
								vec3d pos = grid_offset + cell_off + vec3d(vec3s(sx, sy, sz)) * sub + vec3d(a, b, c);
								if (pred(pos)) {
									particle p;
									p.old_position = p.position = pos;
									p.velocity = velocity;
									p.raw_cell_index = raw;
									_particles.emplace_back(p);
								}
							}
				}
	}
	inline void simulation::seed_box(vec3d start, vec3d size, vec3d vel, std::size_t dens) {
		vec3d end = start + size;
		vec3s s = world_position_to_cell_index_unclamped(start), e = world_position_to_cell_index_unclamped(end);
		seed_func(s, vec3s(e.x - s.x + 1, e.y - s.y + 1, e.z - s.z + 1), [&](vec3d p) {
			return p.x > start.x && p.y > start.y && p.z > start.z && p.x < end.x && p.y < end.y && p.z < end.z;
		}, vel, dens);
	}
	inline void simulation::seed_sphere(vec3d center, double radius, vec3d vel, std::size_t dens) {
		vec3d r(radius, radius, radius);
		vec3s s = world_position_to_cell_index_unclamped(center - r), e = world_position_to_cell_index_unclamped(center + r);
		double r2 = radius * radius;
		seed_func(s, vec3s(e.x - s.x + 1, e.y - s.y + 1, e.z - s.z + 1),
		          [&](vec3d p) { return (p - center).squared_length() < r2; }, vel, dens);
	}

	template <typename Cb> void simulation::_for_all_nearby_particles(vec3s c, Cb &&cb) {
		vec3s n = _space_hash.get_size();
		std::size_t x0 = c.x < 1 ? 0 : c.x - 1, y0 = c.y < 1 ? 0 : c.y - 1, z0 = c.z < 1 ? 0 : c.z - 1;
		std::size_t x1 = std::min(c.x + 2, n.x), y1 = std::min(c.y + 2, n.y), z1 = std::min(c.z + 2, n.z);
		for (std::size_t z = z0; z < z1; ++z)
			for (std::size_t y = y0; y < y1; ++y)
				for (std::size_t x = x0; x < x1; ++x) {
					const _cell_particles &cp = _space_hash(x, y, z);
					for (std::size_t k = 0; k < cp.count; ++k) cb(_particles[cp.begin + k]);
				}
	}

	/// simulation::_advect_particles (src/simulation.cpp:226-249).
	inline void simulation::_advect_particles(double dt) {
		for (auto &src : sources) {
			if (!src->active || !src->coerce_velocity) continue;
			for (vec3s v : src->cells) {
				_cell_particles cp = _space_hash(v);
				for (std::size_t k = 0; k < cp.count; ++k) {
					particle &p = _particles[cp.begin + k];
					p.velocity = src->velocity;
					p.cx = p.cy = p.cz = vec3d();
				}
			}
		}
		vec3d skin(boundary_skin_width, boundary_skin_width, boundary_skin_width);
		vec3d lo = grid_offset + skin, hi = cell_size * vec3d(_grid.grid().get_size()) + grid_offset - skin;
		for (particle &p : _particles) {
			p.position += p.velocity * dt;
			for (int a = 0; a < 3; ++a) p.position[a] = std::clamp(p.position[a], lo[a], hi[a]);
		}
	}

	/// simulation::_correct_positions (src/simulation.cpp:562-610): pairwise springs over the 27-cell neighbourhood.
	inline void simulation::_correct_positions(double dt) {
		const double re = cell_size / std::sqrt(2.0);
		std::vector<vec3d> moved(_particles.size());
		const int count = static_cast<int>(_particles.size());
#pragma omp parallel
		{
			pcg32 jitter_rng(std::random_device{}());
			std::uniform_real_distribution<double> dist(-1.0, 1.0);
#pragma omp for
			for (int i = 0; i < count; ++i) {
				const particle &p = _particles[static_cast<std::size_t>(i)];
				vec3d spring;
				_for_all_nearby_particles(p.compute_cell_index(grid_offset, cell_size), [&](const particle &o) {
					if (&o == &p) return;
					vec3d d = p.position - o.position;
					double d2 = d.squared_length();
					if (d2 < 1e-12) {
						double a = dist(jitter_rng), b = dist(jitter_rng), c = dist(jitter_rng);
						spring += vec3d(a, b, c);
					} else {
						double k = 1.0 - d2 / (re * re), w = k > 0.0 ? k * k * k : 0.0;
						spring += (w / std::sqrt(d2)) * d;
					}
				});
				moved[static_cast<std::size_t>(i)] = p.position + spring * (dt * correction_stiffness * re);
			}
		}
		vec3d hi = grid_offset + vec3d(_grid.grid().get_size()) * cell_size;
		for (std::size_t i = 0; i < _particles.size(); ++i)
			for (int a = 0; a < 3; ++a) _particles[i].position[a] = std::clamp(moved[i][a], grid_offset[a], hi[a]);
	}

	/// simulation::_detect_collisions (src/simulation.cpp:612-683) with grid::march_cells (grid.h:140-209): up to three
	/// bounces of the segment old_position -> position against solid cells / the domain walls, then skin push-out.
	inline void simulation::_detect_collisions() {
		const vec3s n = _grid.grid().get_size();
		auto solid_at = [&](int x, int y, int z) {
			if (x < 0 || y < 0 || z < 0) return true;
			if (static_cast<std::size_t>(x) >= n.x || static_cast<std::size_t>(y) >= n.y || static_cast<std::size_t>(z) >= n.z) return true;
			return _grid.grid()(x, y, z).cell_type == mac_grid::cell::type::solid;
		};
		const int count = static_cast<int>(_particles.size());
#pragma omp parallel for
		for (int pi = 0; pi < count; ++pi) {
			particle &p = _particles[static_cast<std::size_t>(pi)];
			vec3d from = p.old_position, to = p.position;
			for (int bounce = 0; bounce < 3; ++bounce) {
				bool hit = false;
				vec3d a = (from - grid_offset) / cell_size, b = (to - grid_offset) / cell_size, diff = b - a, inv, t;
				int cur[3], last[3], adv[3];
				for (int d = 0; d < 3; ++d) {
					cur[d] = static_cast<int>(std::floor(a[d]));
					last[d] = static_cast<int>(std::floor(b[d]));
					adv[d] = diff[d] > 0.0 ? 1 : -1;
					inv[d] = 1.0 / std::abs(diff[d]);
					t[d] = std::abs(static_cast<double>(cur[d] + (diff[d] > 0.0 ? 1 : 0)) - a[d]) * inv[d];
				}
				while (cur[0] != last[0] || cur[1] != last[1] || cur[2] != last[2]) {
					int dim = 0;
					double tmin = 2.0;
					for (int d = 0; d < 3; ++d) if (t[d] < tmin) { tmin = t[d]; dim = d; }
					if (!(tmin <= 1.0)) break;
					cur[dim] += adv[dim];
					if (solid_at(cur[0], cur[1], cur[2])) {
						vec3d normal;
						normal[dim] = -static_cast<double>(adv[dim]);
						vec3d off = to - from;
						double tt = std::max(t[dim] + boundary_skin_width / dot(off, normal), 0.0);
						from = tt * to + (1.0 - tt) * from;
						to[dim] = from[dim];
						hit = true;
						break;
					}
					t[dim] += inv[dim];
				}
				if (!hit) break;
			}
			p.position = to;
			vec3d gp = p.position - grid_offset;
			vec3s ci(gp / cell_size);
			vec3d cp = gp - vec3d(ci) * cell_size;
			const double skin_max = cell_size - boundary_skin_width;
			for (int d = 0; d < 3; ++d) {
				int c[3] = {static_cast<int>(ci.x), static_cast<int>(ci.y), static_cast<int>(ci.z)};
				if (cp[d] < boundary_skin_width) {
					int q[3] = {c[0], c[1], c[2]};
					q[d] -= 1;
					if (ci[d] == 0 || solid_at(q[0], q[1], q[2])) p.position[d] += boundary_skin_width - cp[d];
				}
				if (cp[d] > skin_max) {
					int q[3] = {c[0], c[1], c[2]};
					q[d] += 1;
					if (ci[d] + 1 >= n[d] || solid_at(q[0], q[1], q[2])) p.position[d] += skin_max - cp[d];
				}
			}
		}
	}

	inline void simulation::_update_sources() {
		for (auto &src : sources) {
			if (!src->active) continue;
			for (vec3s v : src->cells) seed_cell(v, src->velocity, src->target_density_cubic_root);
		}
	}

	inline void simulation::time_step(double dt) {
		// ---- device-resident step: nothing needs the host in the middle of the step (no sources, no callbacks)
		if (_dev && sources.empty() && !_any_callback() && device_resident_steps) {
			bool dev = _push_params();
			if (dev && _solids_dirty) { dev = _push_solids(); _solids_dirty = !dev; }
			if (dev && _dev_stale) {
				dev = _ok(lfa_upload_particles(_dev, _particles.data(), _particles.size()));
				_dev_stale = !dev;
			}
			double residual = 0.0;
			std::uint64_t iters = 0;
			if (dev && _ok(lfa_time_step(_dev, dt, &residual, &iters))) {
				_host_stale = true;
				_grid_stale = true;
				return;
			}
			// a device failure is reported through last_status(); fall through to the staged path
		}
		_sync_host();
		_dev_stale = true;
		_grid_stale = false;
		if (pre_time_step_callback) pre_time_step_callback(dt);
		update_and_hash_particles();
		_advect_particles(dt);
		if (post_advection_callback) post_advection_callback(dt);
		_detect_collisions();
		for (particle &p : _particles) p.old_position = p.position;
		update_and_hash_particles();
		_update_sources();
		hash_particles();

		// ---- hot path on the device -------------------------------------------------------------
		bool dev = _dev != nullptr && _push_params();
		if (dev && _solids_dirty) { dev = _push_solids(); _solids_dirty = !dev; }
		dev = dev && _ok(lfa_upload_particles(_dev, _particles.data(), _particles.size()));
		dev = dev && _ok(lfa_hash_particles(_dev));
		dev = dev && _ok(lfa_p2g(_dev));
		if (dev && post_particle_to_grid_transfer_callback) { _pull_grid(); post_particle_to_grid_transfer_callback(dt); }
		dev = dev && _ok(lfa_add_gravity(_dev, dt));
		if (dev && post_gravity_callback) { _pull_grid(); post_gravity_callback(dt); }
		double residual = 0.0;
		std::uint64_t iters = 0;
		dev = dev && _ok(lfa_pcg_solve(_dev, dt, &residual, &iters));
		if (dev && post_pressure_solve_callback) {
			std::vector<double> pressure(lfa_num_fluid_cells(_dev));
			_ok(lfa_download_pressure(_dev, pressure.data(), pressure.size()));
			std::vector<double> before = pressure;
			post_pressure_solve_callback(dt, pressure, residual, static_cast<std::size_t>(iters));
			if (pressure != before && pressure.size() == before.size())  // the callback may edit it (simulation.h:166)
				_ok(lfa_upload_pressure(_dev, pressure.data(), pressure.size()));
		}
		dev = dev && _ok(lfa_apply_pressure(_dev, dt));
		if (dev && post_apply_pressure_callback) { _pull_grid(); post_apply_pressure_callback(dt); }

		_correct_positions(dt);
		if (post_correction_callback) post_correction_callback(dt);
		_detect_collisions();
		for (particle &p : _particles) p.old_position = p.position;

		dev = dev && _ok(lfa_extrapolate(_dev));  // uses the fluid-cell set of the P2G-time hash, like the reference
		// G2P samples at the corrected positions (src/simulation.cpp:110-121): positions go up again, velocities come back
		dev = dev && _ok(lfa_upload_particles(_dev, _particles.data(), _particles.size()));
		dev = dev && _ok(lfa_hash_particles(_dev));
		dev = dev && _ok(lfa_g2p(_dev));
		dev = dev && _ok(lfa_download_particles(_dev, _particles.data(), _particles.size(), LFA_DL_KEEP_RAW));
		if (dev) _pull_grid();
		if (post_grid_to_particle_transfer_callback) post_grid_to_particle_transfer_callback(dt);
	}
}  // namespace fluid_amd
