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
// What runs where: EVERY stage of time_step runs on the device (there is no host implementation of any stage and no CPU
// fallback). Without callbacks a step is one lfa_time_step call; with callbacks the same device stages are called one by one and
// the callbacks are invoked between them in the reference's order. The particles and the grid stay on the device:
// particles() / grid() download lazily, only when something (a callback, the host after the step) actually calls them, and an
// edit made through the mutable references they return is detected (64-bit content hash) and uploaded before the next device
// stage. The testbed's three callbacks (testbed/main.cpp:101-123: print dt, read the pressure vector, read the particles after
// the step) therefore cost one n-double download and one particle download per step, not a PCIe round trip per stage.
// The device fuses the collision handling into the advection and the position correction; a host that installs
// post_advection_callback or post_correction_callback gets the split stages instead (lfa_advect / lfa_correct, the callback,
// lfa_collide), i.e. the callback sits exactly where src/simulation.cpp:51-59,111-117 has it and sees the moved, not yet
// collided particles with old_position = the position of before the move (no host in the reference tree installs either).
// The position correction - particle positions only - runs on a second HIP stream beside the pressure solve - grid only -
// (`overlap_correction`, default on), in the staged step as well: callbacks between the P2G and the correction that read the
// pressure or the grid never notice; one that asks for particles() or edits the solid cells there has the device take the
// correction back (exactly) and run it at the reference's place.
// The class never throws on the step path. A device error is latched: last_status() < 0, last_error() has the text, the failed
// time_step() leaves the particles untouched and every later time_step()/update() returns immediately until clear_status().
#pragma once

#include <algorithm>
#include <chrono>
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
#include "types.h"

namespace fluid_amd {
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
			/// src/simulation.cpp:13-23 (truncation, no clamp: positions are kept inside the grid by the collision handling)
			vec3s compute_cell_index(vec3d off, double h) const { return vec3s((position - off) / h); }
			std::pair<vec3s, vec3d> compute_cell_index_and_position(vec3d off, double h) const {
				const vec3d f = (position - off) / h;
				const vec3s i(f);
				return {i, f - vec3d(i)};
			}
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
			_grid = detail::make_mac_grid(sz);
			_space_hash = grid3<_cell_particles>(sz);
			if (_dev) { lfa_destroy(_dev); _dev = nullptr; }
			const auto t0 = std::chrono::steady_clock::now();
			_status = lfa_create(&_dev, sz.x, sz.y, sz.z, device);
			_create_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
			if (_status != LFA_OK) _error = lfa_last_error(nullptr);
			_solids_dirty = true;
			_host_stale = _grid_stale = false;  // the new handle holds nothing yet
			_dev_stale = true;
			_particles_handed_out = _grid_handed_out = false;
			_rehash_grid();
		}

		/// simulation::update (src/simulation.cpp:31-41).
		void update(double dt) {
			while (_status >= 0) {
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
			if (_dev && _host_stale) {  // the device holds the current particles
				double out = 0.0;
				if (lfa_cfl(_dev, &out) == LFA_OK) return out;
			}
			const_cast<simulation*>(this)->_sync_host();
			double m = 0.0;
			for (const particle &p : _particles) m = std::max(m, p.velocity.squared_length());
			return cell_size / std::sqrt(m);
		}

		// State lives on the device; these accessors synchronise lazily. The non-const overloads hand out mutable references
		// ("do not store references", simulation.h:141): what the caller does with them is found out by comparing a content hash
		// before the next device stage (_flush_host_edits), so a read through a non-const simulation costs a download, not an upload.
		mac_grid &grid() { _sync_grid(); _grid_handed_out = true; return _grid; }
		const mac_grid &grid() const { const_cast<simulation*>(this)->_sync_grid(); return _grid; }
		std::vector<particle> &particles() { _sync_host(); _particles_handed_out = true; return _particles; }
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
		int apic_unscaled_kernel = 1;               ///< 1: the reference's APIC hat on world distances (simulation.cpp:367-369)
		int pcg_warm_start = 0;                     ///< 1: the PCG starts from the previous step's pressure (0: from p = 0 like the reference)
		bool overlap_correction = true;             ///< the position correction runs on a second stream beside the pressure solve
		int p2g_variant = LFA_P2G_LDS_BINNED, precond = LFA_PRECOND_MULTIGRID, pcg_dtype = LFA_PCG_F32;
		double pcg_tau = 0.97, pcg_sigma = 0.25, pcg_tolerance = 1e-6;   ///< pressure_solver.h:39-41
		std::size_t pcg_max_iterations = 200;                             ///< pressure_solver.h:42
		int last_status() const { return _status; }
		/// Wall milliseconds the last resize() spent in lfa_create (cheap after the first handle of a size: csrc/pool.hip).
		double device_create_ms() const { return _create_ms; }
		void clear_status() { _status = LFA_OK; _error.clear(); }
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
		double _create_ms = 0.0;  // wall time of the last lfa_create (device_create_ms())
		bool _solids_dirty = true;
		bool _host_stale = false;  // the device holds newer particles than _particles
		bool _dev_stale = true;    // _particles must be uploaded before the next device stage
		bool _grid_stale = false;  // the device holds a newer grid than _grid
		bool _particles_handed_out = false, _grid_handed_out = false;  // a mutable reference went out since the last hash
		std::uint64_t _particles_hash = 0, _grid_vel_hash = 0, _grid_solid_hash = 0;
		bool _in_step = false;     // between two device stages of a staged time_step (edits need a re-binning / a grid upload)
		// Staged step: the correction has been started on the device's second stream right after the P2G
		// (lfa_correct_collide_begin) although the reference runs it after the pressure gradient. Callbacks in between that only
		// look at the pressure / the grid velocities never notice; one that asks for particles() or changes the solid cells takes
		// the correction back (lfa_correct_collide_undo: exact) and the stage runs where the reference has it.
		bool _corr_in_flight = false;

		/// 64-bit content hash of words [first_word, first_word + n_words) of every record of `stride_words` 64-bit words:
		/// multiply-xorshift per word, chunks of 16 Ki records combined with their index. At most 8 OpenMP threads, and only for
		/// arrays worth it (a wide team on a many-core host costs far more in wake-ups and spinning than the hash itself).
		static std::uint64_t _hash_words(const void *data, std::size_t bytes, std::size_t stride_words, std::size_t first_word,
		                                 std::size_t n_words) {
			const std::uint64_t *w = static_cast<const std::uint64_t*>(data);
			const std::size_t records = bytes / (8 * stride_words);
			const std::size_t chunk = 1 << 14;
			const long n_chunks = static_cast<long>((records + chunk - 1) / chunk);
			std::uint64_t total = 0x9E3779B97F4A7C15ull ^ records;
	#pragma omp parallel for reduction(^ : total) schedule(static) num_threads(8) if (n_chunks >= 64)
			for (long c = 0; c < n_chunks; ++c) {
				std::uint64_t h = 0xD6E8FEB86659FD93ull + static_cast<std::uint64_t>(c);
				const std::size_t r1 = std::min(records, (static_cast<std::size_t>(c) + 1) * chunk);
				for (std::size_t r = static_cast<std::size_t>(c) * chunk; r < r1; ++r)
					for (std::size_t k = 0; k < n_words; ++k) {
						h ^= w[r * stride_words + first_word + k];
						h *= 0xFF51AFD7ED558CCDull;
						h ^= h >> 32;
					}
				total ^= h * (2 * static_cast<std::uint64_t>(c) + 1);
			}
			return total;
		}
		void _rehash_particles() { _particles_hash = _hash_words(_particles.data(), _particles.size() * sizeof(particle), 19, 0, 19); }
		void _rehash_grid() {
			const std::size_t bytes = detail::cell_count(_grid.grid()) * sizeof(mac_grid::cell);
			_grid_vel_hash = _hash_words(detail::cell_data(_grid.grid()), bytes, 4, 0, 3);
			_grid_solid_hash = _solid_hash();
		}
		std::uint64_t _solid_hash() const {
			// only WHICH cells are solid matters to the device (air / fluid are recomputed by every P2G)
			const mac_grid::cell *c = detail::cell_data(_grid.grid());
			const std::size_t n = detail::cell_count(_grid.grid());
			std::uint64_t h = 0x2545F4914F6CDD1Dull;
			for (std::size_t i = 0; i < n; ++i)
				if (c[i].cell_type == mac_grid::cell::type::solid) { h ^= i + 0x9E3779B97F4A7C15ull; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
			return h;
		}

		/// Takes back a correction that runs ahead of the reference's step order (see _corr_in_flight).
		bool _take_back_correction() {
			if (!_corr_in_flight) return true;
			_corr_in_flight = false;
			return _ok(lfa_correct_collide_undo(_dev));
		}
		void _sync_host() {
			if (_host_stale && _dev) {
				if (!_take_back_correction()) return;
				_particles.resize(static_cast<std::size_t>(lfa_num_particles(_dev)));  // sources create particles on the device
				_ok(lfa_download_particles(_dev, _particles.data(), _particles.size(), LFA_DL_POSITIONS));
				_host_stale = false;
				_rehash_particles();
				_particles_handed_out = false;
			}
		}
		void _sync_grid() {
			if (_grid_stale && _dev) {
				_ok(lfa_download_cells(_dev, detail::cell_data(_grid.grid())));
				_grid_stale = false;
				_rehash_grid();
				_grid_handed_out = false;
			}
		}
		bool _any_stage_callback() const {
			return post_advection_callback || post_particle_to_grid_transfer_callback ||
			       post_gravity_callback || post_pressure_solve_callback || post_apply_pressure_callback ||
			       post_correction_callback || post_grid_to_particle_transfer_callback;
		}

		bool _ok(int rc) {
			if (rc < 0) { _status = rc; _error = _dev ? lfa_last_error(_dev) : lfa_last_error(nullptr); return false; }
			return true;
		}
		bool _flush_host_edits();
		bool _push_sources();
		void _device_advanced() { _host_stale = true; _grid_stale = true; }
		bool _push_params();
		bool _push_solids();
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
		p.p2g_variant = p2g_variant; p.precond = precond; p.pcg_dtype = pcg_dtype; p.apic_unscaled_kernel = apic_unscaled_kernel; p.pcg_warm_start = pcg_warm_start;
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
			// `vec3d(dist(random), dist(random), dist(random))` (simulation.cpp:145): the language leaves the order of the three draws
			// open; g++ - what the reference is built with here (oracle/Makefile) - makes them right to left, so z gets the first
			// draw. With the same generator state this seeds the very particles the reference does (tests/test_ref_callers.py).
			// A host whose own reference build draws left to right (clang, MSVC: the Maya plugin on Windows) defines
			// LFA_SEED_DRAW_ORDER_LTR to seed what ITS reference seeds.
#ifdef LFA_SEED_DRAW_ORDER_LTR
			const double a = dist(random), b = dist(random), c = dist(random);
#else
			const double c = dist(random), b = dist(random), a = dist(random);
#endif
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
#ifdef LFA_SEED_DRAW_ORDER_LTR
								const double a = dist(random), b = dist(random), c = dist(random);
#else
								const double c = dist(random), b = dist(random), a = dist(random);  // z first: see seed_cell
#endif
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

	/// Sources go to the device as the flat int[3k] lists the Maya plugin holds (grid_node.cpp:295-303); re-sending an unchanged
	/// list is free on the device side.
	inline bool simulation::_push_sources() {
		if (!_ok(lfa_clear_sources(_dev))) return false;
		std::vector<std::int32_t> xyz;
		for (const auto &src : sources) {
			if (!src) continue;
			xyz.clear();
			for (const vec3s &c : src->cells) {
				xyz.push_back(static_cast<std::int32_t>(c.x)); xyz.push_back(static_cast<std::int32_t>(c.y));
				xyz.push_back(static_cast<std::int32_t>(c.z));
			}
			const double vel[3] = {src->velocity.x, src->velocity.y, src->velocity.z};
			if (!_ok(lfa_add_source(_dev, xyz.data(), xyz.size() / 3, vel, src->target_density_cubic_root, src->active ? 1 : 0,
			                        src->coerce_velocity ? 1 : 0)))
				return false;
		}
		return true;
	}

	/// Brings the device up to date with whatever the host did through particles() / grid() / the seeding functions since the last
	/// device stage. Between steps only the solid mask of the grid matters (the P2G rebuilds velocities and air / fluid types);
	/// inside a staged step an edited grid is uploaded whole and edited particles are re-uploaded and re-binned.
	inline bool simulation::_flush_host_edits() {
		if (_particles_handed_out && !_host_stale) {
			const std::uint64_t before = _particles_hash;
			_rehash_particles();
			if (_particles_hash != before) _dev_stale = true;
			_particles_handed_out = false;
		}
		if (_grid_handed_out && !_grid_stale) {
			const std::uint64_t vel_before = _grid_vel_hash, solid_before = _grid_solid_hash;
			_rehash_grid();
			if (_grid_solid_hash != solid_before) _solids_dirty = true;
			if (_solids_dirty && !_take_back_correction()) return false;  // the correction collides against the solid cells
			if (_in_step && (_grid_vel_hash != vel_before || _grid_solid_hash != solid_before)) {
				if (!_ok(lfa_upload_cells(_dev, detail::cell_data(_grid.grid())))) return false;
				_solids_dirty = false;
			}
			_grid_handed_out = false;
		}
		if (_solids_dirty) {
			if (!_push_solids()) return false;
			_solids_dirty = false;
			_grid_solid_hash = _solid_hash();
		}
		if (_dev_stale) {
			// (a correction running ahead worked on the particles this upload replaces: lfa_upload_particles joins it, and the
			// stage runs again, serially, where the reference has it)
			_corr_in_flight = false;
			if (!_ok(lfa_upload_particles(_dev, _particles.data(), _particles.size()))) return false;
			_dev_stale = false;
			_host_stale = false;
			_rehash_particles();
			if (_in_step && !_ok(lfa_hash_particles(_dev))) return false;  // the stages of the running step need the binning
		}
		return true;
	}

	inline void simulation::time_step(double dt) {
		if (_status < 0) return;  // a latched device failure: see clear_status()
		if (!_dev) { _status = LFA_E_NO_DEVICE; _error = "no device handle: resize() failed or was not called (there is no CPU fallback)"; return; }
		if (pre_time_step_callback) pre_time_step_callback(dt);
		_in_step = false;
		if (!_push_params() || !_push_sources() || !_flush_host_edits()) return;  // nothing has been advanced

		double residual = 0.0;
		std::uint64_t iters = 0;
		if (!_any_stage_callback()) {
			// ---- the whole step in one call (src/simulation.cpp:43-125 incl. sources)
			if (_ok(lfa_time_step(_dev, dt, &residual, &iters))) _device_advanced();
			return;
		}
		// ---- the same device stages one by one, callbacks in the reference's order between them. After every callback the
		// host's edits (if any) are flushed; a failure stops the step where it is (the status is latched).
		auto stage = [&](int rc) { if (!_ok(rc)) return false; _device_advanced(); return true; };
		auto after = [&](const std::function<void(double)> &cb) {
			if (!cb) return true;
			cb(dt);
			return _flush_host_edits();
		};
		_in_step = true;
		// (a host that installs post_advection_callback gets it where the reference has it, between the advection and its
		// collision handling, simulation.cpp:50-59: the two device stages are then called separately)
		bool ok = (post_advection_callback
		               ? stage(lfa_advect(_dev, dt)) && after(post_advection_callback) && stage(lfa_collide(_dev))
		               : stage(lfa_advect_collide(_dev, dt))) &&
		          stage(lfa_hash_particles(_dev)) && (sources.empty() || stage(lfa_update_sources(_dev, nullptr))) &&
		          stage(lfa_p2g(_dev)) && after(post_particle_to_grid_transfer_callback);
		// From here to the correction the reference's stages touch the grid only (simulation.cpp:82-99): the correction starts now,
		// on the second stream (see _corr_in_flight).
		// (not with a post_correction_callback: the correction then runs where the reference has it, split from its collisions)
		if (ok && overlap_correction && !post_correction_callback) {
			ok = _ok(lfa_correct_collide_begin(_dev, dt));
			_corr_in_flight = ok;
		}
		ok = ok && stage(lfa_add_gravity(_dev, dt)) && after(post_gravity_callback) &&
		     stage(lfa_pcg_solve(_dev, dt, &residual, &iters));
		if (ok && post_pressure_solve_callback) {
			// the callback gets the pressure vector by reference and may edit it (simulation.h:166): n doubles down, and up again
			// only if it did
			std::vector<double> pressure(static_cast<std::size_t>(lfa_num_fluid_cells(_dev)));
			ok = _ok(lfa_download_pressure(_dev, pressure.data(), pressure.size()));
			if (ok) {
				const std::vector<double> before = pressure;
				post_pressure_solve_callback(dt, pressure, residual, static_cast<std::size_t>(iters));
				if (pressure.size() == before.size() && pressure != before)
					ok = _ok(lfa_upload_pressure(_dev, pressure.data(), pressure.size()));
				ok = ok && _flush_host_edits();
			}
		}
		ok = ok && stage(lfa_apply_pressure(_dev, dt)) && after(post_apply_pressure_callback);
		if (ok && _corr_in_flight) {  // nobody has asked for it to be taken back: it is the reference's correction, finished early
			_corr_in_flight = false;
			ok = stage(lfa_correct_collide_end(_dev));
		} else if (post_correction_callback) {  // _correct_positions -> callback -> _detect_collisions (simulation.cpp:111-117)
			ok = ok && stage(lfa_correct(_dev, dt)) && after(post_correction_callback) && stage(lfa_collide(_dev));
		} else {
			ok = ok && stage(lfa_correct_collide(_dev, dt));
		}
		_corr_in_flight = false;
		ok = ok && stage(lfa_extrapolate(_dev)) && stage(lfa_g2p(_dev));
		_in_step = false;
		if (ok && post_grid_to_particle_transfer_callback) {
			post_grid_to_particle_transfer_callback(dt);
			_flush_host_edits();
		}
	}
}  // namespace fluid_amd
