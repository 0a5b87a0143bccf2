// libfluid_amd/host/mesher.h -- C++17 host class with the public surface of lukedan/libfluid's `fluid::mesher`
// (include/fluid/mesher.h:14-46): resize(vec3s), generate_mesh(particles, r), and the public fields grid_offset, cell_size,
// particle_extent, cell_radius. Sampling of the surface function and marching cubes run on an MI355X through the C ABI
// (include/libfluid_amd.h, lfa_mesher_*); the mesh comes back as fluid_amd::mesh (host/mesh.h) with positions and
// indices only, like the reference's. Header-only; link with libfluid_amd.so. Bit-exact with the reference
// (tests/test_mesher.py, tests/test_host_mesher.py). Never throws; device errors are kept in last_status()/last_error().
#pragma once

#include <string>
#include <vector>

#include "mesh.h"
#include "simulation.h"

namespace fluid_amd {
	class mesher {
	public:
		using mesh_t = mesh<double, std::size_t, double, double, vec3d>;

		mesher() = default;
		mesher(const mesher &) = delete;
		mesher &operator=(const mesher &) = delete;
		~mesher() { _release(); }

		/// Size of the sampling grid in cells; the surface function has one more point per axis (src/mesher.cpp:320-323).
		void resize(vec3s size) {
			_size = size;
			_release();
		}

		/// src/mesher.cpp:325-328. The particle vector is read, not kept.
		[[nodiscard]] mesh_t generate_mesh(const std::vector<vec3d> &particles, double r) {
			mesh_t out;
			if (!_ensure()) return out;
			static_assert(sizeof(vec3d) == 24, "vec3d must be three packed doubles");
			_status = lfa_mesher_sample(_dev, reinterpret_cast<const double *>(particles.data()), particles.size(), r);
			return _status == LFA_OK ? _extract() : (_error = lfa_mesher_last_error(_dev), out);
		}
		/// The same from the particles resident in a device simulation (no host copy of the positions).
		[[nodiscard]] mesh_t generate_mesh(simulation &sim, double r) {
			mesh_t out;
			if (!_ensure() || !sim.device_handle()) return out;
			_status = lfa_mesher_sample_sim(_dev, sim.device_handle(), r);
			return _status == LFA_OK ? _extract() : (_error = lfa_mesher_last_error(_dev), out);
		}

		vec3d grid_offset;
		double cell_size = 0.0, particle_extent = 0.5;
		std::size_t cell_radius = 2;
		int device = -1;  ///< HIP device (-1: current)

		int last_status() const { return _status; }
		const std::string &last_error() const { return _error; }
	private:
		vec3s _size;
		lfa_mesher *_dev = nullptr;
		// the handle is rebuilt when a public field changed since it was created
		vec3d _dev_offset;
		double _dev_cell = 0.0, _dev_extent = 0.0;
		std::size_t _dev_radius = 0;
		int _status = LFA_OK;
		std::string _error;

		void _release() {
			if (_dev) lfa_mesher_destroy(_dev);
			_dev = nullptr;
		}
		bool _ensure() {
			const bool same = _dev && _dev_offset.x == grid_offset.x && _dev_offset.y == grid_offset.y &&
				_dev_offset.z == grid_offset.z && _dev_cell == cell_size && _dev_extent == particle_extent && _dev_radius == cell_radius;
			if (same) return true;
			_release();
			const std::uint64_t n[3] = {_size.x, _size.y, _size.z};
			const double off[3] = {grid_offset.x, grid_offset.y, grid_offset.z};
			_status = lfa_mesher_create(&_dev, n, off, cell_size, particle_extent, cell_radius, device);
			if (_status != LFA_OK) {
				_error = lfa_last_error(nullptr);
				_dev = nullptr;
				return false;
			}
			_dev_offset = grid_offset;
			_dev_cell = cell_size;
			_dev_extent = particle_extent;
			_dev_radius = cell_radius;
			return true;
		}
		mesh_t _extract() {
			mesh_t out;
			std::uint64_t nv = 0, ni = 0;
			_status = lfa_mesher_marching_cubes(_dev, &nv, &ni);
			if (_status != LFA_OK) {
				_error = lfa_mesher_last_error(_dev);
				return out;
			}
			out.positions.resize(nv);
			std::vector<std::uint64_t> idx(ni);
			_status = lfa_mesher_download_mesh(_dev, reinterpret_cast<double *>(out.positions.data()), idx.data());
			if (_status != LFA_OK) _error = lfa_mesher_last_error(_dev);
			out.indices.assign(idx.begin(), idx.end());
			return out;
		}
	};
}
