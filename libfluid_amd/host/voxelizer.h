// libfluid_amd/host/voxelizer.h -- C++17 host classes with the public surface of lukedan/libfluid's `fluid::voxelizer`
// (include/fluid/voxelizer.h:14-74) and `fluid::obstacle` (include/fluid/data_structures/obstacle.h:11-21), running the
// triangle marking and the exterior flood fill on an MI355X through the C ABI (include/libfluid_amd.h, lfa_voxels_*).
// Header-only; link with libfluid_amd.so. `namespace fluid_amd` mirrors `namespace fluid`.
//
// Drop-in contract: same member names and argument meaning as the reference --
//   get_bounding_box :24-34  resize_reposition_grid :38  resize_reposition_grid_constrained :44
//   get_overlapping_cell_range :48  voxelize_triangle :52  voxelize_mesh_surface :55-63  mark_exterior :66
//   cell_size / voxels / grid_offset :68-70 (public, host-visible between calls)
// `voxels` stays a host grid3<cell_type> that callers may read or edit between the calls, exactly as with the reference;
// every device stage uploads it, runs, and downloads it again (1 byte per voxel). `voxelize_mesh(mesh, ...)` is the
// sequence both hosts of the reference run (obstacle.cpp:12-18, voxelizer_node.cpp:255-268) in one device pass.
// The class never throws; device errors are kept in last_status()/last_error().
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/libfluid_amd.h"
#include "mesh.h"

namespace fluid_amd {
	class voxelizer {
	public:
		enum class cell_type : unsigned char { interior, exterior, surface };  // voxelizer.h:17-21 == LFA_VOX_*

		template <typename It> static std::pair<vec3d, vec3d> get_bounding_box(It beg, It end) {
			if (beg == end) return {vec3d(), vec3d()};
			vec3d mn(*beg), mx(*beg);
			auto it = beg;
			for (++it; it != end; ++it) _update_bounding_box(vec3d(*it), mn, mx);
			return {mn, mx};
		}

		/// src/voxelizer.cpp:12-20
		void resize_reposition_grid(vec3d min, vec3d max) {
			vec3d size = max - min, gs(std::ceil(size.x / cell_size), std::ceil(size.y / cell_size), std::ceil(size.z / cell_size));
			grid_offset = min - 0.5 * (gs * cell_size - size) - vec3d(cell_size, cell_size, cell_size);
			voxels = grid3<cell_type>(vec3s(gs) + vec3s(2, 2, 2), cell_type::interior);
		}
		/// src/voxelizer.cpp:22-39
		vec3i resize_reposition_grid_constrained(vec3d min, vec3d max, double ref_cell_size, vec3d ref_grid_offset) {
			cell_size = ref_cell_size;
			vec3d lo = (min - ref_grid_offset) / cell_size, hi = (max - ref_grid_offset) / cell_size;
			vec3i grid_min(static_cast<int>(std::floor(lo.x)), static_cast<int>(std::floor(lo.y)), static_cast<int>(std::floor(lo.z)));
			vec3i grid_max(static_cast<int>(std::ceil(hi.x)), static_cast<int>(std::ceil(hi.y)), static_cast<int>(std::ceil(hi.z)));
			grid_min -= vec3i(1, 1, 1);
			grid_max += vec3i(1, 1, 1);
			grid_offset = ref_grid_offset + vec3d(grid_min) * cell_size;
			voxels = grid3<cell_type>(vec3s(grid_max - grid_min), cell_type::interior);
			return grid_min;
		}
		/// The five-argument form the Maya VoxelizerNode calls (plugins/maya/nodes/voxelizer_node.cpp:261-270: it also passes the
		/// reference grid's size, which the reference's own header no longer takes - include/fluid/voxelizer.h:44); the size
		/// plays no part in the placement (it is used afterwards, by get_overlapping_cell_range).
		vec3i resize_reposition_grid_constrained(vec3d min, vec3d max, double ref_cell_size, vec3d ref_grid_offset, vec3s) {
			return resize_reposition_grid_constrained(min, max, ref_cell_size, ref_grid_offset);
		}
		/// src/voxelizer.cpp:41-57
		std::pair<vec3s, vec3s> get_overlapping_cell_range(vec3i offset, vec3s ref_grid_size) const {
			vec3s mn, mx, n = voxels.get_size();
			for (std::size_t d = 0; d < 3; ++d) {
				mn[d] = offset[d] < 0 ? static_cast<std::size_t>(-offset[d]) : 0;
				const int hi = offset[d] + static_cast<int>(n[d]);
				const std::size_t c = static_cast<std::size_t>(hi > 0 ? hi : 0);
				mx[d] = c < ref_grid_size[d] ? c : ref_grid_size[d];
			}
			return {mn, mx};
		}

		/// Marks the cells that overlap the triangle (src/voxelizer.cpp:54-81).
		void voxelize_triangle(vec3d p1, vec3d p2, vec3d p3) {
			const double pos[9] = {p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p3.x, p3.y, p3.z};
			const std::uint32_t idx[3] = {0, 1, 2};
			_on_device([&](lfa_voxels *v) { return lfa_voxels_voxelize_triangles(v, pos, 3, idx, 4, 3); });
		}
		/// include/fluid/voxelizer.h:55-63: all triangles of the mesh in one launch.
		template <typename Mesh> void voxelize_mesh_surface(const Mesh &m) {
			std::vector<double> pos;
			std::vector<std::uint64_t> idx;
			_flatten(m, pos, idx);
			_on_device([&](lfa_voxels *v) {
				return lfa_voxels_voxelize_triangles(v, pos.data(), pos.size() / 3, idx.data(), 8, idx.size());
			});
		}
		/// src/voxelizer.cpp:83-124
		void mark_exterior() {
			_on_device([](lfa_voxels *v) { return lfa_voxels_mark_exterior(v); });
		}

		/// get_bounding_box + resize_reposition_grid_constrained + voxelize_mesh_surface + mark_exterior
		/// (obstacle.cpp:12-18, voxelizer_node.cpp:255-268) without intermediate host copies. Returns the grid offset.
		template <typename Mesh> vec3i voxelize_mesh(const Mesh &m, double ref_cell_size, vec3d ref_grid_offset) {
			std::vector<double> pos;
			std::vector<std::uint64_t> idx;
			_flatten(m, pos, idx);
			const double off[3] = {ref_grid_offset.x, ref_grid_offset.y, ref_grid_offset.z};
			lfa_voxels *v = nullptr;
			_status = lfa_voxelize_mesh(&v, pos.data(), pos.size() / 3, idx.data(), 8, idx.size(), ref_cell_size, off, device);
			if (_status != LFA_OK) {
				_error = lfa_last_error(nullptr);
				return vec3i();
			}
			std::int32_t gmin[3];
			std::uint64_t n[3];
			double goff[3];
			lfa_voxels_info(v, gmin, n, goff, &cell_size);
			grid_offset = vec3d(goff[0], goff[1], goff[2]);
			voxels = grid3<cell_type>(vec3s(n[0], n[1], n[2]), cell_type::interior);
			_status = lfa_voxels_download(v, reinterpret_cast<std::uint8_t *>(detail::cell_data(voxels)));
			if (_status != LFA_OK) _error = lfa_voxels_last_error(v);
			lfa_voxels_destroy(v);
			return vec3i(gmin[0], gmin[1], gmin[2]);
		}

		double cell_size = 1.0;
		grid3<cell_type> voxels;
		vec3d grid_offset;
		int device = -1;  ///< HIP device (-1: current)

		int last_status() const { return _status; }
		const std::string &last_error() const { return _error; }
	private:
		int _status = LFA_OK;
		std::string _error;

		static void _update_bounding_box(vec3d p, vec3d &mn, vec3d &mx) {
			for (std::size_t d = 0; d < 3; ++d) {
				mn[d] = std::min(mn[d], p[d]);
				mx[d] = std::max(mx[d], p[d]);
			}
		}
		template <typename Mesh> static void _flatten(const Mesh &m, std::vector<double> &pos, std::vector<std::uint64_t> &idx) {
			pos.reserve(m.positions.size() * 3);
			for (const auto &p : m.positions) {
				pos.push_back(static_cast<double>(p.x));
				pos.push_back(static_cast<double>(p.y));
				pos.push_back(static_cast<double>(p.z));
			}
			idx.assign(m.indices.begin(), m.indices.end());
		}
		/// voxels -> device, stage, device -> voxels.
		template <typename Stage> void _on_device(Stage &&stage) {
			const vec3s n = voxels.get_size();
			if (detail::cell_count(voxels) == 0) return;
			const std::uint64_t size[3] = {n.x, n.y, n.z};
			const double off[3] = {grid_offset.x, grid_offset.y, grid_offset.z};
			lfa_voxels *v = nullptr;
			_status = lfa_voxels_create(&v, size, off, cell_size, device);
			if (_status != LFA_OK) {
				_error = lfa_last_error(nullptr);
				return;
			}
			_status = lfa_voxels_upload(v, reinterpret_cast<const std::uint8_t *>(detail::cell_data(voxels)));
			if (_status == LFA_OK) _status = stage(v);
			if (_status == LFA_OK) _status = lfa_voxels_download(v, reinterpret_cast<std::uint8_t *>(detail::cell_data(voxels)));
			if (_status != LFA_OK) _error = lfa_voxels_last_error(v);
			lfa_voxels_destroy(v);
		}
	};
	static_assert(sizeof(voxelizer::cell_type) == 1, "voxel types cross the C ABI as bytes");

	/// fluid::obstacle (include/fluid/data_structures/obstacle.h:11-21): the mesh and the reference-grid cells it fully
	/// occupies. The list is the interior voxels inside the reference grid in grid3 order, i.e. what
	/// obstacle.cpp:20-28 means and voxelizer_node.cpp:325-343 ("cells_ref", interior only) computes; the reference's
	/// own loop bounds mix voxel- and reference-grid coordinates and are not reproduced (include/libfluid_amd.h).
	class obstacle {
	public:
		using mesh_t = mesh<double, std::size_t, double, double, vec3d>;  // obstacle.h:13
		obstacle() = default;
		obstacle(mesh_t m, double cell_size, vec3d ref_grid_offset, vec3s ref_grid_size) : obstacle_mesh(std::move(m)) {
			voxelizer vox;
			vec3i offset = vox.voxelize_mesh(obstacle_mesh, cell_size, ref_grid_offset);
			status = vox.last_status();
			if (status != LFA_OK) return;
			vox.voxels.for_each([&](vec3s p, voxelizer::cell_type t) {
				if (t != voxelizer::cell_type::interior) return;
				const long long c[3] = {static_cast<long long>(p.x) + offset.x, static_cast<long long>(p.y) + offset.y,
				                        static_cast<long long>(p.z) + offset.z};
				for (std::size_t d = 0; d < 3; ++d)
					if (c[d] < 0 || c[d] >= static_cast<long long>(ref_grid_size[d])) return;
				cells.emplace_back(vec3s(static_cast<std::size_t>(c[0]), static_cast<std::size_t>(c[1]), static_cast<std::size_t>(c[2])));
			});
		}
		mesh_t obstacle_mesh;
		std::vector<vec3s> cells;
		int status = LFA_OK;
	};
}
