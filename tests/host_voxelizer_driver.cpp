// Test driver for libfluid_amd/host/voxelizer.h: drives fluid_amd::voxelizer the way the reference's hosts drive
// fluid::voxelizer (src/data_structures/obstacle.cpp:12-29, plugins/maya/nodes/voxelizer_node.cpp:255-343), once through
// the staged members (resize_reposition_grid_constrained, voxelize_mesh_surface, mark_exterior) and once through
// fluid_amd::obstacle, and writes voxel types + the obstacle cells. Built and run by tests/test_host_voxelizer.py.
//   usage: host_voxelizer_driver mesh.bin cell_size ox oy oz rx ry rz types_out.bin cells_out.bin
//   mesh.bin: uint64 nv, uint64 ni, double[3 nv], uint64[ni]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../libfluid_amd/host/voxelizer.h"

using namespace fluid_amd;

int main(int argc, char **argv) {
	if (argc < 11) return 2;
	FILE *f = std::fopen(argv[1], "rb");
	if (!f) return 2;
	std::uint64_t nv = 0, ni = 0;
	if (std::fread(&nv, 8, 1, f) != 1 || std::fread(&ni, 8, 1, f) != 1) return 2;
	obstacle::mesh_t m;
	m.positions.resize(nv);
	m.indices.resize(ni);
	std::vector<double> p(3 * nv);
	if (nv && std::fread(p.data(), 24, nv, f) != nv) return 2;
	for (std::uint64_t i = 0; i < nv; ++i) m.positions[i] = vec3d(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
	std::vector<std::uint64_t> idx(ni);
	if (ni && std::fread(idx.data(), 8, ni, f) != ni) return 2;
	m.indices.assign(idx.begin(), idx.end());
	std::fclose(f);
	const double cs = std::atof(argv[2]);
	const vec3d off(std::atof(argv[3]), std::atof(argv[4]), std::atof(argv[5]));
	const vec3s ref(std::atoi(argv[6]), std::atoi(argv[7]), std::atoi(argv[8]));

	auto [bmin, bmax] = voxelizer::get_bounding_box(m.positions.begin(), m.positions.end());
	voxelizer vox;
	vec3i o = vox.resize_reposition_grid_constrained(bmin, bmax, cs, off);
	vox.voxelize_mesh_surface(m);
	vox.mark_exterior();
	if (vox.last_status() != LFA_OK) {
		std::fprintf(stderr, "voxelizer failed: %s\n", vox.last_error().c_str());
		return 3;
	}
	auto range = vox.get_overlapping_cell_range(o, ref);
	vec3s n = vox.voxels.get_size();
	std::printf("%d %d %d %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", o.x, o.y, o.z, n.x, n.y, n.z, range.first.x, range.first.y,
	            range.first.z, range.second.x, range.second.y, range.second.z);
	f = std::fopen(argv[9], "wb");
	std::fwrite(detail::cell_data(vox.voxels), 1, detail::cell_count(vox.voxels), f);
	std::fclose(f);

	obstacle obs(m, cs, off, ref);
	if (obs.status != LFA_OK) return 3;
	std::vector<std::int32_t> cells;
	for (vec3s c : obs.cells) {
		cells.push_back(static_cast<std::int32_t>(c.x));
		cells.push_back(static_cast<std::int32_t>(c.y));
		cells.push_back(static_cast<std::int32_t>(c.z));
	}
	f = std::fopen(argv[10], "wb");
	std::fwrite(cells.data(), 4, cells.size(), f);
	std::fclose(f);
	return 0;
}
