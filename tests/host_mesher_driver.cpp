// Test driver for libfluid_amd/host/mesher.h: drives fluid_amd::mesher the way testbed/main.cpp:101-113,328-334 drives
// fluid::mesher (resize, public fields, generate_mesh, save_obj). Built and run by tests/test_host_mesher.py.
//   usage: host_mesher_driver particles.bin nx ny nz ox oy oz cell_size extent radius r mesh_out.bin mesh_out.obj
//   particles.bin = double[3 n]; mesh_out.bin = u64 nv, u64 ni, double[3 nv], u64[ni]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "../libfluid_amd/host/mesher.h"

using namespace fluid_amd;

int main(int argc, char **argv) {
	if (argc < 14) return 2;
	std::ifstream in(argv[1], std::ios::binary);
	std::vector<char> raw((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	const double *d = reinterpret_cast<const double *>(raw.data());
	std::vector<vec3d> pts(raw.size() / 24);
	for (std::size_t i = 0; i < pts.size(); ++i) pts[i] = vec3d(d[3 * i], d[3 * i + 1], d[3 * i + 2]);

	mesher m;
	m.resize(vec3s(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4])));
	m.grid_offset = vec3d(std::atof(argv[5]), std::atof(argv[6]), std::atof(argv[7]));
	m.cell_size = std::atof(argv[8]);
	m.particle_extent = std::atof(argv[9]);
	m.cell_radius = static_cast<std::size_t>(std::atoi(argv[10]));
	mesher::mesh_t mesh = m.generate_mesh(pts, std::atof(argv[11]));
	if (m.last_status() != LFA_OK) {
		std::fprintf(stderr, "mesher failed: %s\n", m.last_error().c_str());
		return 3;
	}
	// a second call on the same object with a changed public field rebuilds the device handle
	m.particle_extent *= 1.0;
	mesher::mesh_t again = m.generate_mesh(pts, std::atof(argv[11]));
	if (again.positions.size() != mesh.positions.size() || again.indices != mesh.indices) return 4;

	std::ofstream out(argv[12], std::ios::binary);
	const std::uint64_t nv = mesh.positions.size(), ni = mesh.indices.size();
	out.write(reinterpret_cast<const char *>(&nv), 8);
	out.write(reinterpret_cast<const char *>(&ni), 8);
	out.write(reinterpret_cast<const char *>(mesh.positions.data()), 24 * nv);
	for (std::size_t i : mesh.indices) {
		const std::uint64_t v = i;
		out.write(reinterpret_cast<const char *>(&v), 8);
	}
	std::ofstream obj(argv[13]);
	mesh.save_obj(obj);
	return 0;
}
