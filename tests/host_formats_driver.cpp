// Test driver for libfluid_amd/host/mesh.h and point_cloud.h, the formats testbed/main.cpp:328-347
// writes with F3 / F4. Pure host code. Built and run by tests/test_formats.py.
//   host_formats_driver points in.bin out.txt            in.bin = double[3 n]
//   host_formats_driver parse in.txt count out.bin
//   host_formats_driver obj mesh.bin mode out.obj [normals.bin]   mesh.bin = u64 nv, u64 ni, double[3 nv], u64[ni], double[2 nv]
//                                                        mode bits: 1 normals, 2 uvs, 4 reverse faces
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "../libfluid_amd/host/mesh.h"
#include "../libfluid_amd/host/point_cloud.h"

using namespace fluid_amd;

static std::vector<char> slurp(const char *path) {
	std::ifstream f(path, std::ios::binary);
	return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv) {
	if (argc < 4) return 2;
	if (!std::strcmp(argv[1], "points")) {
		std::vector<char> raw = slurp(argv[2]);
		const double *d = reinterpret_cast<const double *>(raw.data());
		std::vector<vec3d> pts(raw.size() / 24);
		for (std::size_t i = 0; i < pts.size(); ++i) pts[i] = vec3d(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
		std::ofstream out(argv[3]);
		point_cloud::save_to_naive(out, pts.begin(), pts.end());
		return 0;
	}
	if (!std::strcmp(argv[1], "parse") && argc >= 5) {
		std::ifstream in(argv[2]);
		std::vector<vec3d> pts = point_cloud::load_from_naive(in, static_cast<std::size_t>(std::strtoull(argv[3], nullptr, 10)));
		std::ofstream out(argv[4], std::ios::binary);
		for (const vec3d &p : pts) out.write(reinterpret_cast<const char *>(&p), 24);
		return 0;
	}
	if (!std::strcmp(argv[1], "obj") && argc >= 5) {
		std::vector<char> raw = slurp(argv[2]);
		const char *c = raw.data();
		std::uint64_t nv, ni;
		std::memcpy(&nv, c, 8);
		std::memcpy(&ni, c + 8, 8);
		const double *pos = reinterpret_cast<const double *>(c + 16);
		const std::uint64_t *idx = reinterpret_cast<const std::uint64_t *>(c + 16 + 24 * nv);
		const double *uv = reinterpret_cast<const double *>(c + 16 + 24 * nv + 8 * ni);
		const int mode = std::atoi(argv[3]);
		mesh<double, std::size_t, double, double, vec3d> m;
		for (std::uint64_t i = 0; i < nv; ++i) m.positions.emplace_back(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
		m.indices.assign(idx, idx + ni);
		if (mode & 2) for (std::uint64_t i = 0; i < nv; ++i) m.uvs.emplace_back(uv[2 * i], uv[2 * i + 1]);
		if (mode & 4) m.reverse_face_directions();
		if (mode & 1) m.generate_normals();
		std::ofstream out(argv[4]);
		m.save_obj(out);
		if ((mode & 1) && argc >= 6) {
			std::ofstream nout(argv[5], std::ios::binary);
			for (const vec3d &n : m.normals) nout.write(reinterpret_cast<const char *>(&n), 24);
		}
		return 0;
	}
	return 2;
}
