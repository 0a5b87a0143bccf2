// Test driver for the C++ host class (libfluid_amd/host/simulation.h): runs N `time_step(dt)` on a particle file and writes
// the particles back. Mirrors how testbed/main.cpp:91-99,187-195 drives fluid::simulation (construct, set public fields,
// inject particles and solid cells, step, read particles()). Built and run by tests/test_host_class.py.
//   usage: host_sim_driver nx ny nz method blend dt steps particles_in.bin particles_out.bin solids.bin|- [nocb]
//   `nocb`: no callback is installed, so the class runs whole steps on the device (lfa_time_step) and keeps the
//   particles there until particles() is read at the end
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../libfluid_amd/host/simulation.h"

using fluid_amd::simulation;
using fluid_amd::vec3d;
using fluid_amd::vec3s;

static std::vector<char> slurp(const char *path) {
	std::vector<char> buf;
	if (FILE *f = std::fopen(path, "rb")) {
		std::fseek(f, 0, SEEK_END);
		long n = std::ftell(f);
		std::fseek(f, 0, SEEK_SET);
		buf.resize(static_cast<std::size_t>(n));
		if (n && std::fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear();
		std::fclose(f);
	}
	return buf;
}

int main(int argc, char **argv) {
	if (argc < 10) return 2;
	simulation sim;
	sim.resize(vec3s(std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3])));
	if (sim.last_status() != LFA_OK) {
		std::fprintf(stderr, "device init failed: %s\n", sim.last_error().c_str());
		return 3;
	}
	sim.cell_size = 1.0;
	sim.grid_offset = vec3d();
	sim.gravity = vec3d(0.0, -981.0, 0.0);
	sim.simulation_method = static_cast<simulation::method>(std::atoi(argv[4]));
	sim.blending_factor = std::atof(argv[5]);
	const double dt = std::atof(argv[6]);
	const int steps = std::atoi(argv[7]);
	std::vector<char> in = slurp(argv[8]);
	sim.particles().resize(in.size() / sizeof(simulation::particle));
	std::memcpy(static_cast<void*>(sim.particles().data()), in.data(), in.size());
	const bool with_callback = !(argc > 11 && std::string(argv[11]) == "nocb");
	if (argc > 10 && std::string(argv[10]) != "-") {
		std::vector<char> s = slurp(argv[10]);
		const int *xyz = reinterpret_cast<const int*>(s.data());
		for (std::size_t i = 0; i + 2 < s.size() / sizeof(int); i += 3)
			sim.grid().grid()(xyz[i], xyz[i + 1], xyz[i + 2]).cell_type = fluid_amd::mac_grid::cell::type::solid;
	}
	sim.reset_space_hash();
	std::size_t iters_total = 0;
	int calls = 0;
	if (with_callback)
		sim.post_pressure_solve_callback = [&](double, std::vector<double> &p, double res, std::size_t it) {
			iters_total += it;
			++calls;
			std::printf("solve %d: %zu unknowns, %zu iterations, residual %.3e\n", calls, p.size(), it, res);
		};
	for (int i = 0; i < steps; ++i) {
		sim.time_step(dt);
		if (sim.last_status() < 0) {
			std::fprintf(stderr, "step %d failed (%d): %s\n", i, sim.last_status(), sim.last_error().c_str());
			return 4;
		}
	}
	std::printf("cfl %.9g fluid type of cell0 %d\n", sim.cfl(), static_cast<int>(sim.grid().grid()[0].cell_type));
	if (FILE *f = std::fopen(argv[9], "wb")) {
		std::fwrite(sim.particles().data(), sizeof(simulation::particle), sim.particles().size(), f);
		std::fclose(f);
	} else {
		return 5;
	}
	return 0;
}
