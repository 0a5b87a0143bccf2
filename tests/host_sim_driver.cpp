// Test driver for the C++ host class (libfluid_amd/host/simulation.h): runs N `time_step(dt)` on a particle file and writes
// the particles back. Mirrors how testbed/main.cpp:91-123,187-195 drives fluid::simulation (construct, set public fields,
// install callbacks, inject particles and solid cells, step, read particles()). Built and run by tests/test_host_class.py.
//   usage: host_sim_driver nx ny nz method blend dt steps particles_in.bin particles_out.bin solids.bin|- [mode] [cell_size]
//   mode: nocb      no callback: one lfa_time_step per step
//         callbacks the testbed's three callbacks (testbed/main.cpp:101-123): print dt; iterations, residual, max pressure from
//                   the pressure vector; max particle speed from particles() after the step (default)
//         two       the first two only (nothing touches particles() during the run)
//         edit      callbacks that EDIT device-resident state through the mutable references: the pressure vector, the grid
//                   (post_gravity: zero a face) and the particles (post_g2p: scale one velocity) - the edits must reach the device
//         window    a callback BETWEEN the P2G and the correction (post_apply_pressure) reads particles() and edits one: the class
//                   has started the correction on the second stream by then and must take it back, exactly (prints the sum of
//                   the x coordinates it saw: the positions of BEFORE the correction)
//         placed    post_advection_callback and post_correction_callback installed (they read particles(): the sum of |position -
//                   old_position| they saw is printed as "moved_advect <sum>" / "moved_correct <sum>" - non-zero only if the callback
//                   sits between the move and its collision handling, as in src/simulation.cpp:51-59,111-117)
//         obstacle  nocb, and the solid cells are only put in after the first step, then removed again before the last one
//                   (the testbed's scene reset edits sim.grid() between steps, testbed/main.cpp:125-178)
//         recreate  what the Maya node does per evaluation (plugins/maya/nodes/grid_node.cpp:256-274,350-366): `steps` times
//                   { a NEW simulation, resize(), fields, particles, update(dt), read particles() } - prints
//                   "cycle_ms <mean> resize_ms <mean>" over the cycles after the first (the first one fills the caches)
//   Prints "step_ms <mean wall milliseconds per step>" (after one warm-up step when steps > 2).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

static int recreate_cycles(int argc, char **argv) {
	const int steps = std::atoi(argv[7]);
	const double dt = std::atof(argv[6]);
	std::vector<char> in = slurp(argv[8]);
	double cycle_ms = 0.0, resize_ms = 0.0, create_ms = 0.0;
	int timed = 0;
	std::vector<simulation::particle> last;
	for (int i = 0; i < steps; ++i) {
		const auto t0 = std::chrono::steady_clock::now();
		simulation sim;
		sim.resize(vec3s(std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3])));
		const auto t1 = std::chrono::steady_clock::now();
		if (sim.last_status() != LFA_OK) {
			std::fprintf(stderr, "device init failed: %s\n", sim.last_error().c_str());
			return 3;
		}
		sim.cell_size = 1.0;
		sim.grid_offset = vec3d();
		sim.gravity = vec3d(0.0, -981.0, 0.0);
		sim.simulation_method = static_cast<simulation::method>(std::atoi(argv[4]));
		sim.blending_factor = std::atof(argv[5]);
		sim.particles().resize(in.size() / sizeof(simulation::particle));
		std::memcpy(static_cast<void*>(sim.particles().data()), in.data(), in.size());
		sim.reset_space_hash();
		sim.update(dt);
		if (sim.last_status() < 0) {
			std::fprintf(stderr, "cycle %d failed (%d): %s\n", i, sim.last_status(), sim.last_error().c_str());
			return 4;
		}
		last = sim.particles();
		const auto t2 = std::chrono::steady_clock::now();
		if (i > 0) {
			cycle_ms += std::chrono::duration<double, std::milli>(t2 - t0).count();
			resize_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
			create_ms += sim.device_create_ms();
			++timed;
		}
	}
	std::printf("cycle_ms %.4f resize_ms %.4f create_ms %.4f\n", timed ? cycle_ms / timed : 0.0, timed ? resize_ms / timed : 0.0,
	            timed ? create_ms / timed : 0.0);
	std::printf("step_ms %.4f\n", timed ? cycle_ms / timed : 0.0);
	if (FILE *f = std::fopen(argv[9], "wb")) {
		std::fwrite(last.data(), sizeof(simulation::particle), last.size(), f);
		std::fclose(f);
		return 0;
	}
	return 5;
}

int main(int argc, char **argv) {
	if (argc < 10) return 2;
	if (argc > 11 && std::string(argv[11]) == "recreate") return recreate_cycles(argc, argv);
	simulation sim;
	sim.resize(vec3s(std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3])));
	if (sim.last_status() != LFA_OK) {
		std::fprintf(stderr, "device init failed: %s\n", sim.last_error().c_str());
		return 3;
	}
	const std::string mode = argc > 11 ? argv[11] : "callbacks";
	sim.cell_size = argc > 12 ? std::atof(argv[12]) : 1.0;
	sim.grid_offset = vec3d();
	sim.gravity = vec3d(0.0, -981.0, 0.0);
	sim.simulation_method = static_cast<simulation::method>(std::atoi(argv[4]));
	sim.blending_factor = std::atof(argv[5]);
	const double dt = std::atof(argv[6]);
	const int steps = std::atoi(argv[7]);
	std::vector<char> in = slurp(argv[8]);
	sim.particles().resize(in.size() / sizeof(simulation::particle));
	std::memcpy(static_cast<void*>(sim.particles().data()), in.data(), in.size());
	std::vector<int> solids;
	if (argc > 10 && std::string(argv[10]) != "-") {
		std::vector<char> s = slurp(argv[10]);
		solids.assign(reinterpret_cast<const int*>(s.data()), reinterpret_cast<const int*>(s.data()) + s.size() / sizeof(int));
	}
	auto set_solids = [&](fluid_amd::mac_grid::cell::type t) {
		for (std::size_t i = 0; i + 2 < solids.size(); i += 3) sim.grid().grid()(solids[i], solids[i + 1], solids[i + 2]).cell_type = t;
	};
	if (mode != "obstacle") set_solids(fluid_amd::mac_grid::cell::type::solid);
	sim.reset_space_hash();
	std::size_t iters_total = 0;
	int calls = 0;
	double max_speed = 0.0;
	if (mode == "callbacks" || mode == "two" || mode == "edit" || mode == "window") {
		sim.pre_time_step_callback = [](double step) { std::printf("  time step %g\n", step); };
		sim.post_pressure_solve_callback = [&](double, std::vector<double> &p, double res, std::size_t it) {
			iters_total += it;
			++calls;
			double pmax = 0.0;
			for (double v : p) pmax = v > pmax ? v : pmax;
			std::printf("solve %d: %zu unknowns, %zu iterations, residual %.3e, max pressure %.6g\n", calls, p.size(), it, res, pmax);
			if (mode == "edit" && !p.empty()) p[0] *= 1.0;  // touches the vector without changing it: no upload
		};
	}
	if (mode == "callbacks")
		sim.post_grid_to_particle_transfer_callback = [&](double) {
			double m = 0.0;
			for (const simulation::particle &p : sim.particles()) m = std::max(m, p.velocity.squared_length());
			max_speed = std::sqrt(m);
			std::printf("    max particle velocity = %g\n", max_speed);
		};
	if (mode == "edit") {
		sim.post_gravity_callback = [&](double) { sim.grid().grid()(1, 1, 1).velocities_posface = vec3d(); };
		sim.post_grid_to_particle_transfer_callback = [&](double) { sim.particles()[0].velocity = sim.particles()[0].velocity * 0.5; };
	}
	if (mode == "window")
		sim.post_apply_pressure_callback = [&](double) {
			double sum = 0.0;
			for (const simulation::particle &p : sim.particles()) sum += p.position.x;
			std::printf("window_pos_sum %.9f\n", sum);
			sim.particles()[0].velocity = sim.particles()[0].velocity * 0.5;
		};
	double moved_advect = 0.0, moved_correct = 0.0;
	if (mode == "placed") {
		auto moved = [&]() {
			double sum = 0.0;
			for (const simulation::particle &p : sim.particles()) sum += std::sqrt((p.position - p.old_position).squared_length());
			return sum;
		};
		sim.post_advection_callback = [&](double) { moved_advect += moved(); };
		sim.post_correction_callback = [&](double) { moved_correct += moved(); };
	}
	double wall_ms = 0.0;
	int timed = 0;
	for (int i = 0; i < steps; ++i) {
		if (mode == "obstacle" && i == 1) set_solids(fluid_amd::mac_grid::cell::type::solid);
		if (mode == "obstacle" && i == steps - 1) set_solids(fluid_amd::mac_grid::cell::type::air);
		const auto t0 = std::chrono::steady_clock::now();
		sim.time_step(dt);
		if (sim.device_handle()) lfa_synchronize(sim.device_handle());
		const auto t1 = std::chrono::steady_clock::now();
		if (i > 0 || steps <= 2) { wall_ms += std::chrono::duration<double, std::milli>(t1 - t0).count(); ++timed; }
		if (sim.last_status() < 0) {
			std::fprintf(stderr, "step %d failed (%d): %s\n", i, sim.last_status(), sim.last_error().c_str());
			return 4;
		}
	}
	if (mode == "placed") std::printf("moved_advect %.9g moved_correct %.9g\n", moved_advect, moved_correct);
	std::printf("step_ms %.4f\n", timed ? wall_ms / timed : 0.0);
	std::printf("cfl %.9g fluid type of cell0 %d\n", sim.cfl(), static_cast<int>(sim.grid().grid()[0].cell_type));
	if (FILE *f = std::fopen(argv[9], "wb")) {
		std::fwrite(sim.particles().data(), sizeof(simulation::particle), sim.particles().size(), f);
		std::fclose(f);
	} else {
		return 5;
	}
	return 0;
}
