// tests/callers/reference_callers.cpp -- host code written the way the two hosts of lukedan/libfluid are written, against the
// REFERENCE'S OWN HEADERS AND TYPES (`#include <fluid/simulation.h>`, `fluid::vec3d`, `fluid::grid3`, `fluid::mac_grid::cell`,
// `fluid::source`, `fluid::mesher::mesh_t`, `fluid::vec_ops::dot` ...). It is the proof of the drop-in boundary:
// this one file is compiled, unchanged,
//   (1) against the real reference       : -I/root/reference/include + the reference's sources (oracle/Makefile, target callers_ref)
//   (2) against this repository's device path: -I libfluid_amd/host/shim -I/root/reference/include, linked with libfluid_amd.so only
//   (3) the same without any libfluid checkout: -I libfluid_amd/host/shim -I libfluid_amd/host/shim_standalone (the GPU box)
// and tests/test_ref_callers.py compares what (2) and (3) compute on the MI355X with what (1) computes on the CPU.
// It is written from scratch, but it CALLS the reference's API the way its hosts do, so the lines that are nothing but such a call
// necessarily coincide with theirs (about two dozen: the callback bodies and print strings of testbed/main.cpp:54-121, the
// seed_box / seed_sphere / update / time_step calls of its scenes, the voxelizer node's resize_reposition_grid_constrained /
// voxelize_mesh_surface / mark_exterior sequence). Each scenario names the host code whose calls it makes:
//   testbed    testbed/main.cpp:50-88 (update_simulation), :90-123 (set-up + the three callbacks), :125-185 (scene reset, scenes
//              0-4), :187-195 (update(1/60) / time_step())
//   gridnode   plugins/maya/nodes/grid_node.cpp:256-274 (fresh simulation per evaluation, fields from attributes), :275-343 (sources
//              and obstacle cells from flat int[3k] arrays), :345-366 (particles moved in, hash_particles, update(frame), positions out,
//              particles moved out)
//   mesher     testbed/main.cpp:203-232 (mesher_thread), :328-334 (F3: save_obj)
//   voxelizer  plugins/maya/nodes/voxelizer_node.cpp:222-343 (mesh<double,int,...>, bounding box, constrained grid, surface,
//              exterior, the `cells` / `cells_ref` lists) and src/data_structures/obstacle.cpp:9-29 (fluid::obstacle)
//   points     testbed/main.cpp:335-347 (F4: point_cloud::save_to_naive)
// usage: reference_callers <scenario> <outdir> [numbers...]      (see each function)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <fluid/simulation.h>
#include <fluid/data_structures/grid.h>
#include <fluid/mesher.h>
#include <fluid/voxelizer.h>
#include <fluid/data_structures/obstacle.h>
#include <fluid/data_structures/point_cloud.h>
#include <fluid/data_structures/mesh.h>

using fluid::vec3d;
using fluid::vec3i;
using fluid::vec3s;

namespace {
	// ---- little binary writer: every record is  name\0  u64 count  f64[count]
	struct dump {
		explicit dump(const std::string &path) : out(path, std::ios::binary) {}
		void put(const char *name, const std::vector<double> &v) {
			out.write(name, static_cast<std::streamsize>(std::strlen(name) + 1));
			const std::uint64_t n = v.size();
			out.write(reinterpret_cast<const char *>(&n), 8);
			out.write(reinterpret_cast<const char *>(v.data()), static_cast<std::streamsize>(8 * n));
		}
		void put(const char *name, double v) { put(name, std::vector<double>{v}); }
		std::ofstream out;
	};
	std::vector<char> slurp(const std::string &path) {
		std::ifstream f(path, std::ios::binary);
		return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	}

	// ============================================================================================== testbed
	struct published {
		std::vector<fluid::simulation::particle> particles;
		fluid::grid3<std::size_t> occupation;
		fluid::grid3<vec3d> velocities;
		double energy = 0.0;
	};
	/// What the testbed publishes to its render thread after every step.
	published publish(const fluid::simulation &sim) {
		published out;
		out.particles.assign(sim.particles().begin(), sim.particles().end());
		for (fluid::simulation::particle &p : out.particles) {
			out.energy += 0.5 * p.velocity.squared_length();
			out.energy -= fluid::vec_ops::dot(sim.gravity, p.position);
		}
		out.occupation = fluid::grid3<std::size_t>(sim.grid().grid().get_size(), 0);
		for (const auto &p : sim.particles()) {
			vec3s cell(vec3i((p.position - sim.grid_offset) / sim.cell_size));
			const vec3s n = out.occupation.get_size();
			if (cell.x < n.x && cell.y < n.y && cell.z < n.z) ++out.occupation(cell);
		}
		out.velocities = fluid::grid3<vec3d>(sim.grid().grid().get_size());
		const vec3s n = out.velocities.get_size();
		for (std::size_t z = 0; z < n.z; ++z)
			for (std::size_t y = 0; y < n.y; ++y)
				for (std::size_t x = 0; x < n.x; ++x) out.velocities(x, y, z) = sim.grid().grid()(x, y, z).velocities_posface;
		return out;
	}
	void write_published(dump &d, const published &pub, const std::string &tag) {
		std::vector<double> pos, vel, occ, gv;
		for (const auto &p : pub.particles) {
			for (std::size_t k = 0; k < 3; ++k) { pos.push_back(p.position[k]); vel.push_back(p.velocity[k]); }
		}
		const vec3s n = pub.occupation.get_size();
		for (std::size_t i = 0; i < n.x * n.y * n.z; ++i) {
			occ.push_back(static_cast<double>(pub.occupation[i]));
			for (std::size_t k = 0; k < 3; ++k) gv.push_back(pub.velocities[i][k]);
		}
		d.put((tag + ".pos").c_str(), pos);
		d.put((tag + ".vel").c_str(), vel);
		d.put((tag + ".occupation").c_str(), occ);
		d.put((tag + ".grid_vel").c_str(), gv);
		d.put((tag + ".energy").c_str(), pub.energy);
	}

	/// testbed <outdir> <grid n> <scene 0-4> <frames> : scene set-up scaled from the testbed's 50^3 grid to n^3.
	int testbed(const std::string &outdir, int argc, char **argv) {
		const std::size_t n = argc > 0 ? std::strtoull(argv[0], nullptr, 10) : 50;
		const int scene = argc > 1 ? std::atoi(argv[1]) : 0;
		const int frames = argc > 2 ? std::atoi(argv[2]) : 2;
		const double s = static_cast<double>(n) / 50.0;  // the testbed's coordinates are for 50^3
		const double cell = 1.0;

		fluid::simulation sim;
		sim.resize(vec3s(n, n, n));
		sim.grid_offset = vec3d();
		sim.cell_size = cell;
		sim.simulation_method = fluid::simulation::method::apic;
		sim.blending_factor = 1.0;
		sim.gravity = vec3d(0.0, -981.0, 0.0);

		std::vector<double> dts, iterations, residuals, max_pressures, max_speeds;
		sim.pre_time_step_callback = [&](double dt) {
			std::cout << "  time step " << dt << "\n";
			dts.push_back(dt);
		};
		sim.post_pressure_solve_callback = [&](double, std::vector<double> &pressure, double residual, std::size_t iters) {
			std::cout << "    iterations = " << iters << "\n";
			if (iters > 100) std::cout << "*** WARNING: large number of iterations\n";
			std::cout << "    residual = " << residual << "\n";
			auto top = std::max_element(pressure.begin(), pressure.end());
			if (top != pressure.end()) std::cout << "    max pressure = " << *top << "\n";
			iterations.push_back(static_cast<double>(iters));
			residuals.push_back(residual);
			max_pressures.push_back(top != pressure.end() ? *top : 0.0);
		};
		sim.post_grid_to_particle_transfer_callback = [&sim, &max_speeds](double) {
			double fastest = 0.0;
			for (const fluid::simulation::particle &p : sim.particles()) fastest = std::max(fastest, p.velocity.squared_length());
			std::cout << "    max particle velocity = " << std::sqrt(fastest) << "\n";
			max_speeds.push_back(std::sqrt(fastest));
		};

		// ---- scene reset
		sim.particles().clear();
		sim.grid().grid().for_each([](vec3s, fluid::mac_grid::cell &c) { c.cell_type = fluid::mac_grid::cell::type::air; });
		sim.sources.clear();
		switch (scene) {
		case 0:
			sim.seed_box(vec3d(15, 15, 15) * s, vec3d(20, 20, 20) * s);
			break;
		case 1:
			sim.seed_sphere(vec3d(25.0, 25.0, 25.0) * s, 15.0 * s);
			break;
		case 2:
			sim.seed_sphere(vec3d(25, 44, 25) * s, 5 * s);
			sim.seed_box(vec3d(0, 0, 0), vec3d(50, 15, 50) * s);
			break;
		case 3:
			sim.seed_box(vec3d(0, 0, 0), vec3d(10, 50, 50) * s);
			break;
		case 4: {
			auto inflow = std::make_unique<fluid::source>();
			for (std::size_t x = 1; x < 5; ++x)
				for (std::size_t y = n / 2; y < n / 2 + n / 5; ++y)
					for (std::size_t z = 2 * n / 5; z < 3 * n / 5; ++z) inflow->cells.emplace_back(x, y, z);
			inflow->velocity = vec3d(200.0, 0.0, 0.0);
			inflow->coerce_velocity = true;
			sim.sources.emplace_back(std::move(inflow));
			const vec3d centre = vec3d(25.0, 25.0, 25.0) * s;
			const double r2 = 100.0 * s * s;
			sim.grid().grid().for_each_in_range_unchecked(
				[&](vec3s at, fluid::mac_grid::cell &c) {
					vec3d d = vec3d(at) + 0.5 * vec3d(cell, cell, cell);
					d -= centre;
					if (d.squared_length() < r2) c.cell_type = fluid::mac_grid::cell::type::solid;
				},
				vec3s(3 * n / 10, 3 * n / 10, 3 * n / 10), vec3s(7 * n / 10, 7 * n / 10, 7 * n / 10));
			break;
		}
		default:
			return 2;
		}
		sim.reset_space_hash();

		dump d(outdir + "/testbed.bin");
		write_published(d, publish(sim), "frame0");
		for (int f = 1; f <= frames; ++f) {
			if (f < frames || frames == 1) {
				std::cout << "update\n";
				sim.update(1.0 / 60.0);
			} else {
				sim.time_step();  // the testbed's single-step key
			}
			write_published(d, publish(sim), "frame" + std::to_string(f));
		}
		d.put("dts", dts);
		d.put("iterations", iterations);
		d.put("residuals", residuals);
		d.put("max_pressures", max_pressures);
		d.put("max_speeds", max_speeds);
		return 0;
	}

	// ============================================================================================== Maya GridNode
	/// The attribute values a GridNode evaluation reads (plain arrays stand in for MDataHandle's int3 / double3 / MIntArray).
	struct node_attributes {
		double cell_size = 1.0;
		int grid_size[3] = {0, 0, 0};
		double grid_offset[3] = {0, 0, 0}, gravity[3] = {0, 0, 0};
		short transfer_method = 2;
		struct source_attr {
			std::vector<int> cells;  // x, y, z triples
			float velocity[3] = {0, 0, 0};
			bool enabled = true, coerce_velocity = false;
			int seeding_density = 2;
		};
		std::vector<source_attr> sources;
		std::vector<std::vector<int>> obstacles;  // x, y, z triples each
	};
	struct grid_node_state {
		std::vector<std::vector<double>> particle_cache;  // per frame: x, y, z of every particle (the MPointArray)
		std::vector<fluid::simulation::particle> last_frame_particles;
	};
	/// One GridNode::compute for a frame that is not cached yet.
	bool evaluate(grid_node_state &node, const node_attributes &attr, std::size_t frame, double frame_time) {
		using fluid::simulation;
		using fluid::mac_grid;
		if (frame >= node.particle_cache.size()) {
			simulation sim;
			sim.cell_size = attr.cell_size;
			for (std::size_t i = 0; i < 3; ++i)
				if (attr.grid_size[i] < 0) return false;
			sim.resize(vec3s(vec3i(attr.grid_size[0], attr.grid_size[1], attr.grid_size[2])));
			sim.grid_offset = vec3d(attr.grid_offset[0], attr.grid_offset[1], attr.grid_offset[2]);
			sim.gravity = vec3d(attr.gravity[0], attr.gravity[1], attr.gravity[2]);
			sim.simulation_method = static_cast<simulation::method>(attr.transfer_method);
			for (const node_attributes::source_attr &sa : attr.sources) {
				auto src = std::make_unique<fluid::source>();
				src->active = sa.enabled;
				src->cells.resize(sa.cells.size() / 3);
				std::size_t at = 0;
				for (vec3s &c : src->cells) {
					c.x = static_cast<std::size_t>(sa.cells[at++]);
					c.y = static_cast<std::size_t>(sa.cells[at++]);
					c.z = static_cast<std::size_t>(sa.cells[at++]);
				}
				src->coerce_velocity = sa.coerce_velocity;
				src->target_density_cubic_root = sa.seeding_density;
				src->velocity = vec3d(sa.velocity[0], sa.velocity[1], sa.velocity[2]);
				sim.sources.emplace_back(std::move(src));
			}
			for (const std::vector<int> &cells : attr.obstacles) {
				std::size_t at = 0;
				for (std::size_t i = 0; i < cells.size() / 3; ++i) {
					vec3s c;
					c.x = static_cast<std::size_t>(cells[at++]);
					c.y = static_cast<std::size_t>(cells[at++]);
					c.z = static_cast<std::size_t>(cells[at++]);
					sim.grid().grid()(c).cell_type = mac_grid::cell::type::solid;
				}
			}
			if (node.particle_cache.size() > 0) sim.particles() = std::move(node.last_frame_particles);
			sim.hash_particles();
			do {
				sim.update(frame_time);
				std::vector<double> &points = node.particle_cache.emplace_back(3 * sim.particles().size());
				std::size_t i = 0;
				for (const simulation::particle &p : sim.particles()) {
					points[i++] = p.position.x;
					points[i++] = p.position.y;
					points[i++] = p.position.z;
				}
			} while (frame >= node.particle_cache.size());
			node.last_frame_particles = std::move(sim.particles());
		}
		return true;
	}
	/// gridnode <outdir> <particles.bin|-> <n> <frames> <method> : `particles.bin` (152-byte records) plays the particles a
	/// previous evaluation left behind; with "-" everything comes from the source.
	int gridnode(const std::string &outdir, int argc, char **argv) {
		if (argc < 4) return 2;
		const std::string start = argv[0];
		const int n = std::atoi(argv[1]), frames = std::atoi(argv[2]);
		node_attributes attr;
		attr.cell_size = 0.5;  // the plugin's users pick their own cell size and offset
		attr.grid_size[0] = attr.grid_size[1] = attr.grid_size[2] = n;
		attr.grid_offset[0] = -1.0; attr.grid_offset[1] = 0.25; attr.grid_offset[2] = 2.0;
		attr.gravity[1] = -981.0;
		attr.transfer_method = static_cast<short>(std::atoi(argv[3]));
		std::vector<int> wall;  // an obstacle: a thick wall with a gap, the way voxelizer_node's `cells_ref` output lists one
		for (int z = 0; z < n; ++z)
			for (int y = 0; y < n / 2; ++y)
				for (int x = 2 * n / 3; x < 2 * n / 3 + 2; ++x)
					if (z < n / 3 || z >= n / 2) { wall.push_back(x); wall.push_back(y); wall.push_back(z); }
		attr.obstacles.push_back(wall);
		grid_node_state node;
		if (start != "-") {
			std::vector<char> raw = slurp(start);
			node.last_frame_particles.resize(raw.size() / sizeof(fluid::simulation::particle));
			std::memcpy(static_cast<void *>(node.last_frame_particles.data()), raw.data(),
			            node.last_frame_particles.size() * sizeof(fluid::simulation::particle));
			node.particle_cache.emplace_back();  // "frame 0" is what the file holds
		} else {
			node_attributes::source_attr sa;
			for (int z = n / 3; z < n / 2; ++z)
				for (int y = n / 2; y < n / 2 + 3; ++y)
					for (int x = 1; x < 3; ++x) { sa.cells.push_back(x); sa.cells.push_back(y); sa.cells.push_back(z); }
			sa.velocity[0] = 30.0f;
			sa.coerce_velocity = true;
			attr.sources.push_back(sa);
		}
		const double frame_time = 1.0 / 24.0;
		dump d(outdir + "/gridnode.bin");
		const std::size_t first = node.particle_cache.size();
		for (std::size_t f = first; f < first + static_cast<std::size_t>(frames); ++f) {
			if (!evaluate(node, attr, f, frame_time)) return 3;
			d.put(("frame" + std::to_string(f - first + 1) + ".points").c_str(), node.particle_cache[f]);
		}
		// the particles the node keeps for its next evaluation: velocities and the identity the test put into cx
		std::vector<double> vel, ident;
		for (const auto &p : node.last_frame_particles) {
			for (std::size_t k = 0; k < 3; ++k) vel.push_back(p.velocity[k]);
			ident.push_back(p.cx.x);
		}
		d.put("kept.vel", vel);
		d.put("kept.cx_x", ident);
		return 0;
	}

	// ============================================================================================== mesher thread + F3
	/// mesher <outdir> <points.bin> <n> <cell_size> <r> : points.bin = double[3 k]
	int mesher(const std::string &outdir, int argc, char **argv) {
		if (argc < 4) return 2;
		std::vector<char> raw = slurp(argv[0]);
		const double *xyz = reinterpret_cast<const double *>(raw.data());
		std::vector<fluid::simulation::particle> sim_particles(raw.size() / 24);
		for (std::size_t i = 0; i < sim_particles.size(); ++i) sim_particles[i].position = vec3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);

		std::vector<vec3d> particles;
		for (const fluid::simulation::particle &p : sim_particles) particles.emplace_back(p.position);
		const std::size_t n = std::strtoull(argv[1], nullptr, 10);
		fluid::mesher mesher;
		mesher.particle_extent = 2.0;
		mesher.cell_radius = 3;
		mesher.grid_offset = vec3d(-1.0, -1.0, -1.0);
		mesher.cell_size = std::atof(argv[2]);
		mesher.resize(vec3s(n, n, n));
		fluid::mesher::mesh_t mesh = mesher.generate_mesh(particles, std::atof(argv[3]));
		mesh.generate_normals();

		std::ofstream obj(outdir + "/mesh.obj");
		mesh.save_obj(obj);
		dump d(outdir + "/mesher.bin");
		std::vector<double> pos, nrm, idx;
		for (const auto &p : mesh.positions) for (std::size_t k = 0; k < 3; ++k) pos.push_back(p[k]);
		for (const auto &p : mesh.normals) for (std::size_t k = 0; k < 3; ++k) nrm.push_back(p[k]);
		for (std::size_t i : mesh.indices) idx.push_back(static_cast<double>(i));
		d.put("positions", pos);
		d.put("normals", nrm);
		d.put("indices", idx);
		return 0;
	}

	// ============================================================================================== VoxelizerNode + obstacle
	/// voxelizer <outdir> <mesh.bin> <cell_size> <ox oy oz> <rx ry rz> : mesh.bin = u64 nv, u64 ni, double[3 nv], u64[ni]
	int voxelizer(const std::string &outdir, int argc, char **argv) {
		using fluid::voxelizer;
		if (argc < 8) return 2;
		std::vector<char> raw = slurp(argv[0]);
		std::uint64_t nv = 0, ni = 0;
		std::memcpy(&nv, raw.data(), 8);
		std::memcpy(&ni, raw.data() + 8, 8);
		const double *pos = reinterpret_cast<const double *>(raw.data() + 16);
		const std::uint64_t *tri = reinterpret_cast<const std::uint64_t *>(raw.data() + 16 + 24 * nv);
		const double cell_size = std::atof(argv[1]);
		const double ref_grid_offset[3] = {std::atof(argv[2]), std::atof(argv[3]), std::atof(argv[4])};
		const int ref_grid_size[3] = {std::atoi(argv[5]), std::atoi(argv[6]), std::atoi(argv[7])};
		const bool include_interior = true, include_surface = true;

		fluid::mesh<double, int, double, double, vec3d> vox_mesh;  // the node's mesh type: int indices
		vox_mesh.positions.resize(nv);
		for (std::size_t i = 0; i < nv; ++i) vox_mesh.positions[i] = vec3d(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
		vox_mesh.indices.resize(ni);
		for (std::size_t i = 0; i < ni; ++i) vox_mesh.indices[i] = static_cast<int>(tri[i]);

		auto [bound_min, bound_max] = voxelizer::get_bounding_box(vox_mesh.positions.begin(), vox_mesh.positions.end());
		voxelizer vox;
#ifdef LFA_HOST_SHIM
		// The Maya node's call as it stands in the reference tree (voxelizer_node.cpp:261-270) passes the reference grid's size as a
		// fifth argument, which include/fluid/voxelizer.h:44 no longer declares: it only compiles against this build.
		vec3i grid_offset = vox.resize_reposition_grid_constrained(
			bound_min, bound_max, cell_size, vec3d(ref_grid_offset[0], ref_grid_offset[1], ref_grid_offset[2]),
			vec3s(static_cast<std::size_t>(ref_grid_size[0]), static_cast<std::size_t>(ref_grid_size[1]),
			      static_cast<std::size_t>(ref_grid_size[2])));
#else
		vec3i grid_offset = vox.resize_reposition_grid_constrained(
			bound_min, bound_max, cell_size, vec3d(ref_grid_offset[0], ref_grid_offset[1], ref_grid_offset[2]));
#endif
		vox.voxelize_mesh_surface(vox_mesh);
		vox.mark_exterior();

		std::vector<vec3s> occupied_cells;
		std::vector<double> types;
		vox.voxels.for_each([&](vec3s at, voxelizer::cell_type type) {
			types.push_back(static_cast<double>(static_cast<unsigned char>(type)));
			switch (type) {
			case voxelizer::cell_type::interior:
				if (include_interior) occupied_cells.emplace_back(at);
				break;
			case voxelizer::cell_type::surface:
				if (include_surface) occupied_cells.emplace_back(at);
				break;
			default:
				break;
			}
		});
		std::vector<double> cells, cells_ref;
		for (vec3s v : occupied_cells) {
			cells.push_back(static_cast<double>(static_cast<int>(v.x)));
			cells.push_back(static_cast<double>(static_cast<int>(v.y)));
			cells.push_back(static_cast<double>(static_cast<int>(v.z)));
		}
		for (vec3s v : occupied_cells) {
			vec3i vref = vec3i(v) + grid_offset;
			if (vref.x >= 0 && vref.x < ref_grid_size[0] && vref.y >= 0 && vref.y < ref_grid_size[1] && vref.z >= 0 &&
			    vref.z < ref_grid_size[2]) {
				cells_ref.push_back(vref.x);
				cells_ref.push_back(vref.y);
				cells_ref.push_back(vref.z);
			}
		}
		dump d(outdir + "/voxelizer.bin");
		d.put("grid_offset", std::vector<double>{double(grid_offset.x), double(grid_offset.y), double(grid_offset.z)});
		d.put("grid_size", std::vector<double>{double(vox.voxels.get_size().x), double(vox.voxels.get_size().y), double(vox.voxels.get_size().z)});
		d.put("types", types);
		d.put("cells", cells);
		d.put("cells_ref", cells_ref);

		// fluid::obstacle, the in-tree host of the voxelizer (std::size_t indices)
		fluid::obstacle::mesh_t solid;
		solid.positions = vox_mesh.positions;
		solid.indices.assign(vox_mesh.indices.begin(), vox_mesh.indices.end());
		fluid::obstacle obs(std::move(solid), cell_size, vec3d(ref_grid_offset[0], ref_grid_offset[1], ref_grid_offset[2]),
		                    vec3s(static_cast<std::size_t>(ref_grid_size[0]), static_cast<std::size_t>(ref_grid_size[1]),
		                          static_cast<std::size_t>(ref_grid_size[2])));
		std::vector<double> oc;
		for (vec3s c : obs.cells) for (std::size_t k = 0; k < 3; ++k) oc.push_back(static_cast<double>(c[k]));
		d.put("obstacle_cells", oc);
		return 0;
	}

	// ============================================================================================== F4
	/// points <outdir> <points.bin>
	int points(const std::string &outdir, int argc, char **argv) {
		if (argc < 1) return 2;
		std::vector<char> raw = slurp(argv[0]);
		const double *xyz = reinterpret_cast<const double *>(raw.data());
		std::vector<fluid::simulation::particle> sim_particles(raw.size() / 24);
		for (std::size_t i = 0; i < sim_particles.size(); ++i) sim_particles[i].position = vec3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
		std::vector<vec3d> pts;
		for (const fluid::simulation::particle &p : sim_particles) pts.emplace_back(p.position);
		std::ofstream fout(outdir + "/points.txt");
		fluid::point_cloud::save_to_naive(fout, pts.begin(), pts.end());
		return 0;
	}
}

int main(int argc, char **argv) {
	if (argc < 3) {
		std::fprintf(stderr, "usage: %s testbed|gridnode|mesher|voxelizer|points <outdir> [...]\n", argv[0]);
		return 2;
	}
	const std::string what = argv[1], outdir = argv[2];
	if (what == "testbed") return testbed(outdir, argc - 3, argv + 3);
	if (what == "gridnode") return gridnode(outdir, argc - 3, argv + 3);
	if (what == "mesher") return mesher(outdir, argc - 3, argv + 3);
	if (what == "voxelizer") return voxelizer(outdir, argc - 3, argv + 3);
	if (what == "points") return points(outdir, argc - 3, argv + 3);
	return 2;
}
