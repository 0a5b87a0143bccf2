// libfluid_amd/host/point_cloud.h -- the reference's plain-text point cloud format
// (include/fluid/data_structures/point_cloud.h:13-36, src/data_structures/point_cloud.cpp:8-19): one "x y z" line per
// point, written with the stream's current formatting, read until extraction fails or `count` points were read.
// Pure host code; testbed F4 dumps particle positions with it (testbed/main.cpp:335-347).
#pragma once

#include <cstddef>
#include <istream>
#include <limits>
#include <ostream>
#include <vector>

#include "types.h"

namespace fluid_amd {
	namespace point_cloud {
		template <typename It> inline void save_to_naive(std::ostream &out, It begin, It end) {
			for (It p = begin; p != end; ++p) out << p->x << " " << p->y << " " << p->z << "\n";
		}
		template <typename Callback> inline void load_from_naive(
			std::istream &in, Callback &&cb, std::size_t count = std::numeric_limits<std::size_t>::max()
		) {
			vec3d v;
			for (std::size_t k = 0; k < count && (in >> v.x >> v.y >> v.z); ++k) cb(v);
		}
		inline std::vector<vec3d> load_from_naive(std::istream &in, std::size_t count = std::numeric_limits<std::size_t>::max()) {
			std::vector<vec3d> pts;
			load_from_naive(in, [&pts](vec3d v) { pts.push_back(v); }, count);
			return pts;
		}
	}
}
