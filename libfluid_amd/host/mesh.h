// libfluid_amd/host/mesh.h -- triangle mesh container with the public surface of lukedan/libfluid's `fluid::mesh`
// (include/fluid/data_structures/mesh.h:14-100): positions / normals / indices / colors / uvs, clear(),
// reverse_face_directions(), generate_normals(), save_obj(). Pure host code (on-disk formats);
// the .OBJ text is byte-identical to the reference's for the same stream state (tests/test_formats.py).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <ostream>
#include <vector>

#include "types.h"

#ifndef LFA_HOST_REFERENCE_TYPES  // with the reference's headers on the include path fluid_amd::mesh IS fluid::mesh (types.h)
namespace fluid_amd {
	template <typename PosT, typename IndexT, typename NormalT, typename UvT, typename ColorT>
	struct mesh {
		std::vector<vec3<PosT>> positions;
		std::vector<vec3<NormalT>> normals;
		std::vector<IndexT> indices;
		std::vector<ColorT> colors;
		std::vector<vec2<UvT>> uvs;

		void clear() {
			positions.clear();
			normals.clear();
			indices.clear();
			colors.clear();
			uvs.clear();
		}

		/// Flips every triangle by exchanging its last two corners (mesh.h:30-35).
		void reverse_face_directions() {
			for (std::size_t t = 0; t + 2 < indices.size(); t += 3) std::swap(indices[t + 1], indices[t + 2]);
		}

		/// Area-weighted vertex normals (mesh.h:37-54): face cross products summed per corner, then normalised; a vertex
		/// whose sum is not longer than 1e-6 gets (1, 0, 0) (normalized_checked, include/fluid/math/vec.h:377-399).
		void generate_normals() {
			normals.assign(positions.size(), vec3<NormalT>());
			for (std::size_t t = 0; t + 2 < indices.size(); t += 3) {
				const IndexT a = indices[t], b = indices[t + 1], c = indices[t + 2];
				const vec3<PosT> e1 = positions[b] - positions[a], e2 = positions[c] - positions[a];
				const vec3<NormalT> n(static_cast<NormalT>(e1.y * e2.z - e1.z * e2.y), static_cast<NormalT>(e1.z * e2.x - e1.x * e2.z),
				                      static_cast<NormalT>(e1.x * e2.y - e1.y * e2.x));
				normals[a] += n;
				normals[b] += n;
				normals[c] += n;
			}
			for (vec3<NormalT> &n : normals) {
				const NormalT sq = n.squared_length();
				if (sq <= static_cast<NormalT>(1e-6) * static_cast<NormalT>(1e-6)) n = vec3<NormalT>(1, 0, 0);
				else n = n / std::sqrt(sq);
			}
		}

		/// Wavefront .OBJ (mesh.h:56-99): `v`, optional `vn` and `vt` records, then 1-based `f` records whose corner form
		/// (`i`, `i/i`, `i//i`, `i/i/i`) depends on which attributes exist.
		void save_obj(std::ostream &out) const {
			for (const auto &p : positions) out << "v " << p.x << " " << p.y << " " << p.z << "\n";
			for (const auto &n : normals) out << "vn " << n.x << " " << n.y << " " << n.z << "\n";
			for (const auto &t : uvs) out << "vt " << t.x << " " << t.y << "\n";
			const bool has_n = !normals.empty(), has_t = !uvs.empty();
			for (std::size_t t = 0; t + 2 < indices.size(); t += 3) {
				out << "f";
				for (std::size_t k = 0; k < 3; ++k) {
					const IndexT id = indices[t + k] + 1;
					out << " " << id;
					if (has_n) {
						out << "/";
						if (has_t) out << id;
						out << "/" << id;
					} else if (has_t) {
						out << "/" << id;
					}
				}
				out << "\n";
			}
		}
	};
}
#endif
