// <fluid/data_structures/mesh.h> without the reference's headers: see shim_standalone/fluid/math/vec.h.
#pragma once
#include "../math/vec.h"
#include "../../../mesh.h"
namespace fluid {
	using ::fluid_amd::mesh;
}
