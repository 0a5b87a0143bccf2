// <fluid/math/vec.h> for a build WITHOUT the reference's headers (add `-I <repo>/libfluid_amd/host/shim_standalone` after
// `-I <repo>/libfluid_amd/host/shim`): the self-contained value types of libfluid_amd/host/types.h under their `fluid::` names.
// Never put this directory on the include path together with the reference's own headers.
#pragma once
#define LFA_HOST_OWN_TYPES 1
#include "../../../types.h"
namespace fluid {
	namespace vec_ops = ::fluid_amd::vec_ops;
	using ::fluid_amd::vec2;
	using ::fluid_amd::vec3;
	using ::fluid_amd::vec2d;
	using ::fluid_amd::vec2s;
	using ::fluid_amd::vec3d;
	using ::fluid_amd::vec3f;
	using ::fluid_amd::vec3i;
	using ::fluid_amd::vec3s;
	using ::fluid_amd::grid3;
	using ::fluid_amd::mac_grid;
	using ::fluid_amd::source;
}
