// <fluid/mesher.h> replacement (see shim/fluid/simulation.h): `fluid::mesher` = the device-backed mesher
// (include/fluid/mesher.h:14-46); `mesh_t` is the reference's own fluid::mesh.
#pragma once
#define LFA_HOST_SHIM 1
#include "../../mesher.h"
namespace fluid {
	using mesher = ::fluid_amd::mesher;
}
