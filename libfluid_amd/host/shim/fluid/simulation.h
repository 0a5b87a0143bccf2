// <fluid/simulation.h> for a host of lukedan/libfluid that wants the MI355X path: put `-I <repo>/libfluid_amd/host/shim` BEFORE
// the reference's own include directory and link libfluid_amd.so; the host's sources stay as they are. `fluid::simulation` is
// then the device-backed class (include/fluid/simulation.h:21-280 is what it mirrors, member by member); vec3d, grid3,
// mac_grid, source remain the reference's own types (libfluid_amd/host/types.h). INTEGRATION.md, section A.
#pragma once
#define LFA_HOST_SHIM 1
#include "../../simulation.h"
namespace fluid {
	using simulation = ::fluid_amd::simulation;
}
