// <fluid/data_structures/obstacle.h> replacement (see shim/fluid/simulation.h): `fluid::obstacle`
// (include/fluid/data_structures/obstacle.h:11-21) voxelizes its mesh on the device.
#pragma once
#define LFA_HOST_SHIM 1
#include "../../../voxelizer.h"
namespace fluid {
	using obstacle = ::fluid_amd::obstacle;
}
