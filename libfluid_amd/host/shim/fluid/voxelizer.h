// <fluid/voxelizer.h> replacement (see shim/fluid/simulation.h): `fluid::voxelizer` = the device-backed voxelizer
// (include/fluid/voxelizer.h:14-74).
#pragma once
#define LFA_HOST_SHIM 1
#include "../../voxelizer.h"
namespace fluid {
	using voxelizer = ::fluid_amd::voxelizer;
}
