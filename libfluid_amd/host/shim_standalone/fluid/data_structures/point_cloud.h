// <fluid/data_structures/point_cloud.h> without the reference's headers: see shim_standalone/fluid/math/vec.h.
#pragma once
#include "../math/vec.h"
#include "../../../point_cloud.h"
namespace fluid {
	namespace point_cloud = ::fluid_amd::point_cloud;
}
