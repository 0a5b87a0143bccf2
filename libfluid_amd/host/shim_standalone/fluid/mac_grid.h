// <fluid/mac_grid.h> without the reference's headers: see shim_standalone/fluid/math/vec.h.
#pragma once
#include "math/vec.h"
