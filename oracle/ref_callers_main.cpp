// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// oracle/ref_callers_main.cpp : tests/callers/reference_callers.cpp built against the REAL lukedan/libfluid. A unity translation
// unit like oracle/ref_harness.cpp: it contains no reference code, it #includes the reference's sources where they lie under
// $REFERENCE_DIR and then the caller-shaped host program, so that the very file that drives this repository's device path
// through the shim headers (libfluid_amd/host/shim) also drives the reference - and tests/test_ref_callers.py can compare the two.
// Output: oracle/_ref/callers_ref (git-ignored). pcg32: see ref_harness.cpp.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <random>
#include <stack>
#include <string>
#include <vector>

#include <pcg_random.hpp>
using arrow_vendored::pcg32;

#include "src/mac_grid.cpp"
#include "src/simulation.cpp"
#include "src/pressure_solver.cpp"
#include "src/math/intersection.cpp"
#include "src/voxelizer.cpp"
#include "src/data_structures/obstacle.cpp"
#include "src/mesher.cpp"
#include "src/data_structures/point_cloud.cpp"

#include "../tests/callers/reference_callers.cpp"
