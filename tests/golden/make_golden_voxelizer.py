"""Generates tests/golden/voxelizer.npz from the REAL reference voxelizer (oracle/_ref/libref.so: src/voxelizer.cpp,
src/math/intersection.cpp, src/data_structures/obstacle.cpp compiled in place). Build container only.
A fixture is data: mesh inputs and what the reference computed (voxel types, grid placement, the VoxelizerNode's lists)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import loader as orc  # noqa: E402
from tests import voxel_cases  # noqa: E402

if __name__ == "__main__":
    orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so is not built: /root/reference is needed to generate the golden vectors")
    out = {}
    for name in voxel_cases.NAMES:
        pos, idx, cs, off, rs = voxel_cases.make(name)
        gmin, goff, types = orc.voxelize(pos, idx, cs, off, kind="ref")
        cells = orc.ref_voxel_cells(pos, idx, cs, off, True, False, rs)      # cells_ref, interior only
        cells_all = orc.ref_voxel_cells(pos, idx, cs, off, True, True, None)  # cells, interior + surface
        out.update({f"{name}_pos": pos, f"{name}_idx": idx, f"{name}_cs": np.float64(cs), f"{name}_off": np.asarray(off),
                    f"{name}_ref_size": np.asarray(rs, dtype=np.int64), f"{name}_grid_min": gmin, f"{name}_grid_off": goff,
                    f"{name}_types": types, f"{name}_cells_ref_interior": cells, f"{name}_cells_all": cells_all})
        print(name, types.shape, {k: int((types == k).sum()) for k in range(3)}, len(cells), len(cells_all))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "voxelizer.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
