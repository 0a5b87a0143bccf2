"""Generates tests/golden/mesher.npz from the REAL reference mesher (oracle/_ref/libref.so: src/mesher.cpp compiled in
place; private members reached like the solver's). Build container only. Data only: inputs and what the reference computed
(sampled surface function, vertex positions, index lists)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import loader as orc  # noqa: E402
from tests import mesher_cases as mc  # noqa: E402

if __name__ == "__main__":
    orc.build()
    if not orc.have_ref():
        sys.exit("oracle/_ref/libref.so is not built")
    out = {}
    # all 256 cases of one cell, unit magnitudes (edge midpoints) and fixed random magnitudes (interpolation)
    mags = np.random.default_rng(11).uniform(0.2, 3.0, size=(256, 8))
    pos_all, idx_all, cnt = [], [], []
    for variant in (None, mags):
        for case in range(256):
            v = mc.single_cell_values(case, None if variant is None else variant[case])
            pos, idx = orc.mesher_mesh(None, (1, 1, 1), values=v, kind="ref")
            pos_all.append(pos)
            idx_all.append(idx)
            cnt.append((len(pos), len(idx)))
    out.update(cases_mags=mags, cases_pos=np.concatenate(pos_all), cases_idx=np.concatenate(idx_all),
               cases_counts=np.asarray(cnt, dtype=np.int64))
    for seed, size in ((1, (7, 6, 5)), (2, (1, 9, 1)), (3, (12, 1, 3))):
        v = mc.random_field(seed, size)
        pos, idx = orc.mesher_mesh(None, size, (0.25, -1.5, 3.0), 0.7, values=v, kind="ref")
        out.update({f"field{seed}_values": v, f"field{seed}_pos": pos, f"field{seed}_idx": idx})
        print("field", seed, size, pos.shape, idx.shape)
    for name in mc.PARTICLE_CASES:
        p, kw = mc.particle_case(name)
        vals = orc.mesher_surface(p, kind="ref", **kw)
        pos, idx = orc.mesher_mesh(p, kind="ref", **kw)
        out.update({f"{name}_particles": p, f"{name}_values": vals, f"{name}_pos": pos, f"{name}_idx": idx})
        print(name, vals.shape, int(np.isnan(vals).sum()), "NaN", pos.shape, idx.shape)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mesher.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")
