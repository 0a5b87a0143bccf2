"""Merges the FETCH_SIZE / WRITE_SIZE summaries (tools/pmc_summary.py csv) into profiles/*_pmc_traffic.json.

usage: pmc_traffic.py fetch.csv write.csv out.json "workload text"
FETCH_SIZE is doubled: gfx950 tallies the 128-B read requests of wide coalesced streams at 64 B
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM); WRITE_SIZE is taken as reported.
"""
import csv
import json
import sys

NAMES = {"k_pcg_a<float, false, true>": "pcg_a", "k_pcg_b<float, true>": "pcg_b",
         "k_pcg_a<float, false, false>": "pcg_a", "k_pcg_b<float, false>": "pcg_b",
         "k_spmv<float>": "spmv_dot", "k_axpy_max<float>": "axpy_max", "k_mic_apply<float, 0, true>": "mic_apply_dot",
         "k_update_s<float>": "update_s", "k_p2g_binned<true>": "p2g_scatter", "k_p2g_binned<false>": "p2g_scatter", "k_p2g_binned<true, false>": "p2g_scatter",
         "k_p2g_binned<false, false>": "p2g_scatter", "k_p2g_binned<true, true>": "p2g_scatter", "k_correct_fine": "correct_tiled",
         "k_build_fine_index": "correct_cell_index", "k_advect_collide<false>": "advect_collide", "k_advect_collide<true>": "advect_collide",
         "k_tile_scatter": "bin_scatter", "k_tile_scatter<1>": "bin_scatter", "k_tile_scatter<2>": "bin_scatter", "k_tile_scatter<0>": "bin_scatter",
         "k_mg_residual_restrict<float>": "mg_down0",
         "k_p2g_finalize<true>": "p2g_finalize", "k_p2g_finalize<false>": "p2g_finalize", "k_g2p<2>": "g2p",
         "k_g2p<1>": "g2p", "k_g2p<0>": "g2p", "k_g2p<2, true>": "g2p", "k_g2p<1, true>": "g2p", "k_g2p<0, true>": "g2p",
         "k_tile_scatter": "bin_scatter", "k_tile_scatter<true>": "bin_scatter", "k_tile_scatter<false>": "bin_scatter", "k_gather_vc": "bin_deferred_gather", "k_tile_count": "bin_count", "k_cell_count": "bin_cells",
         "k_mg_axpy_presmooth<float>": "mg_axpy_presmooth", "k_mg_axpy_presmooth<float, 4>": "mg_axpy_presmooth", "k_mg_axpy_presmooth<float, 1>": "mg_axpy_presmooth",
         "k_mg_prolong_postsmooth<float, true, 1>": "mg_up0", "k_mg_coarse<float>": "mg_coarse", "k_mg_coarse<float, MemAgent>": "mg_coarse", "k_mg_prolong_postsmooth<float, true, 1, false>": "mg_up0", "k_pcg_a<float, false, false>": "pcg_a", "k_advect_collide_count<false>": "advect_collide", "k_advect_collide_count<true>": "advect_collide", "k_mg_prolong_postsmooth<float, true>": "mg_up0", "k_mg_tail<float>": "mg_tail"}


# template arguments come and go with the code: the part before '<' decides where an exact name is not listed
BASE = {"k_p2g_binned": "p2g_scatter", "k_p2g_finalize": "p2g_finalize", "k_g2p": "g2p", "k_correct_fine": "correct_tiled",
        "k_build_fine_index": "correct_cell_index", "k_tile_scatter": "bin_scatter", "k_cell_count": "bin_cells",
        "k_advect_collide_count": "advect_collide", "k_advect_collide": "advect_collide", "k_pcg_a": "pcg_a",
        "k_mg_axpy_presmooth": "mg_axpy_presmooth", "k_mg_residual_restrict": "mg_down0", "k_mg_coarse": "mg_coarse",
        "k_mg_prolong_postsmooth": "mg_up0", "k_apply_pressure": "apply_pressure", "k_extrapolate": "extrapolate", "k_rhs": "rhs"}


def load(path, counter):
    out = {}
    for row in csv.DictReader(open(path)):
        if row["counter"] != counter:
            continue
        name = row["kernel"]
        if name == "k_correct_fine<12288, true>":  # (the second pass over crowded parts: a handful of workgroups)
            continue
        key = NAMES.get(name) or BASE.get(name.split("<")[0])
        if key:
            out[key] = max(out.get(key, 0.0), float(row["median_value_per_dispatch"]) * 1024.0)
    return out


def source_hashes():
    """sha256 of every kernel source / header of the library: bench.py refuses a traffic figure whose kernel's source file (or a
    shared header) has changed since these passes were taken (`roofline.traffic` = null, `traffic_age` says why)."""
    import hashlib
    import os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "libfluid_amd", "csrc")
    return {f: hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest()
            for f in sorted(os.listdir(root)) if f.endswith((".hip", ".h"))}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {"workload": sys.argv[4],
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of the bench command "
                 "(--no-cpu-baseline --no-hot-path --no-kernel-timing); FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md "
                 "section HBM)",
       "statistic": "median over the dispatches of a kernel (the first binning after seeding is not steady state)",
       "source_sha256": source_hashes(),
       "hbm_bytes_per_launch": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = 2.0 * fetch.get(k, 0.0), write.get(k, 0.0)
    res["hbm_bytes_per_launch"][k] = {"fetch": f, "write": w, "total": f + w}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps(res["hbm_bytes_per_launch"], indent=1))
