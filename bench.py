#!/usr/bin/env python3
"""bench.py -- particle-steps/s (+ PCG iterations/s) of the MI355X hot path on BASELINE.json's synthetic dam break.

A "step" is one device-resident pass of the hot path over the resident particle set (lfa_step_hot):
    tile binning -> P2G -> gravity -> pressure system + MIC(0)-PCG -> pressure gradient -> extrapolation -> G2P
(reference: src/simulation.cpp:62-66,72-78,83-104,119-121). The stages of simulation::time_step outside SURVEY.md
section 8(a) (advect / collide / position correction, 8(f) "next" rows) are not part of the timed region.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run. W untimed steps, then exactly K timed steps bracketed by barrier + device synchronise, MAX over
ranks, rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable

# algorithmic bytes per unit; n = PCG unknowns, Np particles, Nc cells of the processed tiles
PCG_BYTES_UNFUSED = {"spmv_dot": 17, "axpy_max": 28, "mic_apply_dot": 34, "update_s": 12}
# fused iteration (default): k_pcg_a = update_s + spmv_dot, k_pcg_b = axpy_max + mic_apply_dot; same 91 n in total
PCG_BYTES_FUSED = {"pcg_a": 12 + 17, "pcg_b": 28 + 34}
# multigrid iteration (default on one GPU): search direction + A s, AXPYs + finest-level pre-smoothing, finest-level residual
# + restriction, coarser levels (latency bound: no byte figure), finest-level prolongation + post-smoothing + dot.
# Bytes per unknown = the minimal traffic of each pass with SURVEY 8(d)'s conventions (fp32 vector 4, A byte 1, dot 8, max 4).
PCG_BYTES_MG = {"pcg_a": 12 + 17, "mg_axpy_presmooth": 28 + 5, "mg_down0": 9.5, "mg_coarse": 0, "mg_up0": 21}


def cpu_baseline(sample, steps):
    """The reference's own hot path (oracle/_ref/libref.so: src/simulation.cpp + src/pressure_solver.cpp + src/mac_grid.cpp
    compiled in place, kind "reference") timed on this box's host cores; when that build did not travel, the plain-C
    restatement (oracle/liboracle.so, kind "port")."""
    import numpy as np  # noqa: F401
    from libfluid_amd import scenes
    from oracle import loader as orc
    cfg = scenes.CONFIGS[sample]
    parts = scenes.seed_block(*cfg["block"])
    kind = "ref" if orc.have_ref() else "oracle"
    sim = orc.CpuSim(cfg["size"], method=cfg["method"], blending=cfg["blending"], kind=kind)
    sim.set_particles(parts)
    t0 = time.perf_counter()
    iters = 0
    for _ in range(steps):
        _, _, it = sim.hot_step(0.033)
        iters += it
    dt = time.perf_counter() - t0
    return {
        "value": len(parts) * steps / dt, "unit": "particle-steps/s", "cores": 1,
        "kind": "reference" if kind == "ref" else "port",
        "sample": f"{sample}: {cfg['size'][0]}^3 grid, {len(parts)} particles, {steps} hot-path steps, "
                  f"{iters} PCG iterations, {dt:.1f} s; "
                  + ("the reference's serial _transfer_to_grid / pressure_solver::solve / _transfer_from_grid "
                     "(src/simulation.cpp:293-398, src/pressure_solver.cpp:19-71), g++ -O2 -DNDEBUG" if kind == "ref" else
                     "plain-C restatement, serial like the reference's P2G/PCG/G2P, gcc -O2"),
        "pcg_iters_per_s": iters / dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default=None, help="C2 | C3 | C4 | C5 (default: C4 at 1 GPU, see DESIGN.md)")
    ap.add_argument("--dt", type=float, default=0.033, help="min(3*cfl, 0.033) of simulation::time_step() at rest")
    ap.add_argument("--precond", default=None, choices=["multigrid", "multilevel", "tiled", "exact"],
                    help="default: multigrid (on z-slabs: finest 4 levels distributed, coarser ones replicated)")
    ap.add_argument("--pcg-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--p2g", default="binned", choices=["binned", "atomic"])
    ap.add_argument("--max-iterations", type=int, default=200, help="PCG iteration cap (pressure_solver.h:42)")
    ap.add_argument("--cpu-sample", default="C2")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-full-step", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="one launch per vector operation in the PCG loop (pcg_fused = 0)")
    ap.add_argument("--obstacle", action="store_true", help="BASELINE configs[4]: voxelize a sphere mesh on the device "
                    "(lfa_voxelize_mesh) and mark it solid before the steps")
    ap.add_argument("--mesh", action="store_true", help="BASELINE configs[4]: extract the surface mesh from the resident particles "
                    "after the timed steps (lfa_mesher_sample_sim + marching cubes), timed separately")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent copies of the domain instead of z-slabs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    if args.precond is None:
        args.precond = "multigrid"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import libfluid_amd as lfa
    from libfluid_amd import scenes

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    cfg_name = args.config or "C4"
    cfg = dict(scenes.CONFIGS[cfg_name])
    size, (blo, bhi) = list(cfg["size"]), [list(x) for x in cfg["block"]]
    parallelism = "1 GPU"
    if world > 1 and not args.replicas:
        # weak scaling: the domain and the dam-break block grow along z with the number of GPUs, every rank owns a slab
        # as large as the single-GPU workload (BASELINE configs[3]/[4] decompose along z the same way)
        size[2] *= world
        bhi[2] *= world
    sim = lfa.Sim(size, method=cfg["method"], blending=cfg["blending"], device=local_rank,
                  precond={"exact": lfa.PRECOND_MIC0_EXACT, "tiled": lfa.PRECOND_MIC0_TILED,
                           "multilevel": lfa.PRECOND_MULTILEVEL, "multigrid": lfa.PRECOND_MULTIGRID}[args.precond],
                  pcg_dtype=lfa.PCG_F64 if args.pcg_dtype == "f64" else lfa.PCG_F32,
                  p2g_variant=lfa.P2G_GLOBAL_ATOMIC if args.p2g == "atomic" else lfa.P2G_LDS_BINNED,
                  max_iterations=args.max_iterations, pcg_fused=0 if args.unfused else 1)
    fused = not args.unfused and args.precond != "exact"
    PCG_BYTES = PCG_BYTES_MG if args.precond == "multigrid" else (PCG_BYTES_FUSED if fused else PCG_BYTES_UNFUSED)
    if world > 1 and not args.replicas:
        # one RCCL communicator per handle: rank 0 creates the id, torch.distributed (RCCL) broadcasts it
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(lfa.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        ntz = (size[2] + 7) // 8
        bounds = lfa.balanced_layer_bounds(ntz, world, blo[2] // 8, (bhi[2] + 7) // 8)
        sim.init_rccl_slab(rank, world, uid.cpu().numpy().tobytes(), bounds)
        parallelism = f"{world} z-slabs (tile layers {bounds}), RCCL send/recv halos + scalar all-reduces over xGMI"
    elif world > 1:
        parallelism = f"{world} independent replicas (--replicas)"
    extras = {}
    if args.obstacle:
        # a sphere in the dry part of the tank, in the path of the collapsing column (the classic dam break with an obstacle),
        # voxelized on the device and marked solid without leaving it. It must not overlap the seeded block: particles deep
        # inside a solid give rows without a diagonal, which the reference's MIC(0) turns into 1/sqrt(0) as well.
        rad = 0.16 * min(bhi[0] - blo[0], bhi[1] - blo[1], cfg["block"][1][2] - blo[2])
        ctr = [min(bhi[0] + 2.0 * rad, size[0] - 1.5 * rad), blo[1] + 1.2 * rad, 0.5 * (blo[2] + cfg["block"][1][2])]
        mpos, midx = scenes.icosphere(ctr, rad, 5)
        t0 = time.perf_counter()
        vox = lfa.Voxels.from_mesh(mpos, midx, 1.0, (0.0, 0.0, 0.0), device=local_rank)
        sim.set_solid_from_voxels(vox, True, True)
        extras["voxelizer"] = {"triangles": int(len(midx) // 3), "voxel_grid": list(vox.size), "ms_mesh_to_solid_cells": 1e3 * (time.perf_counter() - t0),
                               "solid_cells": int(len(vox.cells(True, True, size)))}
        vox.close()
    sim.seed_block(blo, bhi)
    sim.enable_timing(True)

    def barrier():
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    iters_total, not_converged = 0, 0
    for _ in range(args.warmup):
        sim.step_hot(args.dt)
    barrier()
    t0 = time.perf_counter()
    stage_ms = {}
    for _ in range(args.steps):
        res, it, rc = sim.step_hot(args.dt)
        iters_total += it
        not_converged += int(rc == lfa.W_PCG_NOT_CONVERGED)
        for k, v in sim.timings().items():
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counts = sim.counts()
    npart, n_unknowns = counts["particles"], counts["unknowns"]
    ncell_proc = counts["processed_tiles"] * 512
    npart_total = npart
    if dist is not None:
        t = torch.tensor([float(npart)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        npart_total = int(t.item())
    value = npart_total * args.steps / elapsed
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    pcg_s = stage_ms["pcg_loop"] * 1e-3 * args.steps

    out = {
        "metric": "particle_steps_per_sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.pcg_dtype == "f32" else "f32 particles/grid, f64 PCG vectors", "data": "synthetic",
        "config": {
            "workload": f"{cfg_name}: {size[0]}x{size[1]}x{size[2]} MAC grid, dam-break block "
                        f"{tuple(blo)}-{tuple(bhi)} cells, {npart_total} particles, "
                        f"{['PIC', 'FLIP', 'APIC'][cfg['method']]} blend {cfg['blending']}, dt {args.dt}, "
                        f"hot path only (bin+P2G+gravity+PCG+apply+extrapolate+G2P)",
            "unknowns": n_unknowns, "particles_per_gpu": npart,
            "precond": {"tiled": "MIC(0) per 8^3 tile", "exact": "MIC(0) exact (tile hyperplanes)",
                        "multilevel": "MIC(0) per 8^3 tile + tile-aggregate coarse correction",
                        "multigrid": "geometric multigrid V(1,1), red-black Gauss-Seidel per tile"}[args.precond],
            "pcg_loop": ("k_pcg_a + AXPYs/pre-smoothing + V-cycle (k_mg_*) per iteration" if args.precond == "multigrid" else
                         "fused: 2 launches per iteration (k_pcg_a, k_pcg_b)" if fused else "5 launches per iteration"),
            "p2g": args.p2g, "pcg_tolerance": 1e-6, "pcg_max_iterations": args.max_iterations,
            "parallelism": parallelism,
        },
        "pcg": {
            "iterations_per_step": iters_total / max(args.steps, 1),
            "iters_per_sec": iters_total / pcg_s if pcg_s > 0 else None,
            "unknown_iters_per_sec": n_unknowns * iters_total / pcg_s if pcg_s > 0 else None,
            "steps_hitting_max_iterations": not_converged,
        },
        "stage_ms": stage_ms,
    }

    if rank == 0 and world == 1 and not args.no_kernel_timing:
        # live HIP-event timing of each hot kernel on the handle's stream (mean of 20 launches)
        apic = cfg["method"] == 2
        kernels = {}
        for name in PCG_BYTES:
            # pcg_b / mic_apply_dot = the launch of the PCG loop: tile sweeps + the embedded coarse-level workgroups
            ms = sim.bench_kernel(name, 20)
            b = int(PCG_BYTES[name] * n_unknowns * (2 if args.pcg_dtype == "f64" else 1))
            kernels[name] = {"ms": ms, "algorithmic_bytes": b, "GBps": b / ms * 1e-6}
        if args.precond == "multilevel" and not fused:
            kernels["mic_sweeps_without_coarse_levels"] = {"ms": sim.bench_kernel("mic_fine", 20), "algorithmic_bytes": 0,
                                                           "GBps": 0.0}
            kernels["coarse_levels_as_own_launch"] = {"ms": sim.bench_kernel("coarse_levels", 20), "algorithmic_bytes": 0,
                                                      "GBps": 0.0}
        p2g_bytes = (60 if apic else 24) * npart
        g2p_bytes = (60 if apic else (36 if cfg["method"] == 1 else 24)) * npart + \
            (24 if cfg["method"] == 1 else 12) * ncell_proc
        # P2G = 60 Np + 14 Nc with Nc = ALL cells (the reference writes every cell, src/simulation.cpp:
        # 296-335); cells outside the processed tiles are implicit here (background value), a legitimate saving
        ncell_all = cfg["size"][0] * cfg["size"][1] * cfg["size"][2]  # per GPU
        fin_bytes = (26 if cfg["method"] == 1 else 14) * ncell_all
        bin_bytes = (2 * 68 + 8) * npart  # 2 x state bytes + keys
        for name, b in (("g2p", g2p_bytes), ("p2g_finalize", fin_bytes), ("p2g_scatter", p2g_bytes), ("bin", bin_bytes)):
            ms = sim.bench_kernel(name, 10)
            kernels[name] = {"ms": ms, "algorithmic_bytes": b, "GBps": b / ms * 1e-6}
        out["kernels"] = kernels
        # dominant kernel = largest share of the step: iterations x per-iteration kernel time vs the one-shot kernels
        it_per_step = iters_total / max(args.steps, 1)
        share = {k: v["ms"] * (it_per_step if k in PCG_BYTES else 1.0) for k, v in kernels.items()
                 if v["algorithmic_bytes"]}
        dom = max(share, key=share.get)
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (same command, same workload)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_c4_pmc_traffic.json")
        if cfg_name == "C4" and args.precond in ("multilevel", "multigrid") and args.pcg_dtype == "f32" and os.path.exists(pmc):
            per_launch = json.load(open(pmc))["hbm_bytes_per_launch"]
            if dom == "bin":  # the binning stage = count + scatter + per-cell histogram launches
                parts = [per_launch.get(k, {}).get("total") for k in ("bin_count", "bin_scatter", "bin_cells")]
                traffic = sum(parts) if all(v is not None for v in parts) else None
            else:
                traffic = per_launch.get(dom, {}).get("total")
        out["roofline"] = {
            "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes": kernels[dom]["algorithmic_bytes"], "ms": kernels[dom]["ms"],
            "share_of_step_ms": share[dom],
        }
        pcg_iter_ms = sum(kernels[k]["ms"] for k in PCG_BYTES)
        # the reference's iteration is 91 n bytes; the V-cycle iteration is priced by its own passes
        pcg_bytes = (sum(PCG_BYTES.values()) if args.precond == "multigrid" else 91) * n_unknowns * (2 if args.pcg_dtype == "f64" else 1)
        out["roofline_groups"] = {
            "pcg_iteration": {"ms": pcg_iter_ms, "algorithmic_bytes": pcg_bytes, "GBps": pcg_bytes / pcg_iter_ms * 1e-6,
                              "frac": pcg_bytes / pcg_iter_ms * 1e-6 / HBM_PEAK_GBS},
            "p2g": {"ms": kernels["p2g_scatter"]["ms"] + kernels["p2g_finalize"]["ms"],
                    "algorithmic_bytes": p2g_bytes + fin_bytes,
                    "GBps": (p2g_bytes + fin_bytes) / (kernels["p2g_scatter"]["ms"] + kernels["p2g_finalize"]["ms"]) * 1e-6,
                    "frac": (p2g_bytes + fin_bytes) / (kernels["p2g_scatter"]["ms"] + kernels["p2g_finalize"]["ms"])
                    * 1e-6 / HBM_PEAK_GBS},
        }
    if world == 1 and not args.no_full_step:
        # beyond the headline: the device-resident simulation::time_step(dt) (hot path + advect/collide/correct, SURVEY 8f
        # rank 1), dt = min(cfl_number * cfl, 0.033) like simulation::time_step() (src/simulation.cpp:127-129)
        fs_ms, fs_iters, n_fs = 0.0, 0, 3
        t1 = time.perf_counter()
        for _ in range(n_fs):
            dt_fs = min(3.0 * sim.cfl(), 0.033)
            _, it, _ = sim.time_step(dt_fs)
            fs_iters += it
        sim.synchronize()
        fs_s = time.perf_counter() - t1
        out["full_time_step"] = {"steps": n_fs, "ms_per_step": 1e3 * fs_s / n_fs, "particle_steps_per_sec": npart * n_fs / fs_s,
                                 "pcg_iterations_per_step": fs_iters / n_fs, "stage_ms": sim.step_timings(),
                                 "note": "device resident: advect+collide, bin, P2G, PCG, apply, correct+collide, "
                                         "extrapolate, G2P (P2G-time order; particles that left their tile: gather kernel)"}
    if args.mesh and world == 1:
        m = lfa.Mesher(size, (0.0, 0.0, 0.0), 1.0, 1.0, 2, device=local_rank)  # mesher settings of testbed/main.cpp:101-107 at cell size 1
        t0 = time.perf_counter()
        m.sample_sim(sim, 0.5)
        t1 = time.perf_counter()
        mpos, midx = m.marching_cubes()
        t2 = time.perf_counter()
        extras["mesher"] = {"grid_points": (size[0] + 1) * (size[1] + 1) * (size[2] + 1), "sample_ms": 1e3 * (t1 - t0),
                            "marching_cubes_ms_incl_download": 1e3 * (t2 - t1), "vertices": int(len(mpos)), "triangles": int(len(midx) // 3)}
        m.close()
    out.update(extras)
    sim.close()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_steps)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
