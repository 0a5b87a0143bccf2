#!/usr/bin/env python3
"""bench.py -- particle-steps/s (+ PCG iterations/s) of the MI355X-native libfluid step on BASELINE.json's synthetic dam break.

A "step" is one device-resident simulation::time_step(dt) (reference: src/simulation.cpp:43-125; SURVEY.md 8(d) metric (1)):
    dt = min(cfl_number * cfl(), 0.033)   (simulation::time_step(), :127-129 -- the max-|v| reduction is inside the timed step)
    advect+collide -> binning -> P2G -> gravity -> pressure system + PCG -> pressure gradient -> position correction+collide
    -> extrapolation -> G2P
over the particles resident in HBM. The timed region starts on a dam that is already breaking: at least 20 untimed steps
(`--warmup`, topped up by a pre-roll when the caller asks for fewer) precede it, so that particles cross cells and tiles, the
binning permutes, the deferred v/C gather fires and the multigrid tile lists are rebuilt as in a production run. The hot path
alone (lfa_step_hot: the rows of SURVEY 8(a)) is reported beside it under "hot_path".

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches one rank per GPU with
torch.distributed.run. W untimed steps, then exactly K timed steps bracketed by barrier + device synchronise, MAX over
ranks, rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); the measured ceiling is reported beside it
MIN_LEAD_IN = 20       # untimed steps before the timed region (warm-up + pre-roll)

# algorithmic bytes per unit; n = PCG unknowns, Np particles, Nc cells
PCG_BYTES_UNFUSED = {"spmv_dot": 17, "axpy_max": 28, "mic_apply_dot": 34, "update_s": 12}
# fused iteration: k_pcg_a = update_s + spmv_dot, k_pcg_b = axpy_max + mic_apply_dot; same 91 n in total
PCG_BYTES_FUSED = {"pcg_a": 12 + 17, "pcg_b": 28 + 34}
# multigrid iteration (default): search direction + A s, AXPYs + finest-level pre-smoothing, finest-level residual
# + restriction, coarser levels (latency bound: no byte figure), finest-level prolongation + post-smoothing + dot.
PCG_BYTES_MG = {"pcg_a": 12 + 17, "mg_axpy_presmooth": 28 + 5, "mg_down0": 9.5, "mg_coarse": 0, "mg_up0": 21}


def cpu_baseline(sample, steps, lfa, skip_full=False):
    """The reference's own hot path (oracle/_ref/libref.so: src/simulation.cpp + src/pressure_solver.cpp + src/mac_grid.cpp
    compiled in place, kind "reference") timed on this box's host cores; when that build did not travel, the plain-C
    restatement (oracle/liboracle.so, kind "port"). The pressures it computes are compared with the device's on the same
    particles (the oracle is the checker here, never the thing measured on the GPU side)."""
    import numpy as np
    from libfluid_amd import scenes
    from oracle import loader as orc
    cfg = scenes.CONFIGS[sample]
    parts = scenes.seed_block(*cfg["block"])
    kind = "ref" if orc.have_ref() else "oracle"
    sim = orc.CpuSim(cfg["size"], method=cfg["method"], blending=cfg["blending"], kind=kind)
    sim.set_particles(parts)
    t0 = time.perf_counter()
    iters, its, p_first, first_cells, first_vel = 0, [], None, None, None
    for k in range(steps):
        p, _, it = sim.hot_step(0.033)
        if k == 0:
            p_first = p.copy()
            t_skip = time.perf_counter()  # (the copies below are the checker's, not the reference's work)
            first_cells = sim.cells()["vel"].copy()
            w = sim.particles()
            first_vel = w["vel"][np.lexsort((w["pos"][:, 2], w["pos"][:, 1], w["pos"][:, 0]))].copy()
            del w
            t0 += time.perf_counter() - t_skip
        iters += it
        its.append(int(it))
    dt = time.perf_counter() - t0
    # The headline metric is the FULL simulation::time_step (SURVEY 8(d) metric (1)): the reference's own, advection, collisions
    # and position correction included (its OpenMP regions, src/simulation.cpp:226-249,562-683, use the threads named below;
    # P2G / PCG / G2P are serial), on the state the hot-path steps above have produced, dt = min(3 cfl, 0.033).
    full = None
    if sim.L.time_step is not None and not skip_full:
        import ctypes as C
        n_full, its_full = max(1, steps), []
        tf = time.perf_counter()
        for _ in range(n_full):
            it = C.c_uint64(0)
            sim.L.time_step(sim.h, min(3.0 * sim.cfl(), 0.033), None, C.byref(it))
            its_full.append(int(it.value))
        df = time.perf_counter() - tf
        full = {"value": len(parts) * n_full / df, "unit": "particle-steps/s", "steps": n_full, "seconds": df,
                "pcg_iterations": its_full,
                "what": "simulation::time_step (src/simulation.cpp:43-125) incl. cfl(), advection, collisions, position correction"}
    cpu_model = "?"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    hot = {"value": len(parts) * steps / dt, "unit": "particle-steps/s", "cores": 1, "steps": steps, "seconds": dt,
           "what": "the serial hot path alone: _transfer_to_grid / pressure_solver::solve / _transfer_from_grid "
                   "(src/simulation.cpp:293-398, src/pressure_solver.cpp:19-71), no advection / collision / correction"}
    # `value` is the SAME metric as the GPU line's `value`: particle-steps/s of the full simulation::time_step (the hot-path-only
    # figure sits beside it under `hot_path`); when the full step was skipped (--cpu-skip-full-step) it is the hot path and says so
    out = {
        "value": full["value"] if full else hot["value"], "unit": "particle-steps/s",
        "cores": (threads if kind == "ref" else 1) if full else 1,
        "what": ("full simulation::time_step (same metric as `value` of this line), OpenMP regions on `cores` threads, "
                 "P2G / PCG / G2P serial as in the reference") if full else hot["what"],
        "hot_path": hot,
        "host": {"cpu_model": cpu_model, "nproc": os.cpu_count(), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
                 "omp_threads_in_parallel_regions": threads if kind == "ref" else 1},
        "full_time_step": full,
        "kind": "reference" if kind == "ref" else "port",
        "sample": f"{sample}: {cfg['size'][0]}^3 grid, {len(parts)} particles, {steps} hot-path steps, "
                  f"{iters} PCG iterations, {dt:.1f} s; "
                  + ("the reference's serial _transfer_to_grid / pressure_solver::solve / _transfer_from_grid "
                     "(src/simulation.cpp:293-398, src/pressure_solver.cpp:19-71), g++ -O2 -DNDEBUG" if kind == "ref" else
                     "plain-C restatement, serial like the reference's P2G/PCG/G2P, gcc -O2"),
        "pcg_iters_per_s": iters / dt, "pcg_iterations": its,
    }
    # the same first step on the device: exact MIC(0) schedule / fp64 vectors (iteration count; skipped above C3 - its 3 N - 2
    # dependent hyperplanes are a parity instrument, not a solver for 8 M unknowns) and the default configuration. Errors against
    # the fp64 CPU result: pressure max-norm relative (north star: <= 1e-4) and pointwise relative with a floor of 1e-4 of the
    # maximum (<= 1e-3), face velocities of the final grid and particle velocities relative to their maxima (<= 1e-4).
    check = {}
    tags = [("default", {})]
    if len(parts) <= 20_000_000:
        tags.insert(0, ("exact_f64", dict(precond=lfa.PRECOND_MIC0_EXACT, pcg_dtype=lfa.PCG_F64)))
    for tag, extra in tags:
        g = lfa.Sim(cfg["size"], method=cfg["method"], blending=cfg["blending"], **extra)
        g.upload_particles(parts)
        _, it, _ = g.step_hot(0.033)
        pg = g.pressure()
        pmax = float(np.abs(p_first).max())
        rec = {"iterations": int(it), "cpu_iterations": its[0], "unknowns": int(len(p_first)), "pressure_max": pmax,
               "pressure_max_rel_err": float(np.abs(pg - p_first).max() / pmax),
               "pressure_pointwise_rel_err_floor_1e-4": float((np.abs(pg - p_first) / (np.abs(p_first) + 1e-4 * pmax)).max())}
        if first_cells is not None:
            gv = g.cells()["vel"]
            rec["face_velocity_max_rel_err"] = float(np.abs(gv - first_cells).max() / np.abs(first_cells).max())
            del gv
        if first_vel is not None:
            got = g.download_particles(into=parts.copy())
            order_g = np.lexsort((got["pos"][:, 2], got["pos"][:, 1], got["pos"][:, 0]))
            rec["particle_velocity_max_rel_err"] = float(np.abs(got["vel"][order_g] - first_vel).max() / np.abs(first_vel).max())
            del got, order_g
        check[tag] = rec
        g.close()
    out["device_vs_cpu_first_step"] = check
    sim.close()
    return out


def build_sim(args, cfg, lfa, torch, dist, tdev, rank, world, local_rank, n_dev, shared_gpu, transport, slabs, strong):
    """The handle of one run: the BASELINE domain (single GPU, or `strong`: split into `world` z-slabs) or - slabs, not strong -
    the weak-scaling domain that grows along z with the number of ranks. Slab runs get their transport here."""
    size, (blo, bhi) = list(cfg["size"]), [list(x) for x in cfg["block"]]
    parallelism = "1 GPU"
    if slabs and not strong:
        # weak scaling: the domain and the dam-break block grow along z with the number of GPUs, every rank owns a slab
        # as large as the single-GPU workload
        size[2] *= world
        bhi[2] *= world
    sim = lfa.Sim(size, method=cfg["method"], blending=cfg["blending"], device=local_rank,
                  precond={"exact": lfa.PRECOND_MIC0_EXACT, "tiled": lfa.PRECOND_MIC0_TILED,
                           "multilevel": lfa.PRECOND_MULTILEVEL, "multigrid": lfa.PRECOND_MULTIGRID}[args.precond],
                  pcg_dtype=lfa.PCG_F64 if args.pcg_dtype == "f64" else lfa.PCG_F32,
                  p2g_variant=lfa.P2G_GLOBAL_ATOMIC if args.p2g == "atomic" else lfa.P2G_LDS_BINNED,
                  max_iterations=args.max_iterations, pcg_fused=0 if args.unfused else 1)
    transport_note = None
    if slabs:
        ntz = (size[2] + 7) // 8
        bounds = lfa.balanced_layer_bounds(ntz, world, blo[2] // 8, (bhi[2] + 7) // 8)
        if transport == "rccl":
            # one RCCL communicator per handle: rank 0 creates the id, torch.distributed (RCCL) broadcasts it
            uid = torch.zeros(128, dtype=torch.uint8, device=tdev)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(lfa.rccl_unique_id()), dtype=torch.uint8))
            if dist is not None:
                dist.broadcast(uid, src=0)
            ok, err = 1, ""
            try:
                sim.init_rccl_slab(rank, world, uid.cpu().numpy().tobytes(), bounds)
            except lfa.LibfluidError as e:
                ok, err = 0, str(e)
            if dist is not None:  # either every rank has its communicator or none uses it
                t = torch.tensor([ok], dtype=torch.int32, device=tdev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                ok = int(t.item())
            if not ok:
                if args.transport != "auto-shm":
                    # one GPU per rank and no communicator: a host-staged run would be a PCIe number under the headline's name.
                    # `--transport auto-shm` asks for that fallback explicitly (the line then says so in `transport`).
                    raise SystemExit(f"rank {rank}: RCCL communicator could not be created: {err or 'failed on another rank'} "
                                     "(--transport auto-shm falls back to the host-staged transport)")
                # the handle that tried keeps no half-built communicator: a fresh one takes the host-staged transport
                sim.abandon_transport()  # (a peer has no communicator: release ours without waiting for it)
                sim.close()
                sim = lfa.Sim(size, method=cfg["method"], blending=cfg["blending"], device=local_rank,
                              precond={"exact": lfa.PRECOND_MIC0_EXACT, "tiled": lfa.PRECOND_MIC0_TILED,
                                       "multilevel": lfa.PRECOND_MULTILEVEL, "multigrid": lfa.PRECOND_MULTIGRID}[args.precond],
                              pcg_dtype=lfa.PCG_F64 if args.pcg_dtype == "f64" else lfa.PCG_F32,
                              p2g_variant=lfa.P2G_GLOBAL_ATOMIC if args.p2g == "atomic" else lfa.P2G_LDS_BINNED,
                              max_iterations=args.max_iterations, pcg_fused=0 if args.unfused else 1)
                transport = "shm"
                # (the ghost-particle exchange of a slab face is the largest message: ~16 B per particle of one tile layer)
                need_mb = int(16 * (bhi[0] - blo[0]) * (bhi[1] - blo[1]) * 8 * 8 * 2 / 2**20) + 8
                transport_note = (f"RCCL communicator could not be created ({err or 'failed on another rank'}): host-staged fallback "
                                  f"(LFA_SHM_SLOT_MB >= {need_mb} carries every message of this size in one round; smaller slots take several)")
                os.environ.setdefault("LFA_SHM_SLOT_MB", str(max(32, need_mb)))
                if rank == 0:
                    print(f"bench.py: {transport_note}", file=sys.stderr)
        if transport == "shm":
            names = [f"/lfa_bench_{os.getpid()}_{int(time.time() * 1e3) & 0xffffff}_{int(strong)}"]
            if dist is not None:
                dist.broadcast_object_list(names, src=0)
            sim.init_shm_slab(names[0], rank, world, bounds)
        parallelism = (f"{world} z-slabs (tile layers {bounds}), {'strong' if strong else 'weak'} scaling, " +
                       ("RCCL send/recv halos + scalar all-reduces + particle migration over xGMI" if transport == "rccl" else
                        "halos, all-reduces and particle migration staged through host shared memory (lfa_dist_init_shm)" +
                        (f", {world} ranks on {n_dev} GPU(s)" if shared_gpu else "")))
    elif world > 1:
        parallelism = f"{world} independent replicas (--replicas)"
    return sim, size, blo, bhi, parallelism, (transport if not transport_note else f"shm ({transport_note})")


def med(xs):
    return statistics.median(xs) if xs else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None, help="C2 | C3 | C4 | C5 (default: C4 at 1 GPU, see DESIGN.md)")
    ap.add_argument("--dt-max", type=float, default=0.033, help="cap of simulation::time_step(): dt = min(3 cfl, 0.033)")
    ap.add_argument("--precond", default="multigrid", choices=["multigrid", "multilevel", "tiled", "exact"])
    ap.add_argument("--pcg-dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--p2g", default="binned", choices=["binned", "atomic"])
    ap.add_argument("--max-iterations", type=int, default=200, help="PCG iteration cap (pressure_solver.h:42)")
    ap.add_argument("--cpu-sample", default="C2")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-skip-full-step", action="store_true",
                    help="cpu_baseline: hot-path steps only (at C4 the reference's full time_step takes minutes per step)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-hot-path", action="store_true")
    ap.add_argument("--no-mic0-record", action="store_true", help="skip the reference-comparable MIC(0)-PCG figure")
    ap.add_argument("--hot-steps", type=int, default=10)
    ap.add_argument("--no-overlap", action="store_true", help="stages of time_step back to back (lfa_set_step_overlap(0)): "
                    "what the per-kernel profiles under profiles/ are taken with")
    ap.add_argument("--no-serial-stages", action="store_true", help="skip the second loop (stages back to back) that the per-stage "
                    "figures come from: stage_ms_* are then the stretched spans of the overlapped steps")
    ap.add_argument("--unfused", action="store_true", help="one launch per vector operation in the PCG loop (pcg_fused = 0)")
    ap.add_argument("--obstacle", action="store_true", help="BASELINE configs[4]: voxelize a sphere mesh on the device "
                    "(lfa_voxelize_mesh) and mark it solid before the steps")
    ap.add_argument("--mesh", action="store_true", help="BASELINE configs[4]: extract the surface mesh from the resident particles "
                    "after the timed steps (lfa_mesher_sample_sim + marching cubes), timed separately")
    ap.add_argument("--replicas", action="store_true", help="N > 1: independent copies of the domain instead of z-slabs")
    ap.add_argument("--force-slabs", action="store_true", help="run the z-slab code path (RCCL communicator, halo calls, global CFL "
                    "reduction) even with one rank: the only way to exercise it end to end on a 1-GPU box")
    ap.add_argument("--late", type=int, default=0, help="after everything else: run on to step N of the dam break (untimed), then time "
                    "`--late-steps` full steps there and report them as `late_phase` (the fluid has spread over many partly filled tiles)")
    ap.add_argument("--late-steps", type=int, default=20)
    ap.add_argument("--transport", choices=["auto", "rccl", "shm", "auto-shm"], default="auto", help="z-slab messages: RCCL send/recv + all-reduce on "
                    "the handle's stream, or staged through host shared memory (lfa_dist_init_shm: functional, two PCIe crossings per "
                    "message). auto = rccl when every rank has its own GPU (a communicator that cannot be created ENDS the run with an "
                    "error: a host-staged number must not appear under the headline's name), shm when ranks share a GPU; auto-shm = as "
                    "auto, but fall back to shm if the communicator cannot be created")
    ap.add_argument("--strong", action="store_true", help="N > 1: the FIXED BASELINE domain (configs[3]/[4]) split into N z-slabs - the "
                    "configuration BASELINE.json quotes its multi-GPU target on, and the default headline of an N > 1 run")
    ap.add_argument("--weak", action="store_true", help="N > 1: make the weak-scaling run (domain and block grow along z with N, every rank "
                    "owns a single-GPU-sized slab) the headline `value` instead of the fixed domain")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: skip the other scaling mode's run (reported under `weak` / `strong`)")
    ap.add_argument("--watchdog-s", type=int, default=1500, help="N > 1: a rank that has printed no result after this many seconds exits with "
                    "code 124 instead of waiting for ever on a peer (0 = off)")
    args = ap.parse_args()
    if args.weak and args.strong:
        raise SystemExit("--weak and --strong exclude each other")

    if args.gpus > 1 and "RANK" not in os.environ:
        # `--gpus N` without N ranks: this process is not a rank. Start the ranks the way the driver does (nothing here has touched
        # a GPU yet - the children are new processes, this one only waits and hands their exit code on) rather than print a
        # one-GPU line labelled as something else.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without a rank environment: launching `%s`" % (args.gpus, " ".join(cmd)), file=sys.stderr)
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run")
    if world > 1 and args.watchdog_s > 0:
        # A rank that waits for a message that never comes (a peer died, a collective out of step) would hang the job for ever:
        # a stream synchronisation has no timeout. The run ends with an error instead of holding the node.
        import threading

        def expired():
            print(f"bench.py: rank {rank}: no result after {args.watchdog_s} s (--watchdog-s) - a rank is stuck in a collective or died; "
                  "aborting", file=sys.stderr, flush=True)
            os._exit(124)
        wd = threading.Timer(args.watchdog_s, expired)
        wd.daemon = True
        wd.start()

    import torch
    import libfluid_amd as lfa
    from libfluid_amd import scenes

    dist = None
    n_dev = torch.cuda.device_count()
    shared_gpu = world > n_dev  # more ranks than GPUs (a 1-GPU box running the N-process path): RCCL cannot be used
    if shared_gpu and args.transport == "rccl":
        raise SystemExit(f"--transport rccl needs one GPU per rank ({world} ranks, {n_dev} GPUs)")
    transport = "shm" if (shared_gpu or args.transport == "shm") else "rccl"
    local_rank %= max(n_dev, 1)
    tdev = "cuda"  # where the tensors of the torch.distributed collectives live
    if world > 1 or (args.force_slabs and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo")
            tdev = "cpu"
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    cfg_name = args.config or "C4"
    cfg = dict(scenes.CONFIGS[cfg_name])
    slabs = (world > 1 and not args.replicas) or args.force_slabs
    # N > 1: the headline is the FIXED BASELINE domain split into N slabs (what configs[3] and the north star's ">= 6x at 8 GPUs"
    # are quoted on); the weak-scaling run is reported beside it (`weak`). --weak swaps the two.
    strong = slabs and not args.weak
    sim, size, blo, bhi, parallelism, transport_used = build_sim(args, cfg, lfa, torch, dist, tdev, rank, world, local_rank, n_dev, shared_gpu,
                                                                 transport, slabs, strong)
    fused = not args.unfused and args.precond != "exact"
    PCG_BYTES = PCG_BYTES_MG if args.precond == "multigrid" else (PCG_BYTES_FUSED if fused else PCG_BYTES_UNFUSED)
    extras = {}
    if slabs:
        extras["transport"] = transport_used
    if args.obstacle:
        # a sphere in the dry part of the tank, in the path of the collapsing column, voxelized on the device and marked solid
        # without leaving it (it must not overlap the seeded block: particles deep inside a solid give rows without a diagonal)
        rad = 0.16 * min(bhi[0] - blo[0], bhi[1] - blo[1], cfg["block"][1][2] - blo[2])
        ctr = [min(bhi[0] + 2.0 * rad, size[0] - 1.5 * rad), blo[1] + 1.2 * rad, 0.5 * (blo[2] + cfg["block"][1][2])]
        mpos, midx = scenes.icosphere(ctr, rad, 5)
        t0 = time.perf_counter()
        vox = lfa.Voxels.from_mesh(mpos, midx, 1.0, (0.0, 0.0, 0.0), device=local_rank)
        sim.set_solid_from_voxels(vox, True, True)
        extras["voxelizer"] = {"triangles": int(len(midx) // 3), "voxel_grid": list(vox.size),
                               "ms_mesh_to_solid_cells": 1e3 * (time.perf_counter() - t0),
                               "solid_cells": int(len(vox.cells(True, True, size)))}
        vox.close()
    sim.seed_block(blo, bhi)
    sim.enable_timing(True)
    if args.no_overlap:
        sim.set_step_overlap(False)

    def barrier():
        sim.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def global_cfl():
        """simulation::cfl over the whole domain: each rank reduces its own particles, the minimum over ranks is the CFL step."""
        c = sim.cfl()
        if slabs and dist is not None:
            t = torch.tensor([c], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            c = float(t.item())
        return c

    marks = []  # host clock each time the CFL read-back returns: the device has finished the step before (one stamp per step)

    def one_step():
        dt = min(3.0 * global_cfl(), args.dt_max)
        marks.append(time.perf_counter())
        res, it, rc = sim.time_step(dt)
        return dt, it, rc

    preroll = max(0, MIN_LEAD_IN - args.warmup)
    t_sim = 0.0
    for _ in range(preroll + args.warmup):
        dt, _, _ = one_step()
        t_sim += dt
    barrier()
    t0 = time.perf_counter()
    iters_total, not_converged, per_step, dts = 0, 0, [], []
    for _ in range(args.steps):
        dt, it, rc = one_step()
        iters_total += it
        dts.append(dt)
        not_converged += int(rc == lfa.W_PCG_NOT_CONVERGED)
        per_step.append(sim.step_timings())
    barrier()
    elapsed = time.perf_counter() - t0
    # wall time of each timed step (CFL read-back to CFL read-back; the last one ends at the closing barrier): min / median / max
    # beside the device's own span of lfa_time_step, so that a gap between `ms_per_step` and the device time is attributable -
    # a constant host cost per step shows in the median, outlier steps in max and p95
    stamps = marks[-args.steps:] + [t0 + elapsed]
    wall = sorted(1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:]))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counts = sim.counts()
    corr_fallback = sim.correction_stats_ex()
    npart, n_unknowns = counts["particles"], counts["unknowns"]
    npart_total = npart
    if dist is not None:
        t = torch.tensor([float(npart)], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        npart_total = int(t.item())
    value = npart_total * args.steps / elapsed
    names = [k for k in per_step[0] if k not in ("pcg_iterations", "overlapped")]
    overlapped = bool(per_step[0].get("overlapped"))
    stage_overlapped = {k: med([s[k] for s in per_step]) for k in names} if overlapped else None
    iters_timed = iters_total
    serial_ms = None
    if overlapped and not args.no_serial_stages:
        # Stage attribution: in the timed steps the position correction runs on its own stream beside the pressure solve, so the
        # stage spans stretch each other and do not add up. The same steps back to back (lfa_set_step_overlap(0)), AFTER the
        # timed region, give the per-stage / per-kernel times every roofline figure below is priced on; `value` is not.
        sim.set_step_overlap(False)
        per_step, iters_total = [], 0
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            dt, it, rc = one_step()
            iters_total += it
            per_step.append(sim.step_timings())
        barrier()
        serial_ms = 1e3 * (time.perf_counter() - ts) / args.steps
        sim.set_step_overlap(True)
    stage_med = {k: med([s[k] for s in per_step]) for k in names}
    stage_p95 = {k: sorted(s[k] for s in per_step)[min(len(per_step) - 1, int(0.95 * len(per_step)))] for k in names}
    pcg_s = sum(s["pcg_loop"] for s in per_step) * 1e-3

    out = {
        "metric": "particle_steps_per_sec", "value": value, "unit": "particle-steps/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak" if ((slabs and not strong) or (world > 1 and not slabs)) else "strong", "vs_baseline": None,
        "dtype": "f32" if args.pcg_dtype == "f32" else "f32 particles/grid, f64 PCG vectors", "data": "synthetic",
        "config": {
            "workload": f"{cfg_name}: {size[0]}x{size[1]}x{size[2]} MAC grid, dam-break block "
                        f"{tuple(blo)}-{tuple(bhi)} cells, {npart_total} particles, "
                        f"{['PIC', 'FLIP', 'APIC'][cfg['method']]} blend {cfg['blending']}, full simulation::time_step "
                        f"(cfl + advect/collide + bin + P2G + gravity + PCG + apply + correct/collide + extrapolate + G2P), "
                        f"dt = min(3 cfl, {args.dt_max}), timed from step {preroll + args.warmup} of the dam break "
                        f"(t = {t_sim:.3f} s .. {t_sim + sum(dts):.3f} s)",
            "preroll_steps": preroll, "dt_mean": sum(dts) / len(dts),
            "unknowns": n_unknowns, "particles_per_gpu": npart,
            "precond": {"tiled": "MIC(0) per 8^3 tile", "exact": "MIC(0) exact (tile hyperplanes)",
                        "multilevel": "MIC(0) per 8^3 tile + tile-aggregate coarse correction",
                        "multigrid": "geometric multigrid V(1,1), red-black Gauss-Seidel per tile"}[args.precond],
            "p2g": args.p2g, "pcg_tolerance": 1e-6, "pcg_max_iterations": args.max_iterations,
            "parallelism": parallelism,
        },
        "pcg": {
            "iterations_per_step": iters_total / max(args.steps, 1),
            "iterations_per_step_timed_region": iters_timed / max(args.steps, 1),
            "iters_per_sec": iters_total / pcg_s if pcg_s > 0 else None,
            "unknown_iters_per_sec": n_unknowns * iters_total / pcg_s if pcg_s > 0 else None,
            "steps_hitting_max_iterations": not_converged,
            "solver_stats_last_solve": sim.solver_stats(),
            "note": "iterations of the preconditioner named in config.precond; the reference-comparable MIC(0) figure is "
                    "`--precond multilevel` (same iteration counts as the reference's MIC(0)-PCG within a few per cent)",
        },
        "stage_ms_median": stage_med, "stage_ms_p95": stage_p95,
        "step_wall_ms": {"min": wall[0], "median": med(wall), "p95": wall[min(len(wall) - 1, int(0.95 * len(wall)))], "max": wall[-1],
                         "device_span_median": (stage_overlapped or stage_med).get("time_step"),
                         "note": "host clock between the CFL read-backs of consecutive timed steps (ms_per_step is their mean); "
                                 "device_span_median = lfa_time_step's own HIP-event span in the same steps: the difference is host-side "
                                 "(the CFL read-back, launch latency after it)"},
        "correction_fallback_half_tiles": {"flagged": corr_fallback[0], "of": corr_fallback[1], "second_pass": corr_fallback[2],
                                           "note": "half tiles of the last step whose neighbourhood did not fit the LDS-tiled correction kernel"},
    }
    if overlapped and serial_ms is not None:
        out["stage_ms_note"] = (f"stage_ms_* and every per-kernel figure: {args.steps} further steps with the stages back to back "
                                f"(lfa_set_step_overlap(0): {serial_ms:.3f} ms per step wall); `value` / ms_per_step: the timed steps "
                                "with the position correction on a second stream beside the pressure solve (the default)")
        out["ms_per_step_serial_stages"] = serial_ms
        out["stage_ms_median_overlapped"] = stage_overlapped

    if rank == 0 and world == 1 and not slabs and not args.no_kernel_timing:
        apic = cfg["method"] == 2
        flip = cfg["method"] == 1
        ncell_all = cfg["size"][0] * cfg["size"][1] * cfg["size"][2]
        ncell_proc = counts["processed_tiles"] * 512
        it_per_step = iters_total / max(args.steps, 1)
        # ---- measured HBM ceiling of this device ("report both")
        copy_gbs, read_gbs = sim.bench_stream(1 << 30, 10)
        out["hbm_ceiling_measured"] = {"device_copy_GBps": copy_gbs, "read_only_GBps": read_gbs, "spec_peak_GBps": HBM_PEAK_GBS,
                                       "how": "lfa_bench_stream: float4 grid-stride kernels over 1 GiB, mean of 10 launches, best of 5 copy "
                                              "variants (" + getattr(sim, "stream_variant", "") + "); MI355X_MICROARCH.md records 6.29 TB/s for a float4 copy"}
        # ---- in-step kernels / kernel groups: MEDIAN device time inside the timed steps (HIP events on the handle's stream)
        # algorithmic bytes: SURVEY 8(d). P2G scatter 60 Np (APIC) / 24 Np; PCG iteration 91 n (the per-kernel split below adds up to 92.5 n for the V-cycle's own
        # passes); G2P 60 Np + 12 Nc (APIC), 36 Np + 24 Nc (FLIP), 24 Np + 12 Nc (PIC); binning: the bytes the deferred scheme
        # moves (key, t, id both ways + source index = 44 Np, + 2 x 36 Np for PIC/FLIP whose C travels with the particle);
        # position correction: positions in and out, 24 Np (the reference's OMP loop reads and writes vec3d positions,
        # src/simulation.cpp:562-610; pair interactions are arithmetic, not traffic)
        # (the iteration is priced at SURVEY 8(d)'s 91 n whatever the preconditioner: what the V-cycle's own passes add is overhead)
        pcg_iter_bytes = 91 * n_unknowns * (2 if args.pcg_dtype == "f64" else 1)
        in_step = {
            "p2g_scatter_kernel": ((60 if apic else 24) * npart, 1.0),
            "pcg_iteration_mean": (pcg_iter_bytes, it_per_step),
            "g2p": ((60 if apic else (36 if flip else 24)) * npart + (24 if flip else 12) * ncell_proc, 1.0),
            "bin": ((44 if apic else 116) * npart, 1.0),
            "correct_tiled_kernel": (24 * npart, 1.0),
            "advect_collide": (44 * npart, 1.0),  # reads key, t, v (28 B), writes key, t (16 B)
        }
        kern = {}
        for k, (b, mult) in in_step.items():
            ms = stage_med[k]
            if ms > 0:
                kern[k] = {"ms_median": ms, "ms_p95": stage_p95[k], "algorithmic_bytes": int(b), "GBps": b / ms * 1e-6,
                           "frac": b / ms * 1e-6 / HBM_PEAK_GBS, "share_of_step_ms": ms * mult}
        out["in_step_kernels"] = dict(kern)
        # `traffic`: HBM bytes per launch from the PMC passes (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 runs of this same command,
        # tools/make_profiles.sh). It is NOT measured in this run: `traffic_source` names the committed file it is read from.
        pmc = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_{cfg_name.lower()}_pmc_traffic.json") for r in (6, 5, 4, 3, 2))
                    if os.path.exists(q)), os.path.join(ROOT, "profiles", f"r06_{cfg_name.lower()}_pmc_traffic.json"))
        pmc_names = {"p2g_scatter_kernel": "p2g_scatter", "correct_tiled_kernel": "correct_tiled", "g2p": "g2p",
                     "advect_collide": "advect_collide", "bin": "bin_scatter", "pcg_a": "pcg_a", "mg_axpy_presmooth": "mg_axpy_presmooth",
                     "mg_down0": "mg_down0", "mg_up0": "mg_up0", "mg_coarse": "mg_coarse"}
        # the source file a kernel lives in: a traffic figure is only as young as that file and the shared headers
        pmc_files = {"p2g_scatter": "p2g.hip", "correct_tiled": "particles.hip", "g2p": "grid_ops.hip", "advect_collide": "particles.hip",
                     "bin_scatter": "core.hip", "pcg_a": "pcg.hip", "mg_axpy_presmooth": "mg.hip", "mg_down0": "mg.hip", "mg_up0": "mg.hip",
                     "mg_coarse": "mg.hip"}
        per_launch, pmc_sha = {}, {}
        if args.pcg_dtype == "f32" and args.p2g == "binned" and os.path.exists(pmc):  # (C4 and C3 have committed PMC passes)
            rec = json.load(open(pmc))
            per_launch, pmc_sha = rec["hbm_bytes_per_launch"], rec.get("source_sha256", {})

        def traffic_of(k):
            """(bytes, source, age): the PMC figure of kernel k if its passes were taken on THIS code - the kernel's source file and
            the shared headers hash to what tools/pmc_traffic.py recorded beside the figure - else (None, None, why)."""
            name = pmc_names.get(k, "")
            if name not in per_launch:
                return None, None, "no PMC pass of this kernel is committed for this configuration"
            import hashlib
            csrc = os.path.join(ROOT, "libfluid_amd", "csrc")
            stale = [f for f in (pmc_files.get(name), "common.h", "pcg.h") if f and
                     pmc_sha.get(f) != hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()]
            if stale:
                return None, None, ("refused: " + os.path.relpath(pmc, ROOT) + " was taken on other code (" + ", ".join(stale) +
                                    (" changed since" if pmc_sha else ": the file records no source hashes") + "); re-run tools/make_profiles.sh")
            return per_launch[name]["total"], os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc passes of the same command, not this run)", \
                "current: kernel source and shared headers hash to what the PMC passes recorded"

        def roofline_of(k, note):
            tb, tsrc, tage = traffic_of(k)
            return {"bound": "hbm", "kernel": k, "achieved": kern[k]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kern[k]["frac"], "traffic": tb, "traffic_source": tsrc, "traffic_age": tage,
                    "algorithmic_bytes": kern[k]["algorithmic_bytes"], "ms": kern[k]["ms_median"],
                    "share_of_step_ms": kern[k]["share_of_step_ms"], "note": note}
        # `roofline`: the dominant single kernel of the HOT PATH (binning, P2G, PCG, G2P - the path north_star
        # names) inside the timed full steps. The PCG loop is 17 launches per iteration, none of them larger than 1.1 ms per step
        # in total; its iteration as a whole is priced under roofline_groups. `roofline_full_step`: the largest kernel of the
        # whole step, which is the position correction's pair kernel (a SURVEY 8f "next" row; VALU bound, not a bandwidth story).
        # The candidates are every kernel measured in this run that belongs to the hot path - the in-step ones and the PCG loop's
        # launches (isolated time x iterations per step) -, the dominant one is picked from that table, not from a fixed list.
        for name in PCG_BYTES:
            if PCG_BYTES[name] <= 0:
                continue  # (the coarse levels' single launch is latency bound: no byte figure to price)
            ms = sim.bench_kernel(name, 20)
            b = int(PCG_BYTES[name] * n_unknowns * (2 if args.pcg_dtype == "f64" else 1))
            kern[name] = {"ms_median": ms, "ms_p95": ms, "algorithmic_bytes": b, "GBps": b / ms * 1e-6, "frac": b / ms * 1e-6 / HBM_PEAK_GBS,
                          "share_of_step_ms": ms * it_per_step, "how": "lfa_bench_kernel: 20 launches back to back on the last state"}
        NOT_HOT = ("correct_tiled_kernel", "advect_collide", "pcg_iteration_mean")  # SURVEY 8f rows; a group of launches
        hot = [k for k in kern if k not in NOT_HOT]
        dom = max(hot, key=lambda k: kern[k]["share_of_step_ms"])
        out["roofline_candidates"] = {k: round(kern[k]["share_of_step_ms"], 4) for k in sorted(hot, key=lambda k: -kern[k]["share_of_step_ms"])}
        out["roofline"] = roofline_of(dom, "dominant single kernel of the hot path (SURVEY 8a) by share of the median full step; "
                                           "duration = in-step median (HIP events on the handle's stream)")
        if stage_overlapped is not None and stage_overlapped.get(dom, 0) > 0:
            # the P2G / G2P / binning kernels run outside the window in which the correction shares the device with the solve: their
            # launch duration IN THE TIMED REGION is the figure the contract asks for (the serial loop's is kept beside it)
            ms_t = stage_overlapped[dom]
            r = out["roofline"]
            r["ms_serial_loop"], r["ms"] = r["ms"], ms_t
            r["achieved"] = r["algorithmic_bytes"] / ms_t * 1e-6
            r["frac"] = r["achieved"] / HBM_PEAK_GBS
            r["note"] = ("dominant single kernel of the hot path (SURVEY 8a) by share of the median full step; duration = median launch "
                         "duration over the TIMED steps (HIP events on the handle's stream; this kernel runs before the correction's "
                         "stream is forked, so the overlap does not stretch it); ms_serial_loop = the same in the stage-attribution loop")
        dom_all = max(kern, key=lambda k: kern[k]["share_of_step_ms"] if k != "pcg_iteration_mean" else 0.0)
        out["roofline_full_step"] = roofline_of(dom_all, "largest kernel of the whole time_step; the position correction is VALU bound "
                                                         "(PMC: 69 % of the SIMD cycles issue VALU), its HBM fraction is not a quality measure")
        p2g_b = (60 if apic else 24) * npart + (26 if flip else 14) * ncell_all
        p2g_pcg_ms = stage_med["p2g"] + stage_med["pcg_loop"]
        p2g_pcg_b = p2g_b + pcg_iter_bytes * it_per_step
        out["roofline_groups"] = {
            "p2g": {"ms": stage_med["p2g"], "algorithmic_bytes": int(p2g_b), "GBps": p2g_b / stage_med["p2g"] * 1e-6,
                    "frac": p2g_b / stage_med["p2g"] * 1e-6 / HBM_PEAK_GBS},
            "pcg_iteration": {"ms": stage_med["pcg_iteration_mean"], "algorithmic_bytes": int(pcg_iter_bytes),
                              "GBps": kern["pcg_iteration_mean"]["GBps"], "frac": kern["pcg_iteration_mean"]["frac"]},
            "p2g_plus_pcg": {"ms": p2g_pcg_ms, "algorithmic_bytes": int(p2g_pcg_b), "GBps": p2g_pcg_b / p2g_pcg_ms * 1e-6,
                             "frac": p2g_pcg_b / p2g_pcg_ms * 1e-6 / HBM_PEAK_GBS,
                             "note": "the north star's target group (>= 0.40): in-step medians, 60 Np + 14 Nc + 91 n x iterations (SURVEY 8d)"},
        }
        # ---- isolated kernels, back to back on the state of the last step (lfa_bench_kernel): the PCG loop's launches
        kernels = {}
        for name in PCG_BYTES:
            ms = kern[name]["ms_median"] if name in kern else sim.bench_kernel(name, 20)
            b = int(PCG_BYTES[name] * n_unknowns * (2 if args.pcg_dtype == "f64" else 1))
            kernels[name] = {"ms": ms, "algorithmic_bytes": b, "GBps": b / ms * 1e-6}
        out["kernels_isolated"] = kernels

    if args.late > 0:
        # (Before the hot-path and MIC(0) sections below: their lfa_step_hot calls - gravity kicks of dt_max without advection - would
        # perturb the flow the window is meant to measure; round 5 had it behind them.)
        # A second timed window late in the run: the sheet has spread, tiles are partly filled, the solve has more of them to visit
        # and the position correction meets crowded blocks where the fluid has piled up. Same step, same accounting.
        done_steps = preroll + args.warmup + args.steps * (2 if (overlapped and not args.no_serial_stages) else 1)
        while done_steps < args.late:
            one_step()
            done_steps += 1
        barrier()
        tl = time.perf_counter()
        late_it, late_stage = 0, []
        for _ in range(args.late_steps):
            _, it, _ = one_step()
            late_it += it
            late_stage.append(sim.step_timings())
        barrier()
        late_s = time.perf_counter() - tl
        if dist is not None:
            t = torch.tensor([late_s], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            late_s = float(t.item())
        lc, lf = sim.counts(), sim.correction_stats_ex()
        late_names = [k for k in late_stage[0] if k not in ("pcg_iterations", "overlapped")]
        extras["late_phase"] = {
            "from_step": done_steps, "steps": args.late_steps, "ms_per_step": 1e3 * late_s / args.late_steps,
            "ratio_to_timed_region": (late_s / args.late_steps) / (elapsed / args.steps),
            "particle_steps_per_sec": lc["particles"] * args.late_steps / late_s,
            "pcg_iterations_per_step": late_it / args.late_steps,
            "particle_tiles": lc["particle_tiles"], "processed_tiles": lc["processed_tiles"], "unknowns": lc["unknowns"],
            "particles_per_particle_tile": lc["particles"] / max(lc["particle_tiles"], 1),
            "particle_tiles_timed_region": counts["particle_tiles"], "processed_tiles_timed_region": counts["processed_tiles"],
            "correction_fallback_half_tiles": {"flagged": lf[0], "of": lf[1], "second_pass": lf[2]},
            "stage_ms_median": {k: med([s_[k] for s_ in late_stage]) for k in late_names},
        }
    if world == 1 and not slabs and not args.no_hot_path:
        # secondary figure: the hot path alone (no advection / correction) on the state the dam has reached
        hot_ms, hot_it = [], 0
        stage = {}
        for _ in range(2):
            sim.step_hot(args.dt_max)
        sim.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.hot_steps):
            _, it, _ = sim.step_hot(args.dt_max)
            hot_it += it
            for k, v in sim.timings().items():
                stage.setdefault(k, []).append(v)
        sim.synchronize()
        hs = time.perf_counter() - t1
        out["hot_path"] = {"steps": args.hot_steps, "ms_per_step": 1e3 * hs / args.hot_steps,
                           "particle_steps_per_sec": npart * args.hot_steps / hs,
                           "pcg_iterations_per_step": hot_it / args.hot_steps,
                           "stage_ms_median": {k: med(v) for k, v in stage.items()},
                           "note": "lfa_step_hot on the state reached by the timed steps: bin + P2G + gravity + PCG + apply + "
                                   "extrapolate + G2P without advection (round 1's headline)"}
    if world == 1 and not slabs and rank == 0 and not args.no_mic0_record and args.precond == "multigrid":
        # The reference's PCG is MIC(0)-preconditioned (src/pressure_solver.cpp:19-71): the iteration rate that can be set beside
        # its iterations/s is the one of the MIC(0)-based preconditioner (tile-local MIC(0) + coarse correction: the reference's
        # iteration counts within a few per cent), measured on the same state with a second handle's worth of parameters.
        sim.set_params(precond=lfa.PRECOND_MULTILEVEL)
        its, ms = [], []
        sim.step_hot(args.dt_max)
        for _ in range(3):
            _, it, rc = sim.step_hot(args.dt_max)
            t_ = sim.timings()
            its.append(int(it)); ms.append(t_["pcg_loop"])
        out["pcg_mic0"] = {"precond": "MIC(0) per 8^3 tile + tile-aggregate coarse correction (--precond multilevel)",
                           "iterations_per_solve": its, "pcg_loop_ms": ms,
                           "iters_per_sec": sum(its) / (sum(ms) * 1e-3) if sum(ms) > 0 else None,
                           "unknown_iters_per_sec": n_unknowns * sum(its) / (sum(ms) * 1e-3) if sum(ms) > 0 else None,
                           "ms_per_iteration": sum(ms) / max(sum(its), 1),
                           "algorithmic_GBps": 91 * n_unknowns * sum(its) / (sum(ms) * 1e-3) * 1e-9 if sum(ms) > 0 else None,
                           "note": "91 n bytes per iteration (SURVEY 8d); the reference does 58 MIC(0) iterations at C2 where this does 77"}
        sim.set_params(precond=lfa.PRECOND_MULTIGRID)
    if args.mesh and (world == 1 or slabs):
        # BASELINE configs[4]: the surface of the resident particles (mesher settings of testbed/main.cpp:101-107 at cell size 1).
        # On slabs every rank meshes its own cell layers (lfa_mesher_create_window) from its own particles and the ghost copies of
        # its neighbours' adjacent tile layers; the only exchange is the exclusive scan of the vertex counts.
        window = None
        if slabs:
            sim.hash()
            lo, hi = sim.slab()
            window = (lo * 8, min(hi * 8, size[2]))
        m = lfa.Mesher(size, (0.0, 0.0, 0.0), 1.0, 1.0, 2, device=local_rank, window=window)
        barrier()
        t0 = time.perf_counter()
        m.sample_sim(sim, 0.5)
        t1 = time.perf_counter()
        m.marching_cubes()
        nv, ni = m._counts
        if slabs and dist is not None:
            counts = torch.zeros(world, dtype=torch.int64, device=tdev)
            counts[rank] = nv
            dist.all_reduce(counts)
            m.rebase(int(counts[:rank].sum().item()))
            tot = torch.tensor([float(nv), float(ni)], dtype=torch.float64, device=tdev)
            dist.all_reduce(tot)
            nv, ni = int(tot[0].item()), int(tot[1].item())
        mpos, midx = m.download_mesh()
        t2 = time.perf_counter()
        extras["mesher"] = {"grid_points": (size[0] + 1) * (size[1] + 1) * (size[2] + 1), "sample_ms": 1e3 * (t1 - t0),
                            "marching_cubes_ms_incl_download": 1e3 * (t2 - t1), "vertices": int(nv), "triangles": int(ni // 3),
                            "windows": world if slabs else 1}
        m.close()
    out.update(extras)
    if dist is not None:  # every rank has finished with the transport before any rank closes its side of it
        sim.synchronize()
        dist.barrier()
    sim.close()
    if slabs and world > 1 and not args.no_secondary:
        # the other scaling mode, beside the headline: same step, same lead-in, same barrier + max-over-ranks timing, no stage breakdown
        other = not strong
        sim2, size2, blo2, bhi2, par2, tr2 = build_sim(args, cfg, lfa, torch, dist, tdev, rank, world, local_rank, n_dev, shared_gpu,
                                                      transport, slabs, other)
        sim2.seed_block(blo2, bhi2)
        if args.no_overlap:
            sim2.set_step_overlap(False)

        def step2():
            c = sim2.cfl()
            if dist is not None:
                t_ = torch.tensor([c], dtype=torch.float64, device=tdev)
                dist.all_reduce(t_, op=dist.ReduceOp.MIN)
                c = float(t_.item())
            return sim2.time_step(min(3.0 * c, args.dt_max))

        def barrier2():
            sim2.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
        for _ in range(preroll + args.warmup):
            step2()
        barrier2()
        t2 = time.perf_counter()
        it2 = 0
        for _ in range(args.steps):
            _, it, _ = step2()
            it2 += it
        barrier2()
        el2 = time.perf_counter() - t2
        n2 = float(sim2.counts()["particles"])
        if dist is not None:
            t_ = torch.tensor([el2, -n2], dtype=torch.float64, device=tdev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            el2 = float(t_[0].item())
            t_ = torch.tensor([n2], dtype=torch.float64, device=tdev)
            dist.all_reduce(t_, op=dist.ReduceOp.SUM)
            n2 = float(t_.item())
        out["strong" if other else "weak"] = {
            "value": n2 * args.steps / el2, "unit": "particle-steps/s", "ms_per_step": 1e3 * el2 / args.steps, "steps": args.steps,
            "scaling": "strong" if other else "weak", "particles": int(n2), "grid": size2, "parallelism": par2, "transport": tr2,
            "pcg_iterations_per_step": it2 / max(args.steps, 1),
            "note": "the other scaling mode of the same command (the line's `value` is the " + ("weak" if other else "fixed-domain") + " run)"}
        barrier2()
        sim2.close()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_steps, lfa, args.cpu_skip_full_step)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
