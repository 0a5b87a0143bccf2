/* libfluid_amd.h -- C ABI of the MI355X-native PIC/FLIP/APIC hot path (drop-in for lukedan/libfluid's per-step path).
 *
 * The reference has no FFI: its boundary is the C++ class fluid::simulation (include/fluid/simulation.h:21-280) used
 * identically by the testbed (testbed/main.cpp:91-99,189) and the Maya GridNode (plugins/maya/nodes/grid_node.cpp:
 * 256-274,351-357). This header is the thin layer the north star asks for *beneath* a class with that surface
 * (libfluid_amd/host/simulation.h): plain pointers and sizes, host layouts exactly as the reference holds them.
 *
 *   particles : simulation::particle, 152-B fp64 AoS {position, velocity, cx, cy, cz, old_position, raw_cell_index}
 *               (include/fluid/simulation.h:24-34)
 *   cells     : mac_grid::cell, 32-B AoS {vec3d velocities_posface, u8 type{air=1,fluid=2,solid=4}, pad}, x fastest
 *               (include/fluid/mac_grid.h:15-27, include/fluid/data_structures/grid.h:11-12)
 *   solids    : flat int[3k] (x,y,z) triples as the Maya plugin passes them (plugins/maya/nodes/grid_node.cpp:330-339)
 *   pressure  : double[n] in the reference's unknown order = ascending raw cell index of occupied cells
 *               (src/simulation.cpp:83-94, include/fluid/simulation.h:166)
 *
 * Every function returns LFA_OK (0) or a negative LFA_E_* code; lfa_last_error() gives the text. The library never
 * keeps a host pointer past the call. One host thread per handle; different handles are independent.
 * There is no CPU fallback: without a usable HIP device lfa_create fails with LFA_E_NO_DEVICE.
 */
#ifndef LIBFLUID_AMD_H
#define LIBFLUID_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lfa_sim lfa_sim;

enum {
	LFA_OK = 0,
	LFA_E_INVALID = -1,      /* bad argument / size / state */
	LFA_E_NO_DEVICE = -2,    /* no HIP device or device init failed */
	LFA_E_HIP = -3,          /* a HIP runtime call failed */
	LFA_E_OOM = -4,          /* device allocation failed */
	LFA_E_NAN = -5,          /* NaN/Inf detected in the pressure solve */
	LFA_E_UNSUPPORTED = -6,  /* parameter combination not implemented on the device path */
	LFA_W_PCG_NOT_CONVERGED = 1 /* warning: PCG stopped at max_iterations (reference is silent: pressure_solver.cpp:45) */
};

/* simulation::method, include/fluid/simulation.h:44-48 */
enum { LFA_PIC = 0, LFA_FLIP_BLEND = 1, LFA_APIC = 2 };
/* P2G scatter variant (BASELINE config 2: "P2G atomics vs LDS-binned") */
enum { LFA_P2G_LDS_BINNED = 0, LFA_P2G_GLOBAL_ATOMIC = 1 };
/* MIC(0) schedule. EXACT = hyperplane order over tiles: the same recurrence as pressure_solver.cpp:244-332, hence the
 * same iteration counts as the reference. TILED = MIC(0) restricted to 8x8x8 tiles (couplings across tile faces
 * dropped from the preconditioner only): one launch per application, more iterations, same converged pressure. */
enum { LFA_PRECOND_MIC0_TILED = 0, LFA_PRECOND_MIC0_EXACT = 1,
       /* MULTILEVEL = TILED + additive coarse-space correction: one piecewise-constant unknown per 8^3 tile (Galerkin
        * operator, block MIC(0) again on 8^3 blocks of tiles) and a dense solve of the 64^3-cell aggregates on top.
        * Restores mesh-independent-ish convergence at one launch per level. */
       LFA_PRECOND_MULTILEVEL = 2,
       /* MULTIGRID = geometric multigrid V-cycle (levels of 2x coarser cells down to one tile, rediscretised operators,
        * red-black Gauss-Seidel inside tiles / Jacobi across tile faces, piecewise-constant transfers): the work per
        * iteration stays O(unknowns) and the iteration count barely grows with the grid (about 20 at 512^3 where MIC(0)
        * needs 180). Runs on a slab decomposition too (finest levels distributed, coarser ones replicated, see the slab
        * section below). Same converged pressure, same stopping rule. The default. */
       LFA_PRECOND_MULTIGRID = 3 };
/* arithmetic type of the PCG vectors */
enum { LFA_PCG_F32 = 0, LFA_PCG_F64 = 1 };

/* Public fields of fluid::simulation (include/fluid/simulation.h:179-190) and fluid::pressure_solver
 * (include/fluid/pressure_solver.h:38-42), same names and defaults, plus device-path selectors. */
typedef struct lfa_params {
	double grid_offset[3];
	double gravity[3];
	double cell_size;             /* reference default NaN: must be set */
	double blending_factor;       /* 1.0 */
	double density;               /* 1.0 */
	double boundary_skin_width;   /* 0.1 (used by the "next" rows only) */
	double correction_stiffness;  /* 5.0 (used by the "next" rows only) */
	double cfl_number;            /* 3.0 */
	uint64_t velocity_extrapolation_iterations; /* 1; device path supports 0..8 */
	int32_t simulation_method;    /* LFA_APIC */
	double tau, sigma, tolerance; /* 0.97, 0.25, 1e-6 */
	uint64_t max_iterations;      /* 200 */
	int32_t p2g_variant;          /* LFA_P2G_LDS_BINNED */
	int32_t precond;              /* LFA_PRECOND_MULTIGRID (single domain and slabs) */
	int32_t pcg_dtype;            /* LFA_PCG_F32 */
	int32_t apic_unscaled_kernel; /* 1 (default) = keep the reference quirk simulation.cpp:367-369: the APIC P2G hat is
	                                 evaluated on world-space distances (only differs when cell_size != 1: narrower hat for
	                                 cell_size > 1, wider - truncated by the 27-cell gather - for cell_size < 1);
	                                 0 = divide by cell_size like the PIC transfer does */
	int32_t pcg_fused;            /* 1 = two launches per PCG iteration (k_pcg_a: search direction + A s, k_pcg_b: AXPYs +
	                                 MIC(0) sweeps) where the schedule allows it (single domain, tile-local MIC(0));
	                                 0 = one launch per vector operation. Same arithmetic either way. */
	int32_t pcg_warm_start;       /* 0 (default) = every solve starts from p = 0 like the reference (src/pressure_solver.cpp:36).
	                                 1 = the PCG starts from the pressure of the previous solve where a tile was solved then
	                                 (r = b - A p): same system, same stopping rule, fewer iterations when consecutive steps
	                                 resemble each other (a settling pool; in the violent phase of the C4 dam break 16 -> 15).
	                                 Never with LFA_PRECOND_MIC0_EXACT (the reference-parity schedule) or slabs. */
} lfa_params;

/* -- lifetime ------------------------------------------------------------------------------------------------ */
void lfa_default_params(lfa_params *p);
/* simulation::resize (include/fluid/simulation.h:57) + device selection. device < 0 => current device. */
int lfa_create(lfa_sim **out, uint64_t nx, uint64_t ny, uint64_t nz, int device);
void lfa_destroy(lfa_sim *s);
const char *lfa_last_error(const lfa_sim *s); /* s may be NULL: error of the last failed lfa_create on this thread */
int lfa_set_params(lfa_sim *s, const lfa_params *p);
int lfa_get_params(const lfa_sim *s, lfa_params *p);
int lfa_synchronize(lfa_sim *s);
/* HIP stream all kernels of this handle are launched on (hipStream_t as void*), for external timing/graphs. */
void *lfa_stream(lfa_sim *s);

/* -- data in / out (reference host layouts) ------------------------------------------------------------------ */
/* Replaces the particle set (simulation::particles(), include/fluid/simulation.h:142). Converts to the device layout
 * (fp32 SoA, cell-relative positions) and computes the clamped cell key of src/simulation.cpp:251-261 in fp64. */
int lfa_upload_particles(lfa_sim *s, const void *aos152, uint64_t n);
/* Writes velocity, cx, cy, cz and raw_cell_index (x-fastest raw index, as the reference stores it) of every particle
 * back into the caller's array, in upload order (particle i of the upload is element i).
 * flags: LFA_DL_POSITIONS also writes position/old_position (reconstructed from the device's cell-relative
 * representation); LFA_DL_KEEP_RAW leaves raw_cell_index untouched (the reference's value is the one of the last
 * hash, src/simulation.cpp:62, not the one of the final position). */
enum { LFA_DL_POSITIONS = 1, LFA_DL_KEEP_RAW = 2 };
int lfa_download_particles(lfa_sim *s, void *aos152, uint64_t n, int flags);
uint64_t lfa_num_particles(const lfa_sim *s);
/* With a slab decomposition particles migrate between ranks: lfa_num_particles is the number resident on this rank
 * after the last lfa_hash_particles, lfa_download_particles writes them in storage order, and this call returns the
 * global id of each record (upload index / index in the seeded block). Single domain: ids[i] = i. */
int lfa_download_particle_ids(lfa_sim *s, uint32_t *ids, uint64_t n);
/* Synthetic dam-break block [lo,hi) in cells, 8 jittered particles per cell, generated on the device; bit-identical
 * to libfluid_amd/scenes.py:seed_block. */
int lfa_seed_block(lfa_sim *s, const int64_t lo[3], const int64_t hi[3], uint64_t seed);
/* Marks cells solid (flat int[3k] triples). lfa_clear_solid_cells resets every cell to non-solid. */
int lfa_set_solid_cells(lfa_sim *s, const int32_t *xyz, uint64_t k);
int lfa_clear_solid_cells(lfa_sim *s);
/* Whole MAC grid in the reference layout (32-B AoS, x fastest). upload takes velocities and solid flags. */
int lfa_upload_cells(lfa_sim *s, const void *aos32);
int lfa_download_cells(lfa_sim *s, void *aos32);
int lfa_download_old_cells(lfa_sim *s, void *aos32); /* FLIP's _old_grid (src/simulation.cpp:340-344) */

/* -- hot-path stages, individually callable for parity tests ---------------------------------- */
/* a2: bins particles by 8x8x8 tile and counts particles per cell (src/simulation.cpp:251-291). */
int lfa_hash_particles(lfa_sim *s);
uint64_t lfa_num_fluid_cells(lfa_sim *s);
/* _fluid_cells: ascending raw indices of cells holding particles (include/fluid/simulation.h:209). */
int lfa_download_fluid_cells(lfa_sim *s, uint64_t *raw, uint64_t n);
/* per-cell particle counts (the `count` half of _space_hash, include/fluid/simulation.h:193-197,207), x fastest. */
int lfa_download_cell_counts(lfa_sim *s, uint32_t *count);
/* a4-a6: simulation::_transfer_to_grid (src/simulation.cpp:400-412). */
int lfa_p2g(lfa_sim *s);
/* a7: gravity loop (src/simulation.cpp:72-78). */
int lfa_add_gravity(lfa_sim *s, double dt);
/* a8-a12: unknown set, A bits, divergence rhs, MIC(0) factor (src/pressure_solver.cpp:19-25,150-294). */
int lfa_build_system(lfa_sim *s, double dt);
int lfa_download_abits(lfa_sim *s, uint8_t *bits, uint64_t n);   /* bits0-2 nonsolid, 3 xpos, 4 ypos, 5 zpos */
int lfa_download_rhs(lfa_sim *s, double *b, uint64_t n);
int lfa_download_precon(lfa_sim *s, double *precon, uint64_t n);
/* a13/a14 in isolation (vectors in the reference's unknown order). */
int lfa_apply_preconditioner(lfa_sim *s, const double *r, double *z, uint64_t n);
int lfa_apply_a(lfa_sim *s, const double *v, double *out, uint64_t n);
/* a16: pressure_solver::solve (src/pressure_solver.cpp:19-71); includes lfa_build_system like the reference's solve().
 * Returns LFA_W_PCG_NOT_CONVERGED if it stopped at max_iterations. */
int lfa_pcg_solve(lfa_sim *s, double dt, double *residual, uint64_t *iterations);
int lfa_download_pressure(lfa_sim *s, double *p, uint64_t n);
int lfa_upload_pressure(lfa_sim *s, const double *p, uint64_t n); /* a host callback may edit it (simulation.h:166) */
/* a17: pressure_solver::apply_pressure (src/pressure_solver.cpp:73-148). */
int lfa_apply_pressure(lfa_sim *s, double dt);
/* a18: simulation::_extrapolate_velocities (src/simulation.cpp:685-754). */
int lfa_extrapolate(lfa_sim *s);
/* a19-a20: simulation::_transfer_from_grid (src/simulation.cpp:548-560). Works on the particle order of the last
 * lfa_hash_particles: particles that lfa_correct_collide has since moved out of their tile are transferred by a gather
 * kernel (the new cell must still lie in the processed tiles, i.e. within a tile of the P2G-time particles), so no second
 * binning is needed before it (lfa_time_step relies on this). */
int lfa_g2p(lfa_sim *s);
/* a21: simulation::cfl (src/simulation.cpp:199-205); +inf when every velocity is zero. */
int lfa_cfl(lfa_sim *s, double *out);

/* One pass of the hot path, device resident, no host synchronisation except the PCG convergence polls:
 * hash -> P2G -> gravity -> build+PCG -> apply pressure -> extrapolate -> G2P  (src/simulation.cpp:62-66,72-78,
 * 83-104,119-121). residual/iterations may be NULL. */
int lfa_step_hot(lfa_sim *s, double dt, double *residual, uint64_t *iterations);

/* -- the per-step particle stages around the hot path, device resident ----------------------------
 * lfa_advect_collide : simulation::_advect_particles (src/simulation.cpp:226-249, with the sources' velocity coercion) fused with the
 *                      _detect_collisions that follows it (:612-683, grid::march_cells grid.h:140-209; from = old position)
 * lfa_correct_collide: simulation::_correct_positions (:562-610) fused with the _detect_collisions after it (:114-117);
 *                      needs lfa_hash_particles of the current positions
 * lfa_time_step      : simulation::time_step(dt) (:43-125) entirely on the device: advect+collide, hash, P2G, gravity,
 *                      pressure solve, pressure gradient, correct+collide, extrapolate, hash, G2P; sources included, no callbacks
 *                      (the host class falls back to stage calls when it needs them).
 * With a slab decomposition both move stages end with the particle migration: particles whose cell left the owned tile
 * layers are packed (68 B records) and handed to the neighbour rank, arrivals are appended; lfa_correct_collide first
 * fetches ghost copies (key + fraction) of the neighbours' adjacent tile layers, so pairs across a slab face interact. */
/* Fluid sources: simulation::sources (include/fluid/simulation.h:179; include/fluid/data_structures/source.h:12-22) as the
 * hosts fill them (testbed/main.cpp:141-165, plugins/maya/nodes/grid_node.cpp:295-303: flat int[3k] cell triples).
 *   lfa_clear_sources / lfa_add_source : replace the list (sources keep their order: a later coercing source wins a cell, a
 *                        later seeding source tops a cell up beyond an earlier one's target, exactly as the sequential
 *                        loops of src/simulation.cpp:227-238 and :756-765 do)
 *   lfa_update_sources : simulation::_update_sources (:756-765) + the hash_particles that follows it (:64): every cell of an
 *                        active source is topped up to target_density_cubic_root^3 particles (seed_cell, :136-151: uniformly
 *                        random positions inside the cell, the source's velocity, C = 0), counted by the last
 *                        lfa_hash_particles. The positions come from a counter-based generator - the reference draws from
 *                        its pcg32 member in an unspecified argument order, so parity is the particle count
 *                        per cell and every non-random field. New particles get the next ids (download order).
 * lfa_advect_collide applies the velocity coercion of _advect_particles (:227-238: velocity = the source's, C = 0 for every
 * particle inside a cell of an active coercing source) before it moves the particles; lfa_time_step runs the seeding
 * between its two binnings when a source is active. Slab decompositions: every rank is handed the whole list (like the solid
 * cells) and tops up the cells of its own tile layers; lfa_update_sources is a collective then (every rank calls it). */
int lfa_clear_sources(lfa_sim *s);
int lfa_add_source(lfa_sim *s, const int32_t *xyz, uint64_t k, const double velocity[3], uint64_t target_density_cubic_root,
                   int active, int coerce_velocity);
int lfa_update_sources(lfa_sim *s, uint64_t *n_seeded);
int lfa_advect_collide(lfa_sim *s, double dt);
int lfa_correct_collide(lfa_sim *s, double dt);
/* The same two stages with the collision handling split off, for hosts that install post_advection_callback or
 * post_correction_callback: simulation::time_step runs _advect_particles -> callback -> _detect_collisions
 * (src/simulation.cpp:50-59) and _correct_positions -> callback -> _detect_collisions (:111-117). lfa_advect / lfa_correct move
 * the particles and keep the positions of before on the device; a download in between reports them as old_position;
 * lfa_collide runs _detect_collisions (:612-683) from there to the current positions. After an upload in between lfa_collide
 * starts from the uploaded positions (from = to: the skin push-out alone). Slab decompositions: the particles change rank in
 * lfa_collide, once they have their final positions (a collective: every rank calls it). */
int lfa_advect(lfa_sim *s, double dt);
int lfa_correct(lfa_sim *s, double dt);
int lfa_collide(lfa_sim *s);
int lfa_time_step(lfa_sim *s, double dt, double *residual, uint64_t *iterations);
/* Device time of the stages of the last lfa_time_step in milliseconds (timing enabled), HIP events on the handle's stream:
 * [0] advect+collide  [1] binning  [2] P2G (scatter + finalize + gravity)  [3] P2G scatter kernel alone
 * [4] pressure system + preconditioner set-up  [5] PCG loop  [6] pressure gradient  [7] cell index of the position
 * correction (k_build_fine_index)  [8] LDS-tiled correction kernel alone  [9] correct+collide as a whole (7 + 8 + fallback)
 * [10] extrapolation  [11] G2P  [12] whole step  [13] PCG iterations of the step (a count, not a time)
 * [14] mean PCG iteration ([5] / [13])  [15] 1 if the correction ran beside the solve (lfa_set_step_overlap), else 0.
 * Overlapped, [7] [8] [9] are spans on the correction's own stream and [4] [5] [6] [10] spans on the main one: each is
 * stretched by the other side's kernels sharing the device, and they no longer add up to [12]. */
#define LFA_NUM_STEP_TIMERS 16
int lfa_get_step_timings(lfa_sim *s, double ms[LFA_NUM_STEP_TIMERS]);
/* lfa_time_step on a single domain runs the position correction (particle arrays only; simulation.cpp:99-106) on a second HIP
 * stream beside the pressure solve, the pressure gradient and the extrapolation (grid arrays only; :82-97, :119): forked after
 * the P2G, joined before the G2P. Results are identical either way; on = 0 runs the stages back to back (stage attribution).
 * Default: on. A slab decomposition overlaps the same way: the ghost-particle exchange happens on the main stream before the fork,
 * the migration after the join - no communication is issued from the correction's stream. */
int lfa_set_step_overlap(lfa_sim *s, int on);
/* The same fork / join for a host that calls the stages one by one (libfluid_amd/host/simulation.h with stage callbacks):
 * _begin enqueues lfa_correct_collide on the second stream behind everything enqueued so far and returns; until _end the
 * particle arrays and the solid mask are the correction's - grid-only calls (lfa_add_gravity, lfa_build_system, lfa_pcg_solve,
 * lfa_download/upload_pressure, lfa_apply_pressure, lfa_extrapolate, lfa_download_cells) run beside it, every other entry
 * point joins first (as if _end had been called). _end makes the main stream wait for it (no-op if nothing is in flight).
 * _undo (only between _begin and _end) joins and puts back the positions of before the correction - exactly: the correction
 * keeps its inputs - for a host whose callback asks for particles(), or changes the solid mask, at a point of the reference's
 * step order that lies before the correction (simulation.cpp:82-99); it then calls lfa_correct_collide where the reference does.
 * Single domain only (LFA_E_UNSUPPORTED on a slab decomposition). */
int lfa_correct_collide_begin(lfa_sim *s, double dt);
int lfa_correct_collide_end(lfa_sim *s);
int lfa_correct_collide_undo(lfa_sim *s);

/* -- multi-GPU: z-slab domain decomposition ---------------------------------------------------------
 * One handle per GPU/process, every handle created with the GLOBAL grid size. Rank r owns the tile layers
 * [bounds[r], bounds[r+1]) (a tile layer = 8 cells in z) and the particles inside them; one ghost tile layer on each side
 * is refreshed by nearest-neighbour exchanges (tile flags, P2G boundary planes, u/v/w/type halos, one z-slice of the PCG
 * search vector per iteration) and two scalar collectives per PCG iteration. The multigrid preconditioner runs the same
 * V-cycle as on a single domain: its finest levels (while no tile layer straddles a slab face: pick layer_bounds that are
 * multiples of 8 tile layers to keep four levels distributed) exchange one z-slice per face and smoothing step, the
 * coarser levels are replicated through one sum all-reduce of the restricted residual per V-cycle. The reference has no
 * distributed mode; this replaces nothing in it.
 *   lfa_dist_unique_id  : rank 0 fills a 128-byte RCCL id, the caller broadcasts it (e.g. torch.distributed)
 *   lfa_dist_init_rccl  : ncclCommInitRank on this handle's device; send/recv to z+-1 and all-reduce run on the
 *                         handle's stream over xGMI
 *   lfa_dist_local_*    : the same protocol between handles of ONE process (one host thread per handle), device-to-device
 *                         copies instead of RCCL: how the slab logic is tested on a single GPU ("virtual slabs")
 *   lfa_dist_init_shm   : one PROCESS per rank, messages staged through the POSIX shared-memory segment `name` ("/..."; rank 0
 *                         creates it, every rank of the job passes the same name; LFA_SHM_SLOT_MB = per-rank slot, default 32;
 *                         LFA_SHM_TIMEOUT_S = how long a rank waits for its peers before the call fails, default 60).
 *                         No RCCL and no peer access, ranks may share a GPU: the functional fallback when the communicator
 *                         cannot be created, and the way N processes are exercised on a 1-GPU box. Two PCIe crossings per
 *                         message: not a transport to quote throughput on. */
int lfa_dist_unique_id(void *id128);
int lfa_dist_init_rccl(lfa_sim *s, int rank, int nranks, const void *id128, const int32_t *layer_bounds);
typedef struct lfa_hub lfa_hub;
lfa_hub *lfa_dist_local_hub_create(int nranks);
void lfa_dist_local_hub_destroy(lfa_hub *h);
int lfa_dist_init_local(lfa_sim *s, lfa_hub *h, int rank, const int32_t *layer_bounds);
int lfa_dist_init_shm(lfa_sim *s, const char *name, int rank, int nranks, const int32_t *layer_bounds);
/* Owned tile layers of this handle ([0, ntz) without a decomposition). */
int lfa_dist_get_slab(const lfa_sim *s, int32_t *lo, int32_t *hi);
/* Marks the handle's transport as given up by the job (a PEER failed to create its communicator, a rank died): lfa_destroy then
 * releases it without waiting for the peers (ncclCommAbort). A healthy handle is not marked: its communicator is drained and
 * destroyed. Nothing in the reference (it is single-process). */
int lfa_dist_abandon(lfa_sim *s);

/* -- solid-boundary voxelizer ----------------------------------------------------------------
 * Replaces fluid::voxelizer (include/fluid/voxelizer.h:14-74, src/voxelizer.cpp:12-136) as the Maya VoxelizerNode
 * drives it (plugins/maya/nodes/voxelizer_node.cpp:255-343; fluid::obstacle, src/data_structures/obstacle.cpp:9-29, runs
 * the same sequence). Cell classification is bit-exact (fp64, the reference's operation
 * order). Host layouts: positions = double[3 nv]; indices = uint32 or uint64 triples (mesh<..., int|size_t, ...>);
 * voxel types = one byte per cell, x fastest (grid3<cell_type>); cell lists = int32 (x,y,z) triples in grid3::for_each
 * order, as the plugin passes them on to lfa_set_solid_cells (grid_node.cpp:330-339).
 *   lfa_voxels_create            : grid3<cell_type>(size, interior) at a given offset  (resize_reposition_grid*)
 *   lfa_voxels_voxelize_triangles: voxelizer::voxelize_mesh_surface (voxelizer.h:55-63)
 *   lfa_voxels_mark_exterior     : voxelizer::mark_exterior (voxelizer.cpp:83-124)
 *   lfa_voxelize_mesh            : get_bounding_box + resize_reposition_grid_constrained + the two above, i.e. the whole
 *                                  sequence of obstacle.cpp:12-18 / voxelizer_node.cpp:255-268
 *   lfa_voxels_cells             : ref_grid_size == NULL: voxel-grid coordinates of the selected types ("cells",
 *                                  voxelizer_node.cpp:285-323); otherwise reference-grid coordinates clipped to the
 *                                  reference grid ("cells_ref" :325-343). obstacle.cpp:20-28 means the same list
 *                                  (interior only) but bounds its walk over the voxel grid with a corner in
 *                                  reference-grid coordinates, which is out of range for positive offsets; that
 *                                  defect is not reproduced
 *   lfa_set_solid_from_voxels    : the same selection marked solid in a simulation grid without leaving the device */
typedef struct lfa_voxels lfa_voxels;
enum { LFA_VOX_INTERIOR = 0, LFA_VOX_EXTERIOR = 1, LFA_VOX_SURFACE = 2 }; /* voxelizer::cell_type, voxelizer.h:17-21 */
int lfa_voxels_create(lfa_voxels **out, const uint64_t size[3], const double grid_offset[3], double cell_size, int device);
void lfa_voxels_destroy(lfa_voxels *v);
const char *lfa_voxels_last_error(const lfa_voxels *v);
int lfa_voxels_info(const lfa_voxels *v, int32_t grid_min[3], uint64_t size[3], double grid_offset[3], double *cell_size);
int lfa_voxels_upload(lfa_voxels *v, const uint8_t *types);
int lfa_voxels_download(lfa_voxels *v, uint8_t *types);
int lfa_voxels_voxelize_triangles(lfa_voxels *v, const double *positions, uint64_t n_vertices, const void *indices,
                                  int index_bytes, uint64_t n_indices);
int lfa_voxels_mark_exterior(lfa_voxels *v);
int lfa_voxelize_mesh(lfa_voxels **out, const double *positions, uint64_t n_vertices, const void *indices, int index_bytes,
                      uint64_t n_indices, double cell_size, const double ref_grid_offset[3], int device);
int lfa_voxels_count(lfa_voxels *v, int include_interior, int include_surface, const int64_t *ref_grid_size, uint64_t *count);
int lfa_voxels_cells(lfa_voxels *v, int include_interior, int include_surface, const int64_t *ref_grid_size, int32_t *xyz,
                     uint64_t capacity, uint64_t *count);
int lfa_set_solid_from_voxels(lfa_sim *s, lfa_voxels *v, int include_interior, int include_surface);

/* -- surface mesher ---------------------------------------------------------------------------
 * Replaces fluid::mesher (include/fluid/mesher.h:14-46, src/mesher.cpp:320-515): implicit surface function sampled from
 * particle positions + marching cubes with shared vertices. Bit-exact: sampled values, vertex positions, vertex order
 * and index list equal the reference's (fp64, its summation order and its vertex numbering).
 * Host layouts: particle positions = double[3 n] (std::vector<vec3d>); values = double per grid point, x fastest,
 * (size + 1)^3 of them (grid3<double> _surface_function); mesh = double[3 nv] positions + uint64 indices (mesh_t).
 *   lfa_mesher_create          : mesher::resize(size) + the public fields grid_offset, cell_size, particle_extent,
 *                                cell_radius (mesher.h:28-32)
 *   lfa_mesher_sample          : mesher::_sample_surface_function(particles, r) (mesher.cpp:333-376)
 *   lfa_mesher_marching_cubes  : mesher::_marching_cubes() (mesher.cpp:400-515); the two together = generate_mesh (:325)
 *   lfa_mesher_upload_values / download_values : the sampled function, for stage-level parity tests */
/* A z-window of the grid (slab decompositions, BASELINE configs[4] on 8 GPUs): the handle stores, samples and meshes only the
 * cell layers [zlo, zhi) of the whole grid `size` - plus what it needs around them: the point planes zlo - 1 .. zhi and the
 * cells cell_radius beyond those. All arithmetic uses the whole grid's coordinates, so N windows that partition [0, size[2])
 * produce, concatenated in z order, exactly the single-grid mesh: the same vertex positions in the same order, and the same
 * index list once every window's indices have been shifted by the number of vertices of all windows below it
 * (lfa_mesher_rebase; the vertices on the plane between two windows belong to the lower one, the upper one refers to them).
 * The only thing the windows have to tell each other is a vertex count: an exclusive scan over the ranks.
 *   lfa_mesher_create_window : as lfa_mesher_create for the layers [zlo, zhi)
 *   lfa_mesher_window        : first stored point plane, number of stored planes (what lfa_mesher_download_values returns),
 *                              first and one-past-last own cell layer
 *   lfa_mesher_sample_ids    : lfa_mesher_sample with an order key per particle: inside a cell the reference visits the newest
 *                              particle first (space_hashing.h:55-62), i.e. descending index in the host's array; a rank that
 *                              holds its particles in some other order passes their global indices
 *   lfa_mesher_rebase        : adds `vertices_below` to every index of the mesh just extracted
 * lfa_mesher_sample_sim works on a handle with a slab decomposition too: the rank's own particles plus the ghost copies of its
 * neighbours' adjacent tile layers, ordered by global id (the window must not need more than those 8 cells beyond the slab). */
typedef struct lfa_mesher lfa_mesher;
int lfa_mesher_create_window(lfa_mesher **out, const uint64_t size[3], const double grid_offset[3], double cell_size,
                             double particle_extent, uint64_t cell_radius, uint64_t zlo, uint64_t zhi, int device);
int lfa_mesher_window(const lfa_mesher *m, uint64_t *z0, uint64_t *n_planes, uint64_t *own_lo, uint64_t *own_hi);
int lfa_mesher_sample_ids(lfa_mesher *m, const double *positions, const uint32_t *ids, uint64_t n, double r);
int lfa_mesher_rebase(lfa_mesher *m, uint64_t vertices_below);
int lfa_mesher_create(lfa_mesher **out, const uint64_t size[3], const double grid_offset[3], double cell_size,
                      double particle_extent, uint64_t cell_radius, int device);
void lfa_mesher_destroy(lfa_mesher *m);
const char *lfa_mesher_last_error(const lfa_mesher *m);
int lfa_mesher_sample(lfa_mesher *m, const double *positions, uint64_t n, double r);
/* the same from the particles resident in a simulation handle (upload order, the positions LFA_DL_POSITIONS reports):
 * what testbed/main.cpp:52-61,101-113 does through a host copy, without the PCIe round trip */
int lfa_mesher_sample_sim(lfa_mesher *m, lfa_sim *s, double r);
int lfa_mesher_download_values(lfa_mesher *m, double *values);
int lfa_mesher_upload_values(lfa_mesher *m, const double *values);
int lfa_mesher_marching_cubes(lfa_mesher *m, uint64_t *n_vertices, uint64_t *n_indices);
int lfa_mesher_download_mesh(lfa_mesher *m, double *positions, uint64_t *indices);

/* -- measurement --------------------------------------------------------------------------------------------- */
/* Per-stage device time of the last lfa_step_hot, measured with HIP events on the handle's stream (milliseconds):
 * [0] hash/bin [1] P2G [2] gravity [3] build system [4] PCG loop [5] apply pressure [6] extrapolate [7] G2P
 * [8] P2G scatter kernel alone [9] mean PCG iteration (PCG loop / iterations). Enabled by lfa_enable_timing(s,1). */
#define LFA_NUM_TIMERS 10
int lfa_enable_timing(lfa_sim *s, int on);
int lfa_get_timings(lfa_sim *s, double ms[LFA_NUM_TIMERS]);
/* counts of the last step: [0] particles [1] unknowns (fluid cells) [2] tiles holding particles [3] tiles processed
 * by grid kernels (dilated set) [4] padded cell count */
int lfa_get_counts(lfa_sim *s, uint64_t counts[5]);
/* The last pressure solve, per PCG iteration (no member of the reference corresponds: pressure_solver::solve,
 * src/pressure_solver.cpp:45-69, is one host loop): [0] kernel launches of one iteration [1] transport calls (neighbour
 * exchanges + all-reduces; 0 on a single domain) of one iteration [2] levels of the multigrid hierarchy (0: another
 * preconditioner) [3] first level that runs inside the single coarse-level launch [4] iterations of the solve
 * [5] transport calls of the whole solve [6] reserved (always 0: the one-launch solve of small systems left the library in
 * round 5) [7] device-side waits given up on this handle so far (0 in a healthy run; after the first one the handle keeps
 * to the launch-per-phase path, and the solve that met it was repeated there). */
#define LFA_NUM_SOLVER_STATS 8
int lfa_get_solver_stats(lfa_sim *s, uint64_t stats[LFA_NUM_SOLVER_STATS]);
/* Active tiles (8^3 cells of the level) per level of the last multigrid hierarchy, finest first; levels beyond the hierarchy
 * are 0. Diagnostic (sparse scenes: a kernel of the V-cycle pays per tile); no member of the reference corresponds. */
#define LFA_MAX_MG_LEVELS 12
int lfa_get_mg_level_tiles(lfa_sim *s, uint64_t tiles[LFA_MAX_MG_LEVELS]);
/* The last position correction: [0] half tiles handled by the LDS-tiled kernel's fallback (a thread per particle gathering from
 * global memory: crowded blocks of more than 12288 staged particles) [1] half tiles in all. Joins a correction in
 * flight. A large [0] / [1] means the scene is far denser than 8 particles per cell and the correction runs slowly.
 * _ex adds [2]: half tiles that took the tiled kernel's second pass (more than 5632 staged particles: one workgroup per CU). */
int lfa_get_correction_stats(lfa_sim *s, uint64_t stats[2]);
int lfa_get_correction_stats_ex(lfa_sim *s, uint64_t stats[3]);

/* Times `reps` back-to-back launches of one hot-path kernel on the state left by the last lfa_step_hot, with HIP
 * events on the handle's stream; returns the mean launch duration in milliseconds. For bench.py's roofline object. */
enum {
	LFA_K_SPMV_DOT = 0,    /* z = A s, dot(z,s)                         algorithmic 17 n bytes */
	LFA_K_AXPY_MAX = 1,    /* p += a s, r -= a z, max r                 algorithmic 28 n bytes */
	LFA_K_MIC_APPLY = 2,   /* z = M^-1 r (fwd+bwd), dot(z,r)            algorithmic 34 n bytes */
	LFA_K_UPDATE_S = 3,    /* s = z + b s                               algorithmic 12 n bytes */
	LFA_K_P2G_SCATTER = 4, /* particles -> per-tile (sum wv, sum w)     algorithmic 60 Np (APIC) / 24 Np bytes */
	LFA_K_P2G_FINALIZE = 5,/* normalise + type + gravity                algorithmic 14 Nc bytes (+12 Nc FLIP) */
	LFA_K_G2P = 6,         /* grid -> particles                         algorithmic 60 Np + 12 Nc (APIC) */
	LFA_K_BIN = 7,         /* tile binning (count + scatter)            algorithmic 2*68 Np + 8 Np bytes */
	LFA_K_MIC_FINE = 8,    /* the tile-level sweep kernel of LFA_K_MIC_APPLY alone (k_mic_apply) */
	LFA_K_COARSE = 9,      /* the coarse levels of the multilevel preconditioner alone (side stream in the solve) */
	LFA_K_PCG_A = 10,      /* fused: s = z + beta s, q = A s, dot(q,s)   algorithmic 12 n + 17 n bytes */
	LFA_K_PCG_B = 11,      /* fused: p += a s, r -= a q, max r, z = M^-1 r, dot(z,r)   algorithmic 28 n + 34 n bytes */
	/* LFA_PRECOND_MULTIGRID: the parts of one iteration besides LFA_K_PCG_A */
	LFA_K_MG_AXPY_PRESMOOTH = 12, /* p += a s, r -= a q, max r + red-black pre-smoothing of the finest level  28 n + 5 n */
	LFA_K_MG_DOWN0 = 13,          /* finest level: residual + restriction                                     9.5 n */
	LFA_K_MG_COARSE = 14,         /* all coarser levels: down, single-workgroup tail, up (latency bound)          */
	LFA_K_MG_UP0 = 15             /* finest level: prolongation + post-smoothing + dot(z, r)                   21 n */
};
int lfa_bench_kernel(lfa_sim *s, int which, int reps, double *mean_ms);
/* Measured HBM ceilings of this device beside the 8 TB/s spec peak (SURVEY.md 8d "report both"): a float4 grid-stride device
 * copy of `bytes` (read + write counted) and a read-only pass over the same buffer, GB/s, mean of `reps` launches. */
int lfa_bench_stream(lfa_sim *s, uint64_t bytes, int reps, double *copy_gbs, double *read_gbs);
/* Handle re-creation is cheap (fluid::simulation::resize() per Maya evaluation, plugins/maya/nodes/grid_node.cpp:256-274): device
 * blocks, streams, events and the pinned page of destroyed handles are cached process-wide and adopted by the next lfa_create /
 * allocation of the same size. lfa_pool_trim releases the cache to the driver (shutdown, memory pressure; LFA_POOL_MAX_BYTES
 * bounds it, default a quarter of the device memory). lfa_pool_stats: [0] cached bytes [1] cached blocks [2] allocations served
 * from the cache [3] allocations that went to the driver. */
void lfa_pool_trim(void);
void lfa_pool_stats(uint64_t stats[4]);
/* Which copy kernel gave the figure of the last lfa_bench_stream (a static string; "" before the first call). */
const char *lfa_bench_stream_variant(void);

#ifdef __cplusplus
}
#endif
#endif
