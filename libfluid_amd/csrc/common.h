// libfluid_amd/csrc/common.h -- device state, blocked-grid indexing and small device helpers shared by all kernels.
//
// Data layout in HBM:
//  * The MAC grid is stored TILE-MAJOR: 8x8x8 tiles (512 cells, 2 KiB per fp32 field), tile id x-fastest, cells inside a
//    tile x-fastest. A tile is the unit of work of every grid kernel (one wave owns one tile in the PCG kernels, one
//    workgroup in the transfer kernels), so every field access of a tile is one contiguous 2 KiB run.
//    The reference's x-fastest dense layout (include/fluid/data_structures/grid.h:11-12,23-32) exists only at the
//    boundary (upload/download kernels re-index).
//  * Particles are fp32 SoA, binned by tile (not by cell): key = blocked cell index, t = position inside the cell in
//    cell units (exact integer part kept in the key, so precision is 2^-24 of a cell everywhere in the domain).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/libfluid_amd.h"

// Every device allocation of the library goes through the process-wide block cache (pool.hip): the HIP names are redirected
// here, after hip_runtime.h has declared the real ones. hipFree keeps its synchronising semantics unless the caller has
// synchronised itself (lfa_pool_nosync_begin / _end around the releases of lfa_destroy).
hipError_t lfa_pool_malloc(void **p, size_t bytes);
hipError_t lfa_pool_free(void *p);
void lfa_pool_nosync_begin();
void lfa_pool_nosync_end();
template <typename T> inline hipError_t lfa_hip_malloc(T **p, size_t bytes) { return lfa_pool_malloc((void **)p, bytes); }
#define hipMalloc(p, n) lfa_hip_malloc((p), (n))
#define hipFree(p) lfa_pool_free((void *)(p))
struct lfa_stream_set;  // streams, events and the pinned page of a destroyed handle, parked for the next lfa_create (pool.hip)
lfa_stream_set *lfa_pool_take_set(int device);
void lfa_pool_park_set(lfa_stream_set *q);

#define LFA_TILE 8
#define LFA_TILE_CELLS 512
#define LFA_HALO 10            // tile + 1-cell ring
#define LFA_HALO_CELLS 1000
#define LFA_WAVE 64

// cell type bits: reference values (include/fluid/mac_grid.h:17-21) + "outside the grid" marker for padding cells.
#define CT_AIR 1
#define CT_FLUID 2
#define CT_SOLID 4
#define CT_OUTSIDE 0x80

// abits byte: bits0-2 nonsolid_neighbors, bit3/4/5 fluid_{x,y,z}pos (include/fluid/pressure_solver.h:17-26),
// bit6 = "is an unknown" (cell holds particles), bit7 = cell type is fluid.
#define AB_UNKNOWN 0x40
#define AB_FLUID 0x80

struct ParticleSoA {
	uint32_t *key = nullptr;  // blocked cell index
	float *t[3] = {nullptr, nullptr, nullptr};
	float *v[3] = {nullptr, nullptr, nullptr};
	float *c[9] = {nullptr};  // cx.xyz, cy.xyz, cz.xyz
	uint32_t *id = nullptr;   // index in the caller's array (upload order)
	void *base = nullptr;
};

#define LFA_CLOSED_MAX_UNKNOWNS 64  // a closed tile (lfa_sim::tile_closed) holds at most this many unknowns: a few dozen sweeps solve it

struct GridDims {
	int nx, ny, nz;     // real grid
	int ntx, nty, ntz;  // tiles
	int nt;             // ntx*nty*ntz
};

/// The switches of a handle, read from the environment ONCE, by lfa_create (core.hip: lfa_knobs_parse) - nothing below an entry point
/// calls getenv. None is needed in normal use: each one selects a supported configuration so that the A/Bs quoted in DESIGN.md can
/// be repeated; what was measured and dropped is in docs/experiments.md, not behind a switch.
struct lfa_knobs {
	int mg_no_persist = 0;    // LFA_MG_NO_PERSIST=1: a launch per coarse-level phase instead of k_mg_coarse (the path a handle falls back
	                          // to when a device-side wait was given up; bit-identical: the A/B of the tests)
	int mg_dist_single = 0;   // LFA_MG_DIST_SINGLE=1: the slab mode of the hierarchy with a one-rank communicator (tests, one-rank overhead)
	int mg_co_fault = 0;      // LFA_MG_CO_FAULT=n (tests): workgroup n - 1 of k_mg_coarse never raises its first flag
	int mg_no_top = 0;        // LFA_MG_NO_TOP=1: the level above k_mg_coarse's first one keeps its three launches (the A/B of fusing it
	                          // into the launch in launch order)
	int mg_no_prune = 0;      // LFA_MG_NO_PRUNE=1: every parent of a particle tile is an active level-1 tile, also one that holds no unknown
	                          // (the bitwise A/B of round 6's pruned level-1 set)
	int mg_dist_levels = 0;   // LFA_MG_DIST_LEVELS=n (slabs): at most n distributed multigrid levels instead of up to 4 - 1: only the finest level
	                          // is distributed, 4 transport calls per PCG iteration instead of 2 D + 2; the first replicated level's
	                          // right-hand side and types then travel PACKED (its active tiles only). 0: round 5's rule
	int mg_no_closed = 0;     // LFA_MG_NO_CLOSED=1: closed tiles (lfa_sim::tile_closed) stay in the PCG like every other tile (the A/B)
	int mg_no_tagged = 0;     // LFA_MG_NO_TAGGED=1: k_mg_coarse hands over through the level arrays + ready flags also with fp32 vectors
	                          // (what fp64 vectors always do; the bitwise A/B of the tagged hand-off)
};
// (process-wide, read once: LFA_PCG_GRID_CAP - workgroups of the solve's streaming kernels, core.hip; LFA_POOL_MAX_BYTES - the block
// cache, pool.hip; LFA_SHM_TIMEOUT_S / LFA_SHM_SLOT_MB - the host-staged transport, dist.hip)

struct lfa_sim {
	int device = 0;
	lfa_knobs knobs;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr;  // side stream: the coarse levels of the preconditioner run beside the fine sweep
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	// lfa_time_step runs the position correction (VALU-bound, particle arrays only) on stream3 beside the pressure solve
	// (launch- and HBM-bound, grid arrays only): fork after the P2G, join before the G2P
	hipStream_t stream3 = nullptr;
	hipEvent_t ev_cfork = nullptr, ev_cjoin = nullptr;
	bool overlap_correction = true;
	bool corr_in_flight = false;   // lfa_correct_collide_begin .. _end: the particle arrays belong to the correction on stream3
	bool counts_fresh = false;     // tile_count / rank were produced by the advection of lfa_time_step: the binning skips its pass 1
	bool move_pending = false;     // lfa_advect / lfa_correct have moved the particles and lfa_collide is due: the positions of before sit in the other buffer's key / t
	bool corr_begun = false;       // the last lfa_correct_collide_begin started a correction (false: no particles, nothing to take back)
	bool corr_undo_valid = false;  // nothing has changed positions, binning or solids since: lfa_correct_collide_undo can restore
	uint32_t *corr_ovf = nullptr;  // tiled correction: word 0 = number of flagged (overflowing) half tiles, then their bitmap
	int corr_parts_tiles = 0;      // particle tiles of the last correction
	GridDims g{};
	size_t nc = 0, ncp = 0;  // real / padded cell count
	lfa_params prm{};
	std::string err;

	// particles
	size_t np = 0, pcap = 0;
	ParticleSoA pb[2];
	int cur = 0;
	uint32_t *rank = nullptr;
	bool binned = false;

	// z-slab domain decomposition (dist.hip). Every rank indexes the GLOBAL grid; it owns the tile layers
	// [slab_lo, slab_hi) and mirrors one ghost tile layer on each side. dist == nullptr: single domain.
	struct lfa_dist *dist = nullptr;
	int slab_lo = 0, slab_hi = 0;           // owned tile layers (z)
	int slab_align = 0;                     // every interior slab bound is a multiple of 2^slab_align tile layers
	int *ptiles_all = nullptr;              // ghost-lo | owned | ghost-hi particle tiles, ascending tile id
	int p_off = 0, n_ptiles_all = 0;        // ptiles == ptiles_all + p_off
	int n_own_first = 0, n_own_last = 0;    // owned particle tiles in the first / last owned layer
	int n_ghost_lo = 0, n_ghost_hi = 0;     // particle tiles of the neighbours' adjacent layers
	int *halo_tiles = nullptr;              // [send_lo | send_hi | recv_lo | recv_hi] processed tiles of the boundary layers
	int n_halo[4] = {0, 0, 0, 0};
	void *xbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // send_lo, send_hi, recv_lo, recv_hi
	size_t xcap[4] = {0, 0, 0, 0};
	double *dist_red = nullptr;             // all-reduced scalars
	size_t np_live = 0;
	bool holes = false;  // slabs: leavers were invalidated in place since the last binning
	size_t arrivals_at = 0, n_arrivals = 0;  // slabs: particles received since the last binning sit at [arrivals_at, arrivals_at + n_arrivals)
	size_t ghost_at[2] = {0, 0}, n_ghost_particles = 0;  // ghost particles (key, t only) behind the live ones

	// tiles
	uint32_t *tile_count = nullptr, *tile_start = nullptr;  // nt, nt+1
	uint32_t *tile_flag = nullptr, *tile_scan = nullptr;    // nt
	uint32_t *grid_flag = nullptr;  // nt: dilated-set membership at the time of the last P2G (the tiles the grid is explicit on)
	int *ptiles = nullptr, *dtiles = nullptr;               // nt
	int *tile_pslot = nullptr;                               // nt : slot in ptiles or -1
	int n_ptiles = 0, n_dtiles = 0;
	uint32_t *scan_tmp = nullptr;
	size_t scan_tmp_len = 0;
	uint32_t *h_pinned = nullptr;  // small pinned scratch for device->host scalars

	// grid (blocked layout, ncp entries)
	float *u = nullptr, *v = nullptr, *w = nullptr;
	float *uo = nullptr, *vo = nullptr, *wo = nullptr;
	uint8_t *ctype = nullptr, *solid = nullptr;
	uint32_t *cell_count = nullptr;
	uint32_t *fine_start = nullptr;  // position correction: first record of every fine cell of every tile (particles.hip, FT_STRIDE per tile)
	float *stage = nullptr;  // P2G per-tile partial sums [n_ptiles][6][1000]
	size_t stage_tiles = 0;
	float *acc = nullptr;    // global-atomic P2G accumulators [6][ncp]
	double bg[3] = {0, 0, 0};  // velocity of every cell outside the processed (dilated) tile set
	bool grid_valid = false;   // P2G has run since the last upload

	// pressure system
	uint8_t *abits = nullptr;
	void *vp = nullptr, *vr = nullptr, *vz = nullptr, *vs = nullptr, *vpre = nullptr, *vq = nullptr;
	struct lfa_mg *mg = nullptr;  // multigrid hierarchy (mg.hip)
	int *nbr_table = nullptr;  // per particle-tile slot: neighbour tile ids + level-1 indices (k_build_nbr_table)
	void *vs2 = nullptr;  // second search-direction buffer of the fused iteration (k_pcg_a reads one, writes the other)
	size_t vec_elem = 0;  // element size the vectors are currently allocated for
	double *partials = nullptr;  // reduction partials
	int *pcg_state = nullptr;    // [0] done_iter  [1] nan flag
	double *pcg_hist = nullptr;  // residual per iteration
	int *level_tiles = nullptr;  // ptile slots sorted by tile level (exact MIC)
	std::vector<int> level_offsets;  // host: start of every level in level_tiles
	// coarse levels of the multilevel preconditioner (pcg.hip): level 1 = one unknown per particle tile, stored as a
	// tile-major field over the grid of tiles; level 2 = one unknown per level-1 tile, dense inverse
	GridDims g1{};
	size_t ncp1 = 0;
	float *c_diag = nullptr, *c_w[3] = {nullptr, nullptr, nullptr};
	uint8_t *c_unk = nullptr;
	void *c_pre = nullptr, *c_r = nullptr, *c_x = nullptr, *c_r2 = nullptr, *c_x2 = nullptr, *a2inv = nullptr;
	void *c_as = nullptr;     // level-1 restriction of A s (fused iteration)
	void *c_r_cur = nullptr;  // overrides c_r as the coarse right-hand side (parity buffer of the fused iteration)
	int *slot_l1 = nullptr, *l1_tiles = nullptr, *l1_l2 = nullptr;
	int n_l1tiles = 0, n2 = 0, a2cap = 0;
	size_t coarse_elem = 0;
	double a_scale = 0.0, sys_dt = -1.0;
	uint64_t n_unknowns = 0;
	bool system_valid = false, unknown_count_valid = false;
	double last_residual = 0.0;
	uint64_t last_iters = 0;
	// lfa_get_solver_stats
	// single domain, multigrid: the iteration whose residual maxima (cur_rmax_parts, one per workgroup of the AXPY kernel) the
	// V-cycle's first kernel may test (-1: none) - mg.hip: MgStop
	int cur_iter = -1, cur_rmax_n = 0;
	const double *cur_rmax_parts = nullptr;
	uint64_t stat_launches_iter = 0, stat_transport_iter = 0, stat_transport_solve = 0, stat_mg_levels = 0, stat_mg_first_co = 0;
	// kernels whose workgroups wait for each other (k_mg_coarse, k_pcg_small): a wait that was given up (mg.hip: co_wait) ends
	// their use on this handle; the solve that met it is repeated on the launch-per-phase path
	// The solver's arrays may hold non-finite values (a solve that met a NaN; lfa_bench_kernel's repeated launches): entries
	// that are no unknowns are assumed to be zero and are never rewritten, so the next system build re-creates them (pcg_scrub)
	bool pcg_poisoned = false;
	bool last_rhs_zero = false;  // the previous solve was the early-out of a zero right-hand side (pcg.hip: k_check_rhs)
	bool co_disabled = false;
	uint64_t stat_co_aborts = 0, stat_co_reason = 0;
	bool gate_counted = false;  // this handle is in the device's count of live handles (lfa_co_gate_handle)
	// warm start of the PCG (lfa_params.pcg_warm_start)
	uint32_t *tile_epoch = nullptr;   // [nt] solve counter of the last solve a tile took part in
	// [nt] 1: the tile's unknowns couple to nothing outside the tile (spray) and are few: its block of the pressure matrix is solved on
	// its own, once, and the tile stays out of the PCG's tile list (k_abits writes it for every particle tile; mg.hip, round 6)
	uint8_t *tile_closed = nullptr;
	uint32_t solve_epoch = 0;         // counter of system builds
	uint32_t pressure_epoch = 0;      // solve the pressure in vp belongs to (0: none / replaced by an upload)
	bool warm_started = false;        // the system just built starts from the previous pressure: r = b - A p is still due
	// Deferred half of the binning (APIC): lfa_hash_particles moves key, t, id (20 of the 68 bytes) and records
	// where each particle came from; v and C stay in the other buffer until the P2G has read them through that index -
	// the G2P then writes the new v, C straight into the binned order. Anything else that reads v or C calls
	// lfa_particles_materialize first.
	uint32_t *vc_src = nullptr;             // index in pb[cur ^ 1] of the particle now at i (valid while vc_pending)
	size_t vc_extent = 0;                   // records of pb[cur ^ 1] that vc_src may point at (slabs: arrivals append theirs behind)
	bool vc_pending = false;
	bool vmax2_valid = false;               // pcg_state[7] holds max |v|^2 of the particles (written by the last G2P, nothing has touched v since)
	// PIC / FLIP never change C (their P2G does not read it, their G2P does not write it; the hosts' particle records carry it
	// through): instead of moving 36 bytes per particle with every binning, C is parked in a HOME array indexed by the particle
	// id (stable; single domain) and only a download, a change to APIC or an attach to slabs brings it back (core.hip: c_home_*)
	float *c_home = nullptr;                // [9][c_home_cap]
	size_t c_home_cap = 0;
	bool c_home_valid = false;              // C lives in c_home (the c arrays of both particle buffers are scratch)
	bool vc_with_c = false;                 // C is deferred as well (APIC); PIC / FLIP move C with the particle and defer v only
	uint8_t *tile_clear = nullptr;          // [nt] no solid cell within one tile (+ [nt] scratch: tile holds a solid cell)
	unsigned clear_epoch = 0;               // solid_epoch tile_clear was computed for
	unsigned solid_epoch = 1;               // bumped whenever the solid mask changes (caches keyed on it: mg.hip)

	// fluid sources (particles.hip): the host-side list as handed in, and its flattened device form (rebuilt when it changes)
	struct SourceHost {
		std::vector<int32_t> xyz;
		double vel[3];
		uint64_t root;
		bool active, coerce;
		bool operator==(const SourceHost &o) const {
			return xyz == o.xyz && vel[0] == o.vel[0] && vel[1] == o.vel[1] && vel[2] == o.vel[2] && root == o.root &&
			       active == o.active && coerce == o.coerce;
		}
	};
	std::vector<SourceHost> sources, sources_built;
	bool sources_valid = false;       // the device arrays below describe `sources`
	uint32_t *src_cell = nullptr;     // [n_src_entries] blocked cell index of a seeding entry
	uint32_t *src_lo = nullptr;       // count the entry tops up FROM: 0xFFFFFFFF = the binning's, else the previous entry's target
	uint32_t *src_target = nullptr;   // target_density_cubic_root^3
	uint32_t *src_of = nullptr;       // source index of the entry
	uint32_t *src_need = nullptr;     // scratch: particles to create per entry, then their exclusive scan
	float *src_vel = nullptr;         // [n sources][3]
	size_t n_src_entries = 0, src_cap = 0;
	uint8_t *coerce_map = nullptr;    // [ncp] 1 + index of the LAST active coercing source that lists the cell, 0: none
	bool any_coerce = false;
	uint64_t source_epoch = 0;        // counter of seeding calls: part of the counter-based generator's key
	uint64_t next_global_id = 0;      // slabs: the id the next seeded particle of the whole job gets (ids are unique across ranks)

	// boundary scratch
	void *io_buf = nullptr;
	size_t io_cap = 0;
	uint32_t *raw_scan = nullptr;  // nc+1 : raw-order unknown numbering

	// timing
	bool timing = false;
	hipEvent_t ev[48];  // 0-9, 16-18: lfa_step_hot; 20-21: lfa_bench_kernel; 24-..: lfa_time_step (LFA_ST_* boundaries)
	bool ev_created = false;
	double ms[LFA_NUM_TIMERS] = {0};
	double ms_next[LFA_NUM_STEP_TIMERS] = {0};  // lfa_get_step_timings
};

// ---------------------------------------------------------------------------------------------------- error handling
int lfa_fail(lfa_sim *s, int code, const char *fmt, ...);
#define LFA_HIP(s, call)                                                                                   \
	do {                                                                                                    \
		hipError_t e_ = (call);                                                                             \
		if (e_ != hipSuccess)                                                                               \
			return lfa_fail((s), e_ == hipErrorOutOfMemory ? LFA_E_OOM : LFA_E_HIP, "%s failed: %s (%s:%d)", \
			                #call, hipGetErrorString(e_), __FILE__, __LINE__);                              \
	} while (0)
#define LFA_TRY(call)          \
	do {                       \
		int rc_ = (call);      \
		if (rc_ < 0) return rc_; \
	} while (0)
#define LFA_LAUNCH_CHECK(s) LFA_HIP(s, hipGetLastError())

// ---------------------------------------------------------------------------------------------------- indexing
__host__ __device__ inline uint32_t blocked_index(const GridDims &g, int x, int y, int z) {
	int tile = (x >> 3) + g.ntx * ((y >> 3) + g.nty * (z >> 3));
	return (uint32_t)tile * LFA_TILE_CELLS + (uint32_t)((x & 7) | ((y & 7) << 3) | ((z & 7) << 6));
}
__host__ __device__ inline void tile_coords(const GridDims &g, int tile, int &tx, int &ty, int &tz) {
	tx = tile % g.ntx;
	int r = tile / g.ntx;
	ty = r % g.nty;
	tz = r / g.nty;
}
__host__ __device__ inline bool in_grid(const GridDims &g, int x, int y, int z) {
	return (unsigned)x < (unsigned)g.nx && (unsigned)y < (unsigned)g.ny && (unsigned)z < (unsigned)g.nz;
}
/// Raw (reference, x-fastest) index of a blocked index; valid only for cells inside the real grid.
__host__ __device__ inline uint64_t raw_from_blocked(const GridDims &g, uint32_t b) {
	int tile = (int)(b >> 9), l = (int)(b & 511);
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
	return (uint64_t)x + (uint64_t)g.nx * ((uint64_t)y + (uint64_t)g.ny * (uint64_t)z);
}

// ---------------------------------------------------------------------------------------------------- wave helpers
/// Binning, pass 1 (core.hip: k_tile_count; particles.hip: the advection of lfa_time_step does it on the way): a wave holds
/// TC_CHUNKS x 64 consecutive particles, tile[c] = tile of the particle of chunk c in this lane (0xFFFFFFFF: none). One atomic per
/// distinct tile among them; a wave's particles of one tile get consecutive ranks in input order.
#define TC_CHUNKS 8
template <int CH>
__device__ inline void lfa_wave_tile_ranks(const uint32_t (&tile)[CH], uint32_t *tile_count, uint32_t (&my_rank)[CH]) {
	const int lane = threadIdx.x & 63;
	const unsigned long long lt = (1ull << lane) - 1ull;
	unsigned long long todo[CH];
#pragma unroll
	for (int c = 0; c < CH; ++c) {
		todo[c] = __ballot(tile[c] != 0xFFFFFFFFu);
		my_rank[c] = 0;
	}
	for (;;) {
		// the first pending particle (chunk-major) names the tile of this round
		uint32_t t = 0xFFFFFFFFu;
		bool found = false;
#pragma unroll
		for (int c = 0; c < CH; ++c)
			if (!found && todo[c]) {
				t = __shfl(tile[c], __ffsll((long long)todo[c]) - 1, 64);
				found = true;
			}
		if (!found) break;
		unsigned long long same[CH];
		uint32_t total = 0;
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			same[c] = __ballot(tile[c] == t) & todo[c];
			total += (uint32_t)__popcll(same[c]);
		}
		uint32_t base = 0;
		if (lane == 0) base = atomicAdd(&tile_count[t], total);
		base = __shfl(base, 0, 64);
#pragma unroll
		for (int c = 0; c < CH; ++c) {
			if ((same[c] >> lane) & 1ull) my_rank[c] = base + (uint32_t)__popcll(same[c] & lt);
			base += (uint32_t)__popcll(same[c]);
			todo[c] &= ~same[c];
		}
	}
}

/// s + c v for a coupling c that is exactly 0 or 1 (or -1): the product is exact, so the fused form rounds once - in the addition,
/// like `s += c * v` compiled with -ffp-contract=off. Bit-identical, one instruction instead of two; the stencil sums of the
/// smoothers, residuals and the matrix-vector product are made of these.
__device__ inline float madd01(float c, float v, float s) { return __builtin_fmaf(c, v, s); }
__device__ inline double madd01(double c, double v, double s) { return __builtin_fma(c, v, s); }
/// a b + c in one rounding (the sources are compiled with -ffp-contract=off: fusing is always written out)
__device__ inline float fma_r(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ inline double fma_r(double a, double b, double c) { return __builtin_fma(a, b, c); }
/// A thread's share of the per-workgroup partials of the previous kernel (256 threads, n <= 8 x 256 + a tail): all of its loads are
/// issued before the first is consumed - ONE round trip to the L2, where `for (i = tid; i < n; i += 256) a += p[i]` waits for each
/// load in turn (three in a row at 768 partials, in front of everything else the kernel does). Same order of additions.
__device__ inline double strided_partial_sum(const double *p, int n) {
	double v[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int i = (int)threadIdx.x + 256 * k;
		v[k] = i < n ? p[i] : 0.0;
	}
	double a = v[0];
#pragma unroll
	for (int k = 1; k < 8; ++k) a += v[k];
	for (int i = (int)threadIdx.x + 2048; i < n; i += 256) a += p[i];
	return a;
}
/// The same for a maximum; `nan` is set if any partial is a NaN.
__device__ inline double strided_partial_max(const double *p, int n, bool &nan) {
	double v[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const int i = (int)threadIdx.x + 256 * k;
		v[k] = i < n ? p[i] : -INFINITY;
	}
	double a = -INFINITY;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		nan |= v[k] != v[k];
		a = v[k] > a ? v[k] : a;
	}
	for (int i = (int)threadIdx.x + 2048; i < n; i += 256) {
		const double x = p[i];
		nan |= x != x;
		a = x > a ? x : a;
	}
	return a;
}
template <typename T> __device__ inline T wave_sum(T v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
	return v;
}
template <typename T> __device__ inline T wave_max(T v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		T other = __shfl_xor(v, o, 64);
		v = other > v ? other : v;
	}
	return v;
}

// generic exclusive scan of uint32 (scan.hip); out may alias in. Returns total in *total_dev (device) if non-null.
int lfa_exclusive_scan_u32(lfa_sim *s, const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_dev);

// stage entry points implemented per file
int lfa_particles_alloc(lfa_sim *s, size_t n);
int lfa_hash_particles_impl(lfa_sim *s, bool counts_done);  // counts_done: tile_count / rank already hold pass 1 (lfa_time_step)
int lfa_particles_materialize(lfa_sim *s);  // completes a deferred binning (no-op otherwise)
int lfa_c_home_restore(lfa_sim *s);         // C back from its home array into the current buffer (no-op unless c_home_valid)
int lfa_c_home_ensure(lfa_sim *s, size_t n);  // capacity of the home array (keeps the entries of the resident particles)
int lfa_ensure_io(lfa_sim *s, size_t bytes);
int lfa_pcg_alloc(lfa_sim *s);
int lfa_number_unknowns(lfa_sim *s);

// ---------------------------------------------------------------------------------------------------- slabs (dist.hip)
/// Transport between z-slab neighbours. lo = rank-1, hi = rank+1; all pointers are device pointers, sizes in bytes.
/// Where cell (hx, hy, hz) of a particle tile's 10 x 10 x 10 partial-sum array lives in its staging slab (P2G scatter -> finalize):
/// the 8-wide interior rows first (hx = 1..8: 32-byte rows, 800 floats), then the planes hx = 0 and hx = 9 as contiguous 10 x 10
/// planes. In a plain row-major array the x faces - one float of every 40-byte row - pull every line of the block into the two
/// x neighbours' finalize workgroups (1.95 GB fetched per launch at C4 for 0.39 GB of partial sums).
__host__ __device__ inline int lfa_stage_index(int hx, int hy, int hz) {
	return (unsigned)(hx - 1) < 8u ? (hx - 1) + 8 * (hy + 10 * hz) : 800 + (hx ? 100 : 0) + hy + 10 * hz;
}
/// the inverse: the row-major index hx + 10 hy + 100 hz of staging slot o
__host__ __device__ inline int lfa_stage_source(int o) {
	if (o < 800) return 1 + (o & 7) + 10 * (o >> 3);  // (o >> 3 = hy + 10 hz)
	const int q = o - 800, pl = q >= 100 ? 1 : 0, r = q - 100 * pl;
	return 9 * pl + 10 * r;
}

struct lfa_dist {
	int rank = 0, nranks = 1;
	int device_share = 1;  // ranks of this job that run on this handle's GPU (virtual slabs, processes sharing a device)
	// every call of the three below is one transport call of lfa_get_solver_stats (a grouped send/recv pair or one collective)
	int exchange(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
	             size_t n_send_hi, void *recv_hi, size_t n_recv_hi) {
		++calls;
		return exchange_impl(s, send_lo, n_send_lo, recv_lo, n_recv_lo, send_hi, n_send_hi, recv_hi, n_recv_hi);
	}
	int allreduce(lfa_sim *s, double *dev, int count, bool is_max) {
		++calls;
		return allreduce_impl(s, dev, count, is_max);
	}
	/// in-place all-reduce of a device array: dtype LFA_RED_U8 / F32 / F64, sum or max
	int allreduce_buf(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) {
		++calls;
		return allreduce_buf_impl(s, dev, count, dtype, is_max);
	}
	uint64_t calls = 0;
	bool broken = false;  // a transport call failed, or the job abandons this transport (lfa_dist_abandon): close without waiting for peers
	virtual int exchange_impl(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo,
	                          const void *send_hi, size_t n_send_hi, void *recv_hi, size_t n_recv_hi) = 0;
	virtual int allreduce_impl(lfa_sim *s, double *dev, int count, bool is_max) = 0;
	virtual int allreduce_buf_impl(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) = 0;
	virtual ~lfa_dist() {}
};
enum { LFA_RED_U8 = 0, LFA_RED_F32 = 1, LFA_RED_F64 = 2 };
inline bool lfa_has_lo(const lfa_sim *s) { return s->dist && s->dist->rank > 0; }
inline bool lfa_has_hi(const lfa_sim *s) { return s->dist && s->dist->rank + 1 < s->dist->nranks; }
int lfa_dist_exchange_tile_layers_u32(lfa_sim *s, uint32_t *per_tile);  // own boundary layers -> neighbours' ghost layers
int lfa_dist_build_halo_lists(lfa_sim *s);
int lfa_dist_exchange_fields(lfa_sim *s, int nfields, void *const *fields, const int *elem_bytes);
int lfa_dist_exchange_p2g_planes(lfa_sim *s, float *stage_all);
int lfa_dist_exchange_slices(lfa_sim *s, void *vec, int elem_bytes);  // elem_bytes 1, 4 or 8
/// The same for a whole tile layer of a tile-major array with `tiles_per_layer` tiles per layer (multigrid levels): slice 0 of
/// layer lo_layer goes down, slice 7 of layer hi_layer - 1 goes up; they land in layers lo_layer - 1 and hi_layer.
int lfa_dist_exchange_layer_slices(lfa_sim *s, void *vec, int elem_bytes, int tiles_per_layer, int lo_layer, int hi_layer);
int lfa_dist_allreduce(lfa_sim *s, const double *partials, int n, int slot, bool is_max);  // result in dist_red[slot]
#define LFA_DIST_ALPHA_OFF (64 + 2 * (3 * 32 + 1))  // dist_red: [0, 64) scalars | two gather buffers (<= 32 ranks) | alpha of the single-reduction CG
double *lfa_dist_gather_buf(lfa_sim *s, int parity);  // [max per rank | sum per rank] of the last lfa_dist_gather_pair
int lfa_dist_gather_pair(lfa_sim *s, const double *pmax, int n_max, const double *psum, int n_sum, int parity);
int lfa_dist_gather_triple(lfa_sim *s, const double *pmax, int n_max, const double *psum, int n_sum, const double *psum2, int n_sum2,
                           int parity);  // [max | sum | sum2 per rank]
int lfa_dist_ensure_xbuf(lfa_sim *s, int which, size_t bytes);
int lfa_dist_migrate(lfa_sim *s, bool vc_dead = false);
int lfa_dist_exchange_ghost_particles(lfa_sim *s);
int lfa_particles_reserve(lfa_sim *s, size_t n_keep, size_t n_total);
/// Orders the main stream behind a correction that lfa_correct_collide_begin has running on stream3; every entry point that
/// touches particle arrays or the solid mask calls it first (no-op otherwise).
int lfa_corr_join(lfa_sim *s);
/// The same for entry points that change positions, the binning or the solid mask: the correction can no longer be undone.
inline int lfa_corr_commit(lfa_sim *s) {
	s->corr_undo_valid = false;
	return lfa_corr_join(s);
}
#define LFA_EV_CORRECT_END 30  // ev[] slot: end of the correction (B_CORRECT of lfa_time_step)
/// Handles alive on a device (lfa_create +1, lfa_destroy -1). With more than one, the launches of kernels whose workgroups wait for
/// each other are chained through events across the handles' streams (mg.hip: CoGate): two of them resident half each would wait
/// for ever.
void lfa_co_gate_handle(int device, int delta);
int lfa_sources_sync(lfa_sim *s);  // flattens `sources` to the device arrays if they changed (particles.hip)
