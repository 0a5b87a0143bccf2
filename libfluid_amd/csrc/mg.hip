// libfluid_amd/csrc/mg.hip -- geometric multigrid V-cycle as the preconditioner of the pressure PCG
// (LFA_PRECOND_MULTIGRID). Not in the reference (its preconditioner is MIC(0), src/pressure_solver.cpp:244-332): the
// converged pressure is the same (the stopping rule of pressure_solver.cpp:54 is unchanged), the iteration count drops
// from ~180 (MIC(0), C4) to ~20 because the work per iteration no longer depends on how far information has to travel.
//
// Hierarchy: level l has cells of 2^l fine cells, stored tile-major exactly like the fine grid (8^3 tiles of that
// level's cells), so every level runs the same wave-per-tile kernels. A fine tile is one octant of its parent tile.
//   types     : a coarse cell is AIR if any child is air, else FLUID if any child is an unknown, else SOLID
//               (McAdams, Sifakis, Teran 2010: the Dirichlet surface moves inwards, never outwards)
//   operator  : the 7-point operator rediscretised on those types, in the fine level's encoding (one A byte per cell:
//               non-solid neighbour count + "positive neighbour is fluid" bits), UNSCALED (the factor dt/(rho h^2) of
//               pressure_solver.cpp:22 is applied once, to the result)
//   transfers : piecewise constant; restriction = sum of the 8 children, halved (the Galerkin operator of piecewise
//               constant interpolation is twice as stiff as the rediscretised one)
//   smoother  : red-black Gauss-Seidel (over-relaxed, MG_OMEGA) inside a tile, Jacobi across tile faces ("hybrid"); MG_INNER_SWEEPS sweeps
//               red->black on the way down from a zero guess (which makes them tile-local: no halo), as many black->red
//               on the way up with the ring values frozen. The two are adjoint, so the V-cycle is a symmetric positive
//               definite operator (tested).
// Launches per V-cycle and level: k_mg_presmooth, k_mg_residual_restrict on the way down, k_mg_prolong_postsmooth on the
// way up; the coarsest level (one tile) is solved by many sweeps inside one wave.
// (Measured and dropped: restriction fused with the pre-smoothing of the parent tile, one workgroup per parent - the
// serial chain children -> parent costs more than the launch it saves, level-0 restriction 25 -> 45 us; all coarse levels
// in one cooperative launch with grid barriers - a barrier across 8 XCDs costs what a kernel boundary costs; the
// argument-invariant middle of the cycle replayed as a hipGraph - 0.311 vs 0.301 ms per iteration with plain launches.)
// Slabs (dist.hip): the finest levels are distributed like the fine grid (own tiles, one slice per slab face exchanged where
// a stencil crosses it), the coarser ones are replicated through one sum all-reduce of the restricted residual -- the same
// V-cycle as on a single domain, so the iteration count does not depend on the decomposition (see lfa_mg::n_dist).
#include <mutex>

#include "pcg.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define MG_MAX_LEVELS 12
#define MG_NBR_STRIDE 8  // per slot: six face-neighbour tile ids (-1: inactive) + own tile id + pad
// Sweeps per smoothing step. The values across the tile faces stay frozen during a step, so the extra sweep only touches LDS:
// no HBM traffic, and a third fewer iterations (C4: 31 -> 19). A third sweep LOSES (tools/ab_sweeps.sh, round 3: 16.0 / 16.6 instead
// of 15.1 / 15.9 iterations at C3 / C4 and 7 % more time per iteration; C3 at step 550: 22.2 against 22.1 iterations).
#ifndef MG_INNER_SWEEPS
#define MG_INNER_SWEEPS 2
#endif
// Over-relaxation of the Gauss-Seidel update (red-black SOR as the smoother; forward and backward sweeps stay adjoint).
// Measured iterations at C2 / C3 / C4: 1.0: 15 / 18 / 19, 1.08: 14 / 16 / 17, 1.15: - / 15 / 16, 1.2: 13 / 15 / 16, 1.3: - / 17 / 17, 1.5: - / 29 / 31.
#define MG_OMEGA 1.15
// Also measured (C4 / C3, with omega 1.15 = 16 / 15 iterations): 3 sweeps per step on the levels below the finest: 15 / 14 but
// 8 % more time per iteration; restriction factor 0.4 / 0.45 / 0.6 instead of 0.5: 19 / 16 / 23 iterations at C4.

struct lfa_mg_level {
	GridDims g{};          // cells of this level
	size_t ncp = 0;        // padded cells
	int n_tiles = 0;       // active tiles (hold unknowns or children that do)
	int *tiles = nullptr;  // device, ascending tile id
	int *nbr = nullptr;    // device, MG_NBR_STRIDE ints per slot
	uint8_t *ctype = nullptr;  // device, whole padded grid: MT_SOLID / MT_FLUID / MT_AIR (levels >= 1)
	uint8_t *abits = nullptr;  // device, whole padded grid                            (levels >= 1; level 0: s->abits)
	// device, whole padded grid: right-hand side, pre-smoothed iterate, final iterate (the up-kernel reads the ring of x while
	// neighbouring waves store y)                                                     (levels >= 1; level 0: vr, vq, vz)
	void *b = nullptr, *x = nullptr, *y = nullptr;
	size_t cap_tiles = 0;
	int lo_layer = 0, hi_layer = 0;  // owned tile layers of this level (slabs: distributed levels only)
	uint32_t *flag = nullptr, *prev_flag = nullptr;  // device, per tile of the level: active now / at the last set-up (single domain)
	unsigned *ready = nullptr;  // device, 3 x tiles of the level: ready flags of k_mg_coarse (levels >= 1; never cleared: tags are unique)
	// device, 3 x whole padded grid of 8-byte words {launch tag, fp32 value}: what the workgroups of ONE k_mg_coarse launch hand to
	// each other - the pre-smoothed iterate, the right-hand side (children's shares), the result (round 5: the tagged hand-off;
	// allocated when the level first runs inside the launch, fp32 vectors only)
	unsigned long long *xq = nullptr;
	int xq_sections = 0;  // 3: [x | b | y] (a level inside the launch), 1: x alone (the level fused in front of it)
};
struct lfa_mg {
	int n_levels = 0;
	lfa_mg_level lv[MG_MAX_LEVELS];
	size_t elem = 0;
	std::vector<int> host_tiles;  // particle tiles the tile lists / neighbour tables on the device were built for
	// Slabs: levels 0 .. n_dist - 1 are distributed like the fine grid (every rank runs its own tiles, one slice per slab face
	// and level is exchanged); the levels from n_dist on are small and replicated: every rank holds the whole level, the
	// restricted residual is combined by a sum all-reduce and each rank runs the identical remaining V-cycle.
	int n_dist = 0;
	std::vector<int> host_top;    // active tiles of level n_dist (all ranks') the replicated lists were built for
	// LFA_MG_DIST_LEVELS (round 6): the first replicated level's right-hand side (every iteration) and types (every set-up) cross the
	// ranks PACKED - its active tiles only, resp. those and their face neighbours - instead of as whole padded arrays (level 1 of C4:
	// 2 400 of 32 768 tiles; 4.9 instead of 67 MB per iteration)
	bool top_packed = false;
	std::vector<int> host_ring;   // active tiles of level n_dist and their face neighbours, ascending
	int *top_ring = nullptr;      // device copy
	size_t top_ring_cap = 0;
	void *top_buf = nullptr;      // device, the packed buffer
	size_t top_buf_cap = 0;
	// Round 6: closed tiles (lfa_sim::tile_closed - spray whose unknowns couple to nothing outside their tile) are no tiles of level 0:
	// the PCG iterates over lv[0].tiles, a compacted list, and finds a neighbour's slot in slot0 (device, per tile of the grid, -1:
	// not in the list). Off (null, lv[0].tiles = the binning's list): slabs, LFA_MG_NO_CLOSED=1, a grid of one tile.
	int *slot0 = nullptr;
	bool closed_out = false;  // the last set-up left closed tiles out
	uint8_t *l1_dirty = nullptr;  // device, per level-1 tile: a child tile was flagged at the last set-up (k_mg_types_from_fine_dirty)
	// device, per level-1 tile: the tile holds an unknown of level 1 (k_mg_types_from_fine_dirty). Round 6: a level-1 tile WITHOUT
	// one - the parent of spray, of a film thinner than a coarse cell - is no longer active (single domain): it computed zeros.
	// Late in the C3 run 805 instead of 2 996 level-1 tiles (and 95 .. 200 instead of 458 on level 2) are left; bit-identical.
	uint8_t *l1_has_fluid = nullptr;
	unsigned solid_epoch = 0;     // solid mask the level-1 types were computed for
	uint32_t *counts = nullptr;   // device, active tiles per level (single-domain set-up, read back once)
	unsigned co_tag = 0;          // k_mg_coarse: launch counter = the value its ready flags are raised to (lfa_mg_level::ready)
	int launches_per_cycle = 0;   // launches of the last V-cycle incl. the AXPY / pre-smoothing kernel (lfa_get_solver_stats)
	int first_co = 0;             // first level inside k_mg_coarse at the last V-cycle (0: launch-per-phase path)
};
// A level stays distributed while no tile layer straddles a slab face, and its ghost types follow from the one fine ghost tile
// layer a rank mirrors (8 cells = one slice of level 3).
#define MG_DIST_MAX 4
#define MG_CO_MAX_LEVELS 8  // levels inside k_mg_coarse (a 2048^3 grid has 9 levels in all)

namespace {
#define MG_FENCE()                                             \
	do {                                                       \
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
		__builtin_amdgcn_wave_barrier();                       \
	} while (0)

// ordered so that the coarsening rule is a maximum (slabs combine the types of a replicated level by a max all-reduce)
enum { MT_SOLID = 0, MT_FLUID = 1, MT_AIR = 2 };

// ------------------------------------------------------------------------------------------------ set-up kernels
/// Level-0 cell type for the coarsening rule, from the simulation's own arrays (stage_halo_types / is_unknown_at of
/// grid_ops.hip): tiles that hold particles have explicit types, every other cell is solid or air by the solid mask.
__device__ inline int fine_type(const uint32_t *tile_flag, const uint32_t *cell_count, const uint8_t *ctype, const uint8_t *solid,
                                uint32_t b) {
	if (!tile_flag[b >> 9]) return solid[b] ? MT_SOLID : MT_AIR;
	// a solid cell that holds particles is an unknown on the finest level (src/simulation.cpp:83-94) but takes no pressure
	// from its lower neighbours and gives none to them: for the coarse levels it is a wall
	if ((ctype[b] & 7) == CT_SOLID) return MT_SOLID;
	return cell_count[b] > 0 ? MT_FLUID : MT_AIR;
}

/// Types of level 1 from the fine grid, one thread per coarse cell c0 .. c0 + count - 1 of the padded coarse grid. Children
/// with z outside [zlo, zhi) (fine cells) count as walls: the neutral element of the rule (slabs, replicated level).
__global__ void k_mg_types_from_fine(GridDims gf, GridDims gc, size_t c0, size_t count, int zlo, int zhi, const uint32_t *tile_flag,
                                     const uint32_t *cell_count, const uint8_t *ctype, const uint8_t *solid, uint8_t *out) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	const size_t c = c0 + i;
	const int tile = (int)(c >> 9), l = (int)(c & 511);
	int tx, ty, tz;
	tile_coords(gc, tile, tx, ty, tz);
	const int X = tx * 8 + (l & 7), Y = ty * 8 + ((l >> 3) & 7), Z = tz * 8 + (l >> 6);
	bool any_air = false, any_fluid = false;
	for (int k = 0; k < 8; ++k) {
		const int x = 2 * X + (k & 1), y = 2 * Y + ((k >> 1) & 1), z = 2 * Z + (k >> 2);
		int t = MT_SOLID;  // outside the grid: a wall (mac_grid.cpp:26-31)
		if (in_grid(gf, x, y, z) && z >= zlo && z < zhi) t = fine_type(tile_flag, cell_count, ctype, solid, blocked_index(gf, x, y, z));
		any_air |= t == MT_AIR;
		any_fluid |= t == MT_FLUID;
	}
	out[c] = (uint8_t)(!in_grid(gc, X, Y, Z) ? MT_SOLID : (any_air ? MT_AIR : (any_fluid ? MT_FLUID : MT_SOLID)));
}
/// The same, single domain, one workgroup per level-1 tile, skipping the tiles that cannot have changed: outside the processed
/// (dilated) fine tiles a cell is wall or air by the solid mask alone, so a level-1 tile whose 8 child tiles are unflagged now
/// and were unflagged at the last set-up keeps its types (force: first set-up, or the solid mask changed).
/// At C4 2 600 of 32 768 tiles are recomputed per step: 217 -> 45 us.
__global__ void __launch_bounds__(256)
k_mg_types_from_fine_dirty(GridDims gf, GridDims gc, const uint32_t *tile_flag, const uint32_t *cell_count, const uint8_t *ctype,
                           const uint8_t *solid, uint8_t *out, uint8_t *was_dirty, int force, uint8_t *has_fluid) {
	for (int tile = blockIdx.x; tile < gc.nt; tile += gridDim.x) {
		int tx, ty, tz;
		tile_coords(gc, tile, tx, ty, tz);
		int mine = 0;
		if (threadIdx.x < 8) {
			const int cx = 2 * tx + (threadIdx.x & 1), cy = 2 * ty + ((threadIdx.x >> 1) & 1), cz = 2 * tz + (threadIdx.x >> 2);
			if (cx < gf.ntx && cy < gf.nty && cz < gf.ntz) mine = tile_flag[cx + gf.ntx * (cy + gf.nty * cz)] != 0;
		}
		const int dirty = __syncthreads_or(mine);
		if (!force && !dirty && !was_dirty[tile]) continue;  // uniform per workgroup
		int unknowns = 0;
		for (int l = threadIdx.x; l < 512; l += 256) {
			const int X = tx * 8 + (l & 7), Y = ty * 8 + ((l >> 3) & 7), Z = tz * 8 + (l >> 6);
			bool any_air = false, any_fluid = false;
			for (int k = 0; k < 8; ++k) {
				const int x = 2 * X + (k & 1), y = 2 * Y + ((k >> 1) & 1), z = 2 * Z + (k >> 2);
				int t = MT_SOLID;
				if (in_grid(gf, x, y, z)) t = fine_type(tile_flag, cell_count, ctype, solid, blocked_index(gf, x, y, z));
				any_air |= t == MT_AIR;
				any_fluid |= t == MT_FLUID;
			}
			const int ty_ = !in_grid(gc, X, Y, Z) ? MT_SOLID : (any_air ? MT_AIR : (any_fluid ? MT_FLUID : MT_SOLID));
			out[(size_t)tile * 512 + l] = (uint8_t)ty_;
			unknowns |= ty_ == MT_FLUID;
		}
		// (the barrier: was_dirty[tile] was read by every thread above; a tile that is skipped keeps its types, hence its has_fluid)
		const int any_unknown = __syncthreads_or(unknowns);
		if (threadIdx.x == 0) {
			was_dirty[tile] = (uint8_t)dirty;
			has_fluid[tile] = (uint8_t)(any_unknown != 0);
		}
	}
}
/// Types of level l + 1 from level l.
__global__ void k_mg_types_coarsen(GridDims gf, GridDims gc, size_t c0, size_t count, int zlo, int zhi, const uint8_t *tf, uint8_t *out) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= count) return;
	const size_t c = c0 + i;
	const int tile = (int)(c >> 9), l = (int)(c & 511);
	int tx, ty, tz;
	tile_coords(gc, tile, tx, ty, tz);
	const int X = tx * 8 + (l & 7), Y = ty * 8 + ((l >> 3) & 7), Z = tz * 8 + (l >> 6);
	bool any_air = false, any_fluid = false;
	for (int k = 0; k < 8; ++k) {
		const int x = 2 * X + (k & 1), y = 2 * Y + ((k >> 1) & 1), z = 2 * Z + (k >> 2);
		const int t = (in_grid(gf, x, y, z) && z >= zlo && z < zhi) ? tf[blocked_index(gf, x, y, z)] : MT_SOLID;
		any_air |= t == MT_AIR;
		any_fluid |= t == MT_FLUID;
	}
	out[c] = (uint8_t)(!in_grid(gc, X, Y, Z) ? MT_SOLID : (any_air ? MT_AIR : (any_fluid ? MT_FLUID : MT_SOLID)));
}
/// A bytes of a coarse level from its types (the encoding of k_abits, grid_ops.hip), for the active tiles.
__global__ void __launch_bounds__(256) k_mg_abits(const int *tiles, int n_tiles, GridDims g, const uint8_t *t, uint8_t *abits) {
	for (int slot = blockIdx.x; slot < n_tiles; slot += gridDim.x) {
		const int tile = tiles[slot];
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		for (int l = threadIdx.x; l < 512; l += 256) {
			const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
			const size_t b = (size_t)tile * 512 + l;
			uint8_t a = 0;
			if (t[b] == MT_FLUID) {
				auto ty_at = [&](int xx, int yy, int zz) -> int { return in_grid(g, xx, yy, zz) ? t[blocked_index(g, xx, yy, zz)] : MT_SOLID; };
				const int xp = ty_at(x + 1, y, z), yp = ty_at(x, y + 1, z), zp = ty_at(x, y, z + 1);
				const int ns = (xp != MT_SOLID) + (yp != MT_SOLID) + (zp != MT_SOLID) + (ty_at(x - 1, y, z) != MT_SOLID) +
				               (ty_at(x, y - 1, z) != MT_SOLID) + (ty_at(x, y, z - 1) != MT_SOLID);
				a = (uint8_t)(ns | ((xp == MT_FLUID) << 3) | ((yp == MT_FLUID) << 4) | ((zp == MT_FLUID) << 5) | AB_UNKNOWN | AB_FLUID);
			}
			abits[b] = a;
		}
	}
}
// ---- single domain: active-tile lists and neighbour tables of every level without a host round trip per level
/// flag0[t] = tile t holds particles (slot table of the binning)
/// (`closed`, may be null: ... and is not a closed tile, lfa_sim::tile_closed)
__global__ void k_mg_flag_level0(const int *tile_pslot, const uint8_t *closed, uint32_t *flag, int nt) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nt) flag[t] = (tile_pslot[t] >= 0 && !(closed && closed[t])) ? 1u : 0u;
}
/// k_mg_compact that also leaves every tile's slot in the list (-1: not in it)
__global__ void k_mg_compact_slots(const uint32_t *flag, const uint32_t *scan, int *list, int *slot, int nt) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nt) return;
	if (flag[t]) list[scan[t]] = t;
	slot[t] = flag[t] ? (int)scan[t] : -1;
}
/// a tile is active if one of its (up to 8) child tiles is
/// (`has_unknown`, level 1 of a single domain: ... and holds an unknown of its own - lfa_mg::l1_has_fluid)
__global__ void k_mg_flag_parents(GridDims gf, GridDims gc, const uint32_t *ff, uint32_t *fc, const uint8_t *has_unknown) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= gc.nt) return;
	int tx, ty, tz;
	tile_coords(gc, t, tx, ty, tz);
	uint32_t any = 0;
	for (int k = 0; k < 8; ++k) {
		const int cx = 2 * tx + (k & 1), cy = 2 * ty + ((k >> 1) & 1), cz = 2 * tz + (k >> 2);
		if (cx < gf.ntx && cy < gf.nty && cz < gf.ntz) any |= ff[cx + gf.ntx * (cy + gf.nty * cz)];
	}
	if (has_unknown && !has_unknown[t]) any = 0;
	fc[t] = any ? 1u : 0u;
}
__global__ void k_mg_compact(const uint32_t *flag, const uint32_t *scan, int *list, int nt) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nt && flag[t]) list[scan[t]] = t;
}
/// six face neighbours (tile id, -1: inactive or outside) + own id per slot
/// + in word 7 the mask of the active child tiles (level below; `flag_f` null on the finest level): k_mg_coarse waits for them
__device__ inline void mg_build_nbr_slot(int i, const int *tiles, const GridDims &g, const uint32_t *flag, int *nbr, const GridDims &gf,
                                         const uint32_t *flag_f) {
	const int t = tiles[i], sy = g.ntx, sz = g.ntx * g.nty;
	int tx, ty, tz;
	tile_coords(g, t, tx, ty, tz);
	const int cand[6] = {tx > 0 ? t - 1 : -1, tx + 1 < g.ntx ? t + 1 : -1, ty > 0 ? t - sy : -1,
	                     ty + 1 < g.nty ? t + sy : -1, tz > 0 ? t - sz : -1, tz + 1 < g.ntz ? t + sz : -1};
#pragma unroll
	for (int k = 0; k < 6; ++k) nbr[i * MG_NBR_STRIDE + k] = (cand[k] >= 0 && flag[cand[k]]) ? cand[k] : -1;
	nbr[i * MG_NBR_STRIDE + 6] = t;
	int mask = 0;
	if (flag_f)
		for (int k = 0; k < 8; ++k) {
			const int cx = 2 * tx + (k & 1), cy = 2 * ty + ((k >> 1) & 1), cz = 2 * tz + (k >> 2);
			if (cx < gf.ntx && cy < gf.nty && cz < gf.ntz && flag_f[cx + gf.ntx * (cy + gf.nty * cz)]) mask |= 1 << k;
		}
	nbr[i * MG_NBR_STRIDE + 7] = mask;
}
__global__ void k_mg_build_nbr(const int *tiles, int n_tiles, GridDims g, const uint32_t *flag, int *nbr, GridDims gf, const uint32_t *flag_f) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_tiles) mg_build_nbr_slot(i, tiles, g, flag, nbr, gf, flag_f);
}

/// The levels whose whole tile grid is at most MG_SMALL_NT tiles (C4: levels 2-6, C2: levels 1-4) get their tile lists, neighbour
/// tables and clears from three launches in all instead of seven per level - at these sizes every one of those launches is its
/// 5 us of launch latency and nothing else (35 of them, 0.15 ms per step at C4, before).
#define MG_SMALL_NT 4096
struct MgSmall {
	int first, last;                   // levels first .. last (first >= 1)
	GridDims g[MG_MAX_LEVELS];         // g[first - 1] is read too
	uint32_t *flag[MG_MAX_LEVELS];     // the set being built; flag[first - 1] = the children's set of level `first` (an earlier launch)
	const uint32_t *prev[MG_MAX_LEVELS];  // the set of the last set-up
	int *tiles[MG_MAX_LEVELS], *nbr[MG_MAX_LEVELS];
	void *x[MG_MAX_LEVELS], *b[MG_MAX_LEVELS], *y[MG_MAX_LEVELS];
	uint8_t *abits[MG_MAX_LEVELS];
	const uint8_t *l1_has_unknown;     // level 1 (when it is among these levels): only tiles that hold an unknown are active; null: all
};
/// flags from the children's flags, exclusive scan, ascending list, count: level after level, one workgroup
__global__ void __launch_bounds__(1024) k_mg_small_lists(MgSmall P, uint32_t *counts) {
	__shared__ uint32_t fl[2][MG_SMALL_NT];  // the flags of the level being built and of the one below it
	__shared__ uint32_t wsum[16];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	int cur = 0;
	for (int l = P.first; l <= P.last; ++l, cur ^= 1) {
		const GridDims gf = P.g[l - 1], gc = P.g[l];
		const uint32_t *ff = P.flag[l - 1];
		const bool below_in_lds = l > P.first;
		uint32_t v[4], sum = 0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int t = 4 * (int)threadIdx.x + k;
			uint32_t any = 0;
			if (t < gc.nt) {
				int tx, ty, tz;
				tile_coords(gc, t, tx, ty, tz);
				for (int c = 0; c < 8; ++c) {
					const int cx = 2 * tx + (c & 1), cy = 2 * ty + ((c >> 1) & 1), cz = 2 * tz + (c >> 2);
					if (cx < gf.ntx && cy < gf.nty && cz < gf.ntz) {
						const int ct = cx + gf.ntx * (cy + gf.nty * cz);
						any |= below_in_lds ? fl[cur ^ 1][ct] : ff[ct];
					}
				}
				any = any ? 1u : 0u;
				if (l == 1 && P.l1_has_unknown && !P.l1_has_unknown[t]) any = 0u;
				fl[cur][t] = any;
				P.flag[l][t] = any;
			}
			v[k] = any;
			sum += any;
		}
		uint32_t incl = sum;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t u = __shfl_up(incl, o, 64);
			if (lane >= o) incl += u;
		}
		if (lane == 63) wsum[wid] = incl;
		__syncthreads();
		uint32_t ex = incl - sum, total = 0;
		for (int w = 0; w < 16; ++w) {
			if (w < wid) ex += wsum[w];
			total += wsum[w];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k)
			if (v[k]) P.tiles[l][ex++] = 4 * (int)threadIdx.x + k;
		if (threadIdx.x == 0) counts[l] = total;
		__syncthreads();  // (wsum is reused; fl[cur] is complete before the next level reads it)
	}
}
/// neighbour tables (blockIdx.y = level - first) and the clears a departed tile needs, from the counts on the device
template <typename real> __global__ void __launch_bounds__(256) k_mg_small_tables(MgSmall P, const uint32_t *counts) {
	const int l = P.first + (int)blockIdx.y;
	const int n = (int)counts[l];
	// (the children's current set: flag[l - 1] for a small level below, what the caller put there for level first - 1)
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
		mg_build_nbr_slot(i, P.tiles[l], P.g[l], P.flag[l], P.nbr[l], P.g[l - 1], P.flag[l - 1]);
	// departed tiles (in the last set, not in this one: rare): every lane looks at one tile, the wave clears the ones found together
	const uint32_t *prev = P.prev[l], *cur = P.flag[l];
	real *x = (real *)P.x[l], *b = (real *)P.b[l], *y = (real *)P.y[l];
	const int lane = threadIdx.x & 63, nt = P.g[l].nt;
	for (int t0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63; t0 < nt; t0 += gridDim.x * blockDim.x) {
		const int t = t0 + lane;
		// (departed OR arrived: while a level-1 tile without unknowns is inactive its children still restrict into its right-hand side)
		unsigned long long gone = __ballot(t < nt && (prev[t] != 0) != (cur[t] != 0));
		while (gone) {
			const int k = __ffsll((long long)gone) - 1;
			gone &= gone - 1;
			const size_t base = (size_t)(t0 + k) * 512;
			for (int c = lane; c < 512; c += 64) {
				x[base + c] = (real)0; b[base + c] = (real)0; y[base + c] = (real)0; P.abits[l][base + c] = 0;
			}
		}
	}
}
/// Vectors of levels >= 1 are read where no tile of this solve writes (parents of ring cells): a tile that has left the active
/// set must not leave values behind. One workgroup per tile that was active at the last set-up and is not now.
template <typename real>
__global__ void __launch_bounds__(256)
k_mg_clear_departed(const uint32_t *prev, const uint32_t *cur, int nt, real *x, real *b, real *y, uint8_t *abits) {
	for (int t = blockIdx.x; t < nt; t += gridDim.x) {
		if ((prev[t] != 0) == (cur[t] != 0)) continue;  // (departed or arrived, see k_mg_small_tables)
		for (int l = threadIdx.x; l < 512; l += 256) {
			const size_t c = (size_t)t * 512 + l;
			x[c] = (real)0; b[c] = (real)0; y[c] = (real)0; abits[c] = 0;
		}
	}
}

/// buf[k][512] <-> field[tiles[k]][512] (the packed exchange of the first replicated level, lfa_mg::top_packed), and the clear
template <typename T, bool PACK> __global__ void __launch_bounds__(256) k_mg_tiles_copy(const int *tiles, int n, T *field, T *buf) {
	for (int k = blockIdx.x; k < n; k += gridDim.x)
		for (int c = threadIdx.x; c < 512; c += 256) {
			if (PACK) buf[(size_t)k * 512 + c] = field[(size_t)tiles[k] * 512 + c];
			else field[(size_t)tiles[k] * 512 + c] = buf[(size_t)k * 512 + c];
		}
}
template <typename T> __global__ void __launch_bounds__(256) k_mg_tiles_zero(const int *tiles, int n, T *field) {
	for (int k = blockIdx.x; k < n; k += gridDim.x)
		for (int c = threadIdx.x; c < 512; c += 256) field[(size_t)tiles[k] * 512 + c] = (T)0;
}

// ------------------------------------------------------------------------------------------------ V-cycle kernels
/// What a kernel needs to know about one level.
template <typename real> struct MgLv {
	const int *tiles, *nbr;
	int n_tiles;
	GridDims g;
	const uint8_t *abits;
	real *b, *x, *y;
};

/// pick ? b : a on the bit patterns (exact, and immune to being rewritten into an indexed load - or into branches).
__device__ inline float bit_select(float a, float b, int pick) {
	const uint32_t x = __builtin_bit_cast(uint32_t, a), y = __builtin_bit_cast(uint32_t, b);
	return __builtin_bit_cast(float, x ^ ((x ^ y) & (0u - (uint32_t)pick)));
}
__device__ inline double bit_select(double a, double b, int pick) {
	const uint64_t x = __builtin_bit_cast(uint64_t, a), y = __builtin_bit_cast(uint64_t, b);
	return __builtin_bit_cast(double, x ^ ((x ^ y) & (0ull - (uint64_t)pick)));
}
/// 1 / (number of non-solid neighbours), 1..6: a table lookup instead of an IEEE division in the smoother - as three levels of bit
/// selects on the bits of n. (Written as `n == 6 ? 1/6 : n == 5 ? ...` hipcc turns the chain into nested exec-mask BRANCHES, ~40
/// scalar instructions and eight s_cbranch per Gauss-Seidel update: a third of the smoothing kernels' instruction stream.)
template <typename real> __device__ inline real rcp_diag(uint32_t n) {
	const int b0 = (int)(n & 1u), b1 = (int)((n >> 1) & 1u), b2 = (int)((n >> 2) & 1u);
	const real t1 = bit_select((real)0.5, (real)(1.0 / 3.0), b0);   // n = 2, 3
	const real t2 = bit_select((real)0.25, (real)0.2, b0);          // n = 4, 5
	const real u0 = bit_select((real)1, t1, b1);                    // n = (0,) 1 | 2, 3
	const real u1 = bit_select(t2, (real)(1.0 / 6.0), b1);          // n = 4, 5 | 6 (, 7)
	return bit_select(u0, u1, b2);
}


/// How a kernel reaches the level arrays other workgroups write. MemPlain: ordinary accesses (one launch per phase, the kernel
/// boundary makes them visible). MemAgent: agent-scope relaxed atomics (`sc1` on gfx950: the access is coherent across the eight
/// per-XCD L2s by itself, no bulk write-back / invalidate) - what lets k_mg_coarse run its phases inside ONE launch.
struct MemPlain {
	template <typename T> static __device__ inline T ld(const T *p) { return *p; }
	template <typename T> static __device__ inline void st(T *p, T v) { *p = v; }
};
struct MemAgent {
	template <typename T> static __device__ inline T ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
	template <typename T> static __device__ inline void st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
/// One colour of the Gauss-Seidel update on the column of a lane: x_i = (b_i + sum of coupled neighbours) / diag_i.
/// `h` is the 10^3 halo block of the wave (current values, ring = values of the neighbour tiles or 0).
/// What one Gauss-Seidel update needs of a cell's A byte: the couplings as 0 / 1 factors and the over-relaxed update
///   x <- x + omega (sum / diag - x)  =  (1 - omega) x + (omega / diag) sum
/// as two coefficients. A cell that is no unknown (or has no non-solid neighbour) gets (keep, gain) = (1, 0): the update leaves
/// it as it is, WITHOUT a branch - so the four updates of a colour are one basic block whose 28 LDS reads are issued together
/// instead of four exec-masked blocks that each wait for their own.
template <typename real> struct GsCoef {
	real lo, xp, yp, zp, keep, gain;
};
template <typename real> __device__ inline GsCoef<real> gs_coef(uint32_t a) {
	const int on = ((a & AB_UNKNOWN) && (a & 7)) ? 1 : 0;
	GsCoef<real> c;
	c.lo = (a & AB_FLUID) ? (real)1 : (real)0;
	c.xp = (real)((a >> 3) & 1);
	c.yp = (real)((a >> 4) & 1);
	c.zp = (real)((a >> 5) & 1);
	c.keep = bit_select((real)1, (real)(1.0 - MG_OMEGA), on);
	c.gain = bit_select((real)0, (real)MG_OMEGA * rcp_diag<real>(a & 7), on);
	return c;
}
template <typename real> __device__ inline real gs_update(const GsCoef<real> &c, real bv, real xm, real ym, real zm, real xp, real yp, real zp, real x) {
	real sum = bv;
	sum = madd01(c.lo, xm, sum);
	sum = madd01(c.lo, ym, sum);
	sum = madd01(c.lo, zm, sum);
	sum = madd01(c.xp, xp, sum);
	sum = madd01(c.yp, yp, sum);
	sum = madd01(c.zp, zp, sum);
	return fma_r(sum, c.gain, c.keep * x);
}
/// One colour of the Gauss-Seidel update on the column of a lane: x_i = (b_i + sum of coupled neighbours) / diag_i.
/// `h` is the 10^3 halo block of the wave (current values, ring = values of the neighbour tiles or 0).
template <typename real>
__device__ inline void gs_colour(real *h, const uint32_t (&ab)[8], const real (&bb)[8], int lx, int ly, int colour) {
	// the cells of one colour in the column of lane (x, y) are z = 2 j + z0, z0 = (x + y + colour) & 1: four updates with every
	// lane active (looping over z and skipping the other colour would idle half the wave)
	const int z0 = (lx + ly + colour) & 1;
	const uint32_t zmask = 0u - (uint32_t)z0;
	real out[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		// (a select written as arithmetic: `z0 ? ab[2j+1] : ab[2j]` is turned into a dynamically indexed load from scratch)
		const uint32_t a = ab[2 * j] ^ ((ab[2 * j] ^ ab[2 * j + 1]) & zmask);
		const real bv = bit_select(bb[2 * j], bb[2 * j + 1], z0);
		const int i = (lx + 1) + 10 * (ly + 1) + 100 * (2 * j + z0 + 1);
		// (the compiler keeps the decoded coefficients of all eight cells in registers across the sweeps: 3 waves per SIMD for the
		// finest-level kernels; decoding per update instead - an opaque `asm` on `a` - measured no better, docs/experiments.md)
		out[j] = gs_update<real>(gs_coef<real>(a), bv, h[i - 1], h[i - 10], h[i - 100], h[i + 1], h[i + 10], h[i + 100], h[i]);
	}
	// (cells of one colour do not read each other: the four results are stored behind all the reads)
#pragma unroll
	for (int j = 0; j < 4; ++j) h[(lx + 1) + 10 * (ly + 1) + 100 * (2 * j + z0 + 1)] = out[j];
}

/// Down, one tile: x = one red->black sweep on A x = b from x = 0 (`h`: ring already zero). With a zero guess the values
/// across the tile faces do not enter, so the tile is smoothed on its own. Returns nothing; the column ends in `h`.
template <typename real>
__device__ inline void presmooth_column(real *h, const uint32_t (&ab)[8], const real (&bb)[8], int lx, int ly, int inner) {
	MG_FENCE();
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = (real)0;
	MG_FENCE();
	for (int it = 0; it < inner; ++it) {
		gs_colour<real>(h, ab, bb, lx, ly, 0);
		MG_FENCE();
		gs_colour<real>(h, ab, bb, lx, ly, 1);
		MG_FENCE();
	}
}
/// (LD / ST: how the right-hand side is read and the iterate written - MemAgent where another workgroup of the SAME launch wrote or
/// will read them: k_mg_coarse)
template <typename real, typename LD = MemPlain, typename ST = MemPlain>
__device__ inline void presmooth_tile(const MgLv<real> &L, int slot, real *h, int lane, int inner) {
	const int lx = lane & 7, ly = lane >> 3;
	const size_t base = (size_t)L.tiles[slot] * 512;
	uint32_t ab[8];
	real bb[8];
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) {
		ab[zz] = L.abits[base + zz * 64 + lane];
		bb[zz] = LD::ld(L.b + base + zz * 64 + lane);
	}
	presmooth_column<real>(h, ab, bb, lx, ly, inner);
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) ST::st(L.x + base + zz * 64 + lane, h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)]);
}

/// Halo block of a tile-major vector (k_spmv's layout): interior + the six faces of the neighbour tiles (0 if inactive).
template <typename real, typename LD = MemPlain>
__device__ inline void load_halo(real *h, const real *v, size_t base, const int *nb, int lane, int lx, int ly) {
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = LD::ld(v + base + zz * 64 + lane);
	h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = nb[0] >= 0 ? LD::ld(v + (size_t)nb[0] * 512 + ly * 64 + lx * 8 + 7) : (real)0;
	h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = nb[1] >= 0 ? LD::ld(v + (size_t)nb[1] * 512 + ly * 64 + lx * 8 + 0) : (real)0;
	h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = nb[2] >= 0 ? LD::ld(v + (size_t)nb[2] * 512 + ly * 64 + 7 * 8 + lx) : (real)0;
	h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = nb[3] >= 0 ? LD::ld(v + (size_t)nb[3] * 512 + ly * 64 + 0 * 8 + lx) : (real)0;
	h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = nb[4] >= 0 ? LD::ld(v + (size_t)nb[4] * 512 + 7 * 64 + lane) : (real)0;
	h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = nb[5] >= 0 ? LD::ld(v + (size_t)nb[5] * 512 + 0 * 64 + lane) : (real)0;
}

/// Down, one tile: residual r = b - A x of the level and its restriction to the next: half the sum over the 8 children.
/// (`lds_parent`: the restricted values go to the parent tile's 512-entry block in LDS instead - k_mg_restrict0_pre1)
template <typename real, typename LD = MemPlain, typename ST = MemPlain>
__device__ inline void residual_restrict_tile(const MgLv<real> &L, const GridDims &gc, real *b_coarse, int slot, real *h, int lane,
                                              real *lds_parent = nullptr) {
	const int lx = lane & 7, ly = lane >> 3;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int nb[6] = {nt[0], nt[1], nt[2], nt[3], nt[4], nt[5]}, tile = nt[6];
	const size_t base = (size_t)tile * 512;
	// A bytes and right-hand side of the whole column up front, unconditionally (b is 0 where the cell is no unknown): a load
	// inside the per-cell branch costs one HBM round trip per z
	uint32_t ab[8];
	real bb[8];
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) {
		ab[zz] = L.abits[base + zz * 64 + lane];
		bb[zz] = LD::ld(L.b + base + zz * 64 + lane);
	}
	MG_FENCE();
	load_halo<real, LD>(h, L.x, base, nb, lane, lx, ly);
	MG_FENCE();
	real pair[4];
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) {
		const int i = (lx + 1) + 10 * (ly + 1) + 100 * (zz + 1);
		const uint32_t a = ab[zz];
		real r = (real)0;
		if (a & AB_UNKNOWN) {
			const real F = (a & AB_FLUID) ? (real)1 : (real)0;
			real val = (real)(a & 7) * h[i];
			val = madd01(-F, h[i - 1], val);
			val = madd01(-F, h[i - 10], val);
			val = madd01(-F, h[i - 100], val);
			val = madd01(-(real)((a >> 3) & 1), h[i + 1], val);
			val = madd01(-(real)((a >> 4) & 1), h[i + 10], val);
			val = madd01(-(real)((a >> 5) & 1), h[i + 100], val);
			r = bb[zz] - val;
		}
		if (zz & 1) pair[zz >> 1] += r;
		else pair[zz >> 1] = r;
	}
	int tx, ty, tz;
	tile_coords(L.g, tile, tx, ty, tz);
	const int ptile = (tx >> 1) + gc.ntx * ((ty >> 1) + gc.nty * (tz >> 1));
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		real t = pair[j];
		t += __shfl_xor(t, 1, 64);
		t += __shfl_xor(t, 8, 64);
		if (!(lx & 1) && !(ly & 1)) {
			const int o = ((tz & 1) * 4 + j) * 64 + ((ty & 1) * 4 + (ly >> 1)) * 8 + (tx & 1) * 4 + (lx >> 1);
			if (lds_parent) lds_parent[o] = (real)0.5 * t;
			else ST::st(b_coarse + (size_t)ptile * 512 + o, (real)0.5 * t);
		}
	}
}

/// Up, one tile: x += P e (piecewise constant prolongation of the next level's solution), then one black->red sweep with
/// the corrected values of the neighbour tiles on the ring; the column ends in `h`, `bb` returns the right-hand side.
template <typename real>
__device__ inline void prolong_postsmooth_tile(const MgLv<real> &L, const GridDims &gc, const real *e, int slot, real *h, int lane,
                                               real (&bb)[8], int inner) {
	const int lx = lane & 7, ly = lane >> 3;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int nb[6] = {nt[0], nt[1], nt[2], nt[3], nt[4], nt[5]}, tile = nt[6];
	const size_t base = (size_t)tile * 512;
	int tx, ty, tz;
	tile_coords(L.g, tile, tx, ty, tz);
	// correction of a cell given by its coordinates on this level (any tile): value of its parent cell
	auto corr = [&](int X, int Y, int Z) -> real { return e ? e[blocked_index(gc, X >> 1, Y >> 1, Z >> 1)] : (real)0; };
	uint32_t ab[8];
	MG_FENCE();
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) {
		ab[zz] = L.abits[base + zz * 64 + lane];
		bb[zz] = L.b[base + zz * 64 + lane];
		real v = L.x[base + zz * 64 + lane];
		if (ab[zz] & AB_UNKNOWN) v += corr(tx * 8 + lx, ty * 8 + ly, tz * 8 + zz);
		h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = v;
	}
	// ring: face lanes = (a, b) over the two in-face axes; a ring cell is corrected like its own tile corrects it
	auto ring = [&](int nbk, int cell, int X, int Y, int Z) -> real {
		if (nbk < 0) return (real)0;
		const size_t j = (size_t)nbk * 512 + cell;
		real v = L.x[j];
		if (L.abits[j] & AB_UNKNOWN) v += corr(X, Y, Z);
		return v;
	};
	h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = ring(nb[0], ly * 64 + lx * 8 + 7, tx * 8 - 1, ty * 8 + lx, tz * 8 + ly);
	h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = ring(nb[1], ly * 64 + lx * 8 + 0, tx * 8 + 8, ty * 8 + lx, tz * 8 + ly);
	h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = ring(nb[2], ly * 64 + 7 * 8 + lx, tx * 8 + lx, ty * 8 - 1, tz * 8 + ly);
	h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = ring(nb[3], ly * 64 + 0 * 8 + lx, tx * 8 + lx, ty * 8 + 8, tz * 8 + ly);
	h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = ring(nb[4], 7 * 64 + lane, tx * 8 + lx, ty * 8 + ly, tz * 8 - 1);
	h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = ring(nb[5], 0 * 64 + lane, tx * 8 + lx, ty * 8 + ly, tz * 8 + 8);
	MG_FENCE();
	for (int it = 0; it < inner; ++it) {
		gs_colour<real>(h, ab, bb, lx, ly, 1);
		MG_FENCE();
		gs_colour<real>(h, ab, bb, lx, ly, 0);
		MG_FENCE();
	}
}

/// Coarsest level (a single tile), one wave: NSW red->black sweeps followed by NSW black->red sweeps from zero -- a
/// symmetric operator that is as good as exact for the few unknowns left. `h`: ring already zero.
template <typename real> __device__ inline void coarsest_tile(const MgLv<real> &L, int nsw, real *h, int lane) {
	const int lx = lane & 7, ly = lane >> 3;
	const size_t base = (size_t)L.tiles[0] * 512;
	uint32_t ab[8];
	real bb[8];
	MG_FENCE();
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) {
		ab[zz] = L.abits[base + zz * 64 + lane];
		bb[zz] = L.b[base + zz * 64 + lane];
		h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = (real)0;
	}
	MG_FENCE();
	for (int k = 0; k < 2 * nsw; ++k) {
		const int first = k < nsw ? 0 : 1;
		gs_colour<real>(h, ab, bb, lx, ly, first);
		MG_FENCE();
		gs_colour<real>(h, ab, bb, lx, ly, first ^ 1);
		MG_FENCE();
	}
#pragma unroll
	for (int zz = 0; zz < 8; ++zz) L.y[base + zz * 64 + lane] = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)];
}

template <typename real>
__global__ void __launch_bounds__(256) k_mg_presmooth(MgLv<real> L, const int *state) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	if (state[0] >= 0) return;
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	real *h = halo[wid];
	for (int i = lane; i < LFA_HALO_CELLS; i += 64) h[i] = (real)0;  // the ring stays zero
	for (int slot = blockIdx.x * PCG_WAVES + wid; slot < L.n_tiles; slot += gridDim.x * PCG_WAVES) presmooth_tile<real>(L, slot, h, lane, MG_INNER_SWEEPS);
}

/// The AXPYs of the iteration (k_axpy_max) fused with the pre-smoothing of the finest level: p += alpha s, r -= alpha q,
/// signed max r, then x0 = red->black sweep on the new residual, tile by tile (x0 overwrites q: same tile, same wave).
template <typename real, int MW>
__global__ void __launch_bounds__(256, MW)
k_mg_axpy_presmooth(const int *tiles, int n_tiles, const uint8_t *abits, real *p, const real *sdir, real *r, real *q_x,
                    const double *part_sigma, int n_sigma, const double *part_qs, int n_qs, double *part_rmax, const int *state) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double lds[16];
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	double m = -INFINITY;
	bool nan = false;
	if (state[0] < 0) {
		// software pipeline: the loads of the wave's next tile are in flight during the sweeps of the current one - and those of
		// its first tile while the scalars of the previous kernel are reduced. The tile ids of the wave's next 64 slots sit in
		// the lanes of one register (one load, v_readlane per tile): a load of the id in front of each tile's loads is a second,
		// dependent round trip per tile.
		const int stride = gridDim.x * PCG_WAVES;
		int slot = blockIdx.x * PCG_WAVES + wid, turn = 0;
		auto load_ids = [&](int first) -> int {
			const long long sl = (long long)first + (long long)lane * stride;
			return sl < n_tiles ? tiles[sl] : 0;
		};
		int ids = load_ids(slot);
		uint32_t tab[8];
		real tp[8], ts[8], tr[8], tq[8];
		size_t base = 0;
		auto load_tile = [&](int tile) {
			base = (size_t)tile * 512;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t c = base + zz * 64 + lane;
				tab[zz] = abits[c]; tp[zz] = p[c]; ts[zz] = sdir[c]; tr[zz] = r[c]; tq[zz] = q_x[c];
			}
		};
		if (slot < n_tiles) load_tile(__builtin_amdgcn_readlane(ids, 0));
		double a = strided_partial_sum(part_sigma, n_sigma), b = strided_partial_sum(part_qs, n_qs);
		a = wave_sum(a);
		b = wave_sum(b);
		if (lane == 0) { lds[wid] = a; lds[4 + wid] = b; }
		__syncthreads();
		const real alpha = (real)(((lds[0] + lds[1]) + (lds[2] + lds[3])) / ((lds[4] + lds[5]) + (lds[6] + lds[7])));
		__syncthreads();
		real *h = halo[wid];
		for (int i = lane; i < LFA_HALO_CELLS; i += 64) h[i] = (real)0;
		while (slot < n_tiles) {
			uint32_t ab[8];
			real bb[8];
			const size_t obase = base;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t c = obase + zz * 64 + lane;
				ab[zz] = tab[zz];
				real rn = (real)0;
				if (ab[zz] & AB_UNKNOWN) {
					p[c] = tp[zz] + alpha * ts[zz];
					rn = tr[zz] + (-alpha) * tq[zz];
					r[c] = rn;
					nan |= rn != rn;
					m = (double)rn > m ? (double)rn : m;
				}
				bb[zz] = rn;
			}
			slot += stride;
			turn = (turn + 1) & 63;
			if (turn == 0 && slot < n_tiles) ids = load_ids(slot);  // (a wave with more than 64 tiles: rare)
			if (slot < n_tiles) load_tile(__builtin_amdgcn_readlane(ids, turn));
			presmooth_column<real>(h, ab, bb, lx, ly, MG_INNER_SWEEPS);
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) q_x[obase + zz * 64 + lane] = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)];
		}
	}
	m = wave_max(m);
	nan = __any(nan);
	__syncthreads();
	if (lane == 0) lds[8 + wid] = nan ? NAN : m;
	__syncthreads();
	if (threadIdx.x == 0) {
		double v = lds[8];
		for (int i = 1; i < 4; ++i) v = (v != v || lds[8 + i] != lds[8 + i]) ? NAN : (lds[8 + i] > v ? lds[8 + i] : v);
		part_rmax[blockIdx.x] = v;
	}
}

/// The same stage for the single-reduction CG of slab runs (Chronopoulos / Gear). With w = A z (k_pcg_a on the V-cycle's result) the
/// search direction and its image follow from recurrences, dir = z + beta dir, adir = w + beta adir, and
/// alpha = gamma / (delta - beta gamma / alpha_prev) needs gamma = z.r and delta = w.z only: the three scalars of an iteration -
/// gamma, delta and the signed max of r - travel in ONE collective (lfa_dist_gather_triple) where the textbook form needs dot(q, s)
/// in a second one. Same iterates in exact arithmetic. The lists hold one value per rank (or one reduced value); the stopping rule
/// of pressure_solver::solve (src/pressure_solver.cpp:54-58) on the residual of iteration iter - 1 is evaluated here, the way
/// k_pcg_a does it for the textbook form. alpha_io[iter & 1] = alpha of the previous iteration, alpha_io[~iter & 1] is written.
template <typename real>
__global__ void __launch_bounds__(256, 2)
k_mg_axpy_presmooth_cg(const int *tiles, int n_tiles, const uint8_t *abits, real *p, real *dir, real *adir, const real *z, real *r, real *w_x,
                       const double *gamma, int n_gamma, const double *gamma_old, int n_gamma_old, const double *delta, int n_delta,
                       const double *rmax_prev, int n_rmax, double tol, int iter, double *alpha_io, double *part_rmax, int *state,
                       double *hist) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double lds[8];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	double m = -INFINITY;
	bool nan = false;
	bool run = state[0] < 0;
	if (run && iter > 0) {  // every workgroup evaluates the rule on the same values, workgroup 0 records it
		double rm = rmax_prev[0];
		for (int i = 1; i < n_rmax; ++i) rm = (rm != rm || rmax_prev[i] != rmax_prev[i]) ? NAN : (rmax_prev[i] > rm ? rmax_prev[i] : rm);
		const bool stop = rm != rm || rm < tol;
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			hist[iter - 1] = rm;
			if (stop) {
				if (rm != rm) state[1] = 1;
				*(double *)(state + 16) = rm;
				state[0] = iter;
			}
		}
		run = !stop;
	}
	if (run) {
		double g = 0.0, go = 0.0, d = 0.0;
		for (int i = 0; i < n_gamma; ++i) g += gamma[i];
		for (int i = 0; i < n_delta; ++i) d += delta[i];
		double beta_d = 0.0, alpha_d = g / d;
		if (iter > 0) {
			for (int i = 0; i < n_gamma_old; ++i) go += gamma_old[i];
			beta_d = g / go;
			alpha_d = g / (d - beta_d * g / alpha_io[iter & 1]);
		}
		if (blockIdx.x == 0 && threadIdx.x == 0) alpha_io[(iter & 1) ^ 1] = alpha_d;
		const real alpha = (real)alpha_d, beta = (real)beta_d;
		const bool first = iter == 0;  // (dir, adir hold whatever the last solve left)
		real *h = halo[wid];
		for (int i = lane; i < LFA_HALO_CELLS; i += 64) h[i] = (real)0;
		const int stride = gridDim.x * PCG_WAVES;
		int slot = blockIdx.x * PCG_WAVES + wid;
		uint32_t tab[8];
		real tp[8], td[8], ta[8], tz[8], tr[8], tw[8];
		size_t base = 0;
		auto load_tile = [&](int sl) {
			base = (size_t)tiles[sl] * 512;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t c = base + zz * 64 + lane;
				tab[zz] = abits[c]; tp[zz] = p[c]; tz[zz] = z[c]; tr[zz] = r[c]; tw[zz] = w_x[c];
				td[zz] = first ? (real)0 : dir[c];
				ta[zz] = first ? (real)0 : adir[c];
			}
		};
		if (slot < n_tiles) load_tile(slot);
		while (slot < n_tiles) {
			uint32_t ab[8];
			real bb[8];
			const size_t obase = base;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t c = obase + zz * 64 + lane;
				ab[zz] = tab[zz];
				real rn = (real)0;
				if (ab[zz] & AB_UNKNOWN) {
					const real dn = first ? tz[zz] : tz[zz] + beta * td[zz];
					const real an = first ? tw[zz] : tw[zz] + beta * ta[zz];
					dir[c] = dn;
					adir[c] = an;
					p[c] = tp[zz] + alpha * dn;
					rn = tr[zz] + (-alpha) * an;
					r[c] = rn;
					nan |= rn != rn;
					m = (double)rn > m ? (double)rn : m;
				}
				bb[zz] = rn;
			}
			slot += stride;
			if (slot < n_tiles) load_tile(slot);
			presmooth_column<real>(h, ab, bb, lx, ly, MG_INNER_SWEEPS);
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) w_x[obase + zz * 64 + lane] = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)];
		}
	}
	m = wave_max(m);
	nan = __any(nan);
	__syncthreads();
	if (lane == 0) lds[wid] = nan ? NAN : m;
	__syncthreads();
	if (threadIdx.x == 0) {
		double v = lds[0];
		for (int i = 1; i < 4; ++i) v = (v != v || lds[i] != lds[i]) ? NAN : (lds[i] > v ? lds[i] : v);
		part_rmax[blockIdx.x] = v;
	}
}

/// The stopping rule of pressure_solver::solve on the residual the AXPY kernel in front of this one has just produced
/// (src/pressure_solver.cpp:52-57: the reference leaves its loop BEFORE applying the preconditioner to a converged residual).
/// Without it the rule is evaluated by the NEXT iteration's k_pcg_a, i.e. after a whole V-cycle whose result nobody reads: 0.19 of an
/// iteration's 0.25 ms at C4, once per solve. Every workgroup reduces the same per-workgroup maxima (as k_pcg_a does), workgroup 0
/// records the verdict exactly as k_pcg_a / k_check_converged would (hist[iter], state[0] = iter + 1); the kernels behind this one
/// see state[0] and return.
struct MgStop {
	const double *part_rmax;  // null: no test here (slabs: the maximum is another collective; levels >= 1)
	int n_part, iter;
	double tol;
	int *state;
	double *hist;
};
template <typename real>
__global__ void __launch_bounds__(256) k_mg_residual_restrict(MgLv<real> L, GridDims gc, real *b_coarse, const int *state, MgStop stop) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double red[4];
	if (state[0] >= 0) return;
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *h = halo[wid];
	const int stride = gridDim.x * PCG_WAVES;
	int slot = blockIdx.x * PCG_WAVES + wid;
	int nrow = 0, cur_tile = 0;  // (lane k: entry k & 7 of the table row of the slot after the next, see k_mg_prolong_postsmooth)
	auto load_row = [&](int sl) {
		if (sl < L.n_tiles) nrow = L.nbr[(size_t)sl * MG_NBR_STRIDE + (lane & 7)];
	};
	uint32_t tab[8];
	real tb[8], tx[8], rx[6];
	bool rv[6];
	auto load_tile = [&](const int row) {
		const int nt[6] = {__builtin_amdgcn_readlane(row, 0), __builtin_amdgcn_readlane(row, 1), __builtin_amdgcn_readlane(row, 2),
		                   __builtin_amdgcn_readlane(row, 3), __builtin_amdgcn_readlane(row, 4), __builtin_amdgcn_readlane(row, 5)};
		cur_tile = __builtin_amdgcn_readlane(row, 6);
		const size_t base = (size_t)cur_tile * 512;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			tab[zz] = L.abits[base + zz * 64 + lane];
			tb[zz] = L.b[base + zz * 64 + lane];
			tx[zz] = L.x[base + zz * 64 + lane];
		}
		// ring: face lanes = (a, b) over the two in-face axes; a missing neighbour reads the tile's own cell (masked when consumed)
		const int cell[6] = {ly * 64 + lx * 8 + 7, ly * 64 + lx * 8, ly * 64 + 56 + lx, ly * 64 + lx, 7 * 64 + lane, lane};
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			rv[k] = nt[k] >= 0;
			rx[k] = L.x[(size_t)(rv[k] ? nt[k] : cur_tile) * 512 + cell[k]];
		}
	};
	load_row(slot);
	if (slot < L.n_tiles) load_tile(nrow);
	load_row(slot + stride);
	if (stop.part_rmax) {
		bool nan = false;
		double a = strided_partial_max(stop.part_rmax, stop.n_part, nan);
		a = wave_max(a);
		nan = __any(nan);
		if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nan ? NAN : a;
		__syncthreads();
		double rmax = red[0];
		for (int k = 1; k < 4; ++k) rmax = (rmax != rmax || red[k] != red[k]) ? NAN : (red[k] > rmax ? red[k] : rmax);
		const bool done = rmax != rmax || rmax < stop.tol;
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			stop.hist[stop.iter] = rmax;
			if (done) {
				if (rmax != rmax) stop.state[1] = 1;
				*(double *)(stop.state + 16) = rmax;
				stop.state[0] = stop.iter + 1;
			}
		}
		if (done) return;
	}
	// residual_restrict_tile as a software pipeline with branch-free loads (k_mg_prolong_postsmooth): the table row a tile ahead,
	// every load of the next tile in flight while the current one is worked on - and those of the first during the test above.
	while (slot < L.n_tiles) {
		uint32_t ab[8];
		real bb[8];
		const int tile = cur_tile;
		MG_FENCE();
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			ab[zz] = tab[zz];
			bb[zz] = tb[zz];
			h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = tx[zz];
		}
		h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = rv[0] ? rx[0] : (real)0;
		h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = rv[1] ? rx[1] : (real)0;
		h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = rv[2] ? rx[2] : (real)0;
		h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = rv[3] ? rx[3] : (real)0;
		h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = rv[4] ? rx[4] : (real)0;
		h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = rv[5] ? rx[5] : (real)0;
		slot += stride;
		{
			const int crow = nrow;
			load_row(slot + stride);
			if (slot < L.n_tiles) load_tile(crow);
		}
		MG_FENCE();
		real pair[4];
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const int i = (lx + 1) + 10 * (ly + 1) + 100 * (zz + 1);
			const uint32_t a = ab[zz];
			real r = (real)0;
			if (a & AB_UNKNOWN) {
				const real F = (a & AB_FLUID) ? (real)1 : (real)0;
				real val = (real)(a & 7) * h[i];
				val = madd01(-F, h[i - 1], val);
				val = madd01(-F, h[i - 10], val);
				val = madd01(-F, h[i - 100], val);
				val = madd01(-(real)((a >> 3) & 1), h[i + 1], val);
				val = madd01(-(real)((a >> 4) & 1), h[i + 10], val);
				val = madd01(-(real)((a >> 5) & 1), h[i + 100], val);
				r = bb[zz] - val;
			}
			if (zz & 1) pair[zz >> 1] += r;
			else pair[zz >> 1] = r;
		}
		int tx_, ty_, tz_;
		tile_coords(L.g, tile, tx_, ty_, tz_);
		const int ptile = (tx_ >> 1) + gc.ntx * ((ty_ >> 1) + gc.nty * (tz_ >> 1));
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			real t = pair[j];
			t += __shfl_xor(t, 1, 64);
			t += __shfl_xor(t, 8, 64);
			if (!(lx & 1) && !(ly & 1)) {
				const int o = ((tz_ & 1) * 4 + j) * 64 + ((ty_ & 1) * 4 + (ly >> 1)) * 8 + (tx_ & 1) * 4 + (lx >> 1);
				b_coarse[(size_t)ptile * 512 + o] = (real)0.5 * t;
			}
		}
	}
}

/// LEVEL0: the result is scaled by 1/scale and dot(z, r) is formed.
template <typename real, bool LEVEL0, int MW>
__global__ void __launch_bounds__(256, MW)
k_mg_prolong_postsmooth(MgLv<real> L, GridDims gc, const real *e, real inv_scale, double *part_sigma, const int *state) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double red[4];
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *h = halo[wid];
	double acc = 0.0;
	const bool run = state[0] < 0;
	if (run) {
		// The body of prolong_postsmooth_tile as a software pipeline: every load of the wave's next tile (column, ring, the
		// parents' corrections) is in flight while the current tile is swept in LDS.
		const int stride = gridDim.x * PCG_WAVES;
		int slot = blockIdx.x * PCG_WAVES + wid;
		uint32_t tab[8], rab[6];
		real tb[8], tx[8], tc[4], rx[6], rc[6];
		bool rv[6];
		size_t base = 0;
		// The table row (six neighbour tiles, own tile) of the slot AFTER the next: requested a whole tile ahead, so that the loads of
		// the next tile never wait for their addresses.
		// (lane k holds entry k & 7 - a per-lane value, which the compiler cannot move into scalar registers - and wait for - the
		// moment it is loaded; the entries are taken out with v_readlane when the row is used)
		int nrow = 0;
		auto load_row = [&](int sl) {
			if (sl < L.n_tiles) nrow = L.nbr[(size_t)sl * MG_NBR_STRIDE + (lane & 7)];
		};
		auto load_tile = [&](const int row) {
			// Branch-free: every address follows from the table row and no load sits behind a condition, so all loads of the tile
			// are in flight together. (With the six entries read one by one and the ring's corrections behind `neighbour
			// present ?` the compiler had put an `s_waitcnt vmcnt(0)` behind each of seven dependent loads: seven serialized
			// round trips per tile.)
			const int nt[6] = {__builtin_amdgcn_readlane(row, 0), __builtin_amdgcn_readlane(row, 1), __builtin_amdgcn_readlane(row, 2),
			                   __builtin_amdgcn_readlane(row, 3), __builtin_amdgcn_readlane(row, 4), __builtin_amdgcn_readlane(row, 5)};
			const int tile = __builtin_amdgcn_readlane(row, 6);
			base = (size_t)tile * 512;
			int tx_, ty_, tz_;
			tile_coords(L.g, tile, tx_, ty_, tz_);
			auto corr = [&](int X, int Y, int Z) -> real {
				if (!e) return (real)0;  // (uniform)
				return e[blocked_index(gc, X >> 1, Y >> 1, Z >> 1)];
			};
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				tab[zz] = L.abits[base + zz * 64 + lane];
				tb[zz] = L.b[base + zz * 64 + lane];
				tx[zz] = L.x[base + zz * 64 + lane];
			}
#pragma unroll
			for (int j = 0; j < 4; ++j) tc[j] = corr(tx_ * 8 + lx, ty_ * 8 + ly, tz_ * 8 + 2 * j);
			// ring: face lanes = (a, b) over the two in-face axes; a missing neighbour reads the tile's own cell (masked later)
			const int cell[6] = {ly * 64 + lx * 8 + 7, ly * 64 + lx * 8, ly * 64 + 56 + lx, ly * 64 + lx, 7 * 64 + lane, lane};
			const int RX[6] = {tx_ * 8 - 1, tx_ * 8 + 8, tx_ * 8 + lx, tx_ * 8 + lx, tx_ * 8 + lx, tx_ * 8 + lx};
			const int RY[6] = {ty_ * 8 + lx, ty_ * 8 + lx, ty_ * 8 - 1, ty_ * 8 + 8, ty_ * 8 + ly, ty_ * 8 + ly};
			const int RZ[6] = {tz_ * 8 + ly, tz_ * 8 + ly, tz_ * 8 + ly, tz_ * 8 + ly, tz_ * 8 - 1, tz_ * 8 + 8};
#pragma unroll
			for (int k = 0; k < 6; ++k) {
				rv[k] = nt[k] >= 0;
				const size_t j = (size_t)(rv[k] ? nt[k] : tile) * 512 + cell[k];
				rab[k] = L.abits[j];
				rx[k] = L.x[j];
				// (no neighbour: the parent of the tile's own face cell, 8 cells back along the face axis - in range, masked later)
				const int back = rv[k] ? 0 : ((k & 1) ? -8 : 8);
				rc[k] = corr(RX[k] + (k < 2 ? back : 0), RY[k] + ((k >> 1) == 1 ? back : 0), RZ[k] + (k >= 4 ? back : 0));
			}
		};
		load_row(slot);
		if (slot < L.n_tiles) load_tile(nrow);
		load_row(slot + stride);
		while (slot < L.n_tiles) {
			uint32_t ab[8];
			real bb[8];
			const size_t obase = base;
			MG_FENCE();
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				ab[zz] = tab[zz];
				bb[zz] = tb[zz];
				real v = tx[zz];
				if (ab[zz] & AB_UNKNOWN) v += tc[zz >> 1];
				h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = v;
			}
			real fv[6];
#pragma unroll
			for (int k = 0; k < 6; ++k) {
				real v = rx[k];
				if (rab[k] & AB_UNKNOWN) v += rc[k];
				fv[k] = rv[k] ? v : (real)0;
			}
			h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = fv[0];
			h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = fv[1];
			h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = fv[2];
			h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = fv[3];
			h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = fv[4];
			h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = fv[5];
			slot += stride;
			{
				const int crow = nrow;  // (the row of the next tile: here since the last iteration)
				load_row(slot + stride);
				if (slot < L.n_tiles) load_tile(crow);
			}
			MG_FENCE();
			for (int it = 0; it < MG_INNER_SWEEPS; ++it) {
				gs_colour<real>(h, ab, bb, lx, ly, 1);
				MG_FENCE();
				gs_colour<real>(h, ab, bb, lx, ly, 0);
				MG_FENCE();
			}
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const real v = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)];
				if (LEVEL0) {
					const real zv = v * inv_scale;
					L.y[obase + zz * 64 + lane] = zv;
					acc += (double)zv * (double)bb[zz];
				} else {
					L.y[obase + zz * 64 + lane] = v;
				}
			}
		}
	}
	if (LEVEL0) {
		acc = wave_sum(acc);
		if (lane == 0) red[wid] = acc;
		__syncthreads();
		if (threadIdx.x == 0) part_sigma[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
	}
}

/// gs_colour for ONE cell (halo index i): the same expression, term by term.
template <typename real> __device__ inline void gs_cell(real *H, uint32_t a, real bv, int i) {
	H[i] = gs_update<real>(gs_coef<real>(a), bv, H[i - 1], H[i - 10], H[i - 100], H[i + 1], H[i + 10], H[i + 100], H[i]);
}
// ------------------------------------------------------------------------------------------------ cell-parallel variants
// The levels below the finest hold 1/8, 1/64, ... of the tiles: with one wave per tile (4 cells per lane and colour) a
// kernel of theirs is a handful of waves, each alone on its SIMD, waiting out the latency of ~140 dependent instructions per
// half sweep (s_memtime: 2300 cycles). Here a WORKGROUP takes a tile: a thread per cell of a colour in the half sweeps, two
// cells per thread elsewhere, a barrier between the steps. Same operations in the same order per cell: bit-identical to
// k_mg_presmooth / k_mg_residual_restrict / k_mg_prolong_postsmooth (LFA_MG_NO_CP=1 selects those; tested).
template <typename real> struct CpTile {
	real H[LFA_HALO_CELLS];
	real bb[512];
	uint8_t ab[512];
};
template <typename real> __device__ inline void cp_half_sweep(CpTile<real> &S, int colour) {
	const int t = threadIdx.x, qx = t & 7, qy = (t >> 3) & 7, z = 2 * (t >> 6) + ((qx + qy + colour) & 1), c = qx + 8 * qy + 64 * z;
	gs_cell<real>(S.H, S.ab[c], S.bb[c], (qx + 1) + 10 * (qy + 1) + 100 * (z + 1));
	__syncthreads();
}
__device__ inline int cp_hi(int cell) { return ((cell & 7) + 1) + 10 * (((cell >> 3) & 7) + 1) + 100 * ((cell >> 6) + 1); }

/// Down, one tile, a workgroup of 256: x = `inner` red->black sweeps on A x = b from x = 0.
template <typename real, typename MEM> __device__ inline void cp_presmooth_tile(CpTile<real> &S, const MgLv<real> &L, int slot, int inner) {
	const int t = threadIdx.x;
	const size_t base = (size_t)L.tiles[slot] * 512;
	for (int c = t; c < 512; c += 256) {
		S.ab[c] = L.abits[base + c];
		S.bb[c] = MEM::ld(L.b + base + c);
	}
	for (int i = t; i < LFA_HALO_CELLS; i += 256) S.H[i] = (real)0;
	__syncthreads();
	for (int it = 0; it < inner; ++it) {
		cp_half_sweep<real>(S, 0);
		cp_half_sweep<real>(S, 1);
	}
	for (int c = t; c < 512; c += 256) MEM::st(L.x + base + c, S.H[cp_hi(c)]);
	__syncthreads();
}

/// Ring cell r (0..383) of a tile: face f = r >> 6, in-face lane (a, b) = (r & 7, (r >> 3) & 7) as in load_halo.
__device__ inline void cp_ring(int r, int &f, int &hidx, int &ncell, int &dx, int &dy, int &dz) {
	f = r >> 6;
	const int lane = r & 63, a = lane & 7, b = lane >> 3;
	switch (f) {
	case 0: hidx = 0 + 10 * (a + 1) + 100 * (b + 1); ncell = b * 64 + a * 8 + 7; dx = -1; dy = a; dz = b; break;
	case 1: hidx = 9 + 10 * (a + 1) + 100 * (b + 1); ncell = b * 64 + a * 8; dx = 8; dy = a; dz = b; break;
	case 2: hidx = (a + 1) + 100 * (b + 1); ncell = b * 64 + 56 + a; dx = a; dy = -1; dz = b; break;
	case 3: hidx = (a + 1) + 90 + 100 * (b + 1); ncell = b * 64 + a; dx = a; dy = 8; dz = b; break;
	case 4: hidx = (a + 1) + 10 * (b + 1); ncell = 448 + lane; dx = a; dy = b; dz = -1; break;
	default: hidx = (a + 1) + 10 * (b + 1) + 900; ncell = lane; dx = a; dy = b; dz = 8; break;
	}
}

/// Down, one tile: residual r = b - A x of the level and its restriction to the next (half the sum over the 8 children).
template <typename real, typename MEM>
__device__ inline void cp_residual_restrict_tile(CpTile<real> &S, real *R, const MgLv<real> &L, const GridDims &gc, real *b_coarse, int slot) {
	const int t = threadIdx.x;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int tile = nt[6];
	const size_t base = (size_t)tile * 512;
	for (int c = t; c < 512; c += 256) {
		S.ab[c] = L.abits[base + c];
		S.bb[c] = MEM::ld(L.b + base + c);
		S.H[cp_hi(c)] = MEM::ld(L.x + base + c);
	}
	for (int r = t; r < 384; r += 256) {
		int f, hidx, ncell, dx, dy, dz;
		cp_ring(r, f, hidx, ncell, dx, dy, dz);
		const int nb = nt[f];
		S.H[hidx] = nb >= 0 ? MEM::ld(L.x + (size_t)nb * 512 + ncell) : (real)0;
	}
	__syncthreads();
	for (int c = t; c < 512; c += 256) {
		const int i = cp_hi(c);
		const uint32_t a = S.ab[c];
		real r = (real)0;
		if (a & AB_UNKNOWN) {
			const real F = (a & AB_FLUID) ? (real)1 : (real)0;
			real val = (real)(a & 7) * S.H[i];
			val = madd01(-F, S.H[i - 1], val);
			val = madd01(-F, S.H[i - 10], val);
			val = madd01(-F, S.H[i - 100], val);
			val = madd01(-(real)((a >> 3) & 1), S.H[i + 1], val);
			val = madd01(-(real)((a >> 4) & 1), S.H[i + 10], val);
			val = madd01(-(real)((a >> 5) & 1), S.H[i + 100], val);
			r = S.bb[c] - val;
		}
		R[c] = r;
	}
	__syncthreads();
	if (t < 64) {  // one coarse cell each: its 8 children in the order of the pair sums and the two shuffle steps
		const int X = t & 3, Y = (t >> 2) & 3, Z = t >> 4;
		auto pair = [&](int x, int y) { real p = R[x + 8 * y + 64 * (2 * Z)]; p += R[x + 8 * y + 64 * (2 * Z + 1)]; return p; };
		real v = pair(2 * X, 2 * Y);
		v += pair(2 * X + 1, 2 * Y);
		real w = pair(2 * X, 2 * Y + 1);
		w += pair(2 * X + 1, 2 * Y + 1);
		v += w;
		int tx, ty, tz;
		tile_coords(L.g, tile, tx, ty, tz);
		const int ptile = (tx >> 1) + gc.ntx * ((ty >> 1) + gc.nty * (tz >> 1));
		MEM::st(b_coarse + (size_t)ptile * 512 + ((tz & 1) * 4 + Z) * 64 + ((ty & 1) * 4 + Y) * 8 + (tx & 1) * 4 + X, (real)0.5 * v);
	}
	__syncthreads();
}

/// Up, one tile: x += P e, then `inner` black->red sweeps with the corrected values of the neighbour tiles on the ring.
template <typename real, typename MEM>
__device__ inline void cp_prolong_postsmooth_tile(CpTile<real> &S, const MgLv<real> &L, const GridDims &gc, const real *e, int slot, int inner) {
	const int t = threadIdx.x;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int tile = nt[6];
	const size_t base = (size_t)tile * 512;
	int tx, ty, tz;
	tile_coords(L.g, tile, tx, ty, tz);
	auto corr = [&](int X, int Y, int Z) -> real { return MEM::ld(e + blocked_index(gc, X >> 1, Y >> 1, Z >> 1)); };
	for (int c = t; c < 512; c += 256) {
		const uint8_t a = L.abits[base + c];
		S.ab[c] = a;
		S.bb[c] = MEM::ld(L.b + base + c);
		real v = MEM::ld(L.x + base + c);
		if (a & AB_UNKNOWN) v += corr(tx * 8 + (c & 7), ty * 8 + ((c >> 3) & 7), tz * 8 + (c >> 6));
		S.H[cp_hi(c)] = v;
	}
	for (int r = t; r < 384; r += 256) {
		int f, hidx, ncell, dx, dy, dz;
		cp_ring(r, f, hidx, ncell, dx, dy, dz);
		const int nb = nt[f];
		real v = (real)0;
		if (nb >= 0) {
			const size_t j = (size_t)nb * 512 + ncell;
			v = MEM::ld(L.x + j);
			if (L.abits[j] & AB_UNKNOWN) v += corr(tx * 8 + dx, ty * 8 + dy, tz * 8 + dz);
		}
		S.H[hidx] = v;
	}
	__syncthreads();
	for (int it = 0; it < inner; ++it) {
		cp_half_sweep<real>(S, 1);
		cp_half_sweep<real>(S, 0);
	}
	for (int c = t; c < 512; c += 256) MEM::st(L.y + base + c, S.H[cp_hi(c)]);
	__syncthreads();
}

/// Coarsest level (a single tile): nsw red->black sweeps followed by nsw black->red sweeps from zero (coarsest_tile, cell-parallel).
template <typename real, typename MEM> __device__ inline void cp_coarsest_tile(CpTile<real> &S, const MgLv<real> &L, int nsw) {
	const int t = threadIdx.x;
	const size_t base = (size_t)L.tiles[0] * 512;
	for (int c = t; c < 512; c += 256) {
		S.ab[c] = L.abits[base + c];
		S.bb[c] = MEM::ld(L.b + base + c);
	}
	for (int i = t; i < LFA_HALO_CELLS; i += 256) S.H[i] = (real)0;
	__syncthreads();
	for (int q = 0; q < 2 * nsw; ++q) {
		const int fc = q < nsw ? 0 : 1;
		cp_half_sweep<real>(S, fc);
		cp_half_sweep<real>(S, fc ^ 1);
	}
	for (int c = t; c < 512; c += 256) MEM::st(L.y + base + c, S.H[cp_hi(c)]);
	__syncthreads();
}

template <typename real>
__global__ void __launch_bounds__(256) k_mg_presmooth_cp(MgLv<real> L, int inner, const int *state) {
	__shared__ CpTile<real> S;
	if (state[0] >= 0) return;
	for (int slot = blockIdx.x; slot < L.n_tiles; slot += gridDim.x) cp_presmooth_tile<real, MemPlain>(S, L, slot, inner);
}
template <typename real>
__global__ void __launch_bounds__(256) k_mg_residual_restrict_cp(MgLv<real> L, GridDims gc, real *b_coarse, const int *state) {
	__shared__ CpTile<real> S;
	__shared__ real R[512];
	if (state[0] >= 0) return;
	for (int slot = blockIdx.x; slot < L.n_tiles; slot += gridDim.x) cp_residual_restrict_tile<real, MemPlain>(S, R, L, gc, b_coarse, slot);
}
template <typename real>
__global__ void __launch_bounds__(256) k_mg_prolong_postsmooth_cp(MgLv<real> L, GridDims gc, const real *e, int inner, const int *state) {
	__shared__ CpTile<real> S;
	if (state[0] >= 0) return;
	for (int slot = blockIdx.x; slot < L.n_tiles; slot += gridDim.x) cp_prolong_postsmooth_tile<real, MemPlain>(S, L, gc, e, slot, inner);
}

// ------------------------------------------------------------------------------------------------ the coarse levels in ONE launch
// A V-cycle below the finest level is a chain of short, dependent phases (pre-smoothing, residual + restriction per level, the
// coarsest solve, prolongation + post-smoothing per level): as separate launches each costs 5-8 us of dispatch latency whatever
// it computes (C4: 13.5 launches + the tail workgroup = 81 of an iteration's 315 us; C2: two thirds of the iteration).
// k_mg_coarse runs all of them in one launch, as DATAFLOW between workgroups:
//  * workgroup w owns tile slot w of EVERY level it reaches (a level has at most as many tiles as the one above, so a workgroup
//    works on levels first .. lmax(w) on the way down and rejoins at lmax(w) on the way up). The state of its tile on each level -
//    halo block with the pre-smoothed iterate and the neighbours' ring values, right-hand side, A bytes - stays in LDS from the
//    way down to the way up; static data (tile ids, neighbour tables, A bytes) is fetched once, before anything is waited for.
//  * what crosses workgroups - the ring values of the neighbour tiles, the restricted residual of the 8 children, the correction of
//    the parents - goes through the level arrays in global memory with agent-scope relaxed atomics (MemAgent: `sc1`, coherent per
//    access across the eight per-XCD L2s), so no release / acquire fence with its L2 write-back and invalidate is needed. That
//    fence is what made the earlier cooperative attempt slow (tools/xcd_barrier_probe.hip: 4.6-13 us per round with fences,
//    1.6-2.1 us with per-access coherence).
//  * there is no grid barrier and no atomic read-modify-write: a tile that has stored its output waits for the stores to be
//    acknowledged (`s_waitcnt vmcnt(0)`) and then stores this launch's TAG into its ready flag; a consumer polls the flags of
//    exactly the tiles it reads from (<= 6 neighbours, <= 8 children, <= 4 parents). Tags are unique per launch: flags are never
//    cleared. (A counter per phase was measured first: 256 producers incrementing one word serialise, 4-7 us per phase.)
//  * a half sweep keeps the coefficients of a thread's two cells in registers: 7 LDS reads, the update, a barrier.
// Deadlock freedom: every dependency points to an earlier phase of the same global phase order, and the launch has at most
// MG_CO_MAX_TILES workgroups, which are resident at once (or become resident as other kernels' workgroups retire).
// The arithmetic per cell is cp_*_tile's, so the result is bit-identical to the launch-per-phase path (LFA_MG_NO_PERSIST=1; tested).
template <typename real> struct CoLevel {
	real H[LFA_HALO_CELLS];  // halo block: pre-smoothed iterate + ring (0 where the neighbour tile is inactive)
	real b[512];
	uint8_t ab[512];
	uint8_t rab[384];  // A bytes of the ring cells (cp_ring order)
	int nb[8];         // the slot's row of the neighbour table: six neighbour tile ids, own tile id, mask of active child tiles
};
template <typename real> struct MgCo {
	MgLv<real> lv[MG_MAX_LEVELS];
	unsigned *ready[MG_MAX_LEVELS];  // per level 3 x (tiles of the level): tile has stored [0] its pre-smoothed iterate [1] its share of the next level's right-hand side [2] its result
	int first, last, nsw, inner;
	int top;  // >= 1: level first - 1 runs in the launch as well, top_wgs workgroups per phase in launch order (TAGGED only); -1: none
	int top_wgs;
	unsigned tag;
	unsigned long long *xq[MG_MAX_LEVELS];  // TAGGED hand-off: per level [x | b | y], each the level's padded grid of {tag, value} words
	size_t ncp[MG_MAX_LEVELS];
	int *abort;                  // pcg_state + 2: raised by a workgroup whose wait has passed CO_TIMEOUT_TICKS; every waiter checks it
	int fault;                   // LFA_MG_CO_FAULT=n (tests): workgroup n - 1 never raises its first flag
};

/// Back-off of a polling thread: 64 clocks at first, doubling to 8 K clocks. A producer may be late by far more than a phase -
/// its workgroup may not even be resident yet while another kernel's workgroups hold the CUs (the position correction on its own
/// stream) - and a few hundred workgroups re-reading the same words every 64 clocks starve exactly the kernel they wait for
/// (measured: 1.3 s per C2 step with every thread of a whole-solve kernel's reductions polling without back-off).
__device__ inline void co_backoff(int &n) {
	// (the first polls stay tight: a phase hands over within a few round trips, and 256 clocks of sleep too many per wait cost
	// k_mg_coarse 7 us per launch at C4)
	if (n < 16) __builtin_amdgcn_s_sleep(1);
	else if (n < 32) __builtin_amdgcn_s_sleep(8);
	else if (n < 64) __builtin_amdgcn_s_sleep(32);
	else __builtin_amdgcn_s_sleep(127);
	++n;
}
/// Every device-side wait is bounded: a producer that never shows up - its workgroup not resident because another process's
/// kernel of the same kind holds the compute units, a hardware fault, a bug - must end in an error code, not in a hung GPU.
/// After CO_TIMEOUT_TICKS of the constant 100 MHz clock (50 ms; a hand-off takes microseconds) the polling lane raises the solve's
/// abort word; every other waiter of the launch reads that word in its slow polls and leaves too. The host finds the word at its
/// next poll of the solver state, stops using the kernels that wait and repeats the solve on the launch-per-phase path (pcg.hip).
#define CO_TIMEOUT_TICKS 5000000ull
__device__ inline bool co_poll_expired(int tries, unsigned long long &t0, int *abort_word) {
	if (tries <= 64) return false;  // (the fast polls: a hand-off in time never gets here)
	if (t0 == 0ull) t0 = wall_clock64();
	return __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > CO_TIMEOUT_TICKS;
}
/// A ready flag = (launch tag << 4) | XCC id of the workgroup that raised it.
__device__ inline unsigned co_xcc_id() {
	unsigned x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	return x & 15u;
}
/// All threads: waits until every tile in dep[0 .. n) (n <= 8; -1 entries are skipped) carries this launch's tag in `flag`.
/// A wait that is given up (the ceiling)
/// raises the abort word and RETURNS like any other: the workgroup runs on with void data, every later wait of the launch leaves
/// at its first slow poll, the kernel ends within a millisecond and the host, which finds the word at its next poll, discards the
/// solve. (A uniform early exit would need a workgroup-wide OR per wait: two barriers and an LDS round trip in a kernel whose
/// phases are 2 us - measured: 47 -> 53 us per launch at C4.) Always returns true.
template <typename MEM>
__device__ inline bool co_wait(const unsigned *flag, const int *dep, int n, unsigned tag, int *abort_word) {
	if ((int)threadIdx.x < n) {
		const int d = dep[threadIdx.x];
		if (d >= 0) {
			int tries = 0, bad = 0;
			unsigned long long t0 = 0ull;
			unsigned f;
			while (((f = MEM::ld(flag + d)) >> 4) != tag) {
				co_backoff(tries);
				if (co_poll_expired(tries, t0, abort_word)) {
					bad = 1;
					break;
				}
			}
			if (bad) {
				// (for the record: abort_word[16] = 0x100 "a wait ran out" | the waiter's XCC id)
				if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
					abort_word[16] = 0x100 | (int)co_xcc_id();
				__hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	__syncthreads();
	return true;
}
/// All threads: this workgroup's stores so far have been acknowledged; then the tile's flag is raised.
template <typename MEM> __device__ inline void co_post(unsigned *flag, int tile, unsigned tag) {
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) MEM::st(flag + tile, tag << 4);
}

/// Per-thread constants of the dataflow kernels: the two cells of a thread are column (qx, qy), z = 2 j and 2 j + 1.
struct CoThread {
	int t, wg, qx, qy, qj, c0, c1, h0, h1;
	__device__ inline explicit CoThread(int wg_ = (int)blockIdx.x) {
		t = threadIdx.x; wg = wg_;
		qx = t & 7; qy = (t >> 3) & 7; qj = t >> 6;
		c0 = qx + 8 * qy + 128 * qj; c1 = c0 + 64;
		h0 = (qx + 1) + 10 * (qy + 1) + 100 * (2 * qj + 1); h1 = h0 + 100;
	}
};
/// One half sweep of colour `colour` on the halo block H with the coefficients of the thread's two cells.
template <typename real>
__device__ inline void co_half_sweep(const CoThread &T, real *H, uint32_t a0, uint32_t a1, real b0, real b1, int colour) {
	const int k = (T.qx + T.qy + colour) & 1;
	gs_cell<real>(H, k ? a1 : a0, k ? b1 : b0, k ? T.h1 : T.h0);
	__syncthreads();
}
/// Residual b - A x of the thread's two cells from the halo block (unscaled operator), term by term as cp_residual_restrict_tile.
template <typename real> __device__ inline real co_residual_cell(const real *H, uint32_t a, real b, int i) {
	real r = (real)0;
	if (a & AB_UNKNOWN) {
		const real F = (a & AB_FLUID) ? (real)1 : (real)0;
		real val = (real)(a & 7) * H[i];
		val = madd01(-F, H[i - 1], val);
		val = madd01(-F, H[i - 10], val);
		val = madd01(-F, H[i - 100], val);
		val = madd01(-(real)((a >> 3) & 1), H[i + 1], val);
		val = madd01(-(real)((a >> 4) & 1), H[i + 10], val);
		val = madd01(-(real)((a >> 5) & 1), H[i + 100], val);
		r = b - val;
	}
	return r;
}
/// Restriction of the residuals in R (512 cells of tile `tile` of level grid g) to its share of the parent level's right-hand
/// side: half the sum over the 8 children, in the order of the pair sums and the two shuffle steps of residual_restrict_tile.
template <typename real, typename MEM>
__device__ inline void co_restrict_store(const CoThread &T, const real *R, const GridDims &g, const GridDims &gc, int tile, real *b_coarse) {
	if (T.t < 64) {
		const int X = T.t & 3, Y = (T.t >> 2) & 3, Z = T.t >> 4;
		auto pair = [&](int x, int y) { real p = R[x + 8 * y + 64 * (2 * Z)]; p += R[x + 8 * y + 64 * (2 * Z + 1)]; return p; };
		real v = pair(2 * X, 2 * Y);
		v += pair(2 * X + 1, 2 * Y);
		real w = pair(2 * X, 2 * Y + 1);
		w += pair(2 * X + 1, 2 * Y + 1);
		v += w;
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		const int ptile = (tx >> 1) + gc.ntx * ((ty >> 1) + gc.nty * (tz >> 1));
		MEM::st(b_coarse + (size_t)ptile * 512 + ((tz & 1) * 4 + Z) * 64 + ((ty & 1) * 4 + Y) * 8 + (tx & 1) * 4 + X, (real)0.5 * v);
	}
}
/// dep[0..8) = the active child tiles (level below, grid gf) of `tile` (grid g), -1 where there is none.
__device__ inline void co_child_deps(const CoThread &T, int *dep, const GridDims &g, const GridDims &gf, int tile, int child_mask) {
	if (T.t < 8) {
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		const int cx = 2 * tx + (T.t & 1), cy = 2 * ty + ((T.t >> 1) & 1), cz = 2 * tz + (T.t >> 2);
		dep[T.t] = ((child_mask >> T.t) & 1) ? cx + gf.ntx * (cy + gf.nty * cz) : -1;
	}
	__syncthreads();
}
/// dep[0..7) = the parent tiles (grid gc) of the six neighbour tiles and of the tile itself (nb[0..7) of a level with grid g).
__device__ inline void co_parent_deps(const CoThread &T, int *dep, const GridDims &g, const GridDims &gc, const int *nb) {
	if (T.t < 7) {
		const int src = nb[T.t];
		int d = -1;
		if (src >= 0) {
			int sx, sy, sz;
			tile_coords(g, src, sx, sy, sz);
			d = (sx >> 1) + gc.ntx * ((sy >> 1) + gc.nty * (sz >> 1));
		}
		dep[T.t] = d;
	}
	__syncthreads();
}
/// x += P e on the thread's two cells and on the ring cells that are unknowns of an active neighbour (cp_prolong_postsmooth_tile).
template <typename real, typename MEM>
__device__ inline void co_add_correction(const CoThread &T, real *H, uint32_t a0, uint32_t a1, const uint8_t *rab, const int *nb,
                                         const GridDims &g, const GridDims &gc, int tile, const real *e) {
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	{
		// cells z = 2 j and 2 j + 1 of a column share their parent
		const real corr = MEM::ld(e + blocked_index(gc, (tx * 8 + T.qx) >> 1, (ty * 8 + T.qy) >> 1, (tz * 8 + 2 * T.qj) >> 1));
		if (a0 & AB_UNKNOWN) H[T.h0] = H[T.h0] + corr;
		if (a1 & AB_UNKNOWN) H[T.h1] = H[T.h1] + corr;
	}
	for (int r = T.t; r < 384; r += 256) {
		int f, hidx, ncell, dx, dy, dz;
		cp_ring(r, f, hidx, ncell, dx, dy, dz);
		if (nb[f] >= 0 && (rab[r] & AB_UNKNOWN))
			H[hidx] = H[hidx] + MEM::ld(e + blocked_index(gc, (tx * 8 + dx) >> 1, (ty * 8 + dy) >> 1, (tz * 8 + dz) >> 1));
	}
	__syncthreads();
}
/// Ring of the halo block from the neighbour tiles' values in `v` (inactive neighbours: the ring keeps its zeros).
template <typename real, typename MEM> __device__ inline void co_load_ring(const CoThread &T, real *H, const int *nb, const real *v) {
	for (int r = T.t; r < 384; r += 256) {
		int f, hidx, ncell, dx, dy, dz;
		cp_ring(r, f, hidx, ncell, dx, dy, dz);
		const int n = nb[f];
		if (n >= 0) H[hidx] = MEM::ld(v + (size_t)n * 512 + ncell);
	}
	__syncthreads();
}

// ---- the tagged hand-off (round 5). A value travels as ONE 8-byte word {launch tag, fp32 bits}, stored and loaded with agent-scope
// relaxed 64-bit accesses (single-copy atomic): the consumer polls the very words it needs until they carry this launch's tag.
// Against "store the data, wait for the acknowledgement, barrier, store a flag / poll the flag, barrier, load the data" that is one
// trip through the memory side instead of three: tools/handoff_probe.hip measures 1.1-1.35 us per hand-off against 1.7-2.6
// (16-384 pairs of workgroups), and a V-cycle's coarse levels are a chain of 10-13 hand-offs. Every thread polls only words of its
// own (no two workgroups spin on one address: the "storm" of co_backoff's comment does not arise); waits are bounded like co_wait's.
// fp32 vectors only (a double leaves no room for the tag); fp64 solves keep the flags.
__device__ inline unsigned long long co_tagged(unsigned tag, float v) { return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v); }
__device__ inline void co_put(unsigned long long *p, unsigned tag, float v) {
	__hip_atomic_store(p, co_tagged(tag, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void co_put(unsigned long long *, unsigned, double) {}  // (never called: TAGGED is fp32 only)
/// Polls N words (p[k] == nullptr: none, the value stays 0) until each carries `tag`; all N loads of a round are in flight
/// together. A wait that passes the ceiling raises the abort word and returns what it has (see co_wait).
template <int N> __device__ inline void co_get(const unsigned long long *const (&p)[N], unsigned tag, int *abort_word, float (&v)[N]) {
	unsigned long long w[N];
	bool all = true;
#pragma unroll
	for (int k = 0; k < N; ++k) {
		w[k] = p[k] ? __hip_atomic_load(p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
		all &= (unsigned)(w[k] >> 32) == tag;
	}
	if (!all) {
		int tries = 0;
		unsigned long long t0 = 0ull;
		do {
			co_backoff(tries);
			if (co_poll_expired(tries, t0, abort_word)) {
				if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) abort_word[16] = 0x400 | (int)co_xcc_id();
				__hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				break;
			}
			all = true;
#pragma unroll
			for (int k = 0; k < N; ++k) {
				if ((unsigned)(w[k] >> 32) != tag) w[k] = __hip_atomic_load(p[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				all &= (unsigned)(w[k] >> 32) == tag;
			}
		} while (!all);
	}
#pragma unroll
	for (int k = 0; k < N; ++k) v[k] = __uint_as_float((unsigned)w[k]);
}
/// co_restrict_store with the parent's share going out as tagged words.
template <typename real>
__device__ inline void co_restrict_put(const CoThread &T, const real *R, const GridDims &g, const GridDims &gc, int tile, unsigned long long *bq,
                                       unsigned tag) {
	if (T.t < 64) {
		const int X = T.t & 3, Y = (T.t >> 2) & 3, Z = T.t >> 4;
		auto pair = [&](int x, int y) { real p = R[x + 8 * y + 64 * (2 * Z)]; p += R[x + 8 * y + 64 * (2 * Z + 1)]; return p; };
		real v = pair(2 * X, 2 * Y);
		v += pair(2 * X + 1, 2 * Y);
		real w = pair(2 * X, 2 * Y + 1);
		w += pair(2 * X + 1, 2 * Y + 1);
		v += w;
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		const int ptile = (tx >> 1) + gc.ntx * ((ty >> 1) + gc.nty * (tz >> 1));
		co_put(bq + (size_t)ptile * 512 + ((tz & 1) * 4 + Z) * 64 + ((ty & 1) * 4 + Y) * 8 + (tx & 1) * 4 + X, tag, (real)0.5 * v);
	}
}
/// The thread's two right-hand-side values of a tile whose children hand their shares over as tagged words: a cell of an
/// inactive child (bit of `child_mask` clear) is zero and nobody writes it.
template <typename real>
__device__ inline void co_get_rhs(const CoThread &T, const unsigned long long *bq, size_t base, int child_mask, unsigned tag, int *abort_word,
                                  real &b0, real &b1) {
	const int child = (T.qx >> 2) + 2 * (T.qy >> 2) + 4 * (T.qj >> 1);  // (cells z = 2 j and 2 j + 1 lie in the same child)
	const bool on = (child_mask >> child) & 1;
	const unsigned long long *const p[2] = {on ? bq + base + T.c0 : nullptr, on ? bq + base + T.c1 : nullptr};
	float v[2];
	co_get<2>(p, tag, abort_word, v);
	b0 = (real)v[0];
	b1 = (real)v[1];
}
/// co_load_ring on tagged words.
template <typename real> __device__ inline void co_get_ring(const CoThread &T, real *H, const int *nb, const unsigned long long *xq, unsigned tag,
                                                               int *abort_word) {
	int f0, f1, h0, h1, n0, n1, dx, dy, dz;
	const int r1 = T.t + 256;
	cp_ring(T.t, f0, h0, n0, dx, dy, dz);
	cp_ring(r1 < 384 ? r1 : 0, f1, h1, n1, dx, dy, dz);
	const bool on0 = nb[f0] >= 0, on1 = r1 < 384 && nb[f1] >= 0;
	const unsigned long long *const p[2] = {on0 ? xq + (size_t)nb[f0] * 512 + n0 : nullptr, on1 ? xq + (size_t)nb[f1] * 512 + n1 : nullptr};
	float v[2];
	co_get<2>(p, tag, abort_word, v);
	if (on0) H[h0] = (real)v[0];
	if (on1) H[h1] = (real)v[1];
	__syncthreads();
}
/// co_add_correction on tagged words (the parents' results).
template <typename real>
__device__ inline void co_get_correction(const CoThread &T, real *H, uint32_t a0, uint32_t a1, const uint8_t *rab, const int *nb, const GridDims &g,
                                         const GridDims &gc, int tile, const unsigned long long *yq, unsigned tag, int *abort_word) {
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	int f0, f1, h0, h1, n0, n1, dx0, dy0, dz0, dx1, dy1, dz1;
	const int r1 = T.t + 256;
	cp_ring(T.t, f0, h0, n0, dx0, dy0, dz0);
	cp_ring(r1 < 384 ? r1 : 0, f1, h1, n1, dx1, dy1, dz1);
	const bool on0 = nb[f0] >= 0 && (rab[T.t] & AB_UNKNOWN), on1 = r1 < 384 && nb[f1] >= 0 && (rab[r1 < 384 ? r1 : 0] & AB_UNKNOWN);
	// cells z = 2 j and 2 j + 1 of a column share their parent
	const unsigned long long *const p[3] = {
		yq + blocked_index(gc, (tx * 8 + T.qx) >> 1, (ty * 8 + T.qy) >> 1, (tz * 8 + 2 * T.qj) >> 1),
		on0 ? yq + blocked_index(gc, (tx * 8 + dx0) >> 1, (ty * 8 + dy0) >> 1, (tz * 8 + dz0) >> 1) : nullptr,
		on1 ? yq + blocked_index(gc, (tx * 8 + dx1) >> 1, (ty * 8 + dy1) >> 1, (tz * 8 + dz1) >> 1) : nullptr};
	float v[3];
	co_get<3>(p, tag, abort_word, v);
	const real corr = (real)v[0];
	if (a0 & AB_UNKNOWN) H[T.h0] = H[T.h0] + corr;
	if (a1 & AB_UNKNOWN) H[T.h1] = H[T.h1] + corr;
	if (on0) H[h0] = H[h0] + (real)v[1];
	if (on1) H[h1] = H[h1] + (real)v[2];
	__syncthreads();
}

/// Static data of levels P.first .. lmax into st[] (LDS): neighbour-table rows, A bytes of the own tile and of its ring; the
/// halo blocks are cleared. Two dependent round trips. `state` (may be null): read together with the first batch; returns false
/// (uniformly) when the solve has converged. `prefetch_b`: the first level's right-hand side was written by an earlier kernel
/// and is fetched here too.
template <typename real, typename MEM>
__device__ inline bool co_static(const MgCo<real> &P, const CoThread &T, CoLevel<real> *st, int lmax, const int *state, bool prefetch_b) {
	const int t = T.t, nlev = P.last - P.first + 1;
	int nbv = 0;
	if (t < 8 * nlev && P.first + (t >> 3) <= lmax) nbv = P.lv[P.first + (t >> 3)].nbr[(size_t)T.wg * MG_NBR_STRIDE + (t & 7)];
	if (state) {
		const int st0 = state[0];
		if (st0 >= 0 || state[2] != 0) return false;  // converged (or a wait was given up): the launches queued behind are no-ops
	}
	if (t < 8 * nlev) st[t >> 3].nb[t & 7] = nbv;
	for (int l = P.first; l <= lmax; ++l) {
		CoLevel<real> &S = st[l - P.first];
		for (int i = t; i < LFA_HALO_CELLS; i += 256) S.H[i] = (real)0;
	}
	__syncthreads();
	{
		uint32_t va[MG_CO_MAX_LEVELS];  // four A bytes per level, packed: this batch is the kernel's register peak otherwise
		int r1 = t + 256, f0, f1, hidx, nc0, nc1, dx, dy, dz;
		cp_ring(t, f0, hidx, nc0, dx, dy, dz);
		cp_ring(r1 < 384 ? r1 : 0, f1, hidx, nc1, dx, dy, dz);
#pragma unroll
		for (int k = 0; k < MG_CO_MAX_LEVELS; ++k) {
			const int l = P.first + k;
			va[k] = 0;
			if (l <= lmax) {
				const uint8_t *ab = P.lv[l].abits;
				const int *nb = st[k].nb;
				uint32_t b0 = ab[(size_t)nb[6] * 512 + T.c0], b1 = ab[(size_t)nb[6] * 512 + T.c1], b2 = 0, b3 = 0;
				if (nb[f0] >= 0) b2 = ab[(size_t)nb[f0] * 512 + nc0];
				if (r1 < 384 && nb[f1] >= 0) b3 = ab[(size_t)nb[f1] * 512 + nc1];
				va[k] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
			}
		}
		real fb0 = (real)0, fb1 = (real)0;
		if (prefetch_b && P.first <= lmax) {
			// (issued behind the A bytes, consumed by the first phase: the right-hand side the previous kernel has restricted)
			const size_t base0 = (size_t)st[0].nb[6] * 512;
			fb0 = MEM::ld(P.lv[P.first].b + base0 + T.c0);
			fb1 = MEM::ld(P.lv[P.first].b + base0 + T.c1);
		}
#pragma unroll
		for (int k = 0; k < MG_CO_MAX_LEVELS; ++k)
			if (P.first + k <= lmax) {
				st[k].ab[T.c0] = (uint8_t)(va[k] & 255);
				st[k].ab[T.c1] = (uint8_t)((va[k] >> 8) & 255);
				st[k].rab[t] = (uint8_t)((va[k] >> 16) & 255);
				if (r1 < 384) st[k].rab[r1] = (uint8_t)(va[k] >> 24);
			}
		if (prefetch_b && P.first <= lmax) {
			st[0].b[T.c0] = fb0;
			st[0].b[T.c1] = fb1;
		}
	}
	__syncthreads();
	return true;
}

/// One V-cycle of levels P.first .. P.last: down, coarsest solve (workgroup 0), up. The halo blocks of st[] must be zero.
/// `b_prefetched`: the first level's right-hand side is already in st[0].b (k_mg_coarse); otherwise it is waited for like
/// every other level's (its children sit on level P.first - 1, whose flags P.ready[P.first - 1] must be valid).
/// `post_first_y`: raise the result flag of the first level too (somebody inside this launch consumes it).
/// TAGGED: what the workgroups hand to each other travels as {tag, value} words (P.xq) instead of through the level arrays + ready
/// flags - see co_put / co_get; the first level's right-hand side (in) and result (out) stay plain: other kernels own them.
template <typename real, typename MEM, bool TAGGED>
__device__ inline bool co_cycle(const MgCo<real> &P, const CoThread &T, CoLevel<real> *st, real *R, int *dep, unsigned tag, int lmax,
                                bool b_prefetched, bool post_first_y) {
	const int t = T.t, c0 = T.c0, c1 = T.c1, h0 = T.h0, h1 = T.h1;
	auto xq = [&](int l) { return P.xq[l]; };
	auto bq = [&](int l) { return P.xq[l] + P.ncp[l]; };
	auto yq = [&](int l) { return P.xq[l] + 2 * P.ncp[l]; };
	// ---- down
	for (int l = P.first; l < P.last && l <= lmax; ++l) {
		CoLevel<real> &S = st[l - P.first];
		const MgLv<real> &L = P.lv[l];
		const int tile = S.nb[6], nt = L.g.nt;
		const size_t base = (size_t)tile * 512;
		const uint32_t a0 = S.ab[c0], a1 = S.ab[c1];
		const bool waited = l > P.first || !b_prefetched;
		real b0, b1;
		if (waited && TAGGED) {  // the right-hand side is the restricted residual of the child tiles (level l - 1)
			co_get_rhs<real>(T, bq(l), base, S.nb[7], tag, P.abort, b0, b1);
			S.b[c0] = b0;
			S.b[c1] = b1;
		} else if (waited) {
			const GridDims &gf = P.lv[l - 1].g;
			co_child_deps(T, dep, L.g, gf, tile, S.nb[7]);
			if (!co_wait<MEM>(P.ready[l - 1] + gf.nt, dep, 8, tag, P.abort)) return false;
			b0 = MEM::ld(L.b + base + c0);
			b1 = MEM::ld(L.b + base + c1);
			S.b[c0] = b0;
			S.b[c1] = b1;
		} else {  // fetched with the static data
			b0 = S.b[c0];
			b1 = S.b[c1];
		}
		// pre-smoothing from zero: the ring does not enter (the halo block is still zero here)
		for (int it = 0; it < P.inner; ++it) {
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, 0);
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, 1);
		}
		const bool faulty = P.fault && l == P.first && T.wg == P.fault - 1;  // (fault injection: see MgCo::fault)
		if (TAGGED) {
			if (!faulty) {
				co_put(xq(l) + base + c0, tag, S.H[h0]);
				co_put(xq(l) + base + c1, tag, S.H[h1]);
			}
			// residual: the ring holds the neighbours' pre-smoothed values
			co_get_ring<real>(T, S.H, S.nb, xq(l), tag, P.abort);
		} else {
			MEM::st(L.x + base + c0, S.H[h0]);
			MEM::st(L.x + base + c1, S.H[h1]);
			if (!faulty) co_post<MEM>(P.ready[l], tile, tag);
			if (t < 6) dep[t] = S.nb[t];
			__syncthreads();
			if (!co_wait<MEM>(P.ready[l], dep, 6, tag, P.abort)) return false;
			co_load_ring<real, MEM>(T, S.H, S.nb, L.x);
		}
		R[c0] = co_residual_cell<real>(S.H, a0, b0, h0);
		R[c1] = co_residual_cell<real>(S.H, a1, b1, h1);
		__syncthreads();
		if (TAGGED) {
			co_restrict_put<real>(T, R, L.g, P.lv[l + 1].g, tile, bq(l + 1), tag);
			__syncthreads();  // (R is rewritten by the next level)
		} else {
			co_restrict_store<real, MEM>(T, R, L.g, P.lv[l + 1].g, tile, P.lv[l + 1].b);
			co_post<MEM>(P.ready[l] + nt, tile, tag);
		}
	}
	// ---- coarsest level (one tile, workgroup 0): nsw sweeps red->black, nsw black->red from zero
	if (T.wg == 0) {
		const int l = P.last;
		CoLevel<real> &S = st[l - P.first];
		const MgLv<real> &L = P.lv[l];
		const int tile = S.nb[6];
		const size_t base = (size_t)tile * 512;
		const uint32_t a0 = S.ab[c0], a1 = S.ab[c1];
		const bool waited = l > P.first || !b_prefetched;
		real b0, b1;
		if (waited && TAGGED) {
			co_get_rhs<real>(T, bq(l), base, S.nb[7], tag, P.abort, b0, b1);
		} else if (waited) {
			const GridDims &gf = P.lv[l - 1].g;
			co_child_deps(T, dep, L.g, gf, tile, S.nb[7]);
			if (!co_wait<MEM>(P.ready[l - 1] + gf.nt, dep, 8, tag, P.abort)) return false;
			b0 = MEM::ld(L.b + base + c0);
			b1 = MEM::ld(L.b + base + c1);
		} else {
			b0 = S.b[c0];
			b1 = S.b[c1];
		}
		for (int q = 0; q < 2 * P.nsw; ++q) {
			const int fc = q < P.nsw ? 0 : 1;
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, fc);
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, fc ^ 1);
		}
		if (TAGGED && (l > P.first || post_first_y)) {
			co_put(yq(l) + base + c0, tag, S.H[h0]);
			co_put(yq(l) + base + c1, tag, S.H[h1]);
		} else {
			MEM::st(L.y + base + c0, S.H[h0]);
			MEM::st(L.y + base + c1, S.H[h1]);
			if (l > P.first || post_first_y) co_post<MEM>(P.ready[l] + 2 * L.g.nt, tile, tag);
		}
	}
	// ---- up
	for (int l = (lmax < P.last - 1 ? lmax : P.last - 1); l >= P.first; --l) {
		CoLevel<real> &S = st[l - P.first];
		const MgLv<real> &L = P.lv[l];
		const GridDims &gc = P.lv[l + 1].g;
		const int tile = S.nb[6];
		const size_t base = (size_t)tile * 512;
		const uint32_t a0 = S.ab[c0], a1 = S.ab[c1];
		const real b0 = S.b[c0], b1 = S.b[c1];
		// the corrections come from the parent tile and from the parents of the active neighbour tiles
		if (TAGGED) {
			co_get_correction<real>(T, S.H, a0, a1, S.rab, S.nb, L.g, gc, tile, yq(l + 1), tag, P.abort);
		} else {
			co_parent_deps(T, dep, L.g, gc, S.nb);
			if (!co_wait<MEM>(P.ready[l + 1] + 2 * gc.nt, dep, 7, tag, P.abort)) return false;
			co_add_correction<real, MEM>(T, S.H, a0, a1, S.rab, S.nb, L.g, gc, tile, P.lv[l + 1].y);
		}
		for (int it = 0; it < P.inner; ++it) {
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, 1);
			co_half_sweep<real>(T, S.H, a0, a1, b0, b1, 0);
		}
		if (TAGGED && (l > P.first || post_first_y)) {
			co_put(yq(l) + base + c0, tag, S.H[h0]);
			co_put(yq(l) + base + c1, tag, S.H[h1]);
		} else {
			MEM::st(L.y + base + c0, S.H[h0]);
			MEM::st(L.y + base + c1, S.H[h1]);
			if (l > P.first || post_first_y) co_post<MEM>(P.ready[l] + 2 * L.g.nt, tile, tag);
		}
	}
	return true;
}

// ---- the level above, fused into the launch in LAUNCH ORDER (round 5, tagged hand-off only)
// The first level inside k_mg_coarse is the largest whose tiles are all resident at once; the level above it (C4: level 1, 2 400
// tiles; C3: ~600) kept three launches of its own - pre-smoothing, residual + restriction, prolongation + post-smoothing: 28 us of a
// 258 us iteration at C4 for a few microseconds of work. They now run inside the same launch as extra workgroups in front of and
// behind the resident ones, a workgroup per (phase, tile):
//     [0, n)  pre-smoothing   [n, 2 n)  residual + restriction   [2 n, 2 n + W)  the resident workgroups   [2 n + W, 3 n + W)  the way up
// Every value a workgroup waits for is a tagged word written by a workgroup with a LOWER index (its neighbours' or children's
// earlier phase, or its parent in the resident block). The dispatcher hands out workgroups in index order per XCD, and a workgroup
// that has started never waits for a later one: whoever is waited for is running or done - no residency requirement for the n-tile
// phases, nothing to deadlock. (Round 4's ordered work queue reached the same guarantee with 38 000 same-address atomics per launch
// and was 6x slower; here nothing is shared but the data itself.) The waits are bounded like every other (co_get).
// Arithmetic per cell: cp_presmooth_tile / cp_residual_restrict_tile / cp_prolong_postsmooth_tile = the wave-per-tile kernels'.
/// Phase 1 of the top level: cp_presmooth_tile with the iterate going out as tagged words.
template <typename real>
__device__ inline void top_presmooth(CpTile<real> &S, const MgLv<real> &L, int slot, int inner, unsigned long long *xq, unsigned tag) {
	const int t = threadIdx.x;
	const size_t base = (size_t)L.tiles[slot] * 512;
	for (int c = t; c < 512; c += 256) {
		S.ab[c] = L.abits[base + c];
		S.bb[c] = L.b[base + c];  // (restricted by the previous launch)
	}
	for (int i = t; i < LFA_HALO_CELLS; i += 256) S.H[i] = (real)0;
	__syncthreads();
	for (int it = 0; it < inner; ++it) {
		cp_half_sweep<real>(S, 0);
		cp_half_sweep<real>(S, 1);
	}
	for (int c = t; c < 512; c += 256) co_put(xq + base + c, tag, S.H[cp_hi(c)]);
}
/// Phase 2: cp_residual_restrict_tile; the iterate (own cells and ring) comes in, the parent's share goes out, as tagged words.
template <typename real>
__device__ inline void top_residual_restrict(CpTile<real> &S, real *R, const MgLv<real> &L, const GridDims &gc, int slot,
                                             const unsigned long long *xq, unsigned long long *bq_coarse, unsigned tag, int *abort_word) {
	const int t = threadIdx.x;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int tile = nt[6];
	const size_t base = (size_t)tile * 512;
	int f0, f1, h0, h1, n0, n1, dx, dy, dz;
	const int r1 = t + 256;
	cp_ring(t, f0, h0, n0, dx, dy, dz);
	cp_ring(r1 < 384 ? r1 : 0, f1, h1, n1, dx, dy, dz);
	const int nb0 = nt[f0], nb1 = r1 < 384 ? nt[f1] : -1;
	const unsigned long long *const p[4] = {xq + base + t, xq + base + t + 256, nb0 >= 0 ? xq + (size_t)nb0 * 512 + n0 : nullptr,
	                                        nb1 >= 0 ? xq + (size_t)nb1 * 512 + n1 : nullptr};
	for (int c = t; c < 512; c += 256) {
		S.ab[c] = L.abits[base + c];
		S.bb[c] = L.b[base + c];
	}
	float v[4];
	co_get<4>(p, tag, abort_word, v);
	S.H[cp_hi(t)] = (real)v[0];
	S.H[cp_hi(t + 256)] = (real)v[1];
	S.H[h0] = (real)v[2];  // (0 where the neighbour tile is inactive)
	if (r1 < 384) S.H[h1] = (real)v[3];
	__syncthreads();
	for (int c = t; c < 512; c += 256) {
		const int i = cp_hi(c);
		const uint32_t a = S.ab[c];
		real r = (real)0;
		if (a & AB_UNKNOWN) {
			const real F = (a & AB_FLUID) ? (real)1 : (real)0;
			real val = (real)(a & 7) * S.H[i];
			val = madd01(-F, S.H[i - 1], val);
			val = madd01(-F, S.H[i - 10], val);
			val = madd01(-F, S.H[i - 100], val);
			val = madd01(-(real)((a >> 3) & 1), S.H[i + 1], val);
			val = madd01(-(real)((a >> 4) & 1), S.H[i + 10], val);
			val = madd01(-(real)((a >> 5) & 1), S.H[i + 100], val);
			r = S.bb[c] - val;
		}
		R[c] = r;
	}
	__syncthreads();
	if (t < 64) {  // one coarse cell each: its 8 children in the order of the pair sums and the two shuffle steps
		const int X = t & 3, Y = (t >> 2) & 3, Z = t >> 4;
		auto pair = [&](int x, int y) { real q = R[x + 8 * y + 64 * (2 * Z)]; q += R[x + 8 * y + 64 * (2 * Z + 1)]; return q; };
		real v2 = pair(2 * X, 2 * Y);
		v2 += pair(2 * X + 1, 2 * Y);
		real w = pair(2 * X, 2 * Y + 1);
		w += pair(2 * X + 1, 2 * Y + 1);
		v2 += w;
		int tx, ty, tz;
		tile_coords(L.g, tile, tx, ty, tz);
		const int ptile = (tx >> 1) + gc.ntx * ((ty >> 1) + gc.nty * (tz >> 1));
		co_put(bq_coarse + (size_t)ptile * 512 + ((tz & 1) * 4 + Z) * 64 + ((ty & 1) * 4 + Y) * 8 + (tx & 1) * 4 + X, tag, (real)0.5 * v2);
	}
}
/// Phase 3: cp_prolong_postsmooth_tile; the iterate and the parents' corrections come in as tagged words, the result goes to the
/// level's plain array (the finer level's launch reads it).
template <typename real>
__device__ inline void top_prolong_postsmooth(CpTile<real> &S, const MgLv<real> &L, const GridDims &gc, int slot, int inner,
                                              const unsigned long long *xq, const unsigned long long *yq_coarse, unsigned tag, int *abort_word) {
	const int t = threadIdx.x;
	const int *nt = L.nbr + (size_t)slot * MG_NBR_STRIDE;
	const int tile = nt[6];
	const size_t base = (size_t)tile * 512;
	int tx, ty, tz;
	tile_coords(L.g, tile, tx, ty, tz);
	auto parent = [&](int X, int Y, int Z) { return yq_coarse + blocked_index(gc, X >> 1, Y >> 1, Z >> 1); };
	int f0, f1, h0, h1, n0, n1, dx0, dy0, dz0, dx1, dy1, dz1;
	const int r1 = t + 256;
	cp_ring(t, f0, h0, n0, dx0, dy0, dz0);
	cp_ring(r1 < 384 ? r1 : 0, f1, h1, n1, dx1, dy1, dz1);
	const int nb0 = nt[f0], nb1 = r1 < 384 ? nt[f1] : -1;
	const int c0 = t, c1 = t + 256;
	const uint8_t a0 = L.abits[base + c0], a1 = L.abits[base + c1];
	const bool ru0 = nb0 >= 0 && (L.abits[(size_t)nb0 * 512 + n0] & AB_UNKNOWN), ru1 = nb1 >= 0 && (L.abits[(size_t)nb1 * 512 + n1] & AB_UNKNOWN);
	const unsigned long long *const p[8] = {
		xq + base + c0, xq + base + c1,
		(a0 & AB_UNKNOWN) ? parent(tx * 8 + (c0 & 7), ty * 8 + ((c0 >> 3) & 7), tz * 8 + (c0 >> 6)) : nullptr,
		(a1 & AB_UNKNOWN) ? parent(tx * 8 + (c1 & 7), ty * 8 + ((c1 >> 3) & 7), tz * 8 + (c1 >> 6)) : nullptr,
		nb0 >= 0 ? xq + (size_t)nb0 * 512 + n0 : nullptr, nb1 >= 0 ? xq + (size_t)nb1 * 512 + n1 : nullptr,
		ru0 ? parent(tx * 8 + dx0, ty * 8 + dy0, tz * 8 + dz0) : nullptr, ru1 ? parent(tx * 8 + dx1, ty * 8 + dy1, tz * 8 + dz1) : nullptr};
	S.ab[c0] = a0;
	S.ab[c1] = a1;
	S.bb[c0] = L.b[base + c0];
	S.bb[c1] = L.b[base + c1];
	float v[8];
	co_get<8>(p, tag, abort_word, v);
	{
		real x0 = (real)v[0], x1 = (real)v[1];
		if (a0 & AB_UNKNOWN) x0 += (real)v[2];
		if (a1 & AB_UNKNOWN) x1 += (real)v[3];
		S.H[cp_hi(c0)] = x0;
		S.H[cp_hi(c1)] = x1;
		real g0 = (real)0, g1 = (real)0;
		if (nb0 >= 0) {
			g0 = (real)v[4];
			if (ru0) g0 += (real)v[6];
		}
		if (nb1 >= 0) {
			g1 = (real)v[5];
			if (ru1) g1 += (real)v[7];
		}
		S.H[h0] = g0;
		if (r1 < 384) S.H[h1] = g1;
	}
	__syncthreads();
	for (int it = 0; it < inner; ++it) {
		cp_half_sweep<real>(S, 1);
		cp_half_sweep<real>(S, 0);
	}
	for (int c = t; c < 512; c += 256) L.y[base + c] = S.H[cp_hi(c)];
}

template <typename real, typename MEM, bool TAGGED>
__global__ void __launch_bounds__(256) k_mg_coarse(MgCo<real> P, const int *state) {
	extern __shared__ unsigned char co_smem[];
	__shared__ int dep[8];
	// P.top >= 1 (TAGGED only): the level above runs in this launch too - see "fused in launch order" above. Its n tiles are dealt
	// to P.top_wgs workgroups per phase (slot = workgroup, + top_wgs, ...): fewer, longer workgroups to dispatch around the resident ones.
	const int n_top = TAGGED && P.top >= 0 ? P.top_wgs : 0, W = (int)gridDim.x - 3 * n_top;
	if (TAGGED && n_top) {
		const int b = (int)blockIdx.x;
		if (b < 2 * n_top || b >= 2 * n_top + W) {
			if (state[0] >= 0 || state[2] != 0) return;  // converged, or a wait was given up
			CpTile<real> &S = *(CpTile<real> *)co_smem;
			real *R = (real *)(&S + 1);
			const MgLv<real> &L = P.lv[P.top];
			unsigned long long *xq = P.xq[P.top], *q1 = P.xq[P.first];
			if (b < n_top) {
				for (int slot = b; slot < L.n_tiles; slot += n_top) {
					top_presmooth<real>(S, L, slot, P.inner, xq, P.tag);
					__syncthreads();
				}
			} else if (b < 2 * n_top) {
				for (int slot = b - n_top; slot < L.n_tiles; slot += n_top) {
					top_residual_restrict<real>(S, R, L, P.lv[P.first].g, slot, xq, q1 + P.ncp[P.first], P.tag, P.abort);
					__syncthreads();
				}
			} else {
				// The way up waits for the whole chain of the resident workgroups (tens of microseconds) and hundreds of these
				// workgroups are waiting: ONE thread polls ONE word - the parent's result for the tile's first cell - and only when
				// that has arrived do the 256 threads ask for their eight words each (which are then there, or a microsecond away).
				// Everybody polling everything from the start is 500 000 threads re-reading 64 bytes each: the storm co_backoff's
				// comment describes.
				for (int slot = b - 2 * n_top - W; slot < L.n_tiles; slot += n_top) {
					if (threadIdx.x == 0) {
						int tx, ty, tz;
						tile_coords(L.g, L.nbr[(size_t)slot * MG_NBR_STRIDE + 6], tx, ty, tz);
						const unsigned long long *const gate[1] = {q1 + 2 * P.ncp[P.first] + blocked_index(P.lv[P.first].g, (tx * 8) >> 1, (ty * 8) >> 1, (tz * 8) >> 1)};
						float unused[1];
						co_get<1>(gate, P.tag, P.abort, unused);
					}
					__syncthreads();
					top_prolong_postsmooth<real>(S, L, P.lv[P.first].g, slot, P.inner, xq, q1 + 2 * P.ncp[P.first], P.tag, P.abort);
					__syncthreads();
				}
			}
			return;
		}
	}
	const CoThread T((int)blockIdx.x - 2 * n_top);
	const int nlev = P.last - P.first + 1;
	CoLevel<real> *st = (CoLevel<real> *)co_smem;
	real *R = (real *)(st + nlev);
	int lmax = P.first - 1;  // deepest level this workgroup owns a tile of
	for (int l = P.first; l <= P.last; ++l)
		if (T.wg < P.lv[l].n_tiles) lmax = l;
	const bool with_top = n_top > 0;
	if (!co_static<real, MEM>(P, T, st, lmax, state, !with_top)) return;
	(void)co_cycle<real, MEM, TAGGED>(P, T, st, R, dep, P.tag, lmax, !with_top, with_top);
}

/// The small levels in ONE workgroup of 16 waves: down from level `first` to the single-tile level, the coarsest solve,
/// and up again to `first`, with a workgroup barrier between the phases (their data sits in L2).
#define MG_TAIL_WAVES 8  // 512 threads: a thread per cell in the single-tile chain; tail levels of several tiles take a wave per tile
template <typename real> struct MgTail {
	MgLv<real> lv[MG_MAX_LEVELS];
	int first, last, nsw, inner;
	int chain;  // first level of the trailing run of single-tile levels handled by single_tile_chain (== last: none but the coarsest)
};
#define MG_CHAIN_MAX 4
/// The trailing levels that consist of ONE active tile each (the last two at every BASELINE size), down and up with the
/// whole workgroup on one tile: a thread per cell (512) for the cell-wise phases, a thread per cell of one colour (256)
/// for the Gauss-Seidel half sweeps, a workgroup barrier between them; right-hand sides, iterates and results stay in LDS.
/// A single tile has no active neighbours, so the ring of the halo block stays zero. One wave doing all of this alone
/// (4 cells per lane and colour, 24 dependent half sweeps per V-cycle) spent 22 of the tail's 35 us on instruction latency
/// (s_memtime stamps). Same operations in the same order as presmooth_tile / residual_restrict_tile / coarsest_tile /
/// prolong_postsmooth_tile: bit-identical results (tested against LFA_MG_NO_CHAIN=1).
template <typename real>
__device__ inline void single_tile_chain(const MgTail<real> &T, real *H, real (*cb)[512], real (*cx)[512], real (*cy)[512], real *R,
                                         uint8_t *cab_) {
	uint8_t (*cab)[512] = (uint8_t (*)[512])cab_;
	const int t = threadIdx.x, s1 = T.chain, n = T.last - s1 + 1;
	const bool cellwise = t < 512, colourwise = t < 256;
	// cell of a thread in the cell-wise phases, and its halo index
	const int cell = t & 511, px = cell & 7, py = (cell >> 3) & 7, pz = cell >> 6;
	const int hi = (px + 1) + 10 * (py + 1) + 100 * (pz + 1);
	// cell of a thread in a half sweep of colour c: column (qx, qy), z = 2 j + ((qx + qy + c) & 1)
	const int qx = t & 7, qy = (t >> 3) & 7, qj = (t >> 6) & 3;
	auto half_sweep = [&](int k, int colour) {
		if (colourwise) {
			const int z = 2 * qj + ((qx + qy + colour) & 1), c = qx + 8 * qy + 64 * z;
			gs_cell<real>(H, cab[k][c], cb[k][c], (qx + 1) + 10 * (qy + 1) + 100 * (z + 1));
		}
		__syncthreads();
	};
	for (int i = t; i < LFA_HALO_CELLS; i += MG_TAIL_WAVES * 64) H[i] = (real)0;
	if (cellwise)
		for (int k = 0; k < n; ++k) {
			const MgLv<real> &L = T.lv[s1 + k];
			const size_t base = (size_t)L.tiles[0] * 512;
			cab[k][cell] = L.abits[base + cell];
			cb[k][cell] = k == 0 ? L.b[base + cell] : (real)0;
		}
	__syncthreads();
	for (int k = 0; k + 1 < n; ++k) {  // down
		const MgLv<real> &L = T.lv[s1 + k];
		if (cellwise) H[hi] = (real)0;
		__syncthreads();
		for (int it = 0; it < T.inner; ++it) {
			half_sweep(k, 0);
			half_sweep(k, 1);
		}
		if (cellwise) {
			cx[k][cell] = H[hi];
			const uint32_t a = cab[k][cell];
			real r = (real)0;
			if (a & AB_UNKNOWN) {
				const real F = (a & AB_FLUID) ? (real)1 : (real)0;
				real val = (real)(a & 7) * H[hi];
				val = madd01(-F, H[hi - 1], val);
				val = madd01(-F, H[hi - 10], val);
				val = madd01(-F, H[hi - 100], val);
				val = madd01(-(real)((a >> 3) & 1), H[hi + 1], val);
				val = madd01(-(real)((a >> 4) & 1), H[hi + 10], val);
				val = madd01(-(real)((a >> 5) & 1), H[hi + 100], val);
				r = cb[k][cell] - val;
			}
			R[cell] = r;
		}
		__syncthreads();
		if (t < 64) {  // one coarse cell each: its 8 children in the order of the pair sums and the two shuffle steps
			const int X = t & 3, Y = (t >> 2) & 3, Z = t >> 4;
			auto pair = [&](int x, int y) { real p = R[x + 8 * y + 64 * (2 * Z)]; p += R[x + 8 * y + 64 * (2 * Z + 1)]; return p; };
			real v = pair(2 * X, 2 * Y);
			v += pair(2 * X + 1, 2 * Y);
			real w = pair(2 * X, 2 * Y + 1);
			w += pair(2 * X + 1, 2 * Y + 1);
			v += w;
			int tx, ty, tz;
			tile_coords(L.g, L.tiles[0], tx, ty, tz);
			cb[k + 1][((tz & 1) * 4 + Z) * 64 + ((ty & 1) * 4 + Y) * 8 + (tx & 1) * 4 + X] = (real)0.5 * v;
		}
		__syncthreads();
	}
	{  // coarsest level: nsw sweeps red->black, nsw black->red from zero
		const int k = n - 1;
		if (cellwise) H[hi] = (real)0;
		__syncthreads();
		for (int q = 0; q < 2 * T.nsw; ++q) {
			const int fc = q < T.nsw ? 0 : 1;
			half_sweep(k, fc);
			half_sweep(k, fc ^ 1);
		}
		if (cellwise) {
			cy[k][cell] = H[hi];
			if (n == 1) T.lv[T.last].y[(size_t)T.lv[T.last].tiles[0] * 512 + cell] = H[hi];
		}
		__syncthreads();
	}
	for (int k = n - 2; k >= 0; --k) {  // up
		const MgLv<real> &L = T.lv[s1 + k];
		if (cellwise) {
			int tx, ty, tz;
			tile_coords(L.g, L.tiles[0], tx, ty, tz);
			real v = cx[k][cell];
			const int X = (tx * 8 + px) >> 1, Y = (ty * 8 + py) >> 1, Z = (tz * 8 + pz) >> 1;  // parent cell in the next level's tile
			if (cab[k][cell] & AB_UNKNOWN) v += cy[k + 1][(X & 7) | ((Y & 7) << 3) | ((Z & 7) << 6)];
			H[hi] = v;
		}
		__syncthreads();
		for (int it = 0; it < T.inner; ++it) {
			half_sweep(k, 1);
			half_sweep(k, 0);
		}
		if (cellwise) {
			cy[k][cell] = H[hi];
			if (k == 0) L.y[(size_t)L.tiles[0] * 512 + cell] = H[hi];  // the level above (several tiles, global memory) reads this
		}
		__syncthreads();
	}
}
template <typename real>
__global__ void __launch_bounds__(MG_TAIL_WAVES * 64) k_mg_tail(MgTail<real> T, const int *state) {
	__shared__ real halo[MG_TAIL_WAVES][LFA_HALO_CELLS];
	__shared__ real cb[MG_CHAIN_MAX][512], cx[MG_CHAIN_MAX][512], cy[MG_CHAIN_MAX][512];  // level arrays of the single-tile chain
	__shared__ uint8_t cab[MG_CHAIN_MAX][512];
	if (state[0] >= 0) return;
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *h = halo[wid];
	for (int l = T.first; l < T.chain; ++l) {
		const MgLv<real> &L = T.lv[l];
		for (int i = lane; i < LFA_HALO_CELLS; i += 64) h[i] = (real)0;
		for (int slot = wid; slot < L.n_tiles; slot += MG_TAIL_WAVES) presmooth_tile<real>(L, slot, h, lane, T.inner);
		__syncthreads();
		for (int slot = wid; slot < L.n_tiles; slot += MG_TAIL_WAVES) residual_restrict_tile<real>(L, T.lv[l + 1].g, T.lv[l + 1].b, slot, h, lane);
		__syncthreads();
	}
	single_tile_chain<real>(T, halo[0], cb, cx, cy, halo[1], &cab[0][0]);
	for (int l = T.chain - 1; l >= T.first; --l) {
		const MgLv<real> &L = T.lv[l];
		for (int slot = wid; slot < L.n_tiles; slot += MG_TAIL_WAVES) {
			real bb[8];
			prolong_postsmooth_tile<real>(L, T.lv[l + 1].g, T.lv[l + 1].y, slot, h, lane, bb, T.inner);
			const size_t base = (size_t)L.tiles[slot] * 512;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) L.y[base + zz * 64 + lane] = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)];
		}
		__syncthreads();
	}
}


/// Minimum waves per SIMD the two streaming kernels of the finest level are compiled for (their software pipelines hold a whole
/// tile's loads in registers: unconstrained they take 134 / 154 VGPRs = 3 waves per SIMD). LFA_MG_MW_A / LFA_MG_MW_U select
/// another instantiation for A/B runs.
#ifndef MG_MW_DEFAULT_A
#define MG_MW_DEFAULT_A 3
#endif  // (C4: 65 -> 61 us; 5 / 6 waves spill the pipeline registers: 117 / 149 us)
#ifndef MG_MW_DEFAULT_U
#define MG_MW_DEFAULT_U 3
#endif  // (C4: 54 us; 4 / 5 / 6: 61 / 101 / 142 us)
template <typename real, typename... Args> static void launch_axpy_presmooth(int G, hipStream_t st, Args... a) {
	hipLaunchKernelGGL((k_mg_axpy_presmooth<real, MG_MW_DEFAULT_A>), dim3(G), dim3(256), 0, st, a...);
}
template <typename real, typename... Args> static void launch_up0(int G, hipStream_t st, Args... a) {
	hipLaunchKernelGGL((k_mg_prolong_postsmooth<real, true, MG_MW_DEFAULT_U>), dim3(G), dim3(256), 0, st, a...);
}
int mg_grid(int n_tiles) { return pcg_grid(n_tiles); }
/// Slab mode of the hierarchy. A one-rank communicator needs none of it; LFA_MG_DIST_SINGLE=1 runs it anyway (tests: the array
/// all-reduces then go through the real transport).
bool mg_dist(const lfa_sim *s) { return s->dist && (s->dist->nranks > 1 || s->knobs.mg_dist_single); }
}  // namespace

// ================================================================================================= host side
// Kernels whose workgroups wait for each other (k_mg_coarse) need all of them resident together; `fits` below sizes
// one launch against the whole device. Independent handles on one GPU - several simulations of one host process, each on its own
// thread and stream, as the C ABI allows (include/libfluid_amd.h: "different handles are independent"; the Maya host holds one
// fluid node per simulated object, plugins/maya/nodes/grid_node.cpp:256) - could have two such launches in flight, each with part
// of its workgroups resident and waiting for the rest. While more than one handle is alive on a device every such launch
// therefore waits (an event, on the GPU: no host thread blocks) for the previous one of ANY handle. One handle alone pays nothing.
// Other PROCESSES on the same GPU cannot be chained; for them - and for anything unforeseen - every wait is bounded (co_wait).
namespace {
struct CoGate {
	std::mutex m;
	hipEvent_t ev[2] = {nullptr, nullptr};
	unsigned n = 0;       // launches chained so far
	int handles = 0;
};
CoGate g_co_gate[64];
/// Around a launch of a waiting kernel on s->stream: constructor = chain behind the previous one, done() = publish this one.
struct CoGateScope {
	CoGate &g;
	lfa_sim *s;
	bool on = false;
	int rc = LFA_OK;
	explicit CoGateScope(lfa_sim *s_) : g(g_co_gate[s_->device & 63]), s(s_) {
		g.m.lock();
		on = g.handles > 1;
		if (!on) {
			g.m.unlock();
			return;
		}
		for (hipEvent_t &e : g.ev)
			if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = LFA_E_HIP;
		if (rc == LFA_OK && g.n > 0 && hipStreamWaitEvent(s->stream, g.ev[(g.n - 1) & 1], 0) != hipSuccess) rc = LFA_E_HIP;
	}
	int done() {
		if (!on) return rc;
		if (rc == LFA_OK && hipEventRecord(g.ev[g.n & 1], s->stream) != hipSuccess) rc = LFA_E_HIP;
		++g.n;
		on = false;
		g.m.unlock();
		return rc;
	}
	~CoGateScope() {
		if (on) g.m.unlock();
	}
};
}  // namespace
void lfa_co_gate_handle(int device, int delta) {
	CoGate &g = g_co_gate[device & 63];
	std::lock_guard<std::mutex> lk(g.m);
	g.handles += delta;
}

void lfa_mg_free(lfa_sim *s) {
	if (!s->mg) return;
	for (auto &L : s->mg->lv) {
		void *ptrs[] = {L.tiles, L.nbr, L.ctype, L.abits, L.x, L.b, L.y, L.flag, L.prev_flag, L.ready, L.xq};
		for (void *p : ptrs)
			if (p) (void)hipFree(p);
	}
	if (s->mg->l1_dirty) (void)hipFree(s->mg->l1_dirty);
	if (s->mg->l1_has_fluid) (void)hipFree(s->mg->l1_has_fluid);
	if (s->mg->slot0) (void)hipFree(s->mg->slot0);
	if (s->mg->top_ring) (void)hipFree(s->mg->top_ring);
	if (s->mg->top_buf) (void)hipFree(s->mg->top_buf);
	if (s->mg->counts) (void)hipFree(s->mg->counts);
	delete s->mg;
	s->mg = nullptr;
}

/// Builds the hierarchy for the current unknown set (after lfa_build_rhs: A bytes and types of the step are on the device).
template <typename real> static int mg_setup_t(lfa_sim *s) {
	if (!s->mg) s->mg = new lfa_mg();
	lfa_mg &M = *s->mg;
	// level grids: halve until one tile holds the whole level
	std::vector<GridDims> gs{s->g};
	while ((gs.back().nx > 8 || gs.back().ny > 8 || gs.back().nz > 8) && (int)gs.size() < MG_MAX_LEVELS) {
		GridDims c, f = gs.back();
		c.nx = (f.nx + 1) / 2; c.ny = (f.ny + 1) / 2; c.nz = (f.nz + 1) / 2;
		c.ntx = (c.nx + 7) / 8; c.nty = (c.ny + 7) / 8; c.ntz = (c.nz + 7) / 8;
		c.nt = c.ntx * c.nty * c.ntz;
		gs.push_back(c);
	}
	const int nl = (int)gs.size();
	const bool dist = mg_dist(s);
	const int d_max = s->knobs.mg_dist_levels > 0 ? std::min(s->knobs.mg_dist_levels, MG_DIST_MAX) : MG_DIST_MAX;
	const int D = dist ? std::min({d_max, s->slab_align + 1, nl - 1}) : 0;
	if (dist && D < 1) return lfa_fail(s, LFA_E_INVALID, "multigrid on slabs needs at least two levels");
	const bool realloc_all = M.n_levels != nl || M.elem != sizeof(real) || M.n_dist != D ||
	                         (nl > 1 && (M.lv[1].g.nx != gs[1].nx || M.lv[1].g.ny != gs[1].ny || M.lv[1].g.nz != gs[1].nz));
	if (realloc_all) {
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		for (auto &L : M.lv) {
			void *ptrs[] = {L.tiles, L.nbr, L.ctype, L.abits, L.x, L.b, L.y, L.flag, L.prev_flag, L.ready, L.xq};
			for (void *p : ptrs)
				if (p) LFA_HIP(s, hipFree(p));
			L = lfa_mg_level();
		}
		M.n_levels = nl;
		M.elem = sizeof(real);
		M.n_dist = D;
		M.top_packed = dist && s->knobs.mg_dist_levels > 0;
		M.host_tiles.clear();
		M.host_top.clear();
		if (M.l1_dirty) LFA_HIP(s, hipFree(M.l1_dirty));
		M.l1_dirty = nullptr;
		M.solid_epoch = 0;  // forces a full pass
		if (nl > 1) LFA_HIP(s, hipMalloc(&M.l1_dirty, (size_t)gs[1].nt));
		if (M.l1_has_fluid) LFA_HIP(s, hipFree(M.l1_has_fluid));
		M.l1_has_fluid = nullptr;
		if (M.slot0) LFA_HIP(s, hipFree(M.slot0));
		M.slot0 = nullptr;
		if (nl > 1) {
			LFA_HIP(s, hipMalloc(&M.l1_has_fluid, (size_t)gs[1].nt));
			LFA_HIP(s, hipMemsetAsync(M.l1_has_fluid, 0, (size_t)gs[1].nt, s->stream));  // (a tile no pass has visited has no flagged child either)
		}
		if (!M.counts) LFA_HIP(s, hipMalloc(&M.counts, MG_MAX_LEVELS * 4));
		for (int l = 0; l < nl; ++l) {
			lfa_mg_level &L = M.lv[l];
			L.g = gs[l];
			L.ncp = (size_t)gs[l].nt * 512;
			if (!dist) {
				// single domain: tile lists and neighbour tables are built on the device (mg_device_lists); sized for every tile
				LFA_HIP(s, hipMalloc(&L.flag, (size_t)(gs[l].nt + 1) * 4));
				LFA_HIP(s, hipMalloc(&L.prev_flag, (size_t)(gs[l].nt + 1) * 4));
				LFA_HIP(s, hipMemsetAsync(L.prev_flag, 0, (size_t)(gs[l].nt + 1) * 4, s->stream));
				LFA_HIP(s, hipMalloc(&L.tiles, (size_t)gs[l].nt * 4));
				LFA_HIP(s, hipMalloc(&L.nbr, (size_t)gs[l].nt * MG_NBR_STRIDE * 4));
				L.cap_tiles = (size_t)gs[l].nt;
			}
			if (l == 0) {  // k_pcg_small: ready flags of the finest level (3 x tiles) followed by its search-direction flags (1 x tiles)
				LFA_HIP(s, hipMalloc(&L.ready, (size_t)4 * gs[0].nt * sizeof(unsigned)));
				LFA_HIP(s, hipMemsetAsync(L.ready, 0, (size_t)4 * gs[0].nt * sizeof(unsigned), s->stream));
				continue;
			}
			LFA_HIP(s, hipMalloc(&L.ctype, L.ncp));
			LFA_HIP(s, hipMalloc(&L.abits, L.ncp));
			LFA_HIP(s, hipMalloc(&L.x, L.ncp * sizeof(real)));
			LFA_HIP(s, hipMalloc(&L.b, L.ncp * sizeof(real)));
			LFA_HIP(s, hipMalloc(&L.y, L.ncp * sizeof(real)));
			LFA_HIP(s, hipMalloc(&L.ready, (size_t)3 * gs[l].nt * sizeof(unsigned)));
			LFA_HIP(s, hipMemsetAsync(L.ready, 0, (size_t)3 * gs[l].nt * sizeof(unsigned), s->stream));
			LFA_HIP(s, hipMemsetAsync(L.ctype, MT_SOLID, L.ncp, s->stream));
			if (!dist) {  // the device path clears only what a departed tile leaves behind
				LFA_HIP(s, hipMemsetAsync(L.x, 0, L.ncp * sizeof(real), s->stream));
				LFA_HIP(s, hipMemsetAsync(L.b, 0, L.ncp * sizeof(real), s->stream));
				LFA_HIP(s, hipMemsetAsync(L.y, 0, L.ncp * sizeof(real), s->stream));
				LFA_HIP(s, hipMemsetAsync(L.abits, 0, L.ncp, s->stream));
			}
		}
	}
	for (int l = 0; l < nl; ++l) {
		M.lv[l].lo_layer = dist ? s->slab_lo >> l : 0;
		M.lv[l].hi_layer = (!dist || s->slab_hi == s->g.ntz) ? gs[l].ntz : s->slab_hi >> l;
	}
	// the part of a level's arrays this rank touches: its own tile layers and one ghost layer per side (distributed levels)
	auto layer_range = [&](int l, int &t0, int &t1) {
		const lfa_mg_level &L = M.lv[l];
		const int per = gs[l].ntx * gs[l].nty;
		if (dist && l < D) {
			t0 = std::max(L.lo_layer - 1, 0) * per;
			t1 = std::min(L.hi_layer + 1, gs[l].ntz) * per;
		} else {
			t0 = 0;
			t1 = gs[l].nt;
		}
	};
	bool same_tiles = false;
	if (!dist) {
		// ---- single domain: flags -> scan -> compact per level on the device, ONE read-back of the counts, then the neighbour
		// tables. (The host version below - download of the particle tiles, sorted parent lists, binary-searched neighbours,
		// an upload per level - cost 2.8 ms per step at C4 once the dam moves and the tile set changes every step.)
		// the levels from `small` on have at most MG_SMALL_NT tiles in their grid: their lists / tables come from three launches in all
		int small = nl;
		for (int l = nl - 1; l >= 1 && gs[l].nt <= MG_SMALL_NT; --l) small = l;
		MgSmall SP;
		SP.first = small;
		SP.last = nl - 1;
		for (int l = std::max(small - 1, 0); l < nl; ++l) {
			const lfa_mg_level &L = M.lv[l];
			SP.g[l] = gs[l];
			SP.flag[l] = L.flag; SP.prev[l] = L.prev_flag; SP.tiles[l] = L.tiles; SP.nbr[l] = L.nbr;
			SP.x[l] = L.x; SP.b[l] = L.b; SP.y[l] = L.y; SP.abits[l] = L.abits;
		}
		// level 1's types first: its active set is the parents of the particle tiles THAT HOLD AN UNKNOWN of level 1 (round 6)
		const uint8_t *l1_has = (nl > 1 && !s->knobs.mg_no_prune) ? (const uint8_t *)M.l1_has_fluid : (const uint8_t *)nullptr;
		if (nl > 1) {
			hipLaunchKernelGGL(k_mg_types_from_fine_dirty, dim3(std::min(gs[1].nt, 8192)), dim3(256), 0, s->stream, gs[0], gs[1],
			                   (const uint32_t *)s->tile_flag, (const uint32_t *)s->cell_count, (const uint8_t *)s->ctype,
			                   (const uint8_t *)s->solid, M.lv[1].ctype, M.l1_dirty, M.solid_epoch != s->solid_epoch ? 1 : 0, M.l1_has_fluid);
			M.solid_epoch = s->solid_epoch;
		}
		SP.l1_has_unknown = l1_has;
		// closed tiles are solved on their own (k_mg_solve_closed): level 0 is what is left of the particle tiles
		const bool closed_out = nl > 1 && !s->dist && !s->knobs.mg_no_closed && s->tile_closed;
		M.closed_out = closed_out;
		if (closed_out && !M.slot0) LFA_HIP(s, hipMalloc(&M.slot0, (size_t)gs[0].nt * 4));
		hipLaunchKernelGGL(k_mg_flag_level0, dim3((gs[0].nt + 255) / 256), dim3(256), 0, s->stream, (const int *)s->tile_pslot,
		                   closed_out ? (const uint8_t *)s->tile_closed : (const uint8_t *)nullptr, M.lv[0].flag, gs[0].nt);
		for (int l = 1; l < small; ++l)
			hipLaunchKernelGGL(k_mg_flag_parents, dim3((gs[l].nt + 255) / 256), dim3(256), 0, s->stream, gs[l - 1], gs[l],
			                   (const uint32_t *)M.lv[l - 1].flag, M.lv[l].flag, l == 1 ? l1_has : (const uint8_t *)nullptr);
		LFA_LAUNCH_CHECK(s);
		for (int l = 0; l < small; ++l) {
			lfa_mg_level &L = M.lv[l];
			if (l == 0 && !closed_out) {  // the binning's own list
				LFA_HIP(s, hipMemcpyAsync(L.tiles, s->ptiles, (size_t)s->n_ptiles * 4, hipMemcpyDeviceToDevice, s->stream));
				continue;
			}
			if (l == 0) {
				LFA_TRY(lfa_exclusive_scan_u32(s, L.flag, s->tile_scan, (size_t)gs[0].nt, M.counts));
				hipLaunchKernelGGL(k_mg_compact_slots, dim3((gs[0].nt + 255) / 256), dim3(256), 0, s->stream, (const uint32_t *)L.flag,
				                   (const uint32_t *)s->tile_scan, L.tiles, M.slot0, gs[0].nt);
				LFA_LAUNCH_CHECK(s);
				continue;
			}
			LFA_TRY(lfa_exclusive_scan_u32(s, L.flag, s->tile_scan, (size_t)gs[l].nt, M.counts + l));  // gs[l].nt <= nt / 8
			hipLaunchKernelGGL(k_mg_compact, dim3((gs[l].nt + 255) / 256), dim3(256), 0, s->stream, (const uint32_t *)L.flag,
			                   (const uint32_t *)s->tile_scan, L.tiles, gs[l].nt);
			LFA_LAUNCH_CHECK(s);
		}
		if (small < nl) {
			hipLaunchKernelGGL(k_mg_small_lists, dim3(1), dim3(1024), 0, s->stream, SP, M.counts);
			LFA_LAUNCH_CHECK(s);
		}
		uint32_t *hc = s->h_pinned + 64;
		if (nl > 1) {
			LFA_HIP(s, hipMemcpyAsync(hc, M.counts, (size_t)nl * 4, hipMemcpyDeviceToHost, s->stream));
			LFA_HIP(s, hipStreamSynchronize(s->stream));
		}
		if (small < nl) {
			// (the children's set of the first small level: level small - 1 swaps its flag arrays in the loop below - after it the
			// set of this set-up is its prev_flag; within the small levels the sets are the arrays k_mg_small_lists has just written)
			for (int l = small; l < nl; ++l) M.lv[l].n_tiles = (int)hc[l];
		}
		for (int l = 0; l < small; ++l) {
			lfa_mg_level &L = M.lv[l];
			L.n_tiles = (l == 0 && !closed_out) ? s->n_ptiles : (int)hc[l];
			if (L.n_tiles) {
				hipLaunchKernelGGL(k_mg_build_nbr, dim3((L.n_tiles + 255) / 256), dim3(256), 0, s->stream, (const int *)L.tiles, L.n_tiles,
				                   gs[l], (const uint32_t *)L.flag, L.nbr, gs[l > 0 ? l - 1 : 0],
				                   l > 0 ? (const uint32_t *)M.lv[l - 1].prev_flag : (const uint32_t *)nullptr);  // (already swapped: the current set)
				LFA_LAUNCH_CHECK(s);
			}
			if (l >= 1) {
				hipLaunchKernelGGL(k_mg_clear_departed<real>, dim3(std::min(gs[l].nt, 4096)), dim3(256), 0, s->stream,
				                   (const uint32_t *)L.prev_flag, (const uint32_t *)L.flag, gs[l].nt, (real *)L.x, (real *)L.b, (real *)L.y,
				                   L.abits);
				LFA_LAUNCH_CHECK(s);
			}
			std::swap(L.flag, L.prev_flag);  // prev_flag = this set-up's set; flag is rewritten by the next one
		}
		if (small < nl) {
			SP.flag[small - 1] = M.lv[small - 1].prev_flag;  // (swapped just above: the current set of the level below)
			hipLaunchKernelGGL(k_mg_small_tables<real>, dim3(16, nl - small), dim3(256), 0, s->stream, SP, (const uint32_t *)M.counts);
			LFA_LAUNCH_CHECK(s);
			for (int l = small; l < nl; ++l) std::swap(M.lv[l].flag, M.lv[l].prev_flag);
		}
		same_tiles = true;  // nothing left for the host path / the whole-array clears below
	}
	// active tiles per level on the host: a tile is active if one of its child tiles is
	if (dist) {
		const int n_all = dist ? s->n_ptiles_all : s->n_ptiles;
		std::vector<int> tiles(n_all);  // slabs: with the neighbours' particle tiles of the adjacent layers (ascending ids all the same)
		LFA_HIP(s, hipMemcpyAsync(tiles.data(), dist ? s->ptiles_all : s->ptiles, (size_t)n_all * 4, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		// The tile structure changes far less often than its contents: when the particle tiles are the ones of the last solve, the
		// lists and neighbour tables on the device are kept (and no vector can hold values of a tile that has left the set).
		same_tiles = !realloc_all && tiles == M.host_tiles;
		std::vector<std::vector<int>> act(nl);
		auto parents = [&](int l) {  // active tiles of level l from those of level l - 1
			const GridDims &f = gs[l - 1], &c = gs[l];
			std::vector<int> &o = act[l];
			o.clear();
			o.reserve(act[l - 1].size() / 4 + 8);
			for (int t : act[l - 1]) {
				int tx, ty, tz;
				tile_coords(f, t, tx, ty, tz);
				o.push_back((tx >> 1) + c.ntx * ((ty >> 1) + c.nty * (tz >> 1)));
			}
			std::sort(o.begin(), o.end());
			o.erase(std::unique(o.begin(), o.end()), o.end());
		};
		if (dist || !same_tiles) {
			if (dist) act[0].assign(tiles.begin() + s->p_off, tiles.begin() + s->p_off + s->n_ptiles);
			else act[0] = tiles;
			for (int l = 1; l < nl; ++l) {
				parents(l);
				if (dist && l == D) {
					// the first replicated level: union of every rank's active tiles (one flag byte per tile, max all-reduce)
					std::vector<uint8_t> flag((size_t)gs[l].nt, 0);
					for (int t : act[l]) flag[t] = 1;
					LFA_TRY(lfa_dist_ensure_xbuf(s, 0, flag.size()));
					LFA_HIP(s, hipMemcpyAsync(s->xbuf[0], flag.data(), flag.size(), hipMemcpyHostToDevice, s->stream));
					LFA_TRY(s->dist->allreduce_buf(s, s->xbuf[0], flag.size(), LFA_RED_U8, true));
					LFA_HIP(s, hipMemcpyAsync(flag.data(), s->xbuf[0], flag.size(), hipMemcpyDeviceToHost, s->stream));
					LFA_HIP(s, hipStreamSynchronize(s->stream));
					act[l].clear();
					for (int t = 0; t < gs[l].nt; ++t)
						if (flag[t]) act[l].push_back(t);
					same_tiles = same_tiles && act[l] == M.host_top;
					if (M.top_packed) {
						// the tiles whose TYPES every rank needs: the active ones and their face neighbours (k_mg_abits looks one cell
						// across a tile face; the coarser levels' types follow from these)
						std::vector<int> ring;
						ring.reserve(act[l].size() * 3);
						const GridDims &g = gs[l];
						for (int t : act[l]) {
							int tx, ty, tz;
							tile_coords(g, t, tx, ty, tz);
							ring.push_back(t);
							if (tx > 0) ring.push_back(t - 1);
							if (tx + 1 < g.ntx) ring.push_back(t + 1);
							if (ty > 0) ring.push_back(t - g.ntx);
							if (ty + 1 < g.nty) ring.push_back(t + g.ntx);
							if (tz > 0) ring.push_back(t - g.ntx * g.nty);
							if (tz + 1 < g.ntz) ring.push_back(t + g.ntx * g.nty);
						}
						std::sort(ring.begin(), ring.end());
						ring.erase(std::unique(ring.begin(), ring.end()), ring.end());
						if (ring != M.host_ring) {
							if (ring.size() > M.top_ring_cap) {
								if (M.top_ring) LFA_HIP(s, hipFree(M.top_ring));
								M.top_ring = nullptr;
								M.top_ring_cap = ring.size() + ring.size() / 4 + 64;
								LFA_HIP(s, hipMalloc(&M.top_ring, M.top_ring_cap * 4));
							}
							M.host_ring.swap(ring);  // (kept: the copy below reads it asynchronously)
							if (!M.host_ring.empty())
								LFA_HIP(s, hipMemcpyAsync(M.top_ring, M.host_ring.data(), M.host_ring.size() * 4, hipMemcpyHostToDevice, s->stream));
							LFA_HIP(s, hipStreamSynchronize(s->stream));
						}
					}
				}
			}
		}
		if (!same_tiles) {
			for (int l = 0; l < nl; ++l) {
				lfa_mg_level &L = M.lv[l];
				const GridDims &g = gs[l];
				const std::vector<int> &a = act[l];
				L.n_tiles = (int)a.size();
				if (a.size() > L.cap_tiles) {
					if (L.tiles) LFA_HIP(s, hipFree(L.tiles));
					if (L.nbr) LFA_HIP(s, hipFree(L.nbr));
					L.tiles = L.nbr = nullptr;
					L.cap_tiles = a.size() + a.size() / 4 + 16;
					LFA_HIP(s, hipMalloc(&L.tiles, L.cap_tiles * 4));
					LFA_HIP(s, hipMalloc(&L.nbr, L.cap_tiles * MG_NBR_STRIDE * 4));
				}
				if (a.empty()) continue;
				// neighbour tiles by binary search in the sorted list. Across a slab face of a distributed level the neighbour belongs
				// to another rank: on the finest level it is active if it is one of that rank's particle tiles, on the others it always
				// counts as active (its slice arrives with the layer exchange; an inactive tile holds zeros and no unknowns)
				std::vector<int> nbr(a.size() * MG_NBR_STRIDE, 0);
				const int sy = g.ntx, sz = g.ntx * g.nty;
				for (size_t i = 0; i < a.size(); ++i) {
					int tx, ty, tz;
					tile_coords(g, a[i], tx, ty, tz);
					const int cand[6] = {tx > 0 ? a[i] - 1 : -1,           tx + 1 < g.ntx ? a[i] + 1 : -1, ty > 0 ? a[i] - sy : -1,
					                     ty + 1 < g.nty ? a[i] + sy : -1, tz > 0 ? a[i] - sz : -1,          tz + 1 < g.ntz ? a[i] + sz : -1};
					for (int k = 0; k < 6; ++k) {
						int v = -1;
						const int nz = tz + (k == 4 ? -1 : (k == 5 ? 1 : 0));
						if (cand[k] < 0) v = -1;
						else if (dist && l < D && (nz < L.lo_layer || nz >= L.hi_layer))
							v = (l > 0 || std::binary_search(tiles.begin(), tiles.end(), cand[k])) ? cand[k] : -1;
						else v = std::binary_search(a.begin(), a.end(), cand[k]) ? cand[k] : -1;
						nbr[i * MG_NBR_STRIDE + k] = v;
					}
					nbr[i * MG_NBR_STRIDE + 6] = a[i];
					int mask = 0;  // active child tiles (k_mg_coarse waits for their shares of this tile's right-hand side)
					if (l > 0)
						for (int k = 0; k < 8; ++k) {
							const int cx = 2 * tx + (k & 1), cy = 2 * ty + ((k >> 1) & 1), cz = 2 * tz + (k >> 2);
							if (cx < gs[l - 1].ntx && cy < gs[l - 1].nty && cz < gs[l - 1].ntz &&
							    std::binary_search(act[l - 1].begin(), act[l - 1].end(), cx + gs[l - 1].ntx * (cy + gs[l - 1].nty * cz)))
								mask |= 1 << k;
						}
					nbr[i * MG_NBR_STRIDE + 7] = mask;
				}
				LFA_HIP(s, hipMemcpyAsync(L.tiles, a.data(), a.size() * 4, hipMemcpyHostToDevice, s->stream));
				LFA_HIP(s, hipMemcpyAsync(L.nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice, s->stream));
				LFA_HIP(s, hipStreamSynchronize(s->stream));  // the host vectors go out of scope
			}
			M.host_tiles = tiles;
			if (dist) M.host_top = act[D];
		}
	}
	// types and operators of the coarse levels
	for (int l = 1; l < nl; ++l) {
		lfa_mg_level &L = M.lv[l];
		int t0, t1, zlo = 0, zhi = INT_MAX;
		layer_range(l, t0, t1);
		if (dist && l == D) {
			// the rank's share of the first replicated level: children outside its own layers count as walls (the neutral type),
			// the shares are combined by a max all-reduce (air > fluid > wall = the coarsening rule)
			const lfa_mg_level &F = M.lv[l - 1];
			const int per = gs[l].ntx * gs[l].nty;
			LFA_HIP(s, hipMemsetAsync(L.ctype, MT_SOLID, L.ncp, s->stream));
			t0 = (F.lo_layer >> 1) * per;
			t1 = std::min((F.hi_layer + 1) >> 1, gs[l].ntz) * per;
			zlo = F.lo_layer * 8;
			zhi = F.hi_layer * 8;
		}
		const size_t c0 = (size_t)t0 * 512, count = (size_t)(t1 - t0) * 512;
		const unsigned grid = (unsigned)((count + 255) / 256);
		if (l == 1 && !dist) {
			// (done in front of the tile lists above)
		} else if (l == 1)
			hipLaunchKernelGGL(k_mg_types_from_fine, dim3(grid), dim3(256), 0, s->stream, gs[0], gs[1], c0, count, zlo, zhi,
			                   (const uint32_t *)s->tile_flag, (const uint32_t *)s->cell_count, (const uint8_t *)s->ctype,
			                   (const uint8_t *)s->solid, L.ctype);
		else
			hipLaunchKernelGGL(k_mg_types_coarsen, dim3(grid), dim3(256), 0, s->stream, gs[l - 1], gs[l], c0, count, zlo, zhi,
			                   (const uint8_t *)M.lv[l - 1].ctype, L.ctype);
		LFA_LAUNCH_CHECK(s);
		if (dist && l == D && M.top_packed) {
			// packed: the ring tiles' types cross the ranks; every other tile of the level is a WALL on every rank - the same
			// operator everywhere (a rank's own share alone would know more about its own layers than its peers do)
			const int nr = (int)M.host_ring.size();
			const size_t need = std::max((size_t)nr * 512, (size_t)1);
			if (need > M.top_buf_cap) {
				if (M.top_buf) LFA_HIP(s, hipFree(M.top_buf));
				M.top_buf = nullptr;
				M.top_buf_cap = need + need / 4 + 4096;
				LFA_HIP(s, hipMalloc(&M.top_buf, M.top_buf_cap));
			}
			if (nr) hipLaunchKernelGGL((k_mg_tiles_copy<uint8_t, true>), dim3(std::min(nr, 4096)), dim3(256), 0, s->stream, (const int *)M.top_ring, nr, L.ctype, (uint8_t *)M.top_buf);
			LFA_HIP(s, hipMemsetAsync(L.ctype, MT_SOLID, L.ncp, s->stream));
			LFA_TRY(s->dist->allreduce_buf(s, M.top_buf, (size_t)nr * 512, LFA_RED_U8, true));
			if (nr) hipLaunchKernelGGL((k_mg_tiles_copy<uint8_t, false>), dim3(std::min(nr, 4096)), dim3(256), 0, s->stream, (const int *)M.top_ring, nr, L.ctype, (uint8_t *)M.top_buf);
			LFA_LAUNCH_CHECK(s);
		} else if (dist && l == D)
			LFA_TRY(s->dist->allreduce_buf(s, L.ctype, L.ncp, LFA_RED_U8, true));
		// vectors of levels >= 1 are read where no tile of this solve writes (parents of ring cells): they must be zero there.
		// Tiles of the current set are rewritten by every V-cycle, so clearing is only needed when the set has changed.
		if (!same_tiles) {
			layer_range(l, t0, t1);
			const size_t o = (size_t)t0 * 512, n = (size_t)(t1 - t0) * 512;
			LFA_HIP(s, hipMemsetAsync((real *)L.x + o, 0, n * sizeof(real), s->stream));
			LFA_HIP(s, hipMemsetAsync((real *)L.b + o, 0, n * sizeof(real), s->stream));
			LFA_HIP(s, hipMemsetAsync((real *)L.y + o, 0, n * sizeof(real), s->stream));
			LFA_HIP(s, hipMemsetAsync(L.abits + o, 0, n, s->stream));
		}
		if (L.n_tiles) {
			hipLaunchKernelGGL(k_mg_abits, dim3(std::min(L.n_tiles, 4096)), dim3(256), 0, s->stream, (const int *)L.tiles, L.n_tiles, gs[l],
			                   (const uint8_t *)L.ctype, L.abits);
			LFA_LAUNCH_CHECK(s);
		}
	}
	// slabs: which cells across a slab face are unknowns (the up-sweep corrects ring values of unknowns only)
	for (int l = 0; dist && l < D; ++l) {
		if (l == 0) LFA_TRY(lfa_dist_exchange_slices(s, s->abits, 1));
		else LFA_TRY(lfa_dist_exchange_layer_slices(s, M.lv[l].abits, 1, gs[l].ntx * gs[l].nty, M.lv[l].lo_layer, M.lv[l].hi_layer));
	}
	return LFA_OK;
}

int lfa_mg_setup(lfa_sim *s) {
	return s->prm.pcg_dtype == LFA_PCG_F64 ? mg_setup_t<double>(s) : mg_setup_t<float>(s);
}

/// z = V(r) / scale and the partial sums of dot(z, r) (pcg_grid(n_ptiles) of them) into part_sigma.
/// Level 0 uses the solver's own vectors: b = r (vr), pre-smoothed iterate in vq (free between the AXPYs and the next
/// A s), result in vz. `level0_presmoothed`: vq already holds the pre-smoothed iterate (k_mg_axpy_presmooth).
#define MG_TAIL_TILES 1  // levels with at most this many tiles run inside k_mg_tail: with the cell-parallel kernels for the small
                         // levels only the single-tile ones are worth keeping there (C2 / C4: 1.52 / 8.63 ms per step, 8: 1.55 / 8.69)
#define MG_COARSEST_SWEEPS 2  // measured (moving dam, C2 / C3 / C4): 4: 14.0 / 15.05 / 16.15 iterations, 3: 14.05 / 15.0 / 16.0, 2: 14.1 / 15.05 / 15.85 - and 8 fewer half sweeps of latency per V-cycle
#define MG_CO_MAX_TILES 512  // levels of at most this many tiles run inside k_mg_coarse
#ifndef MG_RR_GRID
#define MG_RR_GRID 2048
#endif
#ifndef MG_TOP_MAX_TILES
#define MG_TOP_MAX_TILES 1024  // the level above joins the launch (in launch order) when it has at most this many tiles
#endif
#ifndef MG_TOP_MAX_WGS
#define MG_TOP_MAX_WGS 1024    // ... dealt to at most this many workgroups per phase
#endif
enum { MG_PART_PRE0 = 1, MG_PART_DOWN0 = 2, MG_PART_COARSE = 4, MG_PART_UP0 = 8, MG_PART_ALL = 15 };
template <typename real> static int mg_apply_t(lfa_sim *s, double *part_sigma, bool level0_presmoothed, int parts = MG_PART_ALL) {
	lfa_mg &M = *s->mg;
	const int nl = M.n_levels;
	// every level down to the single-tile one has active tiles - unless level 1 has none (nothing but spray: no coarse cell is an
	// unknown), which leaves the finest level's sweeps alone
	const int last = (nl > 1 && M.lv[1].n_tiles == 0 && !mg_dist(s)) ? 0 : nl - 1;
	auto lvl = [&](int l) {
		const lfa_mg_level &L = M.lv[l];
		return MgLv<real>{L.tiles, L.nbr, L.n_tiles, L.g, l == 0 ? (const uint8_t *)s->abits : (const uint8_t *)L.abits,
		                  l == 0 ? (real *)s->vr : (real *)L.b, l == 0 ? (real *)s->vq : (real *)L.x, l == 0 ? (real *)s->vz : (real *)L.y};
	};
	const real inv_scale = (real)(1.0 / s->a_scale);
	const int *st = (const int *)s->pcg_state;
	if (last == 0) {  // a single level (a grid of one tile, or no coarse unknown anywhere): the two sweeps alone
		const MgLv<real> L = lvl(0);
		if (!level0_presmoothed && (parts & MG_PART_PRE0))
			hipLaunchKernelGGL(k_mg_presmooth<real>, dim3(std::max(1, std::min((L.n_tiles + PCG_WAVES - 1) / PCG_WAVES, MG_RR_GRID))), dim3(256), 0,
			                   s->stream, L, st);
		if (parts & MG_PART_UP0) launch_up0<real>(mg_grid(L.n_tiles), s->stream, L, L.g, (const real *)nullptr, inv_scale, part_sigma, st);
		LFA_LAUNCH_CHECK(s);
		if (parts == MG_PART_ALL) {
			M.launches_per_cycle = 1 + 1;
			M.first_co = 0;
		}
		return LFA_OK;
	}
	int tail = last;  // first level handled by the single-workgroup tail
	while (tail > 1 && M.lv[tail - 1].n_tiles <= MG_TAIL_TILES) --tail;
	// slabs: levels < D run on the rank's own tiles with one slice per slab face exchanged where a stencil crosses it;
	// levels >= D are replicated (identical work on every rank), so the tail workgroup may only hold replicated levels
	const int D = mg_dist(s) ? M.n_dist : 0;
	tail = std::max(tail, D);
	// Default: every level of at most MG_CO_MAX_TILES tiles (C4: levels >= 2, C2: levels >= 1) runs inside ONE launch, k_mg_coarse
	// (phases chained by completion counters instead of kernel boundaries); the larger ones keep a launch per phase.
	// LFA_MG_NO_PERSIST=1, and a handle that has given a device-side wait up (co_disabled): a launch per phase on every level.
	const bool persist = !s->knobs.mg_no_persist && !s->co_disabled;
	// the tagged hand-off inside the launch (co_put / co_get): fp32 vectors, and the exchange arrays of the levels inside allocated
	bool tagged = persist && sizeof(real) == 4 && !s->knobs.mg_no_tagged;
	int top = -1;  // level fused in front of / behind the resident workgroups of k_mg_coarse
	if (persist) {
		// every workgroup of k_mg_coarse must be resident at the same time (they wait for each other): one workgroup per tile of
		// its first level, LDS per workgroup grows with the number of levels inside
		// (per device: a process may hold handles on several GPUs; what the queries return is kept - they sat in every V-cycle)
		struct DevInfo {
			int n_cu = 0;
			bool attr_set = false;
			int per_cu[2][MG_CO_MAX_LEVELS + 1];  // resident workgroups per CU by [tagged][levels inside], -1: not asked yet
		};
		static DevInfo info[64];
		static std::mutex info_mutex;
		DevInfo di;
		const int dslot = s->device & 63;
		{
			std::lock_guard<std::mutex> lk(info_mutex);
			DevInfo &d = info[dslot];
			if (!d.n_cu) {
				hipDeviceProp_t prop;
				d.n_cu = (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
				for (auto &row : d.per_cu)
					for (int &v : row) v = -1;
			}
			if (!d.attr_set) {
				LFA_HIP(s, hipFuncSetAttribute((const void *)k_mg_coarse<real, MemAgent, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
				LFA_HIP(s, hipFuncSetAttribute((const void *)k_mg_coarse<real, MemAgent, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
				d.attr_set = true;
			}
			di = d;
		}
		const int n_cu = di.n_cu;
		auto fits = [&](int first, bool tg) {
			const int nlev = last - first + 1;
			if (nlev > MG_CO_MAX_LEVELS) return false;
			const size_t lds = (size_t)nlev * sizeof(CoLevel<real>) + 512 * sizeof(real) + 64;
			int per_cu = di.per_cu[tg][nlev];  // by registers and LDS together
			if (per_cu < 0) {
				const void *fn = tg ? (const void *)k_mg_coarse<real, MemAgent, true> : (const void *)k_mg_coarse<real, MemAgent, false>;
				if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess) {
					(void)hipGetLastError();
					per_cu = 0;
				}
				std::lock_guard<std::mutex> lk(info_mutex);
				info[dslot].per_cu[tg][nlev] = di.per_cu[tg][nlev] = per_cu;
			}
			// ranks that share this GPU launch their k_mg_coarse at the same time: all of them must be resident together
			const size_t share = s->dist ? (size_t)std::max(1, s->dist->device_share) : 1;
			return M.lv[first].n_tiles <= MG_CO_MAX_TILES && (size_t)M.lv[first].n_tiles * share <= (size_t)per_cu * (size_t)n_cu;
		};
		auto plan = [&](bool tg) {
			int tl = last;
			while (tl > 1 && tl - 1 >= D && fits(tl - 1, tg)) --tl;
			return std::max(tl, std::max(D, 1));
		};
		tail = plan(tagged);
		auto ensure_xq = [&](int l, int sections) -> bool {
			lfa_mg_level &L = M.lv[l];
			if (L.xq && L.xq_sections >= sections) return true;
			if (L.ncp > ((size_t)1 << 26)) return false;
			if (L.xq) {
				if (hipStreamSynchronize(s->stream) != hipSuccess || hipFree(L.xq) != hipSuccess) return false;
				L.xq = nullptr;
			}
			if (hipMalloc(&L.xq, L.ncp * 8 * (size_t)sections) != hipSuccess) {
				(void)hipGetLastError();
				L.xq = nullptr;
				L.xq_sections = 0;
				return false;
			}
			L.xq_sections = sections;
			return hipMemsetAsync(L.xq, 0, L.ncp * 8 * (size_t)sections, s->stream) == hipSuccess;  // (tag 0 is never used)
		};
		if (tagged) {
			// 3 x 8 bytes per padded cell of every level inside (C4: level 2 is 128^3 cells = 50 MB); a level whose GRID is huge
			// although few of its tiles are active (a small pool in a 2048^3 domain) keeps the flags instead
			for (int l = tail; l <= last && tagged; ++l) tagged = ensure_xq(l, 3);
			if (!tagged) tail = plan(false);
		}
		// the level above the first one inside joins the launch in launch order (k_mg_coarse: "fused in launch order"): never the
		// finest level, never a distributed one, only the whole cycle (the bench entry points run parts of it)
		// Up to MG_TOP_MAX_TILES tiles: measured (profiles/r05_top_ab.txt) C3 - level 1, ~600 tiles - 0.1223 -> 0.1150 ms per
		// iteration; C4 - level 1, 2 400 tiles - 0.2610 -> 0.2658 and the overlapped step + 0.34 ms: 7 200 short workgroups have to be
		// dispatched in front of / behind the resident ones, and the waiting ones hold slots the position correction wants.
		// Not when slab ranks share this GPU (device_share > 1): launch order only holds inside ONE launch - another process's
		// waiting phase-3 workgroups could hold the CU slots this launch's resident workgroups need (and the other way round),
		// a cycle that would end at the 50 ms timeout and retire the kernel for good.
		const bool shared_gpu = s->dist && s->dist->device_share > 1;
		if (tagged && !shared_gpu && parts == MG_PART_ALL && tail - 1 >= std::max(D, 1) && M.lv[tail - 1].n_tiles > 0 && !s->knobs.mg_no_top &&
		    M.lv[tail - 1].n_tiles <= MG_TOP_MAX_TILES && ensure_xq(tail - 1, 1))
			top = tail - 1;
	}
	// cell-parallel kernels for levels of up to 1024 tiles (C4: levels 2 and 3; level 1 with 2048 tiles fills the chip with a
	// wave per tile: 8.58 ms per step against 8.75 with cell-parallel kernels on every coarse level, 8.79 with none)
	const int cp_max = 1024;
	auto exchange_level = [&](int l, void *vec) -> int {
		if (l == 0) return lfa_dist_exchange_slices(s, vec, (int)sizeof(real));
		return lfa_dist_exchange_layer_slices(s, vec, (int)sizeof(real), M.lv[l].g.ntx * M.lv[l].g.nty, M.lv[l].lo_layer, M.lv[l].hi_layer);
	};
	int launches = 0;
	for (int l = 0; l < tail; ++l) {
		const MgLv<real> L = lvl(l);
		const int G = mg_grid(L.n_tiles);
		const bool cp = l >= 1 && L.n_tiles <= cp_max;  // a workgroup per tile on the coarser levels (see k_mg_*_cp)
		const int Gcp = std::max(1, std::min(L.n_tiles, 8192));  // (a slab rank may hold no tile of a level)
		if (l == top) continue;  // (inside k_mg_coarse)
		if (!(l == 0 && level0_presmoothed) && (parts & (l == 0 ? MG_PART_PRE0 : MG_PART_COARSE))) {
			if (cp) hipLaunchKernelGGL(k_mg_presmooth_cp<real>, dim3(Gcp), dim3(256), 0, s->stream, L, MG_INNER_SWEEPS, st);
			else hipLaunchKernelGGL(k_mg_presmooth<real>, dim3(std::max(1, std::min((L.n_tiles + PCG_WAVES - 1) / PCG_WAVES, MG_RR_GRID))), dim3(256), 0, s->stream, L, st);  // (like the residual kernel: no partials, no pipeline)
			++launches;
		}
		if (l < D) {
			LFA_TRY(exchange_level(l, L.x));  // the residual needs the pre-smoothed iterate across the slab faces
			// the first replicated level collects the restricted residual of every rank: zero where this rank has no children
			if (l + 1 == D && M.top_packed) {
				if (M.lv[D].n_tiles)
					hipLaunchKernelGGL(k_mg_tiles_zero<real>, dim3(std::min(M.lv[D].n_tiles, 4096)), dim3(256), 0, s->stream, (const int *)M.lv[D].tiles,
					                   M.lv[D].n_tiles, (real *)M.lv[D].b);
			} else if (l + 1 == D)
				LFA_HIP(s, hipMemsetAsync(M.lv[D].b, 0, M.lv[D].ncp * sizeof(real), s->stream));
		}
		if (parts & (l == 0 ? MG_PART_DOWN0 : MG_PART_COARSE)) {
			if (cp) hipLaunchKernelGGL(k_mg_residual_restrict_cp<real>, dim3(Gcp), dim3(256), 0, s->stream, L, M.lv[l + 1].g, (real *)M.lv[l + 1].b, st);
			// (no partial sums, no software pipeline, 63 VGPRs: this kernel hides its loads with waves, not with the tuned 768-workgroup
			// grid of its neighbours - MG_RR_GRID workgroups)
			else {
				// (the finest level's launch also evaluates the stopping rule on the residual the AXPY kernel has just written)
				MgStop stop{nullptr, 0, 0, 0.0, nullptr, nullptr};
				if (l == 0 && level0_presmoothed && D == 0 && s->cur_iter >= 0 && s->cur_rmax_parts)
					stop = MgStop{s->cur_rmax_parts, s->cur_rmax_n, s->cur_iter, s->prm.tolerance, s->pcg_state, s->pcg_hist};
				hipLaunchKernelGGL(k_mg_residual_restrict<real>, dim3(std::max(1, std::min((L.n_tiles + PCG_WAVES - 1) / PCG_WAVES, MG_RR_GRID))), dim3(256), 0, s->stream, L, M.lv[l + 1].g, (real *)M.lv[l + 1].b, st, stop);
			}
			++launches;
		}
		LFA_LAUNCH_CHECK(s);
		if (l < D && l + 1 == D && M.top_packed) {
			const int nt = M.lv[D].n_tiles;
			const size_t need = std::max((size_t)nt * 512 * sizeof(real), (size_t)1);
			if (need > M.top_buf_cap) {
				LFA_HIP(s, hipStreamSynchronize(s->stream));
				if (M.top_buf) LFA_HIP(s, hipFree(M.top_buf));
				M.top_buf = nullptr;
				M.top_buf_cap = need + need / 4 + 4096;
				LFA_HIP(s, hipMalloc(&M.top_buf, M.top_buf_cap));
			}
			if (nt) hipLaunchKernelGGL((k_mg_tiles_copy<real, true>), dim3(std::min(nt, 4096)), dim3(256), 0, s->stream, (const int *)M.lv[D].tiles, nt, (real *)M.lv[D].b, (real *)M.top_buf);
			LFA_TRY(s->dist->allreduce_buf(s, M.top_buf, (size_t)nt * 512, sizeof(real) == 4 ? LFA_RED_F32 : LFA_RED_F64, false));
			if (nt) hipLaunchKernelGGL((k_mg_tiles_copy<real, false>), dim3(std::min(nt, 4096)), dim3(256), 0, s->stream, (const int *)M.lv[D].tiles, nt, (real *)M.lv[D].b, (real *)M.top_buf);
			LFA_LAUNCH_CHECK(s);
		} else if (l < D && l + 1 == D)
			LFA_TRY(s->dist->allreduce_buf(s, M.lv[D].b, M.lv[D].ncp, sizeof(real) == 4 ? LFA_RED_F32 : LFA_RED_F64, false));
	}
	if ((parts & MG_PART_COARSE) && persist) {
		MgCo<real> C;
		for (int l = tail; l <= last; ++l) C.lv[l] = lvl(l);
		C.first = tail;
		C.last = last;
		C.nsw = MG_COARSEST_SWEEPS;
		C.inner = MG_INNER_SWEEPS;
		for (int l = tail; l <= last; ++l) C.ready[l] = M.lv[l].ready;
		if (M.co_tag >= 0x0FFFFFF0u) {  // (a flag holds tag << 4: after 2^28 launches the flags start over)
			for (int l = 1; l <= last; ++l)
				if (M.lv[l].ready) LFA_HIP(s, hipMemsetAsync(M.lv[l].ready, 0, (size_t)3 * M.lv[l].g.nt * sizeof(unsigned), s->stream));
			M.co_tag = 0;
		}
		if (M.co_tag == 0)  // (first launch, or the tags start over: the tagged words too)
			for (int l = 1; l <= last; ++l)
				if (M.lv[l].xq) LFA_HIP(s, hipMemsetAsync(M.lv[l].xq, 0, M.lv[l].ncp * 8 * (size_t)M.lv[l].xq_sections, s->stream));
		C.tag = ++M.co_tag;  // (0 is what the flags are initialised to)
		for (int l = tail; l <= last; ++l) {
			C.xq[l] = tagged ? M.lv[l].xq : nullptr;
			C.ncp[l] = M.lv[l].ncp;
		}
		C.top = top;
		C.top_wgs = top >= 0 ? std::min(M.lv[top].n_tiles, MG_TOP_MAX_WGS) : 0;
		if (top >= 0) {
			C.lv[top] = lvl(top);
			C.xq[top] = M.lv[top].xq;
			C.ncp[top] = M.lv[top].ncp;
		}
		C.abort = s->pcg_state + 2;
		C.fault = s->knobs.mg_co_fault;
		// one workgroup per tile of its first level: every workgroup owns one tile slot on every level it reaches
		const int W = std::max(1, M.lv[tail].n_tiles) + 3 * C.top_wgs;
		const size_t lds = std::max((size_t)(last - tail + 1) * sizeof(CoLevel<real>) + 512 * sizeof(real), sizeof(CpTile<real>) + 512 * sizeof(real));
		CoGateScope gate(s);
		// (dynamic LDS limit: set where `fits` is)
		if (tagged) hipLaunchKernelGGL((k_mg_coarse<real, MemAgent, true>), dim3(W), dim3(256), lds, s->stream, C, st);
		else hipLaunchKernelGGL((k_mg_coarse<real, MemAgent, false>), dim3(W), dim3(256), lds, s->stream, C, st);
		if (gate.done() != LFA_OK) return lfa_fail(s, LFA_E_HIP, "chaining k_mg_coarse behind the device's previous one failed");
		LFA_LAUNCH_CHECK(s);
		++launches;
	} else if (parts & MG_PART_COARSE) {
		MgTail<real> T;
		for (int l = tail; l <= last; ++l) T.lv[l] = lvl(l);
		T.first = tail;
		T.last = last;
		// the trailing run of single-tile levels (at most MG_CHAIN_MAX of them) stays inside one wave
		T.chain = last;
		while (T.chain > tail && last - (T.chain - 1) + 1 <= MG_CHAIN_MAX && M.lv[T.chain - 1].n_tiles == 1) --T.chain;
		T.nsw = MG_COARSEST_SWEEPS;
		T.inner = MG_INNER_SWEEPS;  // measured at C4: 19 iterations; 1 sweep on the tail levels: 22
		hipLaunchKernelGGL(k_mg_tail<real>, dim3(1), dim3(MG_TAIL_WAVES * 64), 0, s->stream, T, st);
		LFA_LAUNCH_CHECK(s);
		++launches;
	}
	for (int l = tail - 1; l >= 0; --l) {
		const MgLv<real> L = lvl(l);
		const int G = mg_grid(L.n_tiles);
		if (l == top) continue;  // (inside k_mg_coarse)
		if (!(parts & (l == 0 ? MG_PART_UP0 : MG_PART_COARSE))) continue;
		if (l == 0)
			launch_up0<real>(G, s->stream, L, M.lv[1].g, (const real *)M.lv[1].y, inv_scale, part_sigma, st);
		else if (L.n_tiles <= cp_max)
			hipLaunchKernelGGL(k_mg_prolong_postsmooth_cp<real>, dim3(std::max(1, std::min(L.n_tiles, 8192))), dim3(256), 0, s->stream, L, M.lv[l + 1].g,
			                   (const real *)M.lv[l + 1].y, MG_INNER_SWEEPS, st);
		else
			hipLaunchKernelGGL((k_mg_prolong_postsmooth<real, false, 1>), dim3(G), dim3(256), 0, s->stream, L, M.lv[l + 1].g,
			                   (const real *)M.lv[l + 1].y, (real)1, (double *)nullptr, st);
		LFA_LAUNCH_CHECK(s);
		++launches;
		// the finer level corrects its ring cells across a slab face with this level's result there
		if (l >= 1 && l < D) LFA_TRY(exchange_level(l, L.y));
	}
	if (parts == MG_PART_ALL) {
		M.launches_per_cycle = launches + (level0_presmoothed ? 1 : 0);
		M.first_co = persist ? (top >= 0 ? top : tail) : 0;
	}
	return LFA_OK;
}

void lfa_mg_stats(const lfa_sim *s, uint64_t *launches_per_cycle, uint64_t *levels, uint64_t *first_co) {
	*launches_per_cycle = s->mg ? (uint64_t)s->mg->launches_per_cycle : 0;
	*levels = s->mg ? (uint64_t)s->mg->n_levels : 0;
	*first_co = s->mg ? (uint64_t)s->mg->first_co : 0;
}

int lfa_mg_apply(lfa_sim *s, double *part_sigma) {
	return s->prm.pcg_dtype == LFA_PCG_F64 ? mg_apply_t<double>(s, part_sigma, false) : mg_apply_t<float>(s, part_sigma, false);
}

/// One PCG iteration's AXPYs + V-cycle: p += alpha s, r -= alpha q (alpha = sigma / (q.s) from the partials), signed max r
/// into part_rmax, z = V(r) / scale, partials of dot(z, r) into part_sigma_new. `sdir` = current search direction.
template <typename real>
static int mg_axpy_apply_t(lfa_sim *s, const void *sdir, const double *part_sigma, int n_sigma, const double *part_qs, int n_qs,
                           double *part_rmax, double *part_sigma_new) {
	const int *tiles0;
	const int n0 = lfa_mg_level0(s, &tiles0, nullptr);
	const int G = mg_grid(n0);
	launch_axpy_presmooth<real>(G, s->stream, tiles0, n0, (const uint8_t *)s->abits, (real *)s->vp,
	                            (const real *)sdir, (real *)s->vr, (real *)s->vq, part_sigma, n_sigma, part_qs, n_qs, part_rmax,
	                            (const int *)s->pcg_state);
	LFA_LAUNCH_CHECK(s);
	return mg_apply_t<real>(s, part_sigma_new, true);
}
/// Slab runs, single-reduction CG (k_mg_axpy_presmooth_cg): direction in vs, its image in vs2, w = A z arrives in vq.
template <typename real>
static int mg_axpy_apply_cg_t(lfa_sim *s, const double *gamma, int n_gamma, const double *gamma_old, int n_gamma_old, const double *delta,
                              int n_delta, const double *rmax_prev, int n_rmax, int iter, double *alpha_io, double *part_rmax,
                              double *part_sigma_new) {
	const int G = mg_grid(s->n_ptiles);
	hipLaunchKernelGGL(k_mg_axpy_presmooth_cg<real>, dim3(G), dim3(256), 0, s->stream, (const int *)s->ptiles, s->n_ptiles,
	                   (const uint8_t *)s->abits, (real *)s->vp, (real *)s->vs, (real *)s->vs2, (const real *)s->vz, (real *)s->vr, (real *)s->vq,
	                   gamma, n_gamma, gamma_old, n_gamma_old, delta, n_delta, rmax_prev, n_rmax, s->prm.tolerance, iter, alpha_io, part_rmax,
	                   s->pcg_state, s->pcg_hist);
	LFA_LAUNCH_CHECK(s);
	return mg_apply_t<real>(s, part_sigma_new, true);
}
int lfa_mg_axpy_apply_cg(lfa_sim *s, const double *gamma, int n_gamma, const double *gamma_old, int n_gamma_old, const double *delta,
                         int n_delta, const double *rmax_prev, int n_rmax, int iter, double *alpha_io, double *part_rmax,
                         double *part_sigma_new) {
	return s->prm.pcg_dtype == LFA_PCG_F64
	           ? mg_axpy_apply_cg_t<double>(s, gamma, n_gamma, gamma_old, n_gamma_old, delta, n_delta, rmax_prev, n_rmax, iter, alpha_io,
	                                        part_rmax, part_sigma_new)
	           : mg_axpy_apply_cg_t<float>(s, gamma, n_gamma, gamma_old, n_gamma_old, delta, n_delta, rmax_prev, n_rmax, iter, alpha_io,
	                                       part_rmax, part_sigma_new);
}
int lfa_mg_axpy_apply(lfa_sim *s, const void *sdir, const double *part_sigma, int n_sigma, const double *part_qs, int n_qs,
                      double *part_rmax, double *part_sigma_new) {
	return s->prm.pcg_dtype == LFA_PCG_F64
	           ? mg_axpy_apply_t<double>(s, sdir, part_sigma, n_sigma, part_qs, n_qs, part_rmax, part_sigma_new)
	           : mg_axpy_apply_t<float>(s, sdir, part_sigma, n_sigma, part_qs, n_qs, part_rmax, part_sigma_new);
}

/// For lfa_bench_kernel: one part of an iteration on the state left by the last solve. 0: AXPYs + level-0 pre-smoothing,
/// 1: level-0 residual + restriction, 2: all coarser levels (down, single-workgroup tail, up), 3: level-0 prolongation +
/// post-smoothing + dot.
/// The tiles the PCG iterates over (the finest level's list), their number, and (may be null: the binning's slot table) the slot of
/// every tile of the grid in that list. After lfa_mg_setup.
int lfa_mg_level0(const lfa_sim *s, const int **tiles, const int **slot) {
	const lfa_mg &M = *s->mg;
	if (tiles) *tiles = M.closed_out ? (const int *)M.lv[0].tiles : (const int *)s->ptiles;
	if (slot) *slot = M.closed_out ? (const int *)M.slot0 : (const int *)s->tile_pslot;
	return M.closed_out ? M.lv[0].n_tiles : s->n_ptiles;
}

namespace {
/// A closed tile's block of A p = b, solved where it stands: red-black SOR sweeps from zero in LDS (a wave per tile, the smoother's
/// own update; the ring stays zero - nothing outside the tile couples to it) until the residual's maximum is below `tol` or the
/// rounding floor of the right-hand side, at most MG_CLOSED_SWEEPS sweeps (<= 64 unknowns with air all around: a few dozen do).
/// p = x / scale (the operator here is unscaled, like the multigrid's). A NaN raises the solve's NaN word.
#define MG_CLOSED_SWEEPS 512
template <typename real>
__global__ void __launch_bounds__(256)
k_mg_solve_closed(const int *ptiles, int n_ptiles, const uint8_t *closed, const uint8_t *abits, const real *b, real *p, real inv_scale,
                  real tol, int *state) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	if (state[0] >= 0) return;  // (the zero right-hand side's early-out: p stays 0 like the reference's, src/pressure_solver.cpp:33-35)
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *h = halo[wid];
	for (int slot = blockIdx.x * PCG_WAVES + wid; slot < n_ptiles; slot += gridDim.x * PCG_WAVES) {
		const int tile = ptiles[slot];
		if (!closed[tile]) continue;  // (uniform per wave)
		const size_t base = (size_t)tile * 512;
		uint32_t ab[8];
		real bb[8];
		real bmax = (real)0;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			ab[zz] = abits[base + zz * 64 + lane];
			bb[zz] = b[base + zz * 64 + lane];
			bmax = fmax(bmax, fabs(bb[zz]));
		}
		MG_FENCE();
		for (int i = lane; i < LFA_HALO_CELLS; i += 64) h[i] = (real)0;
		MG_FENCE();
#pragma unroll
		for (int o = 32; o; o >>= 1) bmax = fmax(bmax, __shfl_xor(bmax, o, 64));
		const real floor_ = bmax * (real)(sizeof(real) == 4 ? 4e-7 : 1e-15);
		const real stop = tol > floor_ ? tol : floor_;
		bool bad = bmax != bmax;
		for (int sweep = 0; sweep < MG_CLOSED_SWEEPS && !bad; sweep += 4) {
			for (int k = 0; k < 4; ++k) {
				gs_colour<real>(h, ab, bb, lx, ly, 0);
				MG_FENCE();
				gs_colour<real>(h, ab, bb, lx, ly, 1);
				MG_FENCE();
			}
			real rmax = (real)0;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const int i = (lx + 1) + 10 * (ly + 1) + 100 * (zz + 1);
				const uint32_t a = ab[zz];
				if (a & AB_UNKNOWN) {
					const real F = (a & AB_FLUID) ? (real)1 : (real)0;
					real val = (real)(a & 7) * h[i];
					val -= F * (h[i - 1] + h[i - 10] + h[i - 100]);
					val -= (real)((a >> 3) & 1) * h[i + 1] + (real)((a >> 4) & 1) * h[i + 10] + (real)((a >> 5) & 1) * h[i + 100];
					const real r = fabs(bb[zz] - val);
					rmax = r > rmax || r != r ? r : rmax;
				}
			}
#pragma unroll
			for (int o = 32; o; o >>= 1) {
				const real other = __shfl_xor(rmax, o, 64);
				rmax = other > rmax || other != other ? other : rmax;
			}
			if (rmax != rmax) bad = true;
			else if (rmax <= stop) break;
		}
		if (bad && lane == 0) state[1] = 1;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) p[base + zz * 64 + lane] = h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] * inv_scale;
		MG_FENCE();
	}
}
}  // namespace

/// Solves the closed tiles (after lfa_mg_setup and the right-hand side, before or beside the PCG: nothing is shared with it).
int lfa_mg_solve_closed(lfa_sim *s, void *out) {
	if (!s->mg || !s->mg->closed_out || !s->n_ptiles) return LFA_OK;
	if (s->mg->lv[0].n_tiles == s->n_ptiles) return LFA_OK;  // (no tile is closed: early in a run, as a rule)
	const int G = std::max(1, std::min((s->n_ptiles + PCG_WAVES - 1) / PCG_WAVES, 4096));
	if (s->prm.pcg_dtype == LFA_PCG_F64)
		hipLaunchKernelGGL(k_mg_solve_closed<double>, dim3(G), dim3(256), 0, s->stream, (const int *)s->ptiles, s->n_ptiles,
		                   (const uint8_t *)s->tile_closed, (const uint8_t *)s->abits, (const double *)s->vr, (double *)(out ? out : s->vp),
		                   1.0 / s->a_scale, 0.05 * s->prm.tolerance, s->pcg_state);
	else
		hipLaunchKernelGGL(k_mg_solve_closed<float>, dim3(G), dim3(256), 0, s->stream, (const int *)s->ptiles, s->n_ptiles,
		                   (const uint8_t *)s->tile_closed, (const uint8_t *)s->abits, (const float *)s->vr, (float *)(out ? out : s->vp),
		                   (float)(1.0 / s->a_scale), (float)(0.05 * s->prm.tolerance), s->pcg_state);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

extern "C" int lfa_get_mg_level_tiles(lfa_sim *s, uint64_t tiles[LFA_MAX_MG_LEVELS]) {
	if (!s || !tiles) return LFA_E_INVALID;
	static_assert(LFA_MAX_MG_LEVELS == MG_MAX_LEVELS, "public and internal level caps differ");
	for (int l = 0; l < MG_MAX_LEVELS; ++l) tiles[l] = (s->mg && l < s->mg->n_levels) ? (uint64_t)s->mg->lv[l].n_tiles : 0;
	return LFA_OK;
}

int lfa_mg_bench_part(lfa_sim *s, int part) {
	if (!s->mg || !s->mg->n_levels || !s->n_ptiles) return lfa_fail(s, LFA_E_INVALID, "multigrid bench: solve with LFA_PRECOND_MULTIGRID first");
	const bool f64 = s->prm.pcg_dtype == LFA_PCG_F64;
	double *P = s->partials;
	const int *tiles0;
	const int n0 = lfa_mg_level0(s, &tiles0, nullptr);
	const int G = mg_grid(n0);
	if (part == 0) {
		if (f64) launch_axpy_presmooth<double>(G, s->stream, tiles0, n0, (const uint8_t *)s->abits, (double *)s->vp, (const double *)s->vs, (double *)s->vr, (double *)s->vq, (const double *)(P + PART_SIG0), G, (const double *)(P + PART_ZS), G, P + PART_RMAX, (const int *)s->pcg_state);
		else launch_axpy_presmooth<float>(G, s->stream, tiles0, n0, (const uint8_t *)s->abits, (float *)s->vp, (const float *)s->vs, (float *)s->vr, (float *)s->vq, (const double *)(P + PART_SIG0), G, (const double *)(P + PART_ZS), G, P + PART_RMAX, (const int *)s->pcg_state);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	const int parts = part == 1 ? MG_PART_DOWN0 : (part == 2 ? MG_PART_COARSE : MG_PART_UP0);
	return f64 ? mg_apply_t<double>(s, P + PART_SIG1, true, parts) : mg_apply_t<float>(s, P + PART_SIG1, true, parts);
}
