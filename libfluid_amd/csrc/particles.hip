// libfluid_amd/csrc/particles.hip -- the per-step particle stages either side of the hot path:
// advection, collision against solid cells / domain walls, position correction, and the device-resident
// simulation::time_step built from them.
//
// Reference: simulation::_advect_particles src/simulation.cpp:226-249; _detect_collisions :612-683 with
// grid::march_cells include/fluid/data_structures/grid.h:140-209; _correct_positions :562-610; time_step :43-125.
// Positions live on the device as (cell, fraction-in-cell); the marches run in fp64 grid units (cell + fraction), which is
// the reference's (position - grid_offset) / cell_size.
#include <math.h>

#include <algorithm>
#include <utility>

#include "common.h"
#include "pcg.h"

namespace {

__device__ inline bool solid_or_outside(const GridDims &g, const uint8_t *solid, int x, int y, int z) {
	if (!in_grid(g, x, y, z)) return true;
	return solid[blocked_index(g, x, y, z)] != 0;
}

/// _detect_collisions for one particle, in grid units (skin = boundary_skin_width / cell_size): up to three bounces of
/// the segment from -> to (DDA over cells, first solid/outside cell stops it `skin` short of the face and removes the
/// normal component), then the push-out from walls / solid neighbours closer than `skin`. Result in `to`.
__device__ inline void collide(const GridDims &g, const uint8_t *solid, double from[3], double to[3], double skin) {
	for (int bounce = 0; bounce < 3; ++bounce) {
		bool hit = false;
		double diff[3], inv[3], t[3];
		int cur[3], last[3], adv[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			cur[d] = (int)floor(from[d]);
			last[d] = (int)floor(to[d]);
			diff[d] = to[d] - from[d];
			adv[d] = diff[d] > 0.0 ? 1 : -1;
			inv[d] = 1.0 / fabs(diff[d]);
			t[d] = fabs((double)(cur[d] + (diff[d] > 0.0 ? 1 : 0)) - from[d]) * inv[d];
		}
		for (int guard = 0; guard < 4096 && (cur[0] != last[0] || cur[1] != last[1] || cur[2] != last[2]); ++guard) {
			int dim = 0;
			double tmin = 2.0;
			if (t[0] < tmin) { tmin = t[0]; dim = 0; }
			if (t[1] < tmin) { tmin = t[1]; dim = 1; }
			if (t[2] < tmin) { tmin = t[2]; dim = 2; }
			if (!(tmin <= 1.0)) break;  // grid.h:196-199
			// static indexing (no scratch): update the chosen axis
			const int a0 = dim == 0, a1 = dim == 1, a2 = dim == 2;
			cur[0] += a0 ? adv[0] : 0; cur[1] += a1 ? adv[1] : 0; cur[2] += a2 ? adv[2] : 0;
			if (solid_or_outside(g, solid, cur[0], cur[1], cur[2])) {
				const double td = a0 ? t[0] : (a1 ? t[1] : t[2]);
				const double od = a0 ? diff[0] : (a1 ? diff[1] : diff[2]);
				const int ad = a0 ? adv[0] : (a1 ? adv[1] : adv[2]);
				double tt = td + skin / (od * (double)(-ad));
				if (tt < 0.0) tt = 0.0;
#pragma unroll
				for (int d = 0; d < 3; ++d) from[d] = tt * to[d] + (1.0 - tt) * from[d];
				if (a0) to[0] = from[0];
				if (a1) to[1] = from[1];
				if (a2) to[2] = from[2];
				hit = true;
				break;
			}
			t[0] += a0 ? inv[0] : 0.0; t[1] += a1 ? inv[1] : 0.0; t[2] += a2 ? inv[2] : 0.0;
		}
		if (!hit) break;
	}
	int ci[3];
	double cp[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		ci[d] = (int)to[d];
		cp[d] = to[d] - (double)ci[d];
	}
	const double skin_max = 1.0 - skin;
	const int n[3] = {g.nx, g.ny, g.nz};
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		if (cp[d] < skin) {
			if (ci[d] == 0 || solid_or_outside(g, solid, ci[0] - (d == 0), ci[1] - (d == 1), ci[2] - (d == 2))) to[d] += skin - cp[d];
		}
		if (cp[d] > skin_max) {
			if (ci[d] + 1 >= n[d] || solid_or_outside(g, solid, ci[0] + (d == 0), ci[1] + (d == 1), ci[2] + (d == 2)))
				to[d] += skin_max - cp[d];
		}
	}
}

/// What is left of `collide` when no solid cell is within reach: nothing stops the segment, and the skin push-out (:654-681)
/// only ever fires at the domain walls.
__device__ inline void collide_walls_only(const GridDims &g, double to[3], double skin) {
	const int n[3] = {g.nx, g.ny, g.nz};
	const double skin_max = 1.0 - skin;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const int ci = (int)to[d];
		const double cp = to[d] - (double)ci;
		double x = to[d];
		if (cp < skin && ci == 0) x += skin - cp;
		if (cp > skin_max && ci + 1 >= n[d]) x += skin_max - cp;
		to[d] = x;
	}
}

__device__ inline void cell_of_key(const GridDims &g, uint32_t key, int c[3]) {
	int tile = (int)(key >> 9), l = (int)(key & 511), tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	c[0] = tx * 8 + (l & 7); c[1] = ty * 8 + ((l >> 3) & 7); c[2] = tz * 8 + (l >> 6);
}
/// grid-unit position -> (clamped cell, fraction), the same rule as the upload path (core.hip: cell_and_fraction).
__device__ inline void split_position(double p, int n, int &cell, float &t) {
	double m = p < 0.0 ? 0.0 : p;
	int c = m >= (double)n ? n - 1 : (int)m;
	double td = p - (double)c;
	float tf = (float)td;
	if (!(tf > 0.0f)) tf = 0.0f;
	if (td < 1.0 && tf >= 1.0f) tf = 0.99999994f;
	if (tf > 1.0f) tf = 1.0f;
	cell = c;
	t = tf;
}

#ifndef CORR_PARTS
#define CORR_PARTS 2                       // workgroups per tile of the LDS-tiled position correction (split in z; 4: 5.6 instead of 5.0 ms alone, and no better beside the pressure solve)
#endif
#define CORR_THREADS 512                   // 8 waves per workgroup, 2 workgroups per CU (LDS)
#define CORR_THREADS_BIG 1024              // the second pass over crowded parts: 16 waves, one workgroup per CU
#ifndef CORR_STAGE_DEPTH
#define CORR_STAGE_DEPTH 4                  // staged records per thread in flight (k_correct_fine)
#endif
#define CORR_BIG_SLICES 4                  // workgroups that share the own particles of one crowded part

struct MoveParams {
	double dt_over_h;   // dt / cell_size
	double skin;        // boundary_skin_width / cell_size
	double corr;        // dt * correction_stiffness * re / cell_size   (re = cell_size / sqrt 2)
	double inv_re2;     // cell_size^2 / re^2 = 2
	float *c_home;      // non-null: C lives in its home array, [9][c_home_stride] indexed by the particle id (lfa_sim::c_home)
	size_t c_home_stride;
	int collide;        // 1: the collision handling that follows the move in the reference runs inside the kernel (the default); 0: the
	                    // caller runs it later (lfa_collide), after a host callback that sits between the two in simulation::time_step
};

/// _advect_particles (x += v dt, clamp to [skin, n - skin]) fused with _detect_collisions (from = the old position).
/// tile_solid[t] = tile t holds a solid cell (or padding of a ragged grid, which counts as one)
__global__ void __launch_bounds__(256) k_tile_has_solid(const uint8_t *solid, int nt, uint8_t *tile_solid) {
	for (int t = blockIdx.x; t < nt; t += gridDim.x) {
		const uint8_t *c = solid + (size_t)t * LFA_TILE_CELLS;
		const int any = __syncthreads_or((c[threadIdx.x] | c[threadIdx.x + 256]) != 0);
		if (threadIdx.x == 0) tile_solid[t] = (uint8_t)(any ? 1 : 0);
	}
}
/// tile_clear[t] = no solid cell within one tile of tile t: a particle that starts in t and moves less than 8 cells per axis
/// cannot meet one, so its collision handling reduces to the domain walls (which the clamp of the advection already enforces).
__global__ void k_tile_clear(GridDims g, const uint8_t *tile_solid, uint8_t *tile_clear) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= g.nt) return;
	int tx, ty, tz;
	tile_coords(g, t, tx, ty, tz);
	int any = 0;
	for (int dz = -1; dz <= 1; ++dz)
		for (int dy = -1; dy <= 1; ++dy)
			for (int dx = -1; dx <= 1; ++dx) {
				const int x = tx + dx, y = ty + dy, z = tz + dz;
				if ((unsigned)x < (unsigned)g.ntx && (unsigned)y < (unsigned)g.nty && (unsigned)z < (unsigned)g.ntz)
					any |= tile_solid[x + g.ntx * (y + g.nty * z)];
			}
	tile_clear[t] = (uint8_t)(any ? 0 : 1);
}

/// COERCE: the velocity coercion of the fluid sources first (src/simulation.cpp:227-238): a particle inside a cell of an active
/// coercing source takes the source's velocity and C = 0 (the cell is the particle's current one: the reference hashes at the
/// start of the step, :49).
template <bool COERCE>
__device__ inline uint32_t advect_one(size_t i, const ParticleSoA &p, const GridDims &g, const uint8_t *solid, const MoveParams &mp,
                                      const uint8_t *coerce_map, const float *src_vel, const uint8_t *tile_clear) {
	const uint32_t key = p.key[i];
	if (key == 0xFFFFFFFFu) return key;  // outside this rank's slab (dropped at the next binning)
	int c[3];
	cell_of_key(g, key, c);
	const int nn[3] = {g.nx, g.ny, g.nz};
	float vel[3] = {p.v[0][i], p.v[1][i], p.v[2][i]};
	if (COERCE) {
		const uint32_t src = coerce_map[key];
		if (src) {
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				vel[d] = src_vel[3 * (src - 1) + d];
				p.v[d][i] = vel[d];
			}
			if (mp.c_home) {
				const size_t j = p.id[i];
#pragma unroll
				for (int k = 0; k < 9; ++k) mp.c_home[k * mp.c_home_stride + j] = 0.0f;
			} else {
#pragma unroll
				for (int k = 0; k < 9; ++k) p.c[k][i] = 0.0f;
			}
		}
	}
	double from[3], to[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		from[d] = (double)c[d] + (double)p.t[d][i];
		double x = from[d] + (double)vel[d] * mp.dt_over_h;
		const double lo = mp.skin, hi = (double)nn[d] - mp.skin;
		to[d] = x < lo ? lo : (hi < x ? hi : x);
	}
	// No solid cell within a tile of the start and a move of less than a tile: the march of _detect_collisions meets nothing,
	// and of its skin push-out only the domain walls remain - where the clamp above has already left the particle at least
	// `skin` inside (to[d] in [skin, n - skin] => cp >= skin in cell 0, cp <= 1 - skin in cell n - 1). The general path costs
	// a dependent byte load per crossed cell and per near face; most tiles of a scene are nowhere near a solid.
	const bool open_water = (tile_clear[key >> 9] & 1) && fabs(to[0] - from[0]) < 8.0 && fabs(to[1] - from[1]) < 8.0 &&
	                        fabs(to[2] - from[2]) < 8.0;
	if (!open_water && mp.collide) collide(g, solid, from, to, mp.skin);
	int nc[3];
	float nt[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) split_position(to[d], nn[d], nc[d], nt[d]);
	const uint32_t new_key = blocked_index(g, nc[0], nc[1], nc[2]);
	p.key[i] = new_key;
#pragma unroll
	for (int d = 0; d < 3; ++d) p.t[d][i] = nt[d];
	return new_key;
}

template <bool COERCE>
__global__ void __launch_bounds__(256)
k_advect_collide(size_t n, ParticleSoA p, GridDims g, const uint8_t *solid, MoveParams mp, const uint8_t *coerce_map,
                 const float *src_vel, const uint8_t *tile_clear) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) advect_one<COERCE>(i, p, g, solid, mp, coerce_map, src_vel, tile_clear);
}

/// The same with pass 1 of the binning that follows it in lfa_time_step (core.hip: k_tile_count) done on the way: the new keys are
/// in registers, so a wave ranks its AC_CHUNKS x 64 particles inside their new tiles right here (one atomic per distinct tile) and
/// the binning neither reads the keys again nor waits for that pass's returning atomics (0.24 ms at C4).
#define AC_CHUNKS 8  // particles per lane (4: 0.80 ms, 8: 0.74, 16: 0.73 at C4; the plain advection 0.58 + k_tile_count 0.24)
template <bool COERCE>
__global__ void __launch_bounds__(256)
k_advect_collide_count(size_t n, ParticleSoA p, GridDims g, const uint8_t *solid, MoveParams mp, const uint8_t *coerce_map,
                       const float *src_vel, const uint8_t *tile_clear, uint32_t *tile_count, uint32_t *rank) {
	const int lane = threadIdx.x & 63;
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t i0 = wave * (64 * AC_CHUNKS) + lane;
	uint32_t tile[AC_CHUNKS], my_rank[AC_CHUNKS];
#pragma unroll
	for (int c = 0; c < AC_CHUNKS; ++c) {
		const size_t i = i0 + 64 * c;
		const uint32_t k = i < n ? advect_one<COERCE>(i, p, g, solid, mp, coerce_map, src_vel, tile_clear) : 0xFFFFFFFFu;
		tile[c] = k != 0xFFFFFFFFu ? k >> 9 : 0xFFFFFFFFu;
	}
	lfa_wave_tile_ranks(tile, tile_count, my_rank);
#pragma unroll
	for (int c = 0; c < AC_CHUNKS; ++c) {
		const size_t i = i0 + 64 * c;
		if (tile[c] != 0xFFFFFFFFu) rank[i] = my_rank[c];
	}
}

__device__ inline float hash_unit(uint32_t a, uint32_t b, uint32_t k) {
	uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ k * 0xC2B2AE3Du;
	x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
	return (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

// ------------------------------------------------------------------------------------------------ position correction
// _correct_positions (src/simulation.cpp:562-610): pairwise springs between particles closer than re = cell / sqrt 2, all from
// the OLD positions, fused with the _detect_collisions that follows it.
// The pair tests are VALU bound and cannot get cheaper per pair, so the lever is the number of candidates. History (C4, a
// moving dam, ms of the tiled kernel): round 1 walked the 27 CELLS around a particle - 216 candidates for 12 partners - 10.5;
// packed fp32 pair math 9.2; per-particle pruning of cells by box distance 62 (divergent loop control costs more than the pairs
// it saves); FINE cells of 0.72 cells, the halo block re-sorted in LDS per workgroup, 81 candidates: 5.8; the force evaluated
// branch-free for every candidate (below): 5.0. This version: fine cells of 8/11 = 0.7273 cells (still >= re = 0.7071),
// ALIGNED with the 8-cell tiles - 11 per axis - so the per-tile index is sorted by fine cell once, by the index kernel (1.02 ms
// instead of 1.17 for the cell index), and a workgroup's block of 13 x 13 x 8 fine cells is put together from contiguous runs
// of its 27 source tiles without atomics or a second pass: per fine row one cell of the x-1 tile, the eleven cells of the own x
// tile (one run), one cell of the x+1 tile, in that order = sorted by fine cell. The kernel itself stayed at 5.0 ms (PMC: 2.3e9
// VALU wave-instructions, 70 % of them the pair walk, 47 of 64 lanes active on average - the lanes of a wave are particles of ~21
// different fine cells whose runs have different lengths; staging 18 %, the per-particle epilogue 11 %).
#define FT 11                                // fine cells per tile axis
#define FT3 (FT * FT * FT)
#define FT_STRIDE (FT3 + 1)                  // per tile: first record of every fine cell + the end
#define FT_INV 1.375f                        // 11 / 8 (exact)
#define FT_PL ((FT + CORR_PARTS - 1) / CORR_PARTS)  // own fine layers (z) of a part (the last part takes what is left)
#define FB 13                                // block: the own 11 x 11 fine cells + one on each side
#define FBZ (FT_PL + 2)
#define FB_N (FB * FB * FBZ)
#define FB_ROWS (FB * FBZ)
#define FINE_CAP 5632                       // staged particles (13 x 13 x 8 fine cells hold 4160 at 8 per cell)
#define FINE_CAP_BIG 12288                   // the second pass over flagged parts: 144 KB of positions, one workgroup per CU
// (own particles the u16 list of k_correct_fine holds: twice its count array, where it lives; more are found through the row offsets)
#define FIDX_THREADS 512   // (256: 0.75 ms at C4, 512: 0.58; FIDX_STAGE and FIDX_CNT are multiples of it)
#define FIDX_CNT 1536                       // index kernel: >= FT3 + 1, a multiple of FIDX_THREADS
#define FIDX_STAGE 4608                      // records staged in LDS per tile by the index kernel (72 KB)

__device__ inline int fine_coord(int l, float t) {
	const int f = (int)(((float)l + t) * FT_INV);
	return f < FT - 1 ? f : FT - 1;  // (t == 1 in the last cell: the max face belongs to the last fine cell)
}
/// A record of the fine index: the in-cell fractions (exact) + the particle index, and the cell inside the tile in the two top
/// bits of the fractions (a fraction is in [0, 1]: sign and top exponent bit are clear): lx whole (3 bits); of ly, lz the lowest
/// bit - a fine row overlaps two cells per axis at most, floor(8 f / 11) and the next one.
__device__ inline float4 fine_record(int lx, int ly, int lz, float t0, float t1, float t2, uint32_t index) {
	const uint32_t b0 = (__float_as_uint(t0) & 0x3FFFFFFFu) | ((uint32_t)(lx & 3) << 30);
	const uint32_t b1 = (__float_as_uint(t1) & 0x3FFFFFFFu) | ((uint32_t)(lx >> 2) << 31) | ((uint32_t)(ly & 1) << 30);
	const uint32_t b2 = (__float_as_uint(t2) & 0x3FFFFFFFu) | ((uint32_t)(lz & 1) << 31);
	return make_float4(__uint_as_float(b0), __uint_as_float(b1), __uint_as_float(b2), __uint_as_float(index));
}
__device__ inline void fine_decode(const float4 &r, int fy, int fz, float t[3], int l[3]) {
	const uint32_t b0 = __float_as_uint(r.x), b1 = __float_as_uint(r.y), b2 = __float_as_uint(r.z);
	t[0] = __uint_as_float(b0 & 0x3FFFFFFFu); t[1] = __uint_as_float(b1 & 0x3FFFFFFFu); t[2] = __uint_as_float(b2 & 0x3FFFFFFFu);
	l[0] = (int)(b0 >> 30) | ((int)(b1 >> 31) << 2);
	const int y0 = (8 * fy) / FT, z0 = (8 * fz) / FT;
	l[1] = y0 + ((y0 ^ (int)(b1 >> 30)) & 1);
	l[2] = z0 + ((z0 ^ (int)(b2 >> 31)) & 1);
}

/// Per particle tile: its particles grouped by fine cell - fine_start[tile][f] = first record of fine cell f (absolute), records
/// (fine_record) in spos. (What the reference's _space_hash is to its 27-cell walk, include/fluid/simulation.h:193-197.)
__global__ void __launch_bounds__(FIDX_THREADS)
k_build_fine_index(const int *ptiles, int n_ptiles, const uint32_t *key, const float *t0, const float *t1, const float *t2,
                   const uint32_t *tile_start, const uint32_t *tile_count, uint32_t *fine_start, float4 *spos, uint32_t *key_copy) {
	__shared__ uint32_t cnt[FIDX_CNT];
	__shared__ uint32_t wsum[FIDX_THREADS / 64];
	__shared__ float4 stage[FIDX_STAGE];
	static_assert(FIDX_CNT % FIDX_THREADS == 0 && FIDX_STAGE % FIDX_THREADS == 0, "index kernel sizing");
	constexpr int PER = FIDX_CNT / FIDX_THREADS;
	auto fine_of = [&](uint32_t i, int l[3], float t[3]) -> int {
		const uint32_t k = key[i];
		l[0] = (int)(k & 7); l[1] = (int)((k >> 3) & 7); l[2] = (int)((k >> 6) & 7);
		t[0] = t0[i]; t[1] = t1[i]; t[2] = t2[i];
		return fine_coord(l[0], t[0]) + FT * (fine_coord(l[1], t[1]) + FT * fine_coord(l[2], t[2]));
	};
	constexpr int RPT = FIDX_STAGE / FIDX_THREADS;  // particles per thread of a tile that fits the staging area: kept in registers
	for (int slot = blockIdx.x; slot < n_ptiles; slot += gridDim.x) {
		const int tile = ptiles[slot];
		// ghost tiles (slab decomposition) keep their particles behind the live ones: the range end comes from the count
		const uint32_t b = tile_start[tile], e = b + tile_count[tile];
		const bool staged = e - b <= FIDX_STAGE;  // (uniform)
#pragma unroll
		for (int k = 0; k < PER; ++k) cnt[threadIdx.x + FIDX_THREADS * k] = 0;
		__syncthreads();
		// pass 1: histogram. A tile that fits keeps what it has read - (fine cell << 9 | cell in the tile) and the fractions -
		// in registers for pass 2 (reading key and t a second time was a third of the kernel's HBM traffic)
		uint32_t rf[RPT];
		float r0[RPT], r1[RPT], r2[RPT];
		if (staged) {
#pragma unroll
			for (int r = 0; r < RPT; ++r) {
				const uint32_t i = b + threadIdx.x + FIDX_THREADS * r;
				rf[r] = 0xFFFFFFFFu;
				if (i < e) {
					int l[3];
					float t[3];
					const int f = fine_of(i, l, t);
					rf[r] = ((uint32_t)f << 9) | (uint32_t)(l[0] | (l[1] << 3) | (l[2] << 6));
					r0[r] = t[0]; r1[r] = t[1]; r2[r] = t[2];
					atomicAdd(&cnt[f], 1u);
				}
			}
		} else {
			for (uint32_t i = b + threadIdx.x; i < e; i += FIDX_THREADS) {
				int l[3];
				float t[3];
				atomicAdd(&cnt[fine_of(i, l, t)], 1u);
			}
		}
		__syncthreads();
		// exclusive scan: thread t owns entries PER t .. PER t + PER - 1
		uint32_t c[PER], incl = 0;
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			c[k] = cnt[PER * threadIdx.x + k];
			incl += c[k];
		}
		const uint32_t sum = incl;
		const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			uint32_t t = __shfl_up(incl, o, 64);
			if (lane >= o) incl += t;
		}
		if (lane == 63) wsum[wid] = incl;
		__syncthreads();
		uint32_t ex = b + incl - sum;
		for (int w = 0; w < wid; ++w) ex += wsum[w];
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int f = PER * threadIdx.x + k;
			cnt[f] = ex;  // cursor
			if (f <= FT3) fine_start[(size_t)tile * FT_STRIDE + f] = ex;
			ex += c[k];
		}
		__syncthreads();
		// pass 2: the records are put in place in LDS and written out as one contiguous run (scattered 16-B stores straight to
		// HBM cost 1.6x); a tile with more particles than the staging area holds reads them again and writes them directly
		if (staged) {
#pragma unroll
			for (int r = 0; r < RPT; ++r)
				if (rf[r] != 0xFFFFFFFFu) {
					const uint32_t at = atomicAdd(&cnt[rf[r] >> 9], 1u);
					stage[at - b] = fine_record((int)(rf[r] & 7), (int)((rf[r] >> 3) & 7), (int)((rf[r] >> 6) & 7), r0[r], r1[r], r2[r],
					                            b + threadIdx.x + FIDX_THREADS * r);
				}
			// `key_copy`: the keys of before the correction, for its fallback pass and lfa_correct_collide_undo - written here, from
			// registers and together with the other stores (a store between the loads of pass 1 held them up: vmcnt is one
			// in-order counter), instead of by a 2 x 4 Np byte copy of their own
			if (key_copy) {
#pragma unroll
				for (int r = 0; r < RPT; ++r)
					if (rf[r] != 0xFFFFFFFFu) key_copy[b + threadIdx.x + FIDX_THREADS * r] = ((uint32_t)tile << 9) | (rf[r] & 511u);
			}
			__syncthreads();
			for (uint32_t k = threadIdx.x; k < e - b; k += FIDX_THREADS) spos[b + k] = stage[k];
		} else {
			for (uint32_t i = b + threadIdx.x; i < e; i += FIDX_THREADS) {
				int l[3];
				float t[3];
				const uint32_t at = atomicAdd(&cnt[fine_of(i, l, t)], 1u);
				spos[at] = fine_record(l[0], l[1], l[2], t[0], t[1], t[2], i);
				if (key_copy) key_copy[i] = key[i];
			}
		}
		__syncthreads();
	}
}

/// Where block fine cell (gx, gy, gz) - own-tile fine coordinates, -1 .. FT - lives: the source tile (-1: none) and its fine
/// coordinates there.
__device__ inline int fine_source(const GridDims &g, const uint32_t *tile_count, int tx, int ty, int tz, int gx, int gy, int gz, int &fx,
                                  int &fy, int &fz) {
	const int dx = gx < 0 ? -1 : (gx >= FT ? 1 : 0), dy = gy < 0 ? -1 : (gy >= FT ? 1 : 0), dz = gz < 0 ? -1 : (gz >= FT ? 1 : 0);
	fx = gx - FT * dx; fy = gy - FT * dy; fz = gz - FT * dz;
	const int x = tx + dx, y = ty + dy, z = tz + dz;
	if ((unsigned)x >= (unsigned)g.ntx || (unsigned)y >= (unsigned)g.nty || (unsigned)z >= (unsigned)g.ntz) return -1;
	const int tile = x + g.ntx * (y + g.nty * z);
	return tile_count[tile] ? tile : -1;  // (tile_flag marks the DILATED set: only tiles with particles have an index)
}

/// One record of the fallback pass below.
__device__ inline void correct_collide_record(size_t rk, const ParticleSoA &p, uint32_t *out_key, float *out_tx, float *out_ty, float *out_tz,
                                              const GridDims &g, const uint8_t *solid, const uint32_t *tile_count, const uint32_t *fine_start,
                                              const float4 *spos, const MoveParams &mp, const uint32_t *only_flagged, const int *tile_pslot,
                                              int p_off, const uint32_t *key_before) {
	const float4 me = spos[rk];
	const size_t i = __float_as_uint(me.w);
	const uint32_t key_i = key_before ? key_before[i] : p.key[i];
	if (key_i == 0xFFFFFFFFu) return;
	const int tile = (int)(key_i >> 9);
	const int l[3] = {(int)(key_i & 7), (int)((key_i >> 3) & 7), (int)((key_i >> 6) & 7)};
	const float t[3] = {__uint_as_float(__float_as_uint(me.x) & 0x3FFFFFFFu), __uint_as_float(__float_as_uint(me.y) & 0x3FFFFFFFu),
	                    __uint_as_float(__float_as_uint(me.z) & 0x3FFFFFFFu)};
	const int f[3] = {fine_coord(l[0], t[0]), fine_coord(l[1], t[1]), fine_coord(l[2], t[2])};
	if (only_flagged) {
		const int work = CORR_PARTS * (tile_pslot[tile] - p_off) + f[2] / FT_PL;
		if (!((only_flagged[1 + (work >> 5)] >> (work & 31)) & 1u)) return;
	}
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	const int nn[3] = {g.nx, g.ny, g.nz};
	const float inv_re2 = (float)mp.inv_re2;
	double spring[3] = {0.0, 0.0, 0.0};
	for (int dz = -1; dz <= 1; ++dz)
		for (int dy = -1; dy <= 1; ++dy)
			for (int dx = -1; dx <= 1; ++dx) {
				int sf[3];
				const int src = fine_source(g, tile_count, tx, ty, tz, f[0] + dx, f[1] + dy, f[2] + dz, sf[0], sf[1], sf[2]);
				if (src < 0) continue;
				const uint32_t *fs = fine_start + (size_t)src * FT_STRIDE + (sf[0] + FT * (sf[1] + FT * sf[2]));
				const uint32_t kb = fs[0], ke = fs[1];
				int stx, sty, stz;
				tile_coords(g, src, stx, sty, stz);
				for (uint32_t k = kb; k < ke; ++k) {
					const float4 rec = spos[k];  // the neighbours' OLD positions (the outputs go back in place)
					if (__float_as_uint(rec.w) == (uint32_t)i) continue;
					float nt[3];
					int nl[3];
					fine_decode(rec, sf[1], sf[2], nt, nl);
					const float ddx = ((float)(8 * (tx - stx) + l[0] - nl[0]) + t[0]) - nt[0];
					const float ddy = ((float)(8 * (ty - sty) + l[1] - nl[1]) + t[1]) - nt[1];
					const float ddz = ((float)(8 * (tz - stz) + l[2] - nl[2]) + t[2]) - nt[2];
					const float d2 = ddx * ddx + ddy * ddy + ddz * ddz;  // grid units^2
					if (d2 < 1e-12f) {
						// coincident pair: the reference adds a random unit-box vector (:584-587, std::random_device)
						const uint32_t j = __float_as_uint(rec.w);
						spring[0] += hash_unit((uint32_t)i, j, 0); spring[1] += hash_unit((uint32_t)i, j, 1);
						spring[2] += hash_unit((uint32_t)i, j, 2);
					} else {
						const float kl = 1.0f - d2 * inv_re2;
						if (kl > 0.0f) {
							const float fr = kl * kl * kl * rsqrtf(d2);
							spring[0] += (double)(fr * ddx); spring[1] += (double)(fr * ddy); spring[2] += (double)(fr * ddz);
						}
					}
				}
			}
	const int c[3] = {8 * tx + l[0], 8 * ty + l[1], 8 * tz + l[2]};
	double from[3], to[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		from[d] = (double)c[d] + (double)t[d];
		double x = from[d] + spring[d] * mp.corr;
		to[d] = x < 0.0 ? 0.0 : ((double)nn[d] < x ? (double)nn[d] : x);  // clamp to [offset, grid max] (:604-609)
	}
	if (mp.collide) collide(g, solid, from, to, mp.skin);
	int nc[3];
	float ntt[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) split_position(to[d], nn[d], nc[d], ntt[d]);
	out_key[i] = blocked_index(g, nc[0], nc[1], nc[2]);
	out_tx[i] = ntt[0]; out_ty[i] = ntt[1]; out_tz[i] = ntt[2];
}

/// _correct_positions + _detect_collisions for the particles of the flagged half tiles (those the LDS-tiled kernel could not
/// hold) - or, without a flag bitmap, of all: a thread per particle gathers its 27 fine cells from the index in global memory.
/// `n` = live particles = records of the owned tiles (ghost tiles' records lie behind them).
__global__ void __launch_bounds__(256)
k_correct_collide(size_t n, ParticleSoA p, uint32_t *out_key, float *out_tx, float *out_ty, float *out_tz, GridDims g,
                   const uint8_t *solid, const uint32_t *tile_count, const uint32_t *fine_start, const float4 *spos, MoveParams mp,
                   const uint32_t *only_flagged, const int *tile_pslot, int p_off, const uint32_t *key_before) {
	// A thread per RECORD of the index: everything about the particle as it was BEFORE the correction - the tiled kernel has
	// already rewritten, in place, key and fractions of the particles it moved, and which part a particle belongs to depends on
	// its old fraction - comes from the record (fractions, index) and the copy of the old keys.
	if (only_flagged && only_flagged[0] == 0) return;  // (word 0 counts the flagged parts: none, as a rule - the grid is bounded, so that
	                                                     // costs a few thousand workgroups, not n / 256)
	for (size_t rk = (size_t)blockIdx.x * blockDim.x + threadIdx.x; rk < n; rk += (size_t)gridDim.x * blockDim.x)
		correct_collide_record(rk, p, out_key, out_tx, out_ty, out_tz, g, solid, tile_count, fine_start, spos, mp, only_flagged, tile_pslot,
		                       p_off, key_before);
}

/// LDS-tiled _correct_positions + _detect_collisions on the fine index. One workgroup per (particle tile, z part): part 0 moves the
/// particles of FT_PL fine layers (the last part of what is left); the block staged in LDS is those layers + one fine cell all around.
/// CAP: staged particles. <FINE_CAP, false> is the pass over every part (two workgroups per CU); <FINE_CAP_BIG, true> takes the parts
/// the first pass has flagged in `only` (word 0: how many; a crowded neighbourhood late in a run) with the whole LDS of a CU to
/// itself, and flags what does not fit even that for the global-gather kernel.
#ifdef CORR_PROFILE
// (variant builds only: where a work item's time goes - wall-clock ticks of thread 0 summed over the work items of every launch)
__device__ unsigned long long g_corr_prof[8];
#define CORR_STAMP(k)                                                                   \
	do {                                                                                \
		if (!ONLY && threadIdx.x == 0) {                                                \
			const unsigned long long now_ = wall_clock64();                             \
			atomicAdd(&g_corr_prof[k], now_ - stamp_);                                  \
			stamp_ = now_;                                                              \
		}                                                                               \
	} while (0)
#else
#define CORR_STAMP(k) do { } while (0)
#endif
template <int CAP, bool ONLY>
__global__ void __launch_bounds__(ONLY ? CORR_THREADS_BIG : CORR_THREADS, 4)
k_correct_fine(const int *ptiles, int n_ptiles, uint32_t *out_key, float *out_tx, float *out_ty, float *out_tz, GridDims g,
                const uint8_t *solid, const uint32_t *tile_count, const uint32_t *fine_start, const float4 *spos, MoveParams mp,
                uint32_t *overflow_tiles, const uint8_t *tile_clear, const uint32_t *only) {
	constexpr int T = ONLY ? CORR_THREADS_BIG : CORR_THREADS;  // threads of the workgroup
	constexpr int CNT = ((FB_N + 1 + T - 1) / T) * T;        // >= FB_N + 1, a multiple of T
	__shared__ float px[CAP], py[CAP], pz[CAP];
	__shared__ uint32_t fcnt[CNT];          // particles per block fine cell; afterwards the own list (u16)
	__shared__ uint16_t foff[FB_N + 1];        // first staged slot of every block fine cell
	__shared__ uint32_t rowsrc[FB_ROWS * 3];   // first source record of the three runs of a fine row (x-1 tile, own x tile, x+1 tile)
	__shared__ uint32_t ownoff[FT * FT_PL + 1];
	__shared__ int srct[27];                   // the source tiles around the own one
	__shared__ uint32_t wsum[T / 64];
	uint16_t *own = (uint16_t *)fcnt;
	static_assert(CNT > FB_N && CNT % T == 0, "fcnt sizing");
	constexpr int PER = CNT / T;
	const int nn[3] = {g.nx, g.ny, g.nz};
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	if (ONLY && only[0] == 0) return;  // (as a rule)
	// (second pass: CORR_BIG_SLICES workgroups per crowded part, each stages the block and moves every CORR_BIG_SLICES-th round of
	// its own particles - the pair work of such a part is several times a normal one's, and there are fewer of them than CUs)
	constexpr int SLICES = ONLY ? CORR_BIG_SLICES : 1;
	for (int item = blockIdx.x; item < SLICES * CORR_PARTS * n_ptiles; item += gridDim.x) {
		const int work = item / SLICES, slice = item % SLICES;
		if (ONLY && !((only[1 + (work >> 5)] >> (work & 31)) & 1u)) continue;  // (uniform)
		const int tile = ptiles[work / CORR_PARTS], part = work % CORR_PARTS;
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		const int own_lo = part * FT_PL, own_n = (own_lo + FT_PL <= FT ? FT_PL : FT - own_lo);
		const int gz0 = own_lo - 1;  // own-tile fine layer of block layer 0
		const int nzb = own_n + 2;   // block layers
		const int nrows = FB * nzb, nown_rows = FT * (nzb - 2);
		const bool open_water = (tile_clear[tile] & 1) != 0;  // no solid cell within a tile of this one
		__syncthreads();
#ifdef CORR_PROFILE
		unsigned long long stamp_ = wall_clock64();
		if (!ONLY && threadIdx.x == 0) atomicAdd(&g_corr_prof[7], 1ull);
#endif
		// ---- the 27 source tiles (-1: outside the grid or without particles), then the block's fine cells: where their records
		// are, how many
		if (threadIdx.x < 27) {
			int fx, fy, fz;
			srct[threadIdx.x] = fine_source(g, tile_count, tx, ty, tz, FT * ((int)threadIdx.x % 3 - 1), FT * (((int)threadIdx.x / 3) % 3 - 1),
			                                FT * ((int)threadIdx.x / 9 - 1), fx, fy, fz);
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int f = threadIdx.x + T * k;
			uint32_t cnt = 0, src0 = 0;
			if (f < FB * nrows) {
				const int bx = f % FB, row = f / FB, by = row % FB, bz = row / FB;
				const int gx = bx - 1, gy = by - 1, gz = gz0 + bz;
				const int dx = gx < 0 ? 0 : (gx >= FT ? 2 : 1), dy = gy < 0 ? 0 : (gy >= FT ? 2 : 1), dz = gz < 0 ? 0 : (gz >= FT ? 2 : 1);
				const int src = srct[dx + 3 * dy + 9 * dz];
				if (src >= 0) {
					const uint32_t fid = (uint32_t)((gx - FT * (dx - 1)) + FT * ((gy - FT * (dy - 1)) + FT * (gz - FT * (dz - 1))));
					const uint32_t *fs = fine_start + ((uint32_t)src * FT_STRIDE + fid);  // (nt * FT_STRIDE < 2^32 up to 3 M tiles)
					src0 = fs[0];
					cnt = fs[1] - src0;
				}
				if (bx <= 1 || bx == FB - 1) rowsrc[row * 3 + (bx <= 1 ? bx : 2)] = src0;
			}
			fcnt[f] = cnt;
		}
		__syncthreads();
		// ---- exclusive scan over the block's fine cells (thread t owns cells PER t .. PER t + PER - 1)
		uint32_t c4[PER], sum = 0;
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			c4[k] = fcnt[PER * threadIdx.x + k];
			sum += c4[k];
		}
		uint32_t incl = sum;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			uint32_t t = __shfl_up(incl, o, 64);
			if (lane >= o) incl += t;
		}
		if (lane == 63) wsum[wid] = incl;
		__syncthreads();
		uint32_t woff = 0, total = 0;
		for (int w = 0; w < T / 64; ++w) {
			if (w < wid) woff += wsum[w];
			total += wsum[w];
		}
		if (total > (uint32_t)CAP) {  // uniform
			if (threadIdx.x == 0 && slice == 0) {
				atomicOr(&overflow_tiles[1 + (work >> 5)], 1u << (work & 31));
				atomicAdd(&overflow_tiles[0], 1u);
			}
			continue;
		}
		{
			uint32_t ex = woff + incl - sum;
#pragma unroll
			for (int k = 0; k < PER; ++k) {
				const int f = PER * threadIdx.x + k;
				if (f <= FB_N) foff[f] = (uint16_t)ex;
				ex += c4[k];
			}
		}
		__syncthreads();
		// ---- own particles: the middle runs of the rows by = 1 .. FT, bz = 1 .. nzb - 2; their lengths, scanned by the first waves
		uint32_t own_total = 0;
		{
			uint32_t len = 0;
			if (threadIdx.x < (unsigned)nown_rows) {
				const int row = (1 + threadIdx.x % FT) + FB * (1 + threadIdx.x / FT);
				len = (uint32_t)foff[row * FB + FB - 1] - (uint32_t)foff[row * FB + 1];
			}
			uint32_t in2 = len;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				uint32_t t = __shfl_up(in2, o, 64);
				if (lane >= o) in2 += t;
			}
			if (lane == 63) wsum[wid] = in2;
			__syncthreads();
			uint32_t wo = 0;
			for (int w = 0; w < T / 64; ++w) {
				if (w < wid) wo += wsum[w];
				own_total += wsum[w];
			}
			if (threadIdx.x < (unsigned)nown_rows) ownoff[threadIdx.x] = wo + in2 - len;
			if (threadIdx.x == 0) ownoff[nown_rows] = own_total;
		}
		// (a crowded part with more own particles than the list holds - it lives in the count array - finds them through the row
		// offsets instead: uniform)
		const bool listed = own_total <= 2u * CNT;
		__syncthreads();  // (every count has been read: `own` may overwrite the array)
		CORR_STAMP(0);  // counts, scans, own offsets
		// ---- stage the rows: a wave per fine row copies its three runs - one cell of the x-1 tile, the own x tile's eleven, one
		// of the x+1 tile - which follow each other in the block (positions relative to the own tile's origin, in cells)
		// Flat: a thread per staged SLOT, its row found by a seven-step search over the rows' first slots - every thread's record
		// load is issued at once, two per thread in flight. (Round 5 and before: a wave per fine row, one row after the other - 13
		// dependent HBM round trips per wave and work item. Measured, round 6: C4 4.91 -> 4.84 ms, the late C3 sheet 2.07 -> 1.89;
		// 2, 4 or 8 records in flight make no difference - the kernel is bound by its instruction count, not by this latency.)
		{
			auto locate = [&](uint32_t slot, uint32_t &src, int &seg, int &dy, int &dz, int &fy, int &fz, int &ownslot) {
				int r = 0;
#pragma unroll
				for (int step = 64; step; step >>= 1) {
					const int c = r + step;
					if (c < nrows && (uint32_t)foff[c * FB] <= slot) r = c;
				}
				const int by = r % FB, bz = r / FB;
				const int gy = by - 1, gz = gz0 + bz;
				dy = gy < 0 ? -1 : (gy >= FT ? 1 : 0); dz = gz < 0 ? -1 : (gz >= FT ? 1 : 0);
				fy = gy - FT * dy; fz = gz - FT * dz;
				const uint32_t d0 = foff[r * FB], d1 = foff[r * FB + 1], d2 = foff[r * FB + FB - 1];
				seg = slot < d1 ? 0 : (slot < d2 ? 1 : 2);
				src = rowsrc[r * 3 + seg] + (slot - (seg == 0 ? d0 : (seg == 1 ? d1 : d2)));
				// the own list: slot order = fine-cell order
				ownslot = (listed && seg == 1 && by >= 1 && by <= FT && bz >= 1 && bz <= nzb - 2)
				              ? (int)(ownoff[(by - 1) + FT * (bz - 1)] + (slot - d1)) : -1;
			};
			auto put = [&](uint32_t slot, const float4 &rec, int seg, int dy, int dz, int fy, int fz, int ownslot) {
				float t[3];
				int l[3];
				fine_decode(rec, fy, fz, t, l);
				px[slot] = (float)(8 * (seg - 1) + l[0]) + t[0];
				py[slot] = (float)(8 * dy + l[1]) + t[1];
				pz[slot] = (float)(8 * dz + l[2]) + t[2];
				if (ownslot >= 0) own[ownslot] = (uint16_t)slot;
			};
			// CORR_STAGE_DEPTH records per thread in flight (a dense item stages ~3 600 = 7 per thread)
			for (uint32_t slot0 = threadIdx.x; slot0 < total; slot0 += CORR_STAGE_DEPTH * T) {
				uint32_t src[CORR_STAGE_DEPTH];
				int meta[CORR_STAGE_DEPTH], ownslot[CORR_STAGE_DEPTH];  // meta: seg | (dy + 1) << 2 | (dz + 1) << 4 | fy << 6 | fz << 10
				float4 rec[CORR_STAGE_DEPTH];
#pragma unroll
				for (int u = 0; u < CORR_STAGE_DEPTH; ++u) {
					const uint32_t slot = slot0 + u * T;
					src[u] = 0; meta[u] = 0; ownslot[u] = -1;
					if (slot < total) {
						int seg, dy, dz, fy, fz;
						locate(slot, src[u], seg, dy, dz, fy, fz, ownslot[u]);
						meta[u] = seg | ((dy + 1) << 2) | ((dz + 1) << 4) | (fy << 6) | (fz << 10);
					}
				}
#pragma unroll
				for (int u = 0; u < CORR_STAGE_DEPTH; ++u) rec[u] = spos[src[u]];  // (src 0 for a slot beyond the end: any valid record)
#pragma unroll
				for (int u = 0; u < CORR_STAGE_DEPTH; ++u) {
					const uint32_t slot = slot0 + u * T;
					if (slot < total)
						put(slot, rec[u], meta[u] & 3, ((meta[u] >> 2) & 3) - 1, ((meta[u] >> 4) & 3) - 1, (meta[u] >> 6) & 15, (meta[u] >> 10) & 15, ownslot[u]);
				}
			}
		}
		__syncthreads();
		CORR_STAMP(1);  // staging
#ifdef CORR_PROFILE
		if (!ONLY && threadIdx.x == 0) { atomicAdd(&g_corr_prof[5], (unsigned long long)total); atomicAdd(&g_corr_prof[6], (unsigned long long)own_total); }
#endif
		for (uint32_t w = threadIdx.x + T * slice; w < own_total; w += T * SLICES) {
			uint32_t me;
			if (listed) {
				me = own[w];
			} else {  // the own row that holds the w-th own particle: last r with ownoff[r] <= w
				int lo = 0, hi = nown_rows - 1;
				while (lo < hi) {
					const int mid = (lo + hi + 1) >> 1;
					if (ownoff[mid] <= w) lo = mid;
					else hi = mid - 1;
				}
				me = (uint32_t)foff[((1 + lo % FT) + FB * (1 + lo / FT)) * FB + 1] + (w - ownoff[lo]);
			}
#ifdef CORR_PROFILE
			const unsigned long long w0_ = wall_clock64();
#endif
			const float mx = px[me], my = py[me], mz = pz[me];
			// block coordinates of its fine cell (the same arithmetic the index was built with: mx = (float)lx + t exactly)
			int fx = (int)(mx * FT_INV), fy = (int)(my * FT_INV), fz = (int)(mz * FT_INV);
			fx = fx < FT - 1 ? fx : FT - 1; fy = fy < FT - 1 ? fy : FT - 1; fz = fz < FT - 1 ? fz : FT - 1;
			const int bx = fx + 1, by = fy + 1, bz = fz - gz0;
			float sx = 0.f, sy = 0.f, sz = 0.f;
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2 s2x = 0.f, s2y = 0.f, s2z = 0.f;
			const float inv_re2 = (float)mp.inv_re2;
			uint32_t j = 0xFFFFFFFFu;  // (the coincidence hash needs the particle's index: loaded by the rare branch only)
			const uint32_t myrow = by + FB * bz;
			const uint32_t grec = rowsrc[myrow * 3 + 1] + (me - (uint32_t)foff[myrow * FB + 1]);
			auto pair2 = [&](uint32_t q, f2 qx, f2 qy, f2 qz) {
				const f2 dx = mx - qx, dy = my - qy, dz = mz - qz;
				const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
				const f2 kl = __builtin_elementwise_fma(-d2, (f2)inv_re2, (f2)1.0f);
				if (kl.x > 0.0f || kl.y > 0.0f) {
#pragma unroll
					for (int k = 0; k < 2; ++k) {
						const float d2k = k ? d2.y : d2.x, klk = k ? kl.y : kl.x;
						if (!(klk > 0.0f)) continue;
						if (d2k < 1e-12f) {
							// coincident (or the particle itself): the reference adds a random unit-box vector (:584-587)
							if (q + k != me) {
								if (j == 0xFFFFFFFFu) j = __float_as_uint(spos[grec].w);
								sx += hash_unit(j, q + k, 0); sy += hash_unit(j, q + k, 1); sz += hash_unit(j, q + k, 2);
							}
						} else {
							const float f = klk * klk * klk * rsqrtf(d2k);
							sx += f * (k ? dx.y : dx.x); sy += f * (k ? dy.y : dy.x); sz += f * (k ? dz.y : dz.x);
						}
					}
				}
			};
			// The nine x-runs of three fine cells around the particle (always inside the block); `p2` takes two candidates at a time
			// (a far-away dummy pads an odd tail: it contributes exactly 0).
			auto walk = [&](auto &&p2, auto &&p2_centre) {
				for (int dz = -1; dz <= 1; ++dz)
					for (int dy = -1; dy <= 1; ++dy) {
						const int rowf = FB * ((by + dy) + FB * (bz + dz)) + bx;
						const uint32_t b = foff[rowf - 1], e = foff[rowf + 2];
						auto run = [&](auto &&pp) {
							uint32_t q = b;
							for (; q + 4 <= e; q += 4) {
								const float x0 = px[q], x1 = px[q + 1], x2 = px[q + 2], x3 = px[q + 3];
								const float y0 = py[q], y1 = py[q + 1], y2 = py[q + 2], y3 = py[q + 3];
								const float z0 = pz[q], z1 = pz[q + 1], z2 = pz[q + 2], z3 = pz[q + 3];
								pp(q, f2{x0, x1}, f2{y0, y1}, f2{z0, z1});
								pp(q + 2, f2{x2, x3}, f2{y2, y3}, f2{z2, z3});
							}
							if (q + 2 <= e) {
								pp(q, f2{px[q], px[q + 1]}, f2{py[q], py[q + 1]}, f2{pz[q], pz[q + 1]});
								q += 2;
							}
							if (q < e) pp(q, f2{px[q], 1e6f}, f2{py[q], 1e6f}, f2{pz[q], 1e6f});
						};
						if (dy == 0 && dz == 0) run(p2_centre);  // (uniform) the run that holds the particle itself
						else run(p2);
					}
			};
			// Fast walk, no branch per candidate: the force is evaluated for every candidate with the kernel clamped at 0 - the same
			// sums, since fma(0, d, s) == s. The self pair (d = 0 exactly) contributes 0 through d^2 + 1e-30. What this cannot do
			// is the reference's random push for COINCIDENT pairs (d^2 < 1e-12, :584-587): a lane that has met one - a d^2 below
			// the threshold in the eight other runs (running minimum), or a second one in its own run (count) - redoes its
			// particle with the branching walk. (With ~0.72-cell fine cells one candidate in seven is a partner and a wave tests
			// 128 per step: a "nobody has a partner" branch around the force is almost never skipped - measured: 5.77 -> 5.00 ms.)
			float d2_min = 1.0f;
			uint32_t n_tiny = 0;
			auto fast = [&](f2 qx, f2 qy, f2 qz) -> f2 {
				const f2 dx = mx - qx, dy = my - qy, dz = mz - qz;
				const f2 d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dx, dx, (f2)1e-30f)));
				f2 kl;
				kl.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(-d2.x, inv_re2, 1.0f), 0.0f, 1.0f);
				kl.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(-d2.y, inv_re2, 1.0f), 0.0f, 1.0f);
				f2 f = kl * kl * kl;
				f.x *= __builtin_amdgcn_rsqf(d2.x);  // d^2 >= 1e-30: a normal number, the bare v_rsq_f32 (what rsqrtf compiles to
				f.y *= __builtin_amdgcn_rsqf(d2.y);  // behind the d^2 >= 1e-12 test of the branching walk)
				s2x = __builtin_elementwise_fma(f, dx, s2x);
				s2y = __builtin_elementwise_fma(f, dy, s2y);
				s2z = __builtin_elementwise_fma(f, dz, s2z);
				return d2;
			};
			walk([&](uint32_t, f2 qx, f2 qy, f2 qz) {
					     const f2 d2 = fast(qx, qy, qz);
					     d2_min = fminf(d2_min, fminf(d2.x, d2.y));
				     },
				     [&](uint32_t, f2 qx, f2 qy, f2 qz) {
					     const f2 d2 = fast(qx, qy, qz);
					     n_tiny += (d2.x < 1e-12f ? 1u : 0u) + (d2.y < 1e-12f ? 1u : 0u);
				     });
			if (d2_min < 1e-12f || n_tiny != 1u) {  // rare
				s2x = 0.f; s2y = 0.f; s2z = 0.f;
				walk(pair2, pair2);
			}
			sx += s2x.x + s2x.y;
			sy += s2y.x + s2y.y;
			sz += s2z.x + s2z.y;
#ifdef CORR_PROFILE
			if (!ONLY && threadIdx.x == 0) atomicAdd(&g_corr_prof[4], wall_clock64() - w0_);
#endif
			// the particle's exact cell and fractions (the staged copy is tile-relative fp32) and its index
			float tme[3];
			int lme[3];
			const float4 rec = spos[grec];
			fine_decode(rec, fy, fz, tme, lme);
			const uint32_t jj = __float_as_uint(rec.w);
			const int c[3] = {8 * tx + lme[0], 8 * ty + lme[1], 8 * tz + lme[2]};
			const double spring[3] = {(double)sx, (double)sy, (double)sz};
			double from[3], to[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				from[d] = (double)c[d] + (double)tme[d];
				double x = from[d] + spring[d] * mp.corr;
				to[d] = x < 0.0 ? 0.0 : ((double)nn[d] < x ? (double)nn[d] : x);
			}
			// (a correction moves a particle by a fraction of a cell: in open water there is nothing to march against, only the
			// domain walls push back)
			if (mp.collide) {
				if (!open_water || fabs(to[0] - from[0]) >= 7.0 || fabs(to[1] - from[1]) >= 7.0 || fabs(to[2] - from[2]) >= 7.0)
					collide(g, solid, from, to, mp.skin);
				else
					collide_walls_only(g, to, mp.skin);
			}
			int nc[3];
			float nt[3];
#pragma unroll
			for (int d = 0; d < 3; ++d) split_position(to[d], nn[d], nc[d], nt[d]);
			out_key[jj] = blocked_index(g, nc[0], nc[1], nc[2]);
			out_tx[jj] = nt[0]; out_ty[jj] = nt[1]; out_tz[jj] = nt[2];
		}
#ifdef CORR_PROFILE
		CORR_STAMP(2);  // thread 0's own particles
		__syncthreads();
		CORR_STAMP(3);  // waiting for the workgroup's slowest thread
#endif
	}
}

}  // namespace
#ifdef CORR_PROFILE
extern "C" int lfa_debug_corr_prof(uint64_t out[8], int reset) {
	unsigned long long h[8] = {0};
	if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_corr_prof), sizeof h) != hipSuccess) return -1;
	for (int i = 0; i < 8; ++i) out[i] = h[i];
	if (reset) {
		unsigned long long z[8] = {0};
		if (hipMemcpyToSymbol(HIP_SYMBOL(g_corr_prof), z, sizeof z) != hipSuccess) return -1;
	}
	return 0;
}
#endif

static MoveParams move_params(const lfa_sim *s, double dt) {
	MoveParams mp;
	const double h = s->prm.cell_size, re = h / sqrt(2.0);
	mp.dt_over_h = dt / h;
	mp.skin = s->prm.boundary_skin_width / h;
	mp.corr = dt * s->prm.correction_stiffness * re / h;
	mp.inv_re2 = h * h / (re * re);
	mp.collide = 1;
	mp.c_home = s->c_home_valid ? s->c_home : nullptr;
	mp.c_home_stride = s->c_home_cap;
	return mp;
}


// ===================================================================================================== fluid sources
namespace {
__global__ void k_mark_coerce(const uint32_t *cell, const uint32_t *src1, size_t n, uint8_t *map) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) map[cell[i]] = (uint8_t)src1[i];
}
/// seed_cell's `for (; num < target; ++num)` (src/simulation.cpp:144): particles an entry has to create, given the count of
/// the last binning and what earlier entries of the same cell have already topped it up to.
__global__ void k_source_need(const uint32_t *cell, const uint32_t *lo, const uint32_t *target, size_t n, const uint32_t *cell_count,
                              const uint32_t *tile_flag, uint32_t *need) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t b = cell[i];
	// first entry of the cell: the binning's count; later entries: the target the previous entry left in the hash (:150)
	const uint32_t have = lo[i] != 0xFFFFFFFFu ? lo[i] : (tile_flag[b >> 9] ? cell_count[b] : 0u);
	need[i] = target[i] > have ? target[i] - have : 0u;
}
__device__ inline uint64_t mix64(uint64_t x) {
	x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
	x ^= x >> 27; x *= 0x94D049BB133111EBull;
	x ^= x >> 31;
	return x;
}
/// One workgroup-strided thread per seeding entry: position = cell + U[0,1)^3 (seed_cell :141-147), velocity = the source's.
__global__ void k_source_seed(const uint32_t *cell, const uint32_t *src_of, const uint32_t *need, const uint32_t *off, size_t n,
                              const float *src_vel, ParticleSoA p, size_t base, uint64_t seed, uint64_t id_base) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t cnt = need[i], b = cell[i];
	const float *vel = src_vel + 3 * src_of[i];
	for (uint32_t j = 0; j < cnt; ++j) {
		const size_t d = base + off[i] + j;
		p.key[d] = b;
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			// the generator is keyed on the particle's own slot d (unique over every entry of a call, so two entries that top up
			// the same cell never create coincident particles, and no (cell, j) window can alias a neighbouring cell's)
			// (a slab rank keys on the particle's id in the whole job, id_base + its index among this call's new particles)
			const uint64_t r = mix64(seed + ((id_base + (uint64_t)(d - base)) * 3ull + (uint64_t)a + 1ull) * 0x9E3779B97F4A7C15ull + ((uint64_t)b << 40));
			float t = (float)(r >> 40) * (1.0f / 16777216.0f);  // 24 random bits: [0, 1)
			p.t[a][d] = t;
			p.v[a][d] = vel[a];
		}
#pragma unroll
		for (int k = 0; k < 9; ++k) p.c[k][d] = 0.0f;
		p.id[d] = (uint32_t)(id_base + (uint64_t)(d - base));
	}
}
}  // namespace

extern "C" int lfa_clear_sources(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	s->sources.clear();
	s->sources_valid = false;
	return LFA_OK;
}

extern "C" int lfa_add_source(lfa_sim *s, const int32_t *xyz, uint64_t k, const double velocity[3], uint64_t root, int active,
                              int coerce_velocity) {
	if (!s || (!xyz && k) || !velocity) return LFA_E_INVALID;
	if (s->sources.size() >= 254) return lfa_fail(s, LFA_E_UNSUPPORTED, "more than 254 fluid sources");
	if (root > 16) return lfa_fail(s, LFA_E_INVALID, "target_density_cubic_root %llu: more than 4096 particles per cell", (unsigned long long)root);
	for (uint64_t i = 0; i < k; ++i)
		if (!in_grid(s->g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]))
			return lfa_fail(s, LFA_E_INVALID, "source cell (%d, %d, %d) outside the grid", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
	lfa_sim::SourceHost h;
	h.xyz.assign(xyz, xyz + 3 * k);
	for (int d = 0; d < 3; ++d) h.vel[d] = velocity[d];
	h.root = root;
	h.active = active != 0;
	h.coerce = coerce_velocity != 0;
	s->sources.push_back(std::move(h));
	s->sources_valid = false;
	return LFA_OK;
}

/// Flattens the source list for the kernels. Seeding entries keep the reference's sequential semantics: seed_cell ends with
/// `_space_hash(cell).count = target` UNCONDITIONALLY (src/simulation.cpp:150), so the first entry of a cell tops it up from the
/// count of the binning, and every later entry of the same cell from the previous entry's target (a static difference; an
/// entry whose target does not exceed it creates nothing and is dropped). Coercion: the last active coercing source of a cell wins.
int lfa_sources_sync(lfa_sim *s) {
	if (s->sources_valid) return LFA_OK;
	if (s->sources == s->sources_built && (s->sources.empty() || s->src_vel)) {  // the host re-sent the same list
		s->sources_valid = true;
		return LFA_OK;
	}
	LFA_HIP(s, hipSetDevice(s->device));
	// slabs: every rank is handed the whole list (like the solid cells) and seeds the cells of its own tile layers; the coercion
	// map covers the whole grid (a particle's cell is always an owned one)
	const int own_zlo = s->dist ? s->slab_lo * 8 : 0, own_zhi = s->dist ? s->slab_hi * 8 : s->g.nz;
	std::vector<uint32_t> cell, lo, target, of, ccell, csrc;
	std::vector<float> vel;
	{
		std::vector<std::pair<uint32_t, uint32_t>> seen;  // (cell, target of its latest entry), kept sorted by cell
		std::vector<std::pair<uint32_t, uint32_t>> coerce;  // (cell, source + 1), last one wins
		for (size_t si = 0; si < s->sources.size(); ++si) {
			const auto &h = s->sources[si];
			for (int d = 0; d < 3; ++d) vel.push_back((float)h.vel[d]);
			if (!h.active) continue;
			const uint32_t tgt = (uint32_t)(h.root * h.root * h.root);
			for (size_t i = 0; i + 2 < h.xyz.size(); i += 3) {
				const uint32_t b = blocked_index(s->g, h.xyz[i], h.xyz[i + 1], h.xyz[i + 2]);
				if (h.coerce) coerce.push_back(std::make_pair(b, (uint32_t)si + 1));
				if (h.xyz[i + 2] < own_zlo || h.xyz[i + 2] >= own_zhi) continue;  // another rank's cell
				auto it = std::lower_bound(seen.begin(), seen.end(), std::make_pair(b, 0u),
				                           [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &c) { return a.first < c.first; });
				if (it != seen.end() && it->first == b) {
					const uint32_t prev = it->second;
					it->second = tgt;
					if (tgt > prev) {
						cell.push_back(b); lo.push_back(prev); target.push_back(tgt); of.push_back((uint32_t)si);
					}
				} else {
					seen.insert(it, std::make_pair(b, tgt));
					cell.push_back(b); lo.push_back(0xFFFFFFFFu); target.push_back(tgt); of.push_back((uint32_t)si);
				}
			}
		}
		std::stable_sort(coerce.begin(), coerce.end(), [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &c) { return a.first < c.first; });
		for (size_t i = 0; i < coerce.size(); ++i)
			if (i + 1 == coerce.size() || coerce[i + 1].first != coerce[i].first) { ccell.push_back(coerce[i].first); csrc.push_back(coerce[i].second); }
	}
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	const size_t n = cell.size(), need = std::max(n, ccell.size());
	if (need > s->src_cap) {
		// the old arrays are dropped first and the handle never keeps a dangling pointer: a failing hipMalloc leaves nulls and a
		// capacity of 0 behind (lfa_destroy frees what is non-null, the next sync allocates again)
		uint32_t **arr[] = {&s->src_cell, &s->src_lo, &s->src_target, &s->src_of, &s->src_need};
		s->src_cap = 0;
		for (uint32_t **q : arr) {
			if (*q) (void)hipFree(*q);
			*q = nullptr;
		}
		const size_t cap = need + need / 2 + 64;
		LFA_HIP(s, hipMalloc(&s->src_cell, cap * 4));
		LFA_HIP(s, hipMalloc(&s->src_lo, cap * 4));
		LFA_HIP(s, hipMalloc(&s->src_target, cap * 4));
		LFA_HIP(s, hipMalloc(&s->src_of, cap * 4));
		LFA_HIP(s, hipMalloc(&s->src_need, 2 * (cap + 1) * 4));  // counts | their exclusive scan
		s->src_cap = cap;
	}
	if (s->src_vel) LFA_HIP(s, hipFree(s->src_vel));
	s->src_vel = nullptr;
	LFA_HIP(s, hipMalloc(&s->src_vel, (vel.size() + 3) * 4));
	if (!vel.empty()) LFA_HIP(s, hipMemcpy(s->src_vel, vel.data(), vel.size() * 4, hipMemcpyHostToDevice));
	// coercion map first (it borrows the entry arrays), then the seeding entries
	const bool had_coerce = s->any_coerce;
	s->any_coerce = !ccell.empty();
	if (s->any_coerce || had_coerce) {
		if (!s->coerce_map) LFA_HIP(s, hipMalloc(&s->coerce_map, s->ncp));
		LFA_HIP(s, hipMemsetAsync(s->coerce_map, 0, s->ncp, s->stream));
		if (s->any_coerce) {
			LFA_HIP(s, hipMemcpyAsync(s->src_cell, ccell.data(), ccell.size() * 4, hipMemcpyHostToDevice, s->stream));
			LFA_HIP(s, hipMemcpyAsync(s->src_of, csrc.data(), csrc.size() * 4, hipMemcpyHostToDevice, s->stream));
			hipLaunchKernelGGL(k_mark_coerce, dim3((unsigned)((ccell.size() + 255) / 256)), dim3(256), 0, s->stream,
			                   (const uint32_t *)s->src_cell, (const uint32_t *)s->src_of, ccell.size(), s->coerce_map);
			LFA_LAUNCH_CHECK(s);
		}
		LFA_HIP(s, hipStreamSynchronize(s->stream));
	}
	if (n) {
		LFA_HIP(s, hipMemcpy(s->src_cell, cell.data(), n * 4, hipMemcpyHostToDevice));
		LFA_HIP(s, hipMemcpy(s->src_lo, lo.data(), n * 4, hipMemcpyHostToDevice));
		LFA_HIP(s, hipMemcpy(s->src_target, target.data(), n * 4, hipMemcpyHostToDevice));
		LFA_HIP(s, hipMemcpy(s->src_of, of.data(), n * 4, hipMemcpyHostToDevice));
	}
	s->n_src_entries = n;
	s->sources_built = s->sources;
	s->sources_valid = true;
	return LFA_OK;
}

extern "C" int lfa_update_sources(lfa_sim *s, uint64_t *n_seeded) {
	if (!s) return LFA_E_INVALID;
	if (n_seeded) *n_seeded = 0;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_update_sources: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_sources_sync(s));
	const size_t n = s->n_src_entries;
	if (!n && !s->dist) return LFA_OK;
	size_t total = 0;
	uint32_t *off = s->src_need ? s->src_need + s->src_cap + 1 : nullptr;
	if (n) {
		hipLaunchKernelGGL(k_source_need, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (const uint32_t *)s->src_cell,
		                   (const uint32_t *)s->src_lo, (const uint32_t *)s->src_target, n, (const uint32_t *)s->cell_count,
		                   (const uint32_t *)s->tile_flag, s->src_need);
		LFA_LAUNCH_CHECK(s);
		// exclusive scan into the second half of the scratch; the total is read back (the host has to size the particle arrays)
		LFA_TRY(lfa_exclusive_scan_u32(s, s->src_need, off, n, off + n));
		LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 96, off + n, 4, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		total = s->h_pinned[96];
	}
	// slabs: what every rank creates, gathered (one sum all-reduce of a vector with one slot per rank) - the ids of the new
	// particles continue the job-wide numbering in rank order, and either every rank re-bins or none does
	uint64_t id_base = s->np_live, total_all = total;
	if (s->dist) {
		const int nr = s->dist->nranks, me = s->dist->rank;
		double hv[32] = {0};
		hv[me] = (double)total;
		double *dv = s->dist_red + 16;  // (slots 0-5 belong to the solve)
		LFA_HIP(s, hipMemcpyAsync(dv, hv, (size_t)nr * 8, hipMemcpyHostToDevice, s->stream));
		LFA_TRY(s->dist->allreduce_buf(s, dv, (size_t)nr, LFA_RED_F64, false));
		LFA_HIP(s, hipMemcpyAsync(hv, dv, (size_t)nr * 8, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		id_base = s->next_global_id;
		total_all = 0;
		for (int r = 0; r < nr; ++r) {
			if (r < me) id_base += (uint64_t)hv[r];
			total_all += (uint64_t)hv[r];
		}
		if (s->next_global_id + total_all >= ((uint64_t)1 << 32)) return lfa_fail(s, LFA_E_INVALID, "more than 2^32 particles");
		s->next_global_id += total_all;
	}
	if (!total_all) return LFA_OK;  // every source cell is full: the binning stands
	if (s->np_live + total >= ((size_t)1 << 32)) return lfa_fail(s, LFA_E_INVALID, "more than 2^32 particles");
	const size_t base = s->np_live;
	LFA_TRY(lfa_particles_materialize(s));  // the new particles bring their own v / C: a deferred binning is completed first
	LFA_TRY(lfa_particles_reserve(s, base, base + total));  // keeps the live records
	if (s->c_home_valid) {  // the new particles get the ids id_base .. id_base + total - 1 (single domain: = base) and C = 0
		LFA_TRY(lfa_c_home_ensure(s, s->dist ? (size_t)s->next_global_id : base + total));
		for (int k = 0; k < 9 && total; ++k)
			LFA_HIP(s, hipMemsetAsync(s->c_home + (size_t)k * s->c_home_cap + id_base, 0, total * 4, s->stream));
	}
	++s->source_epoch;
	if (total) {
		hipLaunchKernelGGL(k_source_seed, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (const uint32_t *)s->src_cell,
		                   (const uint32_t *)s->src_of, (const uint32_t *)s->src_need, (const uint32_t *)off, n, (const float *)s->src_vel,
		                   s->pb[s->cur], base, 0x5EED50ull + s->source_epoch * 0x632BE59BD9B4E019ull, id_base);
		LFA_LAUNCH_CHECK(s);
	}
	s->np_live = base + total;
	s->np = s->np_live;
	s->vmax2_valid = false;  // the new particles carry their source's velocity
	if (n_seeded) *n_seeded = total;
	return lfa_hash_particles(s);  // the reference re-hashes after seeding (src/simulation.cpp:64)
}

/// Which tiles are nowhere near a solid cell / a domain wall (k_tile_clear): recomputed when the solid mask has changed.
static int refresh_tile_clear(lfa_sim *s) {
	if (s->clear_epoch == s->solid_epoch && s->tile_clear) return LFA_OK;
	if (!s->tile_clear) LFA_HIP(s, hipMalloc(&s->tile_clear, (size_t)s->g.nt * 2));
	uint8_t *tile_solid = s->tile_clear + s->g.nt;
	hipLaunchKernelGGL(k_tile_has_solid, dim3(s->g.nt < 16384 ? s->g.nt : 16384), dim3(256), 0, s->stream, (const uint8_t *)s->solid,
	                   s->g.nt, tile_solid);
	hipLaunchKernelGGL(k_tile_clear, dim3((s->g.nt + 255) / 256), dim3(256), 0, s->stream, s->g, (const uint8_t *)tile_solid,
	                   s->tile_clear);
	LFA_LAUNCH_CHECK(s);
	s->clear_epoch = s->solid_epoch;
	return LFA_OK;
}

/// `with_count`: lfa_time_step's variant - the binning that follows finds tile_count / rank done (lfa_sim::counts_fresh).
/// `split`: _advect_particles alone - the positions of before are kept (save_old_positions) for the lfa_collide that follows.
static int save_old_positions(lfa_sim *s, size_t n);
static int advect_collide(lfa_sim *s, double dt, bool with_count, bool split = false) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_particles_materialize(s));  // reads v
	const size_t n = s->binned ? s->np_live : s->np;
	MoveParams mp = move_params(s, dt);
	s->move_pending = false;
	if (split) {  // (slabs: the particles migrate once they have their final positions, in lfa_collide)
		LFA_TRY(save_old_positions(s, n));
		mp.collide = 0;
	}
	LFA_TRY(lfa_sources_sync(s));
	LFA_TRY(refresh_tile_clear(s));
	s->counts_fresh = false;
	if (n) {
		if (s->any_coerce) s->vmax2_valid = false;  // velocities are overwritten inside the coercing sources' cells
		const uint8_t *cm = s->any_coerce ? (const uint8_t *)s->coerce_map : (const uint8_t *)nullptr;
		const float *sv = s->any_coerce ? (const float *)s->src_vel : (const float *)nullptr;
		if (with_count && !s->dist) {
			LFA_HIP(s, hipMemsetAsync(s->tile_count, 0, (size_t)(s->g.nt + 1) * 4, s->stream));
			const dim3 grid((unsigned)((n + 256 * AC_CHUNKS - 1) / (256 * AC_CHUNKS)));
			if (s->any_coerce)
				hipLaunchKernelGGL(k_advect_collide_count<true>, grid, dim3(256), 0, s->stream, n, s->pb[s->cur], s->g, s->solid,
				                   mp, cm, sv, (const uint8_t *)s->tile_clear, s->tile_count, s->rank);
			else
				hipLaunchKernelGGL(k_advect_collide_count<false>, grid, dim3(256), 0, s->stream, n, s->pb[s->cur], s->g, s->solid,
				                   mp, cm, sv, (const uint8_t *)s->tile_clear, s->tile_count, s->rank);
			s->counts_fresh = true;
		} else {
			const dim3 grid((unsigned)((n + 255) / 256));
			if (s->any_coerce)
				hipLaunchKernelGGL(k_advect_collide<true>, grid, dim3(256), 0, s->stream, n, s->pb[s->cur], s->g, s->solid, mp,
				                   cm, sv, (const uint8_t *)s->tile_clear);
			else
				hipLaunchKernelGGL(k_advect_collide<false>, grid, dim3(256), 0, s->stream, n, s->pb[s->cur], s->g, s->solid, mp,
				                   cm, sv, (const uint8_t *)s->tile_clear);
		}
		LFA_LAUNCH_CHECK(s);
	}
	if (!split) LFA_TRY(lfa_dist_migrate(s));
	s->unknown_count_valid = false;
	s->move_pending = split;
	return LFA_OK;
}
extern "C" int lfa_advect_collide(lfa_sim *s, double dt) { return advect_collide(s, dt, false); }
extern "C" int lfa_advect(lfa_sim *s, double dt) { return advect_collide(s, dt, false, true); }

namespace {
/// _detect_collisions as its own pass (lfa_collide): from = the position saved before the move, to = the current one.
__global__ void __launch_bounds__(256)
k_collide_only(size_t n, ParticleSoA p, const uint32_t *old_key, const float *ot0, const float *ot1, const float *ot2, GridDims g,
               const uint8_t *solid, double skin) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t key = p.key[i];
	if (key == 0xFFFFFFFFu) return;
	const int nn[3] = {g.nx, g.ny, g.nz};
	int c[3], oc[3];
	cell_of_key(g, key, c);
	double from[3], to[3];
	const float ot[3] = {ot0 ? ot0[i] : p.t[0][i], ot1 ? ot1[i] : p.t[1][i], ot2 ? ot2[i] : p.t[2][i]};
	cell_of_key(g, old_key ? old_key[i] : key, oc);
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		to[d] = (double)c[d] + (double)p.t[d][i];
		from[d] = (double)oc[d] + (double)ot[d];
	}
	collide(g, solid, from, to, skin);
	int nc[3];
	float nt[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) split_position(to[d], nn[d], nc[d], nt[d]);
	p.key[i] = blocked_index(g, nc[0], nc[1], nc[2]);
#pragma unroll
	for (int d = 0; d < 3; ++d) p.t[d][i] = nt[d];
}
}  // namespace

/// The positions (key, t) of before a split move go to the other buffer's key / t arrays (free between a binning and the G2P, and
/// after the materialisation at the start of a step), where lfa_collide and a download in between (old_position) find them.
static int save_old_positions(lfa_sim *s, size_t n) {
	if (!n) return LFA_OK;
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	LFA_HIP(s, hipMemcpyAsync(oth.key, cur.key, n * 4, hipMemcpyDeviceToDevice, s->stream));
	for (int d = 0; d < 3; ++d) LFA_HIP(s, hipMemcpyAsync(oth.t[d], cur.t[d], n * 4, hipMemcpyDeviceToDevice, s->stream));
	return LFA_OK;
}

extern "C" int lfa_collide(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	const size_t n = s->binned ? s->np_live : s->np;
	const bool pending = s->move_pending;
	s->move_pending = false;
	if (!n) return lfa_dist_migrate(s);  // (slabs: every rank takes part in the hand-over, also one without particles)
	LFA_TRY(refresh_tile_clear(s));
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	// nothing pending (an upload in between has replaced the particles): from = to, i.e. the skin push-out alone
	hipLaunchKernelGGL(k_collide_only, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, cur,
	                   pending ? (const uint32_t *)oth.key : (const uint32_t *)nullptr, pending ? (const float *)oth.t[0] : (const float *)nullptr,
	                   pending ? (const float *)oth.t[1] : (const float *)nullptr, pending ? (const float *)oth.t[2] : (const float *)nullptr, s->g,
	                   (const uint8_t *)s->solid, move_params(s, 0.0).skin);
	LFA_LAUNCH_CHECK(s);
	s->unknown_count_valid = false;
	// slabs: the move that lfa_advect / lfa_correct left pending is complete now - particles that crossed a slab face change rank
	return lfa_dist_migrate(s);
}

/// The correction's scratch for the cell-ordered positions: four consecutive v / c arrays that are free between the binning and
/// the G2P - the other buffer's v.. ; with a deferred binning, where the other buffer still holds the v (and for APIC the C)
/// that the P2G / G2P read: this buffer's own v.. (APIC: the G2P is yet to fill them) or the other buffer's c[0..3] (PIC /
/// FLIP: its C has moved here already)
static float4 *correction_scratch(lfa_sim *s) {
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	return (float4 *)(!s->vc_pending ? oth.v[0] : (s->vc_with_c ? cur.v[0] : oth.c[0]));
}

/// First half of lfa_correct_collide: per-cell particle lists + cell-ordered positions (reads the (key, t) of the current binning).
static int correct_build_index(lfa_sim *s, bool exchange = true) {
	const size_t n = s->np_live;
	if (!n && !s->dist) return LFA_OK;
	if (!s->fine_start) {
		const size_t bytes = (size_t)s->g.nt * FT_STRIDE * 4;
		hipError_t e = hipMalloc(&s->fine_start, bytes);
		if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of the fine-cell index (%zu bytes) failed", bytes);
	}
	// neighbours within one cell across the slab faces: ghost copies of the adjacent tile layers' particles
	if (s->dist && exchange) LFA_TRY(lfa_dist_exchange_ghost_particles(s));
	LFA_TRY(refresh_tile_clear(s));
	const int n_index = s->dist ? s->n_ptiles_all : s->n_ptiles;
	const int grid = n_index < 16384 ? (n_index > 0 ? n_index : 1) : 16384;
	ParticleSoA &cur = s->pb[s->cur];
	if (s->timing && n) LFA_HIP(s, hipEventRecord(s->ev[40], s->stream));
	hipLaunchKernelGGL(k_build_fine_index, dim3(grid), dim3(FIDX_THREADS), 0, s->stream, s->dist ? s->ptiles_all : s->ptiles, n_index, cur.key,
	                   cur.t[0], cur.t[1], cur.t[2], s->tile_start, s->tile_count, s->fine_start, correction_scratch(s), s->pb[s->cur ^ 1].key);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

/// Second half: the pairwise correction + collision; writes (key, t) in place.
static int correct_apply(lfa_sim *s, double dt, bool migrate = true, bool with_collide = true) {
	const size_t n = s->np_live;
	MoveParams mpc = move_params(s, dt);
	mpc.collide = with_collide ? 1 : 0;
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	float4 *spos = correction_scratch(s);
	if (n) {
		// LDS-tiled pass; half tiles whose neighbourhood exceeds the LDS capacity are flagged and redone, first by the same kernel
		// with the whole LDS of a CU (round 4: late in a run a few hundred crowded half tiles cost the global-gather kernel more
		// than the other 31 000 cost the tiled one), then - what exceeds even that - by the global-gather kernel
		const size_t ovf_words = ((size_t)CORR_PARTS * s->n_ptiles + 31) / 32 + 2;  // word 0: how many are flagged
		const size_t ovf_stride = ((size_t)CORR_PARTS * s->g.nt + 31) / 32 + 2;     // second bitmap: what the big pass has flagged
		if (!s->corr_ovf) {  // 2 x 2 bits per tile of the grid; its own array: the pressure solve may be running beside this
			hipError_t e = hipMalloc(&s->corr_ovf, 2 * ovf_stride * 4);
			if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of the correction's overflow bitmaps failed");
		}
		uint32_t *ovf = s->corr_ovf, *ovf2 = s->corr_ovf + ovf_stride;
		s->corr_parts_tiles = s->n_ptiles;
		LFA_HIP(s, hipMemsetAsync(ovf, 0, ovf_words * 4, s->stream));
		LFA_HIP(s, hipMemsetAsync(ovf2, 0, ovf_words * 4, s->stream));
		{
			const int work = CORR_PARTS * s->n_ptiles, g2 = work < 65536 ? (work > 0 ? work : 1) : 65536;
			// every neighbour position comes from the records of the index built above (the OLD positions), so the new ones are
			// written in place; the fallback pass below takes its particles' old state from the records and the copy of the keys
			// the index kernel has left in the other buffer
			if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[41], s->stream));
			hipLaunchKernelGGL((k_correct_fine<FINE_CAP, false>), dim3(g2), dim3(CORR_THREADS), 0, s->stream, s->ptiles, s->n_ptiles, cur.key,
			                   cur.t[0], cur.t[1], cur.t[2], s->g, s->solid, s->tile_count, s->fine_start, (const float4 *)spos, mpc, ovf,
			                   (const uint8_t *)s->tile_clear, (const uint32_t *)nullptr);
			LFA_LAUNCH_CHECK(s);
			if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[42], s->stream));
			// (a workgroup per CU and a few more: they return at once unless something is flagged)
			hipLaunchKernelGGL((k_correct_fine<FINE_CAP_BIG, true>), dim3(std::min(CORR_BIG_SLICES * g2, 2048)), dim3(CORR_THREADS_BIG), 0, s->stream, s->ptiles,
			                   s->n_ptiles, cur.key, cur.t[0], cur.t[1], cur.t[2], s->g, s->solid, s->tile_count, s->fine_start,
			                   (const float4 *)spos, mpc, ovf2, (const uint8_t *)s->tile_clear, (const uint32_t *)ovf);
			LFA_LAUNCH_CHECK(s);
		}
		const uint32_t *fallback = ovf2;
		hipLaunchKernelGGL(k_correct_collide, dim3((unsigned)std::min<size_t>((n + 255) / 256, 16384)), dim3(256), 0, s->stream, n, cur, cur.key, cur.t[0],
		                   cur.t[1], cur.t[2], s->g, s->solid, s->tile_count, s->fine_start, (const float4 *)spos, mpc,
		                   fallback, (const int *)s->tile_pslot, s->p_off, (const uint32_t *)oth.key);
		LFA_LAUNCH_CHECK(s);
	}
	if (migrate) LFA_TRY(lfa_dist_migrate(s));
	s->unknown_count_valid = false;
	return LFA_OK;
}

extern "C" int lfa_correct_collide(lfa_sim *s, double dt) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_correct_collide: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	if (!s->np_live && !s->dist) return LFA_OK;
	LFA_TRY(correct_build_index(s));
	return correct_apply(s, dt);
}

/// _correct_positions alone (src/simulation.cpp:562-610): the collision handling that follows it in time_step (:115) is left to
/// lfa_collide, so that a host's post_correction_callback sits exactly where the reference has it (:111-117).
extern "C" int lfa_correct(lfa_sim *s, double dt) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_correct: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	s->move_pending = false;
	if (!s->np_live && !s->dist) return LFA_OK;
	LFA_TRY(correct_build_index(s));  // (slabs: with the ghost exchange - every rank takes part)
	// the positions of before: the keys are copied by correct_apply itself (other buffer), the fractions here
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	for (int d = 0; d < 3 && s->np_live; ++d) LFA_HIP(s, hipMemcpyAsync(oth.t[d], cur.t[d], s->np_live * 4, hipMemcpyDeviceToDevice, s->stream));
	LFA_TRY(correct_apply(s, dt, false, false));  // (no migration yet: lfa_collide hands the particles over)
	s->move_pending = true;
	return LFA_OK;
}

int lfa_corr_join(lfa_sim *s) {
	if (!s->corr_in_flight) return LFA_OK;
	s->corr_in_flight = false;
	LFA_HIP(s, hipStreamWaitEvent(s->stream, s->ev_cjoin, 0));
	return LFA_OK;
}

/// `slab_exchanged`: lfa_time_step on a slab decomposition has already exchanged the ghost particles on the main stream and
/// migrates after the join - no communication happens on the correction's stream.
static int correct_begin(lfa_sim *s, double dt, bool slab_exchanged) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_correct_collide_begin: call lfa_hash_particles first");
	if (s->dist && !slab_exchanged)
		return lfa_fail(s, LFA_E_UNSUPPORTED, "lfa_correct_collide_begin: slab decompositions exchange particles inside the correction (use lfa_correct_collide or lfa_time_step)");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	s->corr_begun = false;
	if (!s->np_live && !s->dist) return LFA_OK;  // nothing to correct: _end and _undo are no-ops
	LFA_HIP(s, hipEventRecord(s->ev_cfork, s->stream));
	LFA_HIP(s, hipStreamWaitEvent(s->stream3, s->ev_cfork, 0));
	hipStream_t main_stream = s->stream;
	s->stream = s->stream3;
	int rc = correct_build_index(s, false);
	if (rc == LFA_OK) rc = correct_apply(s, dt, false);
	s->stream = main_stream;
	const hipError_t e1 = s->timing ? hipEventRecord(s->ev[LFA_EV_CORRECT_END], s->stream3) : hipSuccess;
	// the join event is recorded whatever happened: the main stream waits for it before it touches particles again
	const hipError_t e2 = hipEventRecord(s->ev_cjoin, s->stream3);
	s->corr_in_flight = true;
	s->corr_begun = true;
	s->corr_undo_valid = rc == LFA_OK && !s->dist;
	if (rc < 0) return rc;
	LFA_HIP(s, e1);
	LFA_HIP(s, e2);
	return LFA_OK;
}
extern "C" int lfa_correct_collide_begin(lfa_sim *s, double dt) { return correct_begin(s, dt, false); }

extern "C" int lfa_correct_collide_end(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	return lfa_corr_join(s);
}

namespace {
/// (key, t) of before the correction: the keys from their copy, the fractions from the cell-ordered records (record k
/// belongs to particle spos[k].w) - both are inputs the correction does not write.
__global__ void __launch_bounds__(256)
k_correct_undo(size_t n, const float4 *spos, const uint32_t *old_key, uint32_t *key, float *t0, float *t1, float *t2) {
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const float4 rec = spos[k];
	const uint32_t i = __float_as_uint(rec.w);
	// (the records of the fine index keep cell bits in the two top bits of the fractions: fine_record)
	t0[i] = __uint_as_float(__float_as_uint(rec.x) & 0x3FFFFFFFu);
	t1[i] = __uint_as_float(__float_as_uint(rec.y) & 0x3FFFFFFFu);
	t2[i] = __uint_as_float(__float_as_uint(rec.z) & 0x3FFFFFFFu);
	key[k] = old_key[k];
}
}  // namespace

extern "C" int lfa_correct_collide_undo(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	// What decides is whether the correction's inputs are still there (corr_undo_valid), not whether it is still running: entry
	// points that only read (lfa_cfl, lfa_upload_cells of a grid-editing callback, lfa_get_correction_stats, the mesher) join it
	// without invalidating anything, and a begin that found no particles started nothing.
	if (!s->corr_begun) return LFA_OK;
	if (!s->corr_undo_valid)
		return lfa_fail(s, LFA_E_INVALID, "lfa_correct_collide_undo: positions, binning or solid cells have changed since lfa_correct_collide_begin");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));  // joins if still in flight; the restore below is the change that invalidates a second undo
	s->corr_begun = false;
	const size_t n = s->np_live;
	if (!n) return LFA_OK;
	ParticleSoA &cur = s->pb[s->cur], &oth = s->pb[s->cur ^ 1];
	hipLaunchKernelGGL(k_correct_undo, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, (const float4 *)correction_scratch(s),
	                   (const uint32_t *)oth.key, cur.key, cur.t[0], cur.t[1], cur.t[2]);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

/// Device-resident simulation::time_step(dt) (src/simulation.cpp:43-125) without fluid sources and host callbacks:
/// advect+collide, hash, P2G, gravity, pressure solve, pressure gradient, correct+collide, extrapolate, hash, G2P.
int lfa_g2p_stale(lfa_sim *s);  // grid_ops.hip
extern "C" int lfa_time_step(lfa_sim *s, double dt, double *residual, uint64_t *iterations) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	const bool tm = s->timing;
	// stage boundaries: ev[24 + k]
	enum { B_START = 24, B_ADVECT, B_BIN, B_P2G, B_SOLVE, B_APPLY, B_CORRECT, B_EXTRAP, B_G2P, B_JOIN };
	static_assert(B_CORRECT == LFA_EV_CORRECT_END, "event slot of the end of the correction");
	auto rec = [&](int id) -> int {
		if (tm) LFA_HIP(s, hipEventRecord(s->ev[id], s->stream));
		return LFA_OK;
	};
	LFA_TRY(rec(B_START));
	LFA_TRY(advect_collide(s, dt, true));
	LFA_TRY(rec(B_ADVECT));
	{
		const bool counted = s->counts_fresh;
		s->counts_fresh = false;
		LFA_TRY(lfa_hash_particles_impl(s, counted));
	}
	if (!s->sources.empty()) LFA_TRY(lfa_update_sources(s, nullptr));  // _update_sources + hash_particles (:63-64)
	LFA_TRY(rec(B_BIN));
	// The correction reads and writes particle positions only, the pressure solve / gradient / extrapolation grid arrays only:
	// the two run side by side (the solve is a chain of short launch- and HBM-bound kernels, the correction one long VALU-bound
	// one); the correction's stream has the lower priority. (Starting its cell index already beside the P2G - it only reads the
	// binned (key, t) - changed nothing at C4.) Slabs exchange ghost particles and migrate around the correction, on the main
	// stream: they do the exchange before the fork and the migration after the join (the decision must be the same on every
	// rank - a rank without particles included -, or the ranks would issue their collectives in different orders).
	const bool overlap = s->overlap_correction && (s->dist || s->np_live);
	LFA_TRY(lfa_p2g_run(s, true, dt));
	LFA_TRY(rec(B_P2G));
	if (overlap) {
		if (s->dist) LFA_TRY(lfa_dist_exchange_ghost_particles(s));
		LFA_TRY(correct_begin(s, dt, true));
	}
	double res = 0.0;
	uint64_t it = 0;
	int rc = LFA_OK;
	auto grid_stages = [&]() -> int {
		rc = lfa_pcg_solve(s, dt, &res, &it);
		if (rc < 0) return rc;
		LFA_TRY(rec(B_SOLVE));
		LFA_TRY(lfa_apply_pressure(s, dt));
		LFA_TRY(rec(B_APPLY));
		if (!overlap) {
			LFA_TRY(lfa_correct_collide(s, dt));
			LFA_TRY(rec(B_CORRECT));
		}
		LFA_TRY(lfa_extrapolate(s));  // the valid set is the one of the P2G-time hash, like the reference (:119)
		return rec(B_EXTRAP);
	};
	const int rc_grid = grid_stages();
	// joined whatever happened in between: nothing that follows on the main stream may race with the correction
	if (overlap) LFA_TRY(lfa_corr_join(s));
	if (rc_grid < 0) return rc_grid;
	if (overlap && s->dist) LFA_TRY(lfa_dist_migrate(s, true));
	LFA_TRY(rec(B_JOIN));
	// The G2P gathers per tile. The particles keep the order of the P2G-time binning and the few whose corrected position left
	// their tile take the global-gather path (lfa_g2p_stale); like after lfa_advect_collide the order is stale afterwards and
	// the next step re-bins. Slabs: the same - particles that went to a neighbour rank carry an invalid key (skipped), the ones
	// that arrived sit behind the binned ones and join the leaver list. (Round 2 re-binned here: +18 % on a one-rank slab run.)
	LFA_TRY(lfa_g2p_stale(s));
	if (tm) {
		LFA_TRY(rec(B_G2P));
		LFA_HIP(s, hipEventSynchronize(s->ev[B_G2P]));
		auto span = [&](int a, int b, double &out) -> int {
			float ms = 0.f;
			LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[a], s->ev[b]));
			out = ms;
			return LFA_OK;
		};
		double *m = s->ms_next;
		const bool solved = it || s->n_ptiles;  // ev[18] (end of the system build) is recorded inside a non-trivial solve
		LFA_TRY(span(B_START, B_ADVECT, m[0]));
		LFA_TRY(span(B_ADVECT, B_BIN, m[1]));
		LFA_TRY(span(B_BIN, B_P2G, m[2]));
		LFA_TRY(span(16, 17, m[3]));
		if (solved) {
			LFA_TRY(span(B_P2G, 18, m[4]));
			LFA_TRY(span(18, B_SOLVE, m[5]));
		} else {
			LFA_TRY(span(B_P2G, B_SOLVE, m[4]));
			m[5] = 0.0;
		}
		LFA_TRY(span(B_SOLVE, B_APPLY, m[6]));
		if (s->np_live) {
			LFA_TRY(span(40, 41, m[7]));
			LFA_TRY(span(41, 42, m[8]));
		} else {
			m[7] = m[8] = 0.0;
		}
		// overlapped: the correction's own span on its stream (its kernels share the device with the solve's)
		if (overlap && !s->np_live) {
			m[9] = 0.0;
		} else {
			LFA_TRY(span(overlap ? 40 : B_APPLY, B_CORRECT, m[9]));
		}
		LFA_TRY(span(overlap ? B_APPLY : B_CORRECT, B_EXTRAP, m[10]));
		LFA_TRY(span(B_JOIN, B_G2P, m[11]));
		m[15] = overlap ? 1.0 : 0.0;
		LFA_TRY(span(B_START, B_G2P, m[12]));
		m[13] = (double)it;
		m[14] = it ? m[5] / (double)it : 0.0;
	}
	if (residual) *residual = res;
	if (iterations) *iterations = it;
	return rc;
}

extern "C" int lfa_get_correction_stats(lfa_sim *s, uint64_t stats[2]) {
	if (!s || !stats) return LFA_E_INVALID;
	stats[0] = stats[1] = 0;
	uint64_t ex[3];
	LFA_TRY(lfa_get_correction_stats_ex(s, ex));
	stats[0] = ex[0];
	stats[1] = ex[1];
	return LFA_OK;
}

extern "C" int lfa_get_correction_stats_ex(lfa_sim *s, uint64_t stats[3]) {
	if (!s || !stats) return LFA_E_INVALID;
	stats[0] = stats[1] = stats[2] = 0;
	if (!s->corr_ovf) return LFA_OK;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_join(s));
	const size_t ovf_stride = ((size_t)CORR_PARTS * s->g.nt + 31) / 32 + 2;
	uint32_t first = 0, second = 0;
	LFA_HIP(s, hipMemcpyAsync(&first, s->corr_ovf, 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipMemcpyAsync(&second, s->corr_ovf + ovf_stride, 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	stats[0] = second;
	stats[1] = (uint64_t)CORR_PARTS * (uint64_t)s->corr_parts_tiles;
	stats[2] = first;
	return LFA_OK;
}

extern "C" int lfa_set_step_overlap(lfa_sim *s, int on) {
	if (!s) return LFA_E_INVALID;
	s->overlap_correction = on != 0;
	return LFA_OK;
}

extern "C" int lfa_get_step_timings(lfa_sim *s, double ms[LFA_NUM_STEP_TIMERS]) {
	if (!s || !ms) return LFA_E_INVALID;
	for (int k = 0; k < LFA_NUM_STEP_TIMERS; ++k) ms[k] = s->ms_next[k];
	return LFA_OK;
}
