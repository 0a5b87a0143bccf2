// libfluid_amd/csrc/pcg.hip -- MIC(0)-preconditioned conjugate gradient on the 7-point MAC Laplacian
// (reference src/pressure_solver.cpp:19-71,244-370).
//
// Layout: every PCG vector is a tile-major field over the padded grid; only tiles that hold unknowns (the particle
// tiles) are ever touched, and entries of non-unknown cells inside those tiles are kept at exactly 0, so the matrix
// rows need no index map (the reference's grid3<size_t> _fluid_cell_indices, pressure_solver.h:52): the off-diagonal
// of row i towards -d is [type(i)==fluid] (that IS _a[j].fluid_dpos for an unknown j) and towards +d is bit d of a_i.
//
// Work decomposition: ONE WAVE OWNS ONE TILE. Lane = (x,y) column, the 8 z-cells of the column sit in registers, so a
// tile field is read with eight 256-B coalesced wave loads. x/y neighbours and the one-cell halo go through a small
// per-wave LDS block; waves never synchronise with each other (no s_barrier in the loops).
//
// MIC(0): forward/backward substitution only depends on -x,-y,-z / +x,+y,+z neighbours, so cells with equal
// i+j+k are independent. Inside a tile the wave sweeps the 22 hyperplanes with lane (x,y) working on z = level-x-y.
//   LFA_PRECOND_MIC0_TILED  couplings across tile faces are dropped from the preconditioner (block MIC(0)): factor and
//                           both sweeps of a tile stay inside one wave, one launch applies M^-1 to every tile.
//   LFA_PRECOND_MIC0_EXACT  tiles are launched hyperplane by hyperplane (tx+ty+tz) and read their predecessors' face
//                           values: the same recurrence, hence the same numbers, as pressure_solver.cpp:244-332.
//
// Reductions are deterministic (pcg.h): fixed tile->wave->workgroup assignment, fixed-order partial sums.
#include "pcg.h"

#include <math.h>
#include <string.h>

#include <stdlib.h>

#include <algorithm>

namespace {

#define WAVE_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <typename real> struct Vecs {
	real *p, *r, *z, *s, *pre, *q;
};

struct TileCtx {
	const int *ptiles;
	int n_ptiles;
	const int *tile_pslot;
	GridDims g;
};

/// Tile ids of the six face neighbours that hold unknowns (-1 otherwise): order -x,+x,-y,+y,-z,+z.
__device__ inline void face_neighbours(const TileCtx &tc, int tile, int nb[6]) {
	int tx, ty, tz;
	tile_coords(tc.g, tile, tx, ty, tz);
	const int sy = tc.g.ntx, sz = tc.g.ntx * tc.g.nty;
	nb[0] = tx > 0 ? tile - 1 : -1;
	nb[1] = tx + 1 < tc.g.ntx ? tile + 1 : -1;
	nb[2] = ty > 0 ? tile - sy : -1;
	nb[3] = ty + 1 < tc.g.nty ? tile + sy : -1;
	nb[4] = tz > 0 ? tile - sz : -1;
	nb[5] = tz + 1 < tc.g.ntz ? tile + sz : -1;
#pragma unroll
	for (int k = 0; k < 6; ++k)
		if (nb[k] >= 0 && tc.tile_pslot[nb[k]] < 0) nb[k] = -1;
}

/// Sum / max of the per-workgroup partials of the previous kernel, identical in every workgroup (256 threads): strided
/// private sums, an xor butterfly inside each wave (every lane ends with the wave's total), one exchange through LDS.
__device__ inline double reduce_partials_sum(const double *part, int n, double *lds) {
	double a = strided_partial_sum(part, n);
	a = wave_sum(a);
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
	__syncthreads();
	const double r = (lds[0] + lds[1]) + (lds[2] + lds[3]);
	__syncthreads();
	return r;
}
/// Two sums at once (one LDS exchange).
__device__ inline void reduce_partials_sum2(const double *pa, int na, const double *pb, int nb, double *lds, double &ra, double &rb) {
	double a = strided_partial_sum(pa, na), b = strided_partial_sum(pb, nb);
	a = wave_sum(a);
	b = wave_sum(b);
	if ((threadIdx.x & 63) == 0) {
		lds[threadIdx.x >> 6] = a;
		lds[4 + (threadIdx.x >> 6)] = b;
	}
	__syncthreads();
	ra = (lds[0] + lds[1]) + (lds[2] + lds[3]);
	rb = (lds[4] + lds[5]) + (lds[6] + lds[7]);
	__syncthreads();
}
__device__ inline double nan_max(double x, double y) { return (x != x || y != y) ? NAN : (y > x ? y : x); }
__device__ inline double reduce_partials_max(const double *part, int n, double *lds) {
	bool nan = false;
	double a = strided_partial_max(part, n, nan);
	a = wave_max(a);
	nan = __any(nan);
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = nan ? NAN : a;
	__syncthreads();
	const double r = nan_max(nan_max(lds[0], lds[1]), nan_max(lds[2], lds[3]));
	__syncthreads();
	return r;
}
__device__ inline void block_partial_sum(double acc, double *lds, double *out, int slot = -1) {
	acc = wave_sum(acc);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) out[slot >= 0 ? slot : (int)blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// ================================================================================================= SpMV + dot
/// z = A s (pressure_solver::_apply_a, src/pressure_solver.cpp:334-362, same term order) and partial dot(z, s).
template <typename real>
__global__ void __launch_bounds__(256)
k_spmv(TileCtx tc, const uint8_t *abits, const real *s, real *z, real scale, double *part_zs, const int *state) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double red[4];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	double acc = 0.0;
	if (state[0] < 0) {
		real *h = halo[wid];
		for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
			const int tile = tc.ptiles[slot];
			int nb[6];
			face_neighbours(tc, tile, nb);
			const size_t base = (size_t)tile * LFA_TILE_CELLS;
			WAVE_SYNC();
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = s[base + zz * 64 + lane];
			// faces: lane = (a, b) over the two in-face axes
			h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = nb[0] >= 0 ? s[(size_t)nb[0] * 512 + ly * 64 + lx * 8 + 7] : (real)0;
			h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = nb[1] >= 0 ? s[(size_t)nb[1] * 512 + ly * 64 + lx * 8 + 0] : (real)0;
			h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = nb[2] >= 0 ? s[(size_t)nb[2] * 512 + ly * 64 + 7 * 8 + lx] : (real)0;
			h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = nb[3] >= 0 ? s[(size_t)nb[3] * 512 + ly * 64 + 0 * 8 + lx] : (real)0;
			h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = nb[4] >= 0 ? s[(size_t)nb[4] * 512 + 7 * 64 + lane] : (real)0;
			h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = nb[5] >= 0 ? s[(size_t)nb[5] * 512 + 0 * 64 + lane] : (real)0;
			WAVE_SYNC();
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const int i = (lx + 1) + 10 * (ly + 1) + 100 * (zz + 1);
				const uint8_t a = abits[base + zz * 64 + lane];
				real out = (real)0;
				if (a & AB_UNKNOWN) {
					const real F = (a & AB_FLUID) ? (real)1 : (real)0;
					const real si = h[i];
					real val = (real)(a & 7) * si;
					val = madd01(-F, h[i - 1], val);
					val = madd01(-F, h[i - 10], val);
					val = madd01(-F, h[i - 100], val);
					val = madd01(-(real)((a >> 3) & 1), h[i + 1], val);
					val = madd01(-(real)((a >> 4) & 1), h[i + 10], val);
					val = madd01(-(real)((a >> 5) & 1), h[i + 100], val);
					out = scale * val;
					acc += (double)out * (double)si;
				}
				z[base + zz * 64 + lane] = out;
			}
		}
	}
	block_partial_sum(acc, red, part_zs);
}

/// r -= q and (ZERO) p = 0 over the particle tiles: the residual of a warm start, r = b - A p_guess.
template <typename real, bool ZERO>
__global__ void __launch_bounds__(256) k_warm_residual(TileCtx tc, real *r, const real *q, real *p) {
	for (int slot = blockIdx.x; slot < tc.n_ptiles; slot += gridDim.x) {
		const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
		for (int l = threadIdx.x; l < LFA_TILE_CELLS; l += 256) {
			if (ZERO) p[base + l] = (real)0;
			else r[base + l] -= q[base + l];
		}
	}
}

// ================================================================================================= AXPY x2 + max
/// p += alpha s ; r += (-alpha) z (_muladd, src/pressure_solver.cpp:364-370) ; signed max of r over the unknowns (:54).
template <typename real>
__global__ void __launch_bounds__(256)
k_axpy_max(TileCtx tc, const uint8_t *abits, Vecs<real> v, const double *part_sigma, int n_sigma, const double *part_zs,
           int n_part, double *part_rmax, const int *state, real *coarse_r, const int *slot_l1) {
	__shared__ double lds[256];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	double m = -INFINITY;
	bool nan = false;
	if (state[0] < 0) {
		const double sigma = reduce_partials_sum(part_sigma, n_sigma, lds);
		const double zs = reduce_partials_sum(part_zs, n_part, lds);
		const real alpha = (real)(sigma / zs);
		for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
			const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
			double sr = 0.0;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t b = base + zz * 64 + lane;
				if (abits[b] & AB_UNKNOWN) {
					v.p[b] = v.p[b] + alpha * v.s[b];
					const real rn = v.r[b] + (-alpha) * v.z[b];
					v.r[b] = rn;
					nan |= rn != rn;
					m = (double)rn > m ? (double)rn : m;
					sr += (double)rn;
				}
			}
			if (coarse_r) {  // restriction of the new residual for the coarse levels (they run beside the fine sweep)
				sr = wave_sum(sr);
				if (lane == 0) coarse_r[slot_l1[slot]] = (real)sr;
			}
		}
	}
	m = wave_max(m);
	nan = __any(nan);
	__syncthreads();
	if (lane == 0) lds[wid] = nan ? NAN : m;
	__syncthreads();
	if (threadIdx.x == 0) {
		double r = lds[0];
		for (int i = 1; i < 4; ++i) r = (r != r || lds[i] != lds[i]) ? NAN : (lds[i] > r ? lds[i] : r);
		part_rmax[blockIdx.x] = r;
	}
}

// ================================================================================================= MIC(0) factor
/// pressure_solver::_compute_preconditioner (src/pressure_solver.cpp:244-294). EXACT: one launch per tile hyperplane,
/// predecessors' faces read from global memory; otherwise the tile is factored on its own.
template <typename real, bool EXACT>
__global__ void __launch_bounds__(256)
k_mic_factor(TileCtx tc, const int *slots, int n_slots, const uint8_t *abits, real *pre, real scale, real tau,
             real sigma) {
	__shared__ real lpre[PCG_WAVES][LFA_TILE_CELLS];
	__shared__ uint8_t lab[PCG_WAVES][LFA_TILE_CELLS];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *P = lpre[wid];
	uint8_t *A = lab[wid];
	for (int k = blockIdx.x * PCG_WAVES + wid; k < n_slots; k += gridDim.x * PCG_WAVES) {
		const int slot = slots ? slots[k] : k;
		const int tile = tc.ptiles[slot];
		const size_t base = (size_t)tile * LFA_TILE_CELLS;
		int nb[6] = {-1, -1, -1, -1, -1, -1};
		if (EXACT) face_neighbours(tc, tile, nb);
		WAVE_SYNC();
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			A[zz * 64 + lane] = abits[base + zz * 64 + lane];
			P[zz * 64 + lane] = (real)0;
		}
		WAVE_SYNC();
		for (int level = 0; level < 22; ++level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				const uint8_t a = A[idx];
				if (a & AB_UNKNOWN) {
					real neg_e = (real)0, neg_e_tau = (real)0;
#pragma unroll
					for (int d = 0; d < 3; ++d) {
						const int c = d == 0 ? lx : (d == 1 ? ly : zz);
						const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
						uint8_t aj = 0;
						real pj = (real)0;
						if (c > 0) {
							aj = A[idx - step];
							pj = P[idx - step];
						} else if (EXACT && nb[2 * d] >= 0) {
							const size_t j = (size_t)nb[2 * d] * LFA_TILE_CELLS + (idx + 7 * step);
							aj = abits[j];
							pj = pre[j];
						}
						if (aj & AB_UNKNOWN) {
							const int bd = (aj >> (3 + d)) & 1, bo1 = (aj >> (3 + (d + 1) % 3)) & 1,
							          bo2 = (aj >> (3 + (d + 2) % 3)) & 1;
							const real ap = (real)bd * pj;
							neg_e += ap * ap;
							neg_e_tau += (real)(bd * (bo1 + bo2)) * pj * pj;
						}
					}
					const real ns = (real)(a & 7);
					real e = ns - (neg_e + tau * neg_e_tau) * scale;
					if (e < sigma * ns) e = ns;
					P[idx] = (real)1 / sqrt(e * scale);
				}
			}
			WAVE_SYNC();
		}
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) pre[base + zz * 64 + lane] = P[zz * 64 + lane];
	}
}

// ================================================================================================= coarse block
template <typename real> struct CoarseFields {
	const float *diag, *w[3];
	const uint8_t *unk;
	real *pre, *r, *x;
	real w1 = (real)1, w2 = (real)1;  // weights of the level-1 / level-2 corrections in the additive preconditioner
	// fused iteration (k_pcg_b): the restricted residual is r1 = r_prev - alpha * as, with r_prev the exact restriction
	// of the previous residual and as = P^T A s (tile sums written by k_pcg_a)
	const real *as = nullptr;
	real alpha = (real)0;
};


/// All coarse work of one preconditioner application in ONE workgroup of WAVES waves (at most 64 level-1 blocks):
/// per level-1 block a wave stages the coefficients in LDS and runs the weighted forward/backward substitution, then
/// the dense top-level solve, the prolongation of its result onto level 1 and the coarse share of z.r. Latency-bound
/// by design (44 dependent LDS hyperplanes per block), so everything the sweeps touch sits in LDS.
/// LDS: WAVES * 3 * 512 * (sizeof(real) + 1) + 144 * 8 bytes.
template <typename real, int WAVES>
__device__ inline void coarse_block(char *smem, const int *l1_tiles, int n_l1, CoarseFields<real> c, real scale,
                                    const real *a2inv, real *x2_out, double *part_sigma_extra, int cb = 0, int ncb = 1,
                                    double *xchg = nullptr, unsigned *ticket = nullptr) {
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	// per wave: PRE, Q, PQ (real) + the three face weights as bytes (a face of a tile has at most 64 couplings)
	real *wbase = (real *)smem + (size_t)wid * 3 * LFA_TILE_CELLS;
	real *PRE = wbase, *Q = wbase + LFA_TILE_CELLS, *PQ = wbase + 2 * LFA_TILE_CELLS;
	uint8_t *ub = (uint8_t *)((real *)smem + (size_t)WAVES * 3 * LFA_TILE_CELLS) + (size_t)wid * 3 * LFA_TILE_CELLS;
	uint8_t *U[3] = {ub, ub + LFA_TILE_CELLS, ub + 2 * LFA_TILE_CELLS};
	double *r2s = (double *)(smem + (size_t)WAVES * 3 * LFA_TILE_CELLS * (sizeof(real) + 1));
	double *x2s = r2s + 64;
	double *red = x2s + 64;
	// xchg (global, several coarse workgroups): [0,64) r2 per level-1 block, [64, 64+ncb) per-workgroup x1.r1
	if (ncb > 1) __builtin_amdgcn_s_setprio(3);  // these waves are the critical path of the launch they are embedded in
	double dot1 = 0.0;
	for (int k = cb * WAVES + wid; k < n_l1; k += ncb * WAVES) {
		const size_t base = (size_t)l1_tiles[k] * LFA_TILE_CELLS;
		double sr = 0.0;
		real rr[8];
		WAVE_SYNC();
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const int idx = zz * 64 + lane;
			const bool u = c.unk[base + idx] != 0;
			U[0][idx] = (uint8_t)c.w[0][base + idx];
			U[1][idx] = (uint8_t)c.w[1][base + idx];
			U[2][idx] = (uint8_t)c.w[2][base + idx];
			PRE[idx] = u ? c.pre[base + idx] : (real)0;
			rr[zz] = u ? c.r[base + idx] : (real)0;
			if (c.as) rr[zz] = u ? rr[zz] + (-c.alpha) * c.as[base + idx] : (real)0;
			Q[idx] = rr[zz];
			PQ[idx] = (real)0;
			sr += (double)rr[zz];
		}
		sr = wave_sum(sr);
		if (lane == 0) {
			if (ncb > 1) xchg[k] = sr;
			else r2s[k] = sr;
		}
		WAVE_SYNC();
#pragma clang loop unroll(disable)
		for (int level = 0; level < 22; ++level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				real t = (real)0;
				if (lx > 0) t += (real)U[0][idx - 1] * PQ[idx - 1];
				if (ly > 0) t += (real)U[1][idx - 8] * PQ[idx - 8];
				if (zz > 0) t += (real)U[2][idx - 64] * PQ[idx - 64];
				const real p = PRE[idx];
				const real q = (Q[idx] + scale * t) * p;
				Q[idx] = q;
				PQ[idx] = p * q;
			}
			WAVE_SYNC();
		}
#pragma clang loop unroll(disable)
		for (int level = 21; level >= 0; --level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				real t = (real)0;
				if (lx < 7) t += (real)U[0][idx] * Q[idx + 1];
				if (ly < 7) t += (real)U[1][idx] * Q[idx + 8];
				if (zz < 7) t += (real)U[2][idx] * Q[idx + 64];
				const real p = PRE[idx];
				Q[idx] = (Q[idx] + scale * p * t) * p;
			}
			WAVE_SYNC();
		}
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const real x1 = c.w1 * Q[zz * 64 + lane];
			c.x[base + zz * 64 + lane] = x1;
			dot1 += (double)x1 * (double)rr[zz];
		}
	}
	dot1 = wave_sum(dot1);
	if (lane == 0) red[wid] = dot1;
	__syncthreads();
	double dsum = 0.0;
	for (int w = 0; w < WAVES; ++w) dsum += red[w];
	if (ncb > 1) {
		// several coarse workgroups: the last one to arrive does the top-level solve. Hand-off by the agent-scope
		// release / ticket / acquire recipe (cdna_hip_programming.md section 6, Guideline 16): every wave drains its
		// stores, one lane releases and draws the ticket, the last arriver acquires before it reads.
		if (threadIdx.x == 0) xchg[64 + cb] = dsum;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		unsigned *flag = (unsigned *)(red + WAVES);
		if (threadIdx.x == 0) {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const unsigned last = t == (unsigned)(ncb - 1) ? 1u : 0u;
			if (last) {
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
				__hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
			}
			*flag = last;
		}
		__syncthreads();
		if (!*flag) return;
		for (int k = threadIdx.x; k < n_l1; k += WAVES * 64)
			r2s[k] = __hip_atomic_load(xchg + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		dsum = 0.0;
		for (int b = 0; b < ncb; ++b) dsum += __hip_atomic_load(xchg + 64 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__syncthreads();
	// top level: x2 = A2^-1 r2 (dense). Its prolongation onto the tiles happens in k_update_s; its share of z.r is x2.r2.
	// The inverse (<= 64 x 64) is staged in LDS with one coalesced pass (the sweep buffers are free now): a row loop over
	// global memory would be n_l1 dependent L2 round trips on the critical path of the launch.
	real *a2s = (real *)smem;
	for (int i = threadIdx.x; i < n_l1 * n_l1; i += WAVES * 64) a2s[i] = a2inv[i];
	__syncthreads();
	for (int row = threadIdx.x; row < n_l1; row += WAVES * 64) {
		double acc = 0.0;
		for (int k = 0; k < n_l1; ++k) acc += (double)a2s[row * n_l1 + k] * r2s[k];
		acc *= (double)c.w2;
		x2s[row] = acc * r2s[row];
		x2_out[row] = (real)acc;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		double t = dsum;
		for (int k = 0; k < n_l1; ++k) t += x2s[k];
		*part_sigma_extra = t;
	}
}

template <typename real, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
k_coarse_all(const int *l1_tiles, int n_l1, CoarseFields<real> c, real scale, const real *a2inv, real *x2,
             double *part_sigma_extra, const int *state) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	if (state[0] >= 0) return;
	coarse_block<real, WAVES>(smem, l1_tiles, n_l1, c, scale, a2inv, x2, part_sigma_extra);
}

// ================================================================================================= MIC(0) apply
enum { SWEEP_BOTH = 0, SWEEP_FWD = 1, SWEEP_BWD = 2 };

/// pressure_solver::_apply_preconditioner (src/pressure_solver.cpp:296-332).
///  SWEEP_BOTH (tiled): z = M^-1 r for every tile in one launch + partial dot(z, r); checks convergence first.
///  SWEEP_FWD / SWEEP_BWD (exact): one tile hyperplane per launch; q is kept in v.q between the two passes.
struct CoarseArgs {
	const int *l1_tiles;
	int n_l1;
	const void *a2inv;
	void *x2;
	double *xchg;
	unsigned *ticket;
};
#define PCG_COARSE_BLOCKS 4  // workgroups of a k_mic_apply launch that do the coarse levels (16 waves in total)

/// EMBED: workgroup 0 of the launch does the coarse levels of the multilevel preconditioner (coarse_block) while the
/// other workgroups sweep the tiles; it is dispatched first and is the longest-running workgroup, so the coarse levels
/// cost no launch, no stream hand-off and no extra time.
template <typename real, int MODE, bool EMBED>
__global__ void __launch_bounds__(256)
k_mic_apply(TileCtx tc, const int *slots, int n_slots, const uint8_t *abits, Vecs<real> v, real scale,
            double *part_sigma, const int *state, real *coarse_r, const int *slot_l1, CoarseFields<real> cf,
            CoarseArgs ca) {
	constexpr bool EXACT = MODE != SWEEP_BOTH;
	constexpr int FINE_LDS = PCG_WAVES * LFA_TILE_CELLS * (3 * (int)sizeof(real) + 1);
	constexpr int COARSE_LDS = PCG_WAVES * 3 * LFA_TILE_CELLS * ((int)sizeof(real) + 1) + 160 * 8;
	__shared__ __attribute__((aligned(16))) char lds_raw[EMBED ? (COARSE_LDS > FINE_LDS ? COARSE_LDS : FINE_LDS) : FINE_LDS];
	__shared__ double red[4];
	const int nblk = EMBED ? (int)gridDim.x - PCG_COARSE_BLOCKS : (int)gridDim.x,
	          blk = EMBED ? (int)blockIdx.x - PCG_COARSE_BLOCKS : (int)blockIdx.x;
	if (EMBED && blockIdx.x < PCG_COARSE_BLOCKS) {
		if (state[0] < 0)
			coarse_block<real, PCG_WAVES>(lds_raw, ca.l1_tiles, ca.n_l1, cf, scale, (const real *)ca.a2inv, (real *)ca.x2,
			                              part_sigma + nblk, (int)blockIdx.x, PCG_COARSE_BLOCKS, ca.xchg, ca.ticket);
		return;
	}
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *Q = (real *)lds_raw + (size_t)wid * LFA_TILE_CELLS;                                   // r -> q -> z
	real *PQ = (real *)lds_raw + (size_t)(PCG_WAVES + wid) * LFA_TILE_CELLS;                    // pre * q
	real *P = (real *)lds_raw + (size_t)(2 * PCG_WAVES + wid) * LFA_TILE_CELLS;
	uint8_t *A = (uint8_t *)((real *)lds_raw + (size_t)3 * PCG_WAVES * LFA_TILE_CELLS) + (size_t)wid * LFA_TILE_CELLS;
	double acc = 0.0;
	if (state[0] < 0) {
		for (int k = blk * PCG_WAVES + wid; k < n_slots; k += nblk * PCG_WAVES) {
			const int slot = slots ? slots[k] : k;
			const int tile = tc.ptiles[slot];
			const size_t base = (size_t)tile * LFA_TILE_CELLS;
			int nb[6] = {-1, -1, -1, -1, -1, -1};
			if (EXACT) face_neighbours(tc, tile, nb);
			real rr[8];
			WAVE_SYNC();
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const size_t b = base + zz * 64 + lane;
				A[zz * 64 + lane] = abits[b];
				P[zz * 64 + lane] = v.pre[b];
				if (MODE == SWEEP_BWD) {
					Q[zz * 64 + lane] = v.q[b];
				} else {
					rr[zz] = v.r[b];
					Q[zz * 64 + lane] = rr[zz];
					PQ[zz * 64 + lane] = (real)0;
				}
			}
			WAVE_SYNC();
			if (MODE == SWEEP_BOTH && coarse_r) {
				// restriction to the coarse space (one piecewise-constant unknown per tile): r1 = sum of r over the tile
				double sr = 0.0;
#pragma unroll
				for (int zz = 0; zz < 8; ++zz) sr += (double)rr[zz];
				sr = wave_sum(sr);
				if (lane == 0) coarse_r[slot_l1[slot]] = (real)sr;
			}
			if (MODE != SWEEP_BWD) {
				// L q = r
#pragma clang loop unroll(disable)
				for (int level = 0; level < 22; ++level) {
					const int zz = level - lx - ly;
					if (zz >= 0 && zz < 8) {
						const int idx = zz * 64 + lane;
						const uint8_t a = A[idx];
						real q = (real)0;
						if (a & AB_UNKNOWN) {
							const real F = (a & AB_FLUID) ? (real)1 : (real)0;
							real t = (real)0;
#pragma unroll
							for (int d = 0; d < 3; ++d) {
								const int c = d == 0 ? lx : (d == 1 ? ly : zz);
								const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
								real pqj = (real)0;
								if (c > 0) {
									pqj = PQ[idx - step];
								} else if (EXACT && nb[2 * d] >= 0) {
									const size_t j = (size_t)nb[2 * d] * LFA_TILE_CELLS + (idx + 7 * step);
									pqj = v.pre[j] * v.q[j];
								}
								t += F * pqj;
							}
							q = (Q[idx] + scale * t) * P[idx];
							PQ[idx] = P[idx] * q;
						}
						Q[idx] = q;
					}
					WAVE_SYNC();
				}
			}
			if (MODE == SWEEP_FWD) {
#pragma unroll
				for (int zz = 0; zz < 8; ++zz) v.q[base + zz * 64 + lane] = Q[zz * 64 + lane];
				continue;
			}
			// L^T z = q
#pragma clang loop unroll(disable)
			for (int level = 21; level >= 0; --level) {
				const int zz = level - lx - ly;
				if (zz >= 0 && zz < 8) {
					const int idx = zz * 64 + lane;
					const uint8_t a = A[idx];
					real zv = (real)0;
					if (a & AB_UNKNOWN) {
						real t = (real)0;
#pragma unroll
						for (int d = 0; d < 3; ++d) {
							const int c = d == 0 ? lx : (d == 1 ? ly : zz);
							const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
							real zj = (real)0;
							if (c < 7) {
								zj = Q[idx + step];
							} else if (EXACT && nb[2 * d + 1] >= 0) {
								zj = v.z[(size_t)nb[2 * d + 1] * LFA_TILE_CELLS + (idx - 7 * step)];
							}
							t += (real)((a >> (3 + d)) & 1) * zj;
						}
						zv = (Q[idx] + scale * P[idx] * t) * P[idx];
					}
					Q[idx] = zv;
				}
				WAVE_SYNC();
			}
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) {
				const real zv = Q[zz * 64 + lane];
				v.z[base + zz * 64 + lane] = zv;
				if (MODE == SWEEP_BOTH) acc += (double)zv * (double)rr[zz];
			}
		}
	}
	if (MODE == SWEEP_BOTH) block_partial_sum(acc, red, part_sigma, blk);
}

/// partial dot(z, r) (exact-MIC path, where the sweeps are separate launches).
template <typename real>
__global__ void __launch_bounds__(256)
k_dot_zr(TileCtx tc, Vecs<real> v, double *part_sigma, const int *state) {
	__shared__ double red[4];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	double acc = 0.0;
	if (state[0] < 0) {
		for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
			const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
#pragma unroll
			for (int zz = 0; zz < 8; ++zz) acc += (double)v.z[base + zz * 64 + lane] * (double)v.r[base + zz * 64 + lane];
		}
	}
	block_partial_sum(acc, red, part_sigma);
}

/// Stopping rule of pressure_solver::solve (src/pressure_solver.cpp:54-58): signed max(r) < tolerance ends the solve
/// with iteration count i+1. One workgroup; later kernels of the stream see state[0] >= 0 and do nothing.
__global__ void __launch_bounds__(256)
k_check_converged(const double *part_rmax, int n_part, double tol, int iter, int *state, double *hist, const double *gave_up = nullptr) {
	__shared__ double lds[256];
	// slabs: how many ranks gave a device-side wait up (the last word of the gather, dist.hip: k_gather_pair). The retreat is
	// collective: a rank that leaves the loop alone would redo the system build and its collectives while its peers carry on
	// with the V-cycle's, and the transport sequences would no longer pair up.
	if (gave_up && threadIdx.x == 0 && *gave_up > 0.0 && state[2] == 0) state[2] = 1;
	if (state[0] >= 0) return;
	const double rmax = reduce_partials_max(part_rmax, n_part, lds);
	if (threadIdx.x == 0) {
		hist[iter] = rmax;
		if (rmax != rmax) {
			state[1] = 1;
			*(double *)(state + 16) = rmax;
			state[0] = iter + 1;
		} else if (rmax < tol) {
			*(double *)(state + 16) = rmax;  // (the host reads it with the state words: no read-back of its own)
			state[0] = iter + 1;
		}
	}
}

/// The early-out of pressure_solver::solve (src/pressure_solver.cpp:28-35: sum b^2 < 1e-6 returns p = 0, residual 0, 0 iterations)
/// evaluated on the device: the solve is marked done before its first kernel - which, like every kernel of the loop, does nothing
/// then - and the host learns it at its first poll instead of draining the GPU for one number ahead of every solve.
/// state[3] = 1: zero right-hand side, 2: NaN in it.
__global__ void __launch_bounds__(256) k_check_rhs(const double *part_b2, int n_part, int *state) {
	__shared__ double lds[4];
	const double tot = reduce_partials_sum(part_b2, n_part, lds);
	if (threadIdx.x == 0) {
		if (tot != tot) {
			state[3] = 2;
			state[1] = 1;
			state[0] = 0;
		} else if (tot < 1e-6) {
			state[3] = 1;
			*(double *)(state + 16) = 0.0;
			state[0] = 0;
		}
	}
}

/// s = z + beta s (src/pressure_solver.cpp:64-66); first = 1: s = z (:40). With the multilevel preconditioner the
/// coarse part of z (piecewise constant per tile, coarse_x) is added here instead of in a pass of its own.
template <typename real>
__global__ void __launch_bounds__(256)
k_update_s(TileCtx tc, Vecs<real> v, const double *part_sig_new, const double *part_sig_old, int n_part, int first,
           const int *state, const uint8_t *abits, const real *coarse_x, const int *slot_l1, const real *coarse_x2,
           const int *l1_l2) {
	__shared__ double lds[256];
	if (state[0] >= 0) return;
	real beta = (real)0;
	if (!first) {
		const double sn = reduce_partials_sum(part_sig_new, n_part, lds);
		const double so = reduce_partials_sum(part_sig_old, n_part, lds);
		beta = (real)(sn / so);
	}
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
		const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
		real xc = (real)0;
		if (coarse_x) {  // prolongation of both coarse levels: level-1 value of the tile + top-level value of its block
			const int i1 = slot_l1[slot];
			xc = coarse_x[i1] + coarse_x2[l1_l2[i1 >> 9]];
		}
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const size_t b = base + zz * 64 + lane;
			real z = v.z[b];
			if (coarse_x && (abits[b] & AB_UNKNOWN)) z += xc;
			v.s[b] = first ? z : z + beta * v.s[b];
		}
	}
}

// ================================================================================================= coarse levels
// Level 1: one unknown per particle tile, Galerkin operator A1 = P^T A P for piecewise-constant P:
//   A1[I,I] = scale * (sum_i ns_i - 2 * #coupled pairs inside I),  A1[I,I+d] = -scale * #couplings across the +d face.
// The level-1 unknowns live on the grid of tiles, stored tile-major again (8^3 blocks of tiles), so the same
// wave-per-tile hyperplane sweep applies with weights. Level 2: one unknown per level-1 block, A2 = P1^T A1 P1 inverted
// densely on the host (n2 = number of 64^3-cell aggregates that hold fluid: 32 at C4).

/// Level-1 coefficients from the fine A bits.
__global__ void __launch_bounds__(256)
k_coarse_coeffs(TileCtx tc, const uint8_t *abits, const int *slot_l1, float *c_diag, float *c_wx, float *c_wy, float *c_wz,
                uint8_t *c_unk) {
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
		const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
		int diag = 0, wx = 0, wy = 0, wz = 0, unk = 0;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const int a = abits[base + zz * 64 + lane];
			if (!(a & AB_UNKNOWN)) continue;
			unk = 1;
			const int bx = (a >> 3) & 1, by = (a >> 4) & 1, bz = (a >> 5) & 1;
			diag += (a & 7) - 2 * ((lx < 7 ? bx : 0) + (ly < 7 ? by : 0) + (zz < 7 ? bz : 0));
			wx += lx == 7 ? bx : 0;
			wy += ly == 7 ? by : 0;
			wz += zz == 7 ? bz : 0;
		}
		diag = wave_sum(diag); wx = wave_sum(wx); wy = wave_sum(wy); wz = wave_sum(wz);
		unk = __any(unk) ? 1 : 0;
		if (lane == 0) {
			const int i1 = slot_l1[slot];
			c_diag[i1] = (float)diag; c_wx[i1] = (float)wx; c_wy[i1] = (float)wy; c_wz[i1] = (float)wz;
			c_unk[i1] = (uint8_t)unk;
		}
	}
}

/// MIC(0) of a level-1 block (weighted version of k_mic_factor, block-local).
template <typename real>
__global__ void __launch_bounds__(256)
k_coarse_factor(const int *l1_tiles, int n_l1, CoarseFields<real> c, real scale, real tau, real sigma) {
	__shared__ real lpre[PCG_WAVES][LFA_TILE_CELLS];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *P = lpre[wid];
	for (int k = blockIdx.x * PCG_WAVES + wid; k < n_l1; k += gridDim.x * PCG_WAVES) {
		const size_t base = (size_t)l1_tiles[k] * LFA_TILE_CELLS;
		WAVE_SYNC();
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) P[zz * 64 + lane] = (real)0;
		WAVE_SYNC();
		for (int level = 0; level < 22; ++level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				if (c.unk[base + idx]) {
					real neg = (real)0;
#pragma unroll
					for (int d = 0; d < 3; ++d) {
						const int cd = d == 0 ? lx : (d == 1 ? ly : zz);
						const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
						if (cd > 0 && c.unk[base + idx - step]) {
							const size_t j = base + idx - step;
							const real pj = P[idx - step];
							const real ad = scale * (real)c.w[d][j], ao1 = scale * (real)c.w[(d + 1) % 3][j],
							           ao2 = scale * (real)c.w[(d + 2) % 3][j];
							neg += (ad * pj) * (ad * pj) + tau * ad * (ao1 + ao2) * pj * pj;
						}
					}
					const real aii = scale * (real)c.diag[base + idx];
					real e = aii - neg;
					if (e < sigma * aii) e = aii;
					P[idx] = (real)1 / sqrt(e);
				}
			}
			WAVE_SYNC();
		}
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) c.pre[base + zz * 64 + lane] = P[zz * 64 + lane];
	}
}

/// x1 = M1^-1 r1 per level-1 block (weighted forward/backward substitution) and r2 = P1^T r1.
template <typename real>
__global__ void __launch_bounds__(256)
k_coarse_apply(const int *l1_tiles, int n_l1, CoarseFields<real> c, real scale, real *r2) {
	__shared__ real lq[PCG_WAVES][LFA_TILE_CELLS];
	__shared__ real lpq[PCG_WAVES][LFA_TILE_CELLS];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *Q = lq[wid], *PQ = lpq[wid];
	for (int k = blockIdx.x * PCG_WAVES + wid; k < n_l1; k += gridDim.x * PCG_WAVES) {
		const size_t base = (size_t)l1_tiles[k] * LFA_TILE_CELLS;
		double sr = 0.0;
		WAVE_SYNC();
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const int idx = zz * 64 + lane;
			const real r = c.unk[base + idx] ? c.r[base + idx] : (real)0;
			Q[idx] = r;
			PQ[idx] = (real)0;
			sr += (double)r;
		}
		sr = wave_sum(sr);
		if (lane == 0) r2[k] = (real)sr;
		WAVE_SYNC();
		for (int level = 0; level < 22; ++level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				real q = (real)0;
				if (c.unk[base + idx]) {
					real t = (real)0;
#pragma unroll
					for (int d = 0; d < 3; ++d) {
						const int cd = d == 0 ? lx : (d == 1 ? ly : zz);
						const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
						if (cd > 0) t += (real)c.w[d][base + idx - step] * PQ[idx - step];
					}
					const real p = c.pre[base + idx];
					q = (Q[idx] + scale * t) * p;
					PQ[idx] = p * q;
				}
				Q[idx] = q;
			}
			WAVE_SYNC();
		}
		for (int level = 21; level >= 0; --level) {
			const int zz = level - lx - ly;
			if (zz >= 0 && zz < 8) {
				const int idx = zz * 64 + lane;
				real zv = (real)0;
				if (c.unk[base + idx]) {
					real t = (real)0;
#pragma unroll
					for (int d = 0; d < 3; ++d) {
						const int cd = d == 0 ? lx : (d == 1 ? ly : zz);
						const int step = d == 0 ? 1 : (d == 1 ? 8 : 64);
						if (cd < 7) t += (real)c.w[d][base + idx] * Q[idx + step];
					}
					const real p = c.pre[base + idx];
					zv = (Q[idx] + scale * p * t) * p;
				}
				Q[idx] = zv;
			}
			WAVE_SYNC();
		}
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) c.x[base + zz * 64 + lane] = Q[zz * 64 + lane];
	}
}

/// Top level: x2 = A2^-1 r2 (dense), x1 += P1 x2, and the coarse part of the preconditioned dot product
/// sigma_coarse = x1 . r1, written as one more partial. One workgroup.
template <typename real>
__global__ void __launch_bounds__(256)
k_coarse_top(const int *l1_tiles, int n_l1, CoarseFields<real> c, const real *a2inv, const real *r2, real *x2,
             double *part_sigma_extra, const int *state) {
	__shared__ double red[4];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (state[0] >= 0) return;
	for (int row = threadIdx.x; row < n_l1; row += 256) {
		double acc = 0.0;
		for (int k = 0; k < n_l1; ++k) acc += (double)a2inv[(size_t)row * n_l1 + k] * (double)r2[k];
		x2[row] = (real)acc;
	}
	__syncthreads();
	double dotp = 0.0;
	for (int k = wid; k < n_l1; k += 4) {
		const size_t base = (size_t)l1_tiles[k] * LFA_TILE_CELLS;
		const real add = x2[k];
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const size_t i = base + zz * 64 + lane;
			if (c.unk[i]) dotp += ((double)c.x[i] + (double)add) * (double)c.r[i];  // prolongation itself: k_update_s
		}
	}
	dotp = wave_sum(dotp);
	if (lane == 0) red[wid] = dotp;
	__syncthreads();
	if (threadIdx.x == 0) *part_sigma_extra = (red[0] + red[1]) + (red[2] + red[3]);
}

/// z += P x_coarse on the unknowns (only for lfa_apply_preconditioner, which hands z to the caller).
template <typename real>
__global__ void __launch_bounds__(256)
k_add_coarse(TileCtx tc, const uint8_t *abits, real *z, const real *coarse_x, const int *slot_l1, const real *coarse_x2,
             const int *l1_l2) {
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int slot = blockIdx.x * PCG_WAVES + wid; slot < tc.n_ptiles; slot += gridDim.x * PCG_WAVES) {
		const size_t base = (size_t)tc.ptiles[slot] * LFA_TILE_CELLS;
		const int i1 = slot_l1[slot];
		const real xc = coarse_x[i1] + coarse_x2[l1_l2[i1 >> 9]];
#pragma unroll
		for (int zz = 0; zz < 8; ++zz)
			if (abits[base + zz * 64 + lane] & AB_UNKNOWN) z[base + zz * 64 + lane] += xc;
	}
}


// ================================================================================================= fused iteration
// Two launches per PCG iteration instead of five (single domain, tile-local MIC(0) with or without the coarse levels):
//   k_pcg_a : [stopping rule on the previous residual] s = z + P x_coarse + beta s ; q = A s ; partial dot(q, s)
//   k_pcg_b : p += alpha s ; r -= alpha q ; signed max r ; z = M^-1 r ; partial dot(z, r)
// Same arithmetic as k_update_s / k_spmv / k_axpy_max (expression for expression), so the two paths produce the same
// numbers up to the rounding of the re-associated tile sweeps. What the fusion needs:
//  * s is double-buffered: a wave of k_pcg_a recomputes the new s on the one-cell halo of its tile from the neighbours'
//    z and old s, while the neighbours' owners store their new s in the same launch.
//  * q = A s goes to its own buffer (vq), because z is still read as halo by other waves of k_pcg_a.
//  * the coarse levels run inside k_pcg_b beside the tile sweeps, i.e. before the new residual exists; their
//    right-hand side follows from linearity, r1 = P^T r_prev - alpha P^T A s: k_pcg_a writes the tile sums of A s,
//    k_pcg_b writes the exact tile sums of the new residual for the next iteration (no drift).
// The sweeps of k_pcg_b are the substitution W = pre q (forward) / z (backward):
//   W_i = pre_i^2 r_i + b_i (W_{i-x} + W_{i-y} + W_{i-z}),          b_i = scale pre_i^2 [i is fluid]
//   z_i = W_i + c_i ([fluid] z_{i+x} + [fluid] z_{i+y} + [fluid] z_{i+z}),  c_i = scale pre_i^2
// (pressure_solver.cpp:301-331 with q_i = W_i / pre_i), in place in one LDS array with zero planes before and behind
// it and per-lane 0/1 factors for the x/y tile faces: 3 FMAs and 5 LDS operations per cell and sweep.
#define WAVE_FENCE()                                          \
	do {                                                       \
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
		__builtin_amdgcn_wave_barrier();                       \
	} while (0)

template <typename real> struct alignas(2 * sizeof(real)) BCPair { real b, c; };
/// Four consecutive cells of a tile field: one 16-B (fp32) global or LDS access per lane.
template <typename real> struct alignas(4 * sizeof(real)) V4 {
	real e[4];
	__device__ real &operator[](int i) { return e[i]; }
	__device__ const real &operator[](int i) const { return e[i]; }
};
#define SWEEP_X_LEN (LFA_TILE_CELLS + 128)

/// X: cell 0 of a SWEEP_X_LEN array whose first and last 64 entries are zero. In: X = pre^2 r. Out: BC[i].c = z_i.
template <typename real> __device__ inline void fast_tile_sweeps(real *X, BCPair<real> *BC, int lane) {
	const int lx = lane & 7, ly = lane >> 3, t0 = lx + ly;
	const real mxm = lx > 0 ? (real)1 : (real)0, mym = ly > 0 ? (real)1 : (real)0;
	const real mxp = lx < 7 ? (real)1 : (real)0, myp = ly < 7 ? (real)1 : (real)0;
#pragma clang loop unroll(disable)
	for (int level = 0; level < 22; ++level) {
		const int zz = level - t0;
		if ((unsigned)zz < 8u) {
			const int idx = zz * 64 + lane;
			const real sum = fma(mxm, X[idx - 1], fma(mym, X[idx - 8], X[idx - 64]));
			X[idx] = fma(BC[idx].b, sum, X[idx]);
		}
		WAVE_FENCE();
	}
#pragma clang loop unroll(disable)
	for (int level = 21; level >= 0; --level) {
		const int zz = level - t0;
		if ((unsigned)zz < 8u) {
			const int idx = zz * 64 + lane;
			const real sum = fma(mxp, X[idx + 1], fma(myp, X[idx + 8], X[idx + 64]));
			const BCPair<real> bc = BC[idx];
			const real zv = fma(bc.c, sum, X[idx]);
			BC[idx].c = zv;
			X[idx] = bc.b != (real)0 ? zv : (real)0;
		}
		WAVE_FENCE();
	}
}

/// Per particle-tile slot, one 128-B row of ints (scalar loads): [0..5] tile ids of the face neighbours that hold
/// unknowns (-x,+x,-y,+y,-z,+z; -1 otherwise), [6] own tile id, [7] own level-1 index; [8..13] / [14] index into the
/// level-1 correction of the neighbours / of the tile itself, [16..21] / [22] the same for the top-level correction, so
/// that k_pcg_a finds every coarse value with one independent load.
#define NBR_STRIDE 32
/// `p_off`: slot of the first owned tile in tile_pslot's numbering (slabs: ghost tiles come first). A face neighbour on
/// another rank (ghost tile) is entered as missing: its new search direction does not exist while k_pcg_a runs, the term
/// is added by k_ghost_face_rows once the boundary slices have been exchanged.
__global__ void __launch_bounds__(256)
k_build_nbr_table(TileCtx tc, int p_off, const int *slot_l1, const int *l1_l2, int *nbr) {
	const int slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= tc.n_ptiles) return;
	const int tile = tc.ptiles[slot];
	int nb[6];
	face_neighbours(tc, tile, nb);
	int *o = nbr + (size_t)slot * NBR_STRIDE;
	for (int k = 0; k < NBR_STRIDE; ++k) o[k] = 0;
#pragma unroll
	for (int k = 0; k < 6; ++k) {
		const int ns = nb[k] >= 0 ? tc.tile_pslot[nb[k]] - p_off : -1;
		const bool own = ns >= 0 && ns < tc.n_ptiles;
		o[k] = own ? nb[k] : -1;
		if (slot_l1 && own) {
			const int j1 = slot_l1[ns];
			o[8 + k] = j1;
			o[16 + k] = l1_l2[j1 >> 9];
		}
	}
	o[6] = tile;
	if (slot_l1) {
		const int i1 = slot_l1[slot];
		o[7] = i1;
		o[14] = i1;
		o[22] = l1_l2[i1 >> 9];
	}
}

/// Slabs: rows of q = A s that couple across a slab face. k_pcg_a computed them with the face towards the other rank
/// empty; once the neighbour's boundary slice of the new s has arrived in the ghost tile, the 64 rows of the slice are
/// recomputed with k_spmv's expression (so a slab run performs the single-domain arithmetic), and dot(q, s) and the tile
/// sum of q (coarse right-hand side) are corrected by the difference. One wave per boundary tile and side.
template <typename real>
__global__ void __launch_bounds__(256)
k_ghost_face_rows(TileCtx tc, int n_lo, int hi_slot0, int n_hi, const uint8_t *abits, const real *s, real *q, real scale,
                  double *part_extra, real *coarse_as, const int *slot_l1, const int *state) {
	__shared__ double red[4];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	double acc = 0.0;
	if (state[0] < 0) {
		for (int w = blockIdx.x * PCG_WAVES + wid; w < n_lo + n_hi; w += gridDim.x * PCG_WAVES) {
			const bool lo = w < n_lo;
			const int slot = lo ? w : hi_slot0 + (w - n_lo), zs = lo ? 0 : 7;
			const int tile = tc.ptiles[slot];
			int nb[6];
			face_neighbours(tc, tile, nb);
			const int ghost = lo ? nb[4] : nb[5];
			if (ghost < 0) continue;  // no fluid across the face (wave-uniform)
			const size_t base = (size_t)tile * LFA_TILE_CELLS;
			const int idx = zs * 64 + lane;
			const uint32_t a = abits[base + idx];
			double dq = 0.0;
			if (a & AB_UNKNOWN) {
				auto at = [&](int t, int i) -> real { return t >= 0 ? s[(size_t)t * LFA_TILE_CELLS + i] : (real)0; };
				const real F = (a & AB_FLUID) ? (real)1 : (real)0;
				const real si = s[base + idx];
				const real sxm = lx > 0 ? s[base + idx - 1] : at(nb[0], idx + 7), sxp = lx < 7 ? s[base + idx + 1] : at(nb[1], idx - 7);
				const real sym = ly > 0 ? s[base + idx - 8] : at(nb[2], idx + 56), syp = ly < 7 ? s[base + idx + 8] : at(nb[3], idx - 56);
				const real szm = zs > 0 ? s[base + idx - 64] : at(nb[4], idx + 448), szp = zs < 7 ? s[base + idx + 64] : at(nb[5], idx - 448);
				real val = (real)(a & 7) * si;
				val = madd01(-F, sxm, val);
				val = madd01(-F, sym, val);
				val = madd01(-F, szm, val);
				val = madd01(-(real)((a >> 3) & 1), sxp, val);
				val = madd01(-(real)((a >> 4) & 1), syp, val);
				val = madd01(-(real)((a >> 5) & 1), szp, val);
				const real out = scale * val;
				dq = (double)out - (double)q[base + idx];
				q[base + idx] = out;
				acc += dq * (double)si;
			}
			if (coarse_as) {
				dq = wave_sum(dq);
				if (lane == 0) coarse_as[slot_l1[slot]] = (real)((double)coarse_as[slot_l1[slot]] + dq);
			}
		}
	}
	block_partial_sum(acc, red, part_extra);
}

template <typename real, bool EMBED>
__global__ void __launch_bounds__(256)
k_pcg_b(const int *__restrict__ ptiles, int n_ptiles, const uint8_t *__restrict__ abits, Vecs<real> v,
        const real *__restrict__ s_cur, real scale, const double *part_sigma_old, int n_sigma, const double *part_zs,
        int n_zs, double *part_rmax, double *part_sigma_new, const int *state, real *coarse_r_out,
        const int *__restrict__ slot_l1, CoarseFields<real> cf, CoarseArgs ca) {
	constexpr int FINE_LDS = PCG_WAVES * (SWEEP_X_LEN + 2 * LFA_TILE_CELLS) * (int)sizeof(real);
	constexpr int COARSE_LDS = PCG_WAVES * 3 * LFA_TILE_CELLS * ((int)sizeof(real) + 1) + 160 * 8;
	constexpr int RED_LDS = 256 * 8;  // the scalar reductions use the front of the same block before / after the tiles
	constexpr int WORK_LDS = EMBED ? (COARSE_LDS > FINE_LDS ? COARSE_LDS : FINE_LDS) : FINE_LDS;
	__shared__ __attribute__((aligned(16))) char lds_raw[WORK_LDS > RED_LDS ? WORK_LDS : RED_LDS];
	double *red = (double *)lds_raw;
	const int nblk = EMBED ? (int)gridDim.x - PCG_COARSE_BLOCKS : (int)gridDim.x,
	          blk = EMBED ? (int)blockIdx.x - PCG_COARSE_BLOCKS : (int)blockIdx.x;
	const bool coarse_wg = EMBED && blockIdx.x < PCG_COARSE_BLOCKS;
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	// Software pipeline over the tiles of this wave: the loads of the next tile are issued before the sweeps of the
	// current one (which only touch LDS), so HBM latency hides behind the 44 dependent hyperplanes; the loads of the first
	// tile are issued before the scalars of the previous kernel are reduced.
	// Global accesses are 16 B per lane: lane l holds cells [256 k + 4 l, 256 k + 4 l + 4), k = 0, 1, of the tile's 512.
	const int stride = nblk * PCG_WAVES;
	int slot = blk * PCG_WAVES + wid;
	uint32_t ta[2];
	V4<real> tpre[2], tp[2], ts[2], tr[2], tq[2];
	size_t base = 0;
	auto load_tile = [&](int sl) {
		base = (size_t)ptiles[sl] * LFA_TILE_CELLS;
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const size_t b = base + 256 * k + 4 * lane;
			ta[k] = *(const uint32_t *)(abits + b);
			tpre[k] = *(const V4<real> *)(v.pre + b);
			tp[k] = *(const V4<real> *)(v.p + b);
			ts[k] = *(const V4<real> *)(s_cur + b);
			tr[k] = *(const V4<real> *)(v.r + b);
			tq[k] = *(const V4<real> *)(v.q + b);
		}
	};
	if (!coarse_wg && slot < n_ptiles) load_tile(slot);
	if (state[0] >= 0) return;
	double sigma, zs;
	reduce_partials_sum2(part_sigma_old, n_sigma, part_zs, n_zs, red, sigma, zs);
	const real alpha = (real)(sigma / zs);
	if (coarse_wg) {
		cf.alpha = alpha;
		coarse_block<real, PCG_WAVES>(lds_raw, ca.l1_tiles, ca.n_l1, cf, scale, (const real *)ca.a2inv, (real *)ca.x2,
		                              part_sigma_new + nblk, (int)blockIdx.x, PCG_COARSE_BLOCKS, ca.xchg, ca.ticket);
		return;
	}
	real *X = (real *)lds_raw + (size_t)wid * SWEEP_X_LEN + 64;
	BCPair<real> *BC = (BCPair<real> *)((real *)lds_raw + (size_t)PCG_WAVES * SWEEP_X_LEN) + (size_t)wid * LFA_TILE_CELLS;
	X[lane - 64] = (real)0;
	X[LFA_TILE_CELLS + lane] = (real)0;
	double acc = 0.0, m = -INFINITY;
	bool nan = false;
	while (slot < n_ptiles) {
		real rn[8];
		double sr = 0.0;
		WAVE_FENCE();
		// entries of non-unknown cells are exact zeros in r, q and pre, so only p and r are masked
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const size_t b = base + 256 * k + 4 * lane;
			const int c0 = 256 * k + 4 * lane;
			V4<real> pn, rv, xv;
			BCPair<real> bc[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				const uint32_t a = ta[k] >> (8 * j);
				const bool unk = (a & AB_UNKNOWN) != 0;
				const real pj = unk ? tp[k][j] + alpha * ts[k][j] : (real)0;
				const real rr = unk ? tr[k][j] + (-alpha) * tq[k][j] : (real)0;
				nan |= rr != rr;
				if (unk) m = (double)rr > m ? (double)rr : m;
				sr += (double)rr;
				rn[4 * k + j] = rr;
				pn[j] = pj;
				rv[j] = rr;
				const real d = tpre[k][j] * tpre[k][j], c = scale * d;
				xv[j] = d * rr;
				bc[j] = BCPair<real>{(a & AB_FLUID) ? c : (real)0, c};
			}
			*(V4<real> *)(v.p + b) = pn;
			*(V4<real> *)(v.r + b) = rv;
			*(V4<real> *)(X + c0) = xv;
#pragma unroll
			for (int j = 0; j < 4; ++j) BC[c0 + j] = bc[j];
		}
		if (coarse_r_out) {  // exact restriction of the new residual: right-hand side recurrence of the next iteration
			sr = wave_sum(sr);
			if (lane == 0) coarse_r_out[slot_l1[slot]] = (real)sr;
		}
		const size_t obase = base;
		slot += stride;
		if (slot < n_ptiles) load_tile(slot);
		WAVE_FENCE();
		fast_tile_sweeps<real>(X, BC, lane);
#pragma unroll
		for (int k = 0; k < 2; ++k) {
			const int c0 = 256 * k + 4 * lane;
			V4<real> zv;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				zv[j] = BC[c0 + j].c;
				acc += (double)zv[j] * (double)rn[4 * k + j];
			}
			*(V4<real> *)(v.z + obase + c0) = zv;
		}
	}
	m = wave_max(m);
	nan = __any(nan);
	acc = wave_sum(acc);
	__syncthreads();
	if (lane == 0) {
		red[wid] = nan ? NAN : m;
		red[8 + wid] = acc;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		double r = red[0];
		for (int i = 1; i < 4; ++i) r = (r != r || red[i] != red[i]) ? NAN : (red[i] > r ? red[i] : r);
		part_rmax[blk] = r;
		part_sigma_new[blk] = (red[8] + red[9]) + (red[10] + red[11]);
	}
}

template <typename real, bool FIRST, bool COARSE>
__global__ void __launch_bounds__(256)
k_pcg_a(int n_ptiles, const int *__restrict__ nbr, const uint8_t *__restrict__ abits, const real *__restrict__ z,
        const real *__restrict__ s_old, real *__restrict__ s_new, real *__restrict__ q, real scale,
        const double *part_sig_new, int n_sig_new, const double *part_sig_old, int n_sig_old,
        const double *part_rmax, int n_rmax, double tol, int iter, int *state, double *hist, double *part_qs,
        const real *__restrict__ coarse_x,
        const real *__restrict__ coarse_x2, real *coarse_as) {
	__shared__ real halo[PCG_WAVES][LFA_HALO_CELLS];
	__shared__ double lds[256];
	const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
	real *h = halo[wid];
	double acc = 0.0;
	// XCD-aware slot order: workgroup b runs on XCD b % 8 (observed, speed only), so the workgroups of one XCD take one
	// contiguous chunk of the (tile-id ordered) slot list and find their y/z face neighbours in their own L2.
	const int G = (int)gridDim.x, per = G >> 3, rem = G & 7, xcd = (int)blockIdx.x & 7;
	const int vblk = xcd * per + (xcd < rem ? xcd : rem) + ((int)blockIdx.x >> 3);
	const int stride = G * PCG_WAVES;
	// face lanes = (a, b) over the two in-face axes
	const int fo[6] = {ly * 64 + lx * 8 + 7, ly * 64 + lx * 8, ly * 64 + 56 + lx, ly * 64 + lx, 7 * 64 + lane, lane};
	// Software pipeline: every global load of the next tile (interior, six faces, coarse values) is in flight while the
	// current tile is computed.
	int slot = vblk * PCG_WAVES + wid;
	uint32_t ab[8], fab[6];  // one register each (see k_pcg_b)
	real zi[8], si[8], fz[6], fs[6], xc = (real)0, xn[6];
	size_t base = 0;
	int l1 = 0;
	bool fvalid[6];
	auto load_tile = [&](int sl) {
		// branch-free: a missing neighbour reads the tile's own cells (masked when consumed), every index comes from the
		// slot's table row (read as six 16-B scalar loads), so all loads of the tile are independent of each other
		const int4 *row = (const int4 *)(nbr + (size_t)sl * NBR_STRIDE);
		const int4 r0 = row[0], r1 = row[1];
		const int nbk[6] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y};
		const int own = r1.z;
		base = (size_t)own * LFA_TILE_CELLS;
		l1 = r1.w;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const size_t b = base + zz * 64 + lane;
			ab[zz] = abits[b];
			zi[zz] = z[b];
			si[zz] = FIRST ? (real)0 : s_old[b];
		}
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			fvalid[k] = nbk[k] >= 0;
			const size_t j = (size_t)(nbk[k] >= 0 ? nbk[k] : own) * LFA_TILE_CELLS + fo[k];
			fab[k] = abits[j];
			fz[k] = z[j];
			fs[k] = FIRST ? (real)0 : s_old[j];
		}
		if (COARSE) {
			const int4 r2 = row[2], r3 = row[3], r4 = row[4], r5 = row[5];
			const int i1[7] = {r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z}, i2[7] = {r4.x, r4.y, r4.z, r4.w, r5.x, r5.y, r5.z};
#pragma unroll
			for (int k = 0; k < 6; ++k) xn[k] = coarse_x[i1[k]] + coarse_x2[i2[k]];
			xc = coarse_x[i1[6]] + coarse_x2[i2[6]];
		}
	};
	if (slot < n_ptiles) load_tile(slot);  // in flight while the scalars of the previous kernel are reduced
	if (state[0] >= 0) return;
	if (iter > 0) {
		// stopping rule of pressure_solver::solve (src/pressure_solver.cpp:54-58) on the residual of iteration iter-1;
		// every workgroup evaluates it on the same partials, workgroup 0 records it
		const double rmax = reduce_partials_max(part_rmax, n_rmax, lds);
		const bool stop = rmax != rmax || rmax < tol;
		if (blockIdx.x == 0 && threadIdx.x == 0) {
			hist[iter - 1] = rmax;
			if (stop) {
				if (rmax != rmax) state[1] = 1;
				*(double *)(state + 16) = rmax;
				state[0] = iter;
			}
		}
		if (stop) return;
	}
	real beta = (real)0;
	if (!FIRST) {
		double sn, so;
		reduce_partials_sum2(part_sig_new, n_sig_new, part_sig_old, n_sig_old, lds, sn, so);
		beta = (real)(sn / so);
	}
	while (slot < n_ptiles) {
		WAVE_FENCE();
		// the new search direction (k_update_s): z + coarse part on the unknowns, + beta s
		uint32_t ac[8];
		const size_t obase = base;
		const int ol1 = l1;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			real zj = zi[zz];
			ac[zz] = ab[zz];
			if (COARSE && (ab[zz] & AB_UNKNOWN)) zj += xc;
			const real sj = FIRST ? zj : zj + beta * si[zz];
			if (s_new) s_new[obase + zz * 64 + lane] = sj;  // (nullptr: q = A z alone - the single-reduction CG of slab runs)
			h[(lx + 1) + 10 * (ly + 1) + 100 * (zz + 1)] = sj;
		}
		real fv[6];
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			real zj = fz[k];
			if (COARSE && (fab[k] & AB_UNKNOWN)) zj += xn[k];
			fv[k] = FIRST ? zj : zj + beta * fs[k];
			fv[k] = fvalid[k] ? fv[k] : (real)0;
		}
		h[0 + 10 * (lx + 1) + 100 * (ly + 1)] = fv[0];
		h[9 + 10 * (lx + 1) + 100 * (ly + 1)] = fv[1];
		h[(lx + 1) + 10 * 0 + 100 * (ly + 1)] = fv[2];
		h[(lx + 1) + 10 * 9 + 100 * (ly + 1)] = fv[3];
		h[(lx + 1) + 10 * (ly + 1) + 100 * 0] = fv[4];
		h[(lx + 1) + 10 * (ly + 1) + 100 * 9] = fv[5];
		slot += stride;
		if (slot < n_ptiles) load_tile(slot);
		WAVE_FENCE();
		double sq = 0.0;
#pragma unroll
		for (int zz = 0; zz < 8; ++zz) {
			const int i = (lx + 1) + 10 * (ly + 1) + 100 * (zz + 1);
			const uint32_t a = ac[zz];
			real out = (real)0;
			if (a & AB_UNKNOWN) {
				const real F = (a & AB_FLUID) ? (real)1 : (real)0;
				const real sc = h[i];
				real val = (real)(a & 7) * sc;
				val = madd01(-F, h[i - 1], val);
				val = madd01(-F, h[i - 10], val);
				val = madd01(-F, h[i - 100], val);
				val = madd01(-(real)((a >> 3) & 1), h[i + 1], val);
				val = madd01(-(real)((a >> 4) & 1), h[i + 10], val);
				val = madd01(-(real)((a >> 5) & 1), h[i + 100], val);
				out = scale * val;
				acc += (double)out * (double)sc;
				sq += (double)out;
			}
			q[obase + zz * 64 + lane] = out;
		}
		if (COARSE) {
			sq = wave_sum(sq);
			if (lane == 0) coarse_as[ol1] = (real)sq;
		}
	}
	block_partial_sum(acc, lds, part_qs);
}

// ---- boundary helpers: vectors in the reference's unknown order <-> tile-major fields
template <typename T, typename U>
__global__ void k_gather_unknowns(GridDims g, size_t nc, const uint32_t *cell_count, const uint32_t *tile_flag,
                                  const uint32_t *raw_scan, const T *field, U *out, int mask, int z0, int z1) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	if (z >= z0 && z < z1 && tile_flag[b >> 9] && cell_count[b] > 0) {
		T val = field[b];
		if (mask) val = (T)((int)val & mask);
		out[raw_scan[r]] = (U)val;
	}
}
template <typename T>
__global__ void k_scatter_unknowns(GridDims g, size_t nc, const uint32_t *cell_count, const uint32_t *tile_flag,
                                   const uint32_t *raw_scan, T *field, const double *in, int z0, int z1) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	if (z >= z0 && z < z1 && tile_flag[b >> 9] && cell_count[b] > 0) field[b] = (T)in[raw_scan[r]];
}
__global__ void k_zero_tiles(const int *ptiles, int n_ptiles, void *field, int elem) {
	const int slot = blockIdx.x;
	if (slot >= n_ptiles) return;
	uint32_t *p = (uint32_t *)((char *)field + (size_t)ptiles[slot] * LFA_TILE_CELLS * elem);
	for (int i = threadIdx.x; i < LFA_TILE_CELLS * elem / 4; i += 256) p[i] = 0u;
}
}  // namespace

// ================================================================================================= host side
int lfa_pcg_alloc(lfa_sim *s) {
	const size_t elem = s->prm.pcg_dtype == LFA_PCG_F64 ? 8 : 4;
	if (s->vp && s->vec_elem == elem) return LFA_OK;
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	void **vs[] = {&s->vp, &s->vr, &s->vz, &s->vs, &s->vpre, &s->vq, &s->vs2};
	for (void **v : vs) {
		if (*v) LFA_HIP(s, hipFree(*v));
		*v = nullptr;
		hipError_t e = hipMalloc(v, s->ncp * elem);
		if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of a PCG vector (%zu bytes) failed", s->ncp * elem);
		LFA_HIP(s, hipMemsetAsync(*v, 0, s->ncp * elem, s->stream));
	}
	s->vec_elem = elem;
	s->system_valid = false;
	return LFA_OK;
}

static TileCtx make_ctx(lfa_sim *s) { return TileCtx{s->ptiles, s->n_ptiles, s->tile_pslot, s->g}; }

template <typename real> static Vecs<real> make_vecs(lfa_sim *s) {
	return Vecs<real>{(real *)s->vp, (real *)s->vr, (real *)s->vz, (real *)s->vs, (real *)s->vpre, (real *)s->vq};
}

/// Tile hyperplanes for the exact MIC(0) schedule: slots sorted by tx+ty+tz.
static int build_levels(lfa_sim *s) {
	std::vector<int> tiles(s->n_ptiles);
	LFA_HIP(s, hipMemcpyAsync(tiles.data(), s->ptiles, (size_t)s->n_ptiles * 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	const int nlev = s->g.ntx + s->g.nty + s->g.ntz - 2;
	std::vector<int> cnt(nlev + 1, 0), lev(s->n_ptiles);
	for (int i = 0; i < s->n_ptiles; ++i) {
		int tx, ty, tz;
		tile_coords(s->g, tiles[i], tx, ty, tz);
		lev[i] = tx + ty + tz;
		cnt[lev[i] + 1]++;
	}
	for (int l = 0; l < nlev; ++l) cnt[l + 1] += cnt[l];
	s->level_offsets.assign(cnt.begin(), cnt.end());
	std::vector<int> order(s->n_ptiles), cur(cnt.begin(), cnt.end() - 1);
	for (int i = 0; i < s->n_ptiles; ++i) order[cur[lev[i]]++] = i;
	LFA_HIP(s, hipMemcpyAsync(s->level_tiles, order.data(), (size_t)s->n_ptiles * 4, hipMemcpyHostToDevice, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}


// ---- multilevel preconditioner: host-side set-up -----------------------------------------------------------
static bool is_ml(const lfa_sim *s) { return s->prm.precond == LFA_PRECOND_MULTILEVEL; }
static bool is_mg(const lfa_sim *s) { return s->prm.precond == LFA_PRECOND_MULTIGRID; }
/// The tiles the PCG ITERATES over: every particle tile - or, with the multigrid preconditioner on a single domain, those that are
/// not closed (lfa_sim::tile_closed: spray, solved on its own by lfa_mg_solve_closed). Valid after the preconditioner's set-up.
static TileCtx iter_ctx(lfa_sim *s) {
	if (!is_mg(s) || !s->mg) return make_ctx(s);
	const int *tiles, *slot;
	const int n = lfa_mg_level0(s, &tiles, &slot);
	return TileCtx{tiles, n, slot, s->g};
}
/// number of partial sums the consumers of sigma add up: one per workgroup (+1 for the coarse part of z.r)
static int sigma_parts(lfa_sim *s) { return pcg_grid(iter_ctx(s).n_ptiles) + (is_ml(s) ? 1 : 0); }

template <typename real> static CoarseFields<real> make_coarse(lfa_sim *s) {
	CoarseFields<real> cf{s->c_diag, {s->c_w[0], s->c_w[1], s->c_w[2]}, s->c_unk, (real *)s->c_pre,
	                      s->c_r_cur ? (real *)s->c_r_cur : (real *)s->c_r, (real *)s->c_x};
	return cf;
}

/// Dense SPD inverse by Cholesky (n2 is the number of 64^3-cell aggregates that hold fluid: tens, at most 512).
static bool spd_inverse(std::vector<double> &a, int n) {
	std::vector<double> l((size_t)n * n, 0.0);
	for (int j = 0; j < n; ++j) {
		double d = a[(size_t)j * n + j];
		for (int k = 0; k < j; ++k) d -= l[(size_t)j * n + k] * l[(size_t)j * n + k];
		if (!(d > 0.0)) return false;
		d = sqrt(d);
		l[(size_t)j * n + j] = d;
		for (int i = j + 1; i < n; ++i) {
			double v = a[(size_t)i * n + j];
			for (int k = 0; k < j; ++k) v -= l[(size_t)i * n + k] * l[(size_t)j * n + k];
			l[(size_t)i * n + j] = v / d;
		}
	}
	// inverse of L (lower), then A^-1 = L^-T L^-1
	std::vector<double> li((size_t)n * n, 0.0);
	for (int c = 0; c < n; ++c) {
		li[(size_t)c * n + c] = 1.0 / l[(size_t)c * n + c];
		for (int i = c + 1; i < n; ++i) {
			double v = 0.0;
			for (int k = c; k < i; ++k) v -= l[(size_t)i * n + k] * li[(size_t)k * n + c];
			li[(size_t)i * n + c] = v / l[(size_t)i * n + i];
		}
	}
	for (int i = 0; i < n; ++i)
		for (int j = 0; j <= i; ++j) {
			double v = 0.0;
			for (int k = i; k < n; ++k) v += li[(size_t)k * n + i] * li[(size_t)k * n + j];
			a[(size_t)i * n + j] = a[(size_t)j * n + i] = v;
		}
	return true;
}

template <typename real> static int coarse_setup(lfa_sim *s) {
	const GridDims &g = s->g;
	GridDims &g1 = s->g1;
	g1.nx = g.ntx; g1.ny = g.nty; g1.nz = g.ntz;
	g1.ntx = (g1.nx + 7) / 8; g1.nty = (g1.ny + 7) / 8; g1.ntz = (g1.nz + 7) / 8;
	g1.nt = g1.ntx * g1.nty * g1.ntz;
	const size_t ncp1 = (size_t)g1.nt * LFA_TILE_CELLS;
	if (ncp1 != s->ncp1 || s->coarse_elem != sizeof(real)) {
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		void **ptrs[] = {(void **)&s->c_diag, (void **)&s->c_w[0], (void **)&s->c_w[1], (void **)&s->c_w[2],
		                 (void **)&s->c_unk, &s->c_pre, &s->c_r, &s->c_as, &s->c_x, &s->c_r2, &s->c_x2, (void **)&s->slot_l1,
		                 (void **)&s->l1_tiles, (void **)&s->l1_l2};
		for (void **p : ptrs) {
			if (*p) LFA_HIP(s, hipFree(*p));
			*p = nullptr;
		}
		LFA_HIP(s, hipMalloc(&s->c_diag, ncp1 * 4));
		for (int d = 0; d < 3; ++d) LFA_HIP(s, hipMalloc(&s->c_w[d], ncp1 * 4));
		LFA_HIP(s, hipMalloc(&s->c_unk, ncp1));
		LFA_HIP(s, hipMalloc(&s->c_pre, ncp1 * sizeof(real)));
		LFA_HIP(s, hipMalloc(&s->c_r, 2 * ncp1 * sizeof(real)));  // [parity][ncp1], see k_pcg_b
		LFA_HIP(s, hipMalloc(&s->c_as, ncp1 * sizeof(real)));
		LFA_HIP(s, hipMalloc(&s->c_x, ncp1 * sizeof(real)));
		LFA_HIP(s, hipMalloc(&s->c_r2, (size_t)g1.nt * sizeof(real)));
		LFA_HIP(s, hipMalloc(&s->c_x2, (size_t)g1.nt * sizeof(real)));
		LFA_HIP(s, hipMalloc(&s->slot_l1, (size_t)g.nt * 4));
		LFA_HIP(s, hipMalloc(&s->l1_tiles, (size_t)g1.nt * 4));
		LFA_HIP(s, hipMalloc(&s->l1_l2, (size_t)g1.nt * 4));
		s->ncp1 = ncp1;
		s->coarse_elem = sizeof(real);
	}
	// level-1 index of every particle tile, list of level-1 blocks that hold unknowns
	std::vector<int> tiles(s->n_ptiles), sl1(s->n_ptiles), l1flag(g1.nt, 0), l1list, l1l2(g1.nt, -1);
	LFA_HIP(s, hipMemcpyAsync(tiles.data(), s->ptiles, (size_t)s->n_ptiles * 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	for (int i = 0; i < s->n_ptiles; ++i) {
		int tx, ty, tz;
		tile_coords(g, tiles[i], tx, ty, tz);
		sl1[i] = (int)blocked_index(g1, tx, ty, tz);
		l1flag[sl1[i] >> 9] = 1;
	}
	for (int t = 0; t < g1.nt; ++t)
		if (l1flag[t]) { l1l2[t] = (int)l1list.size(); l1list.push_back(t); }
	s->n_l1tiles = s->n2 = (int)l1list.size();
	LFA_HIP(s, hipMemcpyAsync(s->slot_l1, sl1.data(), (size_t)s->n_ptiles * 4, hipMemcpyHostToDevice, s->stream));
	LFA_HIP(s, hipMemcpyAsync(s->l1_tiles, l1list.data(), l1list.size() * 4, hipMemcpyHostToDevice, s->stream));
	LFA_HIP(s, hipMemcpyAsync(s->l1_l2, l1l2.data(), (size_t)g1.nt * 4, hipMemcpyHostToDevice, s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_diag, 0, ncp1 * 4, s->stream));
	for (int d = 0; d < 3; ++d) LFA_HIP(s, hipMemsetAsync(s->c_w[d], 0, ncp1 * 4, s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_unk, 0, ncp1, s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_r, 0, 2 * ncp1 * sizeof(real), s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_as, 0, ncp1 * sizeof(real), s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_x, 0, ncp1 * sizeof(real), s->stream));
	LFA_HIP(s, hipMemsetAsync(s->c_pre, 0, ncp1 * sizeof(real), s->stream));
	TileCtx tc = make_ctx(s);
	hipLaunchKernelGGL(k_coarse_coeffs, dim3(pcg_grid(s->n_ptiles)), dim3(256), 0, s->stream, tc, s->abits, s->slot_l1,
	                   s->c_diag, s->c_w[0], s->c_w[1], s->c_w[2], s->c_unk);
	LFA_LAUNCH_CHECK(s);
	CoarseFields<real> cf = make_coarse<real>(s);
	const int g1grid = (s->n_l1tiles + PCG_WAVES - 1) / PCG_WAVES;
	hipLaunchKernelGGL(k_coarse_factor<real>, dim3(g1grid), dim3(256), 0, s->stream, s->l1_tiles, s->n_l1tiles, cf,
	                   (real)s->a_scale, (real)s->prm.tau, (real)s->prm.sigma);
	LFA_LAUNCH_CHECK(s);
	// level 2: A2 = P1^T A1 P1, assembled and inverted on the host
	std::vector<float> hd(ncp1), hw[3] = {std::vector<float>(ncp1), std::vector<float>(ncp1), std::vector<float>(ncp1)};
	std::vector<uint8_t> hu(ncp1);
	LFA_HIP(s, hipMemcpyAsync(hd.data(), s->c_diag, ncp1 * 4, hipMemcpyDeviceToHost, s->stream));
	for (int d = 0; d < 3; ++d) LFA_HIP(s, hipMemcpyAsync(hw[d].data(), s->c_w[d], ncp1 * 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipMemcpyAsync(hu.data(), s->c_unk, ncp1, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	const int n2 = s->n2;
	std::vector<double> a2((size_t)n2 * n2, 0.0);
	for (int k = 0; k < n2; ++k) {
		const int t1 = l1list[k];
		int bx, by, bz;
		tile_coords(g1, t1, bx, by, bz);
		for (int l = 0; l < LFA_TILE_CELLS; ++l) {
			const size_t i = (size_t)t1 * LFA_TILE_CELLS + l;
			if (!hu[i]) continue;
			a2[(size_t)k * n2 + k] += s->a_scale * hd[i];
			const int x = bx * 8 + (l & 7), y = by * 8 + ((l >> 3) & 7), z = bz * 8 + (l >> 6);
			for (int d = 0; d < 3; ++d) {
				if (hw[d][i] == 0.0f) continue;
				const int xx = x + (d == 0), yy = y + (d == 1), zz = z + (d == 2);
				if (!in_grid(g1, xx, yy, zz)) continue;
				const uint32_t j = blocked_index(g1, xx, yy, zz);
				const int k2 = l1l2[j >> 9];
				if (k2 < 0 || !hu[j]) continue;  // coupling to a tile of another rank's slab: not in this coarse space
				a2[(size_t)k * n2 + k2] -= s->a_scale * hw[d][i];
				a2[(size_t)k2 * n2 + k] -= s->a_scale * hw[d][i];
			}
		}
	}
	std::vector<double> a2copy = a2;
	if (!spd_inverse(a2, n2)) {
		// singular top level (fluid with no free surface anywhere: the reference's matrix is singular too): regularise
		a2 = a2copy;
		double tr = 0.0;
		for (int k = 0; k < n2; ++k) tr += a2[(size_t)k * n2 + k];
		for (int k = 0; k < n2; ++k) a2[(size_t)k * n2 + k] += 1e-8 * tr / n2 + 1e-30;
		if (!spd_inverse(a2, n2)) return lfa_fail(s, LFA_E_NAN, "coarse operator is not positive definite");
	}
	if (n2 > s->a2cap) {
		if (s->a2inv) LFA_HIP(s, hipFree(s->a2inv));
		s->a2inv = nullptr;
		LFA_HIP(s, hipMalloc(&s->a2inv, (size_t)n2 * n2 * sizeof(double)));
		s->a2cap = n2;
	}
	std::vector<real> inv((size_t)n2 * n2);
	for (size_t i = 0; i < inv.size(); ++i) inv[i] = (real)a2[i];
	LFA_HIP(s, hipMemcpyAsync(s->a2inv, inv.data(), inv.size() * sizeof(real), hipMemcpyHostToDevice, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

/// Coarse half of z = M^-1 r: level-1 block MIC(0) + dense top level; writes the coarse share of z.r as partial G.
template <typename real> static int coarse_apply(lfa_sim *s, double *part_sigma, hipStream_t stream) {
	CoarseFields<real> cf = make_coarse<real>(s);
	double *extra = part_sigma + pcg_grid(s->n_ptiles);
	if (s->n_l1tiles <= 64) {
		constexpr int WAVES = sizeof(real) == 4 ? 16 : 8;
		const size_t lds = (size_t)WAVES * 3 * LFA_TILE_CELLS * (sizeof(real) + 1) + (64 + 64 + 16) * sizeof(double);
		static bool attr_set = false;
		if (!attr_set) {
			LFA_HIP(s, hipFuncSetAttribute((const void *)k_coarse_all<real, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize,
			                               (int)lds));
			attr_set = true;
		}
		hipLaunchKernelGGL((k_coarse_all<real, WAVES>), dim3(1), dim3(WAVES * 64), lds, stream, s->l1_tiles, s->n_l1tiles,
		                   cf, (real)s->a_scale, (const real *)s->a2inv, (real *)s->c_x2, extra, s->pcg_state);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	const int g1grid = (s->n_l1tiles + PCG_WAVES - 1) / PCG_WAVES;
	hipLaunchKernelGGL(k_coarse_apply<real>, dim3(g1grid), dim3(256), 0, stream, s->l1_tiles, s->n_l1tiles, cf,
	                   (real)s->a_scale, (real *)s->c_r2);
	hipLaunchKernelGGL(k_coarse_top<real>, dim3(1), dim3(256), 0, stream, s->l1_tiles, s->n_l1tiles, cf,
	                   (const real *)s->a2inv, (const real *)s->c_r2, (real *)s->c_x2, extra, s->pcg_state);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

template <typename real> static int mic_factor(lfa_sim *s) {
	TileCtx tc = make_ctx(s);
	const real scale = (real)s->a_scale, tau = (real)s->prm.tau, sigma = (real)s->prm.sigma;
	if (s->prm.precond == LFA_PRECOND_MIC0_EXACT) {
		LFA_TRY(build_levels(s));
		for (size_t l = 0; l + 1 < s->level_offsets.size(); ++l) {
			const int b = s->level_offsets[l], n = s->level_offsets[l + 1] - b;
			if (!n) continue;
			hipLaunchKernelGGL((k_mic_factor<real, true>), dim3((n + PCG_WAVES - 1) / PCG_WAVES), dim3(256), 0, s->stream, tc,
			                   s->level_tiles + b, n, s->abits, (real *)s->vpre, scale, tau, sigma);
			LFA_LAUNCH_CHECK(s);
		}
	} else {
		if (is_mg(s)) {
			LFA_TRY(lfa_mg_setup(s));  // no factorisation: the smoother needs the A bytes only
		} else {
			hipLaunchKernelGGL((k_mic_factor<real, false>), dim3(pcg_grid(s->n_ptiles)), dim3(256), 0, s->stream, tc,
			                   (const int *)nullptr, s->n_ptiles, s->abits, (real *)s->vpre, scale, tau, sigma);
			LFA_LAUNCH_CHECK(s);
		}
		if (is_ml(s)) LFA_TRY(coarse_setup<real>(s));
		if ((s->prm.pcg_fused || is_mg(s)) && s->n_ptiles) {
			if (!s->nbr_table) LFA_HIP(s, hipMalloc(&s->nbr_table, (size_t)s->g.nt * NBR_STRIDE * sizeof(int)));
			// (multigrid, single domain: the rows of the tiles the PCG iterates over - closed tiles are no neighbours of anybody)
			TileCtx tcs = iter_ctx(s);
			if (tcs.n_ptiles) {
				hipLaunchKernelGGL(k_build_nbr_table, dim3((tcs.n_ptiles + 255) / 256), dim3(256), 0, s->stream, tcs, s->p_off,
				                   is_ml(s) ? (const int *)s->slot_l1 : (const int *)nullptr, (const int *)s->l1_l2, s->nbr_table);
				LFA_LAUNCH_CHECK(s);
			}
		}
	}
	return LFA_OK;
}

/// z = M^-1 r and partial dot(z, r) into part_sigma. `r1_ready`: the restricted residual was already written (by
/// k_axpy_max), so the coarse levels are forked onto the side stream and overlap the fine sweep.
template <typename real> static int mic_apply(lfa_sim *s, double *part_sigma, bool r1_ready = false) {
	if (is_mg(s)) return lfa_mg_apply(s, part_sigma);
	TileCtx tc = make_ctx(s);
	Vecs<real> v = make_vecs<real>(s);
	const real scale = (real)s->a_scale;
	const int G = pcg_grid(s->n_ptiles);
	if (s->prm.precond == LFA_PRECOND_MIC0_EXACT) {
		const int nl = (int)s->level_offsets.size() - 1;
		for (int l = 0; l < nl; ++l) {
			const int b = s->level_offsets[l], n = s->level_offsets[l + 1] - b;
			if (!n) continue;
			hipLaunchKernelGGL((k_mic_apply<real, SWEEP_FWD, false>), dim3((n + PCG_WAVES - 1) / PCG_WAVES), dim3(256), 0, s->stream,
			                   tc, s->level_tiles + b, n, s->abits, v, scale, part_sigma, s->pcg_state, (real *)nullptr,
			                   (const int *)nullptr, CoarseFields<real>{}, CoarseArgs{});
			LFA_LAUNCH_CHECK(s);
		}
		for (int l = nl - 1; l >= 0; --l) {
			const int b = s->level_offsets[l], n = s->level_offsets[l + 1] - b;
			if (!n) continue;
			hipLaunchKernelGGL((k_mic_apply<real, SWEEP_BWD, false>), dim3((n + PCG_WAVES - 1) / PCG_WAVES), dim3(256), 0, s->stream,
			                   tc, s->level_tiles + b, n, s->abits, v, scale, part_sigma, s->pcg_state, (real *)nullptr,
			                   (const int *)nullptr, CoarseFields<real>{}, CoarseArgs{});
			LFA_LAUNCH_CHECK(s);
		}
		hipLaunchKernelGGL(k_dot_zr<real>, dim3(G), dim3(256), 0, s->stream, tc, v, part_sigma, s->pcg_state);
		LFA_LAUNCH_CHECK(s);
	} else {
		// r1_ready (inside the PCG loop): the restricted residual was written by k_axpy_max, so workgroup 0 of this very
		// launch runs the coarse levels beside the tile sweeps. Otherwise (first application, test entry points, or
		// more than 64 level-1 blocks) the coarse levels follow as their own launch.
		const bool embed = is_ml(s) && r1_ready && s->n_l1tiles <= 64;
		if (embed) {
			CoarseArgs ca{s->l1_tiles, s->n_l1tiles, s->a2inv, s->c_x2, (double *)s->pcg_hist + 6144, (unsigned *)(s->pcg_state + 4)};
			hipLaunchKernelGGL((k_mic_apply<real, SWEEP_BOTH, true>), dim3(G + PCG_COARSE_BLOCKS), dim3(256), 0, s->stream, tc,
			                   (const int *)nullptr, s->n_ptiles, s->abits, v, scale, part_sigma, s->pcg_state, (real *)nullptr,
			                   (const int *)s->slot_l1, make_coarse<real>(s), ca);
			LFA_LAUNCH_CHECK(s);
		} else {
			hipLaunchKernelGGL((k_mic_apply<real, SWEEP_BOTH, false>), dim3(G), dim3(256), 0, s->stream, tc,
			                   (const int *)nullptr, s->n_ptiles, s->abits, v, scale, part_sigma, s->pcg_state,
			                   (is_ml(s) && !r1_ready) ? (real *)s->c_r : (real *)nullptr, (const int *)s->slot_l1,
			                   CoarseFields<real>{}, CoarseArgs{});
			LFA_LAUNCH_CHECK(s);
			if (is_ml(s)) LFA_TRY(coarse_apply<real>(s, part_sigma, s->stream));
		}
	}
	return LFA_OK;
}

/// After a solve that met a NaN (or lfa_bench_kernel's repeated launches, which blow the vectors up): entries of the solver's
/// arrays that are no unknowns are assumed to be zero and no kernel rewrites them - since round 5 the smoother computes
/// x + 0 * sum there, which a non-finite neighbour turns into NaN for good. The handle would fail every later solve; instead the
/// vectors are cleared and the multigrid hierarchy is built afresh (its set-up zeroes what it allocates). Failure path only.
static int pcg_scrub(lfa_sim *s) {
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	lfa_mg_free(s);
	void *vs[] = {s->vp, s->vr, s->vz, s->vs, s->vpre, s->vq, s->vs2};
	for (void *v : vs)
		if (v) LFA_HIP(s, hipMemsetAsync(v, 0, s->ncp * s->vec_elem, s->stream));
	s->pressure_epoch = 0;
	s->warm_started = false;
	s->pcg_poisoned = false;
	return LFA_OK;
}

template <typename real> static int build_system_t(lfa_sim *s, double dt) {
	if (s->pcg_poisoned) LFA_TRY(pcg_scrub(s));
	LFA_TRY(lfa_build_rhs(s, dt));
	if (s->n_ptiles || (s->dist && is_mg(s))) LFA_TRY(mic_factor<real>(s));  // the multigrid set-up has collectives: every rank takes part
	s->system_valid = true;
	return LFA_OK;
}

extern "C" int lfa_build_system(lfa_sim *s, double dt) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_pcg_alloc(s));
	return s->prm.pcg_dtype == LFA_PCG_F64 ? build_system_t<double>(s, dt) : build_system_t<float>(s, dt);
}

template <typename real, typename... Args>
static void launch_pcg_a(bool first, bool coarse, int grid, hipStream_t st, Args... a) {
	if (first && coarse) hipLaunchKernelGGL((k_pcg_a<real, true, true>), dim3(grid), dim3(256), 0, st, a...);
	else if (first) hipLaunchKernelGGL((k_pcg_a<real, true, false>), dim3(grid), dim3(256), 0, st, a...);
	else if (coarse) hipLaunchKernelGGL((k_pcg_a<real, false, true>), dim3(grid), dim3(256), 0, st, a...);
	else hipLaunchKernelGGL((k_pcg_a<real, false, false>), dim3(grid), dim3(256), 0, st, a...);
}

/// Launch widths (workgroups) of k_pcg_a / k_pcg_b. k_pcg_a streams: it is sized to be resident at once (4 workgroups per
/// CU by registers) and every wave pipelines several tiles. k_pcg_b alternates HBM phases with latency-bound sweeps, and
/// waves that start together stay in lockstep (their phases add up instead of overlapping): twice as many workgroups as
/// fit, so that the second half starts staggered as the first retires, measured 66 vs 73 us at C4.
static void fused_grids(int G, int &GA, int &GB) {
	static int n_cu = 0;
	if (!n_cu) {
		int dev = 0;
		hipDeviceProp_t prop;
		n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
		        prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
	}
	GA = std::min(G, 4 * n_cu);
	GB = std::min(G, 8 * n_cu);
}

template <typename real> static int solve_t(lfa_sim *s, double dt, double *residual, uint64_t *iterations) {
	if (s->dist && s->prm.precond == LFA_PRECOND_MIC0_EXACT)
		return lfa_fail(s, LFA_E_UNSUPPORTED, "the exact (hyperplane) MIC(0) schedule is single-GPU only");
	LFA_TRY(build_system_t<real>(s, dt));
	if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[18], s->stream));
	// The host polls the solver state between chunks of iterations; iterations launched after convergence are no-ops but
	// still cost their launches (~40 us each). The count barely changes from step to step, so the first chunk is as long
	// as the previous solve, the following ones short. (Identical on every rank of a slab run: the scalars are all-reduced.)
	const int first_chunk = s->last_iters > 0 ? (int)std::min<uint64_t>(s->last_iters, 4096) : 8;  // (+1 / +2: no change, measured)
	const uint64_t calls_at_start = s->dist ? s->dist->calls : 0;
	s->stat_launches_iter = s->stat_transport_iter = s->stat_transport_solve = s->stat_mg_levels = s->stat_mg_first_co = 0;
	s->last_residual = 0.0;
	s->last_iters = 0;
	if (residual) *residual = 0.0;
	if (iterations) *iterations = 0;
	if (!s->n_ptiles && !s->dist) return LFA_OK;
	TileCtx tc = make_ctx(s);
	Vecs<real> v = make_vecs<real>(s);
	// the tiles the iteration runs over (multigrid, single domain: without the closed ones, which lfa_mg_solve_closed takes below)
	const int n_it = iter_ctx(s).n_ptiles;
	const int G = pcg_grid(n_it);
	const real scale = (real)s->a_scale;
	double *P = s->partials;
	const bool dist = s->dist != nullptr;
	int init_state[4] = {-1, 0, 0, 0};
	LFA_HIP(s, hipMemcpyAsync(s->pcg_state, init_state, 16, hipMemcpyHostToDevice, s->stream));
	// early out: sum b^2 < 1e-6 (src/pressure_solver.cpp:29-35). Single domain: decided on the device (k_check_rhs marks the solve
	// done, every kernel below is a no-op then, the host learns it at its first poll) - unless the previous solve ended that way: a
	// scene at rest would queue a chunk of no-op iterations step after step, so it asks first, as slab runs (whose sum is a
	// collective with a read-back anyway) always do.
	const bool host_check = dist || s->last_rhs_zero;
	if (host_check) {
		double tot = 0.0;
		if (dist) {
			if (!s->n_ptiles) LFA_HIP(s, hipMemsetAsync(P + PART_B2, 0, 8, s->stream));
			LFA_TRY(lfa_dist_allreduce(s, P + PART_B2, pcg_grid_uncapped(s->n_ptiles), 0, false));
			LFA_HIP(s, hipMemcpyAsync(&tot, s->dist_red, 8, hipMemcpyDeviceToHost, s->stream));
			LFA_HIP(s, hipStreamSynchronize(s->stream));
		} else {
			std::vector<double> hb(pcg_grid_uncapped(s->n_ptiles));
			LFA_HIP(s, hipMemcpyAsync(hb.data(), P + PART_B2, (size_t)pcg_grid_uncapped(s->n_ptiles) * 8, hipMemcpyDeviceToHost, s->stream));
			LFA_HIP(s, hipStreamSynchronize(s->stream));
			for (double x : hb) tot += x;
		}
		if (tot != tot) {
			s->pcg_poisoned = true;
			return lfa_fail(s, LFA_E_NAN, "NaN in the divergence right-hand side");
		}
		s->last_rhs_zero = tot < 1e-6;
		if (tot < 1e-6) {
			if (s->warm_started) {  // the reference returns p = 0 here (src/pressure_solver.cpp:33-35), not the guess
				hipLaunchKernelGGL((k_warm_residual<real, true>), dim3(G), dim3(256), 0, s->stream, tc, v.r, (const real *)v.q, v.p);
				LFA_LAUNCH_CHECK(s);
			}
			s->pressure_epoch = s->solve_epoch;
			return LFA_OK;
		}
	} else {
		hipLaunchKernelGGL(k_check_rhs, dim3(1), dim3(256), 0, s->stream, (const double *)(P + PART_B2), pcg_grid_uncapped(s->n_ptiles), s->pcg_state);
		LFA_LAUNCH_CHECK(s);
	}
	if (s->warm_started) {
		// r = b - A p_guess: one SpMV and one subtraction before the first preconditioner application
		hipLaunchKernelGGL(k_spmv<real>, dim3(G), dim3(256), 0, s->stream, tc, s->abits, (const real *)v.p, v.q, scale, P + PART_ZS,
		                   s->pcg_state);
		hipLaunchKernelGGL((k_warm_residual<real, false>), dim3(G), dim3(256), 0, s->stream, tc, v.r, (const real *)v.q, v.p);
		LFA_LAUNCH_CHECK(s);
	}
	s->pressure_epoch = 0;  // set to this solve's epoch once it has converged: a NaN or a capped solve is no guess for the next one
	if (is_mg(s)) LFA_TRY(lfa_mg_solve_closed(s));  // (spray: tiles whose unknowns couple to nothing outside - exact, once; a no-op after the early-out)
	if (!n_it && !dist) {  // nothing but closed tiles: no iteration
		s->pressure_epoch = s->solve_epoch;
		return LFA_OK;
	}
	const int NS = sigma_parts(s);
	// where the consumers find a reduced scalar: the per-workgroup partials (they re-add them in a fixed order), or -
	// with slabs - the all-reduced value
	double *red = s->dist_red;
	auto sig_src = [&](int parity) { return dist ? red + 3 + parity : P + (parity ? PART_SIG1 : PART_SIG0); };
	const int n_sig = dist ? 1 : NS, n_zs = dist ? 1 : G, n_max = dist ? 1 : G;
	const real *cx = is_ml(s) ? (const real *)s->c_x : (const real *)nullptr;
	// z = M^-1 r ; s = z ; sigma = z.r
	LFA_TRY(mic_apply<real>(s, P + PART_SIG0));
	if (dist) LFA_TRY(lfa_dist_allreduce(s, P + PART_SIG0, NS, 3, false));
	const int maxit = (int)s->prm.max_iterations;
	const int chunk = 4;
	int done = -1, nan = 0, i = 0;
	int *hstate = (int *)s->h_pinned;
	int aborted = 0;  // a kernel whose workgroups wait for each other gave a wait up (mg.hip: co_wait)
	// fused iteration (k_pcg_a / k_pcg_b): tile-local MIC(0) with or without the coarse levels, single domain or slabs
	// (with the multigrid preconditioner the second kernel is the AXPY/pre-smoothing kernel followed by the V-cycle, mg.hip)
	const bool fused = (is_mg(s) || s->prm.pcg_fused) && s->prm.precond != LFA_PRECOND_MIC0_EXACT && (s->n_ptiles == 0 || s->nbr_table);
	if (!fused) {
		hipLaunchKernelGGL(k_update_s<real>, dim3(G), dim3(256), 0, s->stream, tc, v, sig_src(0), sig_src(0), n_sig, 1,
		                   s->pcg_state, s->abits, cx, (const int *)s->slot_l1, (const real *)s->c_x2, (const int *)s->l1_l2);
		LFA_LAUNCH_CHECK(s);
	}
	const bool embed = fused && is_ml(s) && s->n_l1tiles <= 64;
	real *sbuf[2] = {(real *)s->vs, (real *)s->vs2};
	real *crbuf[2] = {(real *)s->c_r, is_ml(s) ? (real *)s->c_r + s->ncp1 : (real *)nullptr};
	// launch widths of the two fused kernels (workgroups); each kernel reads the other's per-workgroup partials
	int GA, GB;
	fused_grids(G, GA, GB);
	if (is_mg(s)) GB = G;  // the V-cycle kernels write pcg_grid(n_ptiles) partials
	const int NSB = GB + (is_ml(s) ? 1 : 0);
	// slabs: boundary tile layers whose rows couple to the neighbour rank (k_ghost_face_rows)
	const int n_face_lo = dist && lfa_has_lo(s) ? s->n_own_first : 0, n_face_hi = dist && lfa_has_hi(s) ? s->n_own_last : 0;
	const int g_face_lo = std::min(16, (n_face_lo + PCG_WAVES - 1) / PCG_WAVES), g_face_hi = std::min(16, (n_face_hi + PCG_WAVES - 1) / PCG_WAVES);
	// Slab runs with the multigrid preconditioner: the single-reduction form of CG (Chronopoulos / Gear; mg.hip:
	// k_mg_axpy_presmooth_cg). Step A_k: w = A z_k and delta_k = w.z_k (k_pcg_a without a search direction, the slice of z across
	// the slab faces, the rows that needed it), then gamma_k = z_k.r_k, delta_k and the signed max of r_k of every rank in ONE
	// collective (gather buffer k & 1). Step B_k: stopping rule on r_k, the recurrences, r_k+1, the V-cycle -> z_k+1.
	// Per iteration 2 D + 2 transport calls (D distributed levels) instead of 2 D + 3.
	const bool cg1 = fused && dist && is_mg(s);
	if (cg1) {
		const int nr = s->dist->nranks;
		double *alpha_io = s->dist_red + LFA_DIST_ALPHA_OFF;
		auto step_a = [&](int k, const double *rmax_part, int n_rmax_part, const double *gam_part, int n_gam_part) -> int {
			launch_pcg_a<real>(true, false, GA, s->stream, s->n_ptiles, (const int *)s->nbr_table, (const uint8_t *)s->abits, (const real *)v.z,
			                   (const real *)nullptr, (real *)nullptr, v.q, scale, (const double *)nullptr, 0, (const double *)nullptr, 0,
			                   (const double *)nullptr, 0, s->prm.tolerance, 0, s->pcg_state, s->pcg_hist, P + PART_ZS, (const real *)nullptr,
			                   (const real *)nullptr, (real *)nullptr);
			LFA_LAUNCH_CHECK(s);
			LFA_TRY(lfa_dist_exchange_slices(s, v.z, (int)sizeof(real)));
			if (g_face_lo) {
				hipLaunchKernelGGL(k_ghost_face_rows<real>, dim3(g_face_lo), dim3(256), 0, s->stream, tc, n_face_lo, 0, 0, s->abits,
				                   (const real *)v.z, v.q, scale, P + PART_ZS + GA, (real *)nullptr, (const int *)s->slot_l1, s->pcg_state);
				LFA_LAUNCH_CHECK(s);
			}
			if (g_face_hi) {
				hipLaunchKernelGGL(k_ghost_face_rows<real>, dim3(g_face_hi), dim3(256), 0, s->stream, tc, 0, s->n_ptiles - n_face_hi,
				                   n_face_hi, s->abits, (const real *)v.z, v.q, scale, P + PART_ZS + GA + g_face_lo, (real *)nullptr,
				                   (const int *)s->slot_l1, s->pcg_state);
				LFA_LAUNCH_CHECK(s);
			}
			return lfa_dist_gather_triple(s, rmax_part, n_rmax_part, gam_part, n_gam_part, P + PART_ZS, GA + g_face_lo + g_face_hi, k & 1);
		};
		const uint64_t calls_a0 = s->dist->calls;
		LFA_TRY(step_a(0, P + PART_RMAX, 0, P + PART_SIG0, NS));
		const uint64_t calls_a = s->dist->calls - calls_a0;
		while (!aborted && i < maxit && done < 0) {
			const int end = std::min(maxit, i + (i == 0 ? first_chunk : chunk));
			for (; i < end; ++i) {
				const int po = i & 1, pn = po ^ 1;
				const uint64_t calls0 = s->dist->calls;
				const double *cur = lfa_dist_gather_buf(s, po), *old = lfa_dist_gather_buf(s, pn);
				double *sig_new_part = P + (pn ? PART_SIG1 : PART_SIG0);
				LFA_TRY(lfa_mg_axpy_apply_cg(s, cur + nr, nr, old + nr, nr, cur + 2 * nr, nr, cur, nr, i, alpha_io, P + PART_RMAX, sig_new_part));
				LFA_TRY(step_a(i + 1, P + PART_RMAX, GB, sig_new_part, NSB));
				if (i == 0) {
					s->stat_transport_iter = s->dist->calls - calls0;
					uint64_t mgl = 0;
					lfa_mg_stats(s, &mgl, &s->stat_mg_levels, &s->stat_mg_first_co);
					s->stat_launches_iter = 1 + (g_face_lo > 0) + (g_face_hi > 0) + mgl;
				}
			}
			hipLaunchKernelGGL(k_check_converged, dim3(1), dim3(256), 0, s->stream, (const double *)lfa_dist_gather_buf(s, i & 1), nr,
			                   s->prm.tolerance, i - 1, s->pcg_state, s->pcg_hist, (const double *)lfa_dist_gather_buf(s, i & 1) + 3 * nr);
			LFA_LAUNCH_CHECK(s);
			LFA_HIP(s, hipMemcpyAsync(hstate, s->pcg_state, 80, hipMemcpyDeviceToHost, s->stream));
			LFA_HIP(s, hipStreamSynchronize(s->stream));
			done = hstate[0];
			nan = hstate[1];
			aborted = hstate[2];
			if (aborted) break;
		}
		(void)calls_a;
	}
	while (fused && !cg1 && !aborted && i < maxit && done < 0) {
		const int end = std::min(maxit, i + (i == 0 ? first_chunk : chunk));
		for (; i < end; ++i) {
			const int po = i & 1, pn = po ^ 1;
			const uint64_t calls0 = dist ? s->dist->calls : 0;
			// where a kernel finds the scalars of the previous one: per-workgroup partials (the application before the loop
			// wrote NS sigma partials, k_pcg_b writes NSB), or - with slabs - the all-reduced values
			// slabs: sigma_k (k >= 1) and the signed max of the residual it belongs to arrive together, one value per rank
			// (lfa_dist_gather_pair, parity k & 1); sigma_0 of the application before the loop is the all-reduced scalar
			const int nr = dist ? s->dist->nranks : 0;
			const double *sig_po = !dist ? P + (po ? PART_SIG1 : PART_SIG0) : (i == 0 ? red + 3 : lfa_dist_gather_buf(s, po) + nr);
			const double *sig_pn = !dist ? P + (pn ? PART_SIG1 : PART_SIG0) : (i <= 1 ? red + 3 : lfa_dist_gather_buf(s, pn) + nr);
			const int n_sig_po = dist ? (i == 0 ? 1 : nr) : (i == 0 ? NS : NSB), n_sig_pn = dist ? (i <= 1 ? 1 : nr) : (i <= 1 ? NS : NSB);
			const double *rmax_prev = !dist ? P + PART_RMAX : (i == 0 ? red + 2 : lfa_dist_gather_buf(s, po));
			const int n_rmax_prev = dist ? (i == 0 ? 1 : nr) : GB;
			launch_pcg_a<real>(i == 0, is_ml(s), GA, s->stream, n_it, (const int *)s->nbr_table, (const uint8_t *)s->abits,
			                   (const real *)v.z, (const real *)sbuf[po], sbuf[pn], v.q, scale, sig_po, n_sig_po, sig_pn, n_sig_pn,
			                   rmax_prev, n_rmax_prev, s->prm.tolerance, i, s->pcg_state,
			                   s->pcg_hist, P + PART_ZS, cx, (const real *)s->c_x2, is_ml(s) ? (real *)s->c_as : (real *)nullptr);
			LFA_LAUNCH_CHECK(s);
			if (dist) {
				// the new search direction across the slab faces, then the rows of q that needed it
				LFA_TRY(lfa_dist_exchange_slices(s, sbuf[pn], (int)sizeof(real)));
				real *cas = is_ml(s) ? (real *)s->c_as : (real *)nullptr;
				if (g_face_lo) {
					hipLaunchKernelGGL(k_ghost_face_rows<real>, dim3(g_face_lo), dim3(256), 0, s->stream, tc, n_face_lo, 0, 0, s->abits,
					                   (const real *)sbuf[pn], v.q, scale, P + PART_ZS + GA, cas, (const int *)s->slot_l1, s->pcg_state);
					LFA_LAUNCH_CHECK(s);
				}
				if (g_face_hi) {
					hipLaunchKernelGGL(k_ghost_face_rows<real>, dim3(g_face_hi), dim3(256), 0, s->stream, tc, 0, s->n_ptiles - n_face_hi,
					                   n_face_hi, s->abits, (const real *)sbuf[pn], v.q, scale, P + PART_ZS + GA + g_face_lo, cas,
					                   (const int *)s->slot_l1, s->pcg_state);
					LFA_LAUNCH_CHECK(s);
				}
				LFA_TRY(lfa_dist_allreduce(s, P + PART_ZS, GA + g_face_lo + g_face_hi, 1, false));
			}
			const double *zs_src = dist ? red + 1 : P + PART_ZS;
			const int n_zs_src = dist ? 1 : GA;
			double *sig_new_part = P + (pn ? PART_SIG1 : PART_SIG0);
			CoarseFields<real> cf = is_ml(s) ? make_coarse<real>(s) : CoarseFields<real>{};
			if (is_mg(s)) {
				// (single domain: the V-cycle's first kernel tests the residual this AXPY produces - the reference does not
				// precondition a converged residual either)
				s->cur_iter = dist ? -1 : i;
				s->cur_rmax_parts = P + PART_RMAX;
				s->cur_rmax_n = GB;
				const int rc_mg = lfa_mg_axpy_apply(s, sbuf[pn], sig_po, n_sig_po, zs_src, n_zs_src, P + PART_RMAX, sig_new_part);
				s->cur_iter = -1;
				LFA_TRY(rc_mg);
			} else if (embed) {
				cf.r = crbuf[po];
				cf.as = (const real *)s->c_as;
				CoarseArgs ca{s->l1_tiles, s->n_l1tiles, s->a2inv, s->c_x2, (double *)s->pcg_hist + 6144, (unsigned *)(s->pcg_state + 4)};
				hipLaunchKernelGGL((k_pcg_b<real, true>), dim3(GB + PCG_COARSE_BLOCKS), dim3(256), 0, s->stream,
				                   (const int *)s->ptiles, s->n_ptiles, s->abits, v, (const real *)sbuf[pn], scale, sig_po, n_sig_po,
				                   zs_src, n_zs_src, P + PART_RMAX, sig_new_part, s->pcg_state, crbuf[pn], (const int *)s->slot_l1, cf, ca);
				LFA_LAUNCH_CHECK(s);
			} else {
				hipLaunchKernelGGL((k_pcg_b<real, false>), dim3(GB), dim3(256), 0, s->stream, (const int *)s->ptiles,
				                   s->n_ptiles, s->abits, v, (const real *)sbuf[pn], scale, sig_po, n_sig_po, zs_src, n_zs_src,
				                   P + PART_RMAX, sig_new_part, s->pcg_state, crbuf[pn], (const int *)s->slot_l1,
				                   CoarseFields<real>{}, CoarseArgs{});
				LFA_LAUNCH_CHECK(s);
				if (is_ml(s)) {  // more than 64 level-1 blocks: the coarse levels follow as their own launches
					s->c_r_cur = crbuf[pn];
					LFA_TRY(coarse_apply<real>(s, sig_new_part + GB - G, s->stream));
					s->c_r_cur = nullptr;
				}
			}
			if (dist) LFA_TRY(lfa_dist_gather_pair(s, P + PART_RMAX, GB, sig_new_part, NSB, pn));  // one collective for both
			if (i == 0) {  // lfa_get_solver_stats: every iteration issues the same launches and transport calls
				s->stat_transport_iter = dist ? s->dist->calls - calls0 : 0;
				uint64_t mgl = 0;
				if (is_mg(s)) lfa_mg_stats(s, &mgl, &s->stat_mg_levels, &s->stat_mg_first_co);
				s->stat_launches_iter = 1 + (dist ? (g_face_lo > 0) + (g_face_hi > 0) : 0) + (is_mg(s) ? mgl : 1);
			}
		}
		// the residual of the last iteration of the chunk is tested here (k_pcg_a tests the one before it)
		hipLaunchKernelGGL(k_check_converged, dim3(1), dim3(256), 0, s->stream,
		                   dist ? (const double *)lfa_dist_gather_buf(s, i & 1) : P + PART_RMAX, dist ? s->dist->nranks : GB,
		                   s->prm.tolerance, i - 1, s->pcg_state, s->pcg_hist,
		                   dist ? (const double *)lfa_dist_gather_buf(s, i & 1) + 3 * s->dist->nranks : (const double *)nullptr);
		LFA_LAUNCH_CHECK(s);
		LFA_HIP(s, hipMemcpyAsync(hstate, s->pcg_state, 80, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		done = hstate[0];
		nan = hstate[1];
		aborted = hstate[2];
		if (aborted) break;
	}
	while (!fused && !aborted && i < maxit && done < 0) {
		const int end = std::min(maxit, i + (i == 0 ? first_chunk : chunk));
		for (; i < end; ++i) {
			const int po = i & 1, pn = po ^ 1;
			double *sig_new_part = P + (pn ? PART_SIG1 : PART_SIG0);
			if (dist) LFA_TRY(lfa_dist_exchange_slices(s, v.s, (int)sizeof(real)));  // search vector across the slab faces
			hipLaunchKernelGGL(k_spmv<real>, dim3(G), dim3(256), 0, s->stream, tc, s->abits, (const real *)v.s, v.z, scale,
			                   P + PART_ZS, s->pcg_state);
			LFA_LAUNCH_CHECK(s);
			if (dist) LFA_TRY(lfa_dist_allreduce(s, P + PART_ZS, G, 1, false));
			hipLaunchKernelGGL(k_axpy_max<real>, dim3(G), dim3(256), 0, s->stream, tc, s->abits, v, sig_src(po), n_sig,
			                   dist ? red + 1 : P + PART_ZS, n_zs, P + PART_RMAX, s->pcg_state,
			                   is_ml(s) ? (real *)s->c_r : (real *)nullptr, (const int *)s->slot_l1);
			LFA_LAUNCH_CHECK(s);
			if (dist) LFA_TRY(lfa_dist_allreduce(s, P + PART_RMAX, G, 2, true));
			hipLaunchKernelGGL(k_check_converged, dim3(1), dim3(256), 0, s->stream, dist ? red + 2 : P + PART_RMAX, n_max,
			                   s->prm.tolerance, i, s->pcg_state, s->pcg_hist, (const double *)nullptr);
			LFA_LAUNCH_CHECK(s);
			LFA_TRY(mic_apply<real>(s, sig_new_part, true));
			if (dist) LFA_TRY(lfa_dist_allreduce(s, sig_new_part, NS, 3 + pn, false));
			hipLaunchKernelGGL(k_update_s<real>, dim3(G), dim3(256), 0, s->stream, tc, v, sig_src(pn), sig_src(po), n_sig, 0,
			                   s->pcg_state, s->abits, cx, (const int *)s->slot_l1, (const real *)s->c_x2, (const int *)s->l1_l2);
			LFA_LAUNCH_CHECK(s);
		}
		LFA_HIP(s, hipMemcpyAsync(hstate, s->pcg_state, 80, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		done = hstate[0];
		nan = hstate[1];
	}
	if (aborted) {
		// What the wait was for never came (a workgroup that is not resident: another process's kernel of the same kind on this GPU,
		// a fault). Whatever the iterations after it computed is void. This handle stops using the kernels that wait and the solve
		// is repeated from the right-hand side on the launch-per-phase path - once: that path waits for nothing.
		if (s->co_disabled) return lfa_fail(s, LFA_E_HIP, "a device-side wait of the pressure solve was given up twice");
		s->stat_co_reason = (uint64_t)(unsigned)hstate[18];  // (0x100: a wait ran out | the waiter's XCC id; 0 on a rank that retreats with a peer)
		s->co_disabled = true;
		++s->stat_co_aborts;
		s->system_valid = false;
		s->warm_started = false;
		s->pressure_epoch = 0;  // (the pressure the aborted iterations left is no guess)
		return solve_t<real>(s, dt, residual, iterations);
	}
	if (!host_check) {  // what k_check_rhs found
		if (hstate[3] == 2) {
			s->pcg_poisoned = true;
			return lfa_fail(s, LFA_E_NAN, "NaN in the divergence right-hand side");
		}
		s->last_rhs_zero = hstate[3] == 1;
		if (hstate[3] == 1) {
			if (s->warm_started) {  // the reference returns p = 0 here (src/pressure_solver.cpp:33-35), not the guess
				hipLaunchKernelGGL((k_warm_residual<real, true>), dim3(G), dim3(256), 0, s->stream, tc, v.r, (const real *)v.q, v.p);
				LFA_LAUNCH_CHECK(s);
			}
			s->pressure_epoch = s->solve_epoch;
			return LFA_OK;
		}
	}
	const int iters = done >= 0 ? done : maxit;
	double res = 0.0;
	if (done > 0) {
		memcpy(&res, hstate + 16, 8);  // (left beside the state words by the kernel that ended the solve)
	} else if (iters > 0) {
		LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 4, s->pcg_hist + (iters - 1), 8, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		memcpy(&res, s->h_pinned + 4, 8);
	}
	s->last_residual = res;
	s->last_iters = (uint64_t)iters;
	s->stat_transport_solve = s->dist ? s->dist->calls - calls_at_start : 0;
	if (residual) *residual = res;
	if (iterations) *iterations = (uint64_t)iters;
	if (nan) {
		s->pcg_poisoned = true;
		return lfa_fail(s, LFA_E_NAN, "NaN in the PCG residual at iteration %d", iters);
	}
	if (done >= 0) s->pressure_epoch = s->solve_epoch;
	return done >= 0 ? LFA_OK : LFA_W_PCG_NOT_CONVERGED;
}

extern "C" int lfa_get_solver_stats(lfa_sim *s, uint64_t stats[LFA_NUM_SOLVER_STATS]) {
	if (!s || !stats) return LFA_E_INVALID;
	const uint64_t v[LFA_NUM_SOLVER_STATS] = {s->stat_launches_iter, s->stat_transport_iter, s->stat_mg_levels, s->stat_mg_first_co,
	                                          s->last_iters, s->stat_transport_solve, 0, s->stat_co_aborts};
	for (int i = 0; i < LFA_NUM_SOLVER_STATS; ++i) stats[i] = v[i];
	return LFA_OK;
}

extern "C" int lfa_pcg_solve(lfa_sim *s, double dt, double *residual, uint64_t *iterations) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_pcg_alloc(s));
	return s->prm.pcg_dtype == LFA_PCG_F64 ? solve_t<double>(s, dt, residual, iterations)
	                                       : solve_t<float>(s, dt, residual, iterations);
}

// ---- vectors at the boundary --------------------------------------------------------------------------------
template <typename T, typename U> static int gather(lfa_sim *s, const T *field, U *host_out, uint64_t n, int mask) {
	LFA_TRY(lfa_number_unknowns(s));
	if (n != s->n_unknowns) return lfa_fail(s, LFA_E_INVALID, "expected %llu unknowns", (unsigned long long)s->n_unknowns);
	if (!n) return LFA_OK;
	LFA_TRY(lfa_ensure_io(s, n * sizeof(U)));
	hipLaunchKernelGGL((k_gather_unknowns<T, U>), dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream, s->g,
	                   s->nc, s->cell_count, s->tile_flag, s->raw_scan, field, (U *)s->io_buf, mask, s->slab_lo * 8, s->slab_hi * 8);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(host_out, s->io_buf, n * sizeof(U), hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}
template <typename T> static int scatter(lfa_sim *s, T *field, const double *host_in, uint64_t n) {
	LFA_TRY(lfa_number_unknowns(s));
	if (n != s->n_unknowns) return lfa_fail(s, LFA_E_INVALID, "expected %llu unknowns", (unsigned long long)s->n_unknowns);
	if (!n) return LFA_OK;
	LFA_TRY(lfa_ensure_io(s, n * 8));
	LFA_HIP(s, hipMemcpyAsync(s->io_buf, host_in, n * 8, hipMemcpyHostToDevice, s->stream));
	hipLaunchKernelGGL(k_zero_tiles, dim3(s->n_ptiles), dim3(256), 0, s->stream, s->ptiles, s->n_ptiles, (void *)field,
	                   (int)sizeof(T));
	hipLaunchKernelGGL(k_scatter_unknowns<T>, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream, s->g, s->nc,
	                   s->cell_count, s->tile_flag, s->raw_scan, field, (const double *)s->io_buf, s->slab_lo * 8,
	                   s->slab_hi * 8);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}
#define NEED_SYSTEM(s)                                                                               \
	if (!(s)) return LFA_E_INVALID;                                                                   \
	if (!(s)->system_valid) return lfa_fail((s), LFA_E_INVALID, "call lfa_build_system (or lfa_pcg_solve) first"); \
	LFA_HIP((s), hipSetDevice((s)->device));
#define F64(s) ((s)->prm.pcg_dtype == LFA_PCG_F64)

extern "C" int lfa_download_abits(lfa_sim *s, uint8_t *bits, uint64_t n) {
	NEED_SYSTEM(s);
	return gather<uint8_t, uint8_t>(s, s->abits, bits, n, 0x3F);
}
extern "C" int lfa_download_rhs(lfa_sim *s, double *b, uint64_t n) {
	NEED_SYSTEM(s);
	return F64(s) ? gather<double, double>(s, (double *)s->vr, b, n, 0) : gather<float, double>(s, (float *)s->vr, b, n, 0);
}
extern "C" int lfa_download_precon(lfa_sim *s, double *p, uint64_t n) {
	NEED_SYSTEM(s);
	return F64(s) ? gather<double, double>(s, (double *)s->vpre, p, n, 0)
	              : gather<float, double>(s, (float *)s->vpre, p, n, 0);
}
extern "C" int lfa_download_pressure(lfa_sim *s, double *p, uint64_t n) {
	NEED_SYSTEM(s);
	return F64(s) ? gather<double, double>(s, (double *)s->vp, p, n, 0) : gather<float, double>(s, (float *)s->vp, p, n, 0);
}
extern "C" int lfa_upload_pressure(lfa_sim *s, const double *p, uint64_t n) {
	NEED_SYSTEM(s);
	s->pressure_epoch = 0;  // a pressure the host put there is not the previous solve's: the next solve starts from p = 0
	return F64(s) ? scatter<double>(s, (double *)s->vp, p, n) : scatter<float>(s, (float *)s->vp, p, n);
}

template <typename real> static int apply_precon_t(lfa_sim *s, const double *r, double *z, uint64_t n) {
	LFA_TRY(scatter<real>(s, (real *)s->vr, r, n));
	int init_state[4] = {-1, 0, 0, 0};
	LFA_HIP(s, hipMemcpyAsync(s->pcg_state, init_state, 16, hipMemcpyHostToDevice, s->stream));
	if (s->n_ptiles) {
		LFA_TRY(mic_apply<real>(s, s->partials + PART_SIG0));
		// (the closed tiles are no part of the V-cycle: their share of M^-1 is their exact inverse)
		if (is_mg(s)) LFA_TRY(lfa_mg_solve_closed(s, s->vz));
		if (is_ml(s)) {
			hipLaunchKernelGGL(k_add_coarse<real>, dim3(pcg_grid(s->n_ptiles)), dim3(256), 0, s->stream, make_ctx(s), s->abits,
			                   (real *)s->vz, (const real *)s->c_x, (const int *)s->slot_l1, (const real *)s->c_x2,
			                   (const int *)s->l1_l2);
			LFA_LAUNCH_CHECK(s);
		}
	}
	return gather<real, double>(s, (real *)s->vz, z, n, 0);
}
extern "C" int lfa_apply_preconditioner(lfa_sim *s, const double *r, double *z, uint64_t n) {
	NEED_SYSTEM(s);
	if (!r || !z) return LFA_E_INVALID;
	return F64(s) ? apply_precon_t<double>(s, r, z, n) : apply_precon_t<float>(s, r, z, n);
}
template <typename real> static int apply_a_t(lfa_sim *s, const double *vin, double *out, uint64_t n) {
	LFA_TRY(scatter<real>(s, (real *)s->vs, vin, n));
	int init_state[4] = {-1, 0, 0, 0};
	LFA_HIP(s, hipMemcpyAsync(s->pcg_state, init_state, 16, hipMemcpyHostToDevice, s->stream));
	if (s->dist) LFA_TRY(lfa_dist_exchange_slices(s, s->vs, (int)sizeof(real)));
	if (s->n_ptiles) {
		TileCtx tc = make_ctx(s);
		hipLaunchKernelGGL(k_spmv<real>, dim3(pcg_grid(s->n_ptiles)), dim3(256), 0, s->stream, tc, s->abits,
		                   (const real *)s->vs, (real *)s->vz, (real)s->a_scale, s->partials + PART_ZS, s->pcg_state);
		LFA_LAUNCH_CHECK(s);
	}
	return gather<real, double>(s, (real *)s->vz, out, n, 0);
}
extern "C" int lfa_apply_a(lfa_sim *s, const double *v, double *out, uint64_t n) {
	NEED_SYSTEM(s);
	if (!v || !out) return LFA_E_INVALID;
	return F64(s) ? apply_a_t<double>(s, v, out, n) : apply_a_t<float>(s, v, out, n);
}

// ================================================================================================= whole hot path
static int ev_rec(lfa_sim *s, int i) {
	if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[i], s->stream));
	return LFA_OK;
}

extern "C" int lfa_step_hot(lfa_sim *s, double dt, double *residual, uint64_t *iterations) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(ev_rec(s, 0));
	LFA_TRY(lfa_hash_particles(s));
	LFA_TRY(ev_rec(s, 1));
	LFA_TRY(lfa_p2g_run(s, true, dt));  // gravity fused into the normalise pass
	LFA_TRY(ev_rec(s, 2));
	LFA_TRY(ev_rec(s, 3));
	LFA_TRY(lfa_pcg_alloc(s));
	LFA_TRY(ev_rec(s, 4));
	double res = 0.0;
	uint64_t it = 0;
	int rc = lfa_pcg_solve(s, dt, &res, &it);
	if (rc < 0) return rc;
	LFA_TRY(ev_rec(s, 5));
	LFA_TRY(lfa_apply_pressure(s, dt));
	LFA_TRY(ev_rec(s, 6));
	LFA_TRY(lfa_extrapolate(s));
	LFA_TRY(ev_rec(s, 7));
	LFA_TRY(lfa_g2p(s));
	LFA_TRY(ev_rec(s, 8));
	if (residual) *residual = res;
	if (iterations) *iterations = it;
	if (s->timing) {
		LFA_HIP(s, hipEventSynchronize(s->ev[8]));
		float ms = 0.f;
		const int pairs[8][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}};
		for (int k = 0; k < 8; ++k) {
			LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[pairs[k][0]], s->ev[pairs[k][1]]));
			s->ms[k] = ms;
		}
		LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[16], s->ev[17]));
		s->ms[8] = ms;
		if (it || s->n_ptiles) {  // split [3]/[4] at the end of the system build recorded inside the solve
			LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[4], s->ev[18]));
			s->ms[3] = ms;
			LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[18], s->ev[5]));
			s->ms[4] = ms;
		}
		s->ms[9] = it ? s->ms[4] / (double)it : 0.0;
	}
	return rc;
}

// ================================================================================================= measurement
int lfa_p2g_bench(lfa_sim *s, int which);   // p2g.hip
int lfa_g2p_bench(lfa_sim *s);              // grid_ops.hip

template <typename real> static int bench_launch(lfa_sim *s, int which) {
	TileCtx tc = make_ctx(s);
	Vecs<real> v = make_vecs<real>(s);
	const int n_it = iter_ctx(s).n_ptiles;  // (the fused kernels' rows: lfa_sim::nbr_table)
	const int G = pcg_grid(which == LFA_K_PCG_A || which == LFA_K_PCG_B ? n_it : s->n_ptiles);
	const real scale = (real)s->a_scale;
	double *P = s->partials;
	int GA, GB;
	fused_grids(G, GA, GB);
	switch (which) {
	case LFA_K_SPMV_DOT:
		hipLaunchKernelGGL(k_spmv<real>, dim3(G), dim3(256), 0, s->stream, tc, s->abits, (const real *)v.s, v.z, scale,
		                   P + PART_ZS, s->pcg_state);
		break;
	case LFA_K_AXPY_MAX:
		hipLaunchKernelGGL(k_axpy_max<real>, dim3(G), dim3(256), 0, s->stream, tc, s->abits, v, P + PART_SIG0, G, P + PART_ZS,
		                   G, P + PART_RMAX, s->pcg_state, is_ml(s) ? (real *)s->c_r : (real *)nullptr,
		                   (const int *)s->slot_l1);
		break;
	case LFA_K_MIC_APPLY:
		return mic_apply<real>(s, P + PART_SIG1, true);
	case LFA_K_MIC_FINE:
		hipLaunchKernelGGL((k_mic_apply<real, SWEEP_BOTH, false>), dim3(G), dim3(256), 0, s->stream, tc,
		                   (const int *)nullptr, s->n_ptiles, s->abits, v, scale, P + PART_SIG1, s->pcg_state, (real *)nullptr,
		                   (const int *)nullptr, CoarseFields<real>{}, CoarseArgs{});
		break;
	case LFA_K_COARSE:
		if (!is_ml(s)) return lfa_fail(s, LFA_E_INVALID, "no coarse levels with this preconditioner");
		return coarse_apply<real>(s, P + PART_SIG1, s->stream);
	case LFA_K_PCG_A:
		if (!s->nbr_table || !(s->prm.pcg_fused || is_mg(s))) return lfa_fail(s, LFA_E_INVALID, "fused kernels: solve with pcg_fused = 1 first");
		launch_pcg_a<real>(false, is_ml(s), is_mg(s) ? G : GA, s->stream, n_it, (const int *)s->nbr_table, (const uint8_t *)s->abits,
		                   (const real *)v.z, (const real *)v.s, (real *)s->vs2, v.q, scale, (const double *)(P + PART_SIG0), G,
		                   (const double *)(P + PART_SIG0), G, (const double *)(P + PART_RMAX), G, -HUGE_VAL, 1, s->pcg_state,
		                   s->pcg_hist + 4095, P + PART_ZS, is_ml(s) ? (const real *)s->c_x : (const real *)nullptr,
		                   (const real *)s->c_x2, is_ml(s) ? (real *)s->c_as : (real *)nullptr);
		break;
	case LFA_K_PCG_B: {
		if (!s->nbr_table || !s->prm.pcg_fused) return lfa_fail(s, LFA_E_INVALID, "fused kernels: solve with pcg_fused = 1 first");
		const bool embed = is_ml(s) && s->n_l1tiles <= 64;
		real *cr1 = is_ml(s) ? (real *)s->c_r + s->ncp1 : (real *)nullptr;
		if (embed) {
			CoarseFields<real> cf = make_coarse<real>(s);
			cf.as = (const real *)s->c_as;
			CoarseArgs ca{s->l1_tiles, s->n_l1tiles, s->a2inv, s->c_x2, (double *)s->pcg_hist + 6144, (unsigned *)(s->pcg_state + 4)};
			hipLaunchKernelGGL((k_pcg_b<real, true>), dim3(GB + PCG_COARSE_BLOCKS), dim3(256), 0, s->stream,
			                   (const int *)s->ptiles, s->n_ptiles, s->abits, v, (const real *)s->vs2, scale, P + PART_SIG0, G, P + PART_ZS, G, P + PART_RMAX, P + PART_SIG1,
			                   s->pcg_state, cr1, (const int *)s->slot_l1, cf, ca);
		} else {
			hipLaunchKernelGGL((k_pcg_b<real, false>), dim3(GB), dim3(256), 0, s->stream, (const int *)s->ptiles, s->n_ptiles,
			                   s->abits, v, (const real *)s->vs2, scale, P + PART_SIG0, G, P + PART_ZS, G, P + PART_RMAX, P + PART_SIG1, s->pcg_state, cr1,
			                   (const int *)s->slot_l1, CoarseFields<real>{}, CoarseArgs{});
		}
		break;
	}
	case LFA_K_MG_AXPY_PRESMOOTH: return lfa_mg_bench_part(s, 0);
	case LFA_K_MG_DOWN0: return lfa_mg_bench_part(s, 1);
	case LFA_K_MG_COARSE: return lfa_mg_bench_part(s, 2);
	case LFA_K_MG_UP0: return lfa_mg_bench_part(s, 3);
	case LFA_K_UPDATE_S:
		hipLaunchKernelGGL(k_update_s<real>, dim3(G), dim3(256), 0, s->stream, tc, v, P + PART_SIG0, P + PART_SIG0, G, 0,
		                   s->pcg_state, s->abits, is_ml(s) ? (const real *)s->c_x : (const real *)nullptr,
		                   (const int *)s->slot_l1, (const real *)s->c_x2, (const int *)s->l1_l2);
		break;
	}
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

extern "C" int lfa_bench_kernel(lfa_sim *s, int which, int reps, double *mean_ms) {
	if (!s || !mean_ms || reps < 1) return LFA_E_INVALID;
	if (!s->binned || !s->n_ptiles) return lfa_fail(s, LFA_E_INVALID, "lfa_bench_kernel: run lfa_step_hot first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_enable_timing(s, s->timing ? 1 : 0));
	if (!s->ev_created) {
		for (auto &e : s->ev) LFA_HIP(s, hipEventCreate(&e));
		s->ev_created = true;
	}
	const bool is_pcg = which <= LFA_K_UPDATE_S || which == LFA_K_MIC_FINE || which == LFA_K_COARSE ||
	                    which == LFA_K_PCG_A || which == LFA_K_PCG_B || (which >= LFA_K_MG_AXPY_PRESMOOTH && which <= LFA_K_MG_UP0);
	if (is_pcg) {
		if (!s->system_valid) return lfa_fail(s, LFA_E_INVALID, "lfa_bench_kernel: no pressure system on the device");
		int init_state[4] = {-1, 0, 0, 0};  // "still iterating": the kernels early-out once a solve has converged
		LFA_HIP(s, hipMemcpyAsync(s->pcg_state, init_state, 16, hipMemcpyHostToDevice, s->stream));
	}
	auto once = [&]() -> int {
		if (is_pcg) return F64(s) ? bench_launch<double>(s, which) : bench_launch<float>(s, which);
		if (which == LFA_K_G2P) return lfa_g2p_bench(s);
		return lfa_p2g_bench(s, which);
	};
	LFA_TRY(once());  // warm
	float ms = 0.f;
	if (which == LFA_K_BIN) {
		// the binning as the step loop runs it: the deferred half (v, C) is consumed by the P2G / G2P there, so it is completed
		// between the timed calls, outside the timed intervals
		for (int i = 0; i < reps; ++i) {
			LFA_TRY(lfa_particles_materialize(s));
			LFA_HIP(s, hipEventRecord(s->ev[20], s->stream));
			LFA_TRY(once());
			LFA_HIP(s, hipEventRecord(s->ev[21], s->stream));
			LFA_HIP(s, hipEventSynchronize(s->ev[21]));
			float one = 0.f;
			LFA_HIP(s, hipEventElapsedTime(&one, s->ev[20], s->ev[21]));
			ms += one;
		}
	} else {
		LFA_HIP(s, hipEventRecord(s->ev[20], s->stream));
		for (int i = 0; i < reps; ++i) LFA_TRY(once());
		LFA_HIP(s, hipEventRecord(s->ev[21], s->stream));
		LFA_HIP(s, hipEventSynchronize(s->ev[21]));
		LFA_HIP(s, hipEventElapsedTime(&ms, s->ev[20], s->ev[21]));
	}
	*mean_ms = (double)ms / reps;
	s->system_valid = is_pcg ? s->system_valid : false;
	s->pcg_poisoned = s->pcg_poisoned || is_pcg;  // (repeated AXPYs with the same scalars: the vectors may have overflowed)
	return LFA_OK;
}
