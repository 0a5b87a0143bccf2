// libfluid_amd/csrc/grid_ops.hip -- MAC-grid stencil stages around the pressure solve
// and the grid-to-particle transfer (a19, a20).
//
// All kernels walk the tile lists built by lfa_hash_particles: `ptiles` (tiles holding particles = tiles holding PCG
// unknowns) and `dtiles` (their 27-neighbourhoods). A cell outside the dilated set is never read as data: its type is
// solid/air from the persistent solid mask and it is never an unknown (stage_halo_types / is_unknown_at below).
#include "common.h"
#include "pcg.h"

namespace {

struct GridView {
	GridDims g;
	const uint8_t *ctype, *solid;
	const uint32_t *cell_count, *tile_flag;
};
/// Membership in the reference's _fluid_cells list (cells holding particles, src/simulation.cpp:83-94).
__device__ inline bool is_unknown_at(const GridView &gv, int x, int y, int z) {
	if (!in_grid(gv.g, x, y, z)) return false;
	uint32_t b = blocked_index(gv.g, x, y, z);
	return gv.tile_flag[b >> 9] && gv.cell_count[b] > 0;
}

/// The types of a tile's cells and of the ring around it, once per workgroup in LDS: H[hx + 10 hy + 100 hz] = type of cell
/// (8 tx + hx - 1, ..) in bits 0-2 (a cell outside the processed tiles is solid / air by the solid mask), is_unknown_at in bit 3. The per-cell kernels below look at six neighbours per cell:
/// straight from global memory that is a tile lookup + two dependent byte loads each (k_rhs 0.18 ms, k_apply_pressure 0.18 ms,
/// k_abits 0.09 ms at C4 for a few hundred MB).
#define HT_UNKNOWN 8
__device__ inline void stage_halo_types(const GridView &gv, int tx, int ty, int tz, uint8_t *H) {
	for (int i = threadIdx.x; i < LFA_HALO_CELLS; i += 256) {
		const int x = 8 * tx + i % 10 - 1, y = 8 * ty + (i / 10) % 10 - 1, z = 8 * tz + i / 100 - 1;
		uint8_t t = CT_SOLID;  // mac_grid::get_cell_and_type (src/mac_grid.cpp:26-31): outside the grid => solid
		if (in_grid(gv.g, x, y, z)) {
			const uint32_t b = blocked_index(gv.g, x, y, z);
			if (gv.tile_flag[b >> 9]) t = (uint8_t)((gv.ctype[b] & 7) | (gv.cell_count[b] > 0 ? HT_UNKNOWN : 0));
			else t = gv.solid[b] ? CT_SOLID : CT_AIR;
		}
		H[i] = t;
	}
	__syncthreads();
}
__device__ inline int halo_index(int l) { return ((l & 7) + 1) + 10 * (((l >> 3) & 7) + 1) + 100 * ((l >> 6) + 1); }

// ---------------------------------------------------------------------------------------------- a10: A bits
/// pressure_solver::_compute_a_matrix (src/pressure_solver.cpp:160-178) for every cell of every particle tile;
/// non-unknown cells get 0 so that the PCG kernels can use the byte as a mask.
/// `tile_closed` (may be null): 1 for a tile whose unknowns couple to nothing outside the tile - no unknown on a tile face has a
/// coupling across it (the +d bit of the cell, or, towards -d, a fluid cell next to an unknown: pressure_solver.cpp:334-362) - and
/// are at most LFA_CLOSED_MAX_UNKNOWNS: spray. Its block of the matrix is solved on its own (mg.hip: k_mg_solve_closed).
__global__ void __launch_bounds__(256) k_abits(const int *ptiles, int n_ptiles, GridView gv, uint8_t *abits, uint8_t *tile_closed) {
	__shared__ uint8_t H[LFA_HALO_CELLS];
	for (int slot = blockIdx.x; slot < n_ptiles; slot += gridDim.x) {
		const int tile = ptiles[slot];
		int tx, ty, tz;
		tile_coords(gv.g, tile, tx, ty, tz);
		__syncthreads();
		stage_halo_types(gv, tx, ty, tz, H);
		int n_unknown = 0, open = 0;
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			const int h = halo_index(l);
			uint8_t a = 0;
			if (in_grid(gv.g, x, y, z) && (H[h] & HT_UNKNOWN)) {
				const int txp = H[h + 1] & 7, typ = H[h + 10] & 7, tzp = H[h + 100] & 7;
				int ns = (txp != CT_SOLID) + (typ != CT_SOLID) + (tzp != CT_SOLID) + ((H[h - 1] & 7) != CT_SOLID) +
				         ((H[h - 10] & 7) != CT_SOLID) + ((H[h - 100] & 7) != CT_SOLID);
				const bool fluid = (H[h] & 7) == CT_FLUID;
				a = (uint8_t)(ns | ((txp == CT_FLUID) << 3) | ((typ == CT_FLUID) << 4) | ((tzp == CT_FLUID) << 5) |
				              AB_UNKNOWN | (fluid ? AB_FLUID : 0));
				++n_unknown;
				// a coupling that leaves the tile: upwards the cell's own bit, downwards a fluid cell beside an unknown
				const int lx = l & 7, ly = (l >> 3) & 7, lz = l >> 6;
				open |= (lx == 7 && txp == CT_FLUID) | (ly == 7 && typ == CT_FLUID) | (lz == 7 && tzp == CT_FLUID);
				open |= fluid && ((lx == 0 && (H[h - 1] & HT_UNKNOWN)) | (ly == 0 && (H[h - 10] & HT_UNKNOWN)) | (lz == 0 && (H[h - 100] & HT_UNKNOWN)));
			}
			abits[b] = a;
		}
		if (tile_closed) {  // (uniform)
			const int any_open = __syncthreads_or(open);
			const int n1 = __syncthreads_count(n_unknown >= 1), n2 = __syncthreads_count(n_unknown >= 2);
			if (threadIdx.x == 0) tile_closed[tile] = (uint8_t)(!any_open && n1 + n2 >= 1 && n1 + n2 <= LFA_CLOSED_MAX_UNKNOWNS);
		}
	}
}

// ---------------------------------------------------------------------------------------------- a11: divergence rhs
/// pressure_solver::_compute_b_vector (src/pressure_solver.cpp:180-242), same term order; writes r = b for every cell
/// of every particle tile (0 where the cell is not an unknown) and zeroes p.
template <typename real>
__global__ void __launch_bounds__(256)
k_rhs(const int *ptiles, int n_ptiles, GridView gv, const float *u, const float *v, const float *w, real *r, real *p,
      float inv_h, double *part_b2, uint32_t *tile_epoch, uint32_t keep_epoch, uint32_t new_epoch) {
	__shared__ double lds[4];
	__shared__ uint8_t H[LFA_HALO_CELLS];
	double acc = 0.0;
	for (int slot = blockIdx.x; slot < n_ptiles; slot += gridDim.x) {
		const int tile = ptiles[slot];
		int tx, ty, tz;
		tile_coords(gv.g, tile, tx, ty, tz);
		__syncthreads();
		stage_halo_types(gv, tx, ty, tz, H);
		// warm start (lfa_params.pcg_warm_start): the pressure of the previous solve is the initial guess where the tile was
		// solved then (a tile that has been out of the set holds pressures of some older step: dropped)
		const bool keep = keep_epoch != 0 && tile_epoch[tile] == keep_epoch;
		__syncthreads();
		if (threadIdx.x == 0 && tile_epoch) tile_epoch[tile] = new_epoch;
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			const int h = halo_index(l);
			real out = (real)0, guess = (real)0;
			if (in_grid(gv.g, x, y, z) && (H[h] & HT_UNKNOWN)) {
				if (keep) guess = p[b];
				const float vx = u[b], vy = v[b], vz = w[b];
				float val = -(vx + vy + vz);
				if (x > 0) {
					uint32_t n = blocked_index(gv.g, x - 1, y, z);
					float f = u[n];
					val += f;
					if ((H[h - 1] & 7) == CT_SOLID) val -= f;
				}
				if (y > 0) {
					uint32_t n = blocked_index(gv.g, x, y - 1, z);
					float f = v[n];
					val += f;
					if ((H[h - 10] & 7) == CT_SOLID) val -= f;
				}
				if (z > 0) {
					uint32_t n = blocked_index(gv.g, x, y, z - 1);
					float f = w[n];
					val += f;
					if ((H[h - 100] & 7) == CT_SOLID) val -= f;
				}
				if ((H[h + 1] & 7) == CT_SOLID) val += vx;
				if ((H[h + 10] & 7) == CT_SOLID) val += vy;
				if ((H[h + 100] & 7) == CT_SOLID) val += vz;
				out = (real)(inv_h * val);
				acc += (double)out * (double)out;
			}
			r[b] = out;
			p[b] = guess;
		}
	}
	acc = wave_sum(acc);
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) part_b2[blockIdx.x] = (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// ---------------------------------------------------------------------------------------------- a17: pressure gradient
/// pressure_solver::apply_pressure (src/pressure_solver.cpp:73-148) restated per FACE instead of per unknown: the +d
/// face of cell c (neighbour n = c + d) receives, in the reference's visiting order (c before n),
///   from c if c is an unknown:  n non-solid ? u -= coeff*((n fluid ? p[n] : 0) - p[c]) : u = 0
///   from n if n is an unknown:  c air ? u -= coeff*p[n] : (c solid ? u = 0 : nothing)
template <typename real>
__global__ void __launch_bounds__(256)
k_apply_pressure(const int *dtiles, int n_dtiles, GridView gv, float *u, float *v, float *w, const real *p, float coeff) {
	__shared__ uint8_t H[LFA_HALO_CELLS];
	for (int slot = blockIdx.x; slot < n_dtiles; slot += gridDim.x) {
		const int tile = dtiles[slot];
		int tx, ty, tz;
		tile_coords(gv.g, tile, tx, ty, tz);
		__syncthreads();
		stage_halo_types(gv, tx, ty, tz, H);
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
			if (!in_grid(gv.g, x, y, z)) continue;
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			const int h = halo_index(l);
			const bool uc = (H[h] & HT_UNKNOWN) != 0;
			const int tc = H[h] & 7;
			const float pc = uc ? (float)p[b] : 0.0f;
			float *vel[3] = {u, v, w};
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const int nx = x + (d == 0), ny = y + (d == 1), nz = z + (d == 2);
				const int hn = h + (d == 0 ? 1 : (d == 1 ? 10 : 100));
				const int tn = H[hn] & 7;
				const bool un = (H[hn] & HT_UNKNOWN) != 0;
				if (!uc && !un) continue;
				float val = vel[d][b];
				const float pn = un ? (float)p[blocked_index(gv.g, nx, ny, nz)] : 0.0f;
				if (uc) {
					if (tn != CT_SOLID) val -= coeff * ((tn == CT_FLUID ? pn : 0.0f) - pc);
					else val = 0.0f;
				}
				if (un) {
					if (tc == CT_AIR) val -= coeff * pn;
					else if (tc == CT_SOLID) val = 0.0f;
				}
				vel[d][b] = val;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------------- a18: extrapolation
/// One iteration of simulation::_extrapolate_velocities (src/simulation.cpp:700-752). Reads only valid cells and writes
/// only invalid ones, so the sweep is order independent and runs in place.
__global__ void __launch_bounds__(256)
k_extrapolate(const int *dtiles, int n_dtiles, GridView gv, float *u, float *v, float *w, const uint8_t *valid_in,
              uint8_t *valid_out) {
	for (int slot = blockIdx.x; slot < n_dtiles; slot += gridDim.x) {
		const int tile = dtiles[slot];
		int tx, ty, tz;
		tile_coords(gv.g, tile, tx, ty, tz);
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			const int x = tx * 8 + (l & 7), y = ty * 8 + ((l >> 3) & 7), z = tz * 8 + (l >> 6);
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			if (!in_grid(gv.g, x, y, z)) {
				if (valid_out) valid_out[b] = 0;
				continue;
			}
			auto valid_at = [&](int xx, int yy, int zz) -> bool {
				if (!valid_in) return is_unknown_at(gv, xx, yy, zz);
				if (!in_grid(gv.g, xx, yy, zz)) return false;
				uint32_t n = blocked_index(gv.g, xx, yy, zz);
				return gv.tile_flag[n >> 9] && valid_in[n];
			};
			if (valid_at(x, y, z)) {
				if (valid_out) valid_out[b] = 1;
				continue;
			}
			int cnt = 0;
			float sum[3] = {0.f, 0.f, 0.f};
			int tpos[3] = {CT_SOLID, CT_SOLID, CT_SOLID};
#pragma unroll
			for (int d = 0; d < 3; ++d) {
				const int c[3] = {x, y, z};
				const int n_[3] = {gv.g.nx, gv.g.ny, gv.g.nz};
				if (c[d] > 0) {
					const int xx = x - (d == 0), yy = y - (d == 1), zz = z - (d == 2);
					if (valid_at(xx, yy, zz)) {
						uint32_t n = blocked_index(gv.g, xx, yy, zz);
						sum[0] += u[n]; sum[1] += v[n]; sum[2] += w[n];
						++cnt;
					}
				}
				if (c[d] + 1 < n_[d]) {
					const int xx = x + (d == 0), yy = y + (d == 1), zz = z + (d == 2);
					if (valid_at(xx, yy, zz)) {
						uint32_t n = blocked_index(gv.g, xx, yy, zz);
						sum[0] += u[n]; sum[1] += v[n]; sum[2] += w[n];
						tpos[d] = gv.ctype[n] & 7;
						++cnt;
					}
				}
			}
			if (cnt > 0) {
				const int mine = gv.ctype[b] & 7;
				const float fc = (float)cnt;
				if (mine == tpos[0]) u[b] = sum[0] / fc;
				if (mine == tpos[1]) v[b] = sum[1] / fc;
				if (mine == tpos[2]) w[b] = sum[2] / fc;
			}
			if (valid_out) valid_out[b] = cnt > 0 ? 1 : 0;
		}
	}
}

// ---------------------------------------------------------------------------------------------- a19/a20: G2P
/// Staggered sample of mac_grid::get_face_samples (src/mac_grid.cpp:51-112) for halo position (hx,hy,hz) of a tile:
/// out-of-range cells replicate the border cell, and the component along a clamped axis is zero; note that the LAST
/// cell of an axis counts as clamped (_clamp returns `clamped` for val >= max, :42-50).
__device__ inline float clamped_sample(const GridDims &g, const float *f, int comp, int x, int y, int z) {
	const int c[3] = {x, y, z};
	const int n[3] = {g.nx, g.ny, g.nz};
	if (c[comp] < 0 || c[comp] >= n[comp] - 1) return 0.0f;
	const int xx = min(max(x, 0), g.nx - 1), yy = min(max(y, 0), g.ny - 1), zz = min(max(z, 0), g.nz - 1);
	return f[blocked_index(g, xx, yy, zz)];
}

__device__ inline float lerp_ref(float a, float b, float t) { return a * (1.0f - t) + b * t; }  // include/fluid/misc.h:20-22

struct G2PParams {
	int method;
	float blend;
	float inv_h;  // 1 / cell_size: _grad_kernel divides by it (src/simulation.cpp:223); a multiply here (72 IEEE
	              // divisions per particle made the kernel VALU-bound), identical for power-of-two cell sizes
};

/// The transfer of one particle (PIC :447-461, FLIP blend :463-505, APIC + _calculate_c_vector :507-546).
/// corner(b0, b1, b2) names the sample cell particle cell + (b0, b1, b2); fetch(field, comp, corner, k) = clamped staggered
/// sample of field (0-2: u v w, 3-5: FLIP's old grid) at that cell + ((k & 1), (k >> 1) & 1, k >> 2).
/// `pold`, `j`: where FLIP finds the particle's velocity from before the step (a deferred binning leaves it in the other buffer).
/// Returns |v|^2 of the new velocity (the CFL reduction of simulation::cfl rides along, src/simulation.cpp:199-205).
template <int METHOD, typename Corner, typename Fetch>
__device__ inline float g2p_particle(const ParticleSoA &p, uint32_t i, const ParticleSoA &pold, uint32_t j, const G2PParams &gp,
                                    Corner &&corner, Fetch &&fetch) {
	const float t[3] = {p.t[0][i], p.t[1][i], p.t[2][i]};
	float vnew[3], vold[3];
	float cvec[9];
#pragma unroll
	for (int comp = 0; comp < 3; ++comp) {
		// (b, f) per axis exactly as in the P2G scatter: own axis -> (idx-1, t), other axes -> (d-1, tmid)
		int b[3];
		float f[3];
#pragma unroll
		for (int a = 0; a < 3; ++a) {
			if (a == comp) {
				b[a] = t[a] >= 1.0f ? 0 : -1;
				f[a] = t[a] >= 1.0f ? 0.0f : t[a];
			} else {
				b[a] = t[a] < 0.5f ? -1 : 0;
				f[a] = t[a] < 0.5f ? t[a] + 0.5f : t[a] - 0.5f;
			}
		}
		const auto base = corner(b[0], b[1], b[2]);
		float s[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) s[k] = fetch(comp, comp, base, k);
		// trilerp nesting of include/fluid/misc.h:24-36: x innermost, then y, then z
		vnew[comp] = lerp_ref(lerp_ref(lerp_ref(s[0], s[1], f[0]), lerp_ref(s[2], s[3], f[0]), f[1]),
		                      lerp_ref(lerp_ref(s[4], s[5], f[0]), lerp_ref(s[6], s[7], f[0]), f[1]), f[2]);
		if (METHOD == LFA_FLIP_BLEND) {
			float o[8];
#pragma unroll
			for (int k = 0; k < 8; ++k) o[k] = fetch(3 + comp, comp, base, k);
			vold[comp] = lerp_ref(lerp_ref(lerp_ref(o[0], o[1], f[0]), lerp_ref(o[2], o[3], f[0]), f[1]),
			                      lerp_ref(lerp_ref(o[4], o[5], f[0]), lerp_ref(o[6], o[7], f[0]), f[1]), f[2]);
		}
		if (METHOD == LFA_APIC) {
			// _calculate_c_vector: sum_k grad_kernel(f - corner_k) * s_k, _grad_kernel sign rule d > 0 ? -1 : +1
			float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const float px = f[0] - (float)(k & 1), py = f[1] - (float)((k >> 1) & 1), pz = f[2] - (float)(k >> 2);
				const float sx = px > 0.f ? -1.f : 1.f, sy = py > 0.f ? -1.f : 1.f, sz = pz > 0.f ? -1.f : 1.f;
				const float ax = 1.f - fabsf(px), ay = 1.f - fabsf(py), az = 1.f - fabsf(pz);
				cx = cx + (sx * ay * az * gp.inv_h) * s[k];
				cy = cy + (ax * sy * az * gp.inv_h) * s[k];
				cz = cz + (ax * ay * sz * gp.inv_h) * s[k];
			}
			cvec[3 * comp] = cx; cvec[3 * comp + 1] = cy; cvec[3 * comp + 2] = cz;
		}
	}
	float vout[3];
	if (METHOD == LFA_FLIP_BLEND) {
#pragma unroll
		for (int k = 0; k < 3; ++k) vout[k] = vnew[k] + (pold.v[k][j] - vold[k]) * gp.blend;
	} else {
#pragma unroll
		for (int k = 0; k < 3; ++k) vout[k] = vnew[k];
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) p.v[k][i] = vout[k];
	if (METHOD == LFA_APIC) {
#pragma unroll
		for (int k = 0; k < 9; ++k) p.c[k][i] = cvec[k];
	}
	return vout[0] * vout[0] + vout[1] * vout[1] + vout[2] * vout[2];
}

/// One workgroup per particle tile: stage u,v,w (and FLIP's old grid) of the tile + 1-cell ring in LDS with the
/// clamping rule applied, then every particle of the tile gathers its 3x8 samples from LDS.
/// STALE: the particles still sit in the order of the last binning but have moved since (position correction): those
/// whose cell has left the tile are appended to `leavers` for k_g2p_leavers instead (a second binning of all particles
/// for the sake of the 1-3 % that crossed a tile face cost 2.1 ms of the 30 ms full step at C4).
#define G2P_LV_CAP 256
#ifndef G2P_THREADS
#define G2P_THREADS 256  // (C4: 128 threads 1.42 ms, 256 1.28, 512 1.90)
#endif
template <int METHOD, bool STALE>
__global__ void __launch_bounds__(G2P_THREADS)
k_g2p(const int *ptiles, int n_ptiles, GridDims g, ParticleSoA p, const uint32_t *tile_start, const float *u,
      const float *v, const float *w, const float *uo, const float *vo, const float *wo, G2PParams gp, uint32_t *leavers,
      uint32_t *n_leavers, ParticleSoA pold, const uint32_t *from, uint32_t *vmax2_bits) {
	constexpr int NF = METHOD == LFA_FLIP_BLEND ? 6 : 3;
	__shared__ float lds[NF * LFA_HALO_CELLS];
	__shared__ uint32_t lv[STALE ? G2P_LV_CAP : 1], lv_n, lv_base;
	float vmax2 = 0.0f;
	for (int slot = blockIdx.x; slot < n_ptiles; slot += gridDim.x) {
		const int tile = ptiles[slot];
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		__syncthreads();
		if (STALE && threadIdx.x == 0) lv_n = 0;
		for (int i = threadIdx.x; i < LFA_HALO_CELLS; i += G2P_THREADS) {
			const int hx = i % 10, hy = (i / 10) % 10, hz = i / 100;
			const int x = tx * 8 + hx - 1, y = ty * 8 + hy - 1, z = tz * 8 + hz - 1;
			lds[i] = clamped_sample(g, u, 0, x, y, z);
			lds[LFA_HALO_CELLS + i] = clamped_sample(g, v, 1, x, y, z);
			lds[2 * LFA_HALO_CELLS + i] = clamped_sample(g, w, 2, x, y, z);
			if (METHOD == LFA_FLIP_BLEND) {
				lds[3 * LFA_HALO_CELLS + i] = clamped_sample(g, uo, 0, x, y, z);
				lds[4 * LFA_HALO_CELLS + i] = clamped_sample(g, vo, 1, x, y, z);
				lds[5 * LFA_HALO_CELLS + i] = clamped_sample(g, wo, 2, x, y, z);
			}
		}
		__syncthreads();
		const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
		for (uint32_t i = beg + threadIdx.x; i < end; i += G2P_THREADS) {
			const uint32_t key = p.key[i];
			if (STALE && (int)(key >> 9) != tile) {
				const uint32_t at = atomicAdd(&lv_n, 1u);
				if (at < G2P_LV_CAP) lv[at] = i;
				else leavers[atomicAdd(n_leavers, 1u)] = i;  // more than the LDS list holds: straight to the global list
				continue;
			}
			const int l = (int)(key & 511);
			const int cell = ((l & 7) + 1) + 10 * (((l >> 3) & 7) + 1) + 100 * ((l >> 6) + 1);
			const float v2 = g2p_particle<METHOD>(
			    p, i, pold, (METHOD == LFA_FLIP_BLEND && from) ? from[i] : i, gp,
			    [&](int b0, int b1, int b2) { return cell + b0 + 10 * b1 + 100 * b2; },
			    [&](int field, int, int base, int k) { return lds[field * LFA_HALO_CELLS + base + (k & 1) + 10 * ((k >> 1) & 1) + 100 * (k >> 2)]; });
			vmax2 = fmaxf(vmax2, v2);
		}
		if (STALE) {  // one global atomic per tile
			__syncthreads();
			const uint32_t n = lv_n < G2P_LV_CAP ? lv_n : G2P_LV_CAP;
			if (threadIdx.x == 0 && n) lv_base = atomicAdd(n_leavers, n);
			__syncthreads();
			for (uint32_t k = threadIdx.x; k < n; k += G2P_THREADS) leavers[lv_base + k] = lv[k];
		}
	}
	// max |v|^2 over the particles this workgroup transferred: non-negative floats order like their bit patterns
	vmax2 = wave_max(vmax2);
	if ((threadIdx.x & 63) == 0 && vmax2 > 0.0f) atomicMax(vmax2_bits, __float_as_uint(vmax2));
}

/// Slabs: the particles received from the neighbour ranks since the binning, [first, first + count), join the leaver list.
__global__ void k_g2p_append_range(uint32_t *leavers, uint32_t *n_leavers, uint32_t first, uint32_t count) {
	__shared__ uint32_t base;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t mine = count - min(count, blockIdx.x * blockDim.x);
	if (threadIdx.x == 0) base = atomicAdd(n_leavers, mine < blockDim.x ? mine : blockDim.x);
	__syncthreads();
	if (i < count) leavers[base + threadIdx.x] = first + i;
}

/// The particles k_g2p<.., STALE> set aside: the same transfer with the samples gathered from the grid in global memory.
template <int METHOD>
__global__ void __launch_bounds__(256)
k_g2p_leavers(GridDims g, ParticleSoA p, const float *u, const float *v, const float *w, const float *uo, const float *vo,
              const float *wo, G2PParams gp, const uint32_t *leavers, const uint32_t *n_leavers, ParticleSoA pold, const uint32_t *from,
              uint32_t *vmax2_bits) {
	const uint32_t n = *n_leavers;
	float vmax2 = 0.0f;
	for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
		const uint32_t i = leavers[k], key = p.key[i];
		if (key == 0xFFFFFFFFu) continue;
		const int tile = (int)(key >> 9), l = (int)(key & 511);
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		const int cx = tx * 8 + (l & 7), cy = ty * 8 + ((l >> 3) & 7), cz = tz * 8 + (l >> 6);
		const float *F[6] = {u, v, w, uo, vo, wo};
		const float v2 = g2p_particle<METHOD>(
		    p, i, pold, (METHOD == LFA_FLIP_BLEND && from) ? from[i] : i, gp,
		    [&](int b0, int b1, int b2) { return make_int3(cx + b0, cy + b1, cz + b2); },
		    [&](int field, int comp, int3 c, int k) { return clamped_sample(g, F[field], comp, c.x + (k & 1), c.y + ((k >> 1) & 1), c.z + (k >> 2)); });
		vmax2 = fmaxf(vmax2, v2);
	}
	vmax2 = wave_max(vmax2);
	if ((threadIdx.x & 63) == 0 && vmax2 > 0.0f) atomicMax(vmax2_bits, __float_as_uint(vmax2));
}
}  // namespace

static int grid_blocks(int n) { return n < 16384 ? (n > 0 ? n : 1) : 16384; }

/// Refreshes the ghost tile layers of the grid fields from the neighbour ranks (no-op without a decomposition).
/// with_topology: also cell types and particle counts (they change once per step, in the P2G).
int lfa_dist_refresh_grid(lfa_sim *s, bool with_topology) {
	if (!s->dist) return LFA_OK;
	void *f[9];
	int e[9], n = 0;
	f[n] = s->u; e[n++] = 4;
	f[n] = s->v; e[n++] = 4;
	f[n] = s->w; e[n++] = 4;
	if (with_topology) {
		f[n] = s->ctype; e[n++] = 1;
		f[n] = s->cell_count; e[n++] = 4;
		if (s->prm.simulation_method == LFA_FLIP_BLEND && s->uo) {
			f[n] = s->uo; e[n++] = 4;
			f[n] = s->vo; e[n++] = 4;
			f[n] = s->wo; e[n++] = 4;
		}
	}
	return lfa_dist_exchange_fields(s, n, f, e);
}
static GridView make_view(lfa_sim *s) { return GridView{s->g, s->ctype, s->solid, s->cell_count, s->tile_flag}; }

/// a8-a11: unknown set, A bits and divergence; r = b, p = 0. The MIC(0) factor is built by pcg.hip.
int lfa_build_rhs(lfa_sim *s, double dt) {
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_build_system: call lfa_hash_particles first");
	LFA_TRY(lfa_pcg_alloc(s));
	LFA_TRY(lfa_dist_refresh_grid(s, true));  // types / counts / velocities of the neighbour slab's adjacent tile layer
	s->a_scale = dt / (s->prm.density * s->prm.cell_size * s->prm.cell_size);  // src/pressure_solver.cpp:22
	s->sys_dt = dt;
	if (!s->n_ptiles) return LFA_OK;
	GridView gv = make_view(s);
	if (!s->tile_closed) {
		LFA_HIP(s, hipMalloc(&s->tile_closed, (size_t)s->g.nt));
		LFA_HIP(s, hipMemsetAsync(s->tile_closed, 0, (size_t)s->g.nt, s->stream));
	}
	hipLaunchKernelGGL(k_abits, dim3(grid_blocks(s->n_ptiles)), dim3(256), 0, s->stream, s->ptiles, s->n_ptiles, gv,
	                   s->abits, s->tile_closed);
	LFA_LAUNCH_CHECK(s);
	// (once per solve and bandwidth bound: as many workgroups as the partial-sum array holds - the cap of the iteration's kernels,
	// lfa_pcg_grid_cap = 768, is tuned for their latency chains and costs this kernel 83 us at C4: 94 -> 177)
	const int G = pcg_grid_uncapped(s->n_ptiles);
	const float inv_h = (float)(1.0 / s->prm.cell_size);
	// warm start: only for the solver's own preconditioners (the exact MIC(0) schedule is the reference-parity path and keeps
	// the reference's zero guess, src/pressure_solver.cpp:36), only from the pressure of the immediately preceding solve
	if (!s->tile_epoch) {
		LFA_HIP(s, hipMalloc(&s->tile_epoch, (size_t)s->g.nt * 4));
		LFA_HIP(s, hipMemsetAsync(s->tile_epoch, 0, (size_t)s->g.nt * 4, s->stream));
	}
	s->warm_started = s->prm.pcg_warm_start && s->prm.precond != LFA_PRECOND_MIC0_EXACT && !s->dist && s->pressure_epoch == s->solve_epoch &&
	                  s->solve_epoch != 0;
	const uint32_t keep_epoch = s->warm_started ? s->solve_epoch : 0u;
	++s->solve_epoch;
	if (s->solve_epoch == 0) s->solve_epoch = 1;
	if (s->prm.pcg_dtype == LFA_PCG_F64)
		hipLaunchKernelGGL(k_rhs<double>, dim3(G), dim3(256), 0, s->stream, s->ptiles, s->n_ptiles, gv, s->u, s->v, s->w,
		                   (double *)s->vr, (double *)s->vp, inv_h, s->partials + PART_B2, s->tile_epoch, keep_epoch, s->solve_epoch);
	else
		hipLaunchKernelGGL(k_rhs<float>, dim3(G), dim3(256), 0, s->stream, s->ptiles, s->n_ptiles, gv, s->u, s->v, s->w,
		                   (float *)s->vr, (float *)s->vp, inv_h, s->partials + PART_B2, s->tile_epoch, keep_epoch, s->solve_epoch);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

extern "C" int lfa_apply_pressure(lfa_sim *s, double dt) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned || !s->vp) return lfa_fail(s, LFA_E_INVALID, "lfa_apply_pressure: no pressure on the device");
	LFA_HIP(s, hipSetDevice(s->device));
	// the +z face of the top owned cell needs the pressure of the cell above it (owned by the upper neighbour)
	if (s->dist) LFA_TRY(lfa_dist_exchange_slices(s, s->vp, s->prm.pcg_dtype == LFA_PCG_F64 ? 8 : 4));
	if (!s->n_dtiles) return LFA_OK;
	const float coeff = (float)(dt / (s->prm.density * s->prm.cell_size));  // src/pressure_solver.cpp:74
	GridView gv = make_view(s);
	dim3 grid(grid_blocks(s->n_dtiles));
	if (s->prm.pcg_dtype == LFA_PCG_F64)
		hipLaunchKernelGGL(k_apply_pressure<double>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, gv, s->u, s->v,
		                   s->w, (const double *)s->vp, coeff);
	else
		hipLaunchKernelGGL(k_apply_pressure<float>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, gv, s->u, s->v,
		                   s->w, (const float *)s->vp, coeff);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

extern "C" int lfa_extrapolate(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_extrapolate: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	const int iters = (int)s->prm.velocity_extrapolation_iterations;
	if (iters == 0) return LFA_OK;
	LFA_TRY(lfa_dist_refresh_grid(s, false));  // post-projection velocities of the neighbour's adjacent layer
	if (!s->n_dtiles) return LFA_OK;
	GridView gv = make_view(s);
	dim3 grid(grid_blocks(s->n_dtiles));
	if (iters == 1) {
		hipLaunchKernelGGL(k_extrapolate, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, gv, s->u, s->v, s->w,
		                   (const uint8_t *)nullptr, (uint8_t *)nullptr);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	// several iterations: cells made valid by iteration i are used from iteration i+1 on (src/simulation.cpp:694-698).
	// The validity masks live in the (otherwise idle at this point) PCG scratch vector q: two byte planes.
	LFA_TRY(lfa_pcg_alloc(s));
	uint8_t *va = (uint8_t *)s->vq, *vb = va + s->ncp;
	for (int i = 0; i < iters; ++i) {
		hipLaunchKernelGGL(k_extrapolate, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, gv, s->u, s->v, s->w,
		                   i == 0 ? (const uint8_t *)nullptr : (const uint8_t *)va, vb);
		LFA_LAUNCH_CHECK(s);
		uint8_t *t = va; va = vb; vb = t;
		// slabs: the next sweep reads what this one has made valid in the neighbour's adjacent tile layer - its velocities and
		// its validity bytes (a sweep reaches one cell, the ghost layer is eight deep: nothing beyond it can matter)
		if (s->dist && i + 1 < iters) {
			LFA_TRY(lfa_dist_refresh_grid(s, false));
			void *f[1] = {va};
			const int e[1] = {1};
			LFA_TRY(lfa_dist_exchange_fields(s, 1, f, e));
		}
	}
	return LFA_OK;
}

/// stale: the particles have moved since the last binning but keep its order (lfa_time_step after the position correction)
static int g2p_run(lfa_sim *s, bool stale) {
	if (!s) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_g2p: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_dist_refresh_grid(s, false));  // extrapolated velocities of the neighbour's adjacent layer
	const bool arrivals = stale && s->dist && s->n_arrivals;  // (a rank without particle tiles may still have received some)
	if (!s->n_ptiles && !arrivals) {
		s->vc_pending = false;
		return LFA_OK;
	}
	G2PParams gp;
	gp.method = s->prm.simulation_method;
	gp.blend = (float)s->prm.blending_factor;
	gp.inv_h = (float)(1.0 / s->prm.cell_size);
	dim3 grid(grid_blocks(s->n_ptiles));
	const ParticleSoA &p = s->pb[s->cur];
	if (s->prm.simulation_method == LFA_FLIP_BLEND && !s->uo) return lfa_fail(s, LFA_E_INVALID, "lfa_g2p: FLIP needs the old grid written by lfa_p2g");
	// leaver list: the binning's rank array is free between two binnings; its counter sits behind the PCG state words
	uint32_t *leavers = s->rank, *n_leavers = (uint32_t *)(s->pcg_state + 6);
	// a deferred binning (lfa_sim::vc_pending): FLIP's old velocity still sits in the other buffer; PIC and APIC overwrite
	// v, C without reading them. Either way the transfer completes the particle records in the binned order.
	const ParticleSoA &pold = s->vc_pending ? s->pb[s->cur ^ 1] : s->pb[s->cur];
	const uint32_t *from = s->vc_pending ? (const uint32_t *)s->vc_src : (const uint32_t *)nullptr;
	if (stale) LFA_HIP(s, hipMemsetAsync(n_leavers, 0, 4, s->stream));
	// the CFL reduction rides along: max |v|^2 of the velocities this transfer writes (every live particle gets one)
	uint32_t *vmax2_bits = (uint32_t *)(s->pcg_state + 7);
	LFA_HIP(s, hipMemsetAsync(vmax2_bits, 0, 4, s->stream));
	// (the STALE instantiation serves both cases: it allocates 62 VGPRs where the plain one gets 129 - 8 instead of 3 waves
	// per SIMD; on freshly binned particles it finds no leavers)
#define G2P_LAUNCH(M)                                                                                                        \
	do {                                                                                                                     \
		if (s->n_ptiles)                                                                                                     \
			hipLaunchKernelGGL((k_g2p<M, true>), grid, dim3(G2P_THREADS), 0, s->stream, s->ptiles, s->n_ptiles, s->g, p, s->tile_start, \
			                   s->u, s->v, s->w, s->uo, s->vo, s->wo, gp, leavers, n_leavers, pold, from, vmax2_bits);       \
		if (arrivals)                                                                                                        \
			hipLaunchKernelGGL(k_g2p_append_range, dim3((unsigned)((s->n_arrivals + 255) / 256)), dim3(256), 0, s->stream, leavers, \
			                   n_leavers, (uint32_t)s->arrivals_at, (uint32_t)s->n_arrivals);                                 \
		if (stale)                                                                                                           \
			hipLaunchKernelGGL(k_g2p_leavers<M>, dim3(512), dim3(256), 0, s->stream, s->g, p, s->u, s->v, s->w, s->uo, s->vo, \
			                   s->wo, gp, (const uint32_t *)leavers, (const uint32_t *)n_leavers, pold, from, vmax2_bits);   \
	} while (0)
	switch (s->prm.simulation_method) {
	case LFA_PIC: G2P_LAUNCH(LFA_PIC); break;
	case LFA_FLIP_BLEND: G2P_LAUNCH(LFA_FLIP_BLEND); break;
	default: G2P_LAUNCH(LFA_APIC); break;
	}
#undef G2P_LAUNCH
	LFA_LAUNCH_CHECK(s);
	s->vc_pending = false;
	// max |v|^2 goes to the host behind the kernels that reduce it: lfa_cfl (the next step's dt) then only waits for the stream
	// instead of queueing a copy of its own behind an idle device
	LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 100, s->pcg_state + 7, 4, hipMemcpyDeviceToHost, s->stream));
	s->vmax2_valid = true;  // every live particle has just got its velocity: the binned ones, the leavers and (slabs) the arrivals
	return LFA_OK;
}
/// (the public entry point takes the leaver path too: it costs a 4-byte memset and an empty launch on freshly binned
/// particles, and makes the call valid after lfa_correct_collide without a second binning as long as the processed tiles
/// still cover the particles' new cells)
extern "C" int lfa_g2p(lfa_sim *s) { return g2p_run(s, true); }
int lfa_g2p_stale(lfa_sim *s) { return g2p_run(s, true); }

int lfa_g2p_bench(lfa_sim *s) { return lfa_g2p(s); }
