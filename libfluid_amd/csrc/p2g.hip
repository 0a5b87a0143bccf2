// libfluid_amd/csrc/p2g.hip -- particle-to-grid transfer on the MAC grid.
//
// Reference: simulation::_transfer_to_grid_{pic,flip,apic} (src/simulation.cpp:293-398) is a serial GATHER: every cell
// visits the particles of its 27 neighbour cells and evaluates the trilinear hat (_kernel, :207-213) at its three
// face centres. Here it is the transposed SCATTER: each particle touches exactly the 2x2x2 face samples per component
// whose hat weight is non-zero, evaluated in cell-relative coordinates (key = cell, t = fraction in the cell):
//     axis a of component c:  s = t_a - (a == c ? 1 : 1/2),  b = floor(s) in {-1,0},  f = s - b
//     target cell offset  b + i (i = 0,1), 1-D weight  (1-f, f),  (p - face)_a / h = f - i
// which is the same set of (cell, weight) pairs the gather produces.
//
// Two variants (BASELINE config 2):
//   LDS-binned   one workgroup per particle tile accumulates (sum w v, sum w) for its 10x10x10 halo block in LDS
//                (64-bit fixed point, ds_add_u64: order independent), streams the tile's particles once, coalesced, and
//                stores the block to a per-tile staging slab; the finalize kernel adds the up to 8 overlapping slabs per
//                cell in a fixed order. No global atomics.
//   global-atomic one thread per particle, 48 global_atomic_add_f32 into dense accumulators.
// Finalize = normalise (weight > 1e-6 else 0, :324/:383), cell typing (:329-334), APIC boundary-face zeroing
// (_remove_boundary_velocities :428-445), FLIP's old-grid copy (:340-344) and, inside lfa_step_hot, gravity (:72-78).
#include "common.h"

namespace {

/// (b, f) of one axis. `own` = this axis is the component's own (non-staggered) axis.
/// own: s = t - 1  -> b = -1, f = t (b = 0, f = 0 when the particle sits on the max face, t == 1);
/// else: s = t - 1/2 -> (b, f) = (-1, t + 1/2) or (0, t - 1/2): the `tmid` of src/mac_grid.cpp:80-93.
__device__ inline void axis_bf(float t, bool own, int &b, float &f) {
	if (own) {
		b = t >= 1.0f ? 0 : -1;
		f = t >= 1.0f ? 0.0f : t;
	} else {
		b = t < 0.5f ? -1 : 0;
		f = t < 0.5f ? t + 0.5f : t - 0.5f;
	}
}

/// Accumulates one particle into a 10x10x10 block of (sum_wv, sum_w) pairs per component.
/// acc layout: [comp][2][1000], halo cell index = hx + 10 hy + 100 hz.
/// QUIRK (APIC only): the reference evaluates the APIC hat on world-space distances, _kernel(p - face) WITHOUT the division
/// by cell_size that PIC has (src/simulation.cpp:367-369 vs :313-315). For cell_size == 1 both are the same 2x2x2 stencil;
/// otherwise the 1-D weight towards the face of cell (particle cell + o) is max(0, 1 - h |d|) with d the distance in cells,
/// o = -1, 0, +1 being exactly the cells whose 27-neighbourhood gather (simulation.h:212-223) visits the particle:
/// h > 1 narrows the hat inside the 2x2x2 set, h < 1 widens it to (up to) all 3x3x3 cells, truncated there by the gather.
template <bool APIC, bool QUIRK, typename AddFn>
__device__ inline void scatter_particle(int lx, int ly, int lz, const float t[3], const float v[3], const float c[9],
                                        float hworld, int rot, AddFn add) {
	if (QUIRK) {
#pragma unroll
		for (int comp = 0; comp < 3; ++comp) {
			float w1[3][3], a1[3][3];  // [axis][o + 1]
			const float *cc = c + 3 * comp;
#pragma unroll
			for (int a = 0; a < 3; ++a)
#pragma unroll
				for (int o = -1; o <= 1; ++o) {
					const float d = t[a] - (float)o - (a == comp ? 1.0f : 0.5f);  // (p - face)_a in cells
					w1[a][o + 1] = fmaxf(0.0f, 1.0f - hworld * fabsf(d));
					a1[a][o + 1] = -hworld * cc[a] * d;  // c_comp[a] * (face - p)_a, :371-375
				}
			for (int k = 0; k < 3; ++k)
				for (int j = 0; j < 3; ++j)
#pragma unroll
					for (int i = 0; i < 3; ++i) {
						const float wgt = (w1[0][i] * w1[1][j]) * w1[2][k];
						if (wgt > 0.0f) {
							const float val = v[comp] + ((a1[0][i] + a1[1][j]) + a1[2][k]);
							add(comp, lx + i, ly + j, lz + k, wgt * val, wgt);
						}
					}
		}
		return;
	}
#pragma unroll
	for (int comp = 0; comp < 3; ++comp) {
		int b[3];
		float f[3];
#pragma unroll
		for (int a = 0; a < 3; ++a) axis_bf(t[a], a == comp, b[a], f[a]);
		const int hx0 = lx + 1 + b[0], hy0 = ly + 1 + b[1], hz0 = lz + 1 + b[2];
		float wx[2] = {1.0f - f[0], f[0]}, wy[2] = {1.0f - f[1], f[1]}, wz[2] = {1.0f - f[2], f[2]};
		float ax[2] = {0.f, 0.f}, ay[2] = {0.f, 0.f}, az[2] = {0.f, 0.f};
		if (APIC) {
			// affine term dot(c_comp, face - p) (src/simulation.cpp:371-375); (face - p)_a = -(f_a - i) * h
			const float *cc = c + 3 * comp;
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				ax[i] = -hworld * cc[0] * (f[0] - (float)i);
				ay[i] = -hworld * cc[1] * (f[1] - (float)i);
				az[i] = -hworld * cc[2] * (f[2] - (float)i);
			}
		}
		// The eight nodes are visited starting from node `rot` (0: in order). The LDS-binned kernel passes the lane number:
		// consecutive particles of one cell - neighbouring lanes once the flow has made the order inside a tile spatially
		// coherent - then add to eight DIFFERENT nodes at the same time instead of queueing up on one LDS word.
#pragma unroll
		for (int n = 0; n < 8; ++n) {
			const int m = (n + rot) & 7, i = m & 1, j = (m >> 1) & 1, k = m >> 2;
			const float wgt = ((i ? wx[1] : wx[0]) * (j ? wy[1] : wy[0])) * (k ? wz[1] : wz[0]);  // product order of _kernel, :209-212
			const float val = APIC ? v[comp] + (((i ? ax[1] : ax[0]) + (j ? ay[1] : ay[0])) + (k ? az[1] : az[0])) : v[comp];
			add(comp, hx0 + i, hy0 + j, hz0 + k, wgt * val, wgt);
		}
	}
}

// ------------------------------------------------------------------------------------------------ LDS-binned
// Accumulation primitive. Measured on MI355X (tools/lds_atomic_bench.hip, lane-ops/clk/CU, random addresses):
//   ds_add_f32 0.33 | ds_add_f64 2.3 | ds_add_u64 4.3 | ds_add_u32 6.7 | ds_write_b32 7.2
// i.e. the fp32 LDS atomic is ~20x slower than the 64-bit integer one on gfx950. The tile therefore accumulates in
// 64-bit FIXED POINT: integer adds are associative, so the per-tile sums are also bit-reproducible whatever order the
// lanes arrive in. Units: sum w in 2^-40 (a contribution is <= 1, a sum < 2^10), sum w v in 2^-36 (|w v| < 2^15 per
// contribution - the range of the magic-constant conversion below -, |sum| < 2^27). The velocity of a face is
// sum wv / sum w down to sum w = 1e-6 (src/simulation.cpp:324,383), so its error is |v| * err(sum w) / sum w: with the
// 2^-30 units of round 1 a face that only sees the far corner of one hat (sum w ~ 1e-5: routine with the narrow
// un-scaled APIC hat at cell_size > 1) was off by 1e-4 of |v|; now 2^-41 n / 1e-6 < 4e-6 even at the threshold.
#define P2G_FIX_SCALE_W 1099511627776.0   /* 2^40 */
#define P2G_FIX_SCALE_V 68719476736.0     /* 2^36 */
/// float -> fixed via the 1.5*2^52 magic constant: two f64 ops and one 64-bit subtract (no f32->i64 convert on CDNA).
/// Valid for |x * scale| < 2^51.
__device__ inline unsigned long long to_fixed(float x, double scale) {
	const double magic = 6755399441055744.0;  // 1.5 * 2^52
	double d = fma((double)x, scale, magic);
	return (unsigned long long)(__double_as_longlong(d) - __double_as_longlong(magic));
}

/// The 2 x 2 x 2 scatter of one particle straight into the LDS accumulators, trimmed for the binned kernel (its VALU work - 725
/// instructions per particle, 76 % of the SIMD cycles - bounds it as much as the LDS atomics do):
/// * the lane's rotation is an XOR (node n of the unrolled loop is node n ^ rot, a bijection for every rot, so the lanes of a cell
///   still hit eight different nodes at the same time) and is applied ONCE per axis by swapping the axis' pair of weights, affine
///   terms and offsets - the nodes then index those pairs with compile-time bits: no select, no index arithmetic per node;
/// * the fixed-point scales ride on the factors: the x weights carry 2^40, v and the affine terms 2^-4 (powers of two: every
///   product and sum rounds as before, the results are bit-identical), so a value converts with one cvt, one f64 add of the magic
///   constant and one 64-bit subtract instead of cvt + two moves + fma + subtract.
template <bool APIC>
__device__ inline void scatter_particle_lds(unsigned long long *acc, int lx, int ly, int lz, const float t[3], const float v[3],
                                            const float c[9], float hworld, int rot) {
	const bool r0 = (rot & 1) != 0, r1 = (rot & 2) != 0, r2 = (rot & 4) != 0;
	const double magic = 6755399441055744.0;  // 1.5 * 2^52 (to_fixed)
	auto conv = [&](float scaled) -> unsigned long long {
		return (unsigned long long)(__double_as_longlong((double)scaled + magic) - __double_as_longlong(magic));
	};
	const float sw = (float)P2G_FIX_SCALE_W, sv = (float)(P2G_FIX_SCALE_V / P2G_FIX_SCALE_W);
#pragma unroll
	for (int comp = 0; comp < 3; ++comp) {
		int b[3];
		float f[3];
#pragma unroll
		for (int a = 0; a < 3; ++a) axis_bf(t[a], a == comp, b[a], f[a]);
		// byte offsets: node (i, j, k) = base + ox[i] + oy[j] + oz[k] (the sum of w sits LFA_HALO_CELLS entries behind the sum of w v)
		const uint32_t a0 = 8u * (uint32_t)(comp * 2 * LFA_HALO_CELLS + ((lx + 1 + b[0]) + 10 * (ly + 1 + b[1]) + 100 * (lz + 1 + b[2])));
		const float wx_lo = (1.0f - f[0]) * sw, wx_hi = f[0] * sw, wy_lo = 1.0f - f[1], wy_hi = f[1], wz_lo = 1.0f - f[2], wz_hi = f[2];
		const float wx[2] = {r0 ? wx_hi : wx_lo, r0 ? wx_lo : wx_hi}, wy[2] = {r1 ? wy_hi : wy_lo, r1 ? wy_lo : wy_hi},
		            wz[2] = {r2 ? wz_hi : wz_lo, r2 ? wz_lo : wz_hi};
		const uint32_t ox[2] = {r0 ? 8u : 0u, r0 ? 0u : 8u}, oy[2] = {r1 ? 80u : 0u, r1 ? 0u : 80u};
		const uint32_t oz[2] = {a0 + (r2 ? 800u : 0u), a0 + (r2 ? 0u : 800u)};
		const uint32_t oxy[4] = {ox[0] + oy[0], ox[1] + oy[0], ox[0] + oy[1], ox[1] + oy[1]};
		float ax[2] = {0.f, 0.f}, ay[2] = {0.f, 0.f}, az[2] = {0.f, 0.f};
		const float vs = v[comp] * sv;
		if (APIC) {
			// affine term dot(c_comp, face - p) (src/simulation.cpp:371-375); (face - p)_a = -(f_a - i) * h
			const float *cc = c + 3 * comp;
			const float ax_lo = (-hworld * cc[0] * f[0]) * sv, ax_hi = (-hworld * cc[0] * (f[0] - 1.0f)) * sv;
			const float ay_lo = (-hworld * cc[1] * f[1]) * sv, ay_hi = (-hworld * cc[1] * (f[1] - 1.0f)) * sv;
			const float az_lo = (-hworld * cc[2] * f[2]) * sv, az_hi = (-hworld * cc[2] * (f[2] - 1.0f)) * sv;
			ax[0] = r0 ? ax_hi : ax_lo; ax[1] = r0 ? ax_lo : ax_hi;
			ay[0] = r1 ? ay_hi : ay_lo; ay[1] = r1 ? ay_lo : ay_hi;
			az[0] = r2 ? az_hi : az_lo; az[1] = r2 ? az_lo : az_hi;
		}
#pragma unroll
		for (int n = 0; n < 8; ++n) {
			const int i = n & 1, j = (n >> 1) & 1, k = n >> 2;
			const float wgt = (wx[i] * wy[j]) * wz[k];  // product order of _kernel, :209-212 (x 2^40)
			const float val = APIC ? vs + ((ax[i] + ay[j]) + az[k]) : vs;  // (x 2^-4)
			unsigned long long *a = (unsigned long long *)((char *)acc + (oxy[i + 2 * j] + oz[k]));
			atomicAdd(a, conv(wgt * val));
			atomicAdd(a + LFA_HALO_CELLS, conv(wgt));
		}
	}
}

struct ParticleRegs {
	uint32_t key;
	float t[3], v[3], c[9];
};
/// key, t from `p` at i; v, C from `pvc` at j (a deferred binning leaves them in the other buffer: j = vc_src[i], else
/// pvc == p and j == i).
template <bool APIC>
__device__ inline void load_particle(const ParticleSoA &p, const ParticleSoA &pvc, uint32_t i, uint32_t j, ParticleRegs &r) {
	r.key = p.key[i];
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		r.t[k] = p.t[k][i];
		r.v[k] = pvc.v[k][j];
	}
	if (APIC) {
#pragma unroll
		for (int k = 0; k < 9; ++k) r.c[k] = pvc.c[k][j];
	}
}

#ifndef P2G_THREADS
#define P2G_THREADS 512  // threads of a scatter workgroup (3 workgroups per CU by LDS; measured at C4: 256 1.60, 384 1.67, 512 1.53, 768 1.89 ms)
#endif
template <bool APIC, bool QUIRK>
__global__ void __launch_bounds__(P2G_THREADS)
k_p2g_binned(const int *ptiles, int n_ptiles, ParticleSoA p, ParticleSoA pvc, const uint32_t *from, const uint32_t *tile_start,
             float *stage, float hworld, int rot_mask) {
	__shared__ unsigned long long acc[6 * LFA_HALO_CELLS];  // 48 KB: [comp][wv | w][10x10x10]
	for (int slot = blockIdx.x; slot < n_ptiles; slot += gridDim.x) {
		const int tile = ptiles[slot];
		for (int i = threadIdx.x; i < 6 * LFA_HALO_CELLS; i += P2G_THREADS) acc[i] = 0ull;
		__syncthreads();
		const uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
		// software pipeline: the loads of the next particle are in flight while the current one is scattered
		// (with a deferred binning the index of the particle after next is loaded one round ahead of its v, C)
		ParticleRegs cur, nxt;
		uint32_t i = beg + threadIdx.x;
		uint32_t jn = i + P2G_THREADS < end ? (from ? from[i + P2G_THREADS] : i + P2G_THREADS) : 0u;
		if (i < end) load_particle<APIC>(p, pvc, i, from ? from[i] : i, cur);
		for (; i < end; i += P2G_THREADS) {
			const uint32_t in = i + P2G_THREADS;
			const uint32_t jnn = in + P2G_THREADS < end ? (from ? from[in + P2G_THREADS] : in + P2G_THREADS) : 0u;
			if (in < end) load_particle<APIC>(p, pvc, in, jn, nxt);
			jn = jnn;
			const int l = (int)(cur.key & 511);
			if (!QUIRK)
				scatter_particle_lds<APIC>(acc, l & 7, (l >> 3) & 7, l >> 6, cur.t, cur.v, cur.c, hworld, (int)(threadIdx.x & rot_mask));
			else
				scatter_particle<APIC, QUIRK>(l & 7, (l >> 3) & 7, l >> 6, cur.t, cur.v, cur.c, hworld, (int)(threadIdx.x & rot_mask),
				                       [&](int comp, int hx, int hy, int hz, float wv, float wgt) {
					                       unsigned long long *a = acc + comp * 2 * LFA_HALO_CELLS + hx + 10 * hy + 100 * hz;
					                       atomicAdd(a, to_fixed(wv, P2G_FIX_SCALE_V));
					                       atomicAdd(a + LFA_HALO_CELLS, to_fixed(wgt, P2G_FIX_SCALE_W));
				                       });
			cur = nxt;
		}
		__syncthreads();
		float *out = stage + (size_t)slot * 6 * LFA_HALO_CELLS;
		for (int k = threadIdx.x; k < 6 * LFA_HALO_CELLS; k += P2G_THREADS) {  // (slab order: lfa_stage_index)
			const int ch = k / LFA_HALO_CELLS;
			out[k] = (float)((double)(long long)acc[ch * LFA_HALO_CELLS + lfa_stage_source(k - ch * LFA_HALO_CELLS)] *
			                 ((ch & 1) ? 1.0 / P2G_FIX_SCALE_W : 1.0 / P2G_FIX_SCALE_V));
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------------------------------ global atomics
__global__ void k_zero_acc(const int *dtiles, int n_dtiles, float *acc, size_t ncp) {
	int slot = blockIdx.x;
	if (slot >= n_dtiles) return;
	size_t base = (size_t)dtiles[slot] * LFA_TILE_CELLS;
	for (int k = 0; k < 6; ++k) {
		acc[k * ncp + base + threadIdx.x] = 0.0f;
		acc[k * ncp + base + 256 + threadIdx.x] = 0.0f;
	}
}

template <bool APIC, bool QUIRK>
__global__ void __launch_bounds__(256)
k_p2g_atomic(size_t n, ParticleSoA p, float *acc, size_t ncp, GridDims g, float hworld) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t key = p.key[i];
	const int tile = (int)(key >> 9), l = (int)(key & 511);
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	const int cx = tx * 8 + (l & 7), cy = ty * 8 + ((l >> 3) & 7), cz = tz * 8 + (l >> 6);
	float t[3] = {p.t[0][i], p.t[1][i], p.t[2][i]};
	float v[3] = {p.v[0][i], p.v[1][i], p.v[2][i]};
	float c[9];
	if (APIC) {
#pragma unroll
		for (int k = 0; k < 9; ++k) c[k] = p.c[k][i];
	}
	// local coords 0 here: hx = 1 + b + i  => cell offset = hx - 1
	scatter_particle<APIC, QUIRK>(0, 0, 0, t, v, c, hworld, 0, [&](int comp, int hx, int hy, int hz, float wv, float wgt) {
		int x = cx + hx - 1, y = cy + hy - 1, z = cz + hz - 1;
		if (!in_grid(g, x, y, z)) return;
		size_t b = blocked_index(g, x, y, z);
		unsafeAtomicAdd(acc + (size_t)(2 * comp) * ncp + b, wv);
		unsafeAtomicAdd(acc + (size_t)(2 * comp + 1) * ncp + b, wgt);
	});
}

// ------------------------------------------------------------------------------------------------ finalize
struct FinalizeParams {
	int method;        // LFA_PIC / FLIP / APIC
	float g[3];        // gravity * dt when fused, else 0
	int fuse_gravity;
};

template <bool BINNED>
__global__ void __launch_bounds__(256)
k_p2g_finalize(const int *dtiles, int n_dtiles, GridDims g, const int *tile_pslot, const float *stage, const float *acc,
               size_t ncp, const uint32_t *cell_count, const uint8_t *solid, float *u, float *v, float *w, float *uo,
               float *vo, float *wo, uint8_t *ctype, FinalizeParams fp) {
	// (XCD-chunked slot order measured here: 0.42 instead of 0.31 ms at C4 - not used)
	__shared__ int ps27[27];  // staging slot of the 27 tiles around this one (-1: none)
	for (int slot = blockIdx.x; slot < n_dtiles; slot += gridDim.x) {
		const int tile = dtiles[slot];
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		if (BINNED) {
			__syncthreads();
			if (threadIdx.x < 27) {
				const int nx_ = tx + (int)threadIdx.x % 3 - 1, ny_ = ty + ((int)threadIdx.x / 3) % 3 - 1, nz_ = tz + (int)threadIdx.x / 9 - 1;
				const bool in = (unsigned)nx_ < (unsigned)g.ntx && (unsigned)ny_ < (unsigned)g.nty && (unsigned)nz_ < (unsigned)g.ntz;
				ps27[threadIdx.x] = in ? tile_pslot[nx_ + g.ntx * (ny_ + g.nty * nz_)] : -1;
			}
			__syncthreads();
		}
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			const int lx = l & 7, ly = (l >> 3) & 7, lz = l >> 6;
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			const int x = tx * 8 + lx, y = ty * 8 + ly, z = tz * 8 + lz;
			float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
			if (BINNED) {
				// fixed visiting order (ascending tile offsets) => the sum does not depend on which workgroup ran first. Only the
				// blocks that reach this cell are visited: its own tile's and, for a cell on a face of the tile, the neighbour's there
				// (slots from LDS: no dependent global look-up per block. Unrolling the eight combinations with every load in
				// flight at once was measured too: 0.30 instead of 0.245 ms at C4)
				const int ox0 = lx == 0 ? -1 : 0, ox1 = lx == 7 ? 1 : 0, oy0 = ly == 0 ? -1 : 0, oy1 = ly == 7 ? 1 : 0;
				const int oz0 = lz == 0 ? -1 : 0, oz1 = lz == 7 ? 1 : 0;
				for (int oz = oz0; oz <= oz1; ++oz)
					for (int oy = oy0; oy <= oy1; ++oy)
						for (int ox = ox0; ox <= ox1; ++ox) {
							const int ps = ps27[(ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)];
							if (ps < 0) continue;
							const float *src = stage + (size_t)ps * 6 * LFA_HALO_CELLS + lfa_stage_index(lx - 8 * ox + 1, ly - 8 * oy + 1, lz - 8 * oz + 1);
#pragma unroll
							for (int k = 0; k < 6; ++k) s[k] += src[k * LFA_HALO_CELLS];
						}
			} else {
#pragma unroll
				for (int k = 0; k < 6; ++k) s[k] = acc[(size_t)k * ncp + b];
			}
			const bool inside = in_grid(g, x, y, z);
			float vel[3];
#pragma unroll
			for (int k = 0; k < 3; ++k) vel[k] = s[2 * k + 1] > 1e-6f ? s[2 * k] / s[2 * k + 1] : 0.0f;
			uint8_t type;
			if (!inside) {
				vel[0] = vel[1] = vel[2] = 0.0f;
				type = CT_SOLID | CT_OUTSIDE;
			} else if (solid[b]) {
				type = CT_SOLID;
			} else {
				type = cell_count[b] > 0 ? CT_FLUID : CT_AIR;
			}
			const bool bx = x == g.nx - 1, by = y == g.ny - 1, bz = z == g.nz - 1;
			if (fp.method == LFA_FLIP_BLEND && inside) {
				uo[b] = bx ? 0.0f : vel[0];
				vo[b] = by ? 0.0f : vel[1];
				wo[b] = bz ? 0.0f : vel[2];
			}
			if (fp.method == LFA_APIC) {
				if (bx) vel[0] = 0.0f;
				if (by) vel[1] = 0.0f;
				if (bz) vel[2] = 0.0f;
			}
			if (fp.fuse_gravity && inside) {
				vel[0] += fp.g[0]; vel[1] += fp.g[1]; vel[2] += fp.g[2];
			}
			u[b] = vel[0]; v[b] = vel[1]; w[b] = vel[2];
			ctype[b] = type;
		}
	}
}

__global__ void __launch_bounds__(256)
k_add_gravity(const int *dtiles, int n_dtiles, GridDims g, float *u, float *v, float *w, float gx, float gy, float gz) {
	for (int slot = blockIdx.x; slot < n_dtiles; slot += gridDim.x) {
		const int tile = dtiles[slot];
		int tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
#pragma unroll
		for (int half = 0; half < 2; ++half) {
			const int l = threadIdx.x + 256 * half;
			if (!in_grid(g, tx * 8 + (l & 7), ty * 8 + ((l >> 3) & 7), tz * 8 + (l >> 6))) continue;
			const size_t b = (size_t)tile * LFA_TILE_CELLS + l;
			u[b] += gx; v[b] += gy; w[b] += gz;
		}
	}
}
}  // namespace

static int grid_blocks(int n) { return n < 16384 ? (n > 0 ? n : 1) : 16384; }

/// 0 = PIC / FLIP, 1 = APIC, 2 = APIC with the reference's unscaled hat at cell_size != 1 (scatter_particle's QUIRK)
static int scatter_mode(const lfa_sim *s) {
	if (s->prm.simulation_method != LFA_APIC) return 0;
	return (s->prm.apic_unscaled_kernel && s->prm.cell_size != 1.0) ? 2 : 1;
}
static void launch_binned(lfa_sim *s, const ParticleSoA &p, const ParticleSoA &pvc, const uint32_t *from, float *stage_own) {
	const dim3 grid(grid_blocks(s->n_ptiles));
	const float hworld = (float)s->prm.cell_size;
#define LB(A, Q)                                                                                                             \
	hipLaunchKernelGGL((k_p2g_binned<A, Q>), grid, dim3(P2G_THREADS), 0, s->stream, s->ptiles, s->n_ptiles, p, pvc, from, s->tile_start, \
	                   stage_own, hworld, rot_mask)
	const int rot_mask = 7;  // the lane-rotated node order (profiles/r02_p2g_lds_pmc.txt)
	switch (scatter_mode(s)) {
	case 0: LB(false, false); break;
	case 1: LB(true, false); break;
	default: LB(true, true); break;
	}
#undef LB
}
static void launch_atomic(lfa_sim *s, const ParticleSoA &p) {
	const dim3 grid((unsigned)((s->np_live + 255) / 256));
	const float hworld = (float)s->prm.cell_size;
#define LA(A, Q) hipLaunchKernelGGL((k_p2g_atomic<A, Q>), grid, dim3(256), 0, s->stream, s->np_live, p, s->acc, s->ncp, s->g, hworld)
	switch (scatter_mode(s)) {
	case 0: LA(false, false); break;
	case 1: LA(true, false); break;
	default: LA(true, true); break;
	}
#undef LA
}

int lfa_p2g_run(lfa_sim *s, bool fuse_gravity, double dt) {
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "lfa_p2g: call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	const int method = s->prm.simulation_method;
	if (method == LFA_FLIP_BLEND && !s->uo) {
		LFA_HIP(s, hipMalloc(&s->uo, s->ncp * 4));
		LFA_HIP(s, hipMalloc(&s->vo, s->ncp * 4));
		LFA_HIP(s, hipMalloc(&s->wo, s->ncp * 4));
		LFA_HIP(s, hipMemsetAsync(s->uo, 0, s->ncp * 4, s->stream));
		LFA_HIP(s, hipMemsetAsync(s->vo, 0, s->ncp * 4, s->stream));
		LFA_HIP(s, hipMemsetAsync(s->wo, 0, s->ncp * 4, s->stream));
	}
	const bool binned = s->prm.p2g_variant == LFA_P2G_LDS_BINNED;
	if (!binned) LFA_TRY(lfa_particles_materialize(s));  // the global-atomic variant reads v, C in place
	// a deferred binning: v, C come from the other buffer through vc_src
	const ParticleSoA &p = s->pb[s->cur], &pvc = s->vc_pending ? s->pb[s->cur ^ 1] : s->pb[s->cur];
	const uint32_t *from = s->vc_pending ? (const uint32_t *)s->vc_src : (const uint32_t *)nullptr;
	if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[16], s->stream));
	if (binned) {
		if ((size_t)s->n_ptiles_all > s->stage_tiles) {
			if (s->stage) LFA_HIP(s, hipFree(s->stage));
			s->stage = nullptr;
			size_t want = (size_t)s->n_ptiles_all + (size_t)s->n_ptiles_all / 4 + 16;
			hipError_t e = hipMalloc(&s->stage, want * 6 * LFA_HALO_CELLS * 4);
			if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of the P2G staging slabs (%zu tiles) failed", want);
			s->stage_tiles = want;
		}
		if (s->n_ptiles) {
			launch_binned(s, p, pvc, from, s->stage + (size_t)s->p_off * 6 * LFA_HALO_CELLS);
			LFA_LAUNCH_CHECK(s);
		}
		// particles within one cell of a slab face also contribute to the neighbour rank's faces
		LFA_TRY(lfa_dist_exchange_p2g_planes(s, s->stage));
	} else {
		if (s->dist) return lfa_fail(s, LFA_E_UNSUPPORTED, "the global-atomic P2G variant is single-GPU only");
		if (!s->acc) {
			hipError_t e = hipMalloc(&s->acc, s->ncp * 6 * 4);
			if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of the atomic P2G accumulators failed");
		}
		if (s->n_dtiles) {
			hipLaunchKernelGGL(k_zero_acc, dim3(s->n_dtiles), dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, s->acc,
			                   s->ncp);
			LFA_LAUNCH_CHECK(s);
		}
		if (s->np_live) {
			launch_atomic(s, p);
			LFA_LAUNCH_CHECK(s);
		}
	}
	if (s->timing) LFA_HIP(s, hipEventRecord(s->ev[17], s->stream));
	FinalizeParams fp;
	fp.method = method;
	fp.fuse_gravity = fuse_gravity ? 1 : 0;
	for (int k = 0; k < 3; ++k) fp.g[k] = fuse_gravity ? (float)(s->prm.gravity[k] * dt) : 0.0f;
	if (s->n_dtiles) {
		dim3 grid(grid_blocks(s->n_dtiles));
		if (binned)
			hipLaunchKernelGGL(k_p2g_finalize<true>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, s->g,
			                   s->tile_pslot, s->stage, s->acc, s->ncp, s->cell_count, s->solid, s->u, s->v, s->w, s->uo,
			                   s->vo, s->wo, s->ctype, fp);
		else
			hipLaunchKernelGGL(k_p2g_finalize<false>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, s->g,
			                   s->tile_pslot, s->stage, s->acc, s->ncp, s->cell_count, s->solid, s->u, s->v, s->w, s->uo,
			                   s->vo, s->wo, s->ctype, fp);
		LFA_LAUNCH_CHECK(s);
	}
	LFA_HIP(s, hipMemcpyAsync(s->grid_flag, s->tile_flag, (size_t)s->g.nt * 4, hipMemcpyDeviceToDevice, s->stream));
	s->grid_valid = true;
	s->system_valid = false;
	for (int k = 0; k < 3; ++k) s->bg[k] = fuse_gravity ? s->prm.gravity[k] * dt : 0.0;
	return LFA_OK;
}

extern "C" int lfa_p2g(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	return lfa_p2g_run(s, false, 0.0);
}

extern "C" int lfa_add_gravity(lfa_sim *s, double dt) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	if (s->binned && s->n_dtiles) {
		hipLaunchKernelGGL(k_add_gravity, dim3(grid_blocks(s->n_dtiles)), dim3(256), 0, s->stream, s->dtiles, s->n_dtiles,
		                   s->g, s->u, s->v, s->w, (float)(s->prm.gravity[0] * dt), (float)(s->prm.gravity[1] * dt),
		                   (float)(s->prm.gravity[2] * dt));
		LFA_LAUNCH_CHECK(s);
	}
	// every cell outside the processed tiles carries the same increment implicitly (see k_export_cells)
	for (int k = 0; k < 3; ++k) s->bg[k] += s->prm.gravity[k] * dt;
	s->system_valid = false;
	return LFA_OK;
}

/// One launch of the P2G scatter kernel / finalize kernel / binning pass on the current state (lfa_bench_kernel).
int lfa_p2g_bench(lfa_sim *s, int which) {
	if (which == LFA_K_P2G_SCATTER && s->prm.p2g_variant != LFA_P2G_LDS_BINNED) LFA_TRY(lfa_particles_materialize(s));
	const ParticleSoA &p = s->pb[s->cur], &pvc = s->vc_pending ? s->pb[s->cur ^ 1] : s->pb[s->cur];
	const uint32_t *from = s->vc_pending ? (const uint32_t *)s->vc_src : (const uint32_t *)nullptr;
	if (which == LFA_K_P2G_SCATTER) {
		if (s->prm.p2g_variant == LFA_P2G_LDS_BINNED) {
			if (!s->stage || (size_t)s->n_ptiles_all > s->stage_tiles) return lfa_fail(s, LFA_E_INVALID, "no staging slabs");
			launch_binned(s, p, pvc, from, s->stage + (size_t)s->p_off * 6 * LFA_HALO_CELLS);
		} else {
			if (!s->acc) return lfa_fail(s, LFA_E_INVALID, "no accumulators");
			launch_atomic(s, p);
		}
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	if (which == LFA_K_P2G_FINALIZE) {
		FinalizeParams fp;
		fp.method = s->prm.simulation_method;
		fp.fuse_gravity = 0;
		fp.g[0] = fp.g[1] = fp.g[2] = 0.f;
		dim3 grid(grid_blocks(s->n_dtiles));
		if (s->prm.p2g_variant == LFA_P2G_LDS_BINNED)
			hipLaunchKernelGGL(k_p2g_finalize<true>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, s->g,
			                   s->tile_pslot, s->stage, s->acc, s->ncp, s->cell_count, s->solid, s->u, s->v, s->w, s->uo,
			                   s->vo, s->wo, s->ctype, fp);
		else
			hipLaunchKernelGGL(k_p2g_finalize<false>, grid, dim3(256), 0, s->stream, s->dtiles, s->n_dtiles, s->g,
			                   s->tile_pslot, s->stage, s->acc, s->ncp, s->cell_count, s->solid, s->u, s->v, s->w, s->uo,
			                   s->vo, s->wo, s->ctype, fp);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	if (which == LFA_K_BIN) return lfa_hash_particles(s);
	return lfa_fail(s, LFA_E_INVALID, "unknown kernel id %d", which);
}
