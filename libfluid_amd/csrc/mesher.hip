// libfluid_amd/csrc/mesher.hip -- surface mesher on the device.
//
// Reference: fluid::mesher (include/fluid/mesher.h:14-46, src/mesher.cpp:320-515): an implicit surface function is
// sampled at the points of a grid from the particles near each point (weights (1 - d^2/extent^2)^3, value = distance to
// the weighted mean position minus the weighted mean radius), then marching cubes (Bourke's tables) extracts the zero
// surface with shared vertices.
//
// Parity contract: bit-exact -- sampled values, vertex positions, vertex order and the index list equal the
// reference's. That fixes more than the arithmetic (fp64, the reference's operation order, -ffp-contract=off):
//  * the sums over particles run in the reference's order: cells of the neighbourhood in grid order, and inside a cell
//    newest particle first (its space hash is a linked list with head insertion, space_hashing.h:55-62). Particles are
//    therefore grouped by cell with an atomic scatter and each cell's group is then sorted by input index.
//  * the vertex numbering of the sequential sweep (src/mesher.cpp:400-515) is reproduced without the sweep: every edge
//    of the sampling grid has exactly one cell that creates its vertex (the cell that sees it as edge 5, 6 or 10, or a
//    cell on the x = 0 / y = 0 / z = 0 faces), and a cell creates its vertices in a fixed order, so an exclusive scan of
//    the per-cell vertex counts in grid order gives every vertex the index the sweep gives it. Triangles follow from a
//    second scan. Both formulations are checked against each other through the oracle, which keeps the sweep.
#include "common.h"
#include "mc_tables.h"

#include <math.h>
#include <string.h>

#include <algorithm>

struct lfa_mesher {
	int device = 0;
	hipStream_t stream = nullptr;
	uint64_t n[3] = {0, 0, 0};  // cells of the WHOLE grid (mesher::resize); the surface function has n + 1 points per axis
	// z-window (lfa_mesher_create_window; the whole grid: z0 = 0, nzl = n[2], everything from 0 to n[2]): the arrays hold the cell
	// layers [z0, z0 + nzl) / point planes [z0, z0 + nzl]; planes [s_lo, s_hi] are sampled, cell layers [c_lo, c_hi) are
	// classified, and vertices / triangles are emitted for the layers from own_lo on (layer own_lo - 1, when it exists, is only
	// there to number the vertices the cells above it share with it)
	uint64_t z0 = 0, nzl = 0, s_lo = 0, s_hi = 0, c_lo = 0, c_hi = 0, own_lo = 0;
	uint32_t *ids = nullptr;       // order keys of the uploaded particles (lfa_mesher_sample_ids), else input order
	size_t ncell = 0, npts = 0;
	double off[3] = {0, 0, 0}, cs = 0.0, extent = 0.5;
	uint64_t radius = 2;
	double *values = nullptr;      // npts
	uint32_t *cell_start = nullptr;  // ncell + 1
	uint32_t *cell_fill = nullptr;   // ncell
	uint32_t *order = nullptr;     // particle indices grouped by cell
	double *pos = nullptr;         // uploaded particle positions
	double *spos = nullptr;        // the same in the visiting order of `order` (contiguous per cell run)
	uint8_t *blk_flag = nullptr;   // per 8^3 block of cells: holds a particle (lets grid points in empty space skip the row walk)
	size_t n_blk_flag = 0;
	size_t pcap = 0;
	uint32_t *vcount = nullptr, *icount = nullptr;  // per cell (+1): vertices created / indices emitted -> offsets
	uint16_t *created = nullptr;
	uint8_t *occ = nullptr;
	uint32_t *blk = nullptr;       // scan scratch
	size_t nblk = 0;
	double *vpos = nullptr;
	uint64_t *vidx = nullptr;
	size_t vcap = 0, icap = 0;
	uint64_t n_vertices = 0, n_indices = 0;
	bool have_mesh = false;
	std::string err;
};

namespace {
constexpr int SCAN_BLOCK = 2048;

int mfail(lfa_mesher *m, int code, const char *msg) {
	if (m) m->err = msg;
	return lfa_fail(nullptr, code, "%s", msg);
}
#define MSH_HIP(m, call)                                                                            \
	do {                                                                                             \
		hipError_t e_ = (call);                                                                      \
		if (e_ != hipSuccess) return mfail((m), e_ == hipErrorOutOfMemory ? LFA_E_OOM : LFA_E_HIP,  \
		                                   (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); \
	} while (0)

// device copies of the case tables of mc_tables.h (filled at create time)
__device__ uint8_t d_edge_corners[12 * 2];
__device__ uint8_t d_corner_offsets[8 * 3];
// edges created before edge e inside one cell; creation order 0 1 2 3 4 7 8 9 11 5 6 10 (see owned_mask)
__constant__ uint32_t d_before[12] = {0x000, 0x001, 0x003, 0x007, 0x00F, 0xB9F, 0xBBF, 0x01F, 0x09F, 0x19F, 0xBFF, 0x39F};
__device__ uint8_t d_tri_table[256 * 16];

struct MeshGrid {
	uint64_t nx, ny, nz;  // cells of the whole grid
	double ox, oy, oz, cs, extent;
	uint64_t radius;
	uint64_t z0, nzl, s_lo, s_hi, c_lo, c_hi, own_lo;  // window (see lfa_mesher); arrays are indexed with z - z0
};

// ------------------------------------------------------------------------------------------------ exclusive scan
/// out[i] = sum of in[0..i) for n entries, out[n] = total. in may alias out. Block sums -> one-workgroup scan -> apply.
__global__ void __launch_bounds__(256) k_block_sums(const uint32_t *in, size_t n, uint32_t *blk) {
	__shared__ uint32_t red[4];
	const size_t b0 = (size_t)blockIdx.x * SCAN_BLOCK;
	uint32_t c = 0;
	for (int k = 0; k < SCAN_BLOCK / 256; ++k) {
		const size_t i = b0 + (size_t)k * 256 + threadIdx.x;
		if (i < n) c += in[i];
	}
	c = wave_sum(c);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) blk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(1024) k_scan_block_sums(uint32_t *blk, size_t n) {
	__shared__ uint32_t part[1024];
	const size_t per = (n + 1023) / 1024, b = threadIdx.x * per, e = b + per < n ? b + per : n;
	uint32_t sum = 0;
	for (size_t i = b; i < e; ++i) sum += blk[i];
	part[threadIdx.x] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t run = 0;
		for (int i = 0; i < 1024; ++i) { const uint32_t v = part[i]; part[i] = run; run += v; }
		blk[n] = run;
	}
	__syncthreads();
	uint32_t run = part[threadIdx.x];
	for (size_t i = b; i < e; ++i) { const uint32_t v = blk[i]; blk[i] = run; run += v; }
}
__global__ void __launch_bounds__(256) k_block_scan_apply(const uint32_t *in, uint32_t *out, size_t n, const uint32_t *blk,
                                                          size_t nblk) {
	__shared__ uint32_t wsum[4];
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	uint32_t base = blk[blockIdx.x];
	const size_t b0 = (size_t)blockIdx.x * SCAN_BLOCK;
	for (int k = 0; k < SCAN_BLOCK / 256; ++k) {
		const size_t i = b0 + (size_t)k * 256 + threadIdx.x;
		const uint32_t v = i < n ? in[i] : 0u;
		uint32_t incl = v;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const uint32_t t = __shfl_up(incl, o, 64);
			if (lane >= o) incl += t;
		}
		__syncthreads();
		if (lane == 63) wsum[wid] = incl;
		__syncthreads();
		uint32_t woff = 0;
		for (int w = 0; w < wid; ++w) woff += wsum[w];
		if (i < n) out[i] = base + woff + incl - v;
		base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = blk[nblk];
}

// ------------------------------------------------------------------------------------------------ particle grouping
/// Cell of a particle as mesher::_sample_surface_function computes it (src/mesher.cpp:335-340): vec3i((p - offset) /
/// cell_size) by truncation; kept only if every index is > 0 (sic) and inside the hash (space_hashing.h:33-49).
__device__ inline uint32_t particle_cell(const MeshGrid &g, const double *p) {
	const int ix = (int)((p[0] - g.ox) / g.cs), iy = (int)((p[1] - g.oy) / g.cs), iz = (int)((p[2] - g.oz) / g.cs);
	if (ix > 0 && iy > 0 && iz > 0 && (uint64_t)ix < g.nx && (uint64_t)iy < g.ny && (uint64_t)iz < g.nz &&
	    (uint64_t)iz >= g.z0 && (uint64_t)iz < g.z0 + g.nzl)  // (and inside the z-window this handle stores)
		return (uint32_t)((uint64_t)ix + g.nx * ((uint64_t)iy + g.ny * ((uint64_t)iz - g.z0)));
	return 0xFFFFFFFFu;
}
__global__ void k_count_particles(MeshGrid g, const double *pos, size_t np, uint32_t *cell_count, uint8_t *blk_flag) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	const uint32_t c = particle_cell(g, pos + 3 * i);
	if (c == 0xFFFFFFFFu) return;
	atomicAdd(&cell_count[c], 1u);
	const uint64_t x = c % g.nx, y = (c / g.nx) % g.ny, z = c / (g.nx * g.ny);  // z: layer inside the window
	blk_flag[(x >> 3) + ((g.nx + 7) >> 3) * ((y >> 3) + ((g.ny + 7) >> 3) * (z >> 3))] = 1;
}
__global__ void k_scatter_particles(MeshGrid g, const double *pos, size_t np, const uint32_t *cell_start, uint32_t *cell_fill,
                                    uint32_t *order) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	const uint32_t c = particle_cell(g, pos + 3 * i);
	if (c != 0xFFFFFFFFu) order[cell_start[c] + atomicAdd(&cell_fill[c], 1u)] = (uint32_t)i;
}
/// The scatter leaves a cell's particles in arbitrary order; the reference visits them newest first, so each group is put
/// in DESCENDING input order. Cells are x fastest, so the cells [x0, x1) of one row of a neighbourhood are one contiguous
/// run of `order` that already is in the reference's visiting order (cells ascending, newest first inside a cell).
/// Groups are a handful of particles: insertion sort, one thread per cell.
/// `ids` (optional): the order key of particle i instead of i itself - a rank of a slab run holds its particles in storage
/// order and orders them by their global ids, which is the single domain's input order.
__global__ void k_sort_groups(const uint32_t *cell_start, size_t ncell, uint32_t *order, const uint32_t *ids) {
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= ncell) return;
	const uint32_t b = cell_start[c], e = cell_start[c + 1];
	for (uint32_t i = b + 1; i < e; ++i) {
		const uint32_t v = order[i];
		uint32_t j = i;
		while (j > b && (ids ? ids[order[j - 1]] < ids[v] : order[j - 1] < v)) {
			order[j] = order[j - 1];
			--j;
		}
		order[j] = v;
	}
}

/// Positions in visiting order: the sampling loop then streams a row of cells instead of chasing order[k] -> pos (two
/// dependent, uncoalesced loads per particle visit; 134 M particles on 270 M points: 258 -> see DESIGN.md).
__global__ void k_gather_positions(const double *pos, const uint32_t *order, const uint32_t *cell_start, size_t ncell, double *spos) {
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= cell_start[ncell]) return;
	const double *p = pos + 3 * (size_t)order[k];
	spos[3 * k] = p[0];
	spos[3 * k + 1] = p[1];
	spos[3 * k + 2] = p[2];
}

// ------------------------------------------------------------------------------------------------ surface function
/// mesher::_sample_surface_function (src/mesher.cpp:342-375), one thread per grid point.
__global__ void __launch_bounds__(256)
k_sample_surface(MeshGrid g, const double *spos, const uint32_t *cell_start, const uint8_t *blk_flag, double r, double *values) {
	const uint64_t px = g.nx + 1, py = g.ny + 1, pz = g.nzl + 1;
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= px * py * pz) return;
	const uint64_t x = i % px, y = (i / px) % py, z = g.z0 + i / (px * py);  // z: plane of the whole grid
	if (z < g.s_lo || z > g.s_hi) return;
	const double gx = g.ox + g.cs * (double)x, gy = g.oy + g.cs * (double)y, gz = g.oz + g.cs * (double)z;
	const uint64_t R = g.radius;
	// for_each_in_range_checked(center, radius, radius - 1): cells [g - R, g + R - 1], clamped (grid.h:116-135)
	const uint64_t x0 = x < R ? 0 : x - R, y0 = y < R ? 0 : y - R, z0 = z < R ? 0 : z - R;
	const uint64_t x1 = x + R < g.nx ? x + R : g.nx, y1 = y + R < g.ny ? y + R : g.ny, z1 = z + R < g.nz ? z + R : g.nz;
	const double e2 = g.extent * g.extent;
	double tw = 0.0, tr = 0.0, tx = 0.0, ty = 0.0, tz = 0.0;
	bool has = false;
	// most grid points of a scene sit in empty space: a look at the (at most 2 x 2 x 2, radius <= 8) blocks of cells the
	// neighbourhood overlaps replaces two cell_start loads per row
	bool maybe = false;
	if (x0 < x1 && y0 < y1 && z0 < z1) {
		const uint64_t bnx = (g.nx + 7) >> 3, bny = (g.ny + 7) >> 3;
		for (uint64_t bz = (z0 - g.z0) >> 3; bz <= (z1 - 1 - g.z0) >> 3; ++bz)
			for (uint64_t by = y0 >> 3; by <= (y1 - 1) >> 3; ++by)
				for (uint64_t bx = x0 >> 3; bx <= (x1 - 1) >> 3; ++bx) maybe |= blk_flag[bx + bnx * (by + bny * bz)] != 0;
	}
	if (maybe)
	for (uint64_t cz = z0; cz < z1; ++cz)
		for (uint64_t cy = y0; cy < y1; ++cy) {
				const size_t row = (size_t)(g.nx * (cy + g.ny * (cz - g.z0)));
				const uint32_t e = cell_start[row + x1];
				for (uint32_t k = cell_start[row + x0]; k < e; ++k) {
					const double *p = spos + 3 * (size_t)k;
					const double qx = p[0], qy = p[1], qz = p[2];
					has = true;
					const double dx = qx - gx, dy = qy - gy, dz = qz - gz;
					double sq = 0.0;
					sq += dx * dx;
					sq += dy * dy;
					sq += dz * dz;
					double w = 1.0 - sq / e2;  // mesher::_kernel, src/mesher.cpp:325-331
					w = w > 0.0 ? w * w * w : 0.0;
					tw += w;
					tr += w * r;
					tx += w * qx;
					ty += w * qy;
					tz += w * qz;
				}
			}
	double value = 1.0;
	if (has) {
		tr /= tw;
		const double mx = tx / tw - gx, my = ty / tw - gy, mz = tz / tw - gz;
		double sq = 0.0;
		sq += mx * mx;
		sq += my * my;
		sq += mz * mz;
		value = sqrt(sq) - tr;
	}
	values[i] = value;
}

// ------------------------------------------------------------------------------------------------ marching cubes
// Creation order of a cell's vertices in the sweep (src/mesher.cpp:431-492): edges 0 1 2 3 (z == 0 face), 4 (y == 0),
// 7 (x == 0), 8 (x == 0 and y == 0), 9 (y == 0), 11 (x == 0), then the three edges every cell owns: 5, 6, 10.
__device__ inline uint32_t owned_mask(uint64_t x, uint64_t y, uint64_t z) {
	uint32_t m = (1u << 5) | (1u << 6) | (1u << 10);
	if (z == 0) {
		m |= (1u << 1) | (1u << 2);
		if (y == 0) m |= 1u << 0;
		if (x == 0) m |= 1u << 3;
	}
	if (y == 0) m |= (1u << 4) | (1u << 9);
	if (x == 0) m |= (1u << 7) | (1u << 11);
	if (x == 0 && y == 0) m |= 1u << 8;
	return m;
}
/// Edges created before edge e inside one cell.
__device__ inline uint32_t before_mask(int e) { return d_before[e]; }
/// mc_edge_mask on the device: an edge carries a vertex when its corners lie on different sides.
__device__ inline uint32_t edge_mask_dev(uint8_t occ) {
	uint32_t m = 0;
#pragma unroll
	for (int e = 0; e < 12; ++e)
		if (((occ >> d_edge_corners[2 * e]) ^ (occ >> d_edge_corners[2 * e + 1])) & 1) m |= 1u << e;
	return m;
}
struct EdgeRef {
	int64_t dx, dy, dz;
	int e;
};
/// The cell that creates the vertex of edge e of cell (x, y, z), and the edge it is there.
__device__ inline EdgeRef edge_owner(int e, uint64_t x, uint64_t y, uint64_t z) {
	switch (e) {
	case 0: if (y > 0 && z > 0) return {0, -1, -1, 6}; if (z > 0) return {0, 0, -1, 4}; if (y > 0) return {0, -1, 0, 2}; break;
	case 1: if (z > 0) return {0, 0, -1, 5}; break;
	case 2: if (z > 0) return {0, 0, -1, 6}; break;
	case 3: if (x > 0 && z > 0) return {-1, 0, -1, 5}; if (z > 0) return {0, 0, -1, 7}; if (x > 0) return {-1, 0, 0, 1}; break;
	case 4: if (y > 0) return {0, -1, 0, 6}; break;
	case 7: if (x > 0) return {-1, 0, 0, 5}; break;
	case 8: if (x > 0 && y > 0) return {-1, -1, 0, 10}; if (x > 0) return {-1, 0, 0, 9}; if (y > 0) return {0, -1, 0, 11}; break;
	case 9: if (y > 0) return {0, -1, 0, 10}; break;
	case 11: if (x > 0) return {-1, 0, 0, 10}; break;
	default: break;
	}
	return {0, 0, 0, e};
}

__device__ inline uint8_t cell_case(const MeshGrid &g, const double *values, uint64_t x, uint64_t y, uint64_t z, double f[8]) {
	const uint64_t px = g.nx + 1, py = g.ny + 1;
	uint8_t occ = 0;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		f[i] = values[(x + d_corner_offsets[3 * i]) + px * ((y + d_corner_offsets[3 * i + 1]) + py * (z - g.z0 + d_corner_offsets[3 * i + 2]))];
		occ |= (uint8_t)((f[i] < 0 ? 1 : 0) << i);
	}
	return occ;
}

__global__ void __launch_bounds__(256)
k_mc_classify(MeshGrid g, const double *values, uint8_t *occ_out, uint16_t *created, uint32_t *vcount, uint32_t *icount) {
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= g.nx * g.ny * g.nzl) return;
	const uint64_t x = c % g.nx, y = (c / g.nx) % g.ny, z = g.z0 + c / (g.nx * g.ny);  // z: layer of the whole grid
	if (z < g.c_lo || z >= g.c_hi) {
		occ_out[c] = 0; created[c] = 0; vcount[c] = 0; icount[c] = 0;
		return;
	}
	double f[8];
	const uint8_t occ = cell_case(g, values, x, y, z, f);
	const uint32_t mine = edge_mask_dev(occ) & owned_mask(x, y, z);
	int ni = 0;
	while (ni < 16 && d_tri_table[occ * 16 + ni] != MC_END) ++ni;
	if (z < g.own_lo) ni = 0;  // the layer below the window's own ones: its triangles belong to the rank below
	occ_out[c] = occ;
	created[c] = (uint16_t)mine;
	vcount[c] = (uint32_t)__popc(mine);
	icount[c] = (uint32_t)ni;
}

/// mesher::_add_point (src/mesher.cpp:378-392) for every vertex a cell creates, at its index in the sweep's numbering.
__global__ void __launch_bounds__(256)
k_mc_vertices(MeshGrid g, const double *values, const uint16_t *created, const uint32_t *vbase, double *vpos, uint32_t vsub) {
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= g.nx * g.ny * g.nzl) return;
	const uint32_t mine = created[c];
	if (!mine) return;
	const uint64_t x = c % g.nx, y = (c / g.nx) % g.ny, z = g.z0 + c / (g.nx * g.ny);
	if (z < g.own_lo) return;  // numbered here, created by the rank below
	double f[8];
	cell_case(g, values, x, y, z, f);
	const double cell[3] = {(double)x, (double)y, (double)z}, off[3] = {g.ox, g.oy, g.oz};
	for (int e = 0; e < 12; ++e) {
		if (!(mine & (1u << e))) continue;
		const int a = d_edge_corners[2 * e], b = d_edge_corners[2 * e + 1];
		const double v1 = f[a], v2 = f[b], t = v1 / (v1 - v2);
		double *o = vpos + 3 * ((size_t)(vbase[c] - vsub) + __popc(mine & before_mask(e)));
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			// vec3d(cell + offset): integer sum converted to double; lerp(a, b, t) = a (1 - t) + b t (misc.h:20-22)
			const double pa = cell[d] + (double)d_corner_offsets[3 * a + d], pb = cell[d] + (double)d_corner_offsets[3 * b + d];
			o[d] = off[d] + g.cs * (pa * (1.0 - t) + pb * t);
		}
	}
}

__global__ void __launch_bounds__(256)
k_mc_triangles(MeshGrid g, const uint8_t *occ_in, const uint16_t *created, const uint32_t *vbase, const uint32_t *ibase,
               uint64_t *vidx, uint32_t vsub) {
	const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= g.nx * g.ny * g.nzl) return;
	const uint8_t occ = occ_in[c];
	if (d_tri_table[occ * 16] == MC_END) return;
	const uint64_t x = c % g.nx, y = (c / g.nx) % g.ny, z = g.z0 + c / (g.nx * g.ny);
	if (z < g.own_lo || z >= g.c_hi) return;
	uint64_t *o = vidx + ibase[c];
	for (int k = 0; k < 16 && d_tri_table[occ * 16 + k] != MC_END; ++k) {
		const EdgeRef r = edge_owner(d_tri_table[occ * 16 + k], x, y, z);
		const size_t oc = (size_t)((int64_t)c + r.dx + (int64_t)g.nx * (r.dy + (int64_t)g.ny * r.dz));
		// relative to the first vertex this window creates: negative (two's complement) for a vertex of the layer below, which
		// the rank below creates; lfa_mesher_rebase adds the number of vertices of all ranks below
		o[k] = (uint64_t)((int64_t)vbase[oc] - (int64_t)vsub + (int64_t)__popc((uint32_t)created[oc] & before_mask(r.e)));
	}
}

MeshGrid make_grid(const lfa_mesher *m) {
	return MeshGrid{m->n[0], m->n[1], m->n[2], m->off[0], m->off[1], m->off[2], m->cs, m->extent, m->radius,
	                m->z0, m->nzl, m->s_lo, m->s_hi, m->c_lo, m->c_hi, m->own_lo};
}
__global__ void k_rebase(uint64_t *idx, size_t n, int64_t base) {
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) idx[i] = (uint64_t)((int64_t)idx[i] + base);
}

int scan_u32(lfa_mesher *m, const uint32_t *in, uint32_t *out, size_t n) {
	const size_t nblk = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
	hipLaunchKernelGGL(k_block_sums, dim3((unsigned)nblk), dim3(256), 0, m->stream, in, n, m->blk);
	hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, m->stream, m->blk, nblk);
	hipLaunchKernelGGL(k_block_scan_apply, dim3((unsigned)nblk), dim3(256), 0, m->stream, in, out, n, (const uint32_t *)m->blk, nblk);
	MSH_HIP(m, hipGetLastError());
	return LFA_OK;
}
}  // namespace

// ================================================================================================= C ABI
extern "C" int lfa_mesher_create_window(lfa_mesher **out, const uint64_t size[3], const double grid_offset[3], double cell_size,
                                        double particle_extent, uint64_t cell_radius, uint64_t zlo, uint64_t zhi, int device) {
	if (!out || !size || !grid_offset) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create: NULL argument");
	*out = nullptr;
	if (!(cell_size > 0.0) || !(particle_extent > 0.0) || cell_radius < 1 || cell_radius > 64)
		return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create: cell_size, particle_extent must be positive, cell_radius in [1, 64]");
	for (int d = 0; d < 3; ++d)
		if (size[d] == 0 || size[d] > 4096) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create: grid size out of range");
	if (zlo >= zhi || zhi > size[2]) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create_window: need 0 <= zlo < zhi <= size[2]");
	// storage window: the planes to sample are those of the own layers and of the layer below (whose vertices the own cells
	// share), the cells to hash are the ones a sampled point can see (cell_radius around it)
	const uint64_t c_lo = zlo > 0 ? zlo - 1 : 0, s_lo = c_lo, s_hi = zhi;
	const uint64_t z0 = s_lo > cell_radius ? s_lo - cell_radius : 0, z1 = s_hi + cell_radius < size[2] ? s_hi + cell_radius : size[2];
	const uint64_t nzl = z1 - z0;
	if ((size[0] + 1) * (size[1] + 1) * (nzl + 1) >= (1ull << 32))
		return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create: more than 2^32 sample points");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return lfa_fail(nullptr, LFA_E_NO_DEVICE, "no HIP device available (libfluid_amd has no CPU fallback)");
	if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
	if (device >= ndev || hipSetDevice(device) != hipSuccess) return lfa_fail(nullptr, LFA_E_NO_DEVICE, "device %d unusable", device);
	lfa_mesher *m = new lfa_mesher();
	m->device = device;
	for (int d = 0; d < 3; ++d) {
		m->n[d] = size[d];
		m->off[d] = grid_offset[d];
	}
	m->cs = cell_size;
	m->extent = particle_extent;
	m->radius = cell_radius;
	m->z0 = z0; m->nzl = nzl; m->s_lo = s_lo; m->s_hi = s_hi; m->c_lo = c_lo; m->c_hi = zhi; m->own_lo = zlo;
	m->ncell = (size_t)size[0] * size[1] * nzl;
	m->npts = (size_t)(size[0] + 1) * (size[1] + 1) * (nzl + 1);
	m->nblk = (m->ncell + SCAN_BLOCK - 1) / SCAN_BLOCK;
	bool ok = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) == hipSuccess &&
	          hipMalloc(&m->values, m->npts * 8) == hipSuccess && hipMalloc(&m->cell_start, (m->ncell + 1) * 4) == hipSuccess &&
	          hipMalloc(&m->cell_fill, m->ncell * 4) == hipSuccess && hipMalloc(&m->vcount, (m->ncell + 1) * 4) == hipSuccess &&
	          hipMalloc(&m->icount, (m->ncell + 1) * 4) == hipSuccess && hipMalloc(&m->created, m->ncell * 2) == hipSuccess &&
	          hipMalloc(&m->occ, m->ncell) == hipSuccess && hipMalloc(&m->blk, (m->nblk + 1) * 4) == hipSuccess;
	// grid3<double>(size + 1): zero-initialised until the first sampling (src/mesher.cpp:321)
	ok = ok && hipMemsetAsync(m->values, 0, m->npts * 8, m->stream) == hipSuccess &&
	     hipMemcpyToSymbolAsync(HIP_SYMBOL(d_tri_table), MC_TRIANGLES, 256 * 16, 0, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
	     hipMemcpyToSymbolAsync(HIP_SYMBOL(d_edge_corners), MC_EDGE_CORNERS, 24, 0, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
	     hipMemcpyToSymbolAsync(HIP_SYMBOL(d_corner_offsets), MC_CORNER_OFFSETS, 24, 0, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
	     hipStreamSynchronize(m->stream) == hipSuccess;
	if (!ok) {
		lfa_mesher_destroy(m);
		return lfa_fail(nullptr, LFA_E_OOM, "lfa_mesher_create: device allocation failed");
	}
	*out = m;
	return LFA_OK;
}

extern "C" int lfa_mesher_create(lfa_mesher **out, const uint64_t size[3], const double grid_offset[3], double cell_size,
                                 double particle_extent, uint64_t cell_radius, int device) {
	if (!size) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_mesher_create: NULL argument");
	return lfa_mesher_create_window(out, size, grid_offset, cell_size, particle_extent, cell_radius, 0, size[2], device);
}

extern "C" int lfa_mesher_window(const lfa_mesher *m, uint64_t *z0, uint64_t *n_planes, uint64_t *own_lo, uint64_t *own_hi) {
	if (!m) return LFA_E_INVALID;
	if (z0) *z0 = m->z0;
	if (n_planes) *n_planes = m->nzl + 1;
	if (own_lo) *own_lo = m->own_lo;
	if (own_hi) *own_hi = m->c_hi;
	return LFA_OK;
}

extern "C" void lfa_mesher_destroy(lfa_mesher *m) {
	if (!m) return;
	(void)hipSetDevice(m->device);
	if (m->stream) (void)hipStreamSynchronize(m->stream);
	void *ptrs[] = {m->ids, m->values, m->cell_start, m->cell_fill, m->order, m->pos, m->spos, m->blk_flag, m->vcount, m->icount, m->created, m->occ, m->blk,
	                m->vpos, m->vidx};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (m->stream) (void)hipStreamDestroy(m->stream);
	delete m;
}

extern "C" const char *lfa_mesher_last_error(const lfa_mesher *m) { return m ? m->err.c_str() : lfa_last_error(nullptr); }

static int sample_device_positions(lfa_mesher *m, const double *dpos, uint64_t n, double r, const uint32_t *dids = nullptr) {
	const MeshGrid g = make_grid(m);
	MSH_HIP(m, hipMemsetAsync(m->cell_start, 0, (m->ncell + 1) * 4, m->stream));
	MSH_HIP(m, hipMemsetAsync(m->cell_fill, 0, m->ncell * 4, m->stream));
	const size_t nbf = (size_t)((m->n[0] + 7) >> 3) * ((m->n[1] + 7) >> 3) * ((m->nzl + 7) >> 3);
	if (nbf > m->n_blk_flag) {
		if (m->blk_flag) MSH_HIP(m, hipFree(m->blk_flag));
		m->blk_flag = nullptr;
		MSH_HIP(m, hipMalloc(&m->blk_flag, nbf));
		m->n_blk_flag = nbf;
	}
	MSH_HIP(m, hipMemsetAsync(m->blk_flag, 0, nbf, m->stream));
	if (n) {
		hipLaunchKernelGGL(k_count_particles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, g, dpos, (size_t)n,
		                   m->cell_start, m->blk_flag);
		MSH_HIP(m, hipGetLastError());
	}
	int rc = scan_u32(m, m->cell_start, m->cell_start, m->ncell);
	if (rc != LFA_OK) return rc;
	if (n) {
		hipLaunchKernelGGL(k_scatter_particles, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, g, dpos, (size_t)n,
		                   (const uint32_t *)m->cell_start, m->cell_fill, m->order);
		hipLaunchKernelGGL(k_sort_groups, dim3((unsigned)((m->ncell + 255) / 256)), dim3(256), 0, m->stream,
		                   (const uint32_t *)m->cell_start, m->ncell, m->order, dids);
		hipLaunchKernelGGL(k_gather_positions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, dpos,
		                   (const uint32_t *)m->order, (const uint32_t *)m->cell_start, m->ncell, m->spos);
	}
	hipLaunchKernelGGL(k_sample_surface, dim3((unsigned)((m->npts + 255) / 256)), dim3(256), 0, m->stream, g, (const double *)m->spos,
	                   (const uint32_t *)m->cell_start, (const uint8_t *)m->blk_flag, r, m->values);
	MSH_HIP(m, hipGetLastError());
	m->have_mesh = false;
	return LFA_OK;
}

static int ensure_particle_capacity(lfa_mesher *m, size_t n) {
	if (n <= m->pcap) return LFA_OK;
	void *old[] = {m->pos, m->order, m->spos, m->ids};
	for (void *q : old)
		if (q) MSH_HIP(m, hipFree(q));
	m->pos = nullptr; m->order = nullptr; m->spos = nullptr; m->ids = nullptr;
	m->pcap = 0;
	MSH_HIP(m, hipMalloc(&m->pos, n * 24));
	MSH_HIP(m, hipMalloc(&m->spos, n * 24));
	MSH_HIP(m, hipMalloc(&m->order, n * 4));
	MSH_HIP(m, hipMalloc(&m->ids, n * 4));
	m->pcap = n;
	return LFA_OK;
}

extern "C" int lfa_mesher_sample_ids(lfa_mesher *m, const double *positions, const uint32_t *ids, uint64_t n, double r) {
	if (!m || (!positions && n)) return LFA_E_INVALID;
	if (n >= (1ull << 32)) return mfail(m, LFA_E_INVALID, "lfa_mesher_sample: more than 2^32 particles");
	MSH_HIP(m, hipSetDevice(m->device));
	int rc = ensure_particle_capacity(m, (size_t)n);
	if (rc != LFA_OK) return rc;
	if (n) MSH_HIP(m, hipMemcpyAsync(m->pos, positions, (size_t)n * 24, hipMemcpyHostToDevice, m->stream));
	if (n && ids) MSH_HIP(m, hipMemcpyAsync(m->ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
	rc = sample_device_positions(m, m->pos, n, r, ids ? (const uint32_t *)m->ids : (const uint32_t *)nullptr);
	if (rc != LFA_OK) return rc;
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	return LFA_OK;
}
extern "C" int lfa_mesher_sample(lfa_mesher *m, const double *positions, uint64_t n, double r) {
	return lfa_mesher_sample_ids(m, positions, nullptr, n, r);
}

/// World positions of a simulation's resident particles, in upload order: the doubles lfa_download_particles writes with
/// LFA_DL_POSITIONS (core.hip:k_export), so meshing from the device equals meshing the downloaded particles.
__global__ void k_sim_positions(ParticleSoA p, size_t first, size_t n, size_t out_at, int by_slot, GridDims g, double ox, double oy,
                                double oz, double h, double *pos, uint32_t *ids) {
	const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	const size_t i = first + k;
	const uint32_t b = p.key[i];
	const int tile = (int)(b >> 9), l = (int)(b & 511);
	int tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	const int c[3] = {tx * 8 + (l & 7), ty * 8 + ((l >> 3) & 7), tz * 8 + (l >> 6)};
	const double off[3] = {ox, oy, oz};
	// single domain: element id of the array = the particle's place in the host's array (what a download writes); slabs: storage
	// order, with the global id as the order key inside a cell
	const size_t at = by_slot ? out_at + k : (size_t)p.id[i];
	double *q = pos + 3 * at;
#pragma unroll
	for (int d = 0; d < 3; ++d) q[d] = off[d] + ((double)c[d] + (double)p.t[d][i]) * h;
	if (by_slot) ids[at] = p.id[i];
}

extern "C" int lfa_mesher_sample_sim(lfa_mesher *m, lfa_sim *s, double r) {
	if (!m || !s) return LFA_E_INVALID;
	if (s->device != m->device) return mfail(m, LFA_E_INVALID, "lfa_mesher_sample_sim: handles live on different devices");
	MSH_HIP(m, hipSetDevice(m->device));
	if (lfa_corr_join(s) < 0) return mfail(m, LFA_E_HIP, "lfa_mesher_sample_sim: joining the simulation's position correction failed");
	size_t n = s->np, n_live = s->np;
	if (s->dist) {
		// a rank of a slab run: its own particles and the ghost copies of the neighbours' adjacent tile layers (8 cells each side);
		// the window of `m` must only need particles from there
		if (!s->binned || s->holes) return mfail(m, LFA_E_INVALID, "lfa_mesher_sample_sim: slab decomposition: call lfa_hash_particles first");
		const double h = s->prm.cell_size, lo = s->prm.grid_offset[2] + h * 8.0 * (double)(lfa_has_lo(s) ? s->slab_lo - 1 : 0),
		             hi = s->prm.grid_offset[2] + h * 8.0 * (double)(lfa_has_hi(s) ? s->slab_hi + 1 : s->g.ntz);
		const double need_lo = m->off[2] + m->cs * ((double)m->s_lo - (double)m->radius), need_hi = m->off[2] + m->cs * ((double)m->s_hi + (double)m->radius);
		if ((lfa_has_lo(s) && need_lo < lo) || (lfa_has_hi(s) && need_hi > hi))
			return mfail(m, LFA_E_INVALID, "lfa_mesher_sample_sim: the mesher window reaches beyond this rank's slab and ghost tile layers");
		if (lfa_particles_materialize(s) != LFA_OK || lfa_dist_exchange_ghost_particles(s) != LFA_OK)
			return mfail(m, LFA_E_HIP, lfa_last_error(s));
		n_live = s->np_live;
		n = n_live + s->n_ghost_particles;
	}
	if (n >= (1ull << 32)) return mfail(m, LFA_E_INVALID, "lfa_mesher_sample_sim: more than 2^32 particles");
	int rc = ensure_particle_capacity(m, n);
	if (rc != LFA_OK) return rc;
	MSH_HIP(m, hipStreamSynchronize(s->stream));
	if (n) {
		// live particles [0, n_live), ghosts behind them (lfa_dist_exchange_ghost_particles): one launch over the lot
		hipLaunchKernelGGL(k_sim_positions, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, s->pb[s->cur], (size_t)0, n,
		                   (size_t)0, s->dist ? 1 : 0, s->g, s->prm.grid_offset[0], s->prm.grid_offset[1], s->prm.grid_offset[2],
		                   s->prm.cell_size, m->pos, m->ids);
		MSH_HIP(m, hipGetLastError());
	}
	rc = sample_device_positions(m, m->pos, n, r, s->dist ? (const uint32_t *)m->ids : (const uint32_t *)nullptr);
	if (rc != LFA_OK) return rc;
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	return LFA_OK;
}

extern "C" int lfa_mesher_download_values(lfa_mesher *m, double *values) {
	if (!m || !values) return LFA_E_INVALID;
	MSH_HIP(m, hipSetDevice(m->device));
	MSH_HIP(m, hipMemcpyAsync(values, m->values, m->npts * 8, hipMemcpyDeviceToHost, m->stream));
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	return LFA_OK;
}

extern "C" int lfa_mesher_upload_values(lfa_mesher *m, const double *values) {
	if (!m || !values) return LFA_E_INVALID;
	MSH_HIP(m, hipSetDevice(m->device));
	MSH_HIP(m, hipMemcpyAsync(m->values, values, m->npts * 8, hipMemcpyHostToDevice, m->stream));
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	m->have_mesh = false;
	return LFA_OK;
}

extern "C" int lfa_mesher_marching_cubes(lfa_mesher *m, uint64_t *n_vertices, uint64_t *n_indices) {
	if (!m) return LFA_E_INVALID;
	MSH_HIP(m, hipSetDevice(m->device));
	const MeshGrid g = make_grid(m);
	const unsigned grid = (unsigned)((m->ncell + 255) / 256);
	hipLaunchKernelGGL(k_mc_classify, dim3(grid), dim3(256), 0, m->stream, g, (const double *)m->values, m->occ, m->created,
	                   m->vcount, m->icount);
	MSH_HIP(m, hipGetLastError());
	int rc = scan_u32(m, m->vcount, m->vcount, m->ncell);
	if (rc == LFA_OK) rc = scan_u32(m, m->icount, m->icount, m->ncell);
	if (rc != LFA_OK) return rc;
	uint32_t tot[3] = {0, 0, 0};
	const size_t first_own = (size_t)m->n[0] * m->n[1] * (m->own_lo - m->z0);
	MSH_HIP(m, hipMemcpyAsync(&tot[0], m->vcount + m->ncell, 4, hipMemcpyDeviceToHost, m->stream));
	MSH_HIP(m, hipMemcpyAsync(&tot[1], m->icount + m->ncell, 4, hipMemcpyDeviceToHost, m->stream));
	MSH_HIP(m, hipMemcpyAsync(&tot[2], m->vcount + first_own, 4, hipMemcpyDeviceToHost, m->stream));  // vertices of the layer below
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	const uint32_t vsub = tot[2], nv = tot[0] - tot[2];
	if (nv > m->vcap) {
		if (m->vpos) MSH_HIP(m, hipFree(m->vpos));
		m->vpos = nullptr;
		m->vcap = 0;
		MSH_HIP(m, hipMalloc(&m->vpos, (size_t)nv * 24));
		m->vcap = nv;
	}
	if (tot[1] > m->icap) {
		if (m->vidx) MSH_HIP(m, hipFree(m->vidx));
		m->vidx = nullptr;
		m->icap = 0;
		MSH_HIP(m, hipMalloc(&m->vidx, (size_t)tot[1] * 8));
		m->icap = tot[1];
	}
	if (tot[0]) {
		hipLaunchKernelGGL(k_mc_vertices, dim3(grid), dim3(256), 0, m->stream, g, (const double *)m->values,
		                   (const uint16_t *)m->created, (const uint32_t *)m->vcount, m->vpos, vsub);
		hipLaunchKernelGGL(k_mc_triangles, dim3(grid), dim3(256), 0, m->stream, g, (const uint8_t *)m->occ,
		                   (const uint16_t *)m->created, (const uint32_t *)m->vcount, (const uint32_t *)m->icount, m->vidx, vsub);
		MSH_HIP(m, hipGetLastError());
		MSH_HIP(m, hipStreamSynchronize(m->stream));
	}
	m->n_vertices = nv;
	m->n_indices = tot[1];
	m->have_mesh = true;
	if (n_vertices) *n_vertices = nv;
	if (n_indices) *n_indices = tot[1];
	return LFA_OK;
}

extern "C" int lfa_mesher_rebase(lfa_mesher *m, uint64_t vertices_below) {
	if (!m) return LFA_E_INVALID;
	if (!m->have_mesh) return mfail(m, LFA_E_INVALID, "lfa_mesher_rebase: call lfa_mesher_marching_cubes first");
	MSH_HIP(m, hipSetDevice(m->device));
	if (m->n_indices && vertices_below) {
		hipLaunchKernelGGL(k_rebase, dim3((unsigned)((m->n_indices + 255) / 256)), dim3(256), 0, m->stream, m->vidx, (size_t)m->n_indices,
		                   (int64_t)vertices_below);
		MSH_HIP(m, hipGetLastError());
		MSH_HIP(m, hipStreamSynchronize(m->stream));
	}
	return LFA_OK;
}

extern "C" int lfa_mesher_download_mesh(lfa_mesher *m, double *positions, uint64_t *indices) {
	if (!m) return LFA_E_INVALID;
	if (!m->have_mesh) return mfail(m, LFA_E_INVALID, "lfa_mesher_download_mesh: call lfa_mesher_marching_cubes first");
	MSH_HIP(m, hipSetDevice(m->device));
	if (positions && m->n_vertices)
		MSH_HIP(m, hipMemcpyAsync(positions, m->vpos, (size_t)m->n_vertices * 24, hipMemcpyDeviceToHost, m->stream));
	if (indices && m->n_indices)
		MSH_HIP(m, hipMemcpyAsync(indices, m->vidx, (size_t)m->n_indices * 8, hipMemcpyDeviceToHost, m->stream));
	MSH_HIP(m, hipStreamSynchronize(m->stream));
	return LFA_OK;
}
