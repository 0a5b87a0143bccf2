// libfluid_amd/csrc/voxelizer.hip -- solid-boundary voxelizer on the device.
//
// Reference: fluid::voxelizer (include/fluid/voxelizer.h:14-74, src/voxelizer.cpp:12-136), the box/triangle separating
// axis test it calls (src/math/intersection.cpp:31-82, Akenine-Moller) and the two hosts of it, fluid::obstacle
// (src/data_structures/obstacle.cpp:9-29) and the Maya VoxelizerNode (plugins/maya/nodes/voxelizer_node.cpp:255-343).
//
// Parity contract: cell classification is bit-exact. Everything that decides a cell's type is evaluated in fp64 with
// the reference's own operation order (-ffp-contract=off), including the cell centres, which the reference accumulates
// by repeated `+= cell_size` from the first cell of the triangle's bounding box (src/voxelizer.cpp:70-75): the wave
// that owns a triangle reproduces those three running sums in LDS before it tests the cells.
//   surface  : one wave per triangle, lanes stride over the cells of the triangle's bounding box; marking is an
//              idempotent byte store, so overlapping triangles need no atomics.
//   exterior : the reference's stack flood fill from voxel (0,0,0) (src/voxelizer.cpp:83-124) computes the 6-connected
//              component of non-surface cells that contains the corner; here 8^3 blocks relax to their local fixed point
//              in LDS and the launch is repeated until no block changes -- the same set, any order.
//   lists    : ordered stream compaction in raw (x fastest) order = grid3::for_each / for_each_in_range order.
#include "common.h"

#include <math.h>
#include <string.h>

#include <algorithm>

struct lfa_voxels {
	int device = 0;
	hipStream_t stream = nullptr;
	uint64_t n[3] = {0, 0, 0};
	size_t nc = 0;
	int32_t grid_min[3] = {0, 0, 0};  // offset of the voxel grid in the reference grid (resize_reposition_grid_constrained)
	double off[3] = {0, 0, 0}, cell_size = 1.0;
	uint8_t *vox = nullptr;
	uint32_t *blk = nullptr;  // per-block counts / offsets of the compaction
	size_t nblk = 0;
	int *flag = nullptr;
	int32_t *out = nullptr;
	size_t out_cap = 0;
	std::string err;
};

namespace {
constexpr int VOX_CH = 128;       // cells of one axis whose centres sit in LDS at a time
constexpr int COMPACT_BLOCK = 2048;  // cells per workgroup of the compaction

int vfail(lfa_voxels *v, int code, const char *msg) {
	if (v) v->err = msg;
	return lfa_fail(nullptr, code, "%s", msg);
}
#define VOX_HIP(v, call)                                                                           \
	do {                                                                                            \
		hipError_t e_ = (call);                                                                     \
		if (e_ != hipSuccess) return vfail((v), e_ == hipErrorOutOfMemory ? LFA_E_OOM : LFA_E_HIP, \
		                                   (std::string(#call) + ": " + hipGetErrorString(e_)).c_str()); \
	} while (0)

struct D3 {
	double x, y, z;
};
__device__ inline D3 sub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
/// vec_ops::dot (include/fluid/math/vec.h:110-122): result{} += a_i * b_i in component order.
__device__ inline double dot3(D3 a, D3 b) {
	double r = 0.0;
	r += a.x * b.x;
	r += a.y * b.y;
	r += a.z * b.z;
	return r;
}

/// aab_triangle_overlap_bounded_center (src/math/intersection.cpp:31-82): box centred at the origin.
__device__ inline bool tri_box_overlap(double hx, double hy, double hz, D3 p1, D3 p2, D3 p3) {
	const D3 f[3] = {sub(p2, p1), sub(p3, p2), sub(p1, p3)};
	const D3 nrm = {f[0].y * f[1].z - f[0].z * f[1].y, f[0].z * f[1].x - f[0].x * f[1].z, f[0].x * f[1].y - f[0].y * f[1].x};
	const double center_off = dot3(p1, nrm);
	const double radius_n = dot3(D3{fabs(nrm.x), fabs(nrm.y), fabs(nrm.z)}, D3{hx, hy, hz});
	if (fabs(center_off) > fabs(radius_n)) return false;
	const D3 v[3] = {p1, p2, p3};
	// (1, 0, 0) x f: p = v.z * f.y - v.y * f.z ; r = h.y |f.z| + h.z |f.y|
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const D3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		const double p0 = v1.z * fi.y - v1.y * fi.z, p1 = v2.z * fi.y - v2.y * fi.z;
		const double pmin = p1 < p0 ? p1 : p0, pmax = p1 < p0 ? p0 : p1;  // std::minmax(p0, p1)
		const double r = hy * fabs(fi.z) + hz * fabs(fi.y);
		if (pmin > r || pmax < -r) return false;
	}
	// (0, 1, 0) x f: p = v.x * f.z - v.z * f.x ; r = h.x |f.z| + h.z |f.x|
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const D3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		const double p0 = v1.x * fi.z - v1.z * fi.x, p1 = v2.x * fi.z - v2.z * fi.x;
		const double pmin = p1 < p0 ? p1 : p0, pmax = p1 < p0 ? p0 : p1;
		const double r = hx * fabs(fi.z) + hz * fabs(fi.x);
		if (pmin > r || pmax < -r) return false;
	}
	// (0, 0, 1) x f: p = v.y * f.x - v.x * f.y ; r = h.x |f.y| + h.y |f.x|
#pragma unroll
	for (int i = 0; i < 3; ++i) {
		const D3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		const double p0 = v1.y * fi.x - v1.x * fi.y, p1 = v2.y * fi.x - v2.x * fi.y;
		const double pmin = p1 < p0 ? p1 : p0, pmax = p1 < p0 ? p0 : p1;
		const double r = hx * fabs(fi.y) + hy * fabs(fi.x);
		if (pmin > r || pmax < -r) return false;
	}
	return true;
}

struct VoxGrid {
	uint64_t nx, ny, nz;
	double ox, oy, oz, cs;
};

/// voxelizer::voxelize_triangle (src/voxelizer.cpp:54-81), one wave per triangle.
template <typename Index>
__global__ void __launch_bounds__(256)
k_voxelize_triangles(VoxGrid g, const double *pos, const Index *idx, size_t n_tri, uint8_t *vox) {
	__shared__ double seq[4][3][VOX_CH];
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const size_t t = (size_t)blockIdx.x * 4 + wid;
	if (t >= n_tri) return;
	D3 p[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		const size_t vi = (size_t)idx[3 * t + k];
		p[k] = D3{pos[3 * vi], pos[3 * vi + 1], pos[3 * vi + 2]};
	}
	const double cs = g.cs, half = 0.5 * cs;
	const double off[3] = {g.ox, g.oy, g.oz};
	const uint64_t gn[3] = {g.nx, g.ny, g.nz};
	uint64_t lo[3], cnt[3];
	double c0[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const double a = d == 0 ? p[0].x : (d == 1 ? p[0].y : p[0].z), b = d == 0 ? p[1].x : (d == 1 ? p[1].y : p[1].z),
		             c = d == 0 ? p[2].x : (d == 1 ? p[2].y : p[2].z);
		const double mn = fmin(fmin(a, b), c), mx = fmax(fmax(a, b), c);
		// vec3s((min - grid_offset) / cell_size): truncation; the reference assumes the triangle lies inside the grid
		// (negative values are undefined behaviour there), here out-of-range indices are clamped to the grid
		const double qlo = (mn - off[d]) / cs, qhi = (mx - off[d]) / cs;
		const uint64_t ilo = qlo > 0.0 ? (uint64_t)qlo : 0, ihi = qhi > 0.0 ? (uint64_t)qhi : 0;
		lo[d] = ilo < gn[d] ? ilo : gn[d] - 1;
		const uint64_t hi = ihi < gn[d] ? ihi : gn[d] - 1;
		cnt[d] = hi >= lo[d] ? hi - lo[d] + 1 : 0;
		c0[d] = off[d] + (double)lo[d] * cs + half;  // min_center
	}
	double (*S)[VOX_CH] = seq[wid];
	// running sums of the cell centres, chunk by chunk: the value at the start of a chunk continues the sum of the
	// previous one (center += cell_size, src/voxelizer.cpp:70-75); x restarts for every y, y for every z chunk.
	double zc = c0[2];
	for (uint64_t z0 = 0; z0 < cnt[2]; z0 += VOX_CH) {
		const int nzc = (int)(cnt[2] - z0 < VOX_CH ? cnt[2] - z0 : VOX_CH);
		if (lane == 0) {
			double c = zc;
			for (int k = 0; k < nzc; ++k) { S[2][k] = c; c += cs; }
		}
		double yc = c0[1];
		for (uint64_t y0 = 0; y0 < cnt[1]; y0 += VOX_CH) {
			const int nyc = (int)(cnt[1] - y0 < VOX_CH ? cnt[1] - y0 : VOX_CH);
			if (lane == 1) {
				double c = yc;
				for (int k = 0; k < nyc; ++k) { S[1][k] = c; c += cs; }
			}
			double xc = c0[0];
			for (uint64_t x0 = 0; x0 < cnt[0]; x0 += VOX_CH) {
				const int nxc = (int)(cnt[0] - x0 < VOX_CH ? cnt[0] - x0 : VOX_CH);
				if (lane == 2) {
					double c = xc;
					for (int k = 0; k < nxc; ++k) { S[0][k] = c; c += cs; }
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				__builtin_amdgcn_wave_barrier();
				const int total = nxc * nyc * nzc;
				for (int i = lane; i < total; i += 64) {
					const int ix = i % nxc, iy = (i / nxc) % nyc, iz = i / (nxc * nyc);
					const D3 c = {S[0][ix], S[1][iy], S[2][iz]};
					const size_t cell = (size_t)(lo[0] + x0 + ix) + g.nx * ((size_t)(lo[1] + y0 + iy) + g.ny * (size_t)(lo[2] + z0 + iz));
					if (vox[cell] != LFA_VOX_SURFACE &&
					    tri_box_overlap(half, half, half, sub(p[0], c), sub(p[1], c), sub(p[2], c)))
						vox[cell] = LFA_VOX_SURFACE;
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				__builtin_amdgcn_wave_barrier();
				// next chunk starts where this one's sum ended
				for (int k = 0; k < nxc; ++k) xc += cs;
			}
			for (int k = 0; k < nyc; ++k) yc += cs;
		}
		for (int k = 0; k < nzc; ++k) zc += cs;
	}
}

/// One relaxation pass of the exterior flood fill: every 8^3 block (one-cell halo) runs to its local fixed point.
__global__ void __launch_bounds__(512)
k_flood_pass(uint8_t *vox, uint64_t nx, uint64_t ny, uint64_t nz, int *changed) {
	__shared__ uint8_t t[10][10][10];
	__shared__ int any_ext, any_int;
	const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
	const int64_t bx = (int64_t)blockIdx.x * 8, by = (int64_t)blockIdx.y * 8, bz = (int64_t)blockIdx.z * 8;
	if (threadIdx.x == 0) { any_ext = 0; any_int = 0; }
	for (int i = threadIdx.x; i < 1000; i += 512) {
		const int hx = i % 10, hy = (i / 10) % 10, hz = i / 100;
		const int64_t x = bx + hx - 1, y = by + hy - 1, z = bz + hz - 1;
		uint8_t v = LFA_VOX_SURFACE;  // outside the grid: a wall
		if (x >= 0 && y >= 0 && z >= 0 && (uint64_t)x < nx && (uint64_t)y < ny && (uint64_t)z < nz)
			v = vox[(size_t)x + nx * ((size_t)y + ny * (size_t)z)];
		t[hz][hy][hx] = v;
	}
	__syncthreads();
	const int64_t x = bx + lx, y = by + ly, z = bz + lz;
	const bool inside = (uint64_t)x < nx && (uint64_t)y < ny && (uint64_t)z < nz;
	uint8_t mine = t[lz + 1][ly + 1][lx + 1];
	const uint8_t before = mine;
	if (inside && mine == LFA_VOX_INTERIOR) any_int = 1;
	// an exterior cell anywhere in the halo block can start a front
	for (int i = threadIdx.x; i < 1000; i += 512)
		if (t[i / 100][(i / 10) % 10][i % 10] == LFA_VOX_EXTERIOR) any_ext = 1;
	__syncthreads();
	if (!any_ext || !any_int) return;
	for (;;) {
		bool ch = false;
		if (inside && mine == LFA_VOX_INTERIOR) {
			if (t[lz + 1][ly + 1][lx] == LFA_VOX_EXTERIOR || t[lz + 1][ly + 1][lx + 2] == LFA_VOX_EXTERIOR ||
			    t[lz + 1][ly][lx + 1] == LFA_VOX_EXTERIOR || t[lz + 1][ly + 2][lx + 1] == LFA_VOX_EXTERIOR ||
			    t[lz][ly + 1][lx + 1] == LFA_VOX_EXTERIOR || t[lz + 2][ly + 1][lx + 1] == LFA_VOX_EXTERIOR) {
				mine = LFA_VOX_EXTERIOR;
				ch = true;
			}
		}
		__syncthreads();
		if (ch) t[lz + 1][ly + 1][lx + 1] = mine;
		if (!__syncthreads_or(ch)) break;
	}
	if (inside && mine != before) {
		vox[(size_t)x + nx * ((size_t)y + ny * (size_t)z)] = mine;
		*changed = 1;
	}
}

__global__ void k_seed_corner(uint8_t *vox) {
	if (vox[0] != LFA_VOX_SURFACE) vox[0] = LFA_VOX_EXTERIOR;  // src/voxelizer.cpp:88-91
}

struct Select {
	int interior, surface;   // voxelizer_node.cpp:286-301
	int clip;                // keep only cells inside the reference grid (voxelizer_node.cpp:325-343, obstacle.cpp:20-28)
	int64_t gmin[3], ref[3];
};
__device__ inline bool selected(const Select &s, uint8_t type, uint64_t x, uint64_t y, uint64_t z) {
	if (!((type == LFA_VOX_INTERIOR && s.interior) || (type == LFA_VOX_SURFACE && s.surface))) return false;
	if (!s.clip) return true;
	const int64_t rx = (int64_t)x + s.gmin[0], ry = (int64_t)y + s.gmin[1], rz = (int64_t)z + s.gmin[2];
	return rx >= 0 && rx < s.ref[0] && ry >= 0 && ry < s.ref[1] && rz >= 0 && rz < s.ref[2];
}

__global__ void __launch_bounds__(256)
k_select_count(const uint8_t *vox, uint64_t nx, uint64_t ny, size_t nc, Select s, uint32_t *blk) {
	__shared__ uint32_t red[4];
	const size_t b0 = (size_t)blockIdx.x * COMPACT_BLOCK;
	uint32_t c = 0;
	for (int k = 0; k < COMPACT_BLOCK / 256; ++k) {
		const size_t r = b0 + (size_t)k * 256 + threadIdx.x;
		if (r < nc) c += selected(s, vox[r], r % nx, (r / nx) % ny, r / (nx * ny)) ? 1u : 0u;
	}
	c = wave_sum(c);
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) blk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

/// Exclusive scan of the per-block counts by one workgroup; total behind the last entry.
__global__ void __launch_bounds__(1024)
k_scan_blocks(uint32_t *blk, size_t n) {
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

__global__ void __launch_bounds__(256)
k_select_write(const uint8_t *vox, uint64_t nx, uint64_t ny, size_t nc, Select s, const uint32_t *blk, int32_t *out) {
	__shared__ uint32_t wave_cnt[4];
	const size_t b0 = (size_t)blockIdx.x * COMPACT_BLOCK;
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	uint32_t base = blk[blockIdx.x];
	for (int k = 0; k < COMPACT_BLOCK / 256; ++k) {
		const size_t r = b0 + (size_t)k * 256 + threadIdx.x;
		uint64_t x = 0, y = 0, z = 0;
		bool sel = false;
		if (r < nc) {
			x = r % nx; y = (r / nx) % ny; z = r / (nx * ny);
			sel = selected(s, vox[r], x, y, z);
		}
		const uint64_t m = __ballot(sel);
		const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
		__syncthreads();
		if (lane == 0) wave_cnt[wid] = (uint32_t)__popcll(m);
		__syncthreads();
		uint32_t woff = 0;
		for (int w = 0; w < wid; ++w) woff += wave_cnt[w];
		if (sel) {
			const size_t o = (size_t)base + woff + before;
			out[3 * o] = (int32_t)((int64_t)x + (s.clip ? s.gmin[0] : 0));
			out[3 * o + 1] = (int32_t)((int64_t)y + (s.clip ? s.gmin[1] : 0));
			out[3 * o + 2] = (int32_t)((int64_t)z + (s.clip ? s.gmin[2] : 0));
		}
		base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
	}
}

/// Marks the selected voxels solid in a simulation grid (the device-side form of feeding `cells_ref` to
/// grid_node.cpp:330-339 / lfa_set_solid_cells).
__global__ void k_voxels_to_solid(const uint8_t *vox, uint64_t nx, uint64_t ny, size_t nc, Select s, uint8_t *solid,
                                  uint8_t *ctype, GridDims g) {
	const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	const uint64_t x = r % nx, y = (r / nx) % ny, z = r / (nx * ny);
	if (!selected(s, vox[r], x, y, z)) return;
	const int rx = (int)((int64_t)x + s.gmin[0]), ry = (int)((int64_t)y + s.gmin[1]), rz = (int)((int64_t)z + s.gmin[2]);
	if (!in_grid(g, rx, ry, rz)) return;
	const uint32_t b = blocked_index(g, rx, ry, rz);
	solid[b] = 1;
	ctype[b] = CT_SOLID;
}

Select make_select(const lfa_voxels *v, int interior, int surface, const int64_t *ref) {
	Select s{};
	s.interior = interior;
	s.surface = surface;
	s.clip = ref ? 1 : 0;
	for (int d = 0; d < 3; ++d) {
		s.gmin[d] = v->grid_min[d];
		s.ref[d] = ref ? ref[d] : 0;
	}
	return s;
}
}  // namespace

// ================================================================================================= C ABI
extern "C" int lfa_voxels_create(lfa_voxels **out, const uint64_t size[3], const double grid_offset[3], double cell_size,
                                 int device) {
	if (!out || !size || !grid_offset) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxels_create: NULL argument");
	*out = nullptr;
	if (!(cell_size > 0.0)) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxels_create: cell_size must be positive");
	for (int d = 0; d < 3; ++d)
		if (size[d] > 8192) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxels_create: grid size out of range");
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return lfa_fail(nullptr, LFA_E_NO_DEVICE, "no HIP device available (libfluid_amd has no CPU fallback)");
	if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;
	if (device >= ndev) return lfa_fail(nullptr, LFA_E_NO_DEVICE, "device %d out of range", device);
	if (hipSetDevice(device) != hipSuccess) return lfa_fail(nullptr, LFA_E_NO_DEVICE, "hipSetDevice(%d) failed", device);
	lfa_voxels *v = new lfa_voxels();
	v->device = device;
	v->cell_size = cell_size;
	for (int d = 0; d < 3; ++d) {
		v->n[d] = size[d];
		v->off[d] = grid_offset[d];
	}
	v->nc = (size_t)size[0] * size[1] * size[2];
	v->nblk = (v->nc + COMPACT_BLOCK - 1) / COMPACT_BLOCK;
	if (hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) != hipSuccess ||
	    hipMalloc(&v->vox, v->nc ? v->nc : 1) != hipSuccess || hipMalloc(&v->blk, (v->nblk + 1) * sizeof(uint32_t)) != hipSuccess ||
	    hipMalloc(&v->flag, sizeof(int)) != hipSuccess) {
		lfa_voxels_destroy(v);
		return lfa_fail(nullptr, LFA_E_OOM, "lfa_voxels_create: device allocation failed");
	}
	// grid3<cell_type>(size, cell_type::interior) (src/voxelizer.cpp:19,38)
	if (hipMemsetAsync(v->vox, LFA_VOX_INTERIOR, v->nc ? v->nc : 1, v->stream) != hipSuccess) {
		lfa_voxels_destroy(v);
		return lfa_fail(nullptr, LFA_E_HIP, "lfa_voxels_create: memset failed");
	}
	*out = v;
	return LFA_OK;
}

extern "C" void lfa_voxels_destroy(lfa_voxels *v) {
	if (!v) return;
	(void)hipSetDevice(v->device);
	if (v->stream) (void)hipStreamSynchronize(v->stream);
	if (v->vox) (void)hipFree(v->vox);
	if (v->blk) (void)hipFree(v->blk);
	if (v->flag) (void)hipFree(v->flag);
	if (v->out) (void)hipFree(v->out);
	if (v->stream) (void)hipStreamDestroy(v->stream);
	delete v;
}

extern "C" const char *lfa_voxels_last_error(const lfa_voxels *v) { return v ? v->err.c_str() : lfa_last_error(nullptr); }

extern "C" int lfa_voxels_info(const lfa_voxels *v, int32_t grid_min[3], uint64_t size[3], double grid_offset[3],
                               double *cell_size) {
	if (!v) return LFA_E_INVALID;
	for (int d = 0; d < 3; ++d) {
		if (grid_min) grid_min[d] = v->grid_min[d];
		if (size) size[d] = v->n[d];
		if (grid_offset) grid_offset[d] = v->off[d];
	}
	if (cell_size) *cell_size = v->cell_size;
	return LFA_OK;
}

extern "C" int lfa_voxels_upload(lfa_voxels *v, const uint8_t *types) {
	if (!v || !types) return LFA_E_INVALID;
	VOX_HIP(v, hipSetDevice(v->device));
	VOX_HIP(v, hipMemcpyAsync(v->vox, types, v->nc, hipMemcpyHostToDevice, v->stream));
	VOX_HIP(v, hipStreamSynchronize(v->stream));
	return LFA_OK;
}

extern "C" int lfa_voxels_download(lfa_voxels *v, uint8_t *types) {
	if (!v || !types) return LFA_E_INVALID;
	VOX_HIP(v, hipSetDevice(v->device));
	VOX_HIP(v, hipMemcpyAsync(types, v->vox, v->nc, hipMemcpyDeviceToHost, v->stream));
	VOX_HIP(v, hipStreamSynchronize(v->stream));
	return LFA_OK;
}

extern "C" int lfa_voxels_voxelize_triangles(lfa_voxels *v, const double *positions, uint64_t n_vertices, const void *indices,
                                             int index_bytes, uint64_t n_indices) {
	if (!v || (index_bytes != 4 && index_bytes != 8)) return LFA_E_INVALID;
	const uint64_t n_tri = n_indices / 3;  // `i + 2 < indices.size()` (include/fluid/voxelizer.h:56)
	if (n_tri == 0 || v->nc == 0) return LFA_OK;
	if (!positions || !indices) return LFA_E_INVALID;
	// indices are trusted by the reference; here an out-of-range index is refused before anything is launched
	uint64_t max_index = 0;
	if (index_bytes == 4) for (uint64_t i = 0; i < 3 * n_tri; ++i) max_index = std::max<uint64_t>(max_index, ((const uint32_t *)indices)[i]);
	else for (uint64_t i = 0; i < 3 * n_tri; ++i) max_index = std::max<uint64_t>(max_index, ((const uint64_t *)indices)[i]);
	if (max_index >= n_vertices) return vfail(v, LFA_E_INVALID, "lfa_voxels_voxelize_triangles: vertex index out of range");
	VOX_HIP(v, hipSetDevice(v->device));
	double *dpos = nullptr;
	void *didx = nullptr;
	VOX_HIP(v, hipMalloc(&dpos, n_vertices * 24));
	if (hipMalloc(&didx, 3 * n_tri * (size_t)index_bytes) != hipSuccess) {
		(void)hipFree(dpos);
		return vfail(v, LFA_E_OOM, "lfa_voxels_voxelize_triangles: device allocation failed");
	}
	int rc = LFA_OK;
	if (hipMemcpyAsync(dpos, positions, n_vertices * 24, hipMemcpyHostToDevice, v->stream) != hipSuccess ||
	    hipMemcpyAsync(didx, indices, 3 * n_tri * (size_t)index_bytes, hipMemcpyHostToDevice, v->stream) != hipSuccess)
		rc = vfail(v, LFA_E_HIP, "lfa_voxels_voxelize_triangles: upload failed");
	if (rc == LFA_OK) {
		const VoxGrid g{v->n[0], v->n[1], v->n[2], v->off[0], v->off[1], v->off[2], v->cell_size};
		const unsigned grid = (unsigned)((n_tri + 3) / 4);
		if (index_bytes == 4)
			hipLaunchKernelGGL(k_voxelize_triangles<uint32_t>, dim3(grid), dim3(256), 0, v->stream, g, (const double *)dpos,
			                   (const uint32_t *)didx, (size_t)n_tri, v->vox);
		else
			hipLaunchKernelGGL(k_voxelize_triangles<uint64_t>, dim3(grid), dim3(256), 0, v->stream, g, (const double *)dpos,
			                   (const uint64_t *)didx, (size_t)n_tri, v->vox);
		if (hipGetLastError() != hipSuccess || hipStreamSynchronize(v->stream) != hipSuccess)
			rc = vfail(v, LFA_E_HIP, "lfa_voxels_voxelize_triangles: kernel failed");
	}
	(void)hipFree(dpos);
	(void)hipFree(didx);
	return rc;
}

extern "C" int lfa_voxels_mark_exterior(lfa_voxels *v) {
	if (!v) return LFA_E_INVALID;
	if (v->nc == 0) return LFA_OK;  // src/voxelizer.cpp:84-86
	VOX_HIP(v, hipSetDevice(v->device));
	hipLaunchKernelGGL(k_seed_corner, dim3(1), dim3(1), 0, v->stream, v->vox);
	const dim3 grid((unsigned)((v->n[0] + 7) / 8), (unsigned)((v->n[1] + 7) / 8), (unsigned)((v->n[2] + 7) / 8));
	const int batch = 8;
	for (int guard = 0; guard < (1 << 20); ++guard) {
		VOX_HIP(v, hipMemsetAsync(v->flag, 0, sizeof(int), v->stream));
		for (int k = 0; k < batch; ++k)
			hipLaunchKernelGGL(k_flood_pass, grid, dim3(512), 0, v->stream, v->vox, v->n[0], v->n[1], v->n[2], v->flag);
		VOX_HIP(v, hipGetLastError());
		int changed = 0;
		VOX_HIP(v, hipMemcpyAsync(&changed, v->flag, sizeof(int), hipMemcpyDeviceToHost, v->stream));
		VOX_HIP(v, hipStreamSynchronize(v->stream));
		if (!changed) return LFA_OK;
	}
	return vfail(v, LFA_E_HIP, "lfa_voxels_mark_exterior: flood fill did not reach a fixed point");
}

static int select_cells(lfa_voxels *v, int interior, int surface, const int64_t *ref, uint64_t *count, bool write) {
	VOX_HIP(v, hipSetDevice(v->device));
	*count = 0;
	if (v->nc == 0) return LFA_OK;
	const Select s = make_select(v, interior, surface, ref);
	hipLaunchKernelGGL(k_select_count, dim3((unsigned)v->nblk), dim3(256), 0, v->stream, (const uint8_t *)v->vox, v->n[0], v->n[1],
	                   v->nc, s, v->blk);
	hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, v->stream, v->blk, v->nblk);
	VOX_HIP(v, hipGetLastError());
	uint32_t total = 0;
	VOX_HIP(v, hipMemcpyAsync(&total, v->blk + v->nblk, 4, hipMemcpyDeviceToHost, v->stream));
	VOX_HIP(v, hipStreamSynchronize(v->stream));
	*count = total;
	if (!write || total == 0) return LFA_OK;
	if (v->out_cap < total) {
		if (v->out) VOX_HIP(v, hipFree(v->out));
		v->out = nullptr;
		v->out_cap = 0;
		VOX_HIP(v, hipMalloc(&v->out, (size_t)total * 12));
		v->out_cap = total;
	}
	hipLaunchKernelGGL(k_select_write, dim3((unsigned)v->nblk), dim3(256), 0, v->stream, (const uint8_t *)v->vox, v->n[0], v->n[1],
	                   v->nc, s, (const uint32_t *)v->blk, v->out);
	VOX_HIP(v, hipGetLastError());
	return LFA_OK;
}

extern "C" int lfa_voxels_count(lfa_voxels *v, int include_interior, int include_surface, const int64_t *ref_grid_size,
                                uint64_t *count) {
	if (!v || !count) return LFA_E_INVALID;
	return select_cells(v, include_interior, include_surface, ref_grid_size, count, false);
}

extern "C" int lfa_voxels_cells(lfa_voxels *v, int include_interior, int include_surface, const int64_t *ref_grid_size,
                                int32_t *xyz, uint64_t capacity, uint64_t *count) {
	if (!v || !count) return LFA_E_INVALID;
	int rc = select_cells(v, include_interior, include_surface, ref_grid_size, count, true);
	if (rc != LFA_OK || *count == 0) return rc;
	if (!xyz || capacity < *count) return vfail(v, LFA_E_INVALID, "lfa_voxels_cells: output buffer too small");
	VOX_HIP(v, hipMemcpyAsync(xyz, v->out, (size_t)*count * 12, hipMemcpyDeviceToHost, v->stream));
	VOX_HIP(v, hipStreamSynchronize(v->stream));
	return LFA_OK;
}

extern "C" int lfa_voxelize_mesh(lfa_voxels **out, const double *positions, uint64_t n_vertices, const void *indices,
                                 int index_bytes, uint64_t n_indices, double cell_size, const double ref_grid_offset[3],
                                 int device) {
	if (!out || !ref_grid_offset || (!positions && n_vertices)) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxelize_mesh: NULL argument");
	*out = nullptr;
	if (!(cell_size > 0.0)) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxelize_mesh: cell_size must be positive");
	// voxelizer::get_bounding_box (include/fluid/voxelizer.h:24-34): vec3d() for an empty vertex list
	double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
	if (n_vertices) {
		for (int d = 0; d < 3; ++d) mn[d] = mx[d] = positions[d];
		for (uint64_t i = 1; i < n_vertices; ++i)
			for (int d = 0; d < 3; ++d) {
				mn[d] = std::min(mn[d], positions[3 * i + d]);
				mx[d] = std::max(mx[d], positions[3 * i + d]);
			}
	}
	// voxelizer::resize_reposition_grid_constrained (src/voxelizer.cpp:22-39)
	int32_t gmin[3];
	uint64_t size[3];
	double off[3];
	for (int d = 0; d < 3; ++d) {
		const double lo = floor((mn[d] - ref_grid_offset[d]) / cell_size), hi = ceil((mx[d] - ref_grid_offset[d]) / cell_size);
		if (!(fabs(lo) < 1e9) || !(fabs(hi) < 1e9)) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_voxelize_mesh: mesh bounds out of range");
		const int gl = (int)lo - 1, gh = (int)hi + 1;
		gmin[d] = gl;
		size[d] = (uint64_t)(gh - gl);
		off[d] = ref_grid_offset[d] + (double)gl * cell_size;
	}
	lfa_voxels *v = nullptr;
	int rc = lfa_voxels_create(&v, size, off, cell_size, device);
	if (rc != LFA_OK) return rc;
	for (int d = 0; d < 3; ++d) v->grid_min[d] = gmin[d];
	rc = lfa_voxels_voxelize_triangles(v, positions, n_vertices, indices, index_bytes, n_indices);
	if (rc == LFA_OK) rc = lfa_voxels_mark_exterior(v);
	if (rc != LFA_OK) {
		lfa_fail(nullptr, rc, "%s", v->err.c_str());
		lfa_voxels_destroy(v);
		return rc;
	}
	*out = v;
	return LFA_OK;
}

extern "C" int lfa_set_solid_from_voxels(lfa_sim *s, lfa_voxels *v, int include_interior, int include_surface) {
	if (!s || !v) return LFA_E_INVALID;
	++s->solid_epoch;
	if (s->device != v->device) return lfa_fail(s, LFA_E_INVALID, "lfa_set_solid_from_voxels: handles live on different devices");
	if (v->nc == 0) return LFA_OK;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_HIP(s, hipStreamSynchronize(v->stream));
	const int64_t ref[3] = {s->g.nx, s->g.ny, s->g.nz};
	const Select sel = make_select(v, include_interior, include_surface, ref);
	hipLaunchKernelGGL(k_voxels_to_solid, dim3((unsigned)((v->nc + 255) / 256)), dim3(256), 0, s->stream, (const uint8_t *)v->vox,
	                   v->n[0], v->n[1], v->nc, sel, s->solid, s->ctype, s->g);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	s->system_valid = false;
	return LFA_OK;
}
