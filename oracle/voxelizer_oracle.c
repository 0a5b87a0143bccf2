/* TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
 *
 * oracle/voxelizer_oracle.c : plain-C restatement of the reference's solid-boundary voxelizer,
 * the checker of libfluid_amd/csrc/voxelizer.hip. Each function cites the reference lines it follows
 * (paths relative to /root/reference). fp64, same operation order as the reference (compiled -ffp-contract=off).
 * Pinned against the real reference (oracle/_ref, ref_voxelize) by tests/test_voxelizer.py and by the golden vectors
 * in tests/golden/voxelizer_*.npz generated from it.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { VOX_INTERIOR = 0, VOX_EXTERIOR = 1, VOX_SURFACE = 2 }; /* include/fluid/voxelizer.h:17-21 */

typedef struct { double x, y, z; } v3;
static v3 sub3(v3 a, v3 b) { v3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
/* vec_ops::dot, include/fluid/math/vec.h:110-122 */
static double dot3(v3 a, v3 b) { double r = 0.0; r += a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }

/* aab_triangle_overlap_bounded_center, src/math/intersection.cpp:31-82 */
static int tri_box(v3 h, v3 p1, v3 p2, v3 p3) {
	v3 f[3], v[3], n;
	f[0] = sub3(p2, p1); f[1] = sub3(p3, p2); f[2] = sub3(p1, p3);
	n.x = f[0].y * f[1].z - f[0].z * f[1].y; /* vec_ops::cross, vec.h:546-548 */
	n.y = f[0].z * f[1].x - f[0].x * f[1].z;
	n.z = f[0].x * f[1].y - f[0].y * f[1].x;
	{
		v3 an = {fabs(n.x), fabs(n.y), fabs(n.z)};
		double center_off = dot3(p1, n), radius_n = dot3(an, h);
		if (fabs(center_off) > fabs(radius_n)) return 0;
	}
	v[0] = p1; v[1] = p2; v[2] = p3;
	for (int i = 0; i < 3; ++i) { /* :48-57 */
		v3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		double p0 = v1.z * fi.y - v1.y * fi.z, q = v2.z * fi.y - v2.y * fi.z;
		double pmin = q < p0 ? q : p0, pmax = q < p0 ? p0 : q;
		double r = h.y * fabs(fi.z) + h.z * fabs(fi.y);
		if (pmin > r || pmax < -r) return 0;
	}
	for (int i = 0; i < 3; ++i) { /* :59-68 */
		v3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		double p0 = v1.x * fi.z - v1.z * fi.x, q = v2.x * fi.z - v2.z * fi.x;
		double pmin = q < p0 ? q : p0, pmax = q < p0 ? p0 : q;
		double r = h.x * fabs(fi.z) + h.z * fabs(fi.x);
		if (pmin > r || pmax < -r) return 0;
	}
	for (int i = 0; i < 3; ++i) { /* :70-79 */
		v3 v1 = v[i], v2 = v[(i + 2) % 3], fi = f[i];
		double p0 = v1.y * fi.x - v1.x * fi.y, q = v2.y * fi.x - v2.x * fi.y;
		double pmin = q < p0 ? q : p0, pmax = q < p0 ? p0 : q;
		double r = h.x * fabs(fi.y) + h.y * fabs(fi.x);
		if (pmin > r || pmax < -r) return 0;
	}
	return 1;
}

/* voxelizer::get_bounding_box (include/fluid/voxelizer.h:24-34) + resize_reposition_grid_constrained
 * (src/voxelizer.cpp:22-39): offset of the voxel grid in the reference grid, its size and world offset. */
void orc_vox_grid(const double *pos, size_t nv, double cs, const double *ref_off, int32_t *grid_min, uint64_t *size,
                  double *grid_off) {
	double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
	if (nv) for (int d = 0; d < 3; ++d) mn[d] = mx[d] = pos[d];
	for (size_t i = 1; i < nv; ++i)
		for (int d = 0; d < 3; ++d) {
			if (pos[3 * i + d] < mn[d]) mn[d] = pos[3 * i + d];
			if (pos[3 * i + d] > mx[d]) mx[d] = pos[3 * i + d];
		}
	for (int d = 0; d < 3; ++d) {
		int lo = (int)floor((mn[d] - ref_off[d]) / cs), hi = (int)ceil((mx[d] - ref_off[d]) / cs);
		lo -= 1; hi += 1;
		grid_min[d] = lo;
		size[d] = (uint64_t)(hi - lo);
		grid_off[d] = ref_off[d] + (double)lo * cs;
	}
}

/* voxelizer::voxelize_triangle, src/voxelizer.cpp:54-81 */
static void voxelize_triangle(uint8_t *vox, const uint64_t *n, const double *off, double cs, v3 p1, v3 p2, v3 p3) {
	double mn[3] = {p1.x, p1.y, p1.z}, mx[3] = {p1.x, p1.y, p1.z};
	const double q2[3] = {p2.x, p2.y, p2.z}, q3[3] = {p3.x, p3.y, p3.z};
	for (int d = 0; d < 3; ++d) {
		if (q2[d] < mn[d]) mn[d] = q2[d];
		if (q2[d] > mx[d]) mx[d] = q2[d];
		if (q3[d] < mn[d]) mn[d] = q3[d];
		if (q3[d] > mx[d]) mx[d] = q3[d];
	}
	double half = 0.5 * cs;
	v3 h = {half, half, half};
	size_t lo[3], hi[3];
	double c0[3];
	for (int d = 0; d < 3; ++d) {
		lo[d] = (size_t)((mn[d] - off[d]) / cs);
		hi[d] = (size_t)((mx[d] - off[d]) / cs);
		c0[d] = off[d] + (double)lo[d] * cs + half;
	}
	v3 c = {c0[0], c0[1], c0[2]};
	for (size_t z = lo[2]; z <= hi[2]; ++z, c.z += cs) {
		c.y = c0[1];
		for (size_t y = lo[1]; y <= hi[1]; ++y, c.y += cs) {
			c.x = c0[0];
			for (size_t x = lo[0]; x <= hi[0]; ++x, c.x += cs) {
				uint8_t *t = &vox[x + n[0] * (y + n[1] * z)];
				if (*t != VOX_SURFACE && tri_box(h, sub3(p1, c), sub3(p2, c), sub3(p3, c))) *t = VOX_SURFACE;
			}
		}
	}
}

/* voxelizer::mark_exterior, src/voxelizer.cpp:83-124 (explicit stack, same visiting rule) */
static void mark_exterior(uint8_t *vox, const uint64_t *n) {
	size_t nc = (size_t)(n[0] * n[1] * n[2]);
	if (nc == 0 || vox[0] == VOX_SURFACE) return;
	size_t *stack = (size_t *)malloc(nc * sizeof(size_t)), top = 0;
	vox[0] = VOX_EXTERIOR;
	stack[top++] = 0;
	const size_t sy = n[0], sz = n[0] * n[1];
	while (top) {
		size_t r = stack[--top];
		size_t x = r % n[0], y = (r / n[0]) % n[1], z = r / sz;
#define PUSH(cond, rr) if ((cond) && vox[rr] == VOX_INTERIOR) { vox[rr] = VOX_EXTERIOR; stack[top++] = (rr); }
		PUSH(z > 0, r - sz)
		PUSH(y > 0, r - sy)
		PUSH(x > 0, r - 1)
		PUSH(z + 1 < n[2], r + sz)
		PUSH(y + 1 < n[1], r + sy)
		PUSH(x + 1 < n[0], r + 1)
#undef PUSH
	}
	free(stack);
}

/* The sequence of src/data_structures/obstacle.cpp:12-18 / plugins/maya/nodes/voxelizer_node.cpp:255-268.
 * `types` has the size orc_vox_grid reports. */
void orc_voxelize(const double *pos, size_t nv, const uint64_t *idx, size_t ni, double cs, const double *ref_off,
                  uint8_t *types) {
	int32_t gmin[3];
	uint64_t n[3];
	double off[3];
	orc_vox_grid(pos, nv, cs, ref_off, gmin, n, off);
	memset(types, VOX_INTERIOR, (size_t)(n[0] * n[1] * n[2]));
	for (size_t i = 0; i + 2 < ni; i += 3) { /* voxelize_mesh_surface, include/fluid/voxelizer.h:55-63 */
		v3 p[3];
		for (int k = 0; k < 3; ++k) {
			p[k].x = pos[3 * idx[i + k]]; p[k].y = pos[3 * idx[i + k] + 1]; p[k].z = pos[3 * idx[i + k] + 2];
		}
		voxelize_triangle(types, n, off, cs, p[0], p[1], p[2]);
	}
	mark_exterior(types, n);
}
