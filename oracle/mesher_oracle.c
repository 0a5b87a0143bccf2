/* TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
 *
 * oracle/mesher_oracle.c : plain-C restatement of the reference's surface mesher, the checker of
 * libfluid_amd/csrc/mesher.hip. Each function cites the reference lines it follows (paths relative to /root/reference).
 * fp64, same operation and summation order as the reference (compiled -ffp-contract=off). Pinned against the real
 * reference (oracle/_ref: ref_mesher_surface / ref_mesher_mesh) by tests/test_mesher.py and the golden vectors generated
 * from it. The marching-cubes case table is shared as constants with the product header (libfluid_amd/csrc/mc_tables.h).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../libfluid_amd/csrc/mc_tables.h"

typedef struct {
	size_t n[3];     /* cells; the surface function has n+1 points per axis (mesher::resize, src/mesher.cpp:320-323) */
	double off[3], cs, extent;
	size_t radius;
} mesher_cfg;

static mesher_cfg make_cfg(const uint64_t *size, const double *off, double cs, double extent, uint64_t radius) {
	mesher_cfg c;
	for (int d = 0; d < 3; ++d) { c.n[d] = (size_t)size[d]; c.off[d] = off[d]; }
	c.cs = cs; c.extent = extent; c.radius = (size_t)radius;
	return c;
}

/* mesher::_kernel, src/mesher.cpp:325-331 */
static double kernel(double sqr_dist) {
	sqr_dist = 1.0 - sqr_dist;
	return sqr_dist > 0.0 ? sqr_dist * sqr_dist * sqr_dist : 0.0;
}

/* mesher::_sample_surface_function, src/mesher.cpp:333-376. The reference's space_hashing keeps a linked list per cell
 * with head insertion (include/fluid/data_structures/space_hashing.h:55-62), so a cell's particles are visited in
 * reverse insertion order; cells in grid order over [g - radius, g + radius - 1] clamped (grid.h:116-135). */
static void sample_surface(const mesher_cfg *c, const double *pos, size_t np, double r, double *values) {
	const size_t nx = c->n[0], ny = c->n[1], nz = c->n[2], ncell = nx * ny * nz;
	size_t *start = (size_t *)calloc(ncell + 1, sizeof(size_t)), *cell_of = (size_t *)malloc((np ? np : 1) * sizeof(size_t));
	size_t *order = (size_t *)malloc((np ? np : 1) * sizeof(size_t));
	for (size_t i = 0; i < np; ++i) {
		cell_of[i] = (size_t)-1;
		int idx[3];
		for (int d = 0; d < 3; ++d) idx[d] = (int)((pos[3 * i + d] - c->off[d]) / c->cs); /* vec3i(...): truncation */
		if (idx[0] > 0 && idx[1] > 0 && idx[2] > 0 && (size_t)idx[0] < nx && (size_t)idx[1] < ny && (size_t)idx[2] < nz) {
			cell_of[i] = (size_t)idx[0] + nx * ((size_t)idx[1] + ny * (size_t)idx[2]);
			start[cell_of[i] + 1]++;
		}
	}
	for (size_t k = 0; k < ncell; ++k) start[k + 1] += start[k];
	size_t *cur = (size_t *)malloc((ncell ? ncell : 1) * sizeof(size_t));
	memcpy(cur, start, ncell * sizeof(size_t));
	for (size_t i = 0; i < np; ++i) if (cell_of[i] != (size_t)-1) order[cur[cell_of[i]]++] = i; /* insertion order */
	const size_t R = c->radius, px = nx + 1, py = ny + 1, pz = nz + 1;
	for (size_t z = 0; z < pz; ++z)
		for (size_t y = 0; y < py; ++y)
			for (size_t x = 0; x < px; ++x) {
				double tw = 0.0, tr = 0.0, tp[3] = {0.0, 0.0, 0.0};
				const double gp[3] = {c->off[0] + c->cs * (double)x, c->off[1] + c->cs * (double)y, c->off[2] + c->cs * (double)z};
				int has = 0;
				const size_t x0 = x < R ? 0 : x - R, y0 = y < R ? 0 : y - R, z0 = z < R ? 0 : z - R;
				size_t x1 = x + (R - 1) + 1, y1 = y + (R - 1) + 1, z1 = z + (R - 1) + 1;
				if (x1 > nx) x1 = nx;
				if (y1 > ny) y1 = ny;
				if (z1 > nz) z1 = nz;
				for (size_t cz = z0; cz < z1; ++cz)
					for (size_t cy = y0; cy < y1; ++cy)
						for (size_t cx = x0; cx < x1; ++cx) {
							const size_t cell = cx + nx * (cy + ny * cz);
							for (size_t k = start[cell + 1]; k-- > start[cell];) { /* newest first */
								const double *p = pos + 3 * order[k];
								has = 1;
								const double d[3] = {p[0] - gp[0], p[1] - gp[1], p[2] - gp[2]};
								double sq = 0.0;
								sq += d[0] * d[0]; sq += d[1] * d[1]; sq += d[2] * d[2];
								const double w = kernel(sq / (c->extent * c->extent));
								tw += w;
								tr += w * r;
								tp[0] += w * p[0]; tp[1] += w * p[1]; tp[2] += w * p[2];
							}
						}
				double value = 1.0;
				if (has) {
					tr /= tw;
					double sq = 0.0;
					for (int d = 0; d < 3; ++d) { const double q = tp[d] / tw - gp[d]; sq += q * q; }
					value = sqrt(sq) - tr;
				}
				values[x + px * (y + py * z)] = value;
			}
	free(start); free(cell_of); free(order); free(cur);
}

void orc_mesher_surface(const double *pos, size_t n, const uint64_t *size, const double *off, double cs, double extent,
                        uint64_t radius, double r, double *values) {
	mesher_cfg c = make_cfg(size, off, cs, extent, radius);
	sample_surface(&c, pos, n, r, values);
}

typedef struct { double *v; size_t n, cap; } dvec;
typedef struct { uint64_t *v; size_t n, cap; } uvec;
static size_t push_point(dvec *a, const double *p) {
	if (a->n + 3 > a->cap) { a->cap = a->cap ? 2 * a->cap : 3072; a->v = (double *)realloc(a->v, a->cap * sizeof(double)); }
	memcpy(a->v + a->n, p, 24);
	a->n += 3;
	return a->n / 3 - 1;
}
static void push_index(uvec *a, uint64_t i) {
	if (a->n + 1 > a->cap) { a->cap = a->cap ? 2 * a->cap : 3072; a->v = (uint64_t *)realloc(a->v, a->cap * sizeof(uint64_t)); }
	a->v[a->n++] = i;
}

/* mesher::_add_point, src/mesher.cpp:378-392; lerp(a, b, t) = a (1 - t) + b t (include/fluid/misc.h:20-22) */
static size_t add_point(const mesher_cfg *c, dvec *out, const size_t *cell, const double *f, int edge) {
	const int a = MC_EDGE_CORNERS[edge][0], b = MC_EDGE_CORNERS[edge][1];
	const double v1 = f[a], v2 = f[b], t = v1 / (v1 - v2);
	double p[3];
	for (int d = 0; d < 3; ++d) {
		const double pa = (double)(cell[d] + MC_CORNER_OFFSETS[a][d]), pb = (double)(cell[d] + MC_CORNER_OFFSETS[b][d]);
		p[d] = c->off[d] + c->cs * (pa * (1.0 - t) + pb * t);
	}
	return push_point(out, p);
}

/* mesher::_marching_cubes, src/mesher.cpp:400-515: one sweep in z, y, x order; vertices on edges shared with cells that
 * come later are remembered per layer (mid0/mid3 of the previous and the current layer, mid8 of the current row). */
static void marching_cubes(const mesher_cfg *c, const double *values, dvec *vp, uvec *idx) {
	const size_t px = c->n[0] + 1, py = c->n[1] + 1, pz = c->n[2] + 1, layer = px * py;
	size_t *prev0 = (size_t *)calloc(layer, sizeof(size_t)), *prev3 = (size_t *)calloc(layer, sizeof(size_t));
	size_t *cur0 = (size_t *)calloc(layer, sizeof(size_t)), *cur3 = (size_t *)calloc(layer, sizeof(size_t));
	size_t *m8 = (size_t *)calloc(layer, sizeof(size_t));
	for (size_t z = 0; z + 1 < pz; ++z) {
		for (size_t y = 0; y + 1 < py; ++y)
			for (size_t x = 0; x + 1 < px; ++x) {
				double f[8];
				uint8_t occ = 0;
				for (int i = 0; i < 8; ++i) {
					f[i] = values[(x + MC_CORNER_OFFSETS[i][0]) + px * ((y + MC_CORNER_OFFSETS[i][1]) + py * (z + MC_CORNER_OFFSETS[i][2]))];
					occ |= (uint8_t)((f[i] < 0 ? 1 : 0) << i);
				}
				const uint16_t el = mc_edge_mask(occ);
				if (!el) continue;
				const size_t cell[3] = {x, y, z}, l00 = x + px * y, l01 = l00 + 1, l10 = l00 + px;
				size_t ids[12] = {0};
#define HAS(e) (el & (1u << (e)))
				if (z == 0) {
					if (y == 0 && HAS(0)) prev0[l00] = add_point(c, vp, cell, f, 0);
					if (HAS(1)) prev3[l01] = add_point(c, vp, cell, f, 1);
					if (HAS(2)) prev0[l10] = add_point(c, vp, cell, f, 2);
					if (x == 0 && HAS(3)) prev3[l00] = add_point(c, vp, cell, f, 3);
				}
				ids[0] = prev0[l00]; ids[1] = prev3[l01]; ids[2] = prev0[l10]; ids[3] = prev3[l00];
				if (y == 0 && HAS(4)) cur0[l00] = add_point(c, vp, cell, f, 4);
				ids[4] = cur0[l00];
				if (x == 0 && HAS(7)) cur3[l00] = add_point(c, vp, cell, f, 7);
				ids[7] = cur3[l00];
				if (x == 0 && y == 0 && HAS(8)) m8[l00] = add_point(c, vp, cell, f, 8);
				ids[8] = m8[l00];
				if (y == 0 && HAS(9)) m8[l01] = add_point(c, vp, cell, f, 9);
				ids[9] = m8[l01];
				if (x == 0 && HAS(11)) m8[l10] = add_point(c, vp, cell, f, 11);
				ids[11] = m8[l10];
				if (HAS(5)) ids[5] = add_point(c, vp, cell, f, 5);
				if (HAS(6)) ids[6] = add_point(c, vp, cell, f, 6);
				if (HAS(10)) ids[10] = add_point(c, vp, cell, f, 10);
#undef HAS
				cur3[l01] = ids[5];
				cur0[l10] = ids[6];
				m8[l10 + 1] = ids[10];
				for (int k = 0; MC_TRIANGLES[occ][k] != MC_END; ++k) push_index(idx, ids[MC_TRIANGLES[occ][k]]);
			}
		size_t *t = prev0; prev0 = cur0; cur0 = t;
		t = prev3; prev3 = cur3; cur3 = t;
	}
	free(prev0); free(prev3); free(cur0); free(cur3); free(m8);
}

void orc_mesher_mesh(const double *pos, size_t n, const uint64_t *size, const double *off, double cs, double extent,
                     uint64_t radius, double r, const double *values, double *vpos, size_t cap_v, uint64_t *idx, size_t cap_i,
                     uint64_t *counts) {
	mesher_cfg c = make_cfg(size, off, cs, extent, radius);
	const size_t npts = (c.n[0] + 1) * (c.n[1] + 1) * (c.n[2] + 1);
	double *vals = (double *)malloc(npts * sizeof(double));
	if (values) memcpy(vals, values, npts * sizeof(double));
	else sample_surface(&c, pos, n, r, vals);
	dvec vp = {0, 0, 0};
	uvec ix = {0, 0, 0};
	marching_cubes(&c, vals, &vp, &ix);
	counts[0] = vp.n / 3;
	counts[1] = ix.n;
	/* (an empty mesh has null arrays: memcpy must not see them even for 0 bytes - found by the sanitizer run, `make asan`) */
	if (vpos && vp.v) memcpy(vpos, vp.v, (vp.n / 3 < cap_v ? vp.n / 3 : cap_v) * 24);
	if (idx && ix.v) memcpy(idx, ix.v, (ix.n < cap_i ? ix.n : cap_i) * 8);
	free(vals); free(vp.v); free(ix.v);
}
