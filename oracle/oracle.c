/* TEST INFRASTRUCTURE ONLY -- never linked into, imported or called by the product path (libfluid_amd/).
 *
 * oracle/oracle.c : plain-C, fp64 restatement of lukedan/libfluid's per-step hot path
 * (particle hashing, P2G PIC/FLIP/APIC, gravity, pressure system + MIC(0)-PCG, pressure gradient, velocity
 * extrapolation, G2P PIC/FLIP/APIC, CFL). Each function names the reference file:line it follows
 * (paths relative to /root/reference). It keeps the reference's data layout (152-B particle AoS, 32-B cell AoS,
 * x-fastest grids), stage order, serial loop structure and quirks, so it doubles as the on-box CPU baseline
 * (bench.py cpu_baseline, kind "port").
 *
 * PARITY PINNING: the reference ships no tests or golden vectors, so this restatement is pinned
 * against outputs of the reference itself: oracle/_ref/libref.so (the reference's own sources compiled in place,
 * oracle/ref_harness.cpp) live in tests/test_oracle.py, tests/test_configs.py and tests/test_sources.py, and against the fixtures that build generated,
 * committed under tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
/* memcpy / memset of n bytes, n may be 0 and the pointers then null (empty particle or unknown sets): ISO C gives memcpy no
 * licence for null arguments even then (found by `make asan`) */
static void copy_n(void *dst, const void *src, size_t n) { if (n) memcpy(dst, src, n); }
static void zero_n(void *dst, size_t n) { if (n) memset(dst, 0, n); }

typedef struct {
	double pos[3], vel[3], cx[3], cy[3], cz[3], old_pos[3]; /* include/fluid/simulation.h:24-33 */
	uint64_t raw;                                             /* include/fluid/simulation.h:34 */
} orc_particle;
typedef struct {
	double vel[3]; /* velocities_posface, include/fluid/mac_grid.h:25 */
	uint8_t type;  /* include/fluid/mac_grid.h:17-21,26 */
	uint8_t pad[7];
} orc_cell;
_Static_assert(sizeof(orc_particle) == 152, "particle AoS is 152 B");
_Static_assert(sizeof(orc_cell) == 32, "cell AoS is 32 B");

enum { T_AIR = 1, T_FLUID = 2, T_SOLID = 4 };
enum { M_PIC = 0, M_FLIP = 1, M_APIC = 2 }; /* include/fluid/simulation.h:44-48 */
#define NOT_FLUID UINT64_MAX                 /* include/fluid/pressure_solver.h:45 */

typedef struct {
	size_t n[3];
	double h, off[3], g[3], blend, density;
	int method;
	size_t extrap_iters; /* include/fluid/simulation.h:189 */
	double tau, sigma, tol; /* include/fluid/pressure_solver.h:39-41 */
	size_t maxit;           /* include/fluid/pressure_solver.h:42 */
	double skin, stiffness; /* include/fluid/simulation.h:187-188 */

	orc_particle *p, *ptmp;
	size_t np, pcap;
	orc_cell *grid, *old_grid;
	uint64_t *hbegin, *hcount;      /* _space_hash, include/fluid/simulation.h:207 */
	uint64_t *fluid_raw;            /* _fluid_cells (raw), include/fluid/simulation.h:209 */
	size_t nfluid;

	/* pressure system (include/fluid/pressure_solver.h:50-56) */
	uint64_t *cell_to_unknown;
	size_t (*fc)[3];
	uint8_t *abits;
	double a_scale, *precon;
	size_t nsys;

	/* fluid sources (include/fluid/data_structures/source.h:12-22; simulation.h:179 `sources`) */
	struct orc_source { int *xyz; size_t k; double vel[3]; size_t root; int active, coerce; } *src;
	size_t nsrc;
	uint64_t rng; /* seeding draws: splitmix64 state (the reference draws from its pcg32 member, simulation.h:177) */
} orc_ctx;

static size_t ncells(const orc_ctx *c) { return c->n[0] * c->n[1] * c->n[2]; }
/* grid::index_to_raw, include/fluid/data_structures/grid.h:212-222 (x-fastest). */
static size_t raw_of(const orc_ctx *c, size_t x, size_t y, size_t z) { return x + c->n[0] * (y + c->n[1] * z); }

void *orc_create(size_t nx, size_t ny, size_t nz, double h, const double *off, const double *g, int method,
                 double blend, double density) {
	orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
	c->n[0] = nx; c->n[1] = ny; c->n[2] = nz;
	c->h = h; c->blend = blend; c->density = density; c->method = method;
	memcpy(c->off, off, 24); memcpy(c->g, g, 24);
	c->extrap_iters = 1; c->tau = 0.97; c->sigma = 0.25; c->tol = 1e-6; c->maxit = 200;
	c->skin = 0.1; c->stiffness = 5.0;
	size_t nc = ncells(c);
	c->grid = (orc_cell *)calloc(nc, sizeof(orc_cell));
	c->old_grid = (orc_cell *)calloc(nc, sizeof(orc_cell));
	for (size_t i = 0; i < nc; ++i) { c->grid[i].type = T_AIR; c->old_grid[i].type = T_AIR; }
	c->hbegin = (uint64_t *)calloc(nc, 8);
	c->hcount = (uint64_t *)calloc(nc, 8);
	c->fluid_raw = (uint64_t *)malloc(nc * 8);
	c->cell_to_unknown = (uint64_t *)malloc(nc * 8);
	return c;
}
void orc_destroy(void *h) {
	orc_ctx *c = (orc_ctx *)h;
	free(c->p); free(c->ptmp); free(c->grid); free(c->old_grid); free(c->hbegin); free(c->hcount);
	free(c->fluid_raw); free(c->cell_to_unknown); free(c->fc); free(c->abits); free(c->precon);
	for (size_t i = 0; i < c->nsrc; ++i) free(c->src[i].xyz);
	free(c->src); free(c);
}
void orc_set_extrapolation_iterations(void *h, size_t n) { ((orc_ctx *)h)->extrap_iters = n; }
void orc_set_pcg_params(void *h, double tau, double sigma, double tol, size_t maxit) {
	orc_ctx *c = (orc_ctx *)h; c->tau = tau; c->sigma = sigma; c->tol = tol; c->maxit = maxit;
}
void orc_set_solid_cells(void *h, const int *xyz, size_t k) {
	orc_ctx *c = (orc_ctx *)h;
	for (size_t i = 0; i < k; ++i) c->grid[raw_of(c, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])].type = T_SOLID;
}
void orc_set_particles(void *h, const void *aos152, size_t n) {
	orc_ctx *c = (orc_ctx *)h;
	if (n > c->pcap) {
		free(c->p); free(c->ptmp);
		c->p = (orc_particle *)malloc(n * sizeof(orc_particle));
		c->ptmp = (orc_particle *)malloc(n * sizeof(orc_particle));
		c->pcap = n;
	}
	c->np = n;
	copy_n(c->p, aos152, n * sizeof(orc_particle));
}
size_t orc_num_particles(void *h) { return ((orc_ctx *)h)->np; }
void orc_get_particles(void *h, void *aos152) {
	orc_ctx *c = (orc_ctx *)h; copy_n(aos152, c->p, c->np * sizeof(orc_particle));
}
void orc_get_cells(void *h, void *aos32) { orc_ctx *c = (orc_ctx *)h; memcpy(aos32, c->grid, ncells(c) * 32); }
void orc_set_cells(void *h, const void *aos32) { orc_ctx *c = (orc_ctx *)h; memcpy(c->grid, aos32, ncells(c) * 32); }
void orc_get_old_cells(void *h, void *aos32) { orc_ctx *c = (orc_ctx *)h; memcpy(aos32, c->old_grid, ncells(c) * 32); }

/* ------------------------------------------------------------------------------------------------ a1, a2 */

/* simulation::update_and_hash_particles src/simulation.cpp:251-264 + hash_particles :266-291.
 * key = raw index of min(size_t(max(pos,0)), size-1) per axis (true division by cell_size, :253).
 * The reference uses std::sort (unstable, :269); intra-cell order is unspecified there. This restatement uses
 * a stable counting sort, so its intra-cell order is "input order". */
void orc_hash(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	size_t nc = ncells(c);
	for (size_t i = 0; i < c->np; ++i) {
		size_t idx[3];
		for (int d = 0; d < 3; ++d) {
			double gp = (c->p[i].pos[d] - c->off[d]) / c->h;
			double m = gp < 0.0 ? 0.0 : gp; /* std::max(pos, 0.0), :256 */
			size_t v = (size_t)m;
			idx[d] = v < c->n[d] - 1 ? v : c->n[d] - 1;
		}
		c->p[i].raw = raw_of(c, idx[0], idx[1], idx[2]);
	}
	/* reset_space_hash :131-134 */
	memset(c->hbegin, 0, nc * 8);
	memset(c->hcount, 0, nc * 8);
	c->nfluid = 0;
	if (c->np == 0) return;
	for (size_t i = 0; i < c->np; ++i) c->hcount[c->p[i].raw]++;
	{
		uint64_t run = 0;
		for (size_t r = 0; r < nc; ++r) {
			if (c->hcount[r]) { c->hbegin[r] = run; c->fluid_raw[c->nfluid++] = r; run += c->hcount[r]; }
		}
	}
	{
		uint64_t *cursor = (uint64_t *)malloc(nc * 8);
		memcpy(cursor, c->hbegin, nc * 8);
		for (size_t i = 0; i < c->np; ++i) c->ptmp[cursor[c->p[i].raw]++] = c->p[i];
		free(cursor);
		orc_particle *t = c->p; c->p = c->ptmp; c->ptmp = t;
	}
	/* the reference leaves begin=0 for the first occupied cell and for empty cells (:273-289); hbegin of the
	 * first run is 0 here as well. */
}
size_t orc_num_fluid_cells(void *h) { return ((orc_ctx *)h)->nfluid; }
void orc_get_fluid_cells(void *h, uint64_t *out) { orc_ctx *c = (orc_ctx *)h; copy_n(out, c->fluid_raw, c->nfluid * 8); }
void orc_get_space_hash(void *h, uint64_t *begin, uint64_t *count) {
	orc_ctx *c = (orc_ctx *)h; memcpy(begin, c->hbegin, ncells(c) * 8); memcpy(count, c->hcount, ncells(c) * 8);
}

/* ------------------------------------------------------------------------------------------- a3 .. a6 */

/* simulation::_kernel src/simulation.cpp:207-213. */
static double hat3(double x, double y, double z) {
	double a = 1.0 - fabs(x), b = 1.0 - fabs(y), d = 1.0 - fabs(z);
	a = a > 0.0 ? a : 0.0; b = b > 0.0 ? b : 0.0; d = d > 0.0 ? d : 0.0;
	return a * b * d;
}

/* simulation::_remove_boundary_velocities src/simulation.cpp:428-445. */
static void zero_max_faces(const orc_ctx *c, orc_cell *g) {
	if (ncells(c) == 0) return;
	size_t mx = c->n[0] - 1, my = c->n[1] - 1, mz = c->n[2] - 1;
	for (size_t z = 0; z < c->n[2]; ++z) {
		for (size_t y = 0; y < c->n[1]; ++y) g[raw_of(c, mx, y, z)].vel[0] = 0.0;
		for (size_t x = 0; x < c->n[0]; ++x) g[raw_of(c, x, my, z)].vel[1] = 0.0;
	}
	for (size_t y = 0; y < c->n[1]; ++y)
		for (size_t x = 0; x < c->n[0]; ++x) g[raw_of(c, x, y, mz)].vel[2] = 0.0;
}

/* simulation::_transfer_to_grid_pic src/simulation.cpp:293-338 and _transfer_to_grid_apic :346-398.
 * One serial sweep over ALL cells (z,y,x); each cell gathers the particles of its clamped 27-cell neighbourhood
 * (_for_all_nearby_particles include/fluid/simulation.h:212-223; range clamp grid.h:126-135; visit order z,y,x).
 * Face coordinates are accumulated by repeated += cell_size (:296-300 / :349-353).
 * Quirk kept: APIC weights use (p - face) WITHOUT dividing by cell_size (:367-369); PIC divides (:313-315). */
static void p2g_sweep(orc_ctx *c, int apic) {
	const double h = c->h, half = 0.5 * h;
	double zpos = c->off[2] + half;
	for (size_t z = 0; z < c->n[2]; ++z, zpos += h) {
		double zface = zpos + half, ypos = c->off[1] + half;
		for (size_t y = 0; y < c->n[1]; ++y, ypos += h) {
			double yface = ypos + half, xpos = c->off[0] + half;
			for (size_t x = 0; x < c->n[0]; ++x, xpos += h) {
				double xface = xpos + half;
				orc_cell *cell = &c->grid[raw_of(c, x, y, z)];
				double sv[3] = {0, 0, 0}, sw[3] = {0, 0, 0};
				/* face sample points: xface=(xface,ypos,zpos) yface=(xpos,yface,zpos) zface=(xpos,ypos,zface) */
				const double fx[3] = {xface, ypos, zpos}, fy[3] = {xpos, yface, zpos}, fz[3] = {xpos, ypos, zface};
				size_t x0 = x < 1 ? 0 : x - 1, y0 = y < 1 ? 0 : y - 1, z0 = z < 1 ? 0 : z - 1;
				size_t x1 = x + 2 < c->n[0] ? x + 2 : c->n[0], y1 = y + 2 < c->n[1] ? y + 2 : c->n[1],
				       z1 = z + 2 < c->n[2] ? z + 2 : c->n[2];
				for (size_t zz = z0; zz < z1; ++zz)
					for (size_t yy = y0; yy < y1; ++yy)
						for (size_t xx = x0; xx < x1; ++xx) {
							size_t r = raw_of(c, xx, yy, zz);
							const orc_particle *q = c->p + c->hbegin[r];
							for (uint64_t k = 0; k < c->hcount[r]; ++k, ++q) {
								double w[3];
								if (!apic) {
									w[0] = hat3((q->pos[0] - fx[0]) / h, (q->pos[1] - fx[1]) / h, (q->pos[2] - fx[2]) / h);
									w[1] = hat3((q->pos[0] - fy[0]) / h, (q->pos[1] - fy[1]) / h, (q->pos[2] - fy[2]) / h);
									w[2] = hat3((q->pos[0] - fz[0]) / h, (q->pos[1] - fz[1]) / h, (q->pos[2] - fz[2]) / h);
									for (int d = 0; d < 3; ++d) { sw[d] += w[d]; sv[d] += w[d] * q->vel[d]; }
								} else {
									w[0] = hat3(q->pos[0] - fx[0], q->pos[1] - fx[1], q->pos[2] - fx[2]);
									w[1] = hat3(q->pos[0] - fy[0], q->pos[1] - fy[1], q->pos[2] - fy[2]);
									w[2] = hat3(q->pos[0] - fz[0], q->pos[1] - fz[1], q->pos[2] - fz[2]);
									/* vec_ops::dot accumulates 0 + x + y + z (include/fluid/math/vec.h:110-121) */
									double ax = 0.0, ay = 0.0, az = 0.0;
									for (int d = 0; d < 3; ++d) {
										ax += q->cx[d] * (fx[d] - q->pos[d]);
										ay += q->cy[d] * (fy[d] - q->pos[d]);
										az += q->cz[d] * (fz[d] - q->pos[d]);
									}
									const double aff[3] = {ax, ay, az};
									for (int d = 0; d < 3; ++d) { sw[d] += w[d]; sv[d] += w[d] * (q->vel[d] + aff[d]); }
								}
							}
						}
				for (int d = 0; d < 3; ++d) cell->vel[d] = sw[d] > 1e-6 ? sv[d] / sw[d] : 0.0; /* :324 / :383 */
				if (cell->type != T_SOLID) { /* :329-334 / :388-393 */
					cell->type = T_AIR;
					if (c->hcount[raw_of(c, x, y, z)] > 0) cell->type = T_FLUID;
				}
			}
		}
	}
}

/* simulation::_transfer_to_grid src/simulation.cpp:400-412 (+ _transfer_to_grid_flip :340-344). */
void orc_p2g(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	switch (c->method) {
	case M_PIC: p2g_sweep(c, 0); break;
	case M_FLIP:
		p2g_sweep(c, 0);
		memcpy(c->old_grid, c->grid, ncells(c) * sizeof(orc_cell));
		zero_max_faces(c, c->old_grid); /* on _old_grid only: the live grid keeps its wall-face velocities */
		break;
	case M_APIC: p2g_sweep(c, 1); zero_max_faces(c, c->grid); break;
	}
}

/* ---------------------------------------------------------------------------------------------------- a7 */
/* gravity loop src/simulation.cpp:72-78: every cell, all three components, walls and solids included. */
void orc_add_gravity(void *hh, double dt) {
	orc_ctx *c = (orc_ctx *)hh;
	size_t nc = ncells(c);
	const double gx = c->g[0] * dt, gy = c->g[1] * dt, gz = c->g[2] * dt;
	for (size_t i = 0; i < nc; ++i) { c->grid[i].vel[0] += gx; c->grid[i].vel[1] += gy; c->grid[i].vel[2] += gz; }
}

/* ------------------------------------------------------------------------------------------ a8 .. a13 */

/* mac_grid::get_cell_and_type src/mac_grid.cpp:26-31: out of range (incl. wrapped "negative") => solid. */
static int type_at(const orc_ctx *c, size_t x, size_t y, size_t z) {
	if (x >= c->n[0] || y >= c->n[1] || z >= c->n[2]) return T_SOLID;
	return c->grid[raw_of(c, x, y, z)].type;
}
/* pressure_solver::_get_neg_neighbor_index / _get_pos_neighbor_index include/fluid/pressure_solver.h:59-71. */
static uint64_t nb_neg(const orc_ctx *c, const size_t *p, int d) {
	if (p[d] == 0) return NOT_FLUID;
	size_t q[3] = {p[0], p[1], p[2]}; q[d]--;
	return c->cell_to_unknown[raw_of(c, q[0], q[1], q[2])];
}
static uint64_t nb_pos(const orc_ctx *c, const size_t *p, int d) {
	if (p[d] + 1 >= c->n[d]) return NOT_FLUID;
	size_t q[3] = {p[0], p[1], p[2]}; q[d]++;
	return c->cell_to_unknown[raw_of(c, q[0], q[1], q[2])];
}
#define BIT_X(a) (((a) >> 3) & 1)
#define BIT_Y(a) (((a) >> 4) & 1)
#define BIT_Z(a) (((a) >> 5) & 1)
#define NONSOLID(a) ((a) & 7)
static int bit_d(uint8_t a, int d) { return (a >> (3 + d)) & 1; }

/* fluid cell list src/simulation.cpp:83-94 (precise_collision_detection: solid cells holding particles stay);
 * pressure_solver ctor + _compute_fluid_cell_indices src/pressure_solver.cpp:14-17,150-155;
 * _a_scale :22; _compute_a_matrix :160-178; _compute_preconditioner :244-294. */
void orc_build_system(void *hh, double dt) {
	orc_ctx *c = (orc_ctx *)hh;
	size_t nc = ncells(c), n = c->nfluid;
	free(c->fc); free(c->abits); free(c->precon);
	c->fc = (size_t(*)[3])malloc((n ? n : 1) * sizeof(size_t[3]));
	c->abits = (uint8_t *)malloc(n ? n : 1);
	c->precon = (double *)calloc(n ? n : 1, 8);
	c->nsys = n;
	for (size_t i = 0; i < nc; ++i) c->cell_to_unknown[i] = NOT_FLUID;
	for (size_t i = 0; i < n; ++i) {
		uint64_t r = c->fluid_raw[i]; /* grid::index_from_raw, grid.h:224-234 */
		c->fc[i][0] = r % c->n[0]; r /= c->n[0];
		c->fc[i][1] = r % c->n[1]; r /= c->n[1];
		c->fc[i][2] = r % c->n[2];
		c->cell_to_unknown[c->fluid_raw[i]] = i;
	}
	c->a_scale = dt / (c->density * c->h * c->h);
	for (size_t i = 0; i < n; ++i) {
		const size_t *p = c->fc[i];
		unsigned ns = 0;
		ns += type_at(c, p[0] + 1, p[1], p[2]) != T_SOLID;
		ns += type_at(c, p[0], p[1] + 1, p[2]) != T_SOLID;
		ns += type_at(c, p[0], p[1], p[2] + 1) != T_SOLID;
		ns += type_at(c, p[0] - 1, p[1], p[2]) != T_SOLID; /* size_t wrap => out of range => solid */
		ns += type_at(c, p[0], p[1] - 1, p[2]) != T_SOLID;
		ns += type_at(c, p[0], p[1], p[2] - 1) != T_SOLID;
		unsigned fxp = type_at(c, p[0] + 1, p[1], p[2]) == T_FLUID;
		unsigned fyp = type_at(c, p[0], p[1] + 1, p[2]) == T_FLUID;
		unsigned fzp = type_at(c, p[0], p[1], p[2] + 1) == T_FLUID;
		c->abits[i] = (uint8_t)((ns & 7) | (fxp << 3) | (fyp << 4) | (fzp << 5)); /* 3-bit field: 6 fits */
	}
	/* MIC(0), sequential in unknown order */
	for (size_t i = 0; i < n; ++i) {
		const size_t *p = c->fc[i];
		double neg_e = 0.0, neg_e_tau = 0.0;
		for (int d = 0; d < 3; ++d) {
			uint64_t j = nb_neg(c, p, d);
			if (j == NOT_FLUID) continue;
			uint8_t a = c->abits[j];
			double pj = c->precon[j];
			double ap = bit_d(a, d) * pj;
			neg_e += ap * ap;
			int o1 = (d + 1) % 3, o2 = (d + 2) % 3;
			/* x: (ypos+zpos), y: (xpos+zpos), z: (xpos+ypos) -- integer sums, order irrelevant */
			neg_e_tau += (double)(bit_d(a, d) * (bit_d(a, o1) + bit_d(a, o2))) * pj * pj;
		}
		double nsd = (double)NONSOLID(c->abits[i]);
		double e = nsd - (neg_e + c->tau * neg_e_tau) * c->a_scale;
		if (e < c->sigma * nsd) e = nsd;
		c->precon[i] = 1.0 / sqrt(e * c->a_scale);
	}
}
void orc_get_abits(void *h, uint8_t *out) { orc_ctx *c = (orc_ctx *)h; copy_n(out, c->abits, c->nsys); }
void orc_get_precon(void *h, double *out) { orc_ctx *c = (orc_ctx *)h; copy_n(out, c->precon, c->nsys * 8); }

/* pressure_solver::_compute_b_vector src/pressure_solver.cpp:180-242. */
static void rhs(const orc_ctx *c, double *b) {
	const double scale = 1.0 / c->h;
	for (size_t i = 0; i < c->nsys; ++i) {
		const size_t *p = c->fc[i];
		const orc_cell *me = &c->grid[raw_of(c, p[0], p[1], p[2])];
		double v = -(me->vel[0] + me->vel[1] + me->vel[2]);
		for (int d = 0; d < 3; ++d) {
			if (p[d] > 0) {
				size_t q[3] = {p[0], p[1], p[2]}; q[d]--;
				const orc_cell *nb = &c->grid[raw_of(c, q[0], q[1], q[2])];
				v += nb->vel[d];
				if (nb->type == T_SOLID) v -= nb->vel[d]; /* add then subtract, :191-194 */
			}
		}
		for (int d = 0; d < 3; ++d) {
			size_t q[3] = {p[0], p[1], p[2]}; q[d]++;
			if (type_at(c, q[0], q[1], q[2]) == T_SOLID) v += me->vel[d];
		}
		b[i] = scale * v;
	}
}
void orc_get_b(void *h, double *out) { rhs((orc_ctx *)h, out); }

/* pressure_solver::_apply_preconditioner src/pressure_solver.cpp:296-332. */
static void mic_apply(const orc_ctx *c, double *z, double *q, const double *r) {
	size_t n = c->nsys;
	for (size_t i = 0; i < n; ++i) {
		double t = 0.0;
		for (int d = 0; d < 3; ++d) {
			uint64_t j = nb_neg(c, c->fc[i], d);
			if (j != NOT_FLUID) t += bit_d(c->abits[j], d) * c->precon[j] * q[j];
		}
		q[i] = (r[i] + c->a_scale * t) * c->precon[i];
	}
	for (size_t i = n; i > 0;) {
		--i;
		double t = 0.0;
		for (int d = 0; d < 3; ++d) {
			uint64_t j = nb_pos(c, c->fc[i], d);
			if (j != NOT_FLUID) t += bit_d(c->abits[i], d) * z[j];
		}
		z[i] = (q[i] + c->a_scale * c->precon[i] * t) * c->precon[i];
	}
}
/* pressure_solver::_apply_a src/pressure_solver.cpp:334-362. */
static void lap_apply(const orc_ctx *c, double *out, const double *v) {
	for (size_t i = 0; i < c->nsys; ++i) {
		double val = NONSOLID(c->abits[i]) * v[i];
		for (int d = 0; d < 3; ++d) {
			uint64_t j = nb_neg(c, c->fc[i], d);
			if (j != NOT_FLUID) val -= bit_d(c->abits[j], d) * v[j];
		}
		for (int d = 0; d < 3; ++d) {
			uint64_t j = nb_pos(c, c->fc[i], d);
			if (j != NOT_FLUID) val -= bit_d(c->abits[i], d) * v[j];
		}
		out[i] = c->a_scale * val;
	}
}
void orc_apply_precon(void *h, const double *r, double *z) {
	orc_ctx *c = (orc_ctx *)h;
	double *q = (double *)calloc(c->nsys ? c->nsys : 1, 8);
	zero_n(z, c->nsys * 8);
	mic_apply(c, z, q, r);
	free(q);
}
void orc_apply_a(void *h, const double *v, double *out) { lap_apply((orc_ctx *)h, out, v); }

/* ------------------------------------------------------------------------------------------ a14 .. a16 */
static double dotv(const double *a, const double *b, size_t n) { /* vec_ops::dynamic::dot vec.h:162-171 */
	double r = 0.0;
	for (size_t i = 0; i < n; ++i) r += a[i] * b[i];
	return r;
}
/* pressure_solver::solve src/pressure_solver.cpp:19-71 (PCG; SIGNED max stopping rule :54; ++i before break :56).
 * Like the reference, solve() rebuilds the system first. */
void orc_solve(void *hh, double dt, double *p, double *residual, uint64_t *iters) {
	orc_ctx *c = (orc_ctx *)hh;
	orc_build_system(c, dt);
	size_t n = c->nsys;
	double *b = (double *)malloc((n ? n : 1) * 8);
	rhs(c, b);
	for (size_t i = 0; i < n; ++i) p[i] = 0.0;
	*residual = 0.0; *iters = 0;
	double tot = 0.0;
	for (size_t i = 0; i < n; ++i) tot += b[i] * b[i];
	if (tot < 1e-6) { free(b); return; } /* :33-35 */
	double *r = b;
	double *z = (double *)calloc(n, 8), *q = (double *)calloc(n, 8), *s = (double *)malloc(n * 8);
	mic_apply(c, z, q, r);
	copy_n(s, z, n * 8);
	double sigma_ps = dotv(z, r, n), res = 0.0;
	size_t i = 0;
	for (; i < c->maxit; ++i) {
		lap_apply(c, z, s);
		double alpha = sigma_ps / dotv(z, s, n);
		for (size_t k = 0; k < n; ++k) p[k] = p[k] + alpha * s[k];      /* _muladd :364-370 */
		for (size_t k = 0; k < n; ++k) r[k] = r[k] + (-alpha) * z[k];
		res = r[0];
		for (size_t k = 1; k < n; ++k) if (res < r[k]) res = r[k];      /* std::max_element, first max */
		if (res < c->tol) { ++i; break; }
		mic_apply(c, z, q, r);
		double sigma_new = dotv(z, r, n);
		double beta = sigma_new / sigma_ps;
		for (size_t k = 0; k < n; ++k) s[k] = z[k] + beta * s[k];
		sigma_ps = sigma_new;
	}
	*residual = res; *iters = i;
	free(b); free(z); free(q); free(s);
}

/* ---------------------------------------------------------------------------------------------------- a17 */
/* pressure_solver::apply_pressure src/pressure_solver.cpp:73-148. */
void orc_apply_pressure(void *hh, double dt, const double *p) {
	orc_ctx *c = (orc_ctx *)hh;
	const double coeff = dt / (c->density * c->h);
	for (size_t i = 0; i < c->nsys; ++i) {
		const size_t *pos = c->fc[i];
		double cur = p[i];
		orc_cell *cell = &c->grid[raw_of(c, pos[0], pos[1], pos[2])];
		for (int d = 0; d < 3; ++d) {
			size_t q[3] = {pos[0], pos[1], pos[2]}; q[d]++;
			int t = type_at(c, q[0], q[1], q[2]);
			if (t != T_SOLID) {
				double other = 0.0;
				if (t == T_FLUID) other = p[c->cell_to_unknown[raw_of(c, q[0], q[1], q[2])]];
				cell->vel[d] -= coeff * (other - cur);
			} else {
				cell->vel[d] = 0.0;
			}
		}
		for (int d = 0; d < 3; ++d) {
			if (pos[d] == 0) continue; /* wrapped index => get_cell() == nullptr, :124 */
			size_t q[3] = {pos[0], pos[1], pos[2]}; q[d]--;
			orc_cell *nb = &c->grid[raw_of(c, q[0], q[1], q[2])];
			if (nb->type == T_AIR) nb->vel[d] -= coeff * cur;
			else if (nb->type == T_SOLID) nb->vel[d] = 0.0;
		}
	}
}

/* ---------------------------------------------------------------------------------------------------- a18 */
/* simulation::_extrapolate_velocities src/simulation.cpp:685-754. */
void orc_extrapolate(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	size_t nc = ncells(c);
	uint8_t *valid = (uint8_t *)calloc(nc ? nc : 1, 1);
	size_t *fresh = (size_t *)malloc((nc ? nc : 1) * sizeof(size_t)), nfresh = 0;
	for (size_t i = 0; i < c->nsys; ++i) valid[raw_of(c, c->fc[i][0], c->fc[i][1], c->fc[i][2])] = 1;
	for (size_t it = 0; it < c->extrap_iters; ++it) {
		for (size_t k = 0; k < nfresh; ++k) valid[fresh[k]] = 1;
		nfresh = 0;
		for (size_t z = 0; z < c->n[2]; ++z)
			for (size_t y = 0; y < c->n[1]; ++y)
				for (size_t x = 0; x < c->n[0]; ++x) {
					size_t flat = raw_of(c, x, y, z);
					if (valid[flat]) continue;
					const size_t pos[3] = {x, y, z};
					size_t cnt = 0;
					double sum[3] = {0, 0, 0};
					int tpos[3] = {T_SOLID, T_SOLID, T_SOLID};
					for (int d = 0; d < 3; ++d) {
						if (pos[d] > 0) {
							size_t q[3] = {x, y, z}; q[d]--;
							size_t r = raw_of(c, q[0], q[1], q[2]);
							if (valid[r]) { for (int e = 0; e < 3; ++e) sum[e] += c->grid[r].vel[e]; ++cnt; }
						}
						if (pos[d] + 1 < c->n[d]) {
							size_t q[3] = {x, y, z}; q[d]++;
							size_t r = raw_of(c, q[0], q[1], q[2]);
							if (valid[r]) {
								for (int e = 0; e < 3; ++e) sum[e] += c->grid[r].vel[e];
								tpos[d] = c->grid[r].type; ++cnt;
							}
						}
					}
					if (cnt > 0) {
						for (int d = 0; d < 3; ++d)
							if (c->grid[flat].type == tpos[d]) c->grid[flat].vel[d] = sum[d] / (double)cnt;
						fresh[nfresh++] = flat;
					}
				}
	}
	free(valid); free(fresh);
}

/* ------------------------------------------------------------------------------------------ a19, a20 */
typedef struct { double v[8][3]; double tmid[3]; } face_samples;

/* mac_grid::get_face_samples src/mac_grid.cpp:51-112 (+ _clamp :42-50).
 * v[k] with k = 4*iz + 2*iy + ix  <=>  v000,v001,v010,v011,v100,v101,v110,v111 (include/fluid/mac_grid.h:30-40). */
static void face_gather(const orc_ctx *c, const orc_cell *g, const size_t *gi, const double *t, face_samples *out) {
	double vels[3][3][3][3];
	for (size_t dz = 0; dz < 3; ++dz) {
		size_t vz = gi[2] + dz; int zc = 0;
		if (vz < 1) { vz = 1; zc = 1; } else if (vz >= c->n[2]) { vz = c->n[2]; zc = 1; }
		--vz;
		for (size_t dy = 0; dy < 3; ++dy) {
			size_t vy = gi[1] + dy; int yc = 0;
			if (vy < 1) { vy = 1; yc = 1; } else if (vy >= c->n[1]) { vy = c->n[1]; yc = 1; }
			--vy;
			for (size_t dx = 0; dx < 3; ++dx) {
				size_t vx = gi[0] + dx; int xc = 0;
				if (vx < 1) { vx = 1; xc = 1; } else if (vx >= c->n[0]) { vx = c->n[0]; xc = 1; }
				--vx;
				const double *vel = g[raw_of(c, vx, vy, vz)].vel;
				vels[dz][dy][dx][0] = xc ? 0.0 : vel[0];
				vels[dz][dy][dx][1] = yc ? 0.0 : vel[1];
				vels[dz][dy][dx][2] = zc ? 0.0 : vel[2];
			}
		}
	}
	size_t d[3] = {1, 1, 1};
	for (int a = 0; a < 3; ++a) {
		out->tmid[a] = t[a] - 0.5;
		if (out->tmid[a] < 0.0) { d[a] = 0; out->tmid[a] += 1.0; }
	}
	const size_t dx = d[0], dy = d[1], dz = d[2];
	for (size_t iz = 0; iz < 2; ++iz)
		for (size_t iy = 0; iy < 2; ++iy)
			for (size_t ix = 0; ix < 2; ++ix) {
				double *o = out->v[4 * iz + 2 * iy + ix];
				o[0] = vels[dz + iz][dy + iy][ix][0];
				o[1] = vels[dz + iz][iy][dx + ix][1];
				o[2] = vels[iz][dy + iy][dx + ix][2];
			}
}
/* lerp/bilerp/trilerp include/fluid/misc.h:20-36. */
static double lerp1(double a, double b, double t) { return a * (1.0 - t) + b * t; }
static double trilerp_c(const face_samples *s, int comp, double t1, double t2, double t3) {
	double lo = lerp1(lerp1(s->v[0][comp], s->v[1][comp], t3), lerp1(s->v[2][comp], s->v[3][comp], t3), t2);
	double hi = lerp1(lerp1(s->v[4][comp], s->v[5][comp], t3), lerp1(s->v[6][comp], s->v[7][comp], t3), t2);
	return lerp1(lo, hi, t1);
}
/* simulation::_grad_kernel src/simulation.cpp:215-224 and _calculate_c_vector :507-521. */
static void c_vector(const orc_ctx *c, const face_samples *s, int comp, double tx, double ty, double tz, double *out) {
	out[0] = out[1] = out[2] = 0.0;
	for (int k = 0; k < 8; ++k) {
		double p[3] = {tx - (double)(k & 1), ty - (double)((k >> 1) & 1), tz - (double)((k >> 2) & 1)};
		double sg[3], n[3];
		for (int a = 0; a < 3; ++a) { sg[a] = p[a] > 0.0 ? -1.0 : 1.0; n[a] = 1.0 - fabs(p[a]); }
		double gk[3] = {sg[0] * n[1] * n[2] / c->h, n[0] * sg[1] * n[2] / c->h, n[0] * n[1] * sg[2] / c->h};
		for (int a = 0; a < 3; ++a) out[a] = out[a] + gk[a] * s->v[k][comp];
	}
}
/* particle::compute_cell_index_and_position src/simulation.cpp:17-23 (trunc via cast, true division). */
static void cell_and_frac(const orc_ctx *c, const orc_particle *p, size_t *gi, double *t) {
	for (int d = 0; d < 3; ++d) {
		double fi = (p->pos[d] - c->off[d]) / c->h;
		gi[d] = (size_t)fi;
		t[d] = fi - (double)gi[d];
	}
}
static void sample_vel(const face_samples *s, const double *t, double *v) {
	v[0] = trilerp_c(s, 0, s->tmid[2], s->tmid[1], t[0]);
	v[1] = trilerp_c(s, 1, s->tmid[2], t[1], s->tmid[0]);
	v[2] = trilerp_c(s, 2, t[2], s->tmid[1], s->tmid[0]);
}
/* simulation::_transfer_from_grid src/simulation.cpp:548-560; _pic :447-461; _flip :463-505; _apic :523-546. */
void orc_g2p(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	for (size_t i = 0; i < c->np; ++i) {
		orc_particle *p = &c->p[i];
		size_t gi[3]; double t[3];
		face_samples s;
		cell_and_frac(c, p, gi, t);
		if (c->method == M_FLIP) {
			face_samples so;
			double vo[3], vn[3];
			face_gather(c, c->old_grid, gi, t, &so);
			face_gather(c, c->grid, gi, t, &s);
			sample_vel(&so, t, vo);
			sample_vel(&s, t, vn);
			for (int d = 0; d < 3; ++d) p->vel[d] = vn[d] + (p->vel[d] - vo[d]) * c->blend;
			continue;
		}
		face_gather(c, c->grid, gi, t, &s);
		sample_vel(&s, t, p->vel);
		if (c->method == M_APIC) {
			c_vector(c, &s, 0, t[0], s.tmid[1], s.tmid[2], p->cx);
			c_vector(c, &s, 1, s.tmid[0], t[1], s.tmid[2], p->cy);
			c_vector(c, &s, 2, s.tmid[0], s.tmid[1], t[2], p->cz);
		}
	}
}

/* ---------------------------------------------------------------------------------------------------- a21 */
/* simulation::cfl src/simulation.cpp:199-205 (returns +inf when all velocities are zero). */
double orc_cfl(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	double m = 0.0;
	for (size_t i = 0; i < c->np; ++i) {
		const double *v = c->p[i].vel;
		double l = 0.0;
		l += v[0] * v[0]; l += v[1] * v[1]; l += v[2] * v[2];
		if (m < l) m = l;
	}
	return c->h / sqrt(m);
}

/* One pass of the hot path in the reference's order (src/simulation.cpp:62-66,72-78,83-104,119-121, without the
 * out-of-scope advect/collide/correct stages). Used by bench.py's cpu_baseline leg and by the parity tests. */
void orc_hot_step(void *hh, double dt, double *p_out, double *residual, uint64_t *iters) {
	orc_ctx *c = (orc_ctx *)hh;
	orc_hash(c);
	orc_p2g(c);
	orc_add_gravity(c, dt);
	double *p = p_out ? p_out : (double *)malloc((c->nfluid ? c->nfluid : 1) * 8);
	double res; uint64_t it;
	orc_solve(c, dt, p, &res, &it);
	orc_apply_pressure(c, dt, p);
	orc_extrapolate(c);
	orc_g2p(c);
	if (residual) *residual = res;
	if (iters) *iters = it;
	if (!p_out) free(p);
}

/* =================================================================================================================
 * "Next" rows of SURVEY.md 8(f), rank 1: the per-step particle stages either side of the hot path. Same status as the
 * rest of this file: test infrastructure, pinned against oracle/_ref (tests/test_oracle.py).
 * ================================================================================================================= */

/* fluid sources: simulation::sources (include/fluid/simulation.h:179), data_structures/source.h:12-22. */
void orc_clear_sources(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	for (size_t i = 0; i < c->nsrc; ++i) free(c->src[i].xyz);
	free(c->src);
	c->src = NULL;
	c->nsrc = 0;
}
void orc_add_source(void *hh, const int *xyz, size_t k, const double *vel, size_t root, int active, int coerce) {
	orc_ctx *c = (orc_ctx *)hh;
	c->src = (struct orc_source *)realloc(c->src, (c->nsrc + 1) * sizeof *c->src);
	struct orc_source *s = &c->src[c->nsrc++];
	s->xyz = (int *)malloc((k ? k : 1) * 12);
	copy_n(s->xyz, xyz, k * 12);
	s->k = k; s->root = root; s->active = active; s->coerce = coerce;
	memcpy(s->vel, vel, 24);
}
static double orc_draw(orc_ctx *c) { /* U[0,1): splitmix64 */
	uint64_t x = (c->rng += 0x9E3779B97F4A7C15ull);
	x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
	return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
/* simulation::seed_cell src/simulation.cpp:136-151: tops the cell up to density^3 particles at uniformly random positions
 * inside it, velocity = the source's; the cell's count in the space hash becomes the target. The positions are random by
 * design (the reference draws from pcg32 in an unspecified argument order): parity = counts and velocities. */
static void orc_seed_cell(orc_ctx *c, size_t x, size_t y, size_t z, const double *vel, size_t dens) {
	const size_t r = raw_of(c, x, y, z), target = dens * dens * dens;
	size_t num = c->hcount[r];
	for (; num < target; ++num) {
		if (c->np == c->pcap) {
			c->pcap = c->pcap ? 2 * c->pcap : 1024;
			c->p = (orc_particle *)realloc(c->p, c->pcap * sizeof(orc_particle));
			c->ptmp = (orc_particle *)realloc(c->ptmp, c->pcap * sizeof(orc_particle));
		}
		orc_particle *p = &c->p[c->np++];
		memset(p, 0, sizeof *p);
		const size_t cell[3] = {x, y, z};
		for (int d = 0; d < 3; ++d) {
			p->pos[d] = c->off[d] + (double)cell[d] * c->h + orc_draw(c) * c->h;
			p->old_pos[d] = p->pos[d];
			p->vel[d] = vel[d];
		}
		p->raw = r;
	}
	c->hcount[r] = target; /* unconditional, :150: a cell that held MORE than the target is recorded as holding the target, so a
	                        * later source with a larger target tops it up from there (not from its real count) */
}
/* simulation::_update_sources src/simulation.cpp:756-765, followed by the hash_particles of time_step (:64). */
void orc_update_sources(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	for (size_t i = 0; i < c->nsrc; ++i) {
		const struct orc_source *s = &c->src[i];
		if (!s->active) continue;
		for (size_t k = 0; k < s->k; ++k)
			orc_seed_cell(c, (size_t)s->xyz[3 * k], (size_t)s->xyz[3 * k + 1], (size_t)s->xyz[3 * k + 2], s->vel, s->root);
	}
	orc_hash(c);
}

/* simulation::_advect_particles src/simulation.cpp:226-249: velocity coercion inside the cells of active coercing
 * sources (by the space hash of the step's first hash), then x += v dt clamped to the skin. */
void orc_advect(void *hh, double dt) {
	orc_ctx *c = (orc_ctx *)hh;
	for (size_t i = 0; i < c->nsrc; ++i) {
		const struct orc_source *s = &c->src[i];
		if (!s->active || !s->coerce) continue;
		for (size_t k = 0; k < s->k; ++k) {
			const size_t r = raw_of(c, (size_t)s->xyz[3 * k], (size_t)s->xyz[3 * k + 1], (size_t)s->xyz[3 * k + 2]);
			orc_particle *q = c->p + c->hbegin[r];
			for (uint64_t j = 0; j < c->hcount[r]; ++j, ++q) {
				memcpy(q->vel, s->vel, 24);
				memset(q->cx, 0, 72); /* cx = cy = cz = 0, :234 */
			}
		}
	}
	double lo[3], hi[3];
	for (int d = 0; d < 3; ++d) {
		lo[d] = c->off[d] + c->skin;
		hi[d] = c->h * (double)c->n[d] + c->off[d] - c->skin;
	}
	for (size_t i = 0; i < c->np; ++i) {
		orc_particle *p = &c->p[i];
		for (int d = 0; d < 3; ++d) {
			p->pos[d] += p->vel[d] * dt;
			/* std::clamp(v, lo, hi) */
			p->pos[d] = p->pos[d] < lo[d] ? lo[d] : (hi[d] < p->pos[d] ? hi[d] : p->pos[d]);
		}
	}
}

static int solid_or_outside(const orc_ctx *c, int x, int y, int z) {
	if (x < 0 || y < 0 || z < 0) return 1;
	if ((size_t)x >= c->n[0] || (size_t)y >= c->n[1] || (size_t)z >= c->n[2]) return 1;
	return c->grid[raw_of(c, (size_t)x, (size_t)y, (size_t)z)].type == T_SOLID;
}

/* simulation::_detect_collisions src/simulation.cpp:612-683 with grid::march_cells grid.h:140-209, followed by
 * old_position = position (:57-59 / :115-117). */
void orc_detect_collisions(void *hh) {
	orc_ctx *c = (orc_ctx *)hh;
	const double h = c->h, skin = c->skin;
	for (size_t pi = 0; pi < c->np; ++pi) {
		orc_particle *p = &c->p[pi];
		double from[3] = {p->old_pos[0], p->old_pos[1], p->old_pos[2]}, to[3] = {p->pos[0], p->pos[1], p->pos[2]};
		for (int bounce = 0; bounce < 3; ++bounce) {
			int hit = 0;
			double a[3], b[3], diff[3], inv[3], t[3];
			int cur[3], last[3], adv[3];
			for (int d = 0; d < 3; ++d) {
				a[d] = (from[d] - c->off[d]) / h;
				b[d] = (to[d] - c->off[d]) / h;
				cur[d] = (int)floor(a[d]);
				last[d] = (int)floor(b[d]);
				diff[d] = b[d] - a[d];
				adv[d] = diff[d] > 0.0 ? 1 : -1;
				inv[d] = 1.0 / fabs(diff[d]);
				t[d] = fabs((double)(cur[d] + (diff[d] > 0.0 ? 1 : 0)) - a[d]) * inv[d];
			}
			while (cur[0] != last[0] || cur[1] != last[1] || cur[2] != last[2]) {
				int dim = 0;
				double tmin = 2.0;
				for (int d = 0; d < 3; ++d) if (t[d] < tmin) { tmin = t[d]; dim = d; }
				if (!(tmin <= 1.0)) break; /* "emergency break", grid.h:196-199 */
				cur[dim] += adv[dim];
				if (solid_or_outside(c, cur[0], cur[1], cur[2])) {
					/* normal = -advance on `dim`; dot(to - from, normal) */
					double dn = (to[dim] - from[dim]) * (double)(-adv[dim]);
					double tt = t[dim] + skin / dn;
					if (tt < 0.0) tt = 0.0;
					for (int d = 0; d < 3; ++d) from[d] = tt * to[d] + (1.0 - tt) * from[d];
					to[dim] = from[dim];
					hit = 1;
					break;
				}
				t[dim] += inv[dim];
			}
			if (!hit) break;
		}
		for (int d = 0; d < 3; ++d) p->pos[d] = to[d];
		/* skin of neighbouring solid cells / walls (:654-681) */
		double gp[3], cp[3];
		int ci[3];
		for (int d = 0; d < 3; ++d) {
			gp[d] = p->pos[d] - c->off[d];
			ci[d] = (int)(size_t)(gp[d] / h);
			cp[d] = gp[d] - (double)ci[d] * h;
		}
		const double skin_max = h - skin;
		for (int d = 0; d < 3; ++d) {
			if (cp[d] < skin) {
				int q[3] = {ci[0], ci[1], ci[2]};
				q[d] -= 1;
				if (ci[d] == 0 || solid_or_outside(c, q[0], q[1], q[2])) p->pos[d] += skin - cp[d];
			}
			if (cp[d] > skin_max) {
				int q[3] = {ci[0], ci[1], ci[2]};
				q[d] += 1;
				if ((size_t)(ci[d] + 1) >= c->n[d] || solid_or_outside(c, q[0], q[1], q[2])) p->pos[d] += skin_max - cp[d];
			}
		}
		for (int d = 0; d < 3; ++d) p->old_pos[d] = p->pos[d];
	}
}

/* simulation::_correct_positions src/simulation.cpp:562-610. Requires the space hash of the current positions.
 * The coincident-particle jitter (:584-587) is random by design in the reference (std::random_device); here such a
 * pair contributes nothing, and the fixtures contain no coincident particles. */
void orc_correct_positions(void *hh, double dt) {
	orc_ctx *c = (orc_ctx *)hh;
	const double re = c->h / sqrt(2.0);
	double (*moved)[3] = (double(*)[3])malloc((c->np ? c->np : 1) * sizeof(double[3]));
	for (size_t i = 0; i < c->np; ++i) {
		const orc_particle *p = &c->p[i];
		size_t ci[3];
		for (int d = 0; d < 3; ++d) ci[d] = (size_t)((p->pos[d] - c->off[d]) / c->h); /* compute_cell_index :13-15 */
		double spring[3] = {0, 0, 0};
		size_t x0 = ci[0] < 1 ? 0 : ci[0] - 1, y0 = ci[1] < 1 ? 0 : ci[1] - 1, z0 = ci[2] < 1 ? 0 : ci[2] - 1;
		size_t x1 = ci[0] + 2 < c->n[0] ? ci[0] + 2 : c->n[0], y1 = ci[1] + 2 < c->n[1] ? ci[1] + 2 : c->n[1],
		       z1 = ci[2] + 2 < c->n[2] ? ci[2] + 2 : c->n[2];
		for (size_t z = z0; z < z1; ++z)
			for (size_t y = y0; y < y1; ++y)
				for (size_t x = x0; x < x1; ++x) {
					size_t r = raw_of(c, x, y, z);
					const orc_particle *o = c->p + c->hbegin[r];
					for (uint64_t k = 0; k < c->hcount[r]; ++k, ++o) {
						if (o == p) continue;
						double off[3] = {p->pos[0] - o->pos[0], p->pos[1] - o->pos[1], p->pos[2] - o->pos[2]};
						double d2 = 0.0;
						d2 += off[0] * off[0]; d2 += off[1] * off[1]; d2 += off[2] * off[2];
						if (d2 < 1e-12) continue;
						double kl = 1.0 - d2 / (re * re), w = 0.0;
						if (kl > 0.0) w = kl * kl * kl;
						double f = w / sqrt(d2);
						for (int d = 0; d < 3; ++d) spring[d] += f * off[d];
					}
				}
		const double s = dt * c->stiffness * re;
		for (int d = 0; d < 3; ++d) moved[i][d] = p->pos[d] + spring[d] * s;
	}
	for (size_t i = 0; i < c->np; ++i)
		for (int d = 0; d < 3; ++d) {
			double hi = c->off[d] + (double)c->n[d] * c->h, v = moved[i][d];
			c->p[i].pos[d] = v < c->off[d] ? c->off[d] : (hi < v ? hi : v);
		}
	free(moved);
}

/* simulation::time_step(dt) src/simulation.cpp:43-125 without callbacks. */
void orc_time_step(void *hh, double dt, double *residual, uint64_t *iters) {
	orc_ctx *c = (orc_ctx *)hh;
	orc_hash(c);
	orc_advect(c, dt);
	orc_detect_collisions(c);
	orc_hash(c);
	if (c->nsrc) orc_update_sources(c);
	orc_p2g(c);
	orc_add_gravity(c, dt);
	double *p = (double *)malloc((c->nfluid ? c->nfluid : 1) * 8), res;
	uint64_t it;
	orc_solve(c, dt, p, &res, &it);
	orc_apply_pressure(c, dt, p);
	orc_correct_positions(c, dt);
	orc_detect_collisions(c);
	orc_extrapolate(c);
	orc_g2p(c);
	if (residual) *residual = res;
	if (iters) *iters = it;
	free(p);
}

/* simulation::update(dt) src/simulation.cpp:31-41: CFL sub-stepping; returns the number of time steps, their lengths in dts. */
size_t orc_update(void *hh, double dt, double *dts, size_t cap) {
	orc_ctx *c = (orc_ctx *)hh;
	size_t n = 0;
	for (;;) {
		const double ts = 3.0 * orc_cfl(c); /* cfl_number = 3, include/fluid/simulation.h:182 */
		const double step = ts > dt ? dt : ts;
		if (dts && n < cap) dts[n] = step;
		++n;
		orc_time_step(c, step, NULL, NULL);
		if (ts > dt) break;
		dt -= ts;
	}
	return n;
}
