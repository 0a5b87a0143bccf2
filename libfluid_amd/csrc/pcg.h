// libfluid_amd/csrc/pcg.h -- shared constants of the pressure-solve kernels.
#pragma once
#include "common.h"

// Every reduction of the solve is deterministic: workgroup g writes one partial to partials[PART_x + g], and every
// consumer workgroup re-adds the <= PCG_MAX_GRID partials in the same fixed order. No atomics, no host round trip.
#define PCG_MAX_GRID 2048
#define PCG_WAVES 4          // waves (= tiles in flight) per workgroup
// offsets in doubles; one spare slot after the PCG_MAX_GRID workgroup partials (coarse share of sigma)
enum { PART_STRIDE = PCG_MAX_GRID + 64, PART_ZS = 0, PART_RMAX = PART_STRIDE, PART_SIG0 = 2 * PART_STRIDE,
       PART_SIG1 = 3 * PART_STRIDE, PART_B2 = 4 * PART_STRIDE, PART_TOTAL = 5 * PART_STRIDE };

/// Workgroups of the wave-per-tile kernels of the solve (each wave walks tiles slot, slot + 4 G, ...). LFA_PCG_GRID_CAP (read once,
/// process-wide: the partial-sum layout of every handle depends on it) lowers the cap for experiments.
extern int lfa_pcg_grid_cap;  // core.hip
static inline int pcg_grid(int n_ptiles) {
	int g = (n_ptiles + PCG_WAVES - 1) / PCG_WAVES;
	if (g < 1) g = 1;
	return g < lfa_pcg_grid_cap ? g : lfa_pcg_grid_cap;
}

/// The same without the tuning cap: up to PCG_MAX_GRID workgroups (kernels that run once per solve and stream).
static inline int pcg_grid_uncapped(int n_ptiles) {
	int g = (n_ptiles + PCG_WAVES - 1) / PCG_WAVES;
	if (g < 1) g = 1;
	return g < PCG_MAX_GRID ? g : PCG_MAX_GRID;
}

int lfa_build_rhs(lfa_sim *s, double dt);
int lfa_p2g_run(lfa_sim *s, bool fuse_gravity, double dt);
int lfa_dist_refresh_grid(lfa_sim *s, bool with_topology);

// multigrid preconditioner (mg.hip)
int lfa_mg_setup(lfa_sim *s);                       // hierarchy for the current unknown set (after lfa_build_rhs)
int lfa_mg_apply(lfa_sim *s, double *part_sigma);   // vz = V(vr) / scale (vq is scratch), partials of dot(z, r)
// the AXPYs of a PCG iteration fused with the pre-smoothing of the finest level, then the rest of the V-cycle
int lfa_mg_axpy_apply(lfa_sim *s, const void *sdir, const double *part_sigma, int n_sigma, const double *part_qs, int n_qs,
                      double *part_rmax, double *part_sigma_new);
// slab runs: the single-reduction form (gamma = z.r, delta = (A z).z and max r in one collective; mg.hip: k_mg_axpy_presmooth_cg)
int lfa_mg_axpy_apply_cg(lfa_sim *s, const double *gamma, int n_gamma, const double *gamma_old, int n_gamma_old, const double *delta,
                         int n_delta, const double *rmax_prev, int n_rmax, int iter, double *alpha_io, double *part_rmax,
                         double *part_sigma_new);
int lfa_mg_bench_part(lfa_sim *s, int part);
int lfa_mg_level0(const lfa_sim *s, const int **tiles, const int **slot);  // the tiles the PCG iterates over (closed ones left out)
int lfa_mg_solve_closed(lfa_sim *s, void *out = nullptr);                  // the closed tiles' own solves of A x = vr (lfa_sim::tile_closed) -> out (null: the pressure)
void lfa_mg_free(lfa_sim *s);
void lfa_mg_stats(const lfa_sim *s, uint64_t *launches_per_cycle, uint64_t *levels, uint64_t *first_co);
