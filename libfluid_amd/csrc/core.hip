// libfluid_amd/csrc/core.hip -- handle lifetime, parameters, device scan, particle/grid boundary conversion and the
// particle binning stage (reference rows a1, a2, a21, a22 of SURVEY.md section 8).
#include <math.h>

#include <algorithm>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "pcg.h"
#include "stream_set.h"

static thread_local std::string g_create_error;

int lfa_fail(lfa_sim *s, int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (s) s->err = buf;
	else g_create_error = buf;
	return code;
}

// =============================================================================================== exclusive scan
namespace {
constexpr int SCAN_BS = 256, SCAN_IPT = 8, SCAN_TILE = SCAN_BS * SCAN_IPT;

__device__ inline uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds, uint32_t &total) {
	const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
	uint32_t incl = v;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		uint32_t t = __shfl_up(incl, o, 64);
		if (lane >= o) incl += t;
	}
	if (lane == 63) lds[wid] = incl;
	__syncthreads();
	uint32_t woff = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < SCAN_BS / 64; ++i) {
		uint32_t x = lds[i];
		if (i < wid) woff += x;
		tot += x;
	}
	__syncthreads();
	total = tot;
	return woff + incl - v;
}

__global__ void __launch_bounds__(SCAN_BS) k_scan_reduce(const uint32_t *in, uint32_t *block_sums, size_t n) {
	__shared__ uint32_t lds[SCAN_BS / 64];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_IPT;
	uint32_t sum = 0;
#pragma unroll
	for (int k = 0; k < SCAN_IPT; ++k)
		if (base + k < n) sum += in[base + k];
	sum = wave_sum(sum);
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = sum;
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t t = 0;
		for (int i = 0; i < SCAN_BS / 64; ++i) t += lds[i];
		block_sums[blockIdx.x] = t;
	}
}

__global__ void __launch_bounds__(SCAN_BS)
k_scan_local(const uint32_t *in, uint32_t *out, const uint32_t *block_offsets, size_t n, uint32_t *total_out) {
	__shared__ uint32_t lds[SCAN_BS / 64];
	size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_IPT;
	uint32_t x[SCAN_IPT], sum = 0;
#pragma unroll
	for (int k = 0; k < SCAN_IPT; ++k) {
		x[k] = base + k < n ? in[base + k] : 0u;
		sum += x[k];
	}
	uint32_t total;
	uint32_t off = block_exclusive_scan(sum, lds, total) + (block_offsets ? block_offsets[blockIdx.x] : 0u);
#pragma unroll
	for (int k = 0; k < SCAN_IPT; ++k) {
		if (base + k < n) out[base + k] = off;
		off += x[k];
	}
	if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BS - 1) *total_out = off;
}
/// A few tiles (n <= SCAN_SERIAL_MAX) by ONE workgroup, tile after tile with a running offset: one launch instead of three (reduce,
/// scan of the block sums, local scan) - at this size each of them is its launch latency (the tile arrays of a 128^3 grid: 4 096).
constexpr size_t SCAN_SERIAL_MAX = 4 * SCAN_TILE;  // (8 tiles = 16 384 entries, C4's particle-tile arrays, are already slower this way than in three launches)
__global__ void __launch_bounds__(SCAN_BS) k_scan_serial(const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_out) {
	__shared__ uint32_t lds[SCAN_BS / 64];
	uint32_t carry = 0;
	for (size_t tile0 = 0; tile0 < n; tile0 += SCAN_TILE) {
		const size_t base = tile0 + (size_t)threadIdx.x * SCAN_IPT;
		uint32_t x[SCAN_IPT], sum = 0;
#pragma unroll
		for (int k = 0; k < SCAN_IPT; ++k) {
			x[k] = base + k < n ? in[base + k] : 0u;
			sum += x[k];
		}
		uint32_t total;
		uint32_t off = carry + block_exclusive_scan(sum, lds, total);
#pragma unroll
		for (int k = 0; k < SCAN_IPT; ++k) {
			if (base + k < n) out[base + k] = off;
			off += x[k];
		}
		carry += total;
	}
	if (total_out && threadIdx.x == 0) *total_out = carry;
}
}  // namespace

static int scan_rec(lfa_sim *s, const uint32_t *in, uint32_t *out, size_t n, uint32_t *tmp, uint32_t *total_dev) {
	size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
	if (nb > 1 && n <= SCAN_SERIAL_MAX) {
		hipLaunchKernelGGL(k_scan_serial, dim3(1), dim3(SCAN_BS), 0, s->stream, in, out, n, total_dev);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	if (nb == 1) {
		hipLaunchKernelGGL(k_scan_local, dim3(1), dim3(SCAN_BS), 0, s->stream, in, out, (const uint32_t *)nullptr, n,
		                   total_dev);
		LFA_LAUNCH_CHECK(s);
		return LFA_OK;
	}
	hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_BS), 0, s->stream, in, tmp, n);
	LFA_LAUNCH_CHECK(s);
	LFA_TRY(scan_rec(s, tmp, tmp, nb, tmp + ((nb + 3) & ~(size_t)3), nullptr));
	hipLaunchKernelGGL(k_scan_local, dim3((unsigned)nb), dim3(SCAN_BS), 0, s->stream, in, out, (const uint32_t *)tmp, n,
	                   total_dev);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

int lfa_exclusive_scan_u32(lfa_sim *s, const uint32_t *in, uint32_t *out, size_t n, uint32_t *total_dev) {
	if (n == 0) {
		if (total_dev) LFA_HIP(s, hipMemsetAsync(total_dev, 0, 4, s->stream));
		return LFA_OK;
	}
	size_t need = 0;
	for (size_t m = n; m > SCAN_TILE;) {
		m = (m + SCAN_TILE - 1) / SCAN_TILE;
		need += (m + 3) & ~(size_t)3;
	}
	need += 16;
	if (need > s->scan_tmp_len) {
		if (s->scan_tmp) LFA_HIP(s, hipFree(s->scan_tmp));
		s->scan_tmp = nullptr;
		LFA_HIP(s, hipMalloc(&s->scan_tmp, need * 4));
		s->scan_tmp_len = need;
	}
	return scan_rec(s, in, out, n, s->scan_tmp, total_dev);
}

// =============================================================================================== lifetime
extern "C" void lfa_default_params(lfa_params *p) {
	memset(p, 0, sizeof *p);
	// include/fluid/simulation.h:179-190, include/fluid/pressure_solver.h:38-42
	p->cell_size = NAN;
	p->blending_factor = 1.0;
	p->density = 1.0;
	p->boundary_skin_width = 0.1;
	p->correction_stiffness = 5.0;
	p->cfl_number = 3.0;
	p->velocity_extrapolation_iterations = 1;
	p->simulation_method = LFA_APIC;
	p->tau = 0.97;
	p->sigma = 0.25;
	p->tolerance = 1e-6;
	p->max_iterations = 200;
	p->p2g_variant = LFA_P2G_LDS_BINNED;
	p->precond = LFA_PRECOND_MULTIGRID;
	p->pcg_dtype = LFA_PCG_F32;
	p->apic_unscaled_kernel = 1;
	p->pcg_fused = 1;
	p->pcg_warm_start = 0;
}

template <typename T> static int dev_alloc(lfa_sim *s, T **p, size_t count, bool zero) {
	*p = nullptr;
	if (count == 0) count = 1;
	hipError_t e = hipMalloc((void **)p, count * sizeof(T));
	if (e != hipSuccess)
		return lfa_fail(s, LFA_E_OOM, "hipMalloc of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
	if (zero) LFA_HIP(s, hipMemsetAsync(*p, 0, count * sizeof(T), s->stream));
	return LFA_OK;
}

__global__ void k_init_ctype(uint8_t *ctype, uint8_t *solid, GridDims g, size_t ncp) {
	size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= ncp) return;
	int tile = (int)(b >> 9), l = (int)(b & 511), tx, ty, tz;
	tile_coords(g, tile, tx, ty, tz);
	bool in = in_grid(g, tx * 8 + (l & 7), ty * 8 + ((l >> 3) & 7), tz * 8 + (l >> 6));
	ctype[b] = in ? CT_AIR : (CT_SOLID | CT_OUTSIDE);
	solid[b] = in ? 0 : 1;
}

/// The correction's stream yields to the main one: the solve's short kernels should not queue behind a long VALU-bound launch.
static hipError_t create_low_priority_stream(hipStream_t *st) {
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
	return hipStreamCreateWithPriority(st, hipStreamNonBlocking, least);
}

int lfa_pcg_grid_cap = 768;  // (pcg.h: pcg_grid; PCG_MAX_GRID = 2048 is the most the partial arrays hold. 3 workgroups per CU: measured at
                             // C4 / C3 / C2 per iteration: 2048: 0.282 / 0.138 / 0.076 ms, 768: 0.273 / 0.132 / 0.0755, 512: 0.308, 640: 0.289, 896: 0.292)
/// The process environment is read here and nowhere below an entry point (lfa_knobs, common.h).
static void lfa_knobs_parse(lfa_knobs &k) {
	auto flag = [](const char *name, int dflt) -> int {
		const char *e = getenv(name);
		return e ? (atoi(e) != 0 ? 1 : 0) : dflt;
	};
	auto num = [](const char *name, int dflt) -> int {
		const char *e = getenv(name);
		return e ? atoi(e) : dflt;
	};
	k.mg_no_persist = flag("LFA_MG_NO_PERSIST", 0);
	k.mg_dist_single = flag("LFA_MG_DIST_SINGLE", 0);
	k.mg_co_fault = num("LFA_MG_CO_FAULT", 0);
	k.mg_no_tagged = flag("LFA_MG_NO_TAGGED", 0);
	k.mg_no_top = flag("LFA_MG_NO_TOP", 0);
	k.mg_no_prune = flag("LFA_MG_NO_PRUNE", 0);
	k.mg_no_closed = flag("LFA_MG_NO_CLOSED", 0);
	k.mg_dist_levels = num("LFA_MG_DIST_LEVELS", 0);
	{
		static std::once_flag once;
		std::call_once(once, [&] {
			const int cap = num("LFA_PCG_GRID_CAP", lfa_pcg_grid_cap);  // (unset: the measured best above, not the array bound)
			lfa_pcg_grid_cap = cap < 1 ? 1 : (cap > 2048 ? 2048 : cap);  // (small caps: tests - a wave then walks hundreds of tiles)
		});
	}
}

extern "C" int lfa_create(lfa_sim **out, uint64_t nx, uint64_t ny, uint64_t nz, int device) {
	if (!out) return lfa_fail(nullptr, LFA_E_INVALID, "lfa_create: out is NULL");
	*out = nullptr;
	if (nx == 0 || ny == 0 || nz == 0 || nx > 4096 || ny > 4096 || nz > 4096)
		return lfa_fail(nullptr, LFA_E_INVALID, "lfa_create: grid size %llux%llux%llu out of range",
		                (unsigned long long)nx, (unsigned long long)ny, (unsigned long long)nz);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
		return lfa_fail(nullptr, LFA_E_NO_DEVICE, "lfa_create: no HIP device available (there is no CPU fallback)");
	if (device < 0) {
		if (hipGetDevice(&device) != hipSuccess) device = 0;
	}
	if (device >= ndev) return lfa_fail(nullptr, LFA_E_NO_DEVICE, "lfa_create: device %d of %d", device, ndev);
	if (hipSetDevice(device) != hipSuccess) return lfa_fail(nullptr, LFA_E_NO_DEVICE, "hipSetDevice(%d) failed", device);

	lfa_sim *s = new lfa_sim();
	s->device = device;
	lfa_knobs_parse(s->knobs);
	GridDims &g = s->g;
	g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz;
	g.ntx = (g.nx + 7) / 8; g.nty = (g.ny + 7) / 8; g.ntz = (g.nz + 7) / 8;
	g.nt = g.ntx * g.nty * g.ntz;
	s->nc = (size_t)nx * ny * nz;
	s->ncp = (size_t)g.nt * LFA_TILE_CELLS;
	if (s->ncp >= ((size_t)1 << 32)) {
		delete s;
		return lfa_fail(nullptr, LFA_E_INVALID, "lfa_create: more than 2^32 padded cells");
	}
	lfa_default_params(&s->prm);
	int rc = LFA_OK;
	auto chk = [&](int r) { if (rc == LFA_OK && r != LFA_OK) rc = r; };
	// streams, events and the pinned page: adopted from a destroyed handle when there is one (pool.hip)
	if (lfa_stream_set *q = lfa_pool_take_set(device)) {
		s->stream = q->stream; s->stream2 = q->stream2; s->stream3 = q->stream3;
		s->ev_fork = q->ev_fork; s->ev_join = q->ev_join; s->ev_cfork = q->ev_cfork; s->ev_cjoin = q->ev_cjoin;
		for (int i = 0; i < 48; ++i) s->ev[i] = q->ev[i];
		s->ev_created = q->ev_created;
		s->h_pinned = q->h_pinned;
		delete q;
	} else {
		if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
			delete s;
			return lfa_fail(nullptr, LFA_E_HIP, "hipStreamCreate failed");
		}
		if (hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess ||
		    create_low_priority_stream(&s->stream3) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_cfork, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_cjoin, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) != hipSuccess) {
			lfa_destroy(s);
			return lfa_fail(nullptr, LFA_E_HIP, "side stream / event creation failed");
		}
	}
	chk(dev_alloc(s, &s->tile_count, g.nt + 1, true));
	chk(dev_alloc(s, &s->tile_start, g.nt + 1, true));
	chk(dev_alloc(s, &s->tile_flag, g.nt + 1, true));
	chk(dev_alloc(s, &s->tile_scan, g.nt + 1, true));
	chk(dev_alloc(s, &s->grid_flag, g.nt + 1, true));
	chk(dev_alloc(s, &s->ptiles_all, g.nt, true));
	s->ptiles = s->ptiles_all;
	s->slab_lo = 0;
	s->slab_hi = g.ntz;
	chk(dev_alloc(s, &s->dtiles, g.nt, true));
	chk(dev_alloc(s, &s->tile_pslot, g.nt, true));
	chk(dev_alloc(s, &s->level_tiles, g.nt, true));
	chk(dev_alloc(s, &s->u, s->ncp, true));
	chk(dev_alloc(s, &s->v, s->ncp, true));
	chk(dev_alloc(s, &s->w, s->ncp, true));
	chk(dev_alloc(s, &s->ctype, s->ncp, true));
	chk(dev_alloc(s, &s->solid, s->ncp, true));
	chk(dev_alloc(s, &s->cell_count, s->ncp, true));
	chk(dev_alloc(s, &s->abits, s->ncp, true));
	chk(dev_alloc(s, &s->partials, 16384, true));  // >= PART_TOTAL (pcg.h)
	chk(dev_alloc(s, &s->pcg_state, 32, true));  // [0] done [1] NaN [2] a wait was given up [3] 1: zero right-hand side, 2: NaN in it ... [16,17] the final residual (double)
	chk(dev_alloc(s, &s->pcg_hist, 8192, true));  // [0,4096) residual history, [6144,..) coarse r2 hand-off
	if (rc == LFA_OK && !s->h_pinned && hipHostMalloc((void **)&s->h_pinned, 4096, hipHostMallocDefault) != hipSuccess)
		rc = lfa_fail(s, LFA_E_HIP, "hipHostMalloc failed");
	if (rc == LFA_OK) {
		hipLaunchKernelGGL(k_init_ctype, dim3((unsigned)((s->ncp + 255) / 256)), dim3(256), 0, s->stream, s->ctype,
		                   s->solid, s->g, s->ncp);
		if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess)
			rc = lfa_fail(s, LFA_E_HIP, "grid initialisation kernel failed (is the code object built for this GPU?)");
	}
	if (rc != LFA_OK) {
		g_create_error = s->err;
		lfa_destroy(s);
		return rc;
	}
	lfa_co_gate_handle(device, +1);
	s->gate_counted = true;
	*out = s;
	return LFA_OK;
}

static void free_soa(ParticleSoA &p) {
	if (p.base) (void)hipFree(p.base);
	p = ParticleSoA();
}

extern "C" void lfa_destroy(lfa_sim *s) {
	if (!s) return;
	(void)hipSetDevice(s->device);
	if (s->gate_counted) lfa_co_gate_handle(s->device, -1);
	if (s->stream) (void)hipStreamSynchronize(s->stream);
	if (s->stream2) (void)hipStreamSynchronize(s->stream2);
	if (s->stream3) (void)hipStreamSynchronize(s->stream3);
	// the handle's streams are idle: its blocks go back to the cache without a device-wide synchronisation each
	lfa_pool_nosync_begin();
	free_soa(s->pb[0]);
	free_soa(s->pb[1]);
	void *ptrs[] = {s->c_home, s->fine_start, s->corr_ovf, s->tile_clear, s->tile_epoch, s->tile_closed, s->src_cell, s->src_lo, s->src_target, s->src_of, s->src_need, s->src_vel, s->coerce_map, s->grid_flag, s->rank, s->vc_src, s->tile_count, s->tile_start, s->tile_flag, s->tile_scan, s->ptiles_all, s->dtiles, s->halo_tiles, s->dist_red,
	                s->xbuf[0], s->xbuf[1], s->xbuf[2], s->xbuf[3],
	                s->tile_pslot, s->scan_tmp, s->u, s->v, s->w, s->uo, s->vo, s->wo, s->ctype, s->solid,
	                s->cell_count, s->stage, s->acc, s->abits, s->vp, s->vr, s->vz, s->vs, s->vpre, s->vq, s->vs2, s->c_as, s->nbr_table,
	                s->partials, s->pcg_state, s->pcg_hist, s->level_tiles, s->io_buf, s->raw_scan, s->c_diag, s->c_w[0],
	                s->c_w[1], s->c_w[2], s->c_unk, s->c_pre, s->c_r, s->c_x, s->c_r2, s->c_x2, s->a2inv, s->slot_l1,
	                s->l1_tiles, s->l1_l2};
	for (void *p : ptrs)
		if (p) (void)hipFree(p);
	if (s->dist) delete s->dist;
	s->dist = nullptr;
	lfa_mg_free(s);
	lfa_pool_nosync_end();
	// streams, events and the pinned page are parked for the next lfa_create on this device (a handle whose creation failed
	// half way is torn down instead)
	if (s->stream && s->stream2 && s->stream3 && s->ev_fork && s->ev_join && s->ev_cfork && s->ev_cjoin && s->h_pinned) {
		lfa_stream_set *q = new lfa_stream_set();
		q->device = s->device;
		q->stream = s->stream; q->stream2 = s->stream2; q->stream3 = s->stream3;
		q->ev_fork = s->ev_fork; q->ev_join = s->ev_join; q->ev_cfork = s->ev_cfork; q->ev_cjoin = s->ev_cjoin;
		for (int i = 0; i < 48; ++i) q->ev[i] = s->ev[i];
		q->ev_created = s->ev_created;
		q->h_pinned = s->h_pinned;
		lfa_pool_park_set(q);
	} else {
		if (s->h_pinned) (void)hipHostFree(s->h_pinned);
		if (s->ev_created)
			for (auto &e : s->ev) (void)hipEventDestroy(e);
		if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
		if (s->ev_join) (void)hipEventDestroy(s->ev_join);
		if (s->ev_cfork) (void)hipEventDestroy(s->ev_cfork);
		if (s->ev_cjoin) (void)hipEventDestroy(s->ev_cjoin);
		if (s->stream3) (void)hipStreamDestroy(s->stream3);
		if (s->stream2) (void)hipStreamDestroy(s->stream2);
		if (s->stream) (void)hipStreamDestroy(s->stream);
	}
	delete s;
}

extern "C" const char *lfa_last_error(const lfa_sim *s) { return s ? s->err.c_str() : g_create_error.c_str(); }

extern "C" int lfa_set_params(lfa_sim *s, const lfa_params *p) {
	if (!s || !p) return LFA_E_INVALID;
	if (!(p->cell_size > 0.0)) return lfa_fail(s, LFA_E_INVALID, "cell_size must be > 0 (got %g)", p->cell_size);
	if (p->simulation_method < 0 || p->simulation_method > 2) return lfa_fail(s, LFA_E_INVALID, "bad simulation_method");
	if (p->velocity_extrapolation_iterations > 8)
		return lfa_fail(s, LFA_E_UNSUPPORTED, "velocity_extrapolation_iterations > 8 exceeds the 1-tile dilation");
	if (p->max_iterations > 4000) return lfa_fail(s, LFA_E_INVALID, "max_iterations > 4000");
	// every selector is validated before anything is invalidated
	if (p->precond < 0 || p->precond > LFA_PRECOND_MULTIGRID) return lfa_fail(s, LFA_E_INVALID, "bad precond %d", p->precond);
	if (p->pcg_dtype != LFA_PCG_F32 && p->pcg_dtype != LFA_PCG_F64) return lfa_fail(s, LFA_E_INVALID, "bad pcg_dtype %d", p->pcg_dtype);
	if (p->p2g_variant != LFA_P2G_LDS_BINNED && p->p2g_variant != LFA_P2G_GLOBAL_ATOMIC)
		return lfa_fail(s, LFA_E_INVALID, "bad p2g_variant %d", p->p2g_variant);
	if (!(p->density > 0.0)) return lfa_fail(s, LFA_E_INVALID, "density must be > 0 (got %g)", p->density);
	// (a rejected call has no side effects: everything above only reads). An unchanged parameter block - what the host class sends
	// before every step - leaves a correction in flight alone.
	if (memcmp(p, &s->prm, sizeof(lfa_params)) != 0) LFA_TRY(lfa_corr_commit(s));
	if (p->simulation_method != s->prm.simulation_method) {
		LFA_TRY(lfa_particles_materialize(s));  // what is deferred depends on it
		if (p->simulation_method == LFA_APIC) LFA_TRY(lfa_c_home_restore(s));  // APIC reads and writes C in particle order
	}
	if (p->pcg_dtype != s->prm.pcg_dtype || p->precond != s->prm.precond) {
		s->system_valid = false;
		s->pressure_epoch = 0;  // vp is about to be reinterpreted / re-allocated: no warm start from it
	}
	s->prm = *p;
	return LFA_OK;
}
extern "C" int lfa_get_params(const lfa_sim *s, lfa_params *p) {
	if (!s || !p) return LFA_E_INVALID;
	*p = s->prm;
	return LFA_OK;
}
extern "C" int lfa_synchronize(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	if (s->corr_in_flight) LFA_HIP(s, hipStreamSynchronize(s->stream3));  // (still to be joined: lfa_correct_collide_end)
	return LFA_OK;
}
extern "C" void *lfa_stream(lfa_sim *s) { return s ? (void *)s->stream : nullptr; }
extern "C" uint64_t lfa_num_particles(const lfa_sim *s) { return s ? s->np : 0; }

int lfa_ensure_io(lfa_sim *s, size_t bytes) {
	if (bytes <= s->io_cap) return LFA_OK;
	if (s->io_buf) LFA_HIP(s, hipFree(s->io_buf));
	s->io_buf = nullptr;
	s->io_cap = 0;
	hipError_t e = hipMalloc(&s->io_buf, bytes);
	if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc(%zu) for the boundary buffer failed", bytes);
	s->io_cap = bytes;
	return LFA_OK;
}

// =============================================================================================== particles in/out
static int alloc_soa(lfa_sim *s, ParticleSoA &p, size_t cap) {
	p = ParticleSoA();
	hipError_t e = hipMalloc(&p.base, cap * 17 * 4);
	if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of particle SoA (%zu particles) failed", cap);
	float *f = (float *)p.base;
	p.key = (uint32_t *)f;
	for (int k = 0; k < 3; ++k) p.t[k] = f + cap * (1 + k);
	for (int k = 0; k < 3; ++k) p.v[k] = f + cap * (4 + k);
	for (int k = 0; k < 9; ++k) p.c[k] = f + cap * (7 + k);
	p.id = (uint32_t *)(f + cap * 16);
	return LFA_OK;
}

/// (Re)allocates the particle arrays for n particles, DISCARDING their contents. Everything new is allocated before anything
/// old is released, so an out-of-memory failure leaves the handle with its old (still consistent) arrays and capacity.
int lfa_particles_alloc(lfa_sim *s, size_t n) {
	if (n <= s->pcap) return LFA_OK;
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	const size_t cap = (n + 1023) & ~(size_t)1023;
	ParticleSoA nb[2];
	uint32_t *rank = nullptr, *vc_src = nullptr;
	int rc = alloc_soa(s, nb[0], cap);
	if (rc == LFA_OK) rc = alloc_soa(s, nb[1], cap);
	if (rc == LFA_OK && hipMalloc(&rank, cap * 4) != hipSuccess) rc = lfa_fail(s, LFA_E_OOM, "hipMalloc of the rank array failed");
	if (rc == LFA_OK && hipMalloc(&vc_src, cap * 4) != hipSuccess) rc = lfa_fail(s, LFA_E_OOM, "hipMalloc of the source-index array failed");
	if (rc != LFA_OK) {
		free_soa(nb[0]);
		free_soa(nb[1]);
		if (rank) (void)hipFree(rank);
		if (vc_src) (void)hipFree(vc_src);
		return rc;
	}
	free_soa(s->pb[0]);
	free_soa(s->pb[1]);
	if (s->rank) (void)hipFree(s->rank);
	if (s->vc_src) (void)hipFree(s->vc_src);
	s->pb[0] = nb[0];
	s->pb[1] = nb[1];
	s->rank = rank;
	s->vc_src = vc_src;
	s->vc_pending = false;
	s->pcap = cap;
	return LFA_OK;
}

/// Grows the particle arrays to hold n_total particles, keeping the first n_keep of the current buffer. On failure the
/// resident particles stay where they are.
int lfa_particles_reserve(lfa_sim *s, size_t n_keep, size_t n_total) {
	if (n_total <= s->pcap) return LFA_OK;  // (a deferred binning stays deferred: nothing moves)
	LFA_TRY(lfa_particles_materialize(s));  // the reallocation below keeps the current buffer only
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	const size_t cap = ((n_total + n_total / 8) + 1023) & ~(size_t)1023;
	ParticleSoA keep = s->pb[s->cur];
	const size_t old_cap = s->pcap;
	s->pb[s->cur] = ParticleSoA();  // detached: the reallocation below must not free it
	s->pcap = 0;
	int rc = lfa_particles_alloc(s, cap);
	if (rc != LFA_OK) {  // nothing was released: put the live buffer back
		s->pb[s->cur] = keep;
		s->pcap = old_cap;
		return rc;
	}
	ParticleSoA &dst = s->pb[s->cur];
	const uint32_t *src_arrays[17];
	uint32_t *dst_arrays[17];
	src_arrays[0] = keep.key; dst_arrays[0] = dst.key;
	for (int k = 0; k < 3; ++k) { src_arrays[1 + k] = (uint32_t *)keep.t[k]; dst_arrays[1 + k] = (uint32_t *)dst.t[k]; }
	for (int k = 0; k < 3; ++k) { src_arrays[4 + k] = (uint32_t *)keep.v[k]; dst_arrays[4 + k] = (uint32_t *)dst.v[k]; }
	for (int k = 0; k < 9; ++k) { src_arrays[7 + k] = (uint32_t *)keep.c[k]; dst_arrays[7 + k] = (uint32_t *)dst.c[k]; }
	src_arrays[16] = keep.id; dst_arrays[16] = dst.id;
	for (int a = 0; a < 17; ++a)
		if (n_keep) LFA_HIP(s, hipMemcpyAsync(dst_arrays[a], src_arrays[a], n_keep * 4, hipMemcpyDeviceToDevice, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	free_soa(keep);
	return LFA_OK;
}

struct IngestParams {
	double off[3], h;
};

/// Position -> (clamped cell, in-cell fraction). The cell is computed in fp64 with a true division exactly like
/// src/simulation.cpp:253-257 (std::max(pos,0) -> size_t cast -> min(..., size-1)), so keys are bit-exact.
/// The fraction is fp32 in [0,1]; it is 1.0f only when the position lies on/over the max face (the reference's
/// unclamped index == size case, src/simulation.cpp:13-23), otherwise it is kept strictly below 1.
__device__ inline void cell_and_fraction(double pos, double off, double h, int n, int &cell, float &t) {
	double gp = (pos - off) / h;
	double m = gp < 0.0 ? 0.0 : gp;
	int c = m >= (double)n ? n - 1 : (int)m;
	if (c > n - 1) c = n - 1;
	double td = gp - (double)c;
	float tf = (float)td;
	if (!(tf > 0.0f)) tf = 0.0f;
	if (td < 1.0 && tf >= 1.0f) tf = 0.99999994f;
	if (tf > 1.0f) tf = 1.0f;
	cell = c;
	t = tf;
}

__global__ void k_ingest(const double *aos, size_t n, ParticleSoA p, GridDims g, IngestParams ip, int slab_lo, int slab_hi) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const double *q = aos + i * 19;
	int c[3];
	float t[3];
	cell_and_fraction(q[0], ip.off[0], ip.h, g.nx, c[0], t[0]);
	cell_and_fraction(q[1], ip.off[1], ip.h, g.ny, c[1], t[1]);
	cell_and_fraction(q[2], ip.off[2], ip.h, g.nz, c[2], t[2]);
	// a particle outside this rank's tile layers belongs to another rank: dropped by the binning (key = invalid)
	p.key[i] = ((c[2] >> 3) >= slab_lo && (c[2] >> 3) < slab_hi) ? blocked_index(g, c[0], c[1], c[2]) : 0xFFFFFFFFu;
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		p.t[k][i] = t[k];
		p.v[k][i] = (float)q[3 + k];
	}
#pragma unroll
	for (int k = 0; k < 9; ++k) p.c[k][i] = (float)q[6 + k];
	p.id[i] = (uint32_t)i;
}

extern "C" int lfa_upload_particles(lfa_sim *s, const void *aos152, uint64_t n) {
	if (!s || (!aos152 && n)) return LFA_E_INVALID;
	s->vc_pending = false;  // the particle set is replaced
	s->c_home_valid = false;
	s->move_pending = false;
	s->vmax2_valid = false;
	if (!(s->prm.cell_size > 0.0)) return lfa_fail(s, LFA_E_INVALID, "set cell_size (lfa_set_params) before uploading");
	if (n >= ((uint64_t)1 << 32)) return lfa_fail(s, LFA_E_INVALID, "more than 2^32 particles");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_particles_alloc(s, n));
	s->np = n;
	s->np_live = n;
	s->next_global_id = n;  // (slabs: every rank is handed the whole set, ids = upload index)
	s->binned = false;  // the grid stays what it was (G2P re-uploads corrected positions between apply and gather)
	s->system_valid = false;
	s->unknown_count_valid = false;
	s->cur = 0;
	if (n == 0) return LFA_OK;
	LFA_TRY(lfa_ensure_io(s, n * 152));
	LFA_HIP(s, hipMemcpyAsync(s->io_buf, aos152, n * 152, hipMemcpyHostToDevice, s->stream));
	IngestParams ip;
	for (int k = 0; k < 3; ++k) ip.off[k] = s->prm.grid_offset[k];
	ip.h = s->prm.cell_size;
	hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (const double *)s->io_buf,
	                   (size_t)n, s->pb[0], s->g, ip, s->slab_lo, s->slab_hi);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipStreamSynchronize(s->stream));  // the host buffer may be reused by the caller
	return LFA_OK;
}

/// Slabs after a hand-over: valid[i] = record i is a resident particle (not one that went to a neighbour rank); its exclusive
/// scan is the record's slot in a download, so holes never reach the host - and no binning (a collective on slabs) is needed.
__global__ void k_valid_flags(const uint32_t *key, size_t n, uint32_t *valid) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) valid[i] = key[i] != 0xFFFFFFFFu ? 1u : 0u;
}
__global__ void k_export_ids(const uint32_t *key, const uint32_t *id, const uint32_t *slot, size_t n, uint32_t *out) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && key[i] != 0xFFFFFFFFu) out[slot ? slot[i] : i] = id[i];
}

/// `old`: lfa_advect / lfa_correct have moved the particles and lfa_collide is still due - old_position is the position of before
/// the move (what a host callback between the two stages sees in the reference), kept in old.key / old.t.
__global__ void k_export(double *aos, size_t n, ParticleSoA p, GridDims g, IngestParams ip, int flags, int by_slot, const uint32_t *slot,
                         ParticleSoA old, int have_old, const float *c_home, size_t c_home_stride) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t b = p.key[i];
	if (slot && b == 0xFFFFFFFFu) return;
	// slab decomposition: particles migrate between ranks, so a rank's records come out in storage order (ids separately)
	double *q = aos + (by_slot ? (slot ? (size_t)slot[i] : i) : (size_t)p.id[i]) * 19;
	if (flags & LFA_DL_POSITIONS) {
		int tile = (int)(b >> 9), l = (int)(b & 511), tx, ty, tz;
		tile_coords(g, tile, tx, ty, tz);
		int c[3] = {tx * 8 + (l & 7), ty * 8 + ((l >> 3) & 7), tz * 8 + (l >> 6)};
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			double x = ip.off[k] + ((double)c[k] + (double)p.t[k][i]) * ip.h;
			q[k] = x;
			q[15 + k] = x;
		}
		if (have_old) {
			const uint32_t ob = old.key[i];
			int otile = (int)(ob >> 9), ol = (int)(ob & 511), ox, oy, oz;
			tile_coords(g, otile, ox, oy, oz);
			int oc[3] = {ox * 8 + (ol & 7), oy * 8 + ((ol >> 3) & 7), oz * 8 + (ol >> 6)};
#pragma unroll
			for (int k = 0; k < 3; ++k) q[15 + k] = ip.off[k] + ((double)oc[k] + (double)old.t[k][i]) * ip.h;
		}
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) q[3 + k] = (double)p.v[k][i];
#pragma unroll
	for (int k = 0; k < 9; ++k) q[6 + k] = (double)(c_home ? c_home[k * c_home_stride + p.id[i]] : p.c[k][i]);
	if (!(flags & LFA_DL_KEEP_RAW)) ((uint64_t *)q)[18] = raw_from_blocked(g, b);
}

/// Slabs with holes (particles handed over since the last binning): *slot = device array mapping record i of [0, np_live) to its
/// place among the resident ones (the binning's rank array, free between two binnings); nullptr when the records are dense.
static int lfa_slab_download_slots(lfa_sim *s, const uint32_t **slot) {
	*slot = nullptr;
	if (!s->dist || !s->holes || !s->np_live) return LFA_OK;
	hipLaunchKernelGGL(k_valid_flags, dim3((unsigned)((s->np_live + 255) / 256)), dim3(256), 0, s->stream, (const uint32_t *)s->pb[s->cur].key,
	                   s->np_live, s->rank);
	LFA_LAUNCH_CHECK(s);
	uint32_t *tot = (uint32_t *)(s->pcg_state + 5);
	LFA_TRY(lfa_exclusive_scan_u32(s, s->rank, s->rank, s->np_live, tot));
	uint32_t h = 0;
	LFA_HIP(s, hipMemcpyAsync(&h, tot, 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	if ((size_t)h != s->np) return lfa_fail(s, LFA_E_INVALID, "slab download: %u resident records but %zu expected", h, s->np);
	*slot = s->rank;
	return LFA_OK;
}

extern "C" int lfa_download_particles(lfa_sim *s, void *aos152, uint64_t n, int flags) {
	if (!s || (!aos152 && n)) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_particles_materialize(s));
	if (n != s->np) return lfa_fail(s, LFA_E_INVALID, "download of %llu particles but %zu are resident",
	                                (unsigned long long)n, s->np);
	if (n == 0) return LFA_OK;
	if (s->dist && !s->binned)
		return lfa_fail(s, LFA_E_INVALID, "slab decomposition: call lfa_hash_particles before downloading particles");
	LFA_TRY(lfa_corr_join(s));
	const uint32_t *slot = nullptr;
	LFA_TRY(lfa_slab_download_slots(s, &slot));
	LFA_TRY(lfa_ensure_io(s, n * 152));
	// start from the caller's records so fields the device does not own (positions unless asked) survive
	LFA_HIP(s, hipMemcpyAsync(s->io_buf, aos152, n * 152, hipMemcpyHostToDevice, s->stream));
	IngestParams ip;
	for (int k = 0; k < 3; ++k) ip.off[k] = s->prm.grid_offset[k];
	ip.h = s->prm.cell_size;
	// (slabs with holes: the records [0, np_live) include the leavers' holes, and the arrivals sit behind ceil256(np))
	const size_t n_rec = s->binned ? s->np_live : s->np;
	hipLaunchKernelGGL(k_export, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, s->stream, (double *)s->io_buf,
	                   n_rec, s->pb[s->cur], s->g, ip, flags, s->dist ? 1 : 0, slot, s->pb[s->cur ^ 1],
	                   (s->move_pending && !s->dist) ? 1 : 0, s->c_home_valid ? (const float *)s->c_home : (const float *)nullptr,
	                   s->c_home_cap);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(aos152, s->io_buf, n * 152, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

extern "C" int lfa_download_particle_ids(lfa_sim *s, uint32_t *ids, uint64_t n) {
	if (!s || (!ids && n)) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	if (n != s->np) return lfa_fail(s, LFA_E_INVALID, "download of %llu ids but %zu particles are resident", (unsigned long long)n, s->np);
	if (n == 0) return LFA_OK;
	LFA_TRY(lfa_corr_join(s));
	if (!s->dist) {  // single domain: record i of a download IS particle i
		for (uint64_t i = 0; i < n; ++i) ids[i] = (uint32_t)i;
		return LFA_OK;
	}
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "slab decomposition: call lfa_hash_particles before downloading ids");
	const uint32_t *slot = nullptr;
	LFA_TRY(lfa_slab_download_slots(s, &slot));
	if (slot) {
		LFA_TRY(lfa_ensure_io(s, n * 4));
		hipLaunchKernelGGL(k_export_ids, dim3((unsigned)((s->np_live + 255) / 256)), dim3(256), 0, s->stream, (const uint32_t *)s->pb[s->cur].key,
		                   (const uint32_t *)s->pb[s->cur].id, slot, s->np_live, (uint32_t *)s->io_buf);
		LFA_LAUNCH_CHECK(s);
		LFA_HIP(s, hipMemcpyAsync(ids, s->io_buf, n * 4, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		return LFA_OK;
	}
	LFA_HIP(s, hipMemcpyAsync(ids, s->pb[s->cur].id, n * 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

// ---- synthetic dam-break block, twin of libfluid_amd/scenes.py:seed_block
__device__ inline uint64_t splitmix64(uint64_t x) {
	x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
	x ^= x >> 27; x *= 0x94D049BB133111EBull;
	x ^= x >> 31;
	return x;
}
__global__ void k_seed_block(size_t n, size_t first, ParticleSoA p, GridDims g, IngestParams ip, int lox, int loy, int loz,
                             int ex, int ey, uint64_t seed, int global_ids) {
	size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const size_t i = j + first;  // index in the whole block
	size_t ci = i >> 3;
	int sub = (int)(i & 7);
	int cell[3] = {lox + (int)(ci % ex), loy + (int)((ci / ex) % ey), loz + (int)(ci / ((size_t)ex * ey))};
	int sb[3] = {sub & 1, (sub >> 1) & 1, (sub >> 2) & 1};
	int c[3];
	float t[3];
	const int nn[3] = {g.nx, g.ny, g.nz};
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		uint64_t ctr = 3ull * (uint64_t)i + (uint64_t)a;
		uint64_t x = seed + (ctr + 1ull) * 0x9E3779B97F4A7C15ull;
		double uu = (double)(splitmix64(x) >> 11) * (1.0 / 9007199254740992.0);
		double pos = ip.off[a] + ((double)cell[a] + ((double)sb[a] + uu) * 0.5) * ip.h;
		cell_and_fraction(pos, ip.off[a], ip.h, nn[a], c[a], t[a]);
	}
	p.key[j] = blocked_index(g, c[0], c[1], c[2]);
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		p.t[k][j] = t[k];
		p.v[k][j] = 0.0f;
	}
#pragma unroll
	for (int k = 0; k < 9; ++k) p.c[k][j] = 0.0f;
	p.id[j] = (uint32_t)(global_ids ? i : j);
}

extern "C" int lfa_seed_block(lfa_sim *s, const int64_t lo[3], const int64_t hi[3], uint64_t seed) {
	if (!s || !lo || !hi) return LFA_E_INVALID;
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_particles_materialize(s));
	if (!(s->prm.cell_size > 0.0)) return lfa_fail(s, LFA_E_INVALID, "set cell_size before seeding");
	const int nn[3] = {s->g.nx, s->g.ny, s->g.nz};
	for (int a = 0; a < 3; ++a)
		if (lo[a] < 0 || hi[a] > nn[a] || lo[a] >= hi[a]) return lfa_fail(s, LFA_E_INVALID, "seed block outside the grid");
	// with a slab decomposition every rank seeds the part of the block that lies in its own tile layers; the counter of
	// the generator is the particle's index in the WHOLE block, so the union over ranks is the single-domain set
	int64_t zlo = lo[2], zhi = hi[2];
	if (s->dist) {
		zlo = zlo > (int64_t)s->slab_lo * 8 ? zlo : (int64_t)s->slab_lo * 8;
		zhi = zhi < (int64_t)s->slab_hi * 8 ? zhi : (int64_t)s->slab_hi * 8;
		if (zhi < zlo) zhi = zlo;
	}
	const size_t per_layer = (size_t)(hi[0] - lo[0]) * (size_t)(hi[1] - lo[1]) * 8;
	const size_t n = per_layer * (size_t)(zhi - zlo), first = per_layer * (size_t)(zlo - lo[2]);
	if (per_layer * (size_t)(hi[2] - lo[2]) >= ((size_t)1 << 32)) return lfa_fail(s, LFA_E_INVALID, "more than 2^32 particles");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_particles_alloc(s, n));
	s->next_global_id = per_layer * (size_t)(hi[2] - lo[2]);
	s->np = n;
	s->np_live = n;
	s->c_home_valid = false;
	s->vmax2_valid = false;
	s->binned = false;
	s->grid_valid = false;
	s->system_valid = false;
	s->unknown_count_valid = false;
	s->cur = 0;
	IngestParams ip;
	for (int k = 0; k < 3; ++k) ip.off[k] = s->prm.grid_offset[k];
	ip.h = s->prm.cell_size;
	if (n) {
		hipLaunchKernelGGL(k_seed_block, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, first, s->pb[0], s->g,
		                   ip, (int)lo[0], (int)lo[1], (int)lo[2], (int)(hi[0] - lo[0]), (int)(hi[1] - lo[1]), seed, s->dist ? 1 : 0);
		LFA_LAUNCH_CHECK(s);
	}
	return LFA_OK;
}

// =============================================================================================== solids / cells
__global__ void k_set_solid(const int32_t *xyz, size_t k, uint8_t *solid, uint8_t *ctype, GridDims g) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= k) return;
	int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
	if (!in_grid(g, x, y, z)) return;
	uint32_t b = blocked_index(g, x, y, z);
	solid[b] = 1;
	ctype[b] = CT_SOLID;
}

extern "C" int lfa_set_solid_cells(lfa_sim *s, const int32_t *xyz, uint64_t k) {
	if (!s || (!xyz && k)) return LFA_E_INVALID;
	++s->solid_epoch;
	if (k == 0) return LFA_OK;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_ensure_io(s, k * 12));
	LFA_HIP(s, hipMemcpyAsync(s->io_buf, xyz, k * 12, hipMemcpyHostToDevice, s->stream));
	hipLaunchKernelGGL(k_set_solid, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s->stream,
	                   (const int32_t *)s->io_buf, (size_t)k, s->solid, s->ctype, s->g);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	s->system_valid = false;
	return LFA_OK;
}

extern "C" int lfa_clear_solid_cells(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	++s->solid_epoch;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	hipLaunchKernelGGL(k_init_ctype, dim3((unsigned)((s->ncp + 255) / 256)), dim3(256), 0, s->stream, s->ctype, s->solid,
	                   s->g, s->ncp);
	LFA_LAUNCH_CHECK(s);
	s->system_valid = false;
	s->grid_valid = false;
	return LFA_OK;
}

struct CellAos {
	double vel[3];
	uint8_t type;
	uint8_t pad[7];
};

/// dense x-fastest 32-B AoS <- blocked fp32 SoA. Grid kernels only touch the processed (dilated) tile set; every other
/// tile is implicit: its value is the base (0 after a P2G, the stored value after an explicit upload) plus the
/// background `bg` = gravity accumulated since then (src/simulation.cpp:72-78 adds g*dt to EVERY cell).
__global__ void k_export_cells(CellAos *out, GridDims g, size_t nc, const float *u, const float *v, const float *w,
                               const uint8_t *ctype, const uint8_t *solid, const uint32_t *tile_flag, int have_dilated,
                               int explicit_base, double bgx, double bgy, double bgz) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	CellAos c;
	memset(&c, 0, sizeof c);
	bool active = have_dilated && tile_flag[b >> 9] != 0;
	if (active) {
		c.vel[0] = (double)u[b]; c.vel[1] = (double)v[b]; c.vel[2] = (double)w[b];
		c.type = ctype[b] & 7;
	} else if (explicit_base) {
		c.vel[0] = (double)u[b] + bgx; c.vel[1] = (double)v[b] + bgy; c.vel[2] = (double)w[b] + bgz;
		c.type = ctype[b] & 7;
	} else {
		c.vel[0] = bgx; c.vel[1] = bgy; c.vel[2] = bgz;
		c.type = solid[b] ? CT_SOLID : CT_AIR;
	}
	out[r] = c;
}

static int export_cells(lfa_sim *s, void *aos32, const float *u, const float *v, const float *w, bool old) {
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_ensure_io(s, s->nc * 32));
	hipLaunchKernelGGL(k_export_cells, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream,
	                   (CellAos *)s->io_buf, s->g, s->nc, u, v, w, s->ctype, s->solid,
	                   s->grid_valid ? s->grid_flag : s->tile_flag, (s->grid_valid || s->binned) ? 1 : 0,
	                   s->grid_valid ? 0 : 1, old ? 0.0 : s->bg[0], old ? 0.0 : s->bg[1], old ? 0.0 : s->bg[2]);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(aos32, s->io_buf, s->nc * 32, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

extern "C" int lfa_download_cells(lfa_sim *s, void *aos32) {
	if (!s || !aos32) return LFA_E_INVALID;
	return export_cells(s, aos32, s->u, s->v, s->w, false);
}
extern "C" int lfa_download_old_cells(lfa_sim *s, void *aos32) {
	if (!s || !aos32) return LFA_E_INVALID;
	if (!s->uo) return lfa_fail(s, LFA_E_INVALID, "no old grid: FLIP P2G has not run");
	return export_cells(s, aos32, s->uo, s->vo, s->wo, true);
}

__global__ void k_import_cells(const CellAos *in, GridDims g, size_t nc, float *u, float *v, float *w, uint8_t *ctype,
                               uint8_t *solid) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	CellAos c = in[r];
	u[b] = (float)c.vel[0]; v[b] = (float)c.vel[1]; w[b] = (float)c.vel[2];
	ctype[b] = c.type & 7;
	solid[b] = (c.type & CT_SOLID) ? 1 : 0;
}

extern "C" int lfa_upload_cells(lfa_sim *s, const void *aos32) {
	if (!s || !aos32) return LFA_E_INVALID;
	++s->solid_epoch;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_join(s));
	LFA_TRY(lfa_ensure_io(s, s->nc * 32));
	LFA_HIP(s, hipMemcpyAsync(s->io_buf, aos32, s->nc * 32, hipMemcpyHostToDevice, s->stream));
	hipLaunchKernelGGL(k_import_cells, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream,
	                   (const CellAos *)s->io_buf, s->g, s->nc, s->u, s->v, s->w, s->ctype, s->solid);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	// an uploaded grid is fully explicit: no implicit background, every tile counts as processed
	s->grid_valid = false;
	s->bg[0] = s->bg[1] = s->bg[2] = 0.0;
	s->system_valid = false;
	return LFA_OK;
}

// =============================================================================================== binning (a2)
/// Pass 1: particles per tile + rank of each particle inside its tile. A wave takes TC_CHUNKS x 64 consecutive particles and
/// issues one atomic per distinct tile among them: particles arrive tile-coherent (they were binned last step and move < 1
/// tile), so that is 1-3 atomics per 256 particles, and a wave's particles of one tile get consecutive ranks in input order
/// (the scatter then writes runs of up to 1 KB per field). One particle per lane spent its time waiting for the returning
/// atomic (0.54 ms at C4; 4 chunks: see DESIGN.md).
__global__ void __launch_bounds__(256) k_tile_count(const uint32_t *key, size_t n, uint32_t *tile_count, uint32_t *rank) {
	const int lane = threadIdx.x & 63;
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
	const size_t i0 = wave * (64 * TC_CHUNKS) + lane;
	uint32_t tile[TC_CHUNKS], my_rank[TC_CHUNKS];
#pragma unroll
	for (int c = 0; c < TC_CHUNKS; ++c) {
		const size_t i = i0 + 64 * c;
		const uint32_t k = i < n ? key[i] : 0xFFFFFFFFu;
		tile[c] = k != 0xFFFFFFFFu ? k >> 9 : 0xFFFFFFFFu;
	}
	lfa_wave_tile_ranks(tile, tile_count, my_rank);
#pragma unroll
	for (int c = 0; c < TC_CHUNKS; ++c) {
		const size_t i = i0 + 64 * c;
		if (tile[c] != 0xFFFFFFFFu) rank[i] = my_rank[c];
	}
}

__global__ void k_tile_flags(const uint32_t *tile_count, uint32_t *flag, int nt) {
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t < nt) flag[t] = tile_count[t] > 0 ? 1u : 0u;
}
__global__ void k_compact_tiles(const uint32_t *flag, const uint32_t *scan, int *list, int *slot_of, int nt) {
	int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= nt) return;
	if (flag[t]) {
		list[scan[t]] = t;
		if (slot_of) slot_of[t] = (int)scan[t];
	} else if (slot_of) {
		slot_of[t] = -1;
	}
}
/// Every tile within one tile of a particle tile gets processed by the grid kernels (P2G reaches 1 cell, extrapolation
/// up to 8 cells, G2P 2 cells beyond a particle's cell).
__global__ void k_dilate(const int *ptiles, int n_ptiles, uint32_t *flag, GridDims g) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_ptiles * 27) return;
	int t = ptiles[i / 27], o = i % 27, tx, ty, tz;
	tile_coords(g, t, tx, ty, tz);
	int x = tx + o % 3 - 1, y = ty + (o / 3) % 3 - 1, z = tz + o / 9 - 1;
	if ((unsigned)x < (unsigned)g.ntx && (unsigned)y < (unsigned)g.nty && (unsigned)z < (unsigned)g.ntz)
		flag[x + g.ntx * (y + g.nty * z)] = 1u;
}

/// Pass 2: move every particle to its tile's segment. With `shuffle` the slot inside the segment is a multiplicative
/// permutation of the rank, which separates particles of one cell (uploads arrive cell-sorted; neighbouring lanes
/// hitting one cell would serialise the LDS atomics of the P2G scatter).
/// Only key, t, id move (20 of the 68 bytes); from[d] records the source index, v and C follow through it (k_p2g_binned,
/// k_g2p, k_gather_vc) - see lfa_sim::vc_pending. (PIC / FLIP keep C in its home array, indexed by the particle id.)
__global__ void k_tile_scatter(size_t n, ParticleSoA src, ParticleSoA dst, const uint32_t *rank,
                               const uint32_t *tile_start, const uint32_t *tile_count, int shuffle, uint32_t *from) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t key = src.key[i], tile = key >> 9, r = rank[i];
	if (key == 0xFFFFFFFFu) return;
	if (shuffle) {
		uint32_t cnt = tile_count[tile];
		if (cnt % 1000003u != 0) r = (uint32_t)(((uint64_t)r * 1000003ull) % cnt);
	}
	size_t d = (size_t)tile_start[tile] + r;
	dst.key[d] = key;
#pragma unroll
	for (int k = 0; k < 3; ++k) dst.t[k][d] = src.t[k][i];
	dst.id[d] = src.id[i];
	from[d] = (uint32_t)i;
}
/// C to / from its home array (indexed by particle id): see lfa_sim::c_home.
__global__ void k_c_to_home(size_t n, ParticleSoA p, float *home, size_t stride) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || p.key[i] == 0xFFFFFFFFu) return;
	const size_t j = p.id[i];
#pragma unroll
	for (int k = 0; k < 9; ++k) home[k * stride + j] = p.c[k][i];
}
__global__ void k_c_from_home(size_t n, ParticleSoA p, const float *home, size_t stride) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n || p.key[i] == 0xFFFFFFFFu) return;
	const size_t j = p.id[i];
#pragma unroll
	for (int k = 0; k < 9; ++k) p.c[k][i] = home[k * stride + j];
}
/// The deferred half: v, C of the particle now at d from where it was before the binning.
__global__ void k_gather_vc(size_t n, ParticleSoA old, ParticleSoA cur, const uint32_t *from, int with_c) {
	size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (d >= n) return;
	const uint32_t i = from[d];
#pragma unroll
	for (int k = 0; k < 3; ++k) cur.v[k][d] = old.v[k][i];
	if (with_c) {
#pragma unroll
		for (int k = 0; k < 9; ++k) cur.c[k][d] = old.c[k][i];
	}
}

/// Particles per cell for every processed tile (the `count` half of _space_hash, src/simulation.cpp:266-291);
/// tiles of the dilated set that hold no particles get zeros.
__global__ void __launch_bounds__(256)
k_cell_count(const int *dtiles, int n_dtiles, const uint32_t *key, const uint32_t *tile_start, uint32_t *cell_count) {
	__shared__ uint32_t cnt[LFA_TILE_CELLS];
	for (int slot = blockIdx.x; slot < n_dtiles; slot += gridDim.x) {
		int tile = dtiles[slot];
		cnt[threadIdx.x] = 0;
		cnt[threadIdx.x + 256] = 0;
		__syncthreads();
		uint32_t b = tile_start[tile], e = tile_start[tile + 1];
		for (uint32_t i = b + threadIdx.x; i < e; i += 256) atomicAdd(&cnt[key[i] & 511], 1u);
		__syncthreads();
		cell_count[(size_t)tile * LFA_TILE_CELLS + threadIdx.x] = cnt[threadIdx.x];
		cell_count[(size_t)tile * LFA_TILE_CELLS + 256 + threadIdx.x] = cnt[threadIdx.x + 256];
		__syncthreads();
	}
}

__global__ void k_compact_range(const uint32_t *flag, const uint32_t *scan, int *list, int *slot_of, int lo, int hi,
                                int slot_base) {
	int t = lo + blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= hi) return;
	if (flag[t]) {
		list[scan[t]] = t;
		if (slot_of) slot_of[t] = slot_base + (int)scan[t];
	} else if (slot_of) {
		slot_of[t] = -1;
	}
}
__global__ void k_fill_i32(int *p, int n, int v) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}
__global__ void k_gather_u32(const uint32_t *src, const int *idx, int n, uint32_t *dst) {
	int i = threadIdx.x;
	if (i < n) dst[i] = src[idx[i]];
}

/// The same dilation from the slot table (>= 0: the tile holds particles), a thread per tile: needs no tile list - and so no count
/// on the host - to size its launch (single domain: the binning reads its three counts back together, once).
__global__ void k_dilate_slots(const int *tile_pslot, uint32_t *flag, GridDims g) {
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= g.nt || tile_pslot[t] < 0) return;
	int tx, ty, tz;
	tile_coords(g, t, tx, ty, tz);
	for (int o = 0; o < 27; ++o) {
		const int x = tx + o % 3 - 1, y = ty + (o / 3) % 3 - 1, z = tz + o / 9 - 1;
		if ((unsigned)x < (unsigned)g.ntx && (unsigned)y < (unsigned)g.nty && (unsigned)z < (unsigned)g.ntz)
			flag[x + g.ntx * (y + g.nty * z)] = 1u;
	}
}

/// Compacts the flagged tiles of [lo, hi) into `list` (ascending tile id); the count stays on the device (*total_dev).
static int compact_tiles_async(lfa_sim *s, const uint32_t *flag, int lo, int hi, int *list, int *slot_of, uint32_t *total_dev) {
	LFA_TRY(lfa_exclusive_scan_u32(s, flag + lo, s->tile_scan + lo, (size_t)(hi - lo), total_dev));
	hipLaunchKernelGGL(k_compact_range, dim3((hi - lo + 255) / 256), dim3(256), 0, s->stream, flag, s->tile_scan - 0,
	                   list - 0, slot_of, lo, hi, 0);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

/// Compacts the flagged tiles of [lo, hi) into `list` (ascending tile id); the count is returned through the host.
static int compact_tiles(lfa_sim *s, const uint32_t *flag, int lo, int hi, int *list, int *slot_of, int *count) {
	*count = 0;
	if (hi <= lo) return LFA_OK;
	uint32_t *total = (uint32_t *)s->pcg_state + 8;
	LFA_TRY(lfa_exclusive_scan_u32(s, flag + lo, s->tile_scan + lo, (size_t)(hi - lo), total));
	hipLaunchKernelGGL(k_compact_range, dim3((hi - lo + 255) / 256), dim3(256), 0, s->stream, flag, s->tile_scan - 0,
	                   list - 0, slot_of, lo, hi, 0);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(s->h_pinned, total, 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	*count = (int)s->h_pinned[0];
	return LFA_OK;
}

int lfa_c_home_ensure(lfa_sim *s, size_t n) {
	if (n <= s->c_home_cap) return LFA_OK;
	const size_t cap = ((n + n / 8) + 1023) & ~(size_t)1023;
	float *nh = nullptr;
	LFA_HIP(s, hipMalloc(&nh, cap * 9 * sizeof(float)));
	auto fail = [&](hipError_t e, const char *what) {  // (the new array does not outlive a failure)
		(void)hipFree(nh);
		return lfa_fail(s, LFA_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
	};
	if (s->c_home && s->c_home_valid && s->c_home_cap)  // (every entry: on slabs the resident ids are anywhere in the job's range)
		for (int k = 0; k < 9; ++k) {
			const hipError_t e = hipMemcpyAsync(nh + (size_t)k * cap, s->c_home + (size_t)k * s->c_home_cap,
			                                    s->c_home_cap * 4, hipMemcpyDeviceToDevice, s->stream);
			if (e != hipSuccess) return fail(e, "copying C to its larger home array");
		}
	if (s->c_home) {
		const hipError_t e = hipStreamSynchronize(s->stream);
		if (e != hipSuccess) return fail(e, "hipStreamSynchronize");
		LFA_HIP(s, hipFree(s->c_home));
	}
	s->c_home = nh;
	s->c_home_cap = cap;
	return LFA_OK;
}
int lfa_c_home_restore(lfa_sim *s) {
	if (!s->c_home_valid) return LFA_OK;
	s->c_home_valid = false;
	const size_t n = s->binned ? s->np_live : s->np;
	if (!n) return LFA_OK;
	hipLaunchKernelGGL(k_c_from_home, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, s->pb[s->cur], (const float *)s->c_home,
	                   s->c_home_cap);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

int lfa_particles_materialize(lfa_sim *s) {
	if (!s->vc_pending) return LFA_OK;
	LFA_TRY(lfa_corr_commit(s));  // the gather writes the v / C arrays a correction in flight keeps its inputs in
	s->vc_pending = false;
	LFA_HIP(s, hipSetDevice(s->device));
	const size_t n = s->np_live;
	if (!n) return LFA_OK;
	hipLaunchKernelGGL(k_gather_vc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, s->pb[s->cur ^ 1], s->pb[s->cur],
	                   (const uint32_t *)s->vc_src, s->vc_with_c ? 1 : 0);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

extern "C" int lfa_hash_particles(lfa_sim *s) { return lfa_hash_particles_impl(s, false); }
int lfa_hash_particles_impl(lfa_sim *s, bool counts_done) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	LFA_TRY(lfa_particles_materialize(s));  // a binning on top of a deferred one: complete that first
	const GridDims &g = s->g;
	const int nt = g.nt, L = g.ntx * g.nty;
	const size_t n = s->binned ? s->np_live : s->np;
	ParticleSoA &src = s->pb[s->cur], &dst = s->pb[s->cur ^ 1];
	// owned tile range, and the range including the two ghost layers
	const int own_lo = s->slab_lo * L, own_hi = s->slab_hi * L;
	const int all_lo = lfa_has_lo(s) ? own_lo - L : own_lo, all_hi = lfa_has_hi(s) ? own_hi + L : own_hi;

	// pass 1 (particles per tile, rank inside the tile) - unless the advection kernel of lfa_time_step has just done it
	if (!counts_done) {
		LFA_HIP(s, hipMemsetAsync(s->tile_count, 0, (size_t)(nt + 1) * 4, s->stream));
		if (n) {
			hipLaunchKernelGGL(k_tile_count, dim3((unsigned)((n + 256 * TC_CHUNKS - 1) / (256 * TC_CHUNKS))), dim3(256), 0, s->stream, src.key, n,
			                   s->tile_count, s->rank);
			LFA_LAUNCH_CHECK(s);
		}
	}
	// tile_start[0..nt] (exclusive scan; entry nt = number of live particles because tile_count[nt] == 0)
	LFA_TRY(lfa_exclusive_scan_u32(s, s->tile_count, s->tile_start, (size_t)nt + 1, nullptr));
	LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 8, s->tile_start + nt, 4, hipMemcpyDeviceToHost, s->stream));

	// ---- particle tiles: owned ones from the counts, the neighbours' adjacent layers by message
	hipLaunchKernelGGL(k_tile_flags, dim3((nt + 255) / 256), dim3(256), 0, s->stream, s->tile_count, s->tile_flag, nt);
	LFA_LAUNCH_CHECK(s);
	LFA_TRY(lfa_dist_exchange_tile_layers_u32(s, s->tile_flag));
	hipLaunchKernelGGL(k_fill_i32, dim3((nt + 255) / 256), dim3(256), 0, s->stream, s->tile_pslot, nt, -1);
	LFA_LAUNCH_CHECK(s);
	const bool one_sync = !s->dist;  // single domain: the tile lists are built without the host, their counts come back together
	if (one_sync) {
		uint32_t *tot = (uint32_t *)s->pcg_state + 8;  // [8] particle tiles, [9] processed tiles
		LFA_TRY(compact_tiles_async(s, s->tile_flag, 0, nt, s->ptiles_all, s->tile_pslot, tot));
		LFA_HIP(s, hipMemsetAsync(s->tile_flag, 0, (size_t)(nt + 1) * 4, s->stream));
		hipLaunchKernelGGL(k_dilate_slots, dim3((nt + 255) / 256), dim3(256), 0, s->stream, (const int *)s->tile_pslot, s->tile_flag, g);
		LFA_LAUNCH_CHECK(s);
		LFA_TRY(compact_tiles_async(s, s->tile_flag, 0, nt, s->dtiles, nullptr, tot + 1));
		LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 9, tot, 8, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		s->n_ptiles_all = (int)s->h_pinned[9];
		s->n_dtiles = (int)s->h_pinned[10];
		s->p_off = 0;
		s->n_ptiles = s->n_ptiles_all;
		s->n_own_first = s->n_own_last = s->n_ghost_lo = s->n_ghost_hi = 0;
		s->ptiles = s->ptiles_all;
	} else {
		// Slabs: the same without the host in between (round 3 read seven counts back one by one: the particle tiles, the layer
		// marks, the processed tiles, four halo lists). Collectives (the flag exchanges) are stream operations; the halo lists
		// have fixed regions of one tile layer each, so no count is needed to place them.
		uint32_t *tot = (uint32_t *)s->pcg_state + 20;  // [0] particle tiles [1..4] scan at the layer marks [5] processed tiles [6..9] halo lists
		LFA_TRY(compact_tiles_async(s, s->tile_flag, all_lo, all_hi, s->ptiles_all, s->tile_pslot, tot));
		int marks[4] = {own_lo, own_lo + L < own_hi ? own_lo + L : own_hi, own_hi - L > own_lo ? own_hi - L : own_lo, own_hi};
		int *didx = s->pcg_state + 12;
		LFA_HIP(s, hipMemcpyAsync(didx, marks, 16, hipMemcpyHostToDevice, s->stream));
		hipLaunchKernelGGL(k_gather_u32, dim3(1), dim3(64), 0, s->stream, s->tile_scan, didx, 4, tot + 1);  // (before the next scan reuses tile_scan)
		LFA_LAUNCH_CHECK(s);
		LFA_HIP(s, hipMemsetAsync(s->tile_flag, 0, (size_t)(nt + 1) * 4, s->stream));
		hipLaunchKernelGGL(k_dilate_slots, dim3((nt + 255) / 256), dim3(256), 0, s->stream, (const int *)s->tile_pslot, s->tile_flag, g);
		LFA_LAUNCH_CHECK(s);
		if (own_lo > 0) LFA_HIP(s, hipMemsetAsync(s->tile_flag, 0, (size_t)own_lo * 4, s->stream));
		if (own_hi < nt) LFA_HIP(s, hipMemsetAsync(s->tile_flag + own_hi, 0, (size_t)(nt - own_hi) * 4, s->stream));
		LFA_TRY(lfa_dist_exchange_tile_layers_u32(s, s->tile_flag));
		LFA_TRY(compact_tiles_async(s, s->tile_flag, own_lo, own_hi, s->dtiles, nullptr, tot + 5));
		{
			const int hlo[4] = {s->slab_lo * L, (s->slab_hi - 1) * L, (s->slab_lo - 1) * L, s->slab_hi * L};
			const bool on[4] = {lfa_has_lo(s), lfa_has_hi(s), lfa_has_lo(s), lfa_has_hi(s)};
			LFA_HIP(s, hipMemsetAsync(tot + 6, 0, 16, s->stream));
			for (int w = 0; w < 4; ++w)
				if (on[w]) LFA_TRY(compact_tiles_async(s, s->tile_flag, hlo[w], hlo[w] + L, s->halo_tiles + (size_t)w * L, nullptr, tot + 6 + w));
		}
		LFA_HIP(s, hipMemcpyAsync(s->h_pinned + 20, tot, 40, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		const uint32_t *h = s->h_pinned + 20;
		s->n_ptiles_all = (int)h[0];
		int vals[4];
		for (int k = 0; k < 4; ++k) vals[k] = marks[k] >= all_hi ? s->n_ptiles_all : (int)h[1 + k];  // (the scan has no entry at the end of the range)
		s->p_off = vals[0];
		s->n_ptiles = vals[3] - vals[0];
		s->n_own_first = vals[1] - vals[0];
		s->n_own_last = vals[3] - vals[2];
		s->n_ghost_lo = s->p_off;
		s->n_ghost_hi = s->n_ptiles_all - vals[3];
		s->ptiles = s->ptiles_all + s->p_off;
		s->n_dtiles = (int)h[5];
		for (int w = 0; w < 4; ++w) s->n_halo[w] = (int)h[6 + w];
	}
	s->np_live = s->h_pinned[8];
	if (s->dist) s->np = s->np_live;  // particles migrate: the resident count is the live count
	s->holes = false;
	s->n_arrivals = 0;
	s->move_pending = false;  // (a binning re-uses the buffer the positions of before a split move were kept in)
	if (n) {
		// v and C (48 of the 68 bytes) stay behind and are read through vc_src by the P2G; the G2P writes the new ones in the
		// new order (the slab migration, which packs whole records, completes the move first).
		// PIC and FLIP carry C through the step unchanged (the reference's G2P does not touch it): C goes to its home array once and
		// stays there, so their scatter moves key, t, id alone, like APIC's. The array is indexed by the particle id: on slabs by the
		// job-wide id, so every rank holds room for every particle of the job - 36 bytes each, 2.4 GB at C4 - and a particle that
		// changes ranks takes its nine floats along in its record (dist.hip: k_pack_leavers / k_unpack_arrivals)
		const dim3 sgrid((unsigned)((n + 255) / 256));
		const bool home = s->prm.simulation_method != LFA_APIC;
		if (home && !s->c_home_valid) {
			LFA_TRY(lfa_c_home_ensure(s, s->dist ? std::max<size_t>(s->pcap, (size_t)s->next_global_id) : s->pcap));
			hipLaunchKernelGGL(k_c_to_home, sgrid, dim3(256), 0, s->stream, n, src, s->c_home, s->c_home_cap);
			LFA_LAUNCH_CHECK(s);
			s->c_home_valid = true;
		} else if (!home && s->c_home_valid) {  // the method has changed to APIC
			LFA_TRY(lfa_c_home_restore(s));
		}
		// (an upload arrives cell-sorted: the first binning permutes the slots inside a tile so that neighbouring lanes of the P2G
		// do not hit one cell)
		hipLaunchKernelGGL(k_tile_scatter, sgrid, dim3(256), 0, s->stream, n, src, dst, s->rank, s->tile_start, s->tile_count,
		                   s->binned ? 0 : 1, s->vc_src);
		LFA_LAUNCH_CHECK(s);
		s->cur ^= 1;
		s->vc_pending = true;
		s->vc_extent = n;
		s->vc_with_c = !home;
	}
	if (s->n_dtiles) {
		int grid = s->n_dtiles < 8192 ? s->n_dtiles : 8192;
		hipLaunchKernelGGL(k_cell_count, dim3(grid), dim3(256), 0, s->stream, s->dtiles, s->n_dtiles,
		                   s->pb[s->cur].key, s->tile_start, s->cell_count);
		LFA_LAUNCH_CHECK(s);
	}
	s->binned = true;
	s->system_valid = false;
	s->unknown_count_valid = false;
	return LFA_OK;
}

/// Processed tiles of the four boundary layers: [own first | own last | ghost below | ghost above], each in its own region of
/// one tile layer of halo_tiles (the binning builds them itself, with its other lists; this entry point re-builds them alone).
int lfa_dist_build_halo_lists(lfa_sim *s) {
	const int L = s->g.ntx * s->g.nty;
	const int lo[4] = {s->slab_lo * L, (s->slab_hi - 1) * L, (s->slab_lo - 1) * L, s->slab_hi * L};
	const bool on[4] = {lfa_has_lo(s), lfa_has_hi(s), lfa_has_lo(s), lfa_has_hi(s)};
	for (int w = 0; w < 4; ++w) {
		s->n_halo[w] = 0;
		if (on[w]) LFA_TRY(compact_tiles(s, s->tile_flag, lo[w], lo[w] + L, s->halo_tiles + (size_t)w * L, nullptr, &s->n_halo[w]));
	}
	return LFA_OK;
}

// ---- fluid-cell list at the boundary (reference order = ascending raw index)
__global__ void k_raw_unknown_flags(GridDims g, size_t nc, const uint32_t *cell_count, const uint32_t *tile_flag,
                                    uint32_t *flag, int z0, int z1) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	// with slabs only the owned cell layers [z0, z1) are numbered (ghost layers belong to the neighbours)
	flag[r] = (z >= z0 && z < z1 && tile_flag[b >> 9] && cell_count[b] > 0) ? 1u : 0u;
}

/// Numbers the unknowns in the reference's order; raw_scan[r] = unknown index of raw cell r (valid where flagged).
int lfa_number_unknowns(lfa_sim *s) {
	if (s->unknown_count_valid) return LFA_OK;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "call lfa_hash_particles first");
	if (!s->raw_scan) LFA_HIP(s, hipMalloc(&s->raw_scan, (s->nc + 1) * 4));
	hipLaunchKernelGGL(k_raw_unknown_flags, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream, s->g, s->nc,
	                   s->cell_count, s->tile_flag, s->raw_scan, s->slab_lo * 8, s->slab_hi * 8);
	LFA_LAUNCH_CHECK(s);
	LFA_TRY(lfa_exclusive_scan_u32(s, s->raw_scan, s->raw_scan, s->nc, (uint32_t *)s->pcg_state + 10));
	LFA_HIP(s, hipMemcpyAsync(s->h_pinned, (uint32_t *)s->pcg_state + 10, 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	s->n_unknowns = s->h_pinned[0];
	s->unknown_count_valid = true;
	return LFA_OK;
}

extern "C" uint64_t lfa_num_fluid_cells(lfa_sim *s) {
	if (!s || lfa_number_unknowns(s) != LFA_OK) return 0;
	return s->n_unknowns;
}

__global__ void k_export_fluid_cells(GridDims g, size_t nc, const uint32_t *cell_count, const uint32_t *tile_flag,
                                     const uint32_t *raw_scan, uint64_t *out, int z0, int z1) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	if (z >= z0 && z < z1 && tile_flag[b >> 9] && cell_count[b] > 0) out[raw_scan[r]] = r;
}

extern "C" int lfa_download_fluid_cells(lfa_sim *s, uint64_t *raw, uint64_t n) {
	if (!s || (!raw && n)) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_number_unknowns(s));
	if (n != s->n_unknowns) return lfa_fail(s, LFA_E_INVALID, "fluid cell count is %llu", (unsigned long long)s->n_unknowns);
	if (n == 0) return LFA_OK;
	LFA_TRY(lfa_ensure_io(s, n * 8));
	hipLaunchKernelGGL(k_export_fluid_cells, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream, s->g, s->nc,
	                   s->cell_count, s->tile_flag, s->raw_scan, (uint64_t *)s->io_buf, s->slab_lo * 8, s->slab_hi * 8);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(raw, s->io_buf, n * 8, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

__global__ void k_export_counts(GridDims g, size_t nc, const uint32_t *cell_count, const uint32_t *tile_flag,
                                uint32_t *out) {
	size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= nc) return;
	int x = (int)(r % g.nx), y = (int)((r / g.nx) % g.ny), z = (int)(r / ((size_t)g.nx * g.ny));
	uint32_t b = blocked_index(g, x, y, z);
	out[r] = tile_flag[b >> 9] ? cell_count[b] : 0u;
}

extern "C" int lfa_download_cell_counts(lfa_sim *s, uint32_t *count) {
	if (!s || !count) return LFA_E_INVALID;
	if (!s->binned) return lfa_fail(s, LFA_E_INVALID, "call lfa_hash_particles first");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_ensure_io(s, s->nc * 4));
	hipLaunchKernelGGL(k_export_counts, dim3((unsigned)((s->nc + 255) / 256)), dim3(256), 0, s->stream, s->g, s->nc,
	                   s->cell_count, s->tile_flag, (uint32_t *)s->io_buf);
	LFA_LAUNCH_CHECK(s);
	LFA_HIP(s, hipMemcpyAsync(count, s->io_buf, s->nc * 4, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	return LFA_OK;
}

// =============================================================================================== CFL (a21)
__global__ void __launch_bounds__(256) k_max_speed2(size_t n, const float *vx, const float *vy, const float *vz, double *partials) {
	__shared__ double lds[4];
	double m = 0.0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		double x = vx[i], y = vy[i], z = vz[i];
		double l = x * x + y * y + z * z;
		m = l > m ? l : m;
	}
	m = wave_max(m);
	if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int i = 1; i < 4; ++i) m = lds[i] > m ? lds[i] : m;
		partials[blockIdx.x] = m;
	}
}

extern "C" int lfa_cfl(lfa_sim *s, double *out) {
	if (!s || !out) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_join(s));
	double m = 0.0;
	if (s->vmax2_valid && s->np_live) {
		// the last G2P reduced max |v|^2 while it wrote the velocities and sent the 4 bytes to the host behind itself
		// (grid_ops.hip: g2p_run): no pass over the particles, no copy of its own
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		float f;
		memcpy(&f, s->h_pinned + 100, 4);
		m = (double)f;
	} else if (s->np_live) {
		LFA_TRY(lfa_particles_materialize(s));
		int grid = (int)((s->np_live + 255) / 256);
		if (grid > 1024) grid = 1024;
		const ParticleSoA &p = s->pb[s->cur];
		hipLaunchKernelGGL(k_max_speed2, dim3(grid), dim3(256), 0, s->stream, s->np_live, p.v[0], p.v[1], p.v[2], s->partials);
		LFA_LAUNCH_CHECK(s);
		std::vector<double> h(grid);
		LFA_HIP(s, hipMemcpyAsync(h.data(), s->partials, grid * 8, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		for (double x : h) m = x > m ? x : m;
	}
	*out = s->prm.cell_size / sqrt(m);  // +inf when all velocities are zero, like src/simulation.cpp:204
	return LFA_OK;
}

// =============================================================================================== measured HBM ceiling
// The copy is what the step's streaming kernels are compared with, so it must be a good copy: several independent 16-byte loads
// in flight per thread (a one-load-per-iteration grid-stride loop reached 4.6 TB/s where MI355X_MICROARCH.md records 6.29 for a
// float4 copy), optionally non-temporal stores (the destination is not read again). The best variant is reported, with its name.
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stream_copy(const float4 *in, float4 *out, size_t n) {
	const size_t chunk = (size_t)256 * U;
	for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
		float4 v[U];
#pragma unroll
		for (int k = 0; k < U; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			if (i < n) {
				if (NT) {
					typedef float f4v __attribute__((ext_vector_type(4)));
					const f4v w = __builtin_nontemporal_load((const f4v *)in + i);
					v[k] = make_float4(w.x, w.y, w.z, w.w);
				} else {
					v[k] = in[i];
				}
			}
		}
#pragma unroll
		for (int k = 0; k < U; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			if (i < n) {
				if (NT) {
					typedef float f4v __attribute__((ext_vector_type(4)));
					f4v w;
					w.x = v[k].x; w.y = v[k].y; w.z = v[k].z; w.w = v[k].w;
					__builtin_nontemporal_store(w, (f4v *)out + i);
				} else {
					out[i] = v[k];
				}
			}
		}
	}
}
template <int U>
__global__ void __launch_bounds__(256) k_stream_read(const float4 *in, float *sink, size_t n) {
	float acc = 0.f;
	const size_t chunk = (size_t)256 * U;
	for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
		float4 v[U];
#pragma unroll
		for (int k = 0; k < U; ++k) {
			const size_t i = base + (size_t)k * 256 + threadIdx.x;
			v[k] = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);
		}
#pragma unroll
		for (int k = 0; k < U; ++k) acc += (v[k].x + v[k].y) + (v[k].z + v[k].w);
	}
	if (acc == 123.456f) sink[0] = acc;  // never true for the zero-filled buffer: keeps the loads alive
}

static int g_stream_best = -1;
extern "C" const char *lfa_bench_stream_variant(void) {
	static const char *names[] = {"1 x float4 per thread and iteration", "4 x float4 in flight", "4 x float4 in flight, non-temporal",
	                              "8 x float4 in flight, non-temporal", "4 x float4 in flight, non-temporal, 32 workgroups per CU"};
	return g_stream_best >= 0 ? names[g_stream_best] : "";
}

// (one-shot scratch: straight from / back to the driver, never parked in the handle cache)
extern "C" int lfa_bench_stream(lfa_sim *s, uint64_t bytes, int reps, double *copy_gbs, double *read_gbs) {
	if (!s || reps < 1 || bytes < (1u << 20)) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	const size_t n = (size_t)bytes / 16;
	float4 *a = nullptr, *b = nullptr;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	auto cleanup = [&]() {
		if (e0) (void)hipEventDestroy(e0);
		if (e1) (void)hipEventDestroy(e1);
		if (a) (void)(hipFree)(a);
		if (b) (void)(hipFree)(b);
	};
	if ((hipMalloc)((void **)&a, n * 16) != hipSuccess || (hipMalloc)((void **)&b, n * 16) != hipSuccess) {
		cleanup();
		return lfa_fail(s, LFA_E_OOM, "lfa_bench_stream: hipMalloc of 2 x %llu bytes failed", (unsigned long long)bytes);
	}
	int rc = LFA_OK;
	auto run = [&]() -> int {
		LFA_HIP(s, hipEventCreate(&e0));
		LFA_HIP(s, hipEventCreate(&e1));
		LFA_HIP(s, hipMemsetAsync(a, 0, n * 16, s->stream));
		LFA_HIP(s, hipMemsetAsync(b, 0, n * 16, s->stream));
		float ms = 0.f;
		double best_copy = 0.0, best_read = 0.0;
		for (int variant = 0; variant < 7; ++variant) {  // 0-4: copies, 5-6: reads
			const dim3 grid(256 * (variant == 4 ? 32 : 16));  // 16 (32) workgroups per CU, grid-stride over chunks
			for (int r = -1; r < reps; ++r) {  // r == -1: warm-up
				if (r == 0) LFA_HIP(s, hipEventRecord(e0, s->stream));
				switch (variant) {
				case 0: hipLaunchKernelGGL((k_stream_copy<1, false>), grid, dim3(256), 0, s->stream, (const float4 *)a, b, n); break;
				case 1: hipLaunchKernelGGL((k_stream_copy<4, false>), grid, dim3(256), 0, s->stream, (const float4 *)a, b, n); break;
				case 2: hipLaunchKernelGGL((k_stream_copy<4, true>), grid, dim3(256), 0, s->stream, (const float4 *)a, b, n); break;
				case 3: hipLaunchKernelGGL((k_stream_copy<8, true>), grid, dim3(256), 0, s->stream, (const float4 *)a, b, n); break;
				case 4: hipLaunchKernelGGL((k_stream_copy<4, true>), grid, dim3(256), 0, s->stream, (const float4 *)a, b, n); break;
				case 5: hipLaunchKernelGGL(k_stream_read<1>, grid, dim3(256), 0, s->stream, (const float4 *)a, (float *)b, n); break;
				default: hipLaunchKernelGGL(k_stream_read<4>, grid, dim3(256), 0, s->stream, (const float4 *)a, (float *)b, n); break;
				}
			}
			LFA_LAUNCH_CHECK(s);
			LFA_HIP(s, hipEventRecord(e1, s->stream));
			LFA_HIP(s, hipEventSynchronize(e1));
			LFA_HIP(s, hipEventElapsedTime(&ms, e0, e1));
			const double gbs = (double)(n * 16) * reps * (variant < 5 ? 2.0 : 1.0) / ((double)ms * 1e-3) * 1e-9;
			if (variant < 5 && gbs > best_copy) { best_copy = gbs; g_stream_best = variant; }
			if (variant >= 5 && gbs > best_read) best_read = gbs;
		}
		if (copy_gbs) *copy_gbs = best_copy;
		if (read_gbs) *read_gbs = best_read;
		return LFA_OK;
	};
	rc = run();
	cleanup();  // (also on an error path: events and buffers never outlive the call)
	return rc;
}

// =============================================================================================== timing / counts
extern "C" int lfa_enable_timing(lfa_sim *s, int on) {
	if (!s) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	if (on && !s->ev_created) {
		for (auto &e : s->ev) LFA_HIP(s, hipEventCreate(&e));
		s->ev_created = true;
	}
	s->timing = on != 0;
	return LFA_OK;
}
extern "C" int lfa_get_timings(lfa_sim *s, double ms[LFA_NUM_TIMERS]) {
	if (!s || !ms) return LFA_E_INVALID;
	for (int i = 0; i < LFA_NUM_TIMERS; ++i) ms[i] = s->ms[i];
	return LFA_OK;
}
extern "C" int lfa_get_counts(lfa_sim *s, uint64_t counts[5]) {
	if (!s || !counts) return LFA_E_INVALID;
	counts[0] = s->np;
	counts[1] = lfa_num_fluid_cells(s);
	counts[2] = (uint64_t)s->n_ptiles;
	counts[3] = (uint64_t)s->n_dtiles;
	counts[4] = s->ncp;
	return LFA_OK;
}
