// libfluid_amd/csrc/dist.hip -- z-slab domain decomposition across the GPUs of one node.
//
// The reference is single-process (no NCCL/MPI anywhere), so nothing here restates reference code; the exchanges are
// what the stencils of the hot path need across a slab face:
//   binning      : "holds particles" / "is processed" flags of the two boundary tile layers          (2 x ntx*nty u32)
//   P2G          : the boundary plane of the per-tile partial sums (6 x 100 floats per particle tile) - particles within
//                  one cell of the face contribute to the neighbour's faces (src/simulation.cpp:309-320 gathers them)
//   grid halos   : u,v,w,type,count (+ FLIP old grid) of the processed tiles of the neighbour's adjacent tile layer,
//                  before the system build, before extrapolation and before G2P
//   PCG          : one z-slice (64 values per particle tile) of the search vector per iteration, of the pressure once;
//                  the two dot products and the signed max as scalar all-reduces
//   multigrid    : per distributed level one z-slice of the pre-smoothed and of the post-smoothed iterate (whole boundary
//                  tile layer on levels >= 1); one sum all-reduce of the first replicated level's right-hand side (mg.hip)
// Every rank indexes the GLOBAL grid, so a ghost tile is simply a tile id whose data arrives by message; kernels are the
// single-GPU kernels, run over the owned tile lists.
//
// Transports: RCCL (ncclSend/ncclRecv grouped to z-1/z+1 + ncclAllReduce on the handle's stream; librccl is dlopen'ed so
// that single-GPU use has no dependency on it) and an in-process one (one host thread per handle, device-to-device copies
// at a rendezvous) used to test the protocol with several "virtual slabs" on one GPU; and a multi-process one staged through
// host shared memory (no RCCL, no peer access: the functional fallback, and how N processes are run on a 1-GPU box).
#include <dlfcn.h>
#include <fcntl.h>
#include <math.h>
#include <sched.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include "common.h"

// ================================================================================================= pack / unpack
namespace {
struct FieldList {
	void *ptr[8];
	int words[8];  // 32-bit words per tile of each field
	int n, tile_words;
};

/// buffer[k][field][...] <-> field[tile[k]*words + ...]; one workgroup per tile.
template <bool PACK>
__global__ void __launch_bounds__(256) k_tiles_copy(const int *tiles, int n, FieldList f, uint32_t *buf) {
	const int k = blockIdx.x;
	if (k >= n) return;
	const size_t tile = (size_t)tiles[k];
	uint32_t *b = buf + (size_t)k * f.tile_words;
	for (int i = 0; i < f.n; ++i) {
		uint32_t *g = (uint32_t *)f.ptr[i] + tile * f.words[i];
		for (int w = threadIdx.x; w < f.words[i]; w += 256) {
			if (PACK) b[w] = g[w];
			else g[w] = b[w];
		}
		b += f.words[i];
	}
}

/// One z-plane (hz fixed) of the six 10x10x10 partial-sum arrays of a particle tile's staging slab.
template <bool PACK>
__global__ void __launch_bounds__(128) k_planes_copy(float *stage_all, int slot0, int n, int hz, float *buf) {
	const int k = blockIdx.x;
	if (k >= n) return;
	float *slab = stage_all + (size_t)(slot0 + k) * 6 * LFA_HALO_CELLS;
	float *b = buf + (size_t)k * 600;
	for (int i = threadIdx.x; i < 600; i += 128) {
		const int c = i % 100;
		float *g = slab + (i / 100) * LFA_HALO_CELLS + lfa_stage_index(c % 10, c / 10, hz);
		if (PACK) b[i] = *g;
		else *g = b[i];
	}
}

/// One z-slice (64 elements, `slice_words` 32-bit words) of a tile-major vector for a run of particle tiles; both slab
/// faces in one launch (blockIdx.y = face: the packs / unpacks of a halo exchange are launch-latency bound).
struct SliceRun {
	int slot0[2], n[2], zz[2];
	uint32_t *buf[2];
};
template <bool PACK>
__global__ void __launch_bounds__(64) k_slices_copy(const int *ptiles_all, SliceRun r, uint32_t *vec, int slice_words) {
	const int w = blockIdx.y, k = blockIdx.x;
	if (k >= r.n[w]) return;
	uint32_t *g = vec + ((size_t)ptiles_all[r.slot0[w] + k] * 8 + (size_t)r.zz[w]) * slice_words;
	uint32_t *b = r.buf[w] + (size_t)k * slice_words;
	for (int i = threadIdx.x; i < slice_words; i += 64) {
		if (PACK) b[i] = g[i];
		else g[i] = b[i];
	}
}
/// The same for the consecutive tiles tile0 .. tile0 + n - 1 (a whole tile layer).
struct LayerRun {
	int tile0[2], n[2], zz[2];
	uint32_t *buf[2];
};
template <bool PACK>
__global__ void __launch_bounds__(256) k_layer_slices_copy(LayerRun r, uint32_t *vec, int slice_words) {
	const int w = blockIdx.y;
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)r.n[w] * slice_words) return;
	const size_t k = i / slice_words, o = i % slice_words;
	uint32_t *g = vec + ((size_t)(r.tile0[w] + k) * 8 + (size_t)r.zz[w]) * slice_words + o;
	if (PACK) r.buf[w][i] = *g;
	else *g = r.buf[w][i];
}

__global__ void __launch_bounds__(256) k_reduce_partials(const double *part, int n, double *out, int is_max) {
	__shared__ double lds[256];
	double a = is_max ? -INFINITY : 0.0;
	bool nan = false;
	for (int i = threadIdx.x; i < n; i += 256) {
		double x = part[i];
		nan |= x != x;
		a = is_max ? (x > a ? x : a) : a + x;
	}
	lds[threadIdx.x] = nan ? NAN : a;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) {
			double x = lds[threadIdx.x], y = lds[threadIdx.x + o];
			lds[threadIdx.x] = (x != x || y != y) ? NAN : (is_max ? (y > x ? y : x) : x + y);
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) *out = lds[0];
}
/// out[0 .. 2 n) = 0 except out[rank] = max of the first list and out[n + rank] = sum of the second (the fixed-order
/// reductions of k_reduce_partials): a SUM all-reduce of `out` then leaves every rank's pair on every rank (an all-gather
/// that is exact in any order), and the consumers reduce n values instead of their per-workgroup partials.
/// (`psum2`: a third list, summed like the second, for out[2 n + rank]: the single-reduction CG of slab runs; out is 3 n long)
/// out[3 n] = 1 when a device-side wait of this rank's solve has been given up (pcg_state[2], sticky): summed over the ranks it
/// tells EVERY rank at its next poll, so that all of them leave the loop and repeat the solve together (pcg.hip).
__global__ void __launch_bounds__(256) k_gather_pair(const double *pmax, int n_max, const double *psum, int n_sum, const double *psum2,
                                                     int n_sum2, double *out, int n, int rank, const int *pcg_state) {
	__shared__ double lds[3][256];
	double a = -INFINITY, b = 0.0, c2 = 0.0;
	bool nan_a = false, nan_b = false, nan_c = false;
	for (int i = threadIdx.x; i < n_sum2; i += 256) {
		const double x = psum2[i];
		nan_c |= x != x;
		c2 += x;
	}
	lds[2][threadIdx.x] = nan_c ? NAN : c2;
	for (int i = threadIdx.x; i < n_max; i += 256) {
		const double x = pmax[i];
		nan_a |= x != x;
		a = x > a ? x : a;
	}
	for (int i = threadIdx.x; i < n_sum; i += 256) {
		const double x = psum[i];
		nan_b |= x != x;
		b += x;
	}
	lds[0][threadIdx.x] = nan_a ? NAN : a;
	lds[1][threadIdx.x] = nan_b ? NAN : b;
	__syncthreads();
	for (int o = 128; o > 0; o >>= 1) {
		if ((int)threadIdx.x < o) {
			const double x = lds[0][threadIdx.x], y = lds[0][threadIdx.x + o];
			lds[0][threadIdx.x] = (x != x || y != y) ? NAN : (y > x ? y : x);
			const double u = lds[1][threadIdx.x], w = lds[1][threadIdx.x + o];
			lds[1][threadIdx.x] = (u != u || w != w) ? NAN : u + w;
			const double u2 = lds[2][threadIdx.x], w2 = lds[2][threadIdx.x + o];
			lds[2][threadIdx.x] = (u2 != u2 || w2 != w2) ? NAN : u2 + w2;
		}
		__syncthreads();
	}
	for (int i = threadIdx.x; i < 3 * n; i += 256)
		out[i] = i == rank ? lds[0][0] : (i == n + rank ? lds[1][0] : (i == 2 * n + rank ? lds[2][0] : 0.0));
	if (threadIdx.x == 0) out[3 * n] = pcg_state && pcg_state[2] ? 1.0 : 0.0;
}
}  // namespace

int lfa_dist_ensure_xbuf(lfa_sim *s, int which, size_t bytes) {
	if (bytes <= s->xcap[which]) return LFA_OK;
	if (s->xbuf[which]) LFA_HIP(s, hipFree(s->xbuf[which]));
	s->xbuf[which] = nullptr;
	s->xcap[which] = 0;
	size_t want = bytes + bytes / 4 + 4096;
	hipError_t e = hipMalloc(&s->xbuf[which], want);
	if (e != hipSuccess) return lfa_fail(s, LFA_E_OOM, "hipMalloc of a halo buffer (%zu bytes) failed", want);
	s->xcap[which] = want;
	return LFA_OK;
}

/// Own first layer -> lower neighbour's upper ghost layer, own last layer -> upper neighbour's lower ghost layer, in
/// place (a tile layer is a contiguous run of ntx*nty tile ids).
int lfa_dist_exchange_tile_layers_u32(lfa_sim *s, uint32_t *per_tile) {
	if (!s->dist) return LFA_OK;
	const size_t L = (size_t)s->g.ntx * s->g.nty, B = L * 4;
	const bool lo = lfa_has_lo(s), hi = lfa_has_hi(s);
	return s->dist->exchange(s, lo ? per_tile + (size_t)s->slab_lo * L : nullptr, lo ? B : 0,
	                         lo ? per_tile + (size_t)(s->slab_lo - 1) * L : nullptr, lo ? B : 0,
	                         hi ? per_tile + (size_t)(s->slab_hi - 1) * L : nullptr, hi ? B : 0,
	                         hi ? per_tile + (size_t)s->slab_hi * L : nullptr, hi ? B : 0);
}

/// u,v,w,... of the processed tiles of the boundary layers: own layers out, ghost layers in.
int lfa_dist_exchange_fields(lfa_sim *s, int nfields, void *const *fields, const int *elem_bytes) {
	if (!s->dist) return LFA_OK;
	FieldList f;
	f.n = nfields;
	f.tile_words = 0;
	for (int i = 0; i < nfields; ++i) {
		f.ptr[i] = fields[i];
		f.words[i] = LFA_TILE_CELLS * elem_bytes[i] / 4;
		f.tile_words += f.words[i];
	}
	const size_t tb = (size_t)f.tile_words * 4;
	const int Lh = s->g.ntx * s->g.nty;
	const int off[4] = {0, Lh, 2 * Lh, 3 * Lh};  // (every list has a region of one tile layer: core.hip, the binning)
	for (int w = 0; w < 4; ++w) LFA_TRY(lfa_dist_ensure_xbuf(s, w, (size_t)s->n_halo[w] * tb));
	for (int w = 0; w < 2; ++w)
		if (s->n_halo[w]) {
			hipLaunchKernelGGL(k_tiles_copy<true>, dim3(s->n_halo[w]), dim3(256), 0, s->stream, s->halo_tiles + off[w],
			                   s->n_halo[w], f, (uint32_t *)s->xbuf[w]);
			LFA_LAUNCH_CHECK(s);
		}
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], s->n_halo[0] * tb, s->xbuf[2], s->n_halo[2] * tb, s->xbuf[1],
	                          s->n_halo[1] * tb, s->xbuf[3], s->n_halo[3] * tb));
	for (int w = 2; w < 4; ++w)
		if (s->n_halo[w]) {
			hipLaunchKernelGGL(k_tiles_copy<false>, dim3(s->n_halo[w]), dim3(256), 0, s->stream, s->halo_tiles + off[w],
			                   s->n_halo[w], f, (uint32_t *)s->xbuf[w]);
			LFA_LAUNCH_CHECK(s);
		}
	return LFA_OK;
}

/// P2G: plane hz=0 of the first owned layer's slabs goes down (lands as plane 0 of the lower rank's ghost-hi slabs),
/// plane hz=9 of the last owned layer's slabs goes up (lands as plane 9 of the upper rank's ghost-lo slabs).
int lfa_dist_exchange_p2g_planes(lfa_sim *s, float *stage_all) {
	if (!s->dist) return LFA_OK;
	const size_t pb = 600 * 4;
	const int n_send[2] = {lfa_has_lo(s) ? s->n_own_first : 0, lfa_has_hi(s) ? s->n_own_last : 0};
	const int n_recv[2] = {s->n_ghost_lo, s->n_ghost_hi};
	const int send_slot0[2] = {s->p_off, s->p_off + s->n_ptiles - s->n_own_last};
	const int recv_slot0[2] = {0, s->p_off + s->n_ptiles};
	const int send_hz[2] = {0, 9}, recv_hz[2] = {9, 0};
	for (int w = 0; w < 2; ++w) {
		LFA_TRY(lfa_dist_ensure_xbuf(s, w, (size_t)n_send[w] * pb));
		LFA_TRY(lfa_dist_ensure_xbuf(s, 2 + w, (size_t)n_recv[w] * pb));
		if (n_send[w]) {
			hipLaunchKernelGGL(k_planes_copy<true>, dim3(n_send[w]), dim3(128), 0, s->stream, stage_all, send_slot0[w],
			                   n_send[w], send_hz[w], (float *)s->xbuf[w]);
			LFA_LAUNCH_CHECK(s);
		}
	}
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], n_send[0] * pb, s->xbuf[2], n_recv[0] * pb, s->xbuf[1], n_send[1] * pb,
	                          s->xbuf[3], n_recv[1] * pb));
	for (int w = 0; w < 2; ++w)
		if (n_recv[w]) {
			hipLaunchKernelGGL(k_planes_copy<false>, dim3(n_recv[w]), dim3(128), 0, s->stream, stage_all, recv_slot0[w],
			                   n_recv[w], recv_hz[w], (float *)s->xbuf[2 + w]);
			LFA_LAUNCH_CHECK(s);
		}
	return LFA_OK;
}

/// PCG search vector / pressure: slice z=0 of the first owned layer goes down (becomes slice 0 of the lower rank's
/// ghost-hi tiles), slice z=7 of the last owned layer goes up (slice 7 of the upper rank's ghost-lo tiles).
int lfa_dist_exchange_slices(lfa_sim *s, void *vec, int elem_bytes) {
	if (!s->dist) return LFA_OK;
	const int wpe = 16 * elem_bytes;  // 32-bit words per slice
	const size_t sb = (size_t)64 * elem_bytes;
	const int n_send[2] = {lfa_has_lo(s) ? s->n_own_first : 0, lfa_has_hi(s) ? s->n_own_last : 0};
	const int n_recv[2] = {s->n_ghost_lo, s->n_ghost_hi};
	for (int w = 0; w < 2; ++w) {
		LFA_TRY(lfa_dist_ensure_xbuf(s, w, (size_t)n_send[w] * sb));
		LFA_TRY(lfa_dist_ensure_xbuf(s, 2 + w, (size_t)n_recv[w] * sb));
	}
	const SliceRun out{{s->p_off, s->p_off + s->n_ptiles - s->n_own_last}, {n_send[0], n_send[1]}, {0, 7},
	                   {(uint32_t *)s->xbuf[0], (uint32_t *)s->xbuf[1]}};
	const SliceRun in{{0, s->p_off + s->n_ptiles}, {n_recv[0], n_recv[1]}, {7, 0}, {(uint32_t *)s->xbuf[2], (uint32_t *)s->xbuf[3]}};
	if (n_send[0] || n_send[1]) {
		hipLaunchKernelGGL(k_slices_copy<true>, dim3(std::max(n_send[0], n_send[1]), 2), dim3(64), 0, s->stream, s->ptiles_all, out,
		                   (uint32_t *)vec, wpe);
		LFA_LAUNCH_CHECK(s);
	}
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], n_send[0] * sb, s->xbuf[2], n_recv[0] * sb, s->xbuf[1], n_send[1] * sb,
	                          s->xbuf[3], n_recv[1] * sb));
	if (n_recv[0] || n_recv[1]) {
		hipLaunchKernelGGL(k_slices_copy<false>, dim3(std::max(n_recv[0], n_recv[1]), 2), dim3(64), 0, s->stream, s->ptiles_all, in,
		                   (uint32_t *)vec, wpe);
		LFA_LAUNCH_CHECK(s);
	}
	return LFA_OK;
}

int lfa_dist_exchange_layer_slices(lfa_sim *s, void *vec, int elem_bytes, int tiles_per_layer, int lo_layer, int hi_layer) {
	if (!s->dist) return LFA_OK;
	const int sw = 16 * elem_bytes, L = tiles_per_layer;
	const size_t sb = (size_t)64 * elem_bytes * L;
	const bool on[2] = {lfa_has_lo(s), lfa_has_hi(s)};
	if (!on[0] && !on[1]) return LFA_OK;
	for (int w = 0; w < 2; ++w)
		if (on[w]) {
			LFA_TRY(lfa_dist_ensure_xbuf(s, w, sb));
			LFA_TRY(lfa_dist_ensure_xbuf(s, 2 + w, sb));
		}
	const LayerRun out{{lo_layer * L, (hi_layer - 1) * L}, {on[0] ? L : 0, on[1] ? L : 0}, {0, 7},
	                   {(uint32_t *)s->xbuf[0], (uint32_t *)s->xbuf[1]}};
	const LayerRun in{{(lo_layer - 1) * L, hi_layer * L}, {on[0] ? L : 0, on[1] ? L : 0}, {7, 0},
	                  {(uint32_t *)s->xbuf[2], (uint32_t *)s->xbuf[3]}};
	const dim3 grid((unsigned)(((size_t)L * sw + 255) / 256), 2);
	hipLaunchKernelGGL(k_layer_slices_copy<true>, grid, dim3(256), 0, s->stream, out, (uint32_t *)vec, sw);
	LFA_LAUNCH_CHECK(s);
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], on[0] ? sb : 0, s->xbuf[2], on[0] ? sb : 0, s->xbuf[1], on[1] ? sb : 0, s->xbuf[3],
	                          on[1] ? sb : 0));
	hipLaunchKernelGGL(k_layer_slices_copy<false>, grid, dim3(256), 0, s->stream, in, (uint32_t *)vec, sw);
	LFA_LAUNCH_CHECK(s);
	return LFA_OK;
}

/// Local fixed-order reduction of the per-workgroup partials, then the all-reduce over ranks; result in dist_red[slot].
int lfa_dist_allreduce(lfa_sim *s, const double *partials, int n, int slot, bool is_max) {
	hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s->stream, partials, n, s->dist_red + slot, is_max ? 1 : 0);
	LFA_LAUNCH_CHECK(s);
	return s->dist->allreduce(s, s->dist_red + slot, 1, is_max);
}


/// The signed max of `pmax` and the sum of `psum` of every rank in ONE collective: afterwards lfa_dist_gather_buf(s, parity)
/// holds [max of rank 0 .. n-1 | sum of rank 0 .. n-1] on every rank.
double *lfa_dist_gather_buf(lfa_sim *s, int parity) { return s->dist_red + 64 + (size_t)parity * (3 * s->dist->nranks + 1); }
int lfa_dist_gather_pair(lfa_sim *s, const double *pmax, int n_max, const double *psum, int n_sum, int parity) {
	return lfa_dist_gather_triple(s, pmax, n_max, psum, n_sum, nullptr, 0, parity);
}
/// The same with the sum of a third list behind them: [max | sum | sum2] of every rank, still ONE collective.
int lfa_dist_gather_triple(lfa_sim *s, const double *pmax, int n_max, const double *psum, int n_sum, const double *psum2, int n_sum2,
                           int parity) {
	const int n = s->dist->nranks;
	if (n > 32) return lfa_fail(s, LFA_E_UNSUPPORTED, "more than 32 slabs");
	double *out = lfa_dist_gather_buf(s, parity);
	hipLaunchKernelGGL(k_gather_pair, dim3(1), dim3(256), 0, s->stream, pmax, n_max, psum, n_sum, psum2, n_sum2, out, n, s->dist->rank,
	                   (const int *)s->pcg_state);
	LFA_LAUNCH_CHECK(s);
	return s->dist->allreduce_buf(s, out, (size_t)3 * n + 1, LFA_RED_F64, false);  // [max | sum | sum2 | ranks that gave a wait up]
}

// ================================================================================================= particle migration
namespace {
/// Particles whose new cell lies outside the owned tile layers leave for the neighbour slab: their 17-word records are
/// appended to the lo / hi send buffer (wave-aggregated append) and their key is invalidated (the next binning drops it).
__global__ void __launch_bounds__(256)
k_pack_leavers(size_t n, ParticleSoA p, int tiles_per_layer, int slab_lo, int slab_hi, uint32_t *counters, uint32_t *buf_lo,
               uint32_t *buf_hi, int count_only, const float *c_home, size_t c_home_stride, ParticleSoA pvc, const uint32_t *from,
               int c_deferred) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	int dest = -1;
	uint32_t key = 0xFFFFFFFFu;
	if (i < n) {
		key = p.key[i];
		if (key != 0xFFFFFFFFu) {
			const int tz = (int)(key >> 9) / tiles_per_layer;
			dest = tz < slab_lo ? 0 : (tz >= slab_hi ? 1 : -1);
		}
	}
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int d = 0; d < 2; ++d) {
		const unsigned long long m = __ballot(dest == d);
		if (!m) continue;
		const int leader = __ffsll((long long)m) - 1;
		uint32_t base = 0;
		if (lane == leader) base = atomicAdd(&counters[d], (uint32_t)__popcll(m));
		base = __shfl(base, leader, 64);
		if (dest == d && !count_only) {
			uint32_t *rec = (d == 0 ? buf_lo : buf_hi) + (size_t)(base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) * 17;
			rec[0] = key;
			// (a deferred binning, lfa_sim::vc_pending: v - and APIC's C - are still where they were before it)
			const size_t j = from ? (size_t)from[i] : i;
#pragma unroll
			for (int k = 0; k < 3; ++k) {
				rec[1 + k] = __float_as_uint(p.t[k][i]);
				rec[4 + k] = __float_as_uint(from ? pvc.v[k][j] : p.v[k][i]);
			}
#pragma unroll
			for (int k = 0; k < 9; ++k)
				rec[7 + k] = __float_as_uint(c_home ? c_home[k * c_home_stride + p.id[i]]  // (lfa_sim::c_home)
				                                    : (from && c_deferred ? pvc.c[k][j] : p.c[k][i]));
			rec[16] = p.id[i];
			p.key[i] = 0xFFFFFFFFu;
		}
	}
}
__global__ void k_offset_u32(const uint32_t *in, uint32_t *out, int n, uint32_t base) {
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) out[i] = in[i] + base;
}
/// `from` (a deferred binning is pending): v and C of the arrival at d go to record `ext + i` of the other buffer and from[d] says so.
__global__ void __launch_bounds__(256) k_unpack_arrivals(size_t n, const uint32_t *buf, ParticleSoA p, size_t at, float *c_home,
                                                         size_t c_home_stride, ParticleSoA pvc, uint32_t *from, size_t ext) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t *rec = buf + i * 17;
	const size_t d = at + i, e = from ? ext + i : d;
	const ParticleSoA &q = from ? pvc : p;
	p.key[d] = rec[0];
#pragma unroll
	for (int k = 0; k < 3; ++k) {
		p.t[k][d] = __uint_as_float(rec[1 + k]);
		q.v[k][e] = __uint_as_float(rec[4 + k]);
	}
#pragma unroll
	for (int k = 0; k < 9; ++k) {
		if (c_home) c_home[k * c_home_stride + rec[16]] = __uint_as_float(rec[7 + k]);
		else q.c[k][e] = __uint_as_float(rec[7 + k]);
	}
	p.id[d] = rec[16];
	if (from) from[d] = (uint32_t)e;
}
}  // namespace

int lfa_particles_reserve(lfa_sim *s, size_t n_keep, size_t n_total);  // core.hip

/// After a stage that moves particles (advect+collide, correct+collide): hand the particles that left the slab to the
/// neighbour ranks and take theirs. Moves of more than one slab per step are not handled (CFL bounds a step to 3 cells).
/// `vc_dead`: the caller is about to run the G2P, which overwrites v (and, APIC, C) of every live particle without reading them
/// (PIC, APIC; FLIP blends with the old velocity): a deferred binning then need not be completed for the leavers' sake - the
/// records travel with whatever v / C the current buffer holds.
int lfa_dist_migrate(lfa_sim *s, bool vc_dead) {
	if (!s->dist) return LFA_OK;
	// (leavers travel as whole records. A deferred binning stays deferred - round 4: completing it for everybody cost a FLIP run
	// 0.12 ms per migration at C3 -: the pack kernel reads v, C where they are, the arrivals' go behind the other buffer's records)
	(void)vc_dead;
	const size_t n = s->binned ? s->np_live : s->np;
	uint32_t *cnt = (uint32_t *)(s->dist_red + 32);  // [0,1] leaving lo/hi, [2,3] arriving from lo/hi
	LFA_HIP(s, hipMemsetAsync(cnt, 0, 16, s->stream));
	const int tpl = s->g.ntx * s->g.nty;
	const dim3 grid((unsigned)((n + 255) / 256 ? (n + 255) / 256 : 1));
	ParticleSoA &p = s->pb[s->cur];
	hipLaunchKernelGGL(k_pack_leavers, grid, dim3(256), 0, s->stream, n, p, tpl, s->slab_lo, s->slab_hi, cnt, (uint32_t *)nullptr,
	                   (uint32_t *)nullptr, 1, (const float *)nullptr, (size_t)0, p, (const uint32_t *)nullptr, 0);
	LFA_LAUNCH_CHECK(s);
	// how many arrive: the neighbours' leave counts
	LFA_TRY(s->dist->exchange(s, lfa_has_lo(s) ? cnt + 0 : nullptr, lfa_has_lo(s) ? 4 : 0, lfa_has_lo(s) ? cnt + 2 : nullptr,
	                          lfa_has_lo(s) ? 4 : 0, lfa_has_hi(s) ? cnt + 1 : nullptr, lfa_has_hi(s) ? 4 : 0,
	                          lfa_has_hi(s) ? cnt + 3 : nullptr, lfa_has_hi(s) ? 4 : 0));
	uint32_t h[4];
	LFA_HIP(s, hipMemcpyAsync(h, cnt, 16, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	if (!lfa_has_lo(s)) h[0] = h[2] = 0;  // nothing can leave through a domain wall (positions are clamped into the grid)
	if (!lfa_has_hi(s)) h[1] = h[3] = 0;
	// (no early-out when nothing crosses this rank's faces: every rank issues the same sequence of transport calls)
	for (int w = 0; w < 4; ++w) LFA_TRY(lfa_dist_ensure_xbuf(s, w, (size_t)h[w] * 68));
	// (room for the arrivals: in the current buffer behind the live records, and - a deferred binning - for their v, C in the other
	// one behind the records vc_src points at; a reallocation completes the deferred binning first)
	LFA_TRY(lfa_particles_reserve(s, n, std::max(n, s->vc_pending ? s->vc_extent : (size_t)0) + h[2] + h[3]));
	ParticleSoA &q = s->pb[s->cur];
	const bool pend = s->vc_pending;
	const ParticleSoA &qo = s->pb[s->cur ^ 1];
	uint32_t *from = pend ? s->vc_src : nullptr;
	LFA_HIP(s, hipMemsetAsync(cnt, 0, 8, s->stream));
	// (PIC / FLIP: C lives in its home array, indexed by the job-wide id - see lfa_sim::c_home)
	if (s->c_home_valid) LFA_TRY(lfa_c_home_ensure(s, (size_t)s->next_global_id));
	float *chome = s->c_home_valid ? s->c_home : nullptr;
	hipLaunchKernelGGL(k_pack_leavers, grid, dim3(256), 0, s->stream, n, q, tpl, s->slab_lo, s->slab_hi, cnt,
	                   (uint32_t *)s->xbuf[0], (uint32_t *)s->xbuf[1], 0, (const float *)chome, s->c_home_cap, pend ? qo : q,
	                   (const uint32_t *)from, s->vc_with_c ? 1 : 0);
	LFA_LAUNCH_CHECK(s);
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], (size_t)h[0] * 68, s->xbuf[2], (size_t)h[2] * 68, s->xbuf[1], (size_t)h[1] * 68,
	                          s->xbuf[3], (size_t)h[3] * 68));
	size_t at = n;
	for (int w = 2; w < 4; ++w)
		if (h[w]) {
			hipLaunchKernelGGL(k_unpack_arrivals, dim3((h[w] + 255) / 256), dim3(256), 0, s->stream, (size_t)h[w],
			                   (const uint32_t *)s->xbuf[w], q, at, chome, s->c_home_cap, pend ? qo : q, from, s->vc_extent);
			LFA_LAUNCH_CHECK(s);
			at += h[w];
			if (pend) s->vc_extent += h[w];
		}
	// the next binning scans [0, at): leavers carry an invalid key and are dropped there
	s->np_live = at;
	// lfa_num_particles: the resident count follows the hand-over at once (binned: np counted the live particles of the last
	// binning; unbinned: np is the extent of the array, holes included, until the next binning compacts it)
	if (!s->binned) s->np = at;
	else s->np = s->np - h[0] - h[1] + h[2] + h[3];
	s->arrivals_at = n;  // [n, at): in no tile's range until the next binning (the G2P takes them through its leaver path)
	s->n_arrivals = at - n;
	s->holes = true;
	s->vmax2_valid = false;  // the cached max |v|^2 (written by the last G2P) knows nothing of arrivals: lfa_cfl reduces again
	return LFA_OK;
}

/// Position correction needs the particles within one cell across the slab face: the (key, t) of the neighbours' adjacent
/// tile layers are appended behind the live particles (ghost particles, read-only), with tile_count / tile_start of the
/// ghost tiles pointing at them. The binned array is ordered by tile, so a tile layer is one contiguous range: no packing.
int lfa_dist_exchange_ghost_particles(lfa_sim *s) {
	if (!s->dist) return LFA_OK;
	const int L = s->g.ntx * s->g.nty;
	const size_t own_lo = (size_t)s->slab_lo * L, own_hi = (size_t)s->slab_hi * L;
	// per-tile counts of the boundary layers -> the neighbours' ghost layers (in place)
	LFA_TRY(lfa_dist_exchange_tile_layers_u32(s, s->tile_count));
	// my boundary ranges
	uint32_t hs[4];  // tile_start at own_lo, own_lo+L, own_hi-L, own_hi
	const size_t marks[4] = {own_lo, own_lo + L < own_hi ? own_lo + L : own_hi, own_hi - L > own_lo ? own_hi - L : own_lo, own_hi};
	for (int k = 0; k < 4; ++k)
		LFA_HIP(s, hipMemcpyAsync(&hs[k], s->tile_start + marks[k], 4, hipMemcpyDeviceToHost, s->stream));
	// ghost totals: sum of the received counts (scan over the ghost layer gives the per-tile starts as well)
	uint32_t *tot = (uint32_t *)(s->dist_red + 40);
	size_t n_g[2] = {0, 0};
	uint32_t hg[2] = {0, 0};
	if (lfa_has_lo(s)) LFA_TRY(lfa_exclusive_scan_u32(s, s->tile_count + own_lo - L, s->tile_scan + own_lo - L, (size_t)L, tot));
	if (lfa_has_hi(s)) LFA_TRY(lfa_exclusive_scan_u32(s, s->tile_count + own_hi, s->tile_scan + own_hi, (size_t)L, tot + 1));
	LFA_HIP(s, hipMemcpyAsync(hg, tot, 8, hipMemcpyDeviceToHost, s->stream));
	LFA_HIP(s, hipStreamSynchronize(s->stream));
	n_g[0] = lfa_has_lo(s) ? hg[0] : 0;
	n_g[1] = lfa_has_hi(s) ? hg[1] : 0;
	const size_t n = s->np_live;
	LFA_TRY(lfa_particles_reserve(s, n, n + n_g[0] + n_g[1]));
	ParticleSoA &p = s->pb[s->cur];
	// upper ghosts first: tile_start[own_hi] == n already, so the upper ghost layer continues the owned numbering
	const size_t at_hi = n, at_lo = n + n_g[1];
	if (lfa_has_hi(s)) {
		hipLaunchKernelGGL(k_offset_u32, dim3((L + 255) / 256), dim3(256), 0, s->stream, s->tile_scan + own_hi, s->tile_start + own_hi,
		                   L, (uint32_t)at_hi);
		LFA_LAUNCH_CHECK(s);
	}
	if (lfa_has_lo(s)) {
		hipLaunchKernelGGL(k_offset_u32, dim3((L + 255) / 256), dim3(256), 0, s->stream, s->tile_scan + own_lo - L,
		                   s->tile_start + own_lo - L, L, (uint32_t)at_lo);
		LFA_LAUNCH_CHECK(s);
	}
	const size_t send_lo_at = hs[0], send_lo_n = lfa_has_lo(s) ? hs[1] - hs[0] : 0;
	const size_t send_hi_at = hs[2], send_hi_n = lfa_has_hi(s) ? hs[3] - hs[2] : 0;
	// key, in-cell position and the global id (the mesher orders the particles of a cell by it)
	uint32_t *arr[5] = {p.key, (uint32_t *)p.t[0], (uint32_t *)p.t[1], (uint32_t *)p.t[2], p.id};
	for (int a = 0; a < 5; ++a)
		LFA_TRY(s->dist->exchange(s, arr[a] + send_lo_at, send_lo_n * 4, arr[a] + at_lo, n_g[0] * 4, arr[a] + send_hi_at,
		                          send_hi_n * 4, arr[a] + at_hi, n_g[1] * 4));
	s->ghost_at[0] = at_lo;
	s->ghost_at[1] = at_hi;
	s->n_ghost_particles = n_g[0] + n_g[1];
	return LFA_OK;
}

// ================================================================================================= in-process transport
struct lfa_hub {
	int n = 0;
	std::mutex m;
	std::condition_variable cv;
	int arrived = 0;
	long generation = 0;
	struct Mail {
		const void *lo = nullptr, *hi = nullptr;
		size_t n_lo = 0, n_hi = 0;
		double val = 0.0;
		std::vector<uint8_t> buf;  // array all-reduce: this rank's contribution, staged on the host
	};
	std::vector<Mail> mail;
	bool failed = false;
	/// false: a peer failed or did not arrive within 20 s (a rank that errors out of a step must not leave the others
	/// waiting forever)
	bool barrier() {
		std::unique_lock<std::mutex> lk(m);
		if (failed) return false;
		long gen = generation;
		if (++arrived == n) {
			arrived = 0;
			++generation;
			cv.notify_all();
			return true;
		}
		if (!cv.wait_for(lk, std::chrono::seconds(20), [&] { return generation != gen || failed; }) || failed) {
			failed = true;
			cv.notify_all();
			return false;
		}
		return true;
	}
	void fail() {
		std::lock_guard<std::mutex> lk(m);
		failed = true;
		cv.notify_all();
	}
};

namespace {
struct LocalDist : lfa_dist {
	lfa_hub *hub = nullptr;
	int exchange_impl(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
	             size_t n_send_hi, void *recv_hi, size_t n_recv_hi) override {
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		lfa_hub::Mail &me = hub->mail[rank];
		me.lo = send_lo; me.n_lo = n_send_lo; me.hi = send_hi; me.n_hi = n_send_hi;
		if (!hub->barrier()) return lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		int rc = LFA_OK;
		// sizes are checked even when this side expects nothing: over RCCL an unmatched ncclSend (or ncclRecv) hangs, so the
		// in-process transport must refuse what the real one cannot do
		if (rank > 0) {
			const lfa_hub::Mail &nb = hub->mail[rank - 1];
			if (nb.n_hi != n_recv_lo) rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu",
			                                        rank - 1, nb.n_hi, n_recv_lo);
			else if (n_recv_lo && hipMemcpy(recv_lo, nb.hi, n_recv_lo, hipMemcpyDeviceToDevice) != hipSuccess) rc = LFA_E_HIP;
		}
		if (rank + 1 < nranks && rc == LFA_OK) {
			const lfa_hub::Mail &nb = hub->mail[rank + 1];
			if (nb.n_lo != n_recv_hi) rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu",
			                                        rank + 1, nb.n_lo, n_recv_hi);
			else if (n_recv_hi && hipMemcpy(recv_hi, nb.lo, n_recv_hi, hipMemcpyDeviceToDevice) != hipSuccess) rc = LFA_E_HIP;
		}
		if (rc == LFA_OK && hipDeviceSynchronize() != hipSuccess) rc = LFA_E_HIP;  // D2D hipMemcpy may return early
		if (rc != LFA_OK) hub->fail();
		if (!hub->barrier() && rc == LFA_OK)  // nobody reuses a send buffer before every copy out of it is done
			rc = lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		return rc;
	}
	int allreduce_impl(lfa_sim *s, double *dev, int count, bool is_max) override {
		if (count != 1) return lfa_fail(s, LFA_E_INVALID, "local all-reduce supports one scalar");
		double v = 0.0;
		LFA_HIP(s, hipMemcpyAsync(&v, dev, 8, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		hub->mail[rank].val = v;
		if (!hub->barrier()) return lfa_fail(s, LFA_E_HIP, "slab all-reduce: a peer rank failed or timed out");
		double r = hub->mail[0].val;
		for (int i = 1; i < nranks; ++i) {  // fixed rank order: every rank computes the identical value
			const double x = hub->mail[i].val;
			r = is_max ? ((x != x || r != r) ? NAN : (x > r ? x : r)) : r + x;
		}
		if (!hub->barrier()) return lfa_fail(s, LFA_E_HIP, "slab all-reduce: a peer rank failed or timed out");
		LFA_HIP(s, hipMemcpyAsync(dev, &r, 8, hipMemcpyHostToDevice, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		return LFA_OK;
	}
	int allreduce_buf_impl(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) override {
		const size_t es = dtype == LFA_RED_U8 ? 1 : (dtype == LFA_RED_F32 ? 4 : 8), bytes = count * es;
		std::vector<uint8_t> &mine = hub->mail[rank].buf;
		mine.resize(bytes);
		LFA_HIP(s, hipMemcpyAsync(mine.data(), dev, bytes, hipMemcpyDeviceToHost, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		if (!hub->barrier()) return lfa_fail(s, LFA_E_HIP, "slab all-reduce: a peer rank failed or timed out");
		std::vector<uint8_t> out(hub->mail[0].buf);  // fixed rank order: every rank computes the identical array
		bool ok = out.size() == bytes;
		for (int r = 1; r < nranks && ok; ++r) {
			const std::vector<uint8_t> &o = hub->mail[r].buf;
			if (o.size() != bytes) { ok = false; break; }
			if (dtype == LFA_RED_U8) {
				for (size_t i = 0; i < count; ++i) out[i] = is_max ? std::max(out[i], o[i]) : (uint8_t)(out[i] + o[i]);
			} else if (dtype == LFA_RED_F32) {
				float *a = (float *)out.data();
				const float *b = (const float *)o.data();
				for (size_t i = 0; i < count; ++i) a[i] = is_max ? std::max(a[i], b[i]) : a[i] + b[i];
			} else {
				double *a = (double *)out.data();
				const double *b = (const double *)o.data();
				for (size_t i = 0; i < count; ++i) a[i] = is_max ? std::max(a[i], b[i]) : a[i] + b[i];
			}
		}
		if (!ok) hub->fail();
		if (!hub->barrier() || !ok) return lfa_fail(s, LFA_E_HIP, "slab all-reduce: size mismatch or a peer rank failed");
		LFA_HIP(s, hipMemcpyAsync(dev, out.data(), bytes, hipMemcpyHostToDevice, s->stream));
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		return LFA_OK;
	}
};

// ================================================================================================= RCCL transport
struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // (optional)
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool load() {
		if (lib) return true;
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (lib) break;
		}
		if (!lib) return false;
#define LOAD(f) f = (decltype(f))dlsym(lib, "nccl" #f)
		LOAD(GetUniqueId); LOAD(CommInitRank); LOAD(CommDestroy); LOAD(Send); LOAD(Recv); LOAD(AllReduce);
		LOAD(GroupStart); LOAD(GroupEnd); LOAD(GetErrorString); LOAD(CommAbort);
#undef LOAD
		return GetUniqueId && CommInitRank && CommDestroy && Send && Recv && AllReduce && GroupStart && GroupEnd;
	}
};
RcclApi g_rccl;

struct RcclDist : lfa_dist {
	ncclComm_t comm = nullptr;
	~RcclDist() override {
		// A healthy communicator is drained and destroyed (lfa_destroy has synchronised the handle's stream: every send / recv /
		// all-reduce this rank enqueued is complete, so a faster rank never tears its side down under a slower peer's last call).
		// ncclCommAbort - which releases the communicator without waiting for anybody - is for the broken case only: a transport
		// call failed, or the job gives the communicator up because a PEER could not create its own (bench.py's fallback:
		// lfa_dist_abandon).
		if (!comm) return;
		if (broken && g_rccl.CommAbort) (void)g_rccl.CommAbort(comm);
		else (void)g_rccl.CommDestroy(comm);
	}
#define NCCL_TRY(s, call)                                                                                \
	do {                                                                                                  \
		ncclResult_t r_ = (call);                                                                         \
		if (r_ != ncclSuccess) {                                                                          \
			broken = true;                                                                                \
			return lfa_fail((s), LFA_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
		}                                                                                                 \
	} while (0)
	int exchange_impl(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
	             size_t n_send_hi, void *recv_hi, size_t n_recv_hi) override {
		// sizes are known on both sides (they follow from the flags exchanged first), so no size handshake is needed
		NCCL_TRY(s, g_rccl.GroupStart());
		if (rank > 0) {
			if (n_send_lo) NCCL_TRY(s, g_rccl.Send(send_lo, n_send_lo, ncclUint8, rank - 1, comm, s->stream));
			if (n_recv_lo) NCCL_TRY(s, g_rccl.Recv(recv_lo, n_recv_lo, ncclUint8, rank - 1, comm, s->stream));
		}
		if (rank + 1 < nranks) {
			if (n_send_hi) NCCL_TRY(s, g_rccl.Send(send_hi, n_send_hi, ncclUint8, rank + 1, comm, s->stream));
			if (n_recv_hi) NCCL_TRY(s, g_rccl.Recv(recv_hi, n_recv_hi, ncclUint8, rank + 1, comm, s->stream));
		}
		NCCL_TRY(s, g_rccl.GroupEnd());
		return LFA_OK;
	}
	int allreduce_impl(lfa_sim *s, double *dev, int count, bool is_max) override {
		NCCL_TRY(s, g_rccl.AllReduce(dev, dev, (size_t)count, ncclFloat64, is_max ? ncclMax : ncclSum, comm, s->stream));
		return LFA_OK;
	}
	int allreduce_buf_impl(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) override {
		const ncclDataType_t t = dtype == LFA_RED_U8 ? ncclUint8 : (dtype == LFA_RED_F32 ? ncclFloat32 : ncclFloat64);
		NCCL_TRY(s, g_rccl.AllReduce(dev, dev, count, t, is_max ? ncclMax : ncclSum, comm, s->stream));
		return LFA_OK;
	}
};

// ================================================================================================= shared-memory transport
// One PROCESS per rank, messages staged through a POSIX shared-memory segment on the host: device -> segment, rendezvous,
// segment -> device. No peer access and no RCCL, so the ranks may even share one GPU. It is the functional multi-process
// transport (bench.py --transport shm, and the fallback when the RCCL communicator cannot be created); its cost is two PCIe
// crossings per message, so it is never the one a throughput figure should be quoted on.
struct ShmHeader {
	std::atomic<uint32_t> magic;
	uint32_t nranks;
	uint64_t slot_bytes;
	std::atomic<uint32_t> failed;
	alignas(64) std::atomic<uint32_t> arrived;
	alignas(64) std::atomic<uint64_t> generation;
};
struct alignas(64) ShmMail {
	uint64_t n_lo, n_hi, n_buf;
	char device[16];  // PCI bus id of the rank's GPU
};
enum : uint32_t { SHM_MAGIC = 0x4c464131u };

/// how long a rank waits for its peers (segment creation, every rendezvous) before it fails the job: LFA_SHM_TIMEOUT_S, default 60
static int shm_timeout_s() {
	static const int t = [] {
		const char *e = getenv("LFA_SHM_TIMEOUT_S");
		const int v = e ? atoi(e) : 60;
		return v > 0 ? v : 60;
	}();
	return t;
}

struct ShmDist : lfa_dist {
	void *base = nullptr;
	size_t map_bytes = 0;
	bool registered = false;
	ShmHeader *hdr = nullptr;
	ShmMail *mail = nullptr;
	uint8_t *slots = nullptr;
	size_t slot_bytes = 0;
	std::vector<uint8_t> acc;
	~ShmDist() override {
		if (registered) (void)hipHostUnregister(base);
		if (base) munmap(base, map_bytes);
	}
	uint8_t *slot(int r) const { return slots + (size_t)r * slot_bytes; }
	/// all ranks arrive, or false when one of them failed / did not arrive in time (LFA_SHM_TIMEOUT_S, default 60 s)
	bool barrier() {
		if (hdr->failed.load(std::memory_order_acquire)) return false;
		const uint64_t gen = hdr->generation.load(std::memory_order_acquire);
		if (hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
			hdr->arrived.store(0, std::memory_order_relaxed);
			hdr->generation.store(gen + 1, std::memory_order_release);
			return true;
		}
		const auto t0 = std::chrono::steady_clock::now();
		for (unsigned spin = 0;; ++spin) {
			if (hdr->generation.load(std::memory_order_acquire) != gen) return true;
			if (hdr->failed.load(std::memory_order_acquire)) return false;
			if (spin < 2000) continue;
			sched_yield();
			if ((spin & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(shm_timeout_s())) {
				hdr->failed.store(1, std::memory_order_release);
				return false;
			}
		}
	}
	void fail() { hdr->failed.store(1, std::memory_order_release); }
	int exchange_impl(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
	                  size_t n_send_hi, void *recv_hi, size_t n_recv_hi) override {
		if (rank == 0) n_send_lo = 0;  // as over RCCL: nothing leaves through a face without a neighbour
		if (rank + 1 == nranks) n_send_hi = 0;
		// Sizes first; then the messages in rounds of half a slot per direction (lo part at 0, hi part at half a slot) - a message
		// larger than that (the ghost particles of the fixed C4 domain: 34 MB per face) takes several rounds, the same number on
		// every rank: the maximum over everybody's posted sizes.
		int rc = LFA_OK;
		mail[rank].n_lo = n_send_lo;
		mail[rank].n_hi = n_send_hi;
		if (!barrier()) return lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		if (rank > 0 && mail[rank - 1].n_hi != n_recv_lo)
			rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu", rank - 1, (size_t)mail[rank - 1].n_hi, n_recv_lo);
		if (rank + 1 < nranks && rc == LFA_OK && mail[rank + 1].n_lo != n_recv_hi)
			rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu", rank + 1, (size_t)mail[rank + 1].n_lo, n_recv_hi);
		const size_t half = (slot_bytes / 2) & ~(size_t)255;
		size_t longest = 0;
		for (int r = 0; r < nranks; ++r) longest = std::max(longest, std::max((size_t)mail[r].n_lo, (size_t)mail[r].n_hi));
		const size_t rounds = longest ? (longest + half - 1) / half : 0;
		if (rc != LFA_OK) fail();
		auto part = [&](size_t total, size_t off) { return off < total ? std::min(half, total - off) : (size_t)0; };
		for (size_t k = 0; k < rounds; ++k) {
			const size_t off = k * half;
			uint8_t *mine = slot(rank);
			const size_t a = part(n_send_lo, off), b = part(n_send_hi, off);
			if (rc == LFA_OK) {
				if (a && hipMemcpyAsync(mine, (const uint8_t *)send_lo + off, a, hipMemcpyDeviceToHost, s->stream) != hipSuccess) rc = LFA_E_HIP;
				if (b && hipMemcpyAsync(mine + half, (const uint8_t *)send_hi + off, b, hipMemcpyDeviceToHost, s->stream) != hipSuccess) rc = LFA_E_HIP;
				if ((a || b) && hipStreamSynchronize(s->stream) != hipSuccess) rc = LFA_E_HIP;
				if (rc != LFA_OK) fail();
			}
			if (!barrier()) return rc != LFA_OK ? rc : lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
			const size_t c = rank > 0 ? part(n_recv_lo, off) : 0, d = rank + 1 < nranks ? part(n_recv_hi, off) : 0;
			if (c && hipMemcpyAsync((uint8_t *)recv_lo + off, slot(rank - 1) + half, c, hipMemcpyHostToDevice, s->stream) != hipSuccess) rc = LFA_E_HIP;
			if (d && hipMemcpyAsync((uint8_t *)recv_hi + off, slot(rank + 1), d, hipMemcpyHostToDevice, s->stream) != hipSuccess) rc = LFA_E_HIP;
			if ((c || d) && hipStreamSynchronize(s->stream) != hipSuccess) rc = LFA_E_HIP;
			if (rc != LFA_OK) fail();
			if (!barrier() && rc == LFA_OK)  // nobody overwrites its slot before every copy out of it is done
				rc = lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
			if (rc != LFA_OK) return rc;
		}
		// (an empty exchange still ends with a barrier: nobody posts the sizes of its next message while a peer reads these)
		if (!rounds && !barrier() && rc == LFA_OK) rc = lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		return rc;
	}
	int allreduce_impl(lfa_sim *s, double *dev, int count, bool is_max) override {
		return allreduce_buf_impl(s, dev, (size_t)count, LFA_RED_F64, is_max);
	}
	int allreduce_buf_impl(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) override {
		const size_t es = dtype == LFA_RED_U8 ? 1 : (dtype == LFA_RED_F32 ? 4 : 8);
		// (an array larger than the slot goes through in slot-sized pieces: the count is the same on every rank, so are the rounds)
		const size_t per = (slot_bytes / es) & ~(size_t)63;
		size_t first = 0;
		do {
			const size_t n = std::min(per, count - first);
			const int rc = allreduce_piece(s, (uint8_t *)dev + first * es, n, dtype, is_max);
			if (rc != LFA_OK) return rc;
			first += n;
		} while (first < count);
		return LFA_OK;
	}
	int allreduce_piece(lfa_sim *s, void *dev, size_t count, int dtype, bool is_max) {
		const size_t es = dtype == LFA_RED_U8 ? 1 : (dtype == LFA_RED_F32 ? 4 : 8), bytes = count * es;
		int rc = LFA_OK;
		if (bytes && (hipMemcpyAsync(slot(rank), dev, bytes, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
		              hipStreamSynchronize(s->stream) != hipSuccess))
			rc = LFA_E_HIP;
		mail[rank].n_buf = bytes;
		if (rc != LFA_OK) fail();
		if (!barrier()) return rc != LFA_OK ? rc : lfa_fail(s, LFA_E_HIP, "slab all-reduce: a peer rank failed or timed out");
		acc.resize(bytes);
		memcpy(acc.data(), slot(0), bytes);  // fixed rank order: every rank computes the identical array
		bool ok = mail[0].n_buf == bytes;
		for (int r = 1; r < nranks && ok; ++r) {
			if (mail[r].n_buf != bytes) { ok = false; break; }
			if (dtype == LFA_RED_U8) {
				uint8_t *a = acc.data();
				const uint8_t *b = slot(r);
				for (size_t i = 0; i < count; ++i) a[i] = is_max ? std::max(a[i], b[i]) : (uint8_t)(a[i] + b[i]);
			} else if (dtype == LFA_RED_F32) {
				float *a = (float *)acc.data();
				const float *b = (const float *)slot(r);
				for (size_t i = 0; i < count; ++i) a[i] = is_max ? std::max(a[i], b[i]) : a[i] + b[i];
			} else {
				double *a = (double *)acc.data();
				const double *b = (const double *)slot(r);
				for (size_t i = 0; i < count; ++i) {
					const double x = b[i], y = a[i];
					a[i] = is_max ? ((x != x || y != y) ? NAN : (x > y ? x : y)) : y + x;
				}
			}
		}
		if (!ok) fail();
		if (!barrier() || !ok) return lfa_fail(s, LFA_E_HIP, "slab all-reduce: size mismatch or a peer rank failed");
		if (bytes) {
			LFA_HIP(s, hipMemcpyAsync(dev, acc.data(), bytes, hipMemcpyHostToDevice, s->stream));
			LFA_HIP(s, hipStreamSynchronize(s->stream));
		}
		return LFA_OK;
	}
	/// rank 0 creates the segment, the others wait for it; false with `why` set when it cannot be had
	bool open(const char *name, int device, const char **why) {
		size_t mb = 32, kb = 0;
		if (const char *e = getenv("LFA_SHM_SLOT_MB")) mb = (size_t)std::max(1, atoi(e));
		if (const char *e = getenv("LFA_SHM_SLOT_KB")) kb = (size_t)std::max(4, atoi(e));  // (tests: messages in several rounds)
		const size_t head = 4096 + (((size_t)nranks * sizeof(ShmMail) + 4095) & ~(size_t)4095);
		int fd = -1;
		if (rank == 0) {
			shm_unlink(name);
			fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
			if (fd < 0) { *why = "shm_open (create) failed"; return false; }
			slot_bytes = kb ? kb << 10 : mb << 20;
			map_bytes = head + (size_t)nranks * slot_bytes;
			// posix_fallocate, not only ftruncate: a segment larger than /dev/shm must fail here, not with SIGBUS at the first touch
			if (ftruncate(fd, (off_t)map_bytes) != 0 || posix_fallocate(fd, 0, (off_t)map_bytes) != 0) {
				::close(fd);
				shm_unlink(name);
				*why = "the segment does not fit /dev/shm (LFA_SHM_SLOT_MB sets the per-rank slot, default 32)";
				return false;
			}
		} else {
			const auto t0 = std::chrono::steady_clock::now();
			struct stat st;
			for (;;) {  // the segment exists and has its final size
				if (fd < 0) fd = shm_open(name, O_RDWR, 0600);
				if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size > head) break;
				if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(shm_timeout_s())) {
					if (fd >= 0) ::close(fd);
					*why = "rank 0 did not create the segment in time (LFA_SHM_TIMEOUT_S)";
					return false;
				}
				usleep(2000);
			}
			map_bytes = (size_t)st.st_size;
			slot_bytes = (map_bytes - head) / (size_t)nranks;
		}
		base = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		::close(fd);
		if (base == MAP_FAILED) { base = nullptr; if (rank == 0) shm_unlink(name); *why = "mmap of the segment failed"; return false; }
		hdr = (ShmHeader *)base;
		mail = (ShmMail *)((uint8_t *)base + 4096);
		slots = (uint8_t *)base + head;
		if (rank == 0) {  // a fresh segment is zero-filled: counters start at 0
			hdr->nranks = (uint32_t)nranks;
			hdr->slot_bytes = slot_bytes;
			hdr->magic.store(SHM_MAGIC, std::memory_order_release);
		} else {
			const auto t0 = std::chrono::steady_clock::now();
			while (hdr->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
				if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(shm_timeout_s())) { *why = "rank 0 did not initialise the segment"; return false; }
				usleep(1000);
			}
			if (hdr->nranks != (uint32_t)nranks || hdr->slot_bytes != slot_bytes) { *why = "the segment belongs to another job (rank count differs)"; return false; }
		}
		memset(mail[rank].device, 0, sizeof mail[rank].device);
		if (hipDeviceGetPCIBusId(mail[rank].device, (int)sizeof mail[rank].device, device) != hipSuccess) {
			(void)hipGetLastError();
			snprintf(mail[rank].device, sizeof mail[rank].device, "dev%d", device);
		}
		const bool all = barrier();
		if (rank == 0) shm_unlink(name);  // every rank holds its mapping: the name can go, nothing is left behind on a crash
		if (!all) { *why = "a peer rank did not attach"; return false; }
		device_share = 0;
		for (int r = 0; r < nranks; ++r) device_share += strncmp(mail[r].device, mail[rank].device, sizeof mail[r].device) == 0;
		// pinned staging: copies run at PCIe rate; without it they still work
		registered = hipHostRegister(base, map_bytes, hipHostRegisterPortable) == hipSuccess;
		if (!registered) (void)hipGetLastError();
		return true;
	}
};

int attach(lfa_sim *s, lfa_dist *d, const int32_t *bounds) {
	const int lo = bounds[d->rank], hi = bounds[d->rank + 1];
	if (bounds[0] != 0 || bounds[d->nranks] != s->g.ntz || lo >= hi || lo < 0 || hi > s->g.ntz) {
		delete d;
		return lfa_fail(s, LFA_E_INVALID, "layer bounds must partition [0,%d) into non-empty slabs", s->g.ntz);
	}
	if (lfa_c_home_restore(s) < 0) {  // (the next binning takes C home again, into an array sized for the ids of the whole job)
		delete d;
		return LFA_E_HIP;
	}
	if (s->dist) delete s->dist;
	s->dist = d;
	// the multigrid levels whose tile layers do not straddle a slab face stay distributed (mg.hip)
	s->slab_align = 30;
	for (int r = 1; r < d->nranks; ++r) s->slab_align = std::min(s->slab_align, __builtin_ctz((unsigned)bounds[r]));
	s->slab_lo = lo;
	s->slab_hi = hi;
	s->binned = false;
	s->sources_valid = false;  // the seeding entries are the ones of this rank's own tile layers ...
	s->sources_built.clear();  // ... so entries flattened for another range (the whole domain, earlier bounds) are not "the same list"
	s->grid_valid = false;
	s->system_valid = false;
	if (!s->dist_red) LFA_HIP(s, hipMalloc(&s->dist_red, (LFA_DIST_ALPHA_OFF + 8) * 8));  // scalars | 2 gather buffers of 3 x nranks + 1 | alpha of the single-reduction CG (2)
	if (!s->halo_tiles) LFA_HIP(s, hipMalloc(&s->halo_tiles, (size_t)4 * s->g.ntx * s->g.nty * 4));
	return LFA_OK;
}
}  // namespace

extern "C" int lfa_dist_unique_id(void *id128) {
	if (!id128) return LFA_E_INVALID;
	if (!g_rccl.load()) return lfa_fail(nullptr, LFA_E_UNSUPPORTED, "librccl could not be loaded");
	ncclUniqueId id;
	if (g_rccl.GetUniqueId(&id) != ncclSuccess) return lfa_fail(nullptr, LFA_E_HIP, "ncclGetUniqueId failed");
	memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
	return LFA_OK;
}

extern "C" int lfa_dist_init_rccl(lfa_sim *s, int rank, int nranks, const void *id128, const int32_t *layer_bounds) {
	if (!s || !id128 || !layer_bounds || rank < 0 || rank >= nranks) return LFA_E_INVALID;
	if (!g_rccl.load()) return lfa_fail(s, LFA_E_UNSUPPORTED, "librccl could not be loaded");
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	RcclDist *d = new RcclDist();
	d->rank = rank;
	d->nranks = nranks;
	ncclUniqueId id;
	memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
	ncclResult_t r = g_rccl.CommInitRank(&d->comm, nranks, id, rank);
	if (r != ncclSuccess) {
		d->comm = nullptr;
		delete d;
		return lfa_fail(s, LFA_E_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
	}
	return attach(s, d, layer_bounds);
}

extern "C" int lfa_dist_init_shm(lfa_sim *s, const char *name, int rank, int nranks, const int32_t *layer_bounds) {
	if (!s || !name || name[0] != '/' || !layer_bounds || rank < 0 || rank >= nranks) return LFA_E_INVALID;
	LFA_HIP(s, hipSetDevice(s->device));
	LFA_TRY(lfa_corr_commit(s));
	ShmDist *d = new ShmDist();
	d->rank = rank;
	d->nranks = nranks;
	const char *why = "";
	if (!d->open(name, s->device, &why)) {
		delete d;
		return lfa_fail(s, LFA_E_HIP, "shared-memory transport %s: %s", name, why);
	}
	return attach(s, d, layer_bounds);
}

extern "C" lfa_hub *lfa_dist_local_hub_create(int nranks) {
	if (nranks < 1) return nullptr;
	lfa_hub *h = new lfa_hub();
	h->n = nranks;
	h->mail.resize(nranks);
	return h;
}
extern "C" void lfa_dist_local_hub_destroy(lfa_hub *h) { delete h; }

extern "C" int lfa_dist_init_local(lfa_sim *s, lfa_hub *h, int rank, const int32_t *layer_bounds) {
	if (!s || !h || !layer_bounds || rank < 0 || rank >= h->n) return LFA_E_INVALID;
	LFA_TRY(lfa_corr_commit(s));
	LocalDist *d = new LocalDist();
	d->rank = rank;
	d->nranks = h->n;
	d->device_share = h->n;  // (an upper bound: handles of one hub may sit on different devices)
	d->hub = h;
	return attach(s, d, layer_bounds);
}

extern "C" int lfa_dist_abandon(lfa_sim *s) {
	if (!s) return LFA_E_INVALID;
	if (s->dist) s->dist->broken = true;
	return LFA_OK;
}

extern "C" int lfa_dist_get_slab(const lfa_sim *s, int32_t *lo, int32_t *hi) {
	if (!s || !lo || !hi) return LFA_E_INVALID;
	*lo = s->dist ? s->slab_lo : 0;
	*hi = s->dist ? s->slab_hi : s->g.ntz;
	return LFA_OK;
}
