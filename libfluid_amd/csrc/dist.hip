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
// Every rank indexes the GLOBAL grid, so a ghost tile is simply a tile id whose data arrives by message; kernels are the
// single-GPU kernels, run over the owned tile lists.
//
// Transports: RCCL (ncclSend/ncclRecv grouped to z-1/z+1 + ncclAllReduce on the handle's stream; librccl is dlopen'ed so
// that single-GPU use has no dependency on it) and an in-process one (one host thread per handle, device-to-device copies
// at a rendezvous) used to test the protocol with several "virtual slabs" on one GPU.
#include <dlfcn.h>
#include <math.h>
#include <string.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <mutex>

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
	float *slab = stage_all + (size_t)(slot0 + k) * 6 * LFA_HALO_CELLS + 100 * hz;
	float *b = buf + (size_t)k * 600;
	for (int i = threadIdx.x; i < 600; i += 128) {
		float *g = slab + (i / 100) * LFA_HALO_CELLS + (i % 100);
		if (PACK) b[i] = *g;
		else *g = b[i];
	}
}

/// One z-slice (64 elements) of a tile-major vector for a run of particle tiles.
template <bool PACK>
__global__ void __launch_bounds__(64) k_slices_copy(const int *ptiles_all, int slot0, int n, int zz, uint32_t *vec,
                                                   int words_per_elem, uint32_t *buf) {
	const int k = blockIdx.x;
	if (k >= n) return;
	uint32_t *g = vec + ((size_t)ptiles_all[slot0 + k] * LFA_TILE_CELLS + (size_t)zz * 64) * words_per_elem;
	uint32_t *b = buf + (size_t)k * 64 * words_per_elem;
	for (int i = threadIdx.x; i < 64 * words_per_elem; i += 64) {
		if (PACK) b[i] = g[i];
		else g[i] = b[i];
	}
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
	int off[4] = {0, s->n_halo[0], s->n_halo[0] + s->n_halo[1], s->n_halo[0] + s->n_halo[1] + s->n_halo[2]};
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
	const int wpe = elem_bytes / 4;
	const size_t sb = (size_t)64 * elem_bytes;
	const int n_send[2] = {lfa_has_lo(s) ? s->n_own_first : 0, lfa_has_hi(s) ? s->n_own_last : 0};
	const int n_recv[2] = {s->n_ghost_lo, s->n_ghost_hi};
	const int send_slot0[2] = {s->p_off, s->p_off + s->n_ptiles - s->n_own_last};
	const int recv_slot0[2] = {0, s->p_off + s->n_ptiles};
	const int send_z[2] = {0, 7}, recv_z[2] = {7, 0};
	for (int w = 0; w < 2; ++w) {
		LFA_TRY(lfa_dist_ensure_xbuf(s, w, (size_t)n_send[w] * sb));
		LFA_TRY(lfa_dist_ensure_xbuf(s, 2 + w, (size_t)n_recv[w] * sb));
		if (n_send[w]) {
			hipLaunchKernelGGL(k_slices_copy<true>, dim3(n_send[w]), dim3(64), 0, s->stream, s->ptiles_all, send_slot0[w],
			                   n_send[w], send_z[w], (uint32_t *)vec, wpe, (uint32_t *)s->xbuf[w]);
			LFA_LAUNCH_CHECK(s);
		}
	}
	LFA_TRY(s->dist->exchange(s, s->xbuf[0], n_send[0] * sb, s->xbuf[2], n_recv[0] * sb, s->xbuf[1], n_send[1] * sb,
	                          s->xbuf[3], n_recv[1] * sb));
	for (int w = 0; w < 2; ++w)
		if (n_recv[w]) {
			hipLaunchKernelGGL(k_slices_copy<false>, dim3(n_recv[w]), dim3(64), 0, s->stream, s->ptiles_all, recv_slot0[w],
			                   n_recv[w], recv_z[w], (uint32_t *)vec, wpe, (uint32_t *)s->xbuf[2 + w]);
			LFA_LAUNCH_CHECK(s);
		}
	return LFA_OK;
}

/// Local fixed-order reduction of the per-workgroup partials, then the all-reduce over ranks; result in dist_red[slot].
int lfa_dist_allreduce(lfa_sim *s, const double *partials, int n, int slot, bool is_max) {
	hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(256), 0, s->stream, partials, n, s->dist_red + slot, is_max ? 1 : 0);
	LFA_LAUNCH_CHECK(s);
	return s->dist->allreduce(s, s->dist_red + slot, 1, is_max);
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
	int exchange(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
	             size_t n_send_hi, void *recv_hi, size_t n_recv_hi) override {
		LFA_HIP(s, hipStreamSynchronize(s->stream));
		lfa_hub::Mail &me = hub->mail[rank];
		me.lo = send_lo; me.n_lo = n_send_lo; me.hi = send_hi; me.n_hi = n_send_hi;
		if (!hub->barrier()) return lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		int rc = LFA_OK;
		if (rank > 0 && n_recv_lo) {
			const lfa_hub::Mail &nb = hub->mail[rank - 1];
			if (nb.n_hi != n_recv_lo) rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu",
			                                        rank - 1, nb.n_hi, n_recv_lo);
			else if (hipMemcpy(recv_lo, nb.hi, n_recv_lo, hipMemcpyDeviceToDevice) != hipSuccess) rc = LFA_E_HIP;
		}
		if (rank + 1 < nranks && n_recv_hi && rc == LFA_OK) {
			const lfa_hub::Mail &nb = hub->mail[rank + 1];
			if (nb.n_lo != n_recv_hi) rc = lfa_fail(s, LFA_E_INVALID, "slab exchange size mismatch with rank %d: %zu vs %zu",
			                                        rank + 1, nb.n_lo, n_recv_hi);
			else if (hipMemcpy(recv_hi, nb.lo, n_recv_hi, hipMemcpyDeviceToDevice) != hipSuccess) rc = LFA_E_HIP;
		}
		if (rc == LFA_OK && hipDeviceSynchronize() != hipSuccess) rc = LFA_E_HIP;  // D2D hipMemcpy may return early
		if (rc != LFA_OK) hub->fail();
		if (!hub->barrier() && rc == LFA_OK)  // nobody reuses a send buffer before every copy out of it is done
			rc = lfa_fail(s, LFA_E_HIP, "slab exchange: a peer rank failed or timed out");
		return rc;
	}
	int allreduce(lfa_sim *s, double *dev, int count, bool is_max) override {
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
};

// ================================================================================================= RCCL transport
struct RcclApi {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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
		LOAD(GroupStart); LOAD(GroupEnd); LOAD(GetErrorString);
#undef LOAD
		return GetUniqueId && CommInitRank && CommDestroy && Send && Recv && AllReduce && GroupStart && GroupEnd;
	}
};
RcclApi g_rccl;

struct RcclDist : lfa_dist {
	ncclComm_t comm = nullptr;
	~RcclDist() override {
		if (comm) g_rccl.CommDestroy(comm);
	}
#define NCCL_TRY(s, call)                                                                                \
	do {                                                                                                  \
		ncclResult_t r_ = (call);                                                                         \
		if (r_ != ncclSuccess)                                                                            \
			return lfa_fail((s), LFA_E_HIP, "%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
	} while (0)
	int exchange(lfa_sim *s, const void *send_lo, size_t n_send_lo, void *recv_lo, size_t n_recv_lo, const void *send_hi,
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
	int allreduce(lfa_sim *s, double *dev, int count, bool is_max) override {
		NCCL_TRY(s, g_rccl.AllReduce(dev, dev, (size_t)count, ncclFloat64, is_max ? ncclMax : ncclSum, comm, s->stream));
		return LFA_OK;
	}
};

int attach(lfa_sim *s, lfa_dist *d, const int32_t *bounds) {
	const int lo = bounds[d->rank], hi = bounds[d->rank + 1];
	if (bounds[0] != 0 || bounds[d->nranks] != s->g.ntz || lo >= hi || lo < 0 || hi > s->g.ntz) {
		delete d;
		return lfa_fail(s, LFA_E_INVALID, "layer bounds must partition [0,%d) into non-empty slabs", s->g.ntz);
	}
	if (s->dist) delete s->dist;
	s->dist = d;
	s->slab_lo = lo;
	s->slab_hi = hi;
	s->binned = false;
	s->grid_valid = false;
	s->system_valid = false;
	if (!s->dist_red) LFA_HIP(s, hipMalloc(&s->dist_red, 64 * 8));
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
	LocalDist *d = new LocalDist();
	d->rank = rank;
	d->nranks = h->n;
	d->hub = h;
	return attach(s, d, layer_bounds);
}

extern "C" int lfa_dist_get_slab(const lfa_sim *s, int32_t *lo, int32_t *hi) {
	if (!s || !lo || !hi) return LFA_E_INVALID;
	*lo = s->dist ? s->slab_lo : 0;
	*hi = s->dist ? s->slab_hi : s->g.ntz;
	return LFA_OK;
}
