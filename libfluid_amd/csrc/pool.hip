// libfluid_amd/csrc/pool.hip -- process-wide caches that make handle re-creation cheap.
//
// The Maya node builds a fresh fluid::simulation per evaluation (plugins/maya/nodes/grid_node.cpp:256-274,350-366) and the
// testbed resets its scene by resize() (testbed/main.cpp:125-178): a handle is created and destroyed over and over with the same
// grid size. lfa_create used to pay ~20 hipMallocs, 3 streams, 4 events, a pinned page and - later, lazily - another few dozen
// allocations every time. Now:
//  * every hipMalloc / hipFree of the library goes through lfa_pool_malloc / lfa_pool_free (common.h redirects the names): a
//    freed block is kept in a per-device free list keyed by its exact size and handed out again to the next request of that
//    size. A second handle of the same grid size therefore allocates nothing. Blocks come back with their old contents - every
//    consumer in the library initialises what it reads (tested by the whole GPU suite, which re-uses blocks constantly).
//  * the streams, events and the pinned page of a destroyed handle are parked as a set and adopted by the next lfa_create.
// hipFree synchronises the device; code that relies on that (re-allocation paths) still gets it. lfa_destroy synchronises its
// own streams once and releases without further synchronisation.
// The cache is bounded (LFA_POOL_MAX_BYTES, default 1/4 of the device memory): beyond it blocks are really freed.
// lfa_pool_trim() releases everything (hosts call it at shutdown or under memory pressure).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/libfluid_amd.h"

namespace {
struct Block {
	size_t bytes;
	int device;
};
struct Pool {
	std::mutex m;
	std::unordered_map<void *, Block> live;                       // blocks handed out
	std::map<std::pair<int, size_t>, std::vector<void *>> free_;  // (device, bytes) -> cached blocks
	size_t cached_bytes = 0, cached_blocks = 0, hits = 0, misses = 0;
};
Pool &pool() {
	static Pool *p = new Pool();  // never destroyed: handles may be released after static destructors ran
	return *p;
}
thread_local int t_nosync = 0;
/// The cache's budget PER DEVICE (a quarter of that device's memory unless LFA_POOL_MAX_BYTES says otherwise; read once per
/// device, with that device current), and what each device holds.
struct DevBudget {
	size_t max_bytes = 0, cached_bytes = 0;
};
DevBudget g_budget[64];
size_t max_bytes_for(int device) {  // (P.m held)
	DevBudget &b = g_budget[device & 63];
	if (b.max_bytes) return b.max_bytes;
	if (const char *e = getenv("LFA_POOL_MAX_BYTES")) {
		b.max_bytes = (size_t)strtoull(e, nullptr, 10);
		if (!b.max_bytes) b.max_bytes = 1;  // "0" = keep nothing
		return b.max_bytes;
	}
	int cur = 0;
	(void)hipGetDevice(&cur);
	if (cur != device) (void)hipSetDevice(device);
	size_t fr = 0, tot = 0;
	b.max_bytes = (hipMemGetInfo(&fr, &tot) == hipSuccess && tot) ? tot / 4 : ((size_t)8 << 30);
	if (cur != device) (void)hipSetDevice(cur);
	return b.max_bytes;
}
}  // namespace

hipError_t lfa_pool_malloc(void **p, size_t bytes) {
	if (!p) return hipErrorInvalidValue;
	*p = nullptr;
	if (bytes == 0) bytes = 1;
	int dev = 0;
	(void)hipGetDevice(&dev);
	Pool &P = pool();
	{
		std::lock_guard<std::mutex> lk(P.m);
		auto it = P.free_.find({dev, bytes});
		if (it != P.free_.end() && !it->second.empty()) {
			*p = it->second.back();
			it->second.pop_back();
			P.cached_bytes -= bytes;
			g_budget[dev & 63].cached_bytes -= bytes;
			--P.cached_blocks;
			++P.hits;
			P.live[*p] = Block{bytes, dev};
			return hipSuccess;
		}
		++P.misses;
	}
	hipError_t e = (hipMalloc)(p, bytes);
	if (e == hipErrorOutOfMemory) {  // give the cache back to the driver and try once more
		(void)hipGetLastError();
		lfa_pool_trim();
		e = (hipMalloc)(p, bytes);
	}
	if (e != hipSuccess) return e;
	std::lock_guard<std::mutex> lk(P.m);
	P.live[*p] = Block{bytes, dev};
	return hipSuccess;
}

hipError_t lfa_pool_free(void *p) {
	if (!p) return hipSuccess;
	Pool &P = pool();
	Block b{0, 0};
	bool known = false;
	{
		std::lock_guard<std::mutex> lk(P.m);
		auto it = P.live.find(p);
		if (it != P.live.end()) {
			b = it->second;
			P.live.erase(it);
			known = true;
		}
	}
	if (!known) return (hipFree)(p);
	// hipFree waits for the device: a block may still be in use by work queued on some stream - of ITS device, which need not
	// be the current one (a host thread that drives several GPUs)
	hipError_t e = hipSuccess;
	int cur = 0;
	(void)hipGetDevice(&cur);
	if (cur != b.device) (void)hipSetDevice(b.device);
	if (!t_nosync) e = hipDeviceSynchronize();
	bool keep = false;
	{
		std::lock_guard<std::mutex> lk(P.m);
		// kept: what fits the device's budget; a block larger than a quarter of the budget is one-shot scratch (the stream
		// benchmark's two 1 GiB buffers, an outgrown particle capacity) and goes back to the driver
		const size_t budget = max_bytes_for(b.device);
		DevBudget &db = g_budget[b.device & 63];
		if (b.bytes <= budget / 4 && db.cached_bytes + b.bytes <= budget) {
			P.free_[{b.device, b.bytes}].push_back(p);
			P.cached_bytes += b.bytes;
			db.cached_bytes += b.bytes;
			++P.cached_blocks;
			keep = true;
		}
	}
	if (!keep) {  // (outside the lock: the driver's free synchronises, other threads' allocations must not queue behind it)
		const hipError_t ef = (hipFree)(p);
		if (e == hipSuccess) e = ef;
	}
	if (cur != b.device) (void)hipSetDevice(cur);
	return e;
}

void lfa_pool_nosync_begin() { ++t_nosync; }
void lfa_pool_nosync_end() { --t_nosync; }

// ---- streams / events / pinned page of a handle
#include "stream_set.h"
namespace {
std::mutex g_sets_m;
std::vector<lfa_stream_set *> g_sets;
void destroy_set(lfa_stream_set *q) {
	if (!q) return;
	if (q->ev_created)
		for (auto &e : q->ev) (void)hipEventDestroy(e);
	hipEvent_t evs[] = {q->ev_fork, q->ev_join, q->ev_cfork, q->ev_cjoin};
	for (hipEvent_t e : evs)
		if (e) (void)hipEventDestroy(e);
	hipStream_t sts[] = {q->stream3, q->stream2, q->stream};
	for (hipStream_t st : sts)
		if (st) (void)hipStreamDestroy(st);
	if (q->h_pinned) (void)hipHostFree(q->h_pinned);
	delete q;
}
}  // namespace

lfa_stream_set *lfa_pool_take_set(int device) {
	std::lock_guard<std::mutex> lk(g_sets_m);
	for (size_t i = 0; i < g_sets.size(); ++i)
		if (g_sets[i]->device == device) {
			lfa_stream_set *q = g_sets[i];
			g_sets.erase(g_sets.begin() + (long)i);
			return q;
		}
	return nullptr;
}
void lfa_pool_park_set(lfa_stream_set *q) {
	if (!q) return;
	std::lock_guard<std::mutex> lk(g_sets_m);
	if (g_sets.size() >= 16) {
		destroy_set(q);
		return;
	}
	g_sets.push_back(q);
}

extern "C" void lfa_pool_trim(void) {
	Pool &P = pool();
	std::vector<void *> blocks;
	{
		std::lock_guard<std::mutex> lk(P.m);
		for (auto &kv : P.free_)
			for (void *p : kv.second) blocks.push_back(p);
		P.free_.clear();
		P.cached_bytes = 0;
		P.cached_blocks = 0;
		for (DevBudget &b : g_budget) b.cached_bytes = 0;
	}
	for (void *p : blocks) (void)(hipFree)(p);
	std::vector<lfa_stream_set *> sets;
	{
		std::lock_guard<std::mutex> lk(g_sets_m);
		sets.swap(g_sets);
	}
	for (lfa_stream_set *q : sets) destroy_set(q);
}

extern "C" void lfa_pool_stats(uint64_t stats[4]) {
	if (!stats) return;
	Pool &P = pool();
	std::lock_guard<std::mutex> lk(P.m);
	stats[0] = P.cached_bytes;
	stats[1] = P.cached_blocks;
	stats[2] = P.hits;
	stats[3] = P.misses;
}
