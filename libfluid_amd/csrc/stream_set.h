// libfluid_amd/csrc/stream_set.h -- what a destroyed handle parks for the next lfa_create on the same device (pool.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct lfa_stream_set {
	int device = 0;
	hipStream_t stream = nullptr, stream2 = nullptr, stream3 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_cfork = nullptr, ev_cjoin = nullptr;
	hipEvent_t ev[48];
	bool ev_created = false;
	uint32_t *h_pinned = nullptr;
};
