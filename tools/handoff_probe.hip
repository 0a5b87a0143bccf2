// tools/handoff_probe.hip -- what does one producer -> consumer hand-off between two workgroups cost on gfx950 (different XCDs as
// a rule), in the two forms k_mg_coarse could use?
//   mode 0 (today): the producer stores 512 floats with agent-scope relaxed atomics, waits for the acknowledgement (s_waitcnt
//                   vmcnt(0)), barrier, thread 0 stores a flag; the consumer polls the flag (a few threads), barrier, then every
//                   thread loads its two values with agent-scope loads.
//   mode 1: every value travels as an 8-byte word {value, tag}; every consumer thread polls its own two words until the tag is
//           this round's. No acknowledgement wait, no flag, no second round trip.
// Two workgroups play ping-pong for `rounds` rounds (a round = two hand-offs); W pairs run side by side.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/handoff_probe tools/handoff_probe.hip ; run: tools/handoff_probe [pairs] [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline void st32(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline float ld32(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned long long ld64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ void __launch_bounds__(256) k_pingpong(float *data, unsigned long long *data8, unsigned *flags, int rounds, float *out,
                                                   unsigned long long *ticks) {
	const int pair = blockIdx.x >> 1, me = blockIdx.x & 1, t = threadIdx.x;
	float *mine = data + (size_t)(2 * pair + me) * 512, *other = data + (size_t)(2 * pair + (me ^ 1)) * 512;
	unsigned long long *mine8 = data8 + (size_t)(2 * pair + me) * 512, *other8 = data8 + (size_t)(2 * pair + (me ^ 1)) * 512;
	unsigned *fmine = flags + 2 * pair + me, *fother = flags + 2 * pair + (me ^ 1);
	float acc = (float)t;
	const unsigned long long t0 = wall_clock64();
	for (int r = 1; r <= rounds; ++r) {
		for (int turn = 0; turn < 2; ++turn) {
			if (turn == me) {  // produce
				if (MODE == 0) {
					st32(mine + t, acc);
					st32(mine + t + 256, acc + 1.f);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					__syncthreads();
					if (t == 0) __hip_atomic_store(fmine, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				} else {
					st64(mine8 + t, ((unsigned long long)r << 32) | __float_as_uint(acc));
					st64(mine8 + t + 256, ((unsigned long long)r << 32) | __float_as_uint(acc + 1.f));
				}
			} else {  // consume
				if (MODE == 0) {
					if (t < 8) while (__hip_atomic_load(fother, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)r) __builtin_amdgcn_s_sleep(1);
					__syncthreads();
					acc = ld32(other + t) * 0.5f + ld32(other + ((t + 37) & 255) + 256) * 0.25f;
				} else {
					unsigned long long a, b;
					while (((a = ld64(other8 + t)) >> 32) != (unsigned long long)r) __builtin_amdgcn_s_sleep(1);
					while (((b = ld64(other8 + ((t + 37) & 255) + 256)) >> 32) != (unsigned long long)r) __builtin_amdgcn_s_sleep(1);
					acc = __uint_as_float((unsigned)a) * 0.5f + __uint_as_float((unsigned)b) * 0.25f;
					__syncthreads();
				}
			}
		}
	}
	if (t == 0) ticks[blockIdx.x] = wall_clock64() - t0;
	out[blockIdx.x * 256 + t] = acc;
}

int main(int argc, char **argv) {
	const int pairs = argc > 1 ? atoi(argv[1]) : 64, rounds = argc > 2 ? atoi(argv[2]) : 2000;
	float *data, *out;
	unsigned long long *data8, *ticks;
	unsigned *flags;
	CHECK(hipMalloc(&data, (size_t)pairs * 2 * 512 * 4));
	CHECK(hipMalloc(&data8, (size_t)pairs * 2 * 512 * 8));
	CHECK(hipMalloc(&flags, pairs * 2 * 4));
	CHECK(hipMalloc(&out, (size_t)pairs * 2 * 256 * 4));
	CHECK(hipMalloc(&ticks, pairs * 2 * 8));
	for (int mode = 0; mode < 2; ++mode) {
		CHECK(hipMemset(data, 0, (size_t)pairs * 2 * 512 * 4));
		CHECK(hipMemset(data8, 0, (size_t)pairs * 2 * 512 * 8));
		CHECK(hipMemset(flags, 0, pairs * 2 * 4));
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		CHECK(hipEventRecord(e0));
		if (mode == 0) hipLaunchKernelGGL(k_pingpong<0>, dim3(2 * pairs), dim3(256), 0, 0, data, data8, flags, rounds, out, ticks);
		else hipLaunchKernelGGL(k_pingpong<1>, dim3(2 * pairs), dim3(256), 0, 0, data, data8, flags, rounds, out, ticks);
		CHECK(hipEventRecord(e1));
		CHECK(hipDeviceSynchronize());
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("mode %d (%s): %d pairs, %d rounds: %.3f us per hand-off (kernel %.3f ms)\n", mode,
		       mode == 0 ? "data + ack wait + flag + poll + load" : "tagged 8-byte words, every thread polls its own", pairs, rounds,
		       1e3 * ms / (2.0 * rounds), ms);
	}
	return 0;
}
