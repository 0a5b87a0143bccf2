// tools/xcd_barrier_probe.hip -- what does a grid barrier cost on gfx950 when the data the workgroups exchange never sits in a
// cache that needs a bulk write-back / invalidate? Design input for the single-launch coarse levels of the V-cycle (mg.hip).
//   mode 0: exchanged words written / read with agent-scope relaxed atomics (sc1), counter barrier, workgroups on every XCD
//   mode 1: workgroup-scope relaxed atomics (sc0: L1 bypassed, the XCD's L2 is the meeting point), workgroups of ONE XCD only
//           (8 W workgroups are launched, those with blockIdx % 8 != 0 leave at once; HW_REG_XCC_ID is recorded to check)
//   mode 2: plain loads / stores with agent-scope release / acquire fences by every thread (buffer_wbl2 / buffer_inv)
//   mode 3: as 2, one fencing thread per workgroup
// Each round: every thread stores a token, barrier, reads the tokens of two other workgroups and checks them.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_barrier_probe tools/xcd_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x)                                                                      \
	do {                                                                              \
		hipError_t e = (x);                                                           \
		if (e != hipSuccess) {                                                        \
			printf("%s: %s (%d)\n", #x, hipGetErrorString(e), __LINE__);              \
			exit(1);                                                                  \
		}                                                                             \
	} while (0)

template <int MODE> __device__ inline void put(unsigned *p, unsigned v) {
	if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	else *(volatile unsigned *)p = v;
}
template <int MODE> __device__ inline unsigned get(unsigned *p) {
	if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	return *(volatile unsigned *)p;
}

template <int MODE> __device__ inline void grid_barrier(unsigned *counter, unsigned target) {
	if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
	// this thread's stores have been acknowledged (the compiler does NOT put a vmcnt wait in front of s_barrier by itself, and a
	// workgroup-scope release fence emits none either: checked in the ISA)
	else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
		if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	__syncthreads();
	if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(unsigned *buf, unsigned *counter, int W, int rounds, int words, unsigned *errors,
                                                unsigned long long *cycles, unsigned *xcc) {
	int wg = blockIdx.x;
	if (MODE == 1) {
		if (blockIdx.x & 7) return;
		wg = blockIdx.x >> 3;
	}
	unsigned x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	if (threadIdx.x == 0) xcc[wg] = x;
	unsigned err = 0;
	const unsigned long long t0 = wall_clock64();
	for (int r = 0; r < rounds; ++r) {
		unsigned *b = buf + (size_t)(r & 1) * W * words;
		for (int i = threadIdx.x; i < words; i += 256) put<MODE>(b + (size_t)wg * words + i, (unsigned)(r * 4096 + wg) * 1024u + (unsigned)i);
		grid_barrier<MODE>(counter, (unsigned)(r + 1) * (unsigned)W);
		const int o1 = (wg + 1) % W, o2 = (wg + W / 2 + 3) % W;
		for (int i = threadIdx.x; i < words; i += 256) {
			err += get<MODE>(b + (size_t)o1 * words + i) != (unsigned)(r * 4096 + o1) * 1024u + (unsigned)i;
			err += get<MODE>(b + (size_t)o2 * words + i) != (unsigned)(r * 4096 + o2) * 1024u + (unsigned)i;
		}
	}
	const unsigned long long t1 = wall_clock64();
	if (err) atomicAdd(errors, err);
	if (threadIdx.x == 0 && wg == 0) cycles[0] = t1 - t0;
}

template <int MODE> static void run(int W, int rounds, int words) {
	unsigned *buf, *counter, *errors, *xcc;
	unsigned long long *cycles;
	CHECK(hipMalloc(&buf, (size_t)2 * W * words * 4));
	CHECK(hipMalloc(&counter, 4));
	CHECK(hipMalloc(&errors, 4));
	CHECK(hipMalloc(&xcc, (size_t)W * 4));
	CHECK(hipMalloc(&cycles, 8));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	unsigned herr = 0;
	unsigned long long hc = 0;
	std::vector<unsigned> hx(W);
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipMemset(buf, 0xff, (size_t)2 * W * words * 4));
		CHECK(hipMemset(counter, 0, 4));
		CHECK(hipMemset(errors, 0, 4));
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_probe<MODE>, dim3(MODE == 1 ? 8 * W : W), dim3(256), 0, 0, buf, counter, W, rounds, words, errors, cycles, xcc);
		CHECK(hipEventRecord(e1));
		CHECK(hipDeviceSynchronize());
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		best = ms < best ? ms : best;
		unsigned e;
		CHECK(hipMemcpy(&e, errors, 4, hipMemcpyDeviceToHost));
		herr += e;
		CHECK(hipMemcpy(&hc, cycles, 8, hipMemcpyDeviceToHost));
	}
	CHECK(hipMemcpy(hx.data(), xcc, (size_t)W * 4, hipMemcpyDeviceToHost));
	unsigned mask = 0;
	for (int i = 0; i < W; ++i) mask |= 1u << (hx[i] & 15);
	printf("mode %d  W %4d  words %5d  rounds %d : %.3f us per round (event), %.3f us (100 MHz clock)  errors %u  xcc mask 0x%x\n", MODE, W, words,
	       rounds, best * 1000.0 / rounds, hc / 100.0 / rounds, herr, mask);
	CHECK(hipFree(buf));
	CHECK(hipFree(counter));
	CHECK(hipFree(errors));
	CHECK(hipFree(xcc));
	CHECK(hipFree(cycles));
}

__global__ void k_xcc(unsigned *out) {
	unsigned x;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
	if (threadIdx.x == 0) out[blockIdx.x] = x;
}

int main() {
	unsigned *d;
	CHECK(hipMalloc(&d, 64 * 4));
	hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, d);
	unsigned h[64];
	CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
	printf("XCC_ID of workgroups 0..63:");
	for (int i = 0; i < 64; ++i) printf(" %u", h[i] & 15);
	printf("\n");
	const int rounds = 200;
	for (int words : {512, 4096}) {
		for (int W : {8, 32, 64, 128, 256}) run<0>(W, rounds, words);
		for (int W : {8, 32, 64, 128}) run<1>(W, rounds, words);
		for (int W : {8, 32, 64, 128, 256}) run<3>(W, rounds, words);
		for (int W : {32, 128}) run<2>(W, rounds, words);
	}
	return 0;
}
