// Micro-benchmark: throughput of LDS atomic flavours on gfx950 (used to choose the P2G accumulation primitive).
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/lds_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

template <int MODE, int PATTERN>
__global__ void __launch_bounds__(256) k(const uint32_t *idx, float *out, int iters) {
	__shared__ float lds[8192];
	for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f;
	__syncthreads();
	uint32_t a = PATTERN == 0 ? threadIdx.x : idx[blockIdx.x * 256 + threadIdx.x];  // 0: conflict free, 1: random
	if (PATTERN == 2) a = threadIdx.x / 8;  // 8 lanes per address
	if (PATTERN == 3) {  // monotone in the lane with random gaps of 1-2 (k-th particles of consecutive occupied cells)
		uint32_t g = 1 + (idx[blockIdx.x * 256 + threadIdx.x] & 1), incl = g;
		for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(incl, o, 64); if ((threadIdx.x & 63) >= o) incl += t; }
		a = incl + 100 * (threadIdx.x >> 6);
	}
	if (PATTERN == 4) {  // halo index of the cells of an 8 x 8 slab of a tile, each shifted by a random (0|-1, 0|-10, 0|-100)
		const uint32_t r = idx[blockIdx.x * 256 + threadIdx.x], l = threadIdx.x & 63;
		a = 111 + (l & 7) + 10 * (l >> 3) + 100 * (threadIdx.x >> 6) - (r & 1) - 10 * ((r >> 1) & 1) - 100 * ((r >> 2) & 1);
	}
	if (PATTERN == 5 || PATTERN == 6) {  // lanes = the 64 cells of an 8 x 8 slab (x fastest), node = cell + random (0|1)^3, on two accumulator layouts
		const uint32_t r = idx[blockIdx.x * 256 + threadIdx.x], l = threadIdx.x & 63;
		const uint32_t hx = (l & 7) + (r & 1), hy = (l >> 3) + ((r >> 1) & 1), hz = (threadIdx.x >> 6) + ((r >> 2) & 1);
		if (PATTERN == 5) a = hx + 10 * hy + 100 * hz;  // row-major 10 x 10 x 10
		else a = hx >= 1 ? (hx - 1) + 8 * (hy + 10 * hz) : 800 + hy + 10 * hz;  // 8-wide interior rows, the plane hx = 0 behind them
	}
	if (PATTERN == 7) {  // consecutive particles of a cell-sorted tile: 8 lanes per cell, each its own random corner
		const uint32_t r = idx[blockIdx.x * 256 + threadIdx.x], c = threadIdx.x >> 3;
		a = (c & 7) + (r & 1) + 10 * (((c >> 3) & 3) + ((r >> 1) & 1)) + 100 * ((r >> 2) & 1);
	}
	float v = 1.0f + threadIdx.x;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int k = 0; k < 16; ++k) {
			uint32_t addr = (a + k * 257 + it) & 4095;
			if (MODE == 0) atomicAdd(&lds[addr], v);
			else if (MODE == 1) atomicAdd((uint32_t *)&lds[addr], (uint32_t)threadIdx.x);
			else if (MODE == 2) atomicAdd((unsigned long long *)&lds[(addr & 2047) * 2], (unsigned long long)threadIdx.x);
			else if (MODE == 3) atomicAdd((double *)&lds[(addr & 2047) * 2], (double)v);
			else if (MODE == 4) lds[addr] = v;
			else if (MODE == 5) { float o = lds[addr]; lds[addr] = o + v; }
		}
	}
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = lds[5] + lds[77];
}

template <int MODE, int PATTERN> void run(const char *name, const uint32_t *idx, float *out) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * 8, iters = 64;
	k<MODE, PATTERN><<<blocks, 256>>>(idx, out, 4);
	hipEventRecord(e0);
	k<MODE, PATTERN><<<blocks, 256>>>(idx, out, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms;
	hipEventElapsedTime(&ms, e0, e1);
	double ops = (double)blocks * 256 * iters * 16;
	printf("%-28s %8.3f ms  %7.2f lane-ops/clk/CU (2.4GHz, 256 CU)\n", name, ms, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
	std::vector<uint32_t> h(256 * 8 * 256);
	uint32_t s = 12345;
	for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) & 4095; }
	uint32_t *idx; float *out;
	hipMalloc(&idx, h.size() * 4); hipMalloc(&out, 4096 * 4);
	hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
	run<0, 0>("ds_add_f32 conflict-free", idx, out);
	run<0, 1>("ds_add_f32 random", idx, out);
	run<0, 2>("ds_add_f32 8 lanes/addr", idx, out);
	run<1, 0>("ds_add_u32 conflict-free", idx, out);
	run<1, 1>("ds_add_u32 random", idx, out);
	run<1, 2>("ds_add_u32 8 lanes/addr", idx, out);
	run<2, 0>("ds_add_u64 conflict-free", idx, out);
	run<2, 1>("ds_add_u64 random", idx, out);
	run<2, 2>("ds_add_u64 8 lanes/addr", idx, out);
	run<2, 3>("ds_add_u64 monotone+gaps", idx, out);
	run<2, 4>("ds_add_u64 slab cells +-1", idx, out);
	run<2, 5>("ds_add_u64 jds cells 10-row", idx, out);
	run<2, 6>("ds_add_u64 jds cells 8-row", idx, out);
	run<2, 7>("ds_add_u64 cell-sorted 8/cell", idx, out);
	run<1, 3>("ds_add_u32 monotone+gaps", idx, out);
	run<1, 4>("ds_add_u32 slab cells +-1", idx, out);
	run<3, 0>("ds_add_f64 conflict-free", idx, out);
	run<3, 1>("ds_add_f64 random", idx, out);
	run<4, 0>("ds_write_b32 conflict-free", idx, out);
	run<4, 1>("ds_write_b32 random", idx, out);
	run<5, 1>("read+write (non-atomic) random", idx, out);
	return 0;
}
