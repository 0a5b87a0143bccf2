// tools/sweep_probe.hip -- cycles per hyperplane level of the register sweeps (reg_tile_sweeps of pcg.hip), alone on a
// SIMD and with 4 / 8 waves per SIMD, for one tile or two interleaved tiles per wave. No memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ inline float dpp_move(float v) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ inline float lane_fetch(int byte_addr, float v) {
	return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
template <int NT> __device__ inline void sweeps(float (&A)[NT][8], float (&S)[NT][8], int lane) {
	const int lx = lane & 7, ly = lane >> 3, t0 = lx + ly;
	const float mxm = lx > 0, mym = ly > 0, mxp = lx < 7, myp = ly < 7;
	const int up8 = ((lane - 8) & 63) << 2, dn8 = ((lane + 8) & 63) << 2;
#pragma unroll
	for (int L = 0; L < 22; ++L) {
		const int k = L & 7, kp = (L + 7) & 7, zz = L - t0;
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			const float prev = A[t][kp];
			const float wx = dpp_move<0x111>(prev), wy = lane_fetch(up8, prev);
			const float sum = fmaf(mxm, wx, fmaf(mym, wy, zz > 0 ? prev : 0.f));
			const float w = fmaf(S[t][k] > 0.f ? S[t][k] : 0.f, sum, A[t][k]);
			A[t][k] = (unsigned)zz < 8u ? w : A[t][k];
		}
	}
#pragma unroll
	for (int L = 21; L >= 0; --L) {
		const int k = L & 7, kn = (L + 1) & 7, zz = L - t0;
		const bool act = (unsigned)zz < 8u;
#pragma unroll
		for (int t = 0; t < NT; ++t) {
			const float next = A[t][kn];
			const float zx = dpp_move<0x101>(next), zy = lane_fetch(dn8, next);
			const float sum = fmaf(mxp, zx, fmaf(myp, zy, zz < 7 ? next : 0.f));
			const float zv = fmaf(fabsf(S[t][k]), sum, A[t][k]);
			A[t][k] = act ? (S[t][k] > 0.f ? zv : 0.f) : A[t][k];
			S[t][k] = act ? zv : S[t][k];
		}
	}
}
template <int NT> __global__ void __launch_bounds__(256) k_probe(int reps, float *out, long long *ticks) {
	const int lane = threadIdx.x & 63;
	float A[NT][8], S[NT][8];
	for (int t = 0; t < NT; ++t) for (int k = 0; k < 8; ++k) { A[t][k] = 0.001f * (lane + k + t); S[t][k] = 0.01f; }
	const long long t0 = clock64();
	for (int r = 0; r < reps; ++r) sweeps<NT>(A, S, lane);
	const long long t1 = clock64();
	float acc = 0;
	for (int t = 0; t < NT; ++t) for (int k = 0; k < 8; ++k) acc += A[t][k] + S[t][k];
	out[blockIdx.x * 256 + threadIdx.x] = acc;
	if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NT> void run(int blocks_per_cu, float *out, long long *d) {
	const int reps = 200, grid = 256 * blocks_per_cu;
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL(k_probe<NT>, dim3(grid), dim3(256), 0, 0, reps, out, d);
	hipEventRecord(a);
	hipLaunchKernelGGL(k_probe<NT>, dim3(grid), dim3(256), 0, 0, reps, out, d);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
	printf("NT=%d, %d waves/SIMD: %.1f cycles per level per tile-sweep step (wave view), %.3f ms wall -> %.2f ns per level per wave, %.1f tile-sweeps/us chip\n",
	       NT, blocks_per_cu, (double)t / (reps * 44.0), ms, ms * 1e6 / (reps * 44.0), (double)grid * 4 * NT * reps / (ms * 1e3));
}
int main() {
	float *out; long long *d;
	hipMalloc(&out, 4 * 256 * 256 * 16); hipMalloc(&d, 64);
	for (int w : {1, 2, 4, 8}) { run<1>(w, out, d); run<2>(w, out, d); }
	return 0;
}
