// tools/clock_probe.hip -- what clock64() (s_memtime) counts on this box, and the shader clock of an idle chip:
// a spin of N ticks timed with HIP events, and a dependent chain of 1e6 v_add timed in ticks and in wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_spin(long long ticks, long long *out) {
	const long long t0 = clock64();
	long long n = 0;
	while (clock64() - t0 < ticks) ++n;
	if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = n;
}
__global__ void k_chain(int n, float *out, long long *ticks) {
	float a = threadIdx.x;
	const long long t0 = clock64();
#pragma unroll 1
	for (int i = 0; i < n; ++i) {
		asm volatile("v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n"
		             "v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n v_add_f32 %0, %0, 1.0\n" : "+v"(a));
	}
	const long long t1 = clock64();
	out[threadIdx.x] = a;
	if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
	long long *d; float *f;
	hipMalloc(&d, 64); hipMalloc(&f, 1024);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	for (int grid : {1, 1024}) {
		hipEventRecord(a);
		hipLaunchKernelGGL(k_spin, dim3(grid), dim3(64), 0, 0, 1000000LL, d);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		printf("spin 1e6 ticks, grid %d: %.3f ms -> %.1f ticks/us\n", grid, ms, 1e6 / (ms * 1e3));
	}
	for (int grid : {1, 1024}) {
		hipEventRecord(a);
		hipLaunchKernelGGL(k_chain, dim3(grid), dim3(64), 0, 0, 125000, f, d);
		hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
		printf("1e6 dependent v_add_f32, grid %d: %.3f ms wall, %lld ticks -> %.2f ticks per add, %.2f ns per add\n", grid, ms, t, t / 1e6, ms * 1e6 / 1e6);
	}
	return 0;
}
