// tools/stream_bench.hip -- practical HBM ceiling for the PCG access pattern on MI355X: NR read streams + NW write streams
// over a list of 2-KiB tiles, dense (tile k at k*2KiB) or gapped like the C4 dam break in a 512^3 tile-major field
// (16 of 64 tiles per x-row, 32 of 64 rows, 32 of 64 layers). One wave per tile, 16 B per lane, 4 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/stream_bench tools/stream_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int NR, int NW>
__global__ void __launch_bounds__(256) k_stream(const int *tiles, int n, float *const *rd, float *const *wr) {
	const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int slot = blockIdx.x * 4 + wid; slot < n; slot += gridDim.x * 4) {
		const size_t base = (size_t)tiles[slot] * 512 + 4 * lane;
		float4 acc = {0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 2; ++k) {
#pragma unroll
			for (int a = 0; a < NR; ++a) {
				const float4 v = *(const float4 *)(rd[a] + base + 256 * k);
				acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
			}
#pragma unroll
			for (int a = 0; a < NW; ++a) *(float4 *)(wr[a] + base + 256 * k) = acc;
		}
		if (NW == 0 && acc.x == 12345.f) rd[0][0] = acc.y;
	}
}

template <int NR, int NW> void run(const char *name, const int *dtiles, int n, float **drd, float **dwr, int grid) {
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stream<NR, NW>), dim3(grid), dim3(256), 0, 0, dtiles, n, drd, dwr);
	hipEventRecord(a);
	const int reps = 20;
	for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_stream<NR, NW>), dim3(grid), dim3(256), 0, 0, dtiles, n, drd, dwr);
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms = 0;
	hipEventElapsedTime(&ms, a, b);
	const double bytes = (double)n * 2048.0 * (NR + NW);
	printf("  %-8s %dR+%dW grid %4d: %7.1f us  %6.2f TB/s\n", name, NR, NW, grid, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) * 1e-12);
}

int main() {
	const int n = 16384;
	std::vector<int> dense(n), gapped;
	for (int i = 0; i < n; ++i) dense[i] = i;
	for (int tz = 0; tz < 32; ++tz) for (int ty = 0; ty < 32; ++ty) for (int tx = 0; tx < 16; ++tx) gapped.push_back(tx + 64 * (ty + 64 * tz));
	const size_t field = (size_t)64 * 64 * 64 * 512;  // floats of one 512^3 field
	float *arr[10], **drd, **dwr;
	for (int i = 0; i < 10; ++i) { hipMalloc(&arr[i], field * 4); hipMemset(arr[i], 0, field * 4); }
	hipMalloc(&drd, 8 * sizeof(float *)); hipMalloc(&dwr, 8 * sizeof(float *));
	hipMemcpy(drd, arr, 7 * sizeof(float *), hipMemcpyHostToDevice);
	hipMemcpy(dwr, arr + 7, 3 * sizeof(float *), hipMemcpyHostToDevice);
	int *dt[2];
	for (int k = 0; k < 2; ++k) { hipMalloc(&dt[k], n * 4); hipMemcpy(dt[k], k ? gapped.data() : dense.data(), n * 4, hipMemcpyHostToDevice); }
	const char *names[2] = {"dense", "gapped"};
	for (int k = 0; k < 2; ++k) {
		printf("%s tile list (%d tiles, %.1f MB per stream)\n", names[k], n, n * 2048e-6);
		for (int grid : {1024, 2048, 4096}) {
			run<1, 1>(names[k], dt[k], n, drd, dwr, grid);
			run<2, 1>(names[k], dt[k], n, drd, dwr, grid);
			run<4, 2>(names[k], dt[k], n, drd, dwr, grid);
			run<6, 3>(names[k], dt[k], n, drd, dwr, grid);
			run<6, 0>(names[k], dt[k], n, drd, dwr, grid);
		}
	}
	return 0;
}
