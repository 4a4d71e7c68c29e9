// tools/l2_probe_rate.hip -- how many random one-dword probes per second a bitmap of a given size takes from all CUs at once
// (the second-level bitmap of the LAP_ANY scans for >= 4 errors: one probe per survivor that passes the LDS set).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/l2_probe_rate.hip -o /tmp/l2_probe_rate && /tmp/l2_probe_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every thread: `iters` rounds of INFLIGHT independent probes; ACTIVE_PCT of the lanes take part (the others idle, as lanes
// without a first-level candidate do)
template <int INFLIGHT>
__global__ __launch_bounds__(768) void probe(const uint32_t *table, uint32_t mask_words, int iters, uint32_t active_of_256, uint32_t *out)
{
	uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
	for (int i = 0; i < iters; i++) {
		uint32_t v[INFLIGHT];
#pragma unroll
		for (int k = 0; k < INFLIGHT; k++) {
			x = x * 1664525u + 1013904223u;
			const uint32_t r = x >> 8;
			v[k] = 0;
			if (((r >> 16) & 255) < active_of_256)
				v[k] = table[r & mask_words];
		}
#pragma unroll
		for (int k = 0; k < INFLIGHT; k++)
			acc += v[k];
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
	int dev_cus = 0;
	CHECK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
	const int grid = dev_cus * 2, threads = 768, iters = 2000;
	uint32_t *table, *out;
	CHECK(hipMalloc(&table, 64u << 20));
	CHECK(hipMemset(table, 0x55, 64u << 20));
	CHECK(hipMalloc(&out, (size_t)grid * threads * 4));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (uint32_t kib : {64u, 512u, 1024u, 2048u, 4096u, 8192u, 32768u})
		for (uint32_t act : {256u, 144u, 64u}) {
			const uint32_t mask = kib * 256 - 1;
			float best = 1e9f;
			for (int rep = 0; rep < 3; rep++) {
				CHECK(hipEventRecord(e0));
				hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(threads), 0, 0, table, mask, iters, act, out);
				CHECK(hipEventRecord(e1));
				CHECK(hipEventSynchronize(e1));
				float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			const double probes = (double)grid * threads * iters * 4 * act / 256.0;
			printf("table %6u KiB  lanes active %3u/256  %.3f ms  %.1f G probes/s\n", kib, act, best, probes / best / 1e6);
		}
	return 0;
}
