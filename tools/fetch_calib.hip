// tools/fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on a known byte count for the load
// shapes the scan kernels use (MI355X_MICROARCH.md: "calibrate on a known byte count in your own
// access pattern").  Each kernel reads the same 4 GiB buffer exactly once:
//   read_b64    one global_load_dwordx2 per lane, consecutive lanes = consecutive words
//   read_b128   one global_load_dwordx4 per lane
//   read_b64_tiles  the scan kernel's shape: lane loads word[i] and word[i+1] of 1024-word tiles
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void read_b64(const uint64_t *p, uint64_t n, uint64_t *sink)
{
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024)
		acc ^= p[i];
	if (acc == 0x1234567)
		*sink = acc;
}

__global__ __launch_bounds__(1024) void read_b128(const uint4 *p, uint64_t n, uint64_t *sink)
{
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 1024) {
		uint4 v = p[i];
		acc ^= v.x ^ v.y ^ v.z ^ v.w;
	}
	if (acc == 0x1234567)
		*sink = acc;
}

__global__ __launch_bounds__(1024) void read_b64_tiles(const uint64_t *p, uint64_t n, uint64_t *sink)
{
	// exactly scan_lap_any_kernel's loads: a workgroup owns tiles of 1024 words; every lane
	// loads its word and the next one (the 63-symbol halo), two tiles in flight
	uint64_t acc = 0;
	const uint64_t tiles = n / 1024 - 1;
	for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
		const uint64_t *tp = p + t * 1024;
		acc ^= tp[threadIdx.x];
		acc ^= tp[threadIdx.x + 1] >> 1;
	}
	if (acc == 0x1234567)
		*sink = acc;
}

int main()
{
	const uint64_t bytes = 4ull << 30, n = bytes / 8;
	uint64_t *d, *sink;
	if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess)
		return 1;
	(void)hipMemset(d, 0x5a, bytes);
	(void)hipDeviceSynchronize();
	for (int rep = 0; rep < 3; rep++) {
		hipLaunchKernelGGL(read_b64, dim3(256), dim3(1024), 0, 0, d, n, sink);
		hipLaunchKernelGGL(read_b128, dim3(256), dim3(1024), 0, 0, (const uint4 *)d, n / 2, sink);
		hipLaunchKernelGGL(read_b64_tiles, dim3(256), dim3(1024), 0, 0, d, n, sink);
	}
	(void)hipDeviceSynchronize();
	printf("read %llu bytes per kernel\n", (unsigned long long)bytes);
	return 0;
}
