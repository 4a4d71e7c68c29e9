// tools/lds_rate.hip -- micro-benchmark: how random LDS reads and integer VALU work overlap on
// gfx950 at the scan kernel's geometry (1024-thread workgroups, one per CU, tables in LDS).
//   mode 0: VALU only (V ops per step)      mode 1: LDS only (R random ds_read_b32 per step)
//   mode 2: both, reads independent          mode 3: both, second read depends on the first
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_rate.hip -o tools/lds_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define STEPS 4096
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

template <int MODE, int VOPS, int CHAINS>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t seed, uint32_t active_pct)
{
	extern __shared__ uint32_t lds[];
	for (uint32_t i = threadIdx.x; i < 28672; i += blockDim.x)
		lds[i] = i * 2654435761u;
	__syncthreads();
	uint32_t x[CHAINS], acc = 0;
	for (int c = 0; c < CHAINS; c++)
		x[c] = (threadIdx.x * 747796405u + seed + c * 2891336453u);
	const bool live = ((threadIdx.x * 2654435761u) >> 25) % 100 < active_pct;   // fraction of lanes that read
	for (int s = 0; s < STEPS; s++) {
		uint32_t t1[CHAINS], t2[CHAINS], t3[CHAINS];
#pragma unroll
		for (int c = 0; c < CHAINS; c++) {
			x[c] = x[c] * 1664525u + 1013904223u;
			t1[c] = t2[c] = t3[c] = 0;
			if (MODE >= 1 && live) {
				t1[c] = *reinterpret_cast<lds_u32_t *>((x[c] >> 8) & 0x7ffc);            // 32 KiB table
				t2[c] = *reinterpret_cast<lds_u32_t *>(32768 + ((x[c] >> 20) & 0x3ffc));  // 16 KiB table
			}
		}
#pragma unroll
		for (int c = 0; c < CHAINS; c++) {
			uint32_t v = x[c] ^ t1[c] ^ t2[c];
			if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
				for (int j = 0; j < VOPS; j++)
					v = (v >> 3) ^ (v << 5) ^ j;          // 3 full-rate ops per j
			}
			if (MODE == 3 && live)
				t3[c] = *reinterpret_cast<lds_u32_t *>(49152 + ((v >> 3) & 0xfffc));      // 64 KiB bitmap
			acc += v ^ t3[c];
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int VOPS, int CHAINS>
static void run(const char *name, uint32_t *d_out, uint32_t pct)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, VOPS, CHAINS>), hipFuncAttributeMaxDynamicSharedMemorySize, 114688);
	hipLaunchKernelGGL((k<MODE, VOPS, CHAINS>), dim3(256), dim3(1024), 114688, 0, d_out, 1u, pct);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k<MODE, VOPS, CHAINS>), dim3(256), dim3(1024), 114688, 0, d_out, 2u, pct);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	double steps = (double)STEPS * CHAINS * 16;                 // chain-steps per CU (16 waves)
	printf("%-28s chains=%d vops=%2d active=%3u%%  %7.3f ms  %6.1f CU-cycles per wave chain-step\n",
	       name, CHAINS, VOPS * 3, pct, ms, ms * 1e-3 * 2.4e9 / steps);
}

int main()
{
	uint32_t *d_out;
	(void)hipMalloc(&d_out, 256 * 1024 * 4);
	run<0, 6, 2>("VALU only", d_out, 100);
	run<1, 0, 2>("LDS 2 reads only", d_out, 100);
	run<1, 0, 2>("LDS 2 reads only", d_out, 50);
	run<1, 0, 2>("LDS 2 reads only", d_out, 25);
	run<3, 0, 2>("LDS 2+1 dependent reads", d_out, 100);
	run<3, 0, 2>("LDS 2+1 dependent reads", d_out, 50);
	run<2, 6, 2>("VALU + 2 reads", d_out, 100);
	run<3, 6, 2>("VALU + 2+1 reads", d_out, 100);
	run<3, 6, 2>("VALU + 2+1 reads", d_out, 50);
	run<3, 6, 1>("VALU + 2+1 reads", d_out, 50);
	run<3, 6, 4>("VALU + 2+1 reads", d_out, 50);
	run<3, 12, 2>("VALU + 2+1 reads", d_out, 50);
	run<0, 12, 2>("VALU only", d_out, 100);
	return 0;
}
