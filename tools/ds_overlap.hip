// tools/ds_overlap.hip -- micro-benchmark (gfx950): what one DS wave-instruction of each kind
// costs (a) on its own and (b) on top of a VALU-saturated loop, at the scan kernel's geometry
// (one 1024-thread workgroup per CU, all 160 KiB of LDS claimed).  The scan kernel's survivor
// loop is a mix of integer VALU work and random table reads; this tells which DS forms hide
// behind VALU work and which do not.
//   Build: hipcc --offload-arch=gfx950 -O3 tools/ds_overlap.hip -o gpurun_out/ds_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 2048
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint64_t lds_u64_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;

enum { DS_NONE = 0, DS_ADDRONLY, DS_B32_RANDOM, DS_B32_LINEAR, DS_BPERMUTE, DS_B64_RANDOM, DS_U16_RANDOM, DS_B32_SAMEADDR,
       DS_B32_MASKED40, DS_B32_PARKED40, DS_B32_SMALLTAB };

static const char *ds_name[] = { "none", "address VALU only", "read_b32 random 32K", "read_b32 lane-linear", "bpermute_b32", "read_b64 random",
				 "read_u16 random", "read_b32 same addr", "read_b32 exec 40%", "read_b32 idle->addr0 40%",
				 "read_b32 random 1K" };

template <int DS, int NDS, int NVALU>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t seed)
{
	extern __shared__ uint32_t lds[];
	for (uint32_t i = threadIdx.x; i < 40960; i += blockDim.x)
		lds[i] = i * 2654435761u;
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63;
	uint32_t x = threadIdx.x * 747796405u + seed, acc = 0;
	uint32_t f0 = x ^ 1, f1 = x ^ 2, f2 = x ^ 3, f3 = x ^ 4;
	const bool live = ((threadIdx.x * 2654435761u) >> 24) % 100 < 40;
	for (int it = 0; it < ITERS; it++) {
		uint32_t r[NDS > 0 ? NDS : 1];
		if (NDS) {                                               // xorshift32: 6 full-rate VALU per iteration
			x ^= x << 13;
			x ^= x >> 17;
			x ^= x << 5;
		}
#pragma unroll
		for (int d = 0; d < NDS; d++) {
			const uint32_t a = (x >> (d + 2)) ^ (x << (9 - d));   // 3 VALU per address (+1 mask)
			r[d] = 0;
			if (DS == DS_ADDRONLY)
				r[d] = a & 0x7ffc;
			else if (DS == DS_B32_RANDOM)
				r[d] = *reinterpret_cast<lds_u32_t *>(a & 0x7ffc);
			else if (DS == DS_B32_SMALLTAB)
				r[d] = *reinterpret_cast<lds_u32_t *>(a & 0x3fc);
			else if (DS == DS_B32_LINEAR)
				r[d] = *reinterpret_cast<lds_u32_t *>((lane << 2) + ((it * 64 + d * 256) & 0x7f00));
			else if (DS == DS_BPERMUTE)
				r[d] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)a, (int)f0);
			else if (DS == DS_B64_RANDOM) {
				const uint64_t v = *reinterpret_cast<lds_u64_t *>(a & 0x7ff8);
				r[d] = (uint32_t)v ^ (uint32_t)(v >> 32);
			} else if (DS == DS_U16_RANDOM)
				r[d] = *reinterpret_cast<lds_u16_t *>(a & 0x7ffe);
			else if (DS == DS_B32_SAMEADDR)
				r[d] = *reinterpret_cast<lds_u32_t *>((it * 4 + d * 4) & 0x7ffc);
			else if (DS == DS_B32_MASKED40) {
				if (live)
					r[d] = *reinterpret_cast<lds_u32_t *>(a & 0x7ffc);
			} else if (DS == DS_B32_PARKED40)
				r[d] = *reinterpret_cast<lds_u32_t *>(live ? (a & 0x7ffc) : 0u);
		}
		// filler: four independent chains of full-rate ops (v_bitop3 / v_xor / v_lshrrev mix)
#pragma unroll
		for (int j = 0; j < NVALU / 4; j++) {
			f0 = __builtin_amdgcn_bitop3_b32(f0, f1, x, 0x96);
			f1 = (f1 >> 1) ^ f2;                                  // 2 ops; counts as 2 of the 4
			f2 = __builtin_amdgcn_bitop3_b32(f2, f3, f0, 0xe8);
			asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
		}
#pragma unroll
		for (int d = 0; d < NDS; d++)
			acc ^= r[d];                                          // 1 VALU per result (or folded into xor3)
		f3 ^= acc;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc ^ f0 ^ f1 ^ f2 ^ f3;
}

static double g_clock_ghz = 2.4;

template <int DS, int NDS, int NVALU>
static double run(uint32_t *d_out)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<DS, NDS, NVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
	hipLaunchKernelGGL((k<DS, NDS, NVALU>), dim3(256), dim3(1024), 163840, 0, d_out, 1u);
	(void)hipEventRecord(e0);
	hipLaunchKernelGGL((k<DS, NDS, NVALU>), dim3(256), dim3(1024), 163840, 0, d_out, 2u);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	// CU-cycles per loop iteration of ONE wave, if the 16 waves of the CU ran back to back
	return ms * 1e-3 * g_clock_ghz * 1e9 / ((double)ITERS * 16);
}

template <int DS>
static void row(uint32_t *d_out)
{
	// baselines: the same address arithmetic without the DS instruction
	const double a8 = run<DS_ADDRONLY, 8, 0>(d_out), a8v64 = run<DS_ADDRONLY, 8, 64>(d_out);
	const double a4v64 = run<DS_ADDRONLY, 4, 64>(d_out), a8v128 = run<DS_ADDRONLY, 8, 128>(d_out);
	const double d8 = run<DS, 8, 0>(d_out);
	const double d8v64 = run<DS, 8, 64>(d_out);
	const double d8v128 = run<DS, 8, 128>(d_out);
	const double d4v64 = run<DS, 4, 64>(d_out);
	printf("%-26s | 8 DS, no filler %6.1f (addr only %5.1f: +%.2f/op) | +64 VALU %6.1f (addr only %5.1f: +%.2f/op) | 4 DS + 64 VALU %6.1f (+%.2f/op) | "
	       "8 DS + 128 VALU %6.1f (addr only %5.1f: +%.2f/op)\n",
	       ds_name[DS], d8, a8, (d8 - a8) / 8, d8v64, a8v64, (d8v64 - a8v64) / 8, d4v64, (d4v64 - a4v64) / 4,
	       d8v128, a8v128, (d8v128 - a8v128) / 8);
	fflush(stdout);
}

int main()
{
	uint32_t *d_out;
	(void)hipMalloc(&d_out, 256 * 1024 * 4);
	int khz = 0;
	(void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
	if (khz > 0)
		g_clock_ghz = khz * 1e-6;
	printf("clock %.2f GHz; numbers are CU-cycles per wave-iteration (16 waves per CU): DS cost includes 1 VALU per address + 1 per result\n", g_clock_ghz);
	row<DS_B32_RANDOM>(d_out);
	row<DS_B32_SMALLTAB>(d_out);
	row<DS_B32_LINEAR>(d_out);
	row<DS_B32_SAMEADDR>(d_out);
	row<DS_BPERMUTE>(d_out);
	row<DS_B64_RANDOM>(d_out);
	row<DS_U16_RANDOM>(d_out);
	row<DS_B32_MASKED40>(d_out);
	row<DS_B32_PARKED40>(d_out);
	return 0;
}
