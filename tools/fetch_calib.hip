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

// ---- round 5: the access shapes of the packet kernels (the scan's factor 2 above was calibrated on 8-B/lane streaming
// loads only).  FETCH_SIZE tallies every fabric read request at 64 B; what has to be found out per shape is how many bytes
// a request stands for.  One 8-byte word per 128-byte line and one per 64-byte sector over the same span tell: if both
// report the same count, the L2 asks for whole 128-byte lines whatever part is used (doubling is right for sparse shapes
// too); if the second reports twice the first, requests are 64-byte sectors (doubling a sparse shape overstates it 2 x).
__global__ __launch_bounds__(256) void word_per_stride(const uint64_t *p, uint64_t n_items, uint32_t stride_bytes, uint32_t first_byte, uint64_t *sink)
{
	uint64_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (uint64_t)gridDim.x * 256)
		acc ^= *reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(p) + i * stride_bytes + first_byte);
	if (acc == 0x1234567)
		*sink = acc;
}
// decode_hits_kernel's staging: windows of `win_dwords` consecutive dwords, one window per 512-byte slot, starting at a
// dword that is not aligned to anything (12 + 4 * (slot % 8) bytes into the slot), brought in with global_load_lds --
// every lane one dword of the concatenated windows, as the owner map of the kernel does it
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(1))) const uint32_t glb_u32_t;
__global__ __launch_bounds__(256) void windows_to_lds(const uint32_t *p, uint64_t n_slots, uint32_t win_dwords, uint64_t *sink)
{
	__shared__ uint32_t stage[4][64 * 8];
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t acc = 0;
	// a wave takes 64 slots per trip; dword D of the concatenated windows belongs to slot D / win_dwords
	for (uint64_t base = ((uint64_t)blockIdx.x * 4 + wave) * 64; base < n_slots; base += (uint64_t)gridDim.x * 4 * 64) {
		const uint32_t total = 64 * win_dwords;
		for (uint32_t d0 = 0; d0 < total; d0 += 512) {
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) {
				const uint32_t d = d0 + 64 * k + lane;
				if (d < total) {
					const uint64_t slot = base + d / win_dwords;
					const uint32_t *src = p + slot * 128 + 3 + (slot & 7) + d % win_dwords;
					__builtin_amdgcn_global_load_lds((glb_u32_t *)src, (lds_u32_t *)&stage[wave][64 * k], 4, 0, 0);
				}
			}
			asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
			acc ^= stage[wave][lane] ^ stage[wave][lane + 64 * 7];
		}
	}
	if (acc == 0x1234567)
		*sink = acc;
}
// gather_kernel's read: 98 consecutive dwords (391 bytes and a bit) per slot from the same unaligned starts, plain loads,
// consecutive lanes = consecutive dwords of one window
__global__ __launch_bounds__(256) void windows_plain(const uint32_t *p, uint64_t n_slots, uint32_t win_dwords, uint64_t *sink)
{
	uint32_t acc = 0;
	const uint32_t lane = threadIdx.x & 63;
	for (uint64_t slot = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); slot < n_slots; slot += (uint64_t)gridDim.x * 4)
		for (uint32_t d = lane; d < win_dwords; d += 64)
			acc ^= p[slot * 128 + 3 + (slot & 7) + d];
	if (acc == 0x1234567)
		*sink = acc;
}
// 16-byte records, one per lane, consecutive (hit lists, btbbx_pkt_in, the long-payload list)
// = read_b128 above.

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
	for (int rep = 0; rep < 3; rep++) {
		// the same 4 GiB span every time
		hipLaunchKernelGGL(word_per_stride, dim3(2048), dim3(256), 0, 0, d, bytes / 128, 128u, 8u, sink);     // one word per 128-byte line
		hipLaunchKernelGGL(word_per_stride, dim3(2048), dim3(256), 0, 0, d, bytes / 64, 64u, 8u, sink);       // one word per 64-byte sector
		hipLaunchKernelGGL(word_per_stride, dim3(2048), dim3(256), 0, 0, d, bytes / 400, 400u, 8u, sink);     // uap_table_kernel: word 1 of every 400-byte row
		hipLaunchKernelGGL(word_per_stride, dim3(2048), dim3(256), 0, 0, d, bytes / 32, 32u, 8u, sink);       // one word per 32 bytes
	}
	for (int rep = 0; rep < 3; rep++) {
		hipLaunchKernelGGL(windows_to_lds, dim3(2048), dim3(256), 0, 0, (const uint32_t *)d, bytes / 512 - 2, 100u, sink);   // a full five-slot packet
		hipLaunchKernelGGL(windows_to_lds, dim3(2048), dim3(256), 0, 0, (const uint32_t *)d, bytes / 512 - 2, 12u, sink);    // a DM1 / short packet (366 symbols)
		hipLaunchKernelGGL(windows_plain, dim3(2048), dim3(256), 0, 0, (const uint32_t *)d, bytes / 512 - 2, 98u, sink);     // gather_kernel's 391 bytes
	}
	(void)hipDeviceSynchronize();
	printf("read %llu bytes per kernel\n", (unsigned long long)bytes);
	return 0;
}
