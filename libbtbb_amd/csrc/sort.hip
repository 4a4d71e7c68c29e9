// sort.hip -- (stream, offset) order for hit lists while they are still in HBM.
//
// The scan kernels append hits unordered; callers of the reference get them in stream order
// (lib/src/bluetooth_packet.c:444-464 is called on a sliding window).  Sorting 10^6 records on the
// host costs ~60 ms and used to be most of the PCIe-inclusive time of the ingest paths; on the
// device it is a few hundred microseconds.  The sort itself is rocPRIM's LSD radix sort (AMD's own
// primitive library, the plain-library case like a rocBLAS GEMM); this file adds the key
// extraction and the scratch management.
#include <string.h>
#include <mutex>
#include "common.h"
#include <rocprim/rocprim.hpp>

struct __attribute__((aligned(16))) HitRec { uint64_t a, b; };      // a btbbx_hit as an opaque 16-byte value

// The key is (stream << offset_bits) | offset with just as many bits as the list needs: an LSD radix sort pays
// one pass over keys and 16-byte values per digit, and a 4 GiB capture of 79 channels needs 39 key bits, not 64.
// Pass 1 ORs all offsets and all stream numbers (one atomic per wave), pass 2 builds the keys.
__global__ __launch_bounds__(256) void hit_extent_kernel(const btbbx_hit *hits, uint32_t n, unsigned long long *extent)
{
	// grid-stride: a few hundred workgroups, one pair of atomics each (one pair per wave of a 10^6-hit list
	// was 20 000 atomics on two addresses and took longer than the sort passes it saves)
	__shared__ unsigned long long part[2][4];
	unsigned long long off = 0, st = 0;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		off |= (unsigned long long)hits[i].offset;
		st |= hits[i].stream;
	}
	for (int d = 32; d; d >>= 1) {
		off |= __shfl_xor(off, d);
		st |= __shfl_xor(st, d);
	}
	if ((threadIdx.x & 63) == 0) {
		part[0][threadIdx.x >> 6] = off;
		part[1][threadIdx.x >> 6] = st;
	}
	__syncthreads();
	if (threadIdx.x < 2)
		atomicOr(&extent[threadIdx.x], part[threadIdx.x][0] | part[threadIdx.x][1] | part[threadIdx.x][2] | part[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void hit_keys_kernel(const btbbx_hit *hits, uint32_t n, uint64_t *keys, uint32_t offset_bits)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		keys[i] = ((uint64_t)hits[i].stream << offset_bits) | hits[i].offset;
}

// one scratch block per device: keys in | keys out | values out | rocPRIM temporary
struct SortScratch {
	std::mutex lock;
	void *block = nullptr;
	size_t bytes = 0;
};
static SortScratch sort_scratch[BTBBX_MAX_DEVICES];

void sort_scratch_release()         // btbbx_shutdown
{
	int home = 0;
	(void)hipGetDevice(&home);
	for (int d = 0; d < BTBBX_MAX_DEVICES; d++) {
		SortScratch &s = sort_scratch[d];
		std::lock_guard<std::mutex> g(s.lock);
		if (s.block) {
			(void)hipSetDevice(d);
			(void)hipFree(s.block);
		}
		s.block = nullptr;
		s.bytes = 0;
	}
	(void)hipSetDevice(home);
}

extern "C" int btbbx_sort_hits_device(btbbx_hit *d_hits, uint32_t n, void *hip_stream)
{
	if (n < 2)
		return BTBBX_OK;
	hipStream_t stream = (hipStream_t)hip_stream;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BTBBX_MAX_DEVICES)
		dev = 0;
	SortScratch &sc = sort_scratch[dev];
	void *&sort_block = sc.block;
	size_t &sort_block_bytes = sc.bytes;
	std::lock_guard<std::mutex> g(sc.lock);
	size_t tmp_bytes = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (HitRec *)nullptr,
					  (HitRec *)nullptr, (size_t)n, 0u, 64u, stream));
	const size_t key_bytes = ((size_t)n * 8 + 255) & ~(size_t)255, val_bytes = ((size_t)n * 16 + 255) & ~(size_t)255;
	const size_t need = 2 * key_bytes + val_bytes + tmp_bytes + 512;
	if (need > sort_block_bytes) {
		if (sort_block)
			(void)hipFree(sort_block);
		sort_block = nullptr;
		sort_block_bytes = 0;
		const size_t want = need + need / 2;
		hipError_t e = hipMalloc(&sort_block, want);
		if (e != hipSuccess)
			return hip_fail(e, "hipMalloc(sort scratch)");
		sort_block_bytes = want;
	}
	char *p = (char *)sort_block;
	uint64_t *k_in = (uint64_t *)p, *k_out = (uint64_t *)(p + key_bytes);
	HitRec *v_out = (HitRec *)(p + 2 * key_bytes);
	void *tmp = p + 2 * key_bytes + val_bytes;
	unsigned long long *d_extent = (unsigned long long *)(p + 2 * key_bytes + val_bytes + ((tmp_bytes + 255) & ~(size_t)255));
	unsigned long long extent[2] = {0, 0};
	HIP_TRY(hipMemsetAsync(d_extent, 0, sizeof(extent), stream));
	hipLaunchKernelGGL(hit_extent_kernel, dim3((n + 255) / 256 < 512 ? (n + 255) / 256 : 512), dim3(256), 0, stream, d_hits, n, d_extent);
	HIP_TRY(hipMemcpyAsync(extent, d_extent, sizeof(extent), hipMemcpyDeviceToHost, stream));
	HIP_TRY(hipStreamSynchronize(stream));
	uint32_t offset_bits = 1, stream_bits = 0;
	while (offset_bits < 64 && (extent[0] >> offset_bits))
		offset_bits++;
	while (stream_bits < 16 && (extent[1] >> stream_bits))
		stream_bits++;
	if (offset_bits + stream_bits > 64) {       // cannot happen with 16-bit stream numbers and offsets below 2^48
		set_error("btbbx_sort_hits_device: offsets too large for the sort key");
		return BTBBX_E_ARG;
	}
	hipLaunchKernelGGL(hit_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_hits, n, k_in, offset_bits);
	HIP_TRY(hipGetLastError());
	HIP_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, (HitRec *)d_hits, v_out, (size_t)n, 0u,
					  offset_bits + stream_bits, stream));
	HIP_TRY(hipMemcpyAsync(d_hits, v_out, (size_t)n * sizeof(btbbx_hit), hipMemcpyDeviceToDevice, stream));
	// the scratch is shared: finish before another caller may reuse it
	HIP_TRY(hipStreamSynchronize(stream));
	return BTBBX_OK;
}
