// synth.hip -- on-device synthetic traffic: the HIP twin of libbtbb_amd/synth.py
// (noise_words / injection_params / make_stream).  A 4 GiB stream is generated in
// place in HBM in a few milliseconds, and any slice can be regenerated on the host
// for the CPU baseline and for parity checks because the generator is counter based.
//
// The reference has no transmitter; the sync-word encoder below restates the
// Bluetooth (64,30) code as used by btbb_gen_syncword (lib/src/bluetooth_packet.c:188-199).
#include "common.h"

#define INJECT_SALT 0xA5A5A5A55A5A5A5AULL

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
	uint64_t z = x + 0x9E3779B97F4A7C15ULL;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}

struct SynthArgs {
	uint64_t *words;
	uint64_t first_word;
	uint64_t n_words;
	uint64_t seed;
	uint32_t stride;
	int64_t fixed_lap;
	uint32_t err_cycle;
	uint64_t sw_default;
	uint64_t gen_rows[24];
};

__global__ __launch_bounds__(256) void noise_kernel(SynthArgs a)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	for (; i < a.n_words; i += step)
		a.words[i] = splitmix64(a.seed + a.first_word + i);
}

// one thread per injected sync word; distinct injections never share a word
// (stride >= 512, jitter < stride - 256), so plain read-modify-write is race free
__global__ __launch_bounds__(256) void inject_kernel(SynthArgs a, uint64_t k0, uint64_t k_count)
{
	uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= k_count)
		return;
	uint64_t k = k0 + idx;
	uint64_t base = (a.seed ^ INJECT_SALT) + k * 4;
	uint64_t h0 = splitmix64(base), h1 = splitmix64(base + 1), h2 = splitmix64(base + 2);
	uint64_t pos = k * a.stride + 64 + h0 % (uint64_t)(a.stride - 256);
	uint32_t lap = a.fixed_lap >= 0 ? (uint32_t)a.fixed_lap & 0xffffff : (uint32_t)h1 & 0xffffff;
	uint32_t nerr = (uint32_t)(k % a.err_cycle);
	uint64_t mask = 0;
	for (uint32_t j = 0; j < 5; j++)
		if (nerr > j)
			mask ^= 1ULL << (((h2 >> (8 * j)) & 0xff) % 57);
	uint64_t sw = a.sw_default;
	for (int i = 0; i < 24; i++)
		if (lap & (0x800000u >> i))
			sw ^= a.gen_rows[i];
	sw ^= mask;

	uint64_t lo_bit = a.first_word * 64, hi_bit = (a.first_word + a.n_words) * 64;
	if (pos + 64 <= lo_bit || pos >= hi_bit)
		return;
	uint32_t sh = (uint32_t)(pos & 63);
	int64_t w0 = (int64_t)(pos >> 6) - (int64_t)a.first_word;
	if (w0 >= 0 && (uint64_t)w0 < a.n_words) {
		uint64_t m = 0xffffffffffffffffULL << sh;
		a.words[w0] = (a.words[w0] & ~m) | (sw << sh);
	}
	if (sh && w0 + 1 >= 0 && (uint64_t)(w0 + 1) < a.n_words) {
		uint64_t m = 0xffffffffffffffffULL >> (64 - sh);
		a.words[w0 + 1] = (a.words[w0 + 1] & ~m) | (sw >> (64 - sh));
	}
}

extern "C" int btbbx_synth_device(uint64_t *d_words, uint64_t first_word, uint64_t n_words,
				  uint64_t seed, uint32_t stride, int64_t fixed_lap, uint32_t err_cycle,
				  void *hip_stream)
{
	if (!d_words || stride < 512 || err_cycle == 0) {
		set_error("btbbx_synth_device: bad argument (stride >= 512, err_cycle >= 1)");
		return BTBBX_E_ARG;
	}
	if (n_words == 0)
		return BTBBX_OK;
	const HostTables &t = host_tables();
	SynthArgs a;
	a.words = d_words;
	a.first_word = first_word;
	a.n_words = n_words;
	a.seed = seed;
	a.stride = stride;
	a.fixed_lap = fixed_lap;
	a.err_cycle = err_cycle;
	a.sw_default = t.sw_default;
	for (int i = 0; i < 24; i++)
		a.gen_rows[i] = t.gen_rows[i];
	hipStream_t s = (hipStream_t)hip_stream;
	uint64_t blocks = (n_words + 255) / 256;
	if (blocks > 16384) blocks = 16384;
	hipLaunchKernelGGL(noise_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, a);
	uint64_t lo = first_word * 64, hi = (first_word + n_words) * 64;
	uint64_t k0 = lo / stride ? lo / stride - 1 : 0;
	uint64_t k1 = hi / stride + 1;
	uint64_t count = k1 - k0;
	hipLaunchKernelGGL(inject_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, s, a, k0, count);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}
