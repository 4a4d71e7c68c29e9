// hop.hip -- Bluetooth BR hop selection and CLK1-27 reversal on the GPU.
//
// Replaces lib/src/bluetooth_piconet.c:171-362 (precalc, address_precalc, perm5/fast_perm,
// gen_hops), :443-472 (hop, aliased_channel, init_candidates) and :575-645 (channel_winnow,
// btbb_winnow).  The reference materialises the whole 2^27-entry pattern (128 MiB, about a
// second of CPU per address) and then filters candidate clocks by table look-ups.  Here the
// selection kernel is a pure function evaluated where it is needed:
//   * hop_sequence_kernel   fills sequence[first .. first+count) for callers that want the
//                           table (one lane per 64 hops; the permutation is applied to a
//                           32-byte window of the bank table held in registers);
//   * candidates / winnow   evaluate the kernel per candidate clock -- no table at all.
// Candidate lists stay in HBM in ascending order (as the reference keeps them) through
// ballot masks + a prefix over mask words + an ordered scatter.
#include <string.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "common.h"

#define HOP_NCHAN   79
#define HOP_TAB     272          // perm (<32) + e (<128) + f (<79) + 32 = at most 268
#define HOP_GROUPS  (1u << 21)   // values of CLK7-27 = groups of 64 hops
#define HOP_MAX_OBS 1024

struct HopArgs {
	uint32_t a1, b, c1, d1, e;
	uint32_t mod;                // 79, or used_channels under AFH
	uint32_t afh;
	uint8_t bank[80];
};

// butterfly stage s exchanges wires (hop_u(s), hop_v(s)); spec vol 2 part B 2.6.2.3
__device__ __host__ constexpr int hop_u(int s) { constexpr int u[14] = {0, 2, 1, 3, 0, 1, 0, 3, 1, 0, 2, 1, 0, 1}; return u[s]; }
__device__ __host__ constexpr int hop_v(int s) { constexpr int v[14] = {1, 3, 2, 4, 4, 3, 2, 4, 4, 3, 4, 3, 3, 2}; return v[s]; }

__device__ __forceinline__ void hop_build_tab(uint8_t *tab, const HopArgs &h)
{
	// every kernel using the table runs 256 lanes per workgroup
	tab[threadIdx.x] = h.bank[threadIdx.x % h.mod];
	if (threadIdx.x < HOP_TAB - 256)
		tab[256 + threadIdx.x] = h.bank[(256 + threadIdx.x) % h.mod];
	__syncthreads();
}

// index into tab for CLK1-27 value idx: perm5 output + e + f (+32 for odd clocks)
__device__ __forceinline__ uint32_t hop_tab_index(const HopArgs &h, uint32_t idx)
{
	const uint32_t y1 = idx & 1, x = (idx >> 1) & 31, t = idx >> 6;
	const uint32_t a = h.a1 ^ ((t >> 14) & 31);
	const uint32_t c = h.c1 ^ ((t >> 9) & 31) ^ (y1 ? 31u : 0u);
	const uint32_t ctl = (c << 9) | (h.d1 ^ (t & 511));
	uint32_t z = ((x + a) & 31) ^ h.b;
#pragma unroll
	for (int s = 13; s >= 0; s--) {
		const uint32_t sw = ((z >> hop_u(s)) ^ (z >> hop_v(s))) & (ctl >> s) & 1;
		z ^= (sw << hop_u(s)) | (sw << hop_v(s));
	}
	uint32_t f = (16u * t) % HOP_NCHAN;
	if (h.afh)
		f %= h.mod;                           // gen_hops' f_dash (:355), not single_hop's
	return z + h.e + f + 32u * y1;
}

__device__ __forceinline__ int hop_observable(uint32_t ch, int aliased)
{
	return aliased ? (int)((ch + 24) % 25) + 26 : (int)ch;
}

// ---- whole-table generation -----------------------------------------------------------
// One lane produces the 64 hops of one value of CLK7-27.  For a fixed clock parity the 32 hops
// over CLK2-6 = x are  bank'[K + perm(in(x))],  in(x) = ((x + a) mod 32) ^ b:  a 32-byte window W
// of the bank table, indexed through a permutation of x.  The window lives in 8 VGPRs and the
// permutation is applied to the ARRAY instead of to each index: a butterfly stage that exchanges
// index bits (u, v) is a conditional exchange of array elements -- whole registers when both bits
// select the register, byte shuffles with v_perm_b32 when a bit selects the byte -- and the final
// x -> (x + a) mod 32 is a byte rotation of the array.  About 6 VALU ops per hop and 16 LDS reads
// per 64 hops; an earlier version evaluated the permutation bit-sliced and transposed the planes
// (11 ops + 1 LDS read per hop, 48 us per pattern).
#define HOP_PAD 304               // bank' entries incl. padding so that 8 dwords can be read at any K

__device__ __forceinline__ uint32_t vperm(uint32_t hi, uint32_t lo, uint32_t sel)
{
	return __builtin_amdgcn_perm(hi, lo, sel);      // selector byte 0..3 -> lo, 4..7 -> hi
}

__device__ __forceinline__ uint32_t bitsel(uint32_t m, uint32_t one, uint32_t zero)
{
	return (one & m) | (zero & ~m);                 // one v_bitop3
}

// new[z] = old[z with index bits U and V exchanged] where the mask m is all ones, unchanged where
// it is 0.  Index z of the 32-entry array = register (z >> 2), byte (z & 3).
template <int U, int V> __device__ __forceinline__ void swap_index_bits(uint32_t (&r)[8], uint32_t m)
{
	static_assert(U < V && V < 5, "stage wires");
	if constexpr (U >= 2) {
		constexpr int mu = 1 << (U - 2), mv = 1 << (V - 2);
#pragma unroll
		for (int i = 0; i < 8; i++)
			if ((i & mu) && !(i & mv)) {
				const int j = i ^ mu ^ mv;
				const uint32_t a = r[i], b = r[j];
				r[i] = bitsel(m, b, a);
				r[j] = bitsel(m, a, b);
			}
	} else if constexpr (V >= 2) {
		constexpr int mv = 1 << (V - 2);
		// A = register with index bit V clear, B = its partner.  U = 0: A.bytes{1,3} <-> B.bytes{0,2};
		// U = 1: A.bytes{2,3} <-> B.bytes{0,1}.  Selectors: identity ^ (m & difference)
		constexpr uint32_t swa = U == 0 ? 0x06020400u : 0x05040100u, swb = U == 0 ? 0x07030501u : 0x07060302u;
		const uint32_t sa = 0x03020100u ^ (m & (swa ^ 0x03020100u));
		const uint32_t sb = 0x07060504u ^ (m & (swb ^ 0x07060504u));
#pragma unroll
		for (int i = 0; i < 8; i++)
			if (!(i & mv)) {
				const uint32_t a = r[i], b = r[i | mv];
				r[i] = vperm(b, a, sa);
				r[i | mv] = vperm(b, a, sb);
			}
	} else {
		const uint32_t s = 0x03020100u ^ (m & (0x03010200u ^ 0x03020100u));   // bytes 1 <-> 2
#pragma unroll
		for (int i = 0; i < 8; i++)
			r[i] = vperm(r[i], r[i], s);
	}
}

__device__ __forceinline__ uint32_t ctl_mask(uint32_t ctl, int k)
{
	return (uint32_t)__builtin_amdgcn_sbfe((int)ctl, k, 1);          // 0 or ~0
}

// the 14 stages in array order (stage 0 first, see the composition note in NOTEBOOK.md 3.6)
__device__ __forceinline__ void hop_permute_array(uint32_t (&r)[8], uint32_t ctl)
{
	swap_index_bits<0, 1>(r, ctl_mask(ctl, 0));
	swap_index_bits<2, 3>(r, ctl_mask(ctl, 1));
	swap_index_bits<1, 2>(r, ctl_mask(ctl, 2));
	swap_index_bits<3, 4>(r, ctl_mask(ctl, 3));
	swap_index_bits<0, 4>(r, ctl_mask(ctl, 4));
	swap_index_bits<1, 3>(r, ctl_mask(ctl, 5));
	swap_index_bits<0, 2>(r, ctl_mask(ctl, 6));
	swap_index_bits<3, 4>(r, ctl_mask(ctl, 7));
	swap_index_bits<1, 4>(r, ctl_mask(ctl, 8));
	swap_index_bits<0, 3>(r, ctl_mask(ctl, 9));
	swap_index_bits<2, 4>(r, ctl_mask(ctl, 10));
	swap_index_bits<1, 3>(r, ctl_mask(ctl, 11));
	swap_index_bits<0, 3>(r, ctl_mask(ctl, 12));
	swap_index_bits<1, 2>(r, ctl_mask(ctl, 13));
}

// new[x] = old[((x + a) mod 32) ^ b]; b is the same for every lane of the launch
__device__ __forceinline__ void hop_input_map(uint32_t (&r)[8], uint32_t a, uint32_t b)
{
	// z ^ (b & 3): one byte shuffle with a launch-uniform selector
	const uint32_t xsel = (b & 1 ? 0x02030001u : 0x03020100u) ^ (b & 2 ? 0x02020202u : 0u);
#pragma unroll
	for (int i = 0; i < 8; i++)
		r[i] = vperm(r[i], r[i], xsel);
	// z ^ (b & 12): exchange registers
	const uint32_t m4 = 0u - ((b >> 2) & 1u), m8 = 0u - ((b >> 3) & 1u);
#pragma unroll
	for (int i = 0; i < 8; i += 2) {
		const uint32_t p = r[i], q = r[i + 1];
		r[i] = bitsel(m4, q, p);
		r[i + 1] = bitsel(m4, p, q);
	}
#pragma unroll
	for (int i = 0; i < 8; i++)
		if (!(i & 2)) {
			const uint32_t p = r[i], q = r[i + 2];
			r[i] = bitsel(m8, q, p);
			r[i + 2] = bitsel(m8, p, q);
		}
	// rotate by whole registers (a >> 2), then by bytes (a & 3)
#pragma unroll
	for (int k = 0; k < 3; k++) {
		const uint32_t m = ctl_mask(a, 2 + k);
		uint32_t n[8];
#pragma unroll
		for (int i = 0; i < 8; i++)
			n[i] = bitsel(m, r[(i + (1 << k)) & 7], r[i]);
#pragma unroll
		for (int i = 0; i < 8; i++)
			r[i] = n[i];
	}
	const uint32_t s = a & 3;
	const uint32_t first = r[0];
#pragma unroll
	for (int i = 0; i < 7; i++)
		r[i] = __builtin_amdgcn_alignbyte(r[i + 1], r[i], s);
	r[7] = __builtin_amdgcn_alignbyte(first, r[7], s);
}

__global__ __launch_bounds__(256) void hop_sequence_kernel(HopArgs h, uint32_t t0, uint32_t nt, uint4 *out)
{
	// four copies of the bank table, copy s shifted by s bytes: any 32-byte window starts on a dword
	__shared__ uint32_t tabs[4][HOP_PAD / 4];
	__shared__ uint4 stage[4][64 * 5];
	for (uint32_t i = threadIdx.x; i < 4 * HOP_PAD; i += 256) {
		const uint32_t s = i / HOP_PAD, k = i % HOP_PAD, v = k + s;
		reinterpret_cast<uint8_t *>(tabs[s])[k] = v < HOP_TAB ? h.bank[v % h.mod] : 0;
	}
	__syncthreads();
	const uint32_t g = blockIdx.x * 256 + threadIdx.x;
	if (g - (threadIdx.x & 63) >= nt)               // whole wave out of range
		return;
	const uint32_t t = t0 + g;
	const uint32_t a = h.a1 ^ ((t >> 14) & 31);
	const uint32_t c = h.c1 ^ ((t >> 9) & 31);
	const uint32_t d = h.d1 ^ (t & 511);
	uint32_t f = (16u * t) % HOP_NCHAN;
	if (h.afh)
		f %= h.mod;
	const uint32_t k0 = h.e + f;
	const uint32_t *win = &tabs[k0 & 3][k0 >> 2];

	uint32_t r0[8], r1[8];
#pragma unroll
	for (int i = 0; i < 8; i++) {
		r0[i] = win[i];                         // even clocks: window at K
		r1[i] = win[i + 8];                     // odd clocks: window at K + 32
	}
	hop_permute_array(r0, (c << 9) | d);
	hop_permute_array(r1, ((c ^ 31u) << 9) | d);
	hop_input_map(r0, a, h.b);
	hop_input_map(r1, a, h.b);

	// sequence order: x / even, x / odd, x + 1 / even, ...
	uint4 v[4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		v[q].x = vperm(r1[2 * q], r0[2 * q], 0x05010400u);
		v[q].y = vperm(r1[2 * q], r0[2 * q], 0x07030602u);
		v[q].z = vperm(r1[2 * q + 1], r0[2 * q + 1], 0x05010400u);
		v[q].w = vperm(r1[2 * q + 1], r0[2 * q + 1], 0x07030602u);
	}
	// A lane holds 64 consecutive bytes; written directly, one store instruction would touch 64
	// different 64-byte segments.  Transpose through LDS (80-byte lane pitch against bank
	// conflicts) so that every store instruction of a wave writes 1 KiB contiguously.
	uint4 *mine = reinterpret_cast<uint4 *>(stage[threadIdx.x >> 6]);
	const uint32_t lane = threadIdx.x & 63;
#pragma unroll
	for (int q = 0; q < 4; q++)
		mine[lane * 5 + q] = v[q];
	__builtin_amdgcn_wave_barrier();
	uint4 *dst = out + (size_t)(g - lane) * 4;       // the wave's 4 KiB
	const uint32_t wave_n = min(64u, nt - (g - lane)) * 4;   // uint4 items this wave owns
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const uint32_t u = 64 * j + lane;
		if (u < wave_n)
			dst[u] = mine[(u >> 2) * 5 + (u & 3)];
	}
}

__global__ __launch_bounds__(256) void hop_channels_kernel(HopArgs h, const uint32_t *clocks, uint32_t n, uint8_t *channels)
{
	__shared__ uint8_t tab[HOP_TAB];
	hop_build_tab(tab, h);
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n)
		channels[i] = tab[hop_tab_index(h, clocks[i] & (BTBBX_SEQUENCE_LENGTH - 1))];
}

// ---- candidate lists ------------------------------------------------------------------
// init_candidates: item j is the clock known6 + 64 j; one ballot word per wave
__global__ __launch_bounds__(256) void hop_candidate_mask_kernel(HopArgs h, uint32_t known6, int channel, int aliased,
								  uint64_t *masks)
{
	__shared__ uint8_t tab[HOP_TAB];
	hop_build_tab(tab, h);
	const uint32_t j = blockIdx.x * 256 + threadIdx.x;         // grid covers exactly HOP_GROUPS
	const int ch = hop_observable(tab[hop_tab_index(h, known6 + 64u * j)], aliased);
	const uint64_t m = __ballot(ch == channel);
	if ((threadIdx.x & 63) == 0)
		masks[j >> 6] = m;
}

// exclusive prefix of popcounts over the mask words; one workgroup
__global__ __launch_bounds__(1024) void hop_mask_prefix_kernel(const uint64_t *masks, uint32_t nwords, uint32_t *prefix,
								uint32_t *total)
{
	__shared__ uint32_t part[1024];
	const uint32_t per = (nwords + 1023) / 1024;
	const uint32_t lo = threadIdx.x * per, hi = min(lo + per, nwords);
	uint32_t sum = 0;
	for (uint32_t w = lo; w < hi; w++)
		sum += __popcll(masks[w]);
	part[threadIdx.x] = sum;
	__syncthreads();
	for (uint32_t step = 1; step < 1024; step <<= 1) {         // Hillis-Steele, inclusive
		uint32_t v = threadIdx.x >= step ? part[threadIdx.x - step] : 0;
		__syncthreads();
		part[threadIdx.x] += v;
		__syncthreads();
	}
	uint32_t run = part[threadIdx.x] - sum;
	for (uint32_t w = lo; w < hi; w++) {
		prefix[w] = run;
		run += __popcll(masks[w]);
	}
	if (threadIdx.x == 1023)
		*total = part[1023];
}

// ordered scatter: src == nullptr -> the value of item i is base + 64 i
__global__ __launch_bounds__(256) void hop_scatter_kernel(const uint64_t *masks, const uint32_t *prefix, uint32_t nwords,
							   const uint32_t *src, uint32_t base, uint32_t *dst)
{
	const uint32_t w = blockIdx.x * 256 + threadIdx.x;
	if (w >= nwords)
		return;
	uint64_t m = masks[w];
	uint32_t o = prefix[w];
	while (m) {
		const uint32_t i = w * 64 + (uint32_t)__builtin_ctzll(m);
		m &= m - 1;
		dst[o++] = src ? src[i] : base + 64u * i;
	}
}

struct HopObs {                   // one observed hop: clock distance to the first packet, channel
	int32_t offset;
	int32_t channel;              // as the reference's `char channel`: > 127 never matches
};

// per candidate: how many of the observations it agrees with before the first mismatch
__global__ __launch_bounds__(256) void hop_winnow_kernel(HopArgs h, const uint32_t *cand, uint32_t n, const HopObs *obs,
							  uint32_t n_obs, int aliased, uint16_t *agree, uint32_t *hist)
{
	__shared__ uint8_t tab[HOP_TAB];
	__shared__ uint32_t lhist[HOP_MAX_OBS + 1];
	for (uint32_t i = threadIdx.x; i <= n_obs; i += 256)
		lhist[i] = 0;
	hop_build_tab(tab, h);
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;
	if (i < n) {
		const uint32_t c = cand[i];
		uint32_t k = 0;
		for (; k < n_obs; k++) {
			const HopObs o = obs[k];
			const uint32_t idx = (c + (uint32_t)o.offset) & (BTBBX_SEQUENCE_LENGTH - 1);
			if (hop_observable(tab[hop_tab_index(h, idx)], aliased) != o.channel)
				break;
		}
		agree[i] = (uint16_t)k;
		atomicAdd(&lhist[k], 1u);
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k <= n_obs; k += 256)
		if (lhist[k])
			atomicAdd(&hist[k], lhist[k]);
}

struct WinnowVerdict {
	uint32_t stop;        // observations applied before the one that left <= 1 candidate (n_obs if none)
	uint32_t count;       // candidates left after that one (or after all)
	uint32_t keep_above;  // survivors are the candidates with agree > keep_above
	uint32_t cand0;       // first survivor (filled by the scatter pass)
};

// hist[k] = candidates whose first mismatch is observation k (k = n_obs: none)
__global__ __launch_bounds__(1024) void hop_verdict_kernel(const uint32_t *hist, uint32_t n, uint32_t n_obs, WinnowVerdict *v)
{
	__shared__ uint32_t cum[1024];
	__shared__ uint32_t first;
	const uint32_t k = threadIdx.x;
	if (k == 0)
		first = n_obs;
	cum[k] = k < n_obs ? hist[k] : 0;
	__syncthreads();
	for (uint32_t step = 1; step < 1024; step <<= 1) {
		uint32_t x = k >= step ? cum[k - step] : 0;
		__syncthreads();
		cum[k] += x;
		__syncthreads();
	}
	// candidates left after applying observation k = n - cum[k]
	if (k < n_obs && n - cum[k] <= 1)
		atomicMin(&first, k);
	__syncthreads();
	if (k == 0) {
		const uint32_t last = first < n_obs ? first : n_obs - 1;
		v->stop = first;
		v->keep_above = last;
		v->count = n - cum[last];
		v->cand0 = 0;
	}
}

__global__ __launch_bounds__(256) void hop_agree_mask_kernel(const uint16_t *agree, uint32_t n, const WinnowVerdict *v,
							      uint64_t *masks)
{
	const uint32_t i = blockIdx.x * 256 + threadIdx.x;         // grid covers ceil(n / 64) whole words
	const uint64_t m = __ballot(i < n && agree[i] > v->keep_above);
	if ((threadIdx.x & 63) == 0)
		masks[i >> 6] = m;
}


// One workgroup does a whole winnowing call for short lists (the common case after the first
// observed hop): agreement counts, verdict and ordered compaction in a single launch.
#define HOP_SMALL_N 16384
__global__ __launch_bounds__(1024) void hop_winnow_small_kernel(HopArgs h, const uint32_t *cand, uint32_t n,
								 const HopObs *obs, uint32_t n_obs, int aliased,
								 uint32_t *dst, WinnowVerdict *v)
{
	__shared__ uint8_t tab[HOP_TAB];
	__shared__ uint16_t agree[HOP_SMALL_N];
	__shared__ uint32_t cum[1024];
	__shared__ uint32_t wave_cnt[16];
	__shared__ uint32_t first, base;
	const uint32_t tid = threadIdx.x;
	if (tid < 256) {
		tab[tid] = h.bank[tid % h.mod];
		if (tid < HOP_TAB - 256)
			tab[256 + tid] = h.bank[(256 + tid) % h.mod];
	}
	cum[tid] = 0;
	if (tid == 0) {
		first = n_obs;
		base = 0;
	}
	__syncthreads();
	for (uint32_t i = tid; i < n; i += 1024) {
		const uint32_t c = cand[i];
		uint32_t k = 0;
		for (; k < n_obs; k++) {
			const HopObs o = obs[k];
			const uint32_t idx = (c + (uint32_t)o.offset) & (BTBBX_SEQUENCE_LENGTH - 1);
			if (hop_observable(tab[hop_tab_index(h, idx)], aliased) != o.channel)
				break;
		}
		agree[i] = (uint16_t)k;
		if (k < n_obs)
			atomicAdd(&cum[k], 1u);                 // first mismatch at observation k
	}
	__syncthreads();
	for (uint32_t step = 1; step < 1024; step <<= 1) {
		const uint32_t x = tid >= step ? cum[tid - step] : 0;
		__syncthreads();
		cum[tid] += x;
		__syncthreads();
	}
	if (tid < n_obs && n - cum[tid] <= 1)
		atomicMin(&first, tid);
	__syncthreads();
	const uint32_t keep = first < n_obs ? first : n_obs - 1;
	for (uint32_t i0 = 0; i0 < n; i0 += 1024) {                 // ordered compaction, 1024 at a time
		const uint32_t i = i0 + tid;
		const bool live = i < n && agree[i] > keep;
		const uint64_t m = __ballot(live);
		if ((tid & 63) == 0)
			wave_cnt[tid >> 6] = (uint32_t)__popcll(m);
		__syncthreads();
		uint32_t before = base;
		for (uint32_t w = 0; w < (tid >> 6); w++)
			before += wave_cnt[w];
		if (live)
			dst[before + (uint32_t)__popcll(m & ((1ull << (tid & 63)) - 1))] = cand[i];
		__syncthreads();
		if (tid == 0) {
			uint32_t all = 0;
			for (int w = 0; w < 16; w++)
				all += wave_cnt[w];
			base += all;
		}
		__syncthreads();
	}
	if (tid == 0) {
		v->stop = first;
		v->keep_above = keep;
		v->count = n - cum[keep];
		v->cand0 = 0;
	}
}

// cand0 of the verdict, read after the compaction on the same stream
__global__ void hop_first_kernel(const uint32_t *cand, WinnowVerdict *v)
{
	if (v->count)
		v->cand0 = cand[0];
}

// ---- host side ------------------------------------------------------------------------
// hop selection needs a GPU but none of the btbb_init() tables
static int hop_device()
{
	if (ctx().ready)
		return BTBBX_OK;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
		set_error("hop: no usable HIP device");
		return BTBBX_E_NODEVICE;
	}
	return BTBBX_OK;
}

static bool hip_bad(hipError_t e, const char *what) { return e != hipSuccess && hip_fail(e, what) != 0; }

static int hop_args(const btbbx_hop_cfg *cfg, HopArgs *h)
{
	if (!cfg) {
		set_error("hop: NULL configuration");
		return BTBBX_E_ARG;
	}
	const uint32_t address = cfg->address & 0xfffffff;
	h->a1 = (address >> 23) & 0x1f;
	h->b = (address >> 19) & 0x0f;
	h->d1 = (address >> 10) & 0x1ff;
	h->c1 = 0;
	h->e = 0;
	for (int i = 0; i < 5; i++)
		h->c1 |= ((address >> (2 * i)) & 1) << i;
	for (int i = 0; i < 7; i++)
		h->e |= ((address >> (2 * i + 1)) & 1) << i;
	h->afh = cfg->afh ? 1 : 0;
	h->mod = cfg->afh ? cfg->used_channels : HOP_NCHAN;
	if (h->mod == 0 || h->mod > 80) {
		set_error("hop: AFH pattern with %u used channels", h->mod);
		return BTBBX_E_ARG;
	}
	memcpy(h->bank, cfg->bank, sizeof(h->bank));
	return BTBBX_OK;
}

// Device and pinned buffers of one reversal; recycled through a small pool because a handle is
// opened per piconet and per restart (hipMalloc/hipHostMalloc cost far more than the kernels).
struct HopWorkspace {
	void *d_block;            // one allocation, carved below
	uint32_t *d_cand[2];      // ping-pong lists, HOP_GROUPS entries each
	uint64_t *d_masks;        // HOP_GROUPS / 64 words
	uint32_t *d_prefix;
	uint16_t *d_agree;
	uint32_t *d_hist;         // HOP_MAX_OBS + 1
	HopObs *d_obs;
	WinnowVerdict *d_verdict;
	uint32_t *d_total;
	int device;               // the HIP device the block lives on (a workspace is only reused there)
	void *h_block;            // pinned: observations going in, verdict / total coming back
	HopObs *h_obs;
	WinnowVerdict *h_verdict;
	uint32_t *h_total;
	hipStream_t stream;
};

static std::mutex pool_lock;
static std::vector<HopWorkspace *> pool;
#define HOP_POOL_MAX 8

static void workspace_free(HopWorkspace *w)
{
	if (!w)
		return;
	if (w->d_block)
		(void)hipFree(w->d_block);
	if (w->h_block)
		(void)hipHostFree(w->h_block);
	if (w->stream)
		(void)hipStreamDestroy(w->stream);
	free(w);
}

static HopWorkspace *workspace_get()
{
	int dev = 0;
	(void)hipGetDevice(&dev);
	{
		std::lock_guard<std::mutex> g(pool_lock);
		for (size_t i = 0; i < pool.size(); i++)
			if (pool[i]->device == dev) {
				HopWorkspace *w = pool[i];
				pool.erase(pool.begin() + (long)i);
				return w;
			}
	}
	HopWorkspace *w = (HopWorkspace *)calloc(1, sizeof(*w));
	if (!w)
		return nullptr;
	w->device = dev;
	size_t off = 0;
	auto carve = [&off](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
	const size_t o_c0 = carve(sizeof(uint32_t) * HOP_GROUPS), o_c1 = carve(sizeof(uint32_t) * HOP_GROUPS);
	const size_t o_m = carve(sizeof(uint64_t) * (HOP_GROUPS / 64)), o_p = carve(sizeof(uint32_t) * (HOP_GROUPS / 64));
	const size_t o_a = carve(sizeof(uint16_t) * HOP_GROUPS), o_h = carve(sizeof(uint32_t) * (HOP_MAX_OBS + 1));
	const size_t o_o = carve(sizeof(HopObs) * HOP_MAX_OBS), o_v = carve(sizeof(WinnowVerdict)), o_t = carve(sizeof(uint32_t));
	hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
	if (e == hipSuccess)
		e = hipMalloc(&w->d_block, off);
	if (e == hipSuccess)
		e = hipHostMalloc(&w->h_block, sizeof(HopObs) * HOP_MAX_OBS + 256, hipHostMallocDefault);
	if (e != hipSuccess) {
		hip_fail(e, "hop workspace allocation");
		workspace_free(w);
		return nullptr;
	}
	char *d = (char *)w->d_block, *hp = (char *)w->h_block;
	w->d_cand[0] = (uint32_t *)(d + o_c0);
	w->d_cand[1] = (uint32_t *)(d + o_c1);
	w->d_masks = (uint64_t *)(d + o_m);
	w->d_prefix = (uint32_t *)(d + o_p);
	w->d_agree = (uint16_t *)(d + o_a);
	w->d_hist = (uint32_t *)(d + o_h);
	w->d_obs = (HopObs *)(d + o_o);
	w->d_verdict = (WinnowVerdict *)(d + o_v);
	w->d_total = (uint32_t *)(d + o_t);
	w->h_obs = (HopObs *)hp;
	w->h_verdict = (WinnowVerdict *)(hp + sizeof(HopObs) * HOP_MAX_OBS);
	w->h_total = (uint32_t *)(hp + sizeof(HopObs) * HOP_MAX_OBS + 64);
	return w;
}

void hop_pool_release()        // btbbx_shutdown
{
	std::lock_guard<std::mutex> g(pool_lock);
	for (HopWorkspace *w : pool)
		workspace_free(w);
	pool.clear();
}

static void workspace_put(HopWorkspace *w)
{
	if (!w)
		return;
	{
		std::lock_guard<std::mutex> g(pool_lock);
		if (pool.size() < HOP_POOL_MAX) {
			pool.push_back(w);
			return;
		}
	}
	workspace_free(w);
}

struct btbbx_hop_reversal {
	HopArgs h;
	int aliased;
	uint32_t n;               // candidates
	int cur;                  // which of the two lists is current
	HopWorkspace *w;
};

extern "C" {

void btbbx_hop_cfg_init(btbbx_hop_cfg *cfg, uint32_t address, const uint8_t *afh_map)
{
	memset(cfg, 0, sizeof(*cfg));
	cfg->address = address & 0xfffffff;
	int j = 0;
	for (int i = 0; i < HOP_NCHAN; i++) {                       // precalc, bluetooth_piconet.c:171-194
		const int chan = (2 * i) % HOP_NCHAN;
		if (!afh_map)
			cfg->bank[i] = (uint8_t)chan;
		else if (afh_map[chan / 8] & (1 << (chan % 8)))
			cfg->bank[j++] = (uint8_t)chan;
	}
	if (afh_map) {
		cfg->afh = 1;
		for (int i = 0; i < 10; i++)                        // btbb_piconet_set_afh_map, :122-131
			cfg->used_channels += (uint8_t)__builtin_popcount(afh_map[i]);
	} else {
		cfg->used_channels = HOP_NCHAN;
	}
}

int btbbx_hop_sequence_device(const btbbx_hop_cfg *cfg, uint64_t first, uint64_t count, uint8_t *d_sequence,
			      void *hip_stream)
{
	int rc = hop_device();
	if (rc)
		return rc;
	HopArgs h;
	if ((rc = hop_args(cfg, &h)))
		return rc;
	if ((first & 63) || (count & 63) || first + count > BTBBX_SEQUENCE_LENGTH || !d_sequence ||
	    ((uintptr_t)d_sequence & 15)) {
		set_error("hop_sequence: range [%llu, +%llu) must be 64-aligned inside 2^27, buffer 16-byte aligned",
			  (unsigned long long)first, (unsigned long long)count);
		return BTBBX_E_ARG;
	}
	if (!count)
		return BTBBX_OK;
	const uint32_t nt = (uint32_t)(count >> 6);
	hipLaunchKernelGGL(hop_sequence_kernel, dim3((nt + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, h,
			   (uint32_t)(first >> 6), nt, (uint4 *)d_sequence);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

int btbbx_hop_channels_device(const btbbx_hop_cfg *cfg, const uint32_t *d_clocks, uint32_t n, uint8_t *d_channels,
			      void *hip_stream)
{
	int rc = hop_device();
	if (rc)
		return rc;
	HopArgs h;
	if ((rc = hop_args(cfg, &h)))
		return rc;
	if (!n)
		return BTBBX_OK;
	hipLaunchKernelGGL(hop_channels_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, h, d_clocks, n,
			   d_channels);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

void btbbx_hop_reversal_close(btbbx_hop_reversal *r)
{
	if (!r)
		return;
	workspace_put(r->w);
	free(r);
}

static int reversal_compact(btbbx_hop_reversal *r, uint32_t nwords, const uint32_t *src, uint32_t base, uint32_t *dst)
{
	HopWorkspace *w = r->w;
	hipLaunchKernelGGL(hop_mask_prefix_kernel, dim3(1), dim3(1024), 0, w->stream, w->d_masks, nwords, w->d_prefix,
			   w->d_total);
	hipLaunchKernelGGL(hop_scatter_kernel, dim3((nwords + 255) / 256), dim3(256), 0, w->stream, w->d_masks, w->d_prefix,
			   nwords, src, base, dst);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

btbbx_hop_reversal *btbbx_hop_reversal_open(const btbbx_hop_cfg *cfg, uint32_t clk6, uint8_t channel, int aliased,
					    int *n_candidates)
{
	if (hop_device())
		return nullptr;
	btbbx_hop_reversal *r = (btbbx_hop_reversal *)calloc(1, sizeof(*r));
	if (!r || hop_args(cfg, &r->h) || !(r->w = workspace_get())) {
		free(r);
		return nullptr;
	}
	HopWorkspace *w = r->w;
	r->aliased = aliased != 0;
	hipLaunchKernelGGL(hop_candidate_mask_kernel, dim3(HOP_GROUPS / 256), dim3(256), 0, w->stream, r->h, clk6 & 63u,
			   (int)(int8_t)channel, r->aliased, w->d_masks);
	if (reversal_compact(r, HOP_GROUPS / 64, nullptr, clk6 & 63u, w->d_cand[0]) ||
	    hip_bad(hipMemcpyAsync(w->h_total, w->d_total, sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream), "copy") ||
	    hip_bad(hipStreamSynchronize(w->stream), "hop_reversal_open")) {
		btbbx_hop_reversal_close(r);
		return nullptr;
	}
	r->n = *w->h_total;
	r->cur = 0;
	if (n_candidates)
		*n_candidates = (int)r->n;
	return r;
}

int btbbx_hop_reversal_winnow(btbbx_hop_reversal *r, const int32_t *index_offsets, const uint8_t *channels,
			      uint32_t n_obs, uint32_t *stop, uint32_t *count, uint32_t *cand0)
{
	if (!r || (n_obs && (!index_offsets || !channels)) || n_obs > HOP_MAX_OBS) {
		set_error("hop_reversal_winnow: bad arguments (n_obs %u)", n_obs);
		return BTBBX_E_ARG;
	}
	HopWorkspace *w = r->w;
	WinnowVerdict v = {n_obs, r->n, 0, 0};
	if (n_obs && r->n) {
		for (uint32_t k = 0; k < n_obs; k++) {
			w->h_obs[k].offset = index_offsets[k];
			w->h_obs[k].channel = (int)(int8_t)channels[k];
		}
		const uint32_t n = r->n, nwords = (n + 63) / 64;
		uint32_t *src = w->d_cand[r->cur], *dst = w->d_cand[r->cur ^ 1];
		HIP_TRY(hipMemcpyAsync(w->d_obs, w->h_obs, sizeof(HopObs) * n_obs, hipMemcpyHostToDevice, w->stream));
		if (n <= HOP_SMALL_N) {
			hipLaunchKernelGGL(hop_winnow_small_kernel, dim3(1), dim3(1024), 0, w->stream, r->h, src, n, w->d_obs,
					   n_obs, r->aliased, dst, w->d_verdict);
		} else {
			HIP_TRY(hipMemsetAsync(w->d_hist, 0, sizeof(uint32_t) * (n_obs + 1), w->stream));
			hipLaunchKernelGGL(hop_winnow_kernel, dim3((n + 255) / 256), dim3(256), 0, w->stream, r->h, src, n,
					   w->d_obs, n_obs, r->aliased, w->d_agree, w->d_hist);
			hipLaunchKernelGGL(hop_verdict_kernel, dim3(1), dim3(1024), 0, w->stream, w->d_hist, n, n_obs,
					   w->d_verdict);
			hipLaunchKernelGGL(hop_agree_mask_kernel, dim3((nwords * 64 + 255) / 256), dim3(256), 0, w->stream,
					   w->d_agree, n, w->d_verdict, w->d_masks);
			int rc = reversal_compact(r, nwords, src, 0, dst);
			if (rc)
				return rc;
		}
		hipLaunchKernelGGL(hop_first_kernel, dim3(1), dim3(1), 0, w->stream, dst, w->d_verdict);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(w->h_verdict, w->d_verdict, sizeof(v), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		v = *w->h_verdict;
		r->cur ^= 1;
		r->n = v.count;
	} else if (n_obs) {
		// An empty list (channel never produced by this pattern, e.g. >= 79 or outside the AFH bank):
		// the reference's channel_winnow still runs for the first unused observation, finds no
		// candidate and resets the piconet (bluetooth_piconet.c:596-601, 614-620) -- so the walk
		// stops at observation 0 with nothing left.
		v.stop = 0;
		v.count = 0;
	} else if (r->n) {
		HIP_TRY(hipMemcpyAsync(w->h_total, w->d_cand[r->cur], sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		v.cand0 = *w->h_total;
	}
	if (stop) *stop = v.stop;
	if (count) *count = v.count;
	if (cand0) *cand0 = v.cand0;
	return BTBBX_OK;
}

int64_t btbbx_hop_reversal_candidates(btbbx_hop_reversal *r, uint32_t *dst, uint64_t cap)
{
	if (!r)
		return BTBBX_E_ARG;
	const uint64_t k = r->n < cap ? r->n : cap;
	if (k && dst) {
		HIP_TRY(hipMemcpyAsync(dst, r->w->d_cand[r->cur], k * sizeof(uint32_t), hipMemcpyDeviceToHost, r->w->stream));
		HIP_TRY(hipStreamSynchronize(r->w->stream));
	}
	return (int64_t)r->n;
}

} // extern "C"
