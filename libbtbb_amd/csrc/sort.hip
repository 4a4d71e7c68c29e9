// sort.hip -- (stream, offset) order for hit lists while they are still in HBM, without a host round trip.
//
// The scan kernels append hits unordered; callers of the reference get them in stream order
// (lib/src/bluetooth_packet.c:444-464 is called on a sliding window, first match first).  Rounds 1-2 handed
// the list to rocPRIM's general radix sort between two hipStreamSynchronize() calls.  A hit list is not a
// general sorting problem: keys (stream, offset) are UNIQUE -- an offset matches at most once -- and sparse
// (one hit per thousands of offsets), so the order follows from counting:
//
//   1. extent:   max offset / max stream of the list (the count itself is read from HBM: no `n` from the host)
//   2. buckets:  lin = stream * (max_offset + 1) + offset, bucket = lin >> shift with as many buckets as the
//                list can have records (<= 2^24): histogram, exclusive scan, scatter -- records grouped by bucket
//   3. rank:     inside a bucket the rank of a record is the number of bucket-mates with a smaller key: for the
//                usual handful of mates a loop over them; for a crowded bucket (a stream made of sync words) a
//                presence bitmap of the bucket's 2^shift possible keys in LDS and a prefix popcount -- O(k), and
//                exact because keys are unique (a list with repeated keys, which no scan produces, is still
//                ordered correctly: ties go by position, a crowded bucket with a repeated key by all pairs).
//
// Seven small launches and a memset on the caller's stream, no synchronisation, no vendor library on the path.  The caller
// owns the scratch memory (btbbx_order_hits_scratch_bytes) so that concurrent callers on different streams
// share nothing; btbbx_sort_hits_device keeps its old signature on top of a per-device scratch block.
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>
#include "common.h"

struct __attribute__((aligned(16))) HitRec { uint64_t a, b; };      // a btbbx_hit as an opaque 16-byte value

struct OrderParams {
	unsigned long long max_off, max_stream;    // extent of the list (atomicMax targets)
	unsigned long long mul;                    // max_off + 1
	uint32_t n;                                // records in the list = min(*d_count, cap)
	uint32_t shift;                            // bucket = lin >> shift
	uint32_t ticket;                           // last workgroup of the extent pass works the parameters out
	uint32_t crowded;                          // some bucket has more than ORDER_SMALL members (set by the scan of the counts)
	uint32_t any_shared;                       // order_scatter_kernel saw a record that shares its bucket: order_rank_list_kernel has work
	uint32_t pad;
};

#define ORDER_SMALL 48u                        // bucket-mates up to here are ranked by a plain loop
#define ORDER_MAX_LOG2 24                      // at most 1024 workgroups x 1024 threads x 16 counters per thread in the scans of the counts
                                               // (four per thread up to 2^22 buckets: lists of up to 4 M records; sixteen for longer ones --
                                               // the 4 GiB LAP_ANY list of 6.4 M records: 2^22 buckets 0.63 ms, 2^24 0.37 ms of ordering)
#define ORDER_BIG_BITS 20                      // a crowded bucket's presence bitmap covers 2^20 keys at a time (128 KiB of LDS)
#define ORDER_PAIRS 4096u                      // crowded buckets up to here: every record against every other

__device__ __forceinline__ uint64_t order_lin(const btbbx_hit &h, unsigned long long mul)
{
	return (uint64_t)h.stream * mul + h.offset;
}
__device__ __forceinline__ bool order_less(const btbbx_hit &x, const btbbx_hit &y)
{
	return x.stream != y.stream ? x.stream < y.stream : x.offset < y.offset;
}

// x (at position ix of the grouped list) goes in front of y (at iy): smaller key, or the same key and the earlier
// position -- the scans never report an offset twice, but a caller may hand over any list
__device__ __forceinline__ bool order_before(const btbbx_hit &x, uint32_t ix, const btbbx_hit &y, uint32_t iy)
{
	if (x.stream != y.stream)
		return x.stream < y.stream;
	if (x.offset != y.offset)
		return x.offset < y.offset;
	return ix < iy;
}

__global__ __launch_bounds__(256) void order_extent_kernel(const btbbx_hit *hits, const uint32_t *d_count, uint32_t n_imm,
							  uint32_t cap, uint32_t nb_log2, OrderParams *p)
{
	const uint32_t n = d_count ? min(*d_count, cap) : n_imm;
	unsigned long long off = 0, st = 0;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		off = max(off, (unsigned long long)hits[i].offset);
		st = max(st, (unsigned long long)hits[i].stream);
	}
	for (int d = 32; d; d >>= 1) {
		off = max(off, (unsigned long long)__shfl_xor(off, d));
		st = max(st, (unsigned long long)__shfl_xor(st, d));
	}
	if ((threadIdx.x & 63) == 0) {
		atomicMax(&p->max_off, off);
		atomicMax(&p->max_stream, st);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		if (atomicAdd(&p->ticket, 1u) == gridDim.x - 1) {        // every workgroup's maxima are in: derive the bucket shift
			__threadfence();
			const unsigned long long mo = atomicMax(&p->max_off, 0ull), ms = atomicMax(&p->max_stream, 0ull);
			const unsigned long long mul = mo + 1;
			// total = (ms + 1) * mul keys at most; 2^T >= total
			uint32_t T = 64;
			if (__umul64hi(ms + 1, mul) == 0) {
				const unsigned long long total = (ms + 1) * mul;
				T = total > 1 ? 64 - __builtin_clzll(total - 1) : 0;
			}
			p->mul = mul;
			p->n = n;
			p->shift = T > nb_log2 ? T - nb_log2 : 0;
		}
	}
}

// bucket shift for a key space of n_streams x (max_offset + 1) keys and 2^nb_log2 buckets (host and device)
__host__ __device__ inline uint32_t order_shift(unsigned long long n_streams, unsigned long long mul, uint32_t nb_log2)
{
	uint32_t T = 64;
	const unsigned __int128 total = (unsigned __int128)n_streams * mul;
	if (!(total >> 64)) {
		const unsigned long long t = (unsigned long long)total;
		T = t > 1 ? 64 - __builtin_clzll(t - 1) : 0;
	}
	return T > nb_log2 ? T - nb_log2 : 0;
}

// the caller knows the bounds of the list (stream count and search length of the scan that produced it): one thread
// writes the parameters the extent pass would have derived
__global__ void order_bounds_kernel(const uint32_t *d_count, uint32_t n_imm, uint32_t cap, uint32_t nb_log2, uint32_t n_streams,
				    unsigned long long max_offset, OrderParams *p)
{
	const unsigned long long mul = max_offset + 1;
	p->mul = mul;
	p->n = d_count ? min(*d_count, cap) : n_imm;
	p->shift = order_shift(n_streams ? n_streams : 1, mul, nb_log2);
}

__global__ __launch_bounds__(256) void order_hist_kernel(const btbbx_hit *hits, const OrderParams *p, uint32_t *cnt)
{
	const uint32_t n = p->n, shift = p->shift;
	const unsigned long long mul = p->mul;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
		atomicAdd(&cnt[order_lin(hits[i], mul) >> shift], 1u);
}

// exclusive scan of cnt[0 .. nb) in place, three launches: per-block sums, scan of the sums, per-block scan + base.
// cnt[nb] receives the total.
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *lds_wave, uint32_t &total)
{
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint32_t inc = v;
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(inc, d);
		if (lane >= (uint32_t)d)
			inc += t;
	}
	if (lane == 63)
		lds_wave[wave] = inc;
	__syncthreads();
	uint32_t base = 0, tot = 0;
	for (uint32_t w = 0; w < 16; w++) {
		const uint32_t s = lds_wave[w];
		if (w < wave)
			base += s;
		tot += s;
	}
	__syncthreads();
	total = tot;
	return base + inc - v;
}

// (bounds_streams != 0: the caller knows the list's bounds, and thread 0 of the grid writes the parameters
// order_bounds_kernel would have -- one launch less on the stream)
template <int ORDER_SCAN_ITEMS>
__global__ __launch_bounds__(1024) void order_scan_sums_kernel(const uint32_t *cnt, uint32_t nb, uint32_t *block_sums, OrderParams *p,
							       const uint32_t *d_count, uint32_t n_imm, uint32_t cap, uint32_t nb_log2,
							       uint32_t bounds_streams, unsigned long long max_offset, const uint32_t *gate)
{
	__shared__ uint32_t lds_wave[16];
	if (gate && !*gate)
		return;
	if (bounds_streams && blockIdx.x == 0 && threadIdx.x == 0) {
		const unsigned long long mul = max_offset + 1;
		p->mul = mul;
		p->n = d_count ? min(*d_count, cap) : n_imm;
		p->shift = order_shift(bounds_streams, mul, nb_log2);
	}
	const uint32_t base = blockIdx.x * 1024 * ORDER_SCAN_ITEMS + threadIdx.x * ORDER_SCAN_ITEMS;
	uint32_t v = 0;
	bool big = false;
#pragma unroll
	for (int k = 0; k < ORDER_SCAN_ITEMS; k++)
		if (base + k < nb) {
			const uint32_t c = cnt[base + k];
			v += c;
			big |= c > ORDER_SMALL;
		}
	// order_crowded_kernel has nothing to look for when no counter is that large (the usual case: as many buckets as
	// records): it then skips its pass over all the buckets
	if (__ballot(big) && (threadIdx.x & 63) == 0)
		atomicOr(&p->crowded, 1u);
	uint32_t total;
	(void)block_exclusive_scan_1024(v, lds_wave, total);
	if (threadIdx.x == 0)
		block_sums[blockIdx.x] = total;
}

// every workgroup adds up the sums of the workgroups before it itself (at most 1024 numbers from L2: cheaper than a
// launch for a one-workgroup scan in between)
template <int ORDER_SCAN_ITEMS>
__global__ __launch_bounds__(1024) void order_scan_apply_kernel(uint32_t *cnt, uint32_t nb, const uint32_t *block_sums, const uint32_t *gate)
{
	__shared__ uint32_t lds_wave[16];
	__shared__ uint32_t my_base;
	if (gate && !*gate)
		return;
	{
		const uint32_t v = threadIdx.x < blockIdx.x ? block_sums[threadIdx.x] : 0;      // gridDim.x <= 1024
		uint32_t before;
		(void)block_exclusive_scan_1024(v, lds_wave, before);
		if (threadIdx.x == 0)
			my_base = before;
		__syncthreads();
	}
	const uint32_t base = blockIdx.x * 1024 * ORDER_SCAN_ITEMS + threadIdx.x * ORDER_SCAN_ITEMS;
	uint32_t v[ORDER_SCAN_ITEMS], sum = 0;
#pragma unroll
	for (int k = 0; k < ORDER_SCAN_ITEMS; k++) {
		v[k] = base + k < nb ? cnt[base + k] : 0;
		sum += v[k];
	}
	uint32_t total;
	uint32_t run = block_exclusive_scan_1024(sum, lds_wave, total) + my_base;
#pragma unroll
	for (int k = 0; k < ORDER_SCAN_ITEMS; k++) {
		if (base + k < nb)
			cnt[base + k] = run;
		run += v[k];
	}
	if (base < nb && base + ORDER_SCAN_ITEMS >= nb)      // the thread that holds the last counter writes the total behind it
		cnt[nb] = run;
}

// `final` (may be null): where a record that is alone in its bucket goes instead of `grouped` -- its bucket start IS its
// rank, so when the list was parked somewhere else by the scan (btbbx_scan_ordered_device) the scatter is also the copy
// into place and only the records that share a bucket are looked at again (order_rank_list_kernel)
// `work` (with `final`): per record of the list, where in `grouped` it went if it shares its bucket with up to ORDER_SMALL
// others, else ~0 -- four bytes per record that tell order_rank_list_kernel whom to rank (a dense worklist appended with one
// counter atomic per wave made this kernel 130 us: 20 000 atomics on one address, profiles/r04_paths/chain_kernels_worklist.csv)
__global__ __launch_bounds__(256) void order_scatter_kernel(const btbbx_hit *hits, OrderParams *p, const uint32_t *start,
							    uint32_t *cursor, btbbx_hit *grouped, btbbx_hit *final, uint32_t *work, const uint32_t *gate)
{
	if (gate && !*gate)
		return;
	const uint32_t n = p->n, shift = p->shift;
	const unsigned long long mul = p->mul;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const btbbx_hit h = hits[i];
		const uint32_t b = (uint32_t)(order_lin(h, mul) >> shift);
		const uint32_t s0 = start[b], k = start[b + 1] - s0;      // a bucket of one (more than half of the records) needs no cursor
		const uint32_t pos = k == 1 ? s0 : s0 + atomicAdd(&cursor[b], 1u);
		HitRec *dst = reinterpret_cast<HitRec *>(k == 1 && final ? final : grouped);
		dst[pos] = *reinterpret_cast<const HitRec *>(&h);
		if (work) {
			const bool shared = k > 1 && k <= ORDER_SMALL;
			work[i] = shared ? pos : ~0u;
			if (shared)
				p->any_shared = 1;                      // (plain store of a one by whoever sees it: order_rank_list_kernel's cue)
		}
	}
}

// one thread per record, the ones that share a bucket work: rank among the bucket-mates (they sit in L1 / L2), into place
__global__ __launch_bounds__(256) void order_rank_list_kernel(const btbbx_hit *grouped, const OrderParams *p, const uint32_t *start,
							      const uint32_t *work, btbbx_hit *out, const uint32_t *gate)
{
	if ((gate && !*gate) || !p->any_shared)
		return;                                         // every record alone in its bucket (sparse lists: the usual case)
	const uint32_t n = p->n, shift = p->shift;
	const unsigned long long mul = p->mul;
	for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < n; w += gridDim.x * 256) {
		const uint32_t i = work[w];
		if (i == ~0u)
			continue;                                   // alone in its bucket (in place already) or order_crowded_kernel's
		const btbbx_hit h = grouped[i];
		const uint32_t b = (uint32_t)(order_lin(h, mul) >> shift);
		const uint32_t s = start[b], k = start[b + 1] - s;
		uint32_t rank = 0;
		for (uint32_t j = 0; j < k; j++)
			rank += order_before(grouped[s + j], s + j, h, i) ? 1u : 0u;
		reinterpret_cast<HitRec *>(out)[s + rank] = *reinterpret_cast<const HitRec *>(&h);
	}
}

// records of buckets with up to ORDER_SMALL members: rank = bucket-mates with a smaller key (they sit in L1 / L2)
__global__ __launch_bounds__(256) void order_rank_kernel(const btbbx_hit *grouped, const OrderParams *p, const uint32_t *start,
							 btbbx_hit *out)
{
	const uint32_t n = p->n, shift = p->shift;
	const unsigned long long mul = p->mul;
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const btbbx_hit h = grouped[i];
		const uint32_t b = (uint32_t)(order_lin(h, mul) >> shift);
		const uint32_t s = start[b], k = start[b + 1] - s;
		if (k > ORDER_SMALL)
			continue;                                   // order_crowded_kernel's
		uint32_t rank = 0;
		for (uint32_t j = 0; j < k; j++)
			rank += order_before(grouped[s + j], s + j, h, i) ? 1u : 0u;
		reinterpret_cast<HitRec *>(out)[s + rank] = *reinterpret_cast<const HitRec *>(&h);
	}
}

// buckets with more members than that, one at a time per workgroup: up to ORDER_PAIRS members every record against
// every other; beyond that a presence bitmap of the bucket's keys in LDS (2^20 keys per window), rank = set bits below
// the record's own -- exact because keys are unique, and checked: a bucket whose bitmap holds fewer bits than it has
// members (a repeated key) is redone by all pairs.
// (vblock of vgrid: the workgroup's place in the launch -- or 0 of 1 when order_single_kernel runs the whole ordering in one workgroup)
__device__ void order_crowded_body(const btbbx_hit *grouped, const OrderParams *p, const uint32_t *start, uint32_t nb, btbbx_hit *out,
				   uint32_t vblock, uint32_t vgrid)
{
	extern __shared__ uint32_t bits[];                  // 2^shift bits, then 1024 group prefixes
	__shared__ uint32_t lds_wave[16];
	__shared__ uint32_t found[1024], n_found;
	__shared__ unsigned long long win_lo, win_hi;
	if (!p->crowded)
		return;
	const uint32_t shift = p->shift;
	const unsigned long long mul = p->mul;
	// 1024 buckets are looked at per step, one per thread (a workgroup stepping through the buckets one by one spent
	// 5 ms on 2^22 of them waiting for its own loads); the crowded ones among them are then worked off one at a time
	for (uint32_t first = vblock * 1024; first < nb; first += vgrid * 1024) {
		if (threadIdx.x == 0)
			n_found = 0;
		__syncthreads();
		const uint32_t mine_b = first + threadIdx.x;
		if (mine_b < nb && start[mine_b + 1] - start[mine_b] > ORDER_SMALL)
			found[atomicAdd(&n_found, 1u)] = mine_b;
		__syncthreads();
		const uint32_t todo = n_found;
		for (uint32_t f = 0; f < todo; f++) {
			const uint32_t b = found[f];
			const uint32_t s = start[b], k = start[b + 1] - s;   // uniform over the workgroup
			auto rank_all_pairs = [&]() {
				for (uint32_t i = threadIdx.x; i < k; i += 1024) {
					const btbbx_hit h = grouped[s + i];
					uint32_t rank = 0;
					for (uint32_t j = 0; j < k; j++)
						rank += order_before(grouped[s + j], j, h, i) ? 1u : 0u;
					reinterpret_cast<HitRec *>(out)[s + rank] = *reinterpret_cast<const HitRec *>(&h);
				}
			};
			if (k <= ORDER_PAIRS) {                                  // at most 16 M comparisons: cheaper than clearing 128 KiB
				rank_all_pairs();
				continue;
			}
			// Presence bitmap over ORDER_BIG_BITS key bits at a time: the bucket's 2^shift keys are cut into windows of
			// 2^20, only the windows between the smallest and the largest key present are visited, ranks carry over.
			const uint32_t wbits = shift < ORDER_BIG_BITS ? shift : ORDER_BIG_BITS;
			const uint32_t words = wbits >= 5 ? 1u << (wbits - 5) : 1u;  // bitmap words per window
			const uint32_t per = (words + 1023) / 1024;                  // words per thread group
			uint32_t *group_prefix = bits + words;
			const uint64_t low_mask = shift ? ((1ull << shift) - 1) : 0, win_mask = (1ull << wbits) - 1;
			if (threadIdx.x == 0) {
				win_lo = ~0ull;
				win_hi = 0;
			}
			__syncthreads();
			{
				unsigned long long lo = ~0ull, hi = 0;
				for (uint32_t i = threadIdx.x; i < k; i += 1024) {
					const unsigned long long wdw = (order_lin(grouped[s + i], mul) & low_mask) >> wbits;
					lo = min(lo, wdw);
					hi = max(hi, wdw);
				}
				if (lo <= hi) {
					atomicMin(&win_lo, lo);
					atomicMax(&win_hi, hi);
				}
			}
			__syncthreads();
			const unsigned long long w_first = win_lo, w_last = win_hi;
			if (w_last - w_first > 65535) {                          // dense runs far apart inside one bucket: not worth the windows
				rank_all_pairs();
				__syncthreads();
				continue;
			}
			uint32_t base_rank = 0;
			for (unsigned long long wdw = w_first; wdw <= w_last; wdw++) {
				for (uint32_t w = threadIdx.x; w < words; w += 1024)
					bits[w] = 0;
				__syncthreads();
				for (uint32_t i = threadIdx.x; i < k; i += 1024) {
					const uint64_t low = order_lin(grouped[s + i], mul) & low_mask;
					const uint32_t lw = (uint32_t)(low & win_mask);
					if ((low >> wbits) == wdw)
						atomicOr(&bits[lw >> 5], 1u << (lw & 31));
				}
				__syncthreads();
				uint32_t mine = 0;
				for (uint32_t w = threadIdx.x * per; w < min(words, (threadIdx.x + 1) * per); w++)
					mine += __popc(bits[w]);
				uint32_t total;
				group_prefix[threadIdx.x] = block_exclusive_scan_1024(mine, lds_wave, total);
				__syncthreads();
				if (total) {
					for (uint32_t i = threadIdx.x; i < k; i += 1024) {
						const btbbx_hit h = grouped[s + i];
						const uint64_t low = order_lin(h, mul) & low_mask;
						if ((low >> wbits) != wdw)
							continue;
						const uint32_t lw = (uint32_t)(low & win_mask);
						const uint32_t w = lw >> 5, g = w / per;
						uint32_t rank = base_rank + group_prefix[g] + __popc(bits[w] & ((1u << (lw & 31)) - 1));
						for (uint32_t x = g * per; x < w; x++)
							rank += __popc(bits[x]);
						reinterpret_cast<HitRec *>(out)[s + rank] = *reinterpret_cast<const HitRec *>(&h);
					}
				}
				base_rank += total;
				__syncthreads();
			}
			if (base_rank != k) {                                    // a key occurs twice: the bitmap counted it once -- all pairs
				rank_all_pairs();
				__syncthreads();
			}
		}
		__syncthreads();
	}
}

__global__ __launch_bounds__(1024) void order_crowded_kernel(const btbbx_hit *grouped, const OrderParams *p, const uint32_t *start,
							     uint32_t nb, btbbx_hit *out, const uint32_t *gate)
{
	if (gate && !*gate)
		return;
	order_crowded_body(grouped, p, start, nb, out, blockIdx.x, gridDim.x);
}

// The whole ordering of a parked list in ONE workgroup (round 6): what redoes an ordered scan whose stream the segment slots could
// not rank -- one made of sync words.  It follows the gated re-scan on a side stream beside the compaction, returns at its first
// instruction for every other stream, and is one launch where the general path is six that wait for each other (4.7 us apiece
// even when they have nothing to do: 33 us of the 0.6 ms config-3 chain).  Slow -- one CU -- and exact: zero the counters,
// histogram, scan, scatter, ranks of shared buckets, crowded buckets, the steps of order_launch in its own order.
__global__ __launch_bounds__(1024) void order_single_kernel(const btbbx_hit *list, const uint32_t *d_count, uint32_t cap, uint32_t nb, uint32_t nb_log2,
							    uint32_t n_streams, unsigned long long max_offset, OrderParams *p, uint32_t *start,
							    uint32_t *cursor, btbbx_hit *grouped, btbbx_hit *out, uint32_t *work, uint32_t *count_out,
							    const uint32_t *gate)
{
	__shared__ uint32_t lds_wave[16];
	__shared__ uint32_t s_crowded, s_shared;
	if (gate && !*gate)
		return;
	const uint32_t tid = threadIdx.x;
	const uint32_t n = min(*d_count, cap);
	if (tid == 0)
		*count_out = *d_count;                            // (every match counts, also beyond the capacity: btbbx_scan_device's rule)
	const unsigned long long mul = max_offset + 1;
	const uint32_t shift = order_shift(n_streams, mul, nb_log2);
	for (uint32_t i = tid; i <= nb; i += 1024)
		start[i] = 0;
	for (uint32_t i = tid; i < nb; i += 1024)
		cursor[i] = 0;
	if (tid == 0) {
		p->mul = mul;
		p->n = n;
		p->shift = shift;
		p->crowded = 0;
		p->any_shared = 0;
		s_crowded = s_shared = 0;
	}
	__syncthreads();
	for (uint32_t i = tid; i < n; i += 1024)
		atomicAdd(&start[order_lin(list[i], mul) >> shift], 1u);
	__syncthreads();
	{	// exclusive scan of the counts in place, sixteen per thread and step
		uint32_t carry = 0;
		for (uint32_t first = 0; first < nb; first += 1024 * 16) {
			const uint32_t mine = first + tid * 16;
			uint32_t v[16], sum = 0;
			bool big = false;
#pragma unroll
			for (int k = 0; k < 16; k++) {
				v[k] = mine + k < nb ? __hip_atomic_load(&start[mine + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
				sum += v[k];
				big |= v[k] > ORDER_SMALL;
			}
			if (big)
				s_crowded = 1;
			uint32_t total;
			uint32_t run = carry + block_exclusive_scan_1024(sum, lds_wave, total);
#pragma unroll
			for (int k = 0; k < 16; k++) {
				if (mine + k < nb)
					start[mine + k] = run;
				run += v[k];
			}
			carry += total;
		}
		if (tid == 0)
			start[nb] = carry;
	}
	__syncthreads();
	if (tid == 0)
		p->crowded = s_crowded;
	for (uint32_t i = tid; i < n; i += 1024) {           // scatter (order_scatter_kernel)
		const btbbx_hit h = list[i];
		const uint32_t b = (uint32_t)(order_lin(h, mul) >> shift);
		const uint32_t s0 = start[b], k = start[b + 1] - s0;
		const uint32_t pos = k == 1 ? s0 : s0 + atomicAdd(&cursor[b], 1u);
		HitRec *dst = reinterpret_cast<HitRec *>(k == 1 ? out : grouped);
		dst[pos] = *reinterpret_cast<const HitRec *>(&h);
		const bool shared = k > 1 && k <= ORDER_SMALL;
		work[i] = shared ? pos : ~0u;
		if (shared)
			s_shared = 1;
	}
	__syncthreads();
	if (tid == 0)
		p->any_shared = s_shared;
	if (s_shared) {
		for (uint32_t w = tid; w < n; w += 1024) {       // ranks of bucket-mates (order_rank_list_kernel)
			const uint32_t i = work[w];
			if (i == ~0u)
				continue;
			const btbbx_hit h = grouped[i];
			const uint32_t b = (uint32_t)(order_lin(h, mul) >> shift);
			const uint32_t s0 = start[b], k = start[b + 1] - s0;
			uint32_t rank = 0;
			for (uint32_t j = 0; j < k; j++)
				rank += order_before(grouped[s0 + j], s0 + j, h, i) ? 1u : 0u;
			reinterpret_cast<HitRec *>(out)[s0 + rank] = *reinterpret_cast<const HitRec *>(&h);
		}
	}
	__syncthreads();
	order_crowded_body(grouped, p, start, nb, out, 0, 1);
}

// ---- segment slots: the ordered LAP_ANY scan without a sort (round 6) ------------------------------------------------------
//
// scan_slide_kernel<..., ORD> (scan.hip) leaves every hit in a slot of its SEGMENT -- the 63 words of a tile one wave owns, 4032
// offsets -- at its rank among the segment's hits, and the segment's count in cnt[segment]: plain stores next to each other in
// stream order, which the L2 of the XCD that works on that part of the stream merges into whole lines.  The list in (stream, offset)
// order is then a COMPACTION of the slots: counts -> prefix -> one copy, every read and write coalesced; no bucket atomics in the
// scan, no scatter, no ranking of bucket-mates.  Hits ranked beyond the slots (more than SLOT_N in 4032 offsets) wait in an overflow
// list with (segment, rank) and are put at prefix[segment] + rank by one more launch that usually finds nothing.
#define SLOT_N 2u                              // slots per segment (one: 40 % of the benchmark's list through the overflow list, 81 us more -- profiles/r06_order)
#define SLOT_BLOCK 1024u                       // segments per workgroup of the compaction

struct SlotHeader {
	uint32_t ovf_count;        // records in the overflow list (scan)
	uint32_t irregular;        // the scan could not rank a hit (ScanArgs::irregular): the general path redoes the call
	uint32_t total;            // hits found
	uint32_t redo_count;       // the fallback's re-scan counts its hits here (zeroed with the header)
};

// (zeroing as a kernel: the fallback's counters are only cleared when the fallback runs)
__global__ __launch_bounds__(256) void slot_gated_zero_kernel(uint4 *p, uint64_t n16, uint32_t *count, const uint32_t *gate)
{
	if (gate && !*gate)
		return;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256)
		p[i] = make_uint4(0, 0, 0, 0);
	if (count && blockIdx.x == 0 && threadIdx.x == 0)
		*count = 0;
}

// slots -> list: a workgroup takes SLOT_BLOCK x SLOT_PER segments, a thread SLOT_PER consecutive ones (their counts are one 16-byte
// load).  Two launches: the sums of the workgroups' counts; then every workgroup adds up the sums in front of it itself (about a
// thousand numbers out of L2 for a 4 GiB stream) and copies its segments' records to where they go -- the records of a segment's
// slots lie in rank order.  A segment with hits in the overflow list leaves its start for slot_overflow_kernel.
// (One launch with a look-back between the workgroups was tried first: 870 us against 110 -- a release / acquire at device scope
// writes back / invalidates the L2 of an XCD, and every workgroup did both; profiles/r06_order.)
#define SLOT_PER 8u                            // segments per thread
static_assert(SLOT_PER == 8, "a thread's counts are one uint4");

__device__ __forceinline__ uint32_t slot_counts(const uint16_t *cnt, uint32_t n_segs, uint32_t first, uint32_t (&c)[SLOT_PER])
{
	uint32_t sum = 0;
	if (first + SLOT_PER <= n_segs) {                     // (cnt is 256-byte aligned, first a multiple of 8)
		const uint4 v = *reinterpret_cast<const uint4 *>(cnt + first);
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (int k = 0; k < 4; k++) {
			c[2 * k] = w[k] & 0xffffu;
			c[2 * k + 1] = w[k] >> 16;
		}
	} else {
#pragma unroll
		for (uint32_t k = 0; k < SLOT_PER; k++)
			c[k] = first + k < n_segs ? cnt[first + k] : 0u;
	}
#pragma unroll
	for (uint32_t k = 0; k < SLOT_PER; k++)
		sum += c[k];
	return sum;
}

__global__ __launch_bounds__(1024) void slot_sums_kernel(const uint16_t *cnt, uint32_t n_segs, uint32_t *block_sums)
{
	__shared__ uint32_t lds_wave[16];
	uint32_t c[SLOT_PER], total;
	const uint32_t mine = slot_counts(cnt, n_segs, (blockIdx.x * SLOT_BLOCK + threadIdx.x) * SLOT_PER, c);
	(void)block_exclusive_scan_1024(mine, lds_wave, total);
	if (threadIdx.x == 0)
		block_sums[blockIdx.x] = total;
}

// (a wave takes 64 x SLOT_PER consecutive segments, a lane every 64th of them: counts, slots and output are read and written
// by neighbouring lanes next to each other -- eight consecutive segments per lane instead made this kernel 194 us against 75)
__global__ __launch_bounds__(1024) void slot_place_kernel(const uint16_t *cnt, uint32_t n_segs, const uint32_t *block_sums, SlotHeader *hd,
							  const uint64_t *slots, uint32_t segs_per_stream, uint32_t seg_offsets, uint32_t *seg_start,
							  btbbx_hit *out, uint32_t cap, uint32_t *d_count)
{
	__shared__ uint32_t lds_wave[16];
	uint32_t before = 0, all = 0;                         // sums of the workgroups in front of this one / of all of them
	for (uint32_t first = 0; first < gridDim.x; first += 1024) {
		const uint32_t i = first + threadIdx.x;
		const uint32_t v = i < gridDim.x ? block_sums[i] : 0u;
		uint32_t t_before, t_all;
		(void)block_exclusive_scan_1024(i < blockIdx.x ? v : 0u, lds_wave, t_before);
		(void)block_exclusive_scan_1024(v, lds_wave, t_all);
		before += t_before;
		all += t_all;
	}
	if (hd->irregular)
		return;                                           // the general ordering redoes the call (and owns d_count and the list)
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		hd->total = all;
		*d_count = all;
	}
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const uint32_t wfirst = (blockIdx.x * SLOT_BLOCK + wave * 64) * SLOT_PER;      // the wave's first segment
	uint32_t c[SLOT_PER], mine = 0;
#pragma unroll
	for (uint32_t k = 0; k < SLOT_PER; k++) {
		const uint32_t seg = wfirst + k * 64 + lane;
		c[k] = seg < n_segs ? cnt[seg] : 0u;
		mine += c[k];
	}
	for (int d = 32; d; d >>= 1)
		mine += __shfl_xor(mine, d);
	if (lane == 0)
		lds_wave[wave] = mine;
	__syncthreads();
	uint32_t run = before;
	for (uint32_t w = 0; w < wave; w++)
		run += lds_wave[w];
	if (mine == 0)
		return;                                           // (wave-uniform)
#pragma unroll
	for (uint32_t k = 0; k < SLOT_PER; k++) {
		const uint32_t n = c[k];
		uint32_t inc = n;
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t t = __shfl_up(inc, d);
			if (lane >= (uint32_t)d)
				inc += t;
		}
		const uint32_t pos = run + inc - n;
		run += __shfl(inc, 63);
		if (n == 0)
			continue;
		const uint32_t seg = wfirst + k * 64 + lane;
		const uint64_t *src = slots + (uint64_t)seg * SLOT_N;    // (both slots of the lane's eight segments as 16-byte loads ahead of the prefixes: 94 against 77 us)
		const uint32_t stream = seg / segs_per_stream;        // a slot holds what the segment does not say
		const uint64_t seg_first = (uint64_t)(seg - stream * segs_per_stream) * seg_offsets;
#pragma unroll
		for (uint32_t r = 0; r < SLOT_N; r++)
			if (r < n && pos + r < cap) {
				const uint64_t v = src[r];
				btbbx_hit h;
				h.offset = seg_first + (v & 0xfffu);
				h.lap = (uint32_t)(v >> 12) & 0xffffffu;
				h.ac_errors = (uint8_t)(v >> 36);
				h.reserved = 0;
				h.stream = (uint16_t)stream;
				reinterpret_cast<HitRec *>(out)[pos + r] = *reinterpret_cast<const HitRec *>(&h);
			}
		if (n > SLOT_N)
			seg_start[seg] = pos;
	}
}

__global__ __launch_bounds__(256) void slot_overflow_kernel(const SlotHeader *hd, const HitRec *recs, const uint2 *meta, uint32_t ovf_cap,
							     const uint32_t *seg_start, HitRec *out, uint32_t cap)
{
	if (hd->irregular)
		return;
	const uint32_t n = min(hd->ovf_count, ovf_cap);
	for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const uint2 m = meta[i];
		const uint64_t pos = (uint64_t)seg_start[m.x] + m.y;
		if (pos < cap)
			out[pos] = recs[i];
	}
}

// ---- host side ------------------------------------------------------------------------------------------------------

static uint32_t order_nb_log2(uint32_t cap)
{
	// as many buckets as the list can have records, 2^8 .. 2^ORDER_MAX_LOG2
	uint32_t l = 8;
	while (l < ORDER_MAX_LOG2 && (1u << l) < cap)
		l++;
	return l;
}

// A caller that knows the bounds of the list (stream count, search length) gets one more bit of buckets: the key space of
// n_streams x mul keys seldom fills a power of two (79 streams: 62 % of one), so of 2^(nb_log2 + 1) buckets of half the width
// between half and all are in use -- on the config-3 capture 2 048-bit buckets, narrower than the gap between two packets,
// and no record shares its bucket (with 2^nb_log2 buckets of 4 096 bits 45 % did).  -> buckets in use
static uint32_t order_fine_buckets(uint32_t n_streams, unsigned long long mul, uint32_t nb_log2_fine)
{
	const uint32_t shift = order_shift(n_streams, mul, nb_log2_fine);
	const unsigned __int128 total = (unsigned __int128)n_streams * mul, nbu = (total + (((unsigned __int128)1 << shift) - 1)) >> shift;
	const unsigned __int128 most = (unsigned __int128)1 << nb_log2_fine;
	return (uint32_t)(nbu < most ? (nbu ? nbu : 1) : most);
}

struct OrderLayout { size_t params, start, cursor, sums, grouped, parked, work, total; uint32_t nb_log2, nb; };
// n_streams != 0: the fine buckets (nb_log2 is then one more, nb = the buckets in use); the scratch always has room for them
static OrderLayout order_layout(uint32_t cap, uint32_t n_streams = 0, unsigned long long mul = 0)
{
	OrderLayout L;
	L.nb_log2 = order_nb_log2(cap);
	// (at most 2^ORDER_MAX_LOG2 buckets either way: the scans of the counts run as up to 1024 workgroups of 1024 x ORDER_SCAN_ITEMS counters)
	const size_t nb_most = (size_t)1 << (L.nb_log2 < ORDER_MAX_LOG2 ? L.nb_log2 + 1 : ORDER_MAX_LOG2);
	L.nb = 1u << L.nb_log2;
	if (n_streams && L.nb_log2 < ORDER_MAX_LOG2) {
		L.nb_log2 += 1;
		L.nb = order_fine_buckets(n_streams, mul, L.nb_log2);
	}
	const size_t nb = L.nb;
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	L.params = 0;
	L.start = up(sizeof(OrderParams));
	L.cursor = L.start + up((nb + 1) * 4);           // (parameters, counters and cursors of the buckets in use: one memset)
	L.sums = L.cursor + up(nb * 4);
	L.grouped = L.start + up((nb_most + 1) * 4) + up(nb_most * 4) + up(1024 * 4);
	L.parked = L.grouped + up((size_t)cap * sizeof(btbbx_hit));      // where btbbx_scan_ordered_device's scan leaves its list
	L.work = L.parked + up((size_t)cap * sizeof(btbbx_hit));        // positions of the records that share a bucket
	L.total = L.work + up((size_t)cap * 4);
	return L;
}

extern "C" size_t btbbx_order_hits_scratch_bytes(uint32_t cap)
{
	return order_layout(cap ? cap : 1).total;
}

// counted_by_scan: the list lies in the scratch's `parked` region (the scan wrote it there and counted its buckets);
// d_hits only receives the ordered list
static int order_launch(btbbx_hit *d_hits, const uint32_t *d_count, uint32_t n_imm, uint32_t cap, void *d_scratch,
			size_t scratch_bytes, hipStream_t stream, uint32_t n_streams = 0, uint64_t max_offset = 0,
			bool counted_by_scan = false, const uint32_t *gate = nullptr)
{
	if (cap < 2)
		return BTBBX_OK;
	const OrderLayout L = order_layout(cap, n_streams, (unsigned long long)max_offset + 1);
	if (!d_scratch || scratch_bytes < L.total || ((uintptr_t)d_scratch & 15) || ((uintptr_t)d_hits & 15)) {
		set_error("btbbx_order_hits_device: scratch of %zu bytes (16-byte aligned) needed, %zu given", L.total, scratch_bytes);
		return BTBBX_E_ARG;
	}
	char *base = (char *)d_scratch;
	OrderParams *p = (OrderParams *)(base + L.params);
	uint32_t *start = (uint32_t *)(base + L.start), *cursor = (uint32_t *)(base + L.cursor), *sums = (uint32_t *)(base + L.sums);
	btbbx_hit *grouped = (btbbx_hit *)(base + L.grouped);
	const uint32_t nb = L.nb;
	// parameters, bucket counters and cursors are contiguous: one memset (done by btbbx_scan_ordered_device BEFORE its scan
	// when the scan kernel itself counts the buckets)
	if (!counted_by_scan)
		HIP_TRY(hipMemsetAsync(base, 0, L.sums, stream));
	const uint32_t blocks = (uint32_t)std::min<uint64_t>(((uint64_t)cap + 255) / 256, 2048);
	// the crowded-bucket pass wants 132 KiB of dynamic LDS: asked for once per device, before anything is queued (a part
	// that cannot give it makes the call fail here, with nothing half done)
	const uint32_t crowded_lds = 4u * ((1u << (ORDER_BIG_BITS - 5)) + 1024u);
	{
		static std::atomic<uint64_t> attr_set{0};
		int dev = 0;
		HIP_TRY(hipGetDevice(&dev));
		if (dev >= 64 || !((attr_set.load() >> dev) & 1)) {
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(order_crowded_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
						    crowded_lds));
			if (dev < 64)
				attr_set.fetch_or(1ULL << dev);
		}
	}
	const bool bounds = n_streams != 0;
	if (!bounds)
		hipLaunchKernelGGL(order_extent_kernel, dim3(std::min(blocks, 512u)), dim3(256), 0, stream, d_hits, d_count, n_imm, cap, L.nb_log2, p);
	if (!counted_by_scan) {
		if (bounds)     // (the histogram needs the parameters before the scan of its counts can write them)
			hipLaunchKernelGGL(order_bounds_kernel, dim3(1), dim3(1), 0, stream, d_count, n_imm, cap, L.nb_log2, n_streams,
					   (unsigned long long)max_offset, p);
		hipLaunchKernelGGL(order_hist_kernel, dim3(blocks), dim3(256), 0, stream, d_hits, p, start);
	}
	// BTBBX_ORDER_TIMING=1 (a measuring aid, tools/order_probe.py): HIP events between the launches, printed per call; it
	// synchronises the stream
	static const bool timing = getenv("BTBBX_ORDER_TIMING") != nullptr;
	hipEvent_t ev[8];
	int n_ev = 0;
	auto mark = [&]() {
		if (timing && n_ev < 8 && hipEventCreate(&ev[n_ev]) == hipSuccess)
			(void)hipEventRecord(ev[n_ev++], stream);
	};
	mark();
	const uint32_t scan_items = nb <= (1u << 22) ? 4u : 16u;
	const uint32_t scan_blocks = (nb + 1024 * scan_items - 1) / (1024 * scan_items);
	if (scan_blocks > 1024) {                           // (order_layout keeps nb <= 2^ORDER_MAX_LOG2; order_scan_apply_kernel adds up 1024 sums)
		set_error("btbbx_order_hits_device: %u buckets", nb);
		return BTBBX_E_ARG;
	}
	if (scan_items == 4) {
		hipLaunchKernelGGL(order_scan_sums_kernel<4>, dim3(scan_blocks), dim3(1024), 0, stream, start, nb, sums, p, d_count, n_imm, cap, L.nb_log2,
				   bounds && counted_by_scan ? n_streams : 0u, (unsigned long long)max_offset, gate);
		mark();
		hipLaunchKernelGGL(order_scan_apply_kernel<4>, dim3(scan_blocks), dim3(1024), 0, stream, start, nb, sums, gate);
	} else {
		hipLaunchKernelGGL(order_scan_sums_kernel<16>, dim3(scan_blocks), dim3(1024), 0, stream, start, nb, sums, p, d_count, n_imm, cap, L.nb_log2,
				   bounds && counted_by_scan ? n_streams : 0u, (unsigned long long)max_offset, gate);
		mark();
		hipLaunchKernelGGL(order_scan_apply_kernel<16>, dim3(scan_blocks), dim3(1024), 0, stream, start, nb, sums, gate);
	}
	mark();
	if (counted_by_scan) {
		// parked -> d_hits (records alone in their bucket: in place) / grouped (the others), then the records that share a bucket
		const btbbx_hit *parked = (const btbbx_hit *)(base + L.parked);
		uint32_t *work = (uint32_t *)(base + L.work);
		hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, parked, p, start, cursor, grouped, d_hits, work, gate);
		mark();
		hipLaunchKernelGGL(order_rank_list_kernel, dim3(blocks), dim3(256), 0, stream, grouped, p, start, work, d_hits, gate);
	} else {
		hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, d_hits, p, start, cursor, grouped, (btbbx_hit *)nullptr,
				   (uint32_t *)nullptr, gate);
		hipLaunchKernelGGL(order_rank_kernel, dim3(blocks), dim3(256), 0, stream, grouped, p, start, d_hits);
	}
	mark();
	hipLaunchKernelGGL(order_crowded_kernel, dim3(std::min(nb, 256u)), dim3(1024), crowded_lds, stream, grouped, p, start, nb, d_hits, gate);
	mark();
	if (timing && n_ev) {
		(void)hipEventSynchronize(ev[n_ev - 1]);
		fprintf(stderr, "order_launch (%u buckets):", nb);
		for (int i = 1; i < n_ev; i++) {
			float ms = 0;
			(void)hipEventElapsedTime(&ms, ev[i - 1], ev[i]);
			fprintf(stderr, " %.1f", ms * 1e3f);
		}
		fprintf(stderr, " us (sums, apply, scatter, rank, crowded)\n");
		for (int i = 0; i < n_ev; i++)
			(void)hipEventDestroy(ev[i]);
	}
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

// Device-resident count: sorts the first min(*d_count, cap) records of d_hits in place; nothing comes back to the
// host and nothing is synchronised.  d_scratch: btbbx_order_hits_scratch_bytes(cap) bytes owned by the caller.
extern "C" int btbbx_order_hits_device(btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap, void *d_scratch,
				       size_t scratch_bytes, void *hip_stream)
{
	if (!d_hits || !d_count) {
		set_error("btbbx_order_hits_device: null pointer");
		return BTBBX_E_ARG;
	}
	return order_launch(d_hits, d_count, 0, cap, d_scratch, scratch_bytes, (hipStream_t)hip_stream);
}

// ... when the caller knows what produced the list -- a scan of n_streams streams over search_bits offsets each (the
// arguments of btbbx_scan_device) -- the pass over the list that looks for its largest stream number and offset is
// not needed.  Records outside those bounds are a caller error (they would be binned wrongly).
extern "C" int btbbx_order_scan_hits_device(btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap, uint32_t n_streams,
					    uint64_t search_bits, void *d_scratch, size_t scratch_bytes, void *hip_stream)
{
	if (!d_hits || !d_count || !n_streams || !search_bits) {
		set_error("btbbx_order_scan_hits_device: bad argument");
		return BTBBX_E_ARG;
	}
	return order_launch(d_hits, d_count, 0, cap, d_scratch, scratch_bytes, (hipStream_t)hip_stream, n_streams, search_bits - 1);
}

// Scan and order in one call: the scan kernels count every record they write in its bucket (one more atomic beside the
// record, no pass over the list afterwards), then scan of the counts, scatter, rank as above.  The arguments are
// btbbx_scan_device's plus the ordering scratch; d_hits comes back in (stream, offset) order, *d_count as btbbx_scan_device
// leaves it.  Nothing is synchronised.
int launch_scan(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, uint32_t n_streams, uint64_t search_bits,
		uint32_t lap, int max_ac_errors, btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count,
		unsigned long long *d_first, hipStream_t stream, uint32_t *bucket_cnt, uint64_t bucket_mul, uint32_t bucket_shift, bool msb,
		const ScanSlots *slots, const uint32_t *gate);

// The fallback of the segment slots -- the general path, every launch of it gated on SlotHeader::irregular -- is seven launches that
// return at once for any stream but one made of sync words, 4.7 us each when they queue behind one another: 33 us of a 0.6 ms
// chain.  They run on a side stream instead, forked behind the scan (the flag is final then) and joined at the end of the call, beside
// the compaction, which leaves the list alone when the flag is set: exactly one of the two writes d_hits and *d_count.
struct SideLane { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static std::mutex side_lock;
static std::vector<SideLane> side_pool[BTBBX_MAX_DEVICES];
static bool side_acquire(int dev, SideLane *out)
{
	{
		std::lock_guard<std::mutex> g(side_lock);
		if (!side_pool[dev].empty()) {
			*out = side_pool[dev].back();
			side_pool[dev].pop_back();
			return true;
		}
	}
	SideLane l;
	if (hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking) != hipSuccess)
		return false;
	if (hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) {
		if (l.fork) (void)hipEventDestroy(l.fork);
		(void)hipStreamDestroy(l.s);
		return false;
	}
	*out = l;
	return true;
}
static void side_release(int dev, const SideLane &l)      // (work queued on it stays ordered: the next user queues behind it)
{
	std::lock_guard<std::mutex> g(side_lock);
	side_pool[dev].push_back(l);
}
static void side_pool_release()                            // btbbx_shutdown
{
	std::lock_guard<std::mutex> g(side_lock);
	int home = 0;
	(void)hipGetDevice(&home);
	for (int d = 0; d < BTBBX_MAX_DEVICES; d++) {
		if (side_pool[d].empty())
			continue;
		(void)hipSetDevice(d);
		for (SideLane &l : side_pool[d]) {
			(void)hipStreamSynchronize(l.s);
			(void)hipEventDestroy(l.fork);
			(void)hipEventDestroy(l.join);
			(void)hipStreamDestroy(l.s);
		}
		side_pool[d].clear();
	}
	(void)hipSetDevice(home);
}

// ---- segment slots: layout behind the general ordering's scratch (which the fallback uses) ----
bool scan_slot_geometry(uint64_t search_bits, uint32_t n_streams, uint32_t lap, uint32_t *segs_per_stream, uint64_t *n_segs);

struct SlotLayout { size_t header, cnt, sums, seg_start, slots, ovf_recs, ovf_meta, total; uint32_t n_segs, n_blocks, ovf_cap, segs_per_stream; };
static SlotLayout slot_layout(size_t front, uint32_t segs_per_stream, uint64_t n_segs, uint32_t cap)
{
	auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
	SlotLayout S;
	S.n_segs = (uint32_t)n_segs;
	S.segs_per_stream = segs_per_stream;
	S.n_blocks = (uint32_t)((n_segs + SLOT_BLOCK * SLOT_PER - 1) / (SLOT_BLOCK * SLOT_PER));
	S.ovf_cap = cap;
	S.header = up(front);
	S.cnt = S.header + 256;                                   // (header and counts: one memset)
	S.sums = S.cnt + up(n_segs * sizeof(uint16_t) + 16);
	S.seg_start = S.sums + up(((size_t)S.n_blocks + 1) * 4);
	S.slots = S.seg_start + up(n_segs * 4);
	S.ovf_recs = S.slots + up(n_segs * SLOT_N * sizeof(uint64_t));
	S.ovf_meta = S.ovf_recs + up((size_t)cap * sizeof(btbbx_hit));
	S.total = S.ovf_meta + up((size_t)cap * 8);
	return S;
}

// Scratch of btbbx_scan_ordered_device for a scan of these streams: the general ordering's (btbbx_order_hits_scratch_bytes(cap))
// plus, where the scan has the segment-slot form (LAP_ANY with tables for up to two errors), its slots -- 32 bytes per 4032
// offsets + 24 bytes per record of cap.  A caller that hands over btbbx_order_hits_scratch_bytes(cap) only gets the general path.
extern "C" size_t btbbx_scan_ordered_scratch_bytes(uint64_t search_bits, uint32_t n_streams, uint32_t lap, uint32_t cap)
{
	const size_t front = order_layout(cap ? cap : 1, n_streams, search_bits).total;
	uint32_t sps = 0;
	uint64_t n_segs = 0;
	if (!n_streams || !search_bits || ctx_require() || !scan_slot_geometry(search_bits, n_streams, lap, &sps, &n_segs))
		return front;
	return slot_layout(front, sps, n_segs, cap).total;
}

extern "C" int btbbx_scan_ordered_device_fmt(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, uint32_t n_streams,
					     uint64_t search_bits, uint32_t lap, int max_ac_errors, int format, btbbx_hit *d_hits, uint32_t cap,
					     uint32_t *d_count, void *d_scratch, size_t scratch_bytes, void *hip_stream)
{
	if (!d_words || !d_hits || !d_count || !n_streams || !search_bits || cap < 2 ||
	    (format != BTBBX_FMT_PACKED && format != BTBBX_FMT_PACKED_MSB)) {
		set_error("btbbx_scan_ordered_device: bad argument");
		return BTBBX_E_ARG;
	}
	const OrderLayout L = order_layout(cap, n_streams, search_bits);
	if (!d_scratch || scratch_bytes < L.total || ((uintptr_t)d_scratch & 15)) {
		set_error("btbbx_scan_ordered_device: scratch of %zu bytes (16-byte aligned) needed, %zu given", L.total, scratch_bytes);
		return BTBBX_E_ARG;
	}
	hipStream_t stream = (hipStream_t)hip_stream;
	char *base = (char *)d_scratch;
	const uint32_t shift = order_shift(n_streams, search_bits, L.nb_log2);
	const bool msb = format == BTBBX_FMT_PACKED_MSB;
	const uint32_t *gate = nullptr;
	{	// segment slots where the scan has them and the caller's scratch holds them: scan into the slots, compact
		uint32_t sps = 0;
		uint64_t n_segs = 0;
		if (scan_slot_geometry(search_bits, n_streams, lap, &sps, &n_segs) && ((uintptr_t)d_hits & 15) == 0) {
			const SlotLayout S = slot_layout(L.total, sps, n_segs, cap);
			if (scratch_bytes >= S.total) {
				SlotHeader *hd = (SlotHeader *)(base + S.header);
				HIP_TRY(hipMemsetAsync(base + S.header, 0, S.sums - S.header, stream));
				ScanSlots sl;
				sl.slots = (uint64_t *)(base + S.slots);
				sl.seg_offsets = lap == BTBBX_LAP_ANY ? 4032u : 4096u;
				sl.cnt = (uint16_t *)(base + S.cnt);
				sl.slot_n = SLOT_N;
				sl.segs_per_stream = S.segs_per_stream;
				sl.ovf_recs = (btbbx_hit *)(base + S.ovf_recs);
				sl.ovf_meta = base + S.ovf_meta;
				sl.ovf_cap = S.ovf_cap;
				sl.ovf_count = &hd->ovf_count;
				sl.irregular = &hd->irregular;
				int rc = launch_scan(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors, nullptr, 0, d_count,
						     nullptr, stream, nullptr, 0, 0, msb, &sl, nullptr);
				if (rc)
					return rc;
				// the gated fallback beside the compaction (see SideLane); without a side stream it follows on `stream`
				int dev = 0;
				SideLane lane;
				const bool forked = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < BTBBX_MAX_DEVICES && side_acquire(dev, &lane);
				if (forked) {
					const uint32_t *g = &hd->irregular;
					HIP_TRY(hipEventRecord(lane.fork, stream));
					HIP_TRY(hipStreamWaitEvent(lane.s, lane.fork, 0));
					// (the re-scan appends to the parked list behind its own counter in the header: d_count is the compaction's until the
					// single-workgroup ordering has the list in place)
					rc = launch_scan(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors, (btbbx_hit *)(base + L.parked), cap,
							 &hd->redo_count, nullptr, lane.s, nullptr, 0, 0, msb, nullptr, g);
					if (!rc) {
						const uint32_t crowded_lds = 4u * ((1u << (ORDER_BIG_BITS - 5)) + 1024u);
						HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(order_single_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, crowded_lds));
						hipLaunchKernelGGL(order_single_kernel, dim3(1), dim3(1024), crowded_lds, lane.s, (const btbbx_hit *)(base + L.parked),
								   (const uint32_t *)&hd->redo_count, cap, L.nb, L.nb_log2, n_streams, (unsigned long long)(search_bits - 1),
								   (OrderParams *)(base + L.params), (uint32_t *)(base + L.start), (uint32_t *)(base + L.cursor),
								   (btbbx_hit *)(base + L.grouped), d_hits, (uint32_t *)(base + L.work), d_count, g);
					}
					const hipError_t e_join = hipEventRecord(lane.join, lane.s);
					side_release(dev, lane);
					if (rc)
						return rc;
					HIP_TRY(e_join);
				}
				uint32_t *sums = (uint32_t *)(base + S.sums);
				hipLaunchKernelGGL(slot_sums_kernel, dim3(S.n_blocks), dim3(1024), 0, stream, sl.cnt, S.n_segs, sums);
				hipLaunchKernelGGL(slot_place_kernel, dim3(S.n_blocks), dim3(1024), 0, stream, sl.cnt, S.n_segs, sums, hd,
						   (const uint64_t *)(base + S.slots), S.segs_per_stream, sl.seg_offsets, (uint32_t *)(base + S.seg_start), d_hits, cap,
						   d_count);
				hipLaunchKernelGGL(slot_overflow_kernel, dim3(64), dim3(256), 0, stream, hd, (const HitRec *)(base + S.ovf_recs),
						   (const uint2 *)(base + S.ovf_meta), S.ovf_cap, (const uint32_t *)(base + S.seg_start), (HitRec *)d_hits, cap);
				HIP_TRY(hipGetLastError());
				if (forked) {
					HIP_TRY(hipStreamWaitEvent(stream, lane.join, 0));
					return BTBBX_OK;
				}
				// a hit the scan could not rank (ScanArgs::irregular: streams made of sync words): everything below runs again, this
				// time for real; otherwise each of its launches returns at its first instruction
				gate = &hd->irregular;
			}
		}
	}
	if (gate) {
		hipLaunchKernelGGL(slot_gated_zero_kernel, dim3(512), dim3(256), 0, stream, (uint4 *)base, (uint64_t)(L.sums / 16), d_count, gate);
	} else {
		HIP_TRY(hipMemsetAsync(base, 0, L.sums, stream));
		// the list is built from this call's matches only (they are parked in the scratch, not appended to d_hits): the counter
		// starts at zero whatever the caller left in it -- a stale count would send uninitialised parked records through the ordering
		HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(uint32_t), stream));
	}
	// the scan leaves its records in the scratch (and counts each in its bucket); the ordering puts them into d_hits
	int rc = launch_scan(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors, (btbbx_hit *)(base + L.parked), cap, d_count,
			     nullptr, stream, (uint32_t *)(base + L.start), search_bits, shift, msb, nullptr, gate);
	if (rc)
		return rc;
	return order_launch(d_hits, d_count, 0, cap, d_scratch, scratch_bytes, stream, n_streams, search_bits - 1, true, gate);
}

extern "C" int btbbx_scan_ordered_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, uint32_t n_streams,
					 uint64_t search_bits, uint32_t lap, int max_ac_errors, btbbx_hit *d_hits, uint32_t cap,
					 uint32_t *d_count, void *d_scratch, size_t scratch_bytes, void *hip_stream)
{
	return btbbx_scan_ordered_device_fmt(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors, BTBBX_FMT_PACKED,
					     d_hits, cap, d_count, d_scratch, scratch_bytes, hip_stream);
}

// one scratch block per device for the signature without caller scratch
struct SortScratch {
	std::mutex lock;
	void *block = nullptr;
	size_t bytes = 0;
};
static SortScratch sort_scratch[BTBBX_MAX_DEVICES];

void sort_scratch_release()         // btbbx_shutdown
{
	side_pool_release();
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
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= BTBBX_MAX_DEVICES) {
		set_error("btbbx_sort_hits_device: HIP device ordinal outside the %d contexts this library keeps", BTBBX_MAX_DEVICES);
		return BTBBX_E_ARG;
	}
	SortScratch &sc = sort_scratch[dev];
	std::lock_guard<std::mutex> g(sc.lock);
	const size_t need = order_layout(n).total;
	if (need > sc.bytes) {
		if (sc.block)
			(void)hipFree(sc.block);
		sc.block = nullptr;
		sc.bytes = 0;
		const size_t want = need + need / 2;
		hipError_t e = hipMalloc(&sc.block, want);
		if (e != hipSuccess)
			return hip_fail(e, "hipMalloc(sort scratch)");
		sc.bytes = want;
	}
	int rc = order_launch(d_hits, nullptr, n, n, sc.block, sc.bytes, stream);
	if (rc)
		return rc;
	// the scratch is shared between the callers of this signature: finish before another one may reuse it
	HIP_TRY(hipStreamSynchronize(stream));
	return BTBBX_OK;
}
