// scan.hip -- sliding 64-bit access-code correlator for gfx950 (MI355X).
//
// Replaces the per-symbol loops of promiscuous_packet_search()
// (lib/src/bluetooth_packet.c:368-420) and find_known_lap() (:423-441) behind
// btbb_find_ac() (:444-464).  Input is the PACKED stream (1 bit per symbol); one
// lane owns the 64 bit-offsets that start in one 64-bit word.
//
// LAP_ANY, per lane and word (scan_slide_kernel; tables built for three and four errors: the same kernel in its two-level
// form, see SlideStd / Slide4 below; for five errors: scan_lap_any_kernel, which computes the syndrome from two tables in LDS
// and probes a bitmap in L2 per survivor instead of step 2):
//   1. bit-sliced barker pre-filter: seven funnel-shifted copies of the stream give the
//      7-bit window (LAP MSB + 6 barker bits, :378-385) of all 32 offsets of a dword at
//      once; a carry-save adder counts mismatches against 0x27 and `count in {0,1,6,7}`
//      is BARKER_DISTANCE[window] <= 1.  1/8 of the offsets survive.
//   2. bit-sliced check stream (slide.h): the (64,30) code is cyclic, so ONE sparse parity check slides over the
//      window; 19 consecutive bits of the check stream at a survivor's offset are its candidate index.  A chain of 32
//      offsets is a pair of shift registers (survivor mask, check bits) moved down to the survivor in hand, so the
//      index is the low 19 bits: ONE read of a 2^19-bit set in LDS per survivor (the set = every index a window the
//      reference accepts can have: gen_syndrome :147-159 and the map of :161-185 are linear in the same code; it lies
//      there as bit-reversed 16-bit entries, membership is one fast-rate left shift and one signed compare).  0.30 % pass.
//   3. candidates go straight to a per-wave LDS ring and are verified up to 64 at a time with the
//      exact reference rule: full 34-bit syndrome, open-addressing lookup of the
//      error pattern, popcount <= max_ac_errors, LAP from the corrected word
//      (:396-416).  Results are therefore bit-exact, the set only prunes.
// Known LAP: a bit-sliced mismatch count of the top 16 (12 for max_ac_errors < 2) sync-word bits prunes (0.2 % left for
// max_ac_errors = 2), the survivors get the full popcount(window ^ syncword) of :433.
//
// scan_slide_kernel: two persistent 768-thread workgroups per CU (6 waves per SIMD; 64 KiB set + 12 KiB rings of LDS
// each) stride over tiles of 756 words (63 per wave); scan_known_lap_kernel: 256-thread workgroups, tiles of 512 words.  Pure integer
// work, no MFMA; bound by VALU issue, not by HBM (DESIGN.md 3.1 and 6 say what it is bound by).
#include <stdlib.h>
#include <string.h>
#include "common.h"

#define FULL_MASK 0xffffffffffffffffULL

struct ScanArgs {
	const uint64_t *words;
	uint64_t n_words;        // valid words per stream
	uint64_t pitch_words;    // distance between streams
	uint64_t search_bits;    // offsets [0, search_bits) are tested
	uint64_t tiles_per_stream;
	uint64_t n_tiles;
	uint32_t xcd_tiles;      // LAP_ANY: tiles per XCD share (0 = plain round robin over workgroups)
	uint32_t ring_margin;    // scan_slide_kernel: free ring entries below which the pass loop is left for a drain
	uint32_t full_tiles;     // leading tiles of a stream whose words, halo word and offsets are all in range
	uint32_t n_streams;
	uint32_t msb;            // the words hold their symbols MSB first in every byte (BTBBX_FMT_PACKED_MSB): converted in registers
	uint32_t lap;            // known-LAP mode
	uint64_t syncword;       // known-LAP mode
	int max_err;
	btbbx_hit *hits;
	uint32_t hit_cap;
	uint32_t *hit_count;
	unsigned long long *first;   // first-match mode (atomicMin target) or nullptr
	// btbbx_scan_ordered_device: every record written is also counted in the bucket the ordering (sort.hip) will put it in
	// -- the list then needs no histogram pass -- bucket = (stream * bucket_mul + offset) >> bucket_shift; null = off
	uint32_t *bucket_cnt;
	uint64_t bucket_mul;
	uint32_t bucket_shift;
	// btbbx_scan_ordered_device, LAP_ANY with tables for <= 2 errors (scan_slide_kernel<..., ORD>; the ordering itself: sort.hip
	// "segment slots"): a SEGMENT = the 63 words of a tile one wave owns.  All hits of a segment come out of ONE drain of ONE wave,
	// which ranks them by offset among themselves and stores each in the segment's own slots -- plain stores, no counter, no
	// atomic; hits ranked beyond the slots go to an overflow list with (segment, rank).  null = off.
	uint64_t *seg_slots;         // [segments][seg_slot_n]: offset inside the segment (12 bits) | lap << 12 | ac_errors << 36 -- the segment says the rest
	uint16_t *seg_cnt;           // hits of the segment (all of them, also those in the overflow list); zeroed by the caller
	uint32_t seg_slot_n;
	uint32_t segs_per_stream;    // tiles_per_stream x waves per tile
	btbbx_hit *ovf_recs;         // overflow list: records ...
	uint2 *ovf_meta;             // ... and their (segment, rank)
	uint32_t ovf_cap;
	uint32_t *ovf_count;
	uint32_t *irregular;         // set when a hit left outside a drain (a ring without room: a stream of sync words) or the overflow list is full:
	                             // the caller falls back to the general ordering
	const uint32_t *gate;        // the fallback launch itself: returns at once unless *gate != 0
	ScanTables t;
};

// Debug build (-DSCAN_PROFILE): where the LAP_ANY kernel's wave time goes.  Lane 0 of every wave adds the
// s_memtime ticks since its previous mark to a per-wave counter in LDS (global atomics here would stall the
// very loads the loop waits for); the counters go out once at the end and the launcher prints the table.
//   0 = tile loads + barker filter, 1..13 = survivor pass k, 16 = loop exit, 17 = compaction,
//   18 = exact checks, 19 = wait for the prefetched words
#ifdef SCAN_PROFILE
__device__ unsigned long long g_scan_prof[32];
#define PROF_MARK(k) do { uint64_t now_; __builtin_amdgcn_sched_barrier(0); \
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) : : "memory"); __builtin_amdgcn_sched_barrier(0); \
		if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t *>(prof_off + 4u * (k)), (uint32_t)(now_ - prof_t), \
						      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); prof_t = now_; } while (0)
#define PROF_PIN(x) asm volatile("" : "+v"(x))
#else
#define PROF_MARK(k) do { (void)(k); } while (0)
#define PROF_PIN(x) do { } while (0)
#endif

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
	return __builtin_amdgcn_alignbit(hi, lo, sh);
}

__device__ __forceinline__ void count_bucket(const ScanArgs &a, uint32_t stream, uint64_t offset)
{
	if (a.bucket_cnt)
		atomicAdd(&a.bucket_cnt[((uint64_t)stream * a.bucket_mul + offset) >> a.bucket_shift], 1u);
}

__device__ __forceinline__ void emit_hit(const ScanArgs &a, uint32_t stream, uint64_t offset,
					 uint32_t lap, uint32_t nerr)
{
	if (a.first) {
		unsigned long long v = ((unsigned long long)offset << 32) | ((unsigned long long)(lap & 0xffffff) << 8) | nerr;
		atomicMin(a.first, v);
		return;
	}
	uint32_t idx = atomicAdd(a.hit_count, 1u);
	if (idx < a.hit_cap) {
		btbbx_hit h;
		h.offset = offset;
		h.lap = lap;
		h.ac_errors = (uint8_t)nerr;
		h.reserved = 0;
		h.stream = (uint16_t)stream;
		a.hits[idx] = h;
		count_bucket(a, stream, offset);
	}
}

// (Stream words are read once; loading them non-temporally so that they do not push the L2-resident tables of the
// >= 4-error kernels out of the cache changed nothing: 12.0 against 12.06 ms per GiB at five errors, round 3; round 6, the two-level
// form for four errors: 7 % fewer fabric reads, 3 % slower -- profiles/r06_init4.)
__device__ __forceinline__ uint64_t stream_ld(const uint64_t *p)
{
	return *p;
}
// MSB-first bytes (first received symbol in bit 7, the order a radio front end delivers) -> the library's LSB-first dword:
// reverse the dword's 32 bits, put the four bytes back in order (v_bfrev_b32 + v_perm_b32).  The scan kernels do this to the
// four dwords of a lane behind a wave-uniform branch -- in the filter phase, which runs in the other waves' gaps -- instead of
// a conversion pass over the capture in HBM (4 GiB read + 4 GiB written before a 4 GiB scan).
__device__ __forceinline__ uint32_t msb_dword(uint32_t x)
{
	return __builtin_bswap32(__brev(x));
}
__device__ __forceinline__ uint64_t load_word(const uint64_t *base, uint64_t j, uint64_t n_words)
{
	return j < n_words ? stream_ld(base + j) : 0ULL;
}

// The exact acceptance rule of promiscuous_packet_search for one offset that passed the
// barker filter (bluetooth_packet.c:387-416).
// The kernel has no static __shared__, so the dynamic LDS allocation starts at LDS byte 0
// and table addresses are plain byte offsets: every DS access below is `base + offset:imm`
// with the table base folded into the 16-bit immediate.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_ld(uint32_t byte_off) { return *reinterpret_cast<lds_u32_t *>(byte_off); }
__device__ __forceinline__ void lds_st(uint32_t byte_off, uint32_t v) { *reinterpret_cast<lds_u32_t *>(byte_off) = v; }
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
__device__ __forceinline__ uint32_t lds_ld16(uint32_t byte_off) { return *reinterpret_cast<lds_u16_t *>(byte_off); }
// lanes whose 16-bit entry x, shifted LEFT by sh & 15, is negative as a 16-bit number: the fast-rate left shift (see
// scan_slide_kernel) and the 16-bit compare, both as written here (from C the compiler widens the test to v_bfe_u32 + v_cmp_ne_u32)
__device__ __forceinline__ uint64_t sign16_after_shl(uint32_t x, uint32_t sh)
{
	uint32_t r;
	uint64_t m;
	asm("v_lshlrev_b16 %1, %2, %3\n\tv_cmp_gt_i16_e64 %0, 0, %1" : "=s"(m), "=&v"(r) : "v"(sh), "v"(x));
	return m;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
__device__ __forceinline__ u32x4 lds_ld4(uint32_t byte_off) { return *reinterpret_cast<lds_u32x4_t *>(byte_off); }
__device__ __forceinline__ void lds_st4(uint32_t byte_off, u32x4 v) { *reinterpret_cast<lds_u32x4_t *>(byte_off) = v; }
// a ring record as four dword stores (the compiler pairs them into two ds_write2_b32): its dwords come from registers that
// are not neighbours, and one 16-byte store first copies them into four that are -- vector instructions of a one-lane event
__device__ __forceinline__ void lds_st_rec(uint32_t byte_off, u32x4 v)
{
	lds_st(byte_off, v.x);
	lds_st(byte_off + 4u, v.y);
	lds_st(byte_off + 8u, v.z);
	lds_st(byte_off + 12u, v.w);
}

// w = the 64-symbol window at `offset` (the kernel keeps it with the candidate: by the time a
// batch is verified the stream words have long left the L2, and re-reading them cost 40 % extra
// HBM traffic).
template <bool LDS_TABLES = true>
__device__ __forceinline__ bool verify_lap_any(const ScanArgs &a, uint64_t w, uint32_t &lap, uint32_t &nerr_out)
{
	uint32_t win = (uint32_t)(w >> 57);
	uint32_t cls = __popc(win ^ BARKER1) <= 1 ? 1u : 0u;
	uint64_t sw = (w & LOW57) | ((uint64_t)(cls ? BARKER1 : BARKER0) << 57);
	// syndrome of (sw ^ pn): linear in the low 57 window bits plus a class constant.  The low 32
	// bits come from the LDS tables exactly as in the probe; bits 32 and 33 are two parities.  (The
	// byte tables in global memory cost eight divergent loads per candidate, which is what bounded
	// the scan for tables built for three or more errors.)
	const uint64_t low = w & LOW57;
	const uint32_t ia = (uint32_t)(low >> TABA_FIRST) & ((1u << TABA_BITS) - 1), ib = (uint32_t)(low >> (TABA_FIRST + TABA_BITS));
	const uint32_t s_lo = (uint32_t)low ^ (LDS_TABLES ? lds_ld(LDS_OFF_TABA + (ia << 2)) : a.t.tabA[ia])
			      ^ (LDS_TABLES ? lds_ld(LDS_OFF_TABB + (ib << 2)) : a.t.tabB[ib]) ^ (cls ? a.t.kdiff : 0u);
	const uint32_t s_hi = ((uint32_t)(a.t.kclass[cls] >> 32) ^ (__popcll(low & a.t.hi_mask[0]) & 1)
			       ^ ((__popcll(low & a.t.hi_mask[1]) & 1) << 1)) & 3;
	const uint64_t syn = ((uint64_t)s_hi << 32) | s_lo;
	if (a.t.bitmap2) {
		const uint32_t i2 = (s_lo * 0x9E3779B1u) >> a.t.bitmap2_shift;
		if (!((a.t.bitmap2[i2 >> 5] >> (i2 & 31)) & 1))
			return false;
	}
	uint32_t nerr = 0;
	if (syn) {
		uint64_t h = ((((uint32_t)syn ^ (uint32_t)(syn >> 32)) * 0x9E3779B1u) >> (32 - __popcll(a.t.hmask))) & a.t.hmask;
		for (;;) {
			uint64_t slot = a.t.hslots[h];
			if (slot == HSLOT_EMPTY)
				return false;                         // no pattern -> ac_errors = 0xff -> reject
			if ((slot & 0x3ffffffffULL) == syn) {
				uint64_t err = 0;
#pragma unroll
				for (int i = 0; i < 5; i++) {
					uint32_t pos = (uint32_t)(slot >> (34 + 6 * i)) & 63;
					if (pos != 63)
						err |= 1ULL << pos;
				}
				sw ^= err;
				nerr = __popcll(err);
				break;
			}
			h = (h + 1) & a.t.hmask;
		}
	}
	lap = (uint32_t)(sw >> 34) & 0xffffff;
	nerr_out = nerr;
	return (int)nerr <= a.max_err;
}

// ---- LAP_ANY ----------------------------------------------------------------------------

// index of the lowest set bit; 0xffffffff for 0 (v_ffbl_b32), which the callers use as
// "offset 31 of a lane that has nothing left" -- its result is masked out afterwards
__device__ __forceinline__ uint32_t lowest_bit(uint32_t m)
{
	uint32_t p;
	asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(m));
	return p;
}

// three-input boolean in one full-rate instruction (truth table index = a*4 + b*2 + c)
#define BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return BITOP3(a, b, c, 0x96); }

// Barker pre-filter for the 32 offsets whose 7-bit window (LAP MSB + 6 barker bits,
// bluetooth_packet.c:378-385) lives in dh:dm: bit k of the window at offset p is bit
// (p + 25 + k) of dh:dm.  Counts mismatches against BARKER1 = 0b0100111 with a carry-save
// adder of v_bitop3 full adders (inverted planes folded into the truth tables):
//   count in {0,1} -> BARKER_DISTANCE <= 1, corrected to BARKER1 (class 1)
//   count in {6,7} -> BARKER_DISTANCE <= 1, corrected to BARKER0 (class 0)
__device__ __forceinline__ void barker32(uint32_t dm, uint32_t dh, uint32_t valid, uint32_t &pass, uint32_t &cls)
{
	const uint32_t s0 = alignbit(dh, dm, 25), s1 = alignbit(dh, dm, 26), s2 = alignbit(dh, dm, 27);
	const uint32_t s3 = alignbit(dh, dm, 28), s4 = alignbit(dh, dm, 29), s5 = alignbit(dh, dm, 30);
	const uint32_t s6 = alignbit(dh, dm, 31);
	// mismatch planes: m0 = ~s0, m1 = ~s1, m2 = ~s2, m3 = s3, m4 = s4, m5 = ~s5, m6 = s6
	const uint32_t a = BITOP3(s0, s1, s2, 0x69);       // m0 ^ m1 ^ m2
	const uint32_t ca = BITOP3(s0, s1, s2, 0x17);      // maj(m0, m1, m2)
	const uint32_t b = BITOP3(s3, s4, s5, 0x69);       // m3 ^ m4 ^ m5
	const uint32_t cb = BITOP3(s3, s4, s5, 0xd4);      // maj(s3, s4, ~s5)
	const uint32_t cc = BITOP3(a, b, s6, 0xe8);        // carry of the ones column
	// count = ones + 2 (ca + cb + cc): it is 0 or 1 iff the three carries are all clear, 6 or 7 iff they are all set -- one
	// "all three equal" instead of the twos and fours planes and their comparison (third session of round 6: seven instead of
	// eight three-input instructions per 32 offsets)
	pass = BITOP3(ca, cb, cc, 0x81) & valid;
	cls = BITOP3(ca, cb, cc, 0x01);                    // all clear: count in {0, 1}
}

// (scan_lap_any_kernel, tables for five errors)  One survivor costs about 17 VALU + 2 DS instructions:
//   syndrome_low32 = w[31:0] ^ tabA[w[44:34]] ^ tabB[w[56:45]] ^ (class ? kdiff : 0)
// for the window w at offset p of the dword triple (e0,e1,e2), then a probe of the second-level
// bitmap in L2 with a hash of it.  The stages are separate functions so that the survivor loop
// can issue the LDS reads of its two chains back to back, each under the exec mask of the
// lanes that really have a survivor: the DS pipe (shared by the 16 waves of the CU) then
// only pays bank conflicts for useful lanes.
// Instruction choice follows tools/valu_rate.hip: two-operand logic/shift ops and v_bitop3
// issue at full rate on gfx950, v_bfe/v_alignbit/v_lshl_add/v_and_or at half rate.
struct Probe { uint32_t x, offA, offB; };

__device__ __forceinline__ Probe probe_addr(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t cls,
					    uint32_t kdiff, uint32_t p)
{
	Probe r;
	const uint32_t wlo = alignbit(e1, e0, p);
	const uint32_t whi = alignbit(e2, e1, p);
	// whi = window bits 32..63: bits 34..44 sit at 2..12 -- already the byte offset of a u32 entry
	r.offA = whi & (((1u << TABA_BITS) - 1) << 2);
	r.offB = (whi >> (TABA_FIRST - 32 + TABA_BITS - 2)) & (((1u << TABB_BITS) - 1) << 2);
	const uint32_t cmask = (uint32_t)__builtin_amdgcn_sbfe(cls, p, 1);     // 0 or ~0
	r.x = BITOP3(cmask, kdiff, wlo, 0x6a);                                 // wlo ^ (cmask & kdiff)
	return r;
}

// Wave priorities (s_setprio) by phase of a trip.  The four waves of a SIMD otherwise run in step -- all in the
// VALU-dense pre-filter, then all waiting on LDS round trips in the survivor loop -- and compete for the same unit.
// With the pre-filter lowest, the loop above it and the candidate handling (the longest latencies: LDS batches,
// global probes, hit stores) highest, a wave in a latency-bound phase issues as soon as it can and the pre-filter of
// the others fills the gaps: 4.55 -> 4.18 ms.  Measured (filter / loop / candidates): 0/2/3 and 0/1/3 4.17-4.18,
// 1/2/3 4.20, 3/1/0 4.23, 2/0/3 4.24, 1/0/1 and 0/3/3 4.29, 1/0/0 4.42; a fixed priority per wave (no phases): 4.54-4.58.
#ifndef PRIO_FILTER
#define PRIO_FILTER 0
#define PRIO_LOOP 2
#define PRIO_CAND 3
#endif
struct SlideTapList { int n; int k[32]; };
template <uint64_t TAPS>
constexpr SlideTapList slide_tap_list()
{
	SlideTapList l = {0, {0}};
	for (int k = 0; k < 64; k++)
		if ((TAPS >> k) & 1)
			l.k[l.n++] = k;
	return l;
}
// 32 positions of the sliding check stream (slide.h): bit b = parity of the stream bits b + k over the taps k,
// stream bit i = bit i of e2:e1:e0.  Taps and shifts are compile-time constants (a funnel shift by a
// run-time amount costs more, see 3.2 of NOTEBOOK.md).
template <uint64_t TAPS>
__device__ __forceinline__ uint32_t slide32(uint32_t e0, uint32_t e1, uint32_t e2)
{
	constexpr SlideTapList taps = slide_tap_list<TAPS>();
	uint32_t plane[32];
#pragma unroll
	for (int i = 0; i < taps.n; i++) {
		const int k = taps.k[i];
		if (k == 0)
			plane[i] = e0;
		else if (k < 32)
			plane[i] = alignbit(e1, e0, k);
		else if (k == 32)
			plane[i] = e1;
		else
			plane[i] = alignbit(e2, e1, k - 32);
	}
	uint32_t acc = plane[0];
#pragma unroll
	for (int i = 1; i + 1 < taps.n; i += 2)
		acc = xor3(acc, plane[i], plane[i + 1]);
	if ((taps.n & 1) == 0)
		acc ^= plane[taps.n - 1];
	return acc;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_lap_any_kernel(ScanArgs a)
{
	extern __shared__ uint32_t lds[];

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = tid >> 6;
	const uint32_t slot_off = LDS_OFF_PARK + CAND_BYTES * (wave * 64 * PARK_SLOTS + lane * PARK_SLOTS);
	const uint32_t ring_off = LDS_OFF_QUEUE + CAND_BYTES * wave * QRING;
	uint32_t kdiff = a.t.kdiff;
	asm volatile("" : "+v"(kdiff));           // keep it in a VGPR: a VALU op with an SGPR source issues at half rate

	// Tile order.  The dispatcher is observed to place workgroup b on XCD b % 8 (not a contract: a
	// different placement costs L2 sharing, never correctness).  Each XCD gets one contiguous
	// eighth of the tiles and its 32 workgroups walk it interleaved, so that the halo word of a
	// tile -- the first word of the next tile -- is found in the L2 the neighbour workgroup just
	// filled instead of being fetched from HBM a second time by another XCD.
	uint32_t first_tile = blockIdx.x, tile_step = gridDim.x, n_mine;
	if (a.xcd_tiles) {
		const uint32_t xcd = blockIdx.x & 7, lo_t = xcd * a.xcd_tiles;
		const uint32_t hi_t = min((uint64_t)lo_t + a.xcd_tiles, a.n_tiles);
		tile_step = gridDim.x >> 3;
		first_tile = lo_t + (blockIdx.x >> 3);
		n_mine = first_tile < hi_t ? (hi_t - first_tile + tile_step - 1) / tile_step : 0;
	} else {
		n_mine = first_tile < a.n_tiles ? (uint32_t)((a.n_tiles - first_tile + tile_step - 1) / tile_step) : 0;
	}

	// tables -> LDS, 16 bytes per lane per step, coalesced
	{
		char *ldsb = reinterpret_cast<char *>(lds);
		const uint4 *srcA = reinterpret_cast<const uint4 *>(a.t.tabA);
		const uint4 *srcB = reinterpret_cast<const uint4 *>(a.t.tabB);
		uint4 *dA = reinterpret_cast<uint4 *>(ldsb + LDS_OFF_TABA);
		uint4 *dB = reinterpret_cast<uint4 *>(ldsb + LDS_OFF_TABB);
		for (uint32_t i = tid; i < LDS_TABA_WORDS / 4; i += SCAN_THREADS) dA[i] = srcA[i];
		for (uint32_t i = tid; i < LDS_TABB_WORDS / 4; i += SCAN_THREADS) dB[i] = srcB[i];
	}
	__syncthreads();
#ifdef SCAN_PROFILE
	const uint32_t prof_off = LDS_OFF_PROF + 128u * (tid >> 6);
	if ((tid & 63) < 32)
		lds_st(prof_off + 4u * (tid & 63), 0u);
	uint64_t prof_t;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t) : : "memory");
#endif

	// Candidate = passed the bitmap in L2 (a quarter of the survivors with tables for five errors).  Three stages:
	//  1. park: one DS write into a private slot of the lane -- no atomics and no ballots in
	//     the survivor loop (a lane with all slots full verifies in place: adversarial input);
	//  2. compact: at a tile end, once enough lanes hold one, the parked codes are packed
	//     into the wave's ring with ballot + mbcnt;
	//  3. verify: the exact reference rule, 64 ring entries at a time (full wave, and the
	//     compiler merges the 64 hit-counter atomics into one).
	// code = (tile iteration << 12) | (lane that owns the word << 6) | offset in the word
	uint32_t n_parked = 0;
	uint32_t q_head = 0, q_tail = 0;          // wave-uniform ring cursors (free running)
	auto code_word = [&](uint32_t code, uint32_t &stream) {
		// `it` -> tile.  The launcher keeps tile numbers below 2^32 (iterations < 2^20, grid <= CUs),
		// so this is one 32-bit division and only for multi-stream launches -- the 64-bit div + mod
		// that used to sit here cost about 2000 cycles per batch of 64 candidates.
		const uint32_t tile = first_tile + (code >> 12) * tile_step;
		uint32_t t = tile;
		stream = 0;
		if (a.n_streams > 1) {
			stream = tile / (uint32_t)a.tiles_per_stream;
			t = tile - stream * (uint32_t)a.tiles_per_stream;
		}
		return (uint64_t)t * SCAN_THREADS + wave * 64 + ((code >> 6) & 63);
	};
	// Hits of a verified batch are not written one batch at a time: a single counter word in
	// global memory takes ~140 M atomics/s, which capped the scan as soon as batches became
	// frequent (tables for >= 3 errors: 750 k batches per GiB).  Each wave keeps up to 64 pending
	// hit records in registers (one per lane), appends new ones with ds_permute (a lane-to-lane
	// push through the LDS crossbar, no LDS memory), and reserves + writes 64 at a time.
	uint32_t pend = 0;                            // wave-uniform
	uint32_t h_off = 0, h_hi = 0, h_lap = 0;      // lane k < pend: offset low, offset high | stream << 16, lap << 8 | errors
	auto flush_hits = [&]() {
		if (pend == 0)
			return;
		uint32_t base = 0;
		if (lane == 0)
			base = atomicAdd(a.hit_count, pend);
		base = __builtin_amdgcn_readfirstlane(base);
		const uint32_t idx = base + lane;
		if (lane < pend && idx < a.hit_cap) {
			uint4 rec;
			rec.x = h_off;
			rec.y = h_hi & 0xffff;
			rec.z = h_lap >> 8;
			rec.w = (h_lap & 0xff) | (h_hi & 0xffff0000u);
			reinterpret_cast<uint4 *>(a.hits)[idx] = rec;
			count_bucket(a, h_hi >> 16, ((uint64_t)(h_hi & 0xffff) << 32) | h_off);
		}
		pend = 0;
	};
	auto push_hits = [&](bool hit, uint32_t stream, uint64_t offset, uint32_t lap, uint32_t nerr) {
		if (a.first) {                            // first-match mode: atomicMin, hits go out one by one
			if (hit)
				emit_hit(a, stream, offset, lap, nerr);
			return;
		}
		const uint64_t m = __ballot(hit);
		if (!m)
			return;
		const uint32_t c = (uint32_t)__popcll(m);
		if (pend + c > 64)
			flush_hits();
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
		// lanes without a hit push to a lane outside [pend, pend + c), whose result is ignored
		const int dst = (int)((hit ? pend + rank : (pend ? 0u : c)) << 2);
		const uint32_t r_off = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)(uint32_t)offset);
		const uint32_t r_hi = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)((uint32_t)(offset >> 32) | (stream << 16)));
		const uint32_t r_lap = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)((lap << 8) | nerr));
		if (lane - pend < c) {
			h_off = r_off;
			h_hi = r_hi;
			h_lap = r_lap;
		}
		pend += c;
	};
	auto park = [&](uint32_t code, uint32_t wlo, uint32_t whi) {
		if (n_parked < PARK_SLOTS) {
			const u32x4 rec = {code, wlo, whi, 0u};
			lds_st4(slot_off + CAND_BYTES * n_parked, rec);
			n_parked++;
		} else {                                  // all slots taken (adversarial input): verify in place
			uint32_t stream, lap, nerr;
			const uint64_t word = code_word(code, stream);
			if (verify_lap_any(a, ((uint64_t)whi << 32) | wlo, lap, nerr))
				emit_hit(a, stream, word * 64 + (code & 63), lap, nerr);
		}
	};
	auto drain = [&](uint32_t n) {
		PROF_MARK(17);
		bool hit = false;
		uint32_t stream = 0, lap = 0, nerr = 0;
		uint64_t offset = 0;
		if (lane < n) {
			const u32x4 rec = lds_ld4(ring_off + CAND_BYTES * ((q_head + lane) & (QRING - 1)));
			const uint32_t code = rec.x;
			const uint64_t w = ((uint64_t)rec.z << 32) | rec.y;
			offset = code_word(code, stream) * 64 + (code & 63);
			hit = verify_lap_any(a, w, lap, nerr);
		}
		push_hits(hit, stream, offset, lap, nerr);
		q_head += n;
		PROF_MARK(18);
	};
	auto compact = [&](bool final) {
		for (uint32_t k = 0; k < PARK_SLOTS; k++) {
			const uint64_t have = __ballot(n_parked > k);
			if (!have)
				break;
			if (n_parked > k) {
				const uint32_t slot = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(have >> 32),
						__builtin_amdgcn_mbcnt_lo((uint32_t)have, 0));
				const uint32_t from = slot_off + CAND_BYTES * k, to = ring_off + CAND_BYTES * (slot & (QRING - 1));
				lds_st4(to, lds_ld4(from));
			}
			q_tail += __popcll(have);
			while (q_tail - q_head >= 64)       // keeps the ring below 128 entries
				drain(64);
		}
		n_parked = 0;
		if (final && q_tail != q_head)
			drain(q_tail - q_head);
	};

	// tile cursor without divisions: uniform (stream, tile-in-stream) stepped per tile
	// (32-bit: the launcher refuses launches with 2^32 tiles or more, and 64-bit compares of wave-uniform
	// values run on the VALU -- the SALU has none)
	struct Cursor { uint32_t stream; uint32_t t; };
	const uint32_t tiles_per_stream = (uint32_t)a.tiles_per_stream;
	Cursor cur = {a.n_streams, 0};               // stream == n_streams: nothing (left) to do
	uint32_t handed = 0;                          // tiles handed out so far
	if (n_mine) {
		cur.stream = a.n_streams > 1 ? first_tile / tiles_per_stream : 0;
		cur.t = first_tile - cur.stream * tiles_per_stream;
	}
	auto advance = [&](Cursor &c) {
		if (++handed >= n_mine) {
			c.stream = a.n_streams;
			return;
		}
		// (no wrap: the launcher keeps the tile count below 2^20 x grid size)
		c.t += tile_step;
		while (c.t >= tiles_per_stream && c.stream < a.n_streams) {
			c.t -= tiles_per_stream;
			c.stream++;
		}
	};
	// a tile whose 1024 words + halo word and 65536 offsets are all in range needs no masks
	auto tile_full = [&](uint32_t tt) {                         // (one scalar compare; the launcher did the 64-bit arithmetic)
		return tt < a.full_tiles;
	};
	// front set (tables for five errors, round 6): window positions of the second check stream behind offset 63 reach 23 bits into
	// the word after next -- its low dword comes along
	const bool front = a.t.slide4b_bitmap != nullptr;          // launch-uniform
	auto load_pair = [&](const Cursor &c, uint64_t &lo, uint64_t &hi, uint32_t &far) {
		lo = hi = 0;
		far = 0;
		if (c.stream >= a.n_streams)
			return;
		const uint64_t *tp = a.words + (uint64_t)c.stream * a.pitch_words + (uint64_t)c.t * SCAN_THREADS;   // uniform
		if (tile_full(c.t)) {
			lo = stream_ld(tp + tid);
			hi = stream_ld(tp + tid + 1);
		} else {
			const uint64_t w = (uint64_t)c.t * SCAN_THREADS + tid;
			lo = w < a.n_words ? stream_ld(tp + tid) : 0;
			hi = w + 1 < a.n_words ? stream_ld(tp + tid + 1) : 0;
		}
		if (front) {
			const uint64_t w = (uint64_t)c.t * SCAN_THREADS + tid;
			far = w + 2 < a.n_words ? *reinterpret_cast<const uint32_t *>(tp + tid + 2) : 0u;
		}
	};

	// Each trip of the main loop works on UNROLL tiles at once (independent words in the same
	// lane): with one workgroup of 16 waves per CU (the tables fill the LDS) this is what keeps
	// enough independent LDS chains in flight to cover the DS latency.
	constexpr int UNROLL = SCAN_UNROLL;
	Cursor tc[UNROLL];
	uint64_t lo[UNROLL], hi[UNROLL];
	uint32_t far[UNROLL];
#pragma unroll
	for (int u = 0; u < UNROLL; u++) {
		tc[u] = cur;
		load_pair(cur, lo[u], hi[u], far[u]);
		advance(cur);
	}

	for (uint32_t it = 0; tc[0].stream < a.n_streams; it += UNROLL) {
		// software prefetch of the next tiles: the loads fly while these are processed
		Cursor nc[UNROLL];
		uint64_t nlo[UNROLL], nhi[UNROLL];
		uint32_t nfar[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			nc[u] = cur;
			load_pair(cur, nlo[u], nhi[u], nfar[u]);
			advance(cur);
		}

		uint32_t d[UNROLL][4], m[UNROLL][2], cls[UNROLL][2];
		uint32_t c2[UNROLL][3] = {};                             // the second check stream (front set), positions 0 .. 95 of the lane's word
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			d[u][0] = (uint32_t)lo[u]; d[u][1] = (uint32_t)(lo[u] >> 32);
			d[u][2] = (uint32_t)hi[u]; d[u][3] = (uint32_t)(hi[u] >> 32);
			if (a.msb) {
#pragma unroll
				for (int k = 0; k < 4; k++)
					d[u][k] = msb_dword(d[u][k]);
			}
			// offsets of this word that lie inside [0, search_bits)
			uint32_t validA = 0xffffffffu, validB = 0xffffffffu;
			if (tc[u].stream >= a.n_streams) {
				validA = validB = 0;
			} else if (!tile_full(tc[u].t)) {
				const uint64_t first_off = ((uint64_t)tc[u].t * SCAN_THREADS + tid) * 64;
				const uint64_t valid = first_off >= a.search_bits ? 0ULL
					: (a.search_bits - first_off >= 64 ? FULL_MASK : ((1ULL << (a.search_bits - first_off)) - 1));
				validA = (uint32_t)valid;
				validB = (uint32_t)(valid >> 32);
			}
			barker32(d[u][1], d[u][2], validA, m[u][0], cls[u][0]);    // offsets 0..31: window bits 57.. in d1:d2
			barker32(d[u][2], d[u][3], validB, m[u][1], cls[u][1]);    // offsets 32..63
			if (front) {
				const uint32_t d4 = a.msb ? msb_dword(far[u]) : far[u];
				c2[u][0] = slide32<SLIDE4B_TAPS>(d[u][0], d[u][1], d[u][2]);
				c2[u][1] = slide32<SLIDE4B_TAPS>(d[u][1], d[u][2], d[u][3]);
				c2[u][2] = slide32<SLIDE4B_TAPS>(d[u][2], d[u][3], d4);
			}
#ifdef SCAN_PROFILE
			PROF_PIN(m[u][0]); PROF_PIN(m[u][1]);
			if (u == UNROLL - 1) PROF_MARK(14);
#endif
		}

		// Survivor loop: runs while any lane of the wave has survivors; each pass takes one
		// survivor of every 32-offset half in flight (2 * UNROLL independent chains).  The LDS
		// reads of all chains are issued before any result is used.
		// Lanes without a survivor in a chain (45 % of them, measured) read along: their ffbl is ~0, so
		// they form some in-range table address from offset 31, and `m >> p` -- bit 0 set exactly for
		// a lane that has a survivor -- masks their bitmap bit afterwards.  Switching them off in the
		// exec mask instead (a v_cmp, an s_and_saveexec, a skip branch and an s_or per group of reads)
		// was 3 % slower: the loop is bound by instruction issue, not by LDS bank conflicts
		// (profiles/r02_cut).  The bitmap in L2 is still read under exec.
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			PROF_PIN(m[u][0]);
			PROF_PIN(m[u][1]);
		}
		PROF_MARK(0);
		for (uint32_t pass = 1;; pass++) {
			uint32_t any = 0;
#pragma unroll
			for (int u = 0; u < UNROLL; u++)
				any |= m[u][0] | m[u][1];
			if (!__ballot(any != 0))
				break;
			uint32_t p[UNROLL][2], t1[UNROLL][2], t2[UNROLL][2], bw[UNROLL][2], proj[UNROLL][2];
			Probe q[UNROLL][2];
#pragma unroll
			for (int u = 0; u < UNROLL; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					p[u][h] = lowest_bit(m[u][h]);
					q[u][h] = probe_addr(d[u][h], d[u][h + 1], d[u][h + 2], cls[u][h], kdiff, p[u][h]);
					t1[u][h] = lds_ld(LDS_OFF_TABA + q[u][h].offA);
					t2[u][h] = lds_ld(LDS_OFF_TABB + q[u][h].offB);
				}
			uint32_t anybit = 0, bit[UNROLL][2], live[UNROLL][2], i2[UNROLL][2];
			// front set (round 6): 24 positions of the second check stream at the survivor's offset, one dword of a 2 MiB set in L2 per
			// survivor (the four chains' loads in flight together); only its members (22 %) go on to the bitmap over the syndrome.
			// (Sending the front-set loads of pass k + 1 behind the bitmap loads of pass k -- a two-stage pipeline, 107 VGPRs --
			// changed nothing: 7.40 against 7.28 ms per GiB; the kernel runs at the two tables' probe rates, 236 G/s out of the L2 and
			// 88 G/s for the 8 MiB one, not at their latency.  profiles/r06_init5)
			bool go[UNROLL][2];
			if (front) {
				uint32_t v1[UNROLL][2], w1[UNROLL][2];
#pragma unroll
				for (int u = 0; u < UNROLL; u++)
#pragma unroll
					for (int h = 0; h < 2; h++) {
						v1[u][h] = alignbit(c2[u][h + 1], c2[u][h], p[u][h]);
						w1[u][h] = 0;
						if (m[u][h])
							w1[u][h] = a.t.slide4b_bitmap[(v1[u][h] >> 5) & ((1u << (SLIDE4B_BITS - 5)) - 1)];
					}
#pragma unroll
				for (int u = 0; u < UNROLL; u++)
#pragma unroll
					for (int h = 0; h < 2; h++)
						go[u][h] = (int32_t)(w1[u][h] << (v1[u][h] & 31)) < 0;      // (words bit-reversed: member = sign; 0 for an empty chain)
			} else {
#pragma unroll
				for (int u = 0; u < UNROLL; u++)
#pragma unroll
					for (int h = 0; h < 2; h++)
						go[u][h] = m[u][h] != 0;
			}
#pragma unroll
			for (int u = 0; u < UNROLL; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					proj[u][h] = xor3(q[u][h].x, t1[u][h], t2[u][h]);
					// tables for five errors: every value of any set that fits the LDS is a sum of five columns, so the
					// survivors (round 6: those the front set lets through) probe the 2^26-bit bitmap in L2 / Infinity Cache right here
					i2[u][h] = (proj[u][h] * 0x9E3779B1u) >> a.t.bitmap2_shift;
					bw[u][h] = 0;
					if (go[u][h])
						bw[u][h] = a.t.bitmap2[i2[u][h] >> 5];
				}
#pragma unroll
			for (int u = 0; u < UNROLL; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					// only bit 0 counts: (bitmap word >> index) & (m >> p), p = ~0 for m == 0: 0 >> 31
					live[u][h] = m[u][h] >> p[u][h];
					bit[u][h] = bw[u][h] >> (i2[u][h] & 31);
					anybit = BITOP3(bit[u][h], live[u][h], anybit, 0xea);   // anybit |= bit & live, one instruction
					m[u][h] &= m[u][h] - 1;
				}
			if (anybit & 1) {
#pragma unroll
				for (int u = 0; u < UNROLL; u++)
#pragma unroll
					for (int h = 0; h < 2; h++)
						if (bit[u][h] & live[u][h] & 1)  // rare: rebuild the window of this offset and keep it with the code
							park(((it + u) << 12) | (lane << 6) | (h << 5) | (p[u][h] & 31),
							     alignbit(d[u][h + 1], d[u][h], p[u][h]), alignbit(d[u][h + 2], d[u][h + 1], p[u][h]));
			}
			PROF_MARK(pass < 13 ? pass : 13);
		}

		// wave-uniform: compact (and verify) once enough lanes hold a candidate
		PROF_MARK(16);
		if (__popcll(__ballot(n_parked != 0)) >= 24 || __ballot(n_parked >= PARK_SLOTS))
			compact(false);
		PROF_MARK(17);
#ifdef SCAN_PROFILE
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			PROF_PIN(nlo[u]);
			PROF_PIN(nhi[u]);
		}
#endif
		PROF_MARK(19);
#pragma unroll
		for (int u = 0; u < UNROLL; u++) {
			tc[u] = nc[u];
			lo[u] = nlo[u];
			hi[u] = nhi[u];
			far[u] = nfar[u];
		}
	}
	compact(true);
	flush_hits();
#ifdef SCAN_PROFILE
	if (lane < 32)
		atomicAdd(&g_scan_prof[lane], (unsigned long long)lds_ld(prof_off + 4u * lane));
#endif
}


// ---- LAP_ANY, sliding checks (tables for <= 4 errors; two cuts of one kernel: SlideStd / Slide4 below) ----
//
// The kernel of the headline path (promiscuous_packet_search, bluetooth_packet.c:368-420).  Per trip of
// TILES tiles: the bit-sliced barker filter (barker32) and the check stream (slide32, slide.h) for both
// halves of the lane's words, then the lock-step survivor loop -- eight vector instructions and ONE read of the
// 2^SLIDE_BITS-bit candidate set in LDS per survivor (chains as shift registers, round 5).  A candidate goes straight to
// the wave's ring in LDS (the membership compare's lane mask + mbcnt, no atomics); the exact reference rule (verify_lap_any, syndrome tables read
// through L2) runs on ring batches of up to 64.  The ring is the only LDS besides the set, so TWO workgroups
// fit a CU: 2 x 768 threads = 6 waves per SIMD at <= 80 VGPRs (76 KiB of LDS each; a wave owns 63 words of a tile of 756).  Measured on one box,
// 4 GiB, ms per launch (profiles/r03_ab/): one 1024-thread workgroup per CU (4 waves per SIMD) 4.12, 2 x 1024
// (8 waves, 64 VGPRs, spills outside the loop) 3.82-3.90, 2 x 768 3.59, 2 x 896 / 832 / 704 / 640 (waves that do
// not divide evenly over the four SIMDs) 4.4-5.6; the 2^20-bit set (128 KiB, one workgroup per CU only) 3.81.
// Candidates ranked beyond the ring's free entries are checked in place, never dropped (a stream made of sync
// words: tests/test_gpu_scan.py adversarial cases).
// Geometry and tuning (every A/B behind these values is in profiles/: r03_ab, r05_scan).
#define SLIDE_TILES 2                      // tiles a wave works on per trip (2 * SLIDE_TILES chains per lane); 1: +15 %, 3 (80 VGPRs): +1 %
#define SLIDE4_TILES 3                     // ... of the two-level form (tables for three and four errors; 2: +2.5 %, 4: +20 %)
#define SLIDE_WGS 2                        // workgroups per CU the kernel is cut for
#define SLIDE_THREADS 768                  // workgroup size = words per tile (a multiple of 256: whole waves per SIMD); 2 x 1024: +2.5 % (round 5, spills); round 6,
                                           // the ordered form at 64 registers without a spill: 3.24 against 3.01 ms -- eight waves per SIMD are SLOWER (profiles/r06_order)
#define SLIDE_FIXED 6                      // passes run before the first "anything left?" test of a trip (5: +2 %, 7: +1 %)
#define SLIDE_DRAIN_AT 60u                 // 64-entry ring: entries at which a trip end drains it (32 / 48 / 56 / 60: 3.56 / 3.48 / 3.46 / 3.455 ms; round 6 on
                                           // the 63-word kernel: 32 +1.5 %, 40 and 48 nothing -- profiles/r06_scan/ab_b3_drain_threshold.txt)
#define SLIDE_DRAIN_AT_ORD 40u             // ... of the ordered form (see its drain)
// Round 6 measured three more forms of this kernel and dropped them (profiles/r06_scan; the source with the switches is kept there as text):
//   * the fixed passes without compare, scalar OR and branch -- the sign of (set word << index) shifted into a hit register per chain, one look
//     at the registers behind the last pass, the candidate's record carrying its survivor's ordinal for the drain to turn into an offset:
//     bit-exact, 17 % fewer scalar instructions, 5.4 % MORE vector instructions, +10 % time (3.21 against 2.92 ms; fully unrolled 3.03).  The
//     launch follows its vector instruction count; scalar instructions and branches are not what it waits for.
//   * a drain's hits written straight behind one counter atomic each (no pending records in registers: 66 VGPRs): 4.10 ms -- 322 k
//     returning atomics on one address serialise (SQ_WAIT_ANY 2.7 x).
//   * two chains per word walking towards each other (6.09 passes instead of 6.99): not built -- tools/lockstep_model.py prices it at +34 % per
//     chain and pass (64-bit survivor masks, 82 check bits per chain) for -13 % passes.
// The kernel's two cuts.
// SlideStd: tables for <= 2 errors (0.3 % of the survivors are members of the set).  Two workgroups per CU around a 2^19-bit set; six
//   passes run blind, a candidate the ring has no room for is checked in place.
// Slide4: tables for three and four errors (slide.h), where 3 % / 32 % of the survivors are members of any set the LDS can hold.  ONE
//   workgroup per CU around a 2^20-bit set (the whole LDS: 128 KiB + 2 KiB of ring per wave); its members look a second check
//   stream up in a set in L2 before they count as candidates (LEVEL2); the pass loop watches the ring's room and is left for
//   drains (DENSE: the room test in every pass costs the sparse case 4 %, the in-place path costs a dense case a factor of three).
//   INVERT: the chains run on the complemented check stream -- an idle chain indexes 0 or 1, which are members of the set for four
//   errors while their complements are not (context.cpp stores the set accordingly).
// Measured, ms per GiB (tools/init_sweep.py, profiles/r05_init4): three errors 1.72-1.75 (SlideStd in a dense form, rounds 3-4) ->
// 1.42-1.44; four errors 2.78-2.84 (a probe kernel: three table reads per survivor, 58 % of them to L2) -> 2.09-2.13.
struct SlideStd {
	static constexpr int BITS = SLIDE_BITS, THREADS = SLIDE_THREADS, WGS = SLIDE_WGS;
	static constexpr uint64_t TAPS = SLIDE_TAPS, TAPS_B = 0;
	static constexpr bool LEVEL2 = false, INVERT = false, DENSE = false;
};
struct Slide4 {
	static constexpr int BITS = SLIDE4_BITS, THREADS = 1024, WGS = 1;
	static constexpr uint64_t TAPS = SLIDE4_TAPS, TAPS_B = SLIDE4B_TAPS;
	static constexpr bool LEVEL2 = true, INVERT = true, DENSE = true;
};
template <class CFG> struct SlideGeom {
	static constexpr uint32_t SET_WORDS = 1u << (CFG::BITS - 5), SET_BYTES = 4u * SET_WORDS;
	static constexpr uint32_t WAVES_PER_EU = CFG::WGS * CFG::THREADS / 256;
#ifdef SCAN_PROFILE
	static constexpr uint32_t RING = 64;                                    // (the phase counters need 2 KiB of the two-level form's full LDS)
#else
	static constexpr uint32_t RING = CFG::WGS == 2 ? 64 : 128;               // ring entries per wave
#endif
	static constexpr uint32_t LANE_WORDS = 63;                              // words of a tile a wave owns (see the kernel)
	static constexpr uint32_t TILE_WORDS = CFG::THREADS / 64 * LANE_WORDS;
	static constexpr uint32_t RING_END = SET_BYTES + CAND_BYTES * (CFG::THREADS / 64) * RING;
#ifdef SCAN_PROFILE
	static constexpr uint32_t LDS_BYTES = RING_END + 128u * (CFG::THREADS / 64);     // 32 phase counters per wave
#else
	static constexpr uint32_t LDS_BYTES = RING_END;
#endif
	static_assert((uint64_t)LDS_BYTES * CFG::WGS <= 160u * 1024u, "the workgroups a CU is cut for must fit its 160 KiB of LDS");
};

// MSB: the words hold their symbols MSB first in every byte (BTBBX_FMT_PACKED_MSB); a template flag, not a run-time branch: the
// branch alone cost the LSB path 1 % here and 7 % in scan_known_lap_kernel (the words' registers become merge points)
// ORD: hits leave through the segment slots (ScanArgs::seg_slots) instead of the appended list
template <class CFG, int TILES, bool MSB, bool ORD = false>
__global__ __launch_bounds__(CFG::THREADS) __attribute__((amdgpu_waves_per_eu(SlideGeom<CFG>::WAVES_PER_EU, SlideGeom<CFG>::WAVES_PER_EU)))
void scan_slide_kernel(ScanArgs a)
{
	extern __shared__ uint32_t lds[];
	if (a.gate && *a.gate == 0)
		return;
	constexpr uint32_t RING = SlideGeom<CFG>::RING;
	// ABS (third session of round 6, the one-level form): a chain is walked by ABSOLUTE positions -- p = v_ffbl of what is left of
	// its mask, index = the untouched 64-bit check register >> p, mask &= mask - 1 -- instead of the pair of shift registers
	// below.  The same instructions per survivor (v_add + v_and for v_lshrrev + v_and), but no marker to plant per chain and trip,
	// no v_ffbh per candidate event (p IS the offset) and nothing loop-carried but the mask: 2.893 -> 2.879 ms over six
	// alternating pairs (profiles/r06_shift).  The two-level form keeps the shift registers: its events run a pass behind.
	constexpr bool ABS = !CFG::LEVEL2;
	constexpr uint32_t THREADS = CFG::THREADS, SET_WORDS = SlideGeom<CFG>::SET_WORDS, SET_BYTES = SlideGeom<CFG>::SET_BYTES;

	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: ring addresses stay on the SALU
	// A wave owns LANE_WORDS = 63 consecutive words of a tile; its lane 63 works on the NEXT wave's first word, only so that
	// lane 62 gets the check bits behind its own word (positions 64 .. 95) from a neighbour like every other lane.  (With 64
	// words per wave those eighteen bits of lane 63 came from the scalar unit: two readlanes and 22 scalar shifts / XORs per
	// tile and check stream -- 5 % of the instructions a wave issues per trip, for one lane; a lane in 64 idles instead.)
	constexpr uint32_t LANE_WORDS = SlideGeom<CFG>::LANE_WORDS, TILE_WORDS = SlideGeom<CFG>::TILE_WORDS;
	const uint32_t wid = wave * LANE_WORDS + lane;                       // this lane's word in a tile
	uint32_t live = lane != 63 ? 0xffffffffu : 0u;                       // offsets of the lane's word that are its own
	asm volatile("" : "+v"(live));
	const uint32_t ring_off = SET_BYTES + CAND_BYTES * wave * RING;

	// tile order: one contiguous eighth of the tiles per XCD, its workgroups interleaved (see scan_lap_any_kernel)
	uint32_t first_tile = blockIdx.x, tile_step = gridDim.x, n_mine;
	if (a.xcd_tiles) {
		const uint32_t xcd = blockIdx.x & 7, lo_t = xcd * a.xcd_tiles;
		const uint32_t hi_t = min((uint64_t)lo_t + a.xcd_tiles, a.n_tiles);
		tile_step = gridDim.x >> 3;
		first_tile = lo_t + (blockIdx.x >> 3);
		n_mine = first_tile < hi_t ? (hi_t - first_tile + tile_step - 1) / tile_step : 0;
	} else {
		n_mine = first_tile < a.n_tiles ? (uint32_t)((a.n_tiles - first_tile + tile_step - 1) / tile_step) : 0;
	}

	{	// candidate set -> LDS byte 0, 16 bytes per lane per step, as 16-bit entries, every entry bit-reversed: the member bit of
		// index i is bit 15 - (i & 15) of entry i >> 4, so that a LEFT shift by i brings it to the entry's sign bit -- "member" is
		// then one signed 16-bit compare, whose result (a lane mask in scalar registers) is also the ballot the candidate path
		// needs.  Sixteen bits, not thirty-two (rounds 5-6a): on gfx950 v_lshlrev_b32 issues at the slow rate (4.1 cycles per wave,
		// like v_alignbit) while v_lshlrev_b16 and the RIGHT shifts issue at the fast one (2.3-2.5; tools/valu_rate.hip,
		// profiles/r06_scan/valu_rate_shifts.txt) -- one left shift per survivor.
		const uint4 *src = reinterpret_cast<const uint4 *>(CFG::LEVEL2 ? a.t.slide4_bitmap : a.t.slide_bitmap);
		uint4 *dst = reinterpret_cast<uint4 *>(lds);
		auto rev16 = [](uint32_t x) { const uint32_t r = __brev(x); return (r >> 16) | (r << 16); };   // both halves reversed in place
		for (uint32_t i = tid; i < SET_WORDS / 4; i += THREADS) {
			const uint4 v = src[i];
			dst[i] = make_uint4(rev16(v.x), rev16(v.y), rev16(v.z), rev16(v.w));
		}
	}
	__syncthreads();

#ifdef SCAN_PROFILE
	// phases: 0 = tile loads + barker filter + check stream, 1 .. 13 = survivor pass k, 16 = loop exit, 18 = ring drain,
	// 19 = hand-over to the next trip
	const uint32_t prof_off = SlideGeom<CFG>::RING_END + 128u * wave;
	if (lane < 32)
		lds_st(prof_off + 4u * lane, 0u);
	uint64_t prof_t;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t) : : "memory");
#endif
	uint32_t q_head = 0, q_tail = 0;          // wave-uniform ring cursors (free running)
	// code = (tile iteration << 12) | (lane that owns the word << 6) | offset in the word
	uint32_t code_tile = 0;                       // (set by code_word: the tile's number inside its stream)
	auto code_word = [&](uint32_t code, uint32_t &stream) {
		const uint32_t tile = first_tile + (code >> 12) * tile_step;
		uint32_t t = tile;
		stream = 0;
		if (a.n_streams > 1) {
			stream = tile / (uint32_t)a.tiles_per_stream;
			t = tile - stream * (uint32_t)a.tiles_per_stream;
		}
		code_tile = t;
		return (uint64_t)t * TILE_WORDS + wave * LANE_WORDS + ((code >> 6) & 63);
	};
	// hits: up to 64 pending records per wave in registers, written 1 KiB at a time behind one counter atomic
	uint32_t pend = 0;                            // wave-uniform
	uint32_t h_off = 0, h_hi = 0, h_lap = 0;      // lane k < pend: offset low, offset high | stream << 16, lap << 8 | errors
	auto flush_hits = [&]() {
		if (pend == 0)
			return;
		uint32_t base = 0;
		if (lane == 0)
			base = atomicAdd(a.hit_count, pend);
		base = __builtin_amdgcn_readfirstlane(base);
		const uint32_t idx = base + lane;
		if (lane < pend && idx < a.hit_cap) {
			uint4 rec;
			rec.x = h_off;
			rec.y = h_hi & 0xffff;
			rec.z = h_lap >> 8;
			rec.w = (h_lap & 0xff) | (h_hi & 0xffff0000u);
			reinterpret_cast<uint4 *>(a.hits)[idx] = rec;
			count_bucket(a, h_hi >> 16, ((uint64_t)(h_hi & 0xffff) << 32) | h_off);
		}
		pend = 0;
	};
	auto push_hits = [&](bool hit, uint32_t stream, uint64_t offset, uint32_t lap, uint32_t nerr) {
		if (a.first) {                            // first-match mode: atomicMin, hits go out one by one
			if (hit)
				emit_hit(a, stream, offset, lap, nerr);
			return;
		}
		const uint64_t m = __ballot(hit);
		if (!m)
			return;
		const uint32_t c = (uint32_t)__popcll(m);
		if (pend + c > 64)
			flush_hits();
		const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
		const int dst = (int)((hit ? pend + rank : (pend ? 0u : c)) << 2);
		const uint32_t r_off = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)(uint32_t)offset);
		const uint32_t r_hi = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)((uint32_t)(offset >> 32) | (stream << 16)));
		const uint32_t r_lap = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)((lap << 8) | nerr));
		if (lane - pend < c) {
			h_off = r_off;
			h_hi = r_hi;
			h_lap = r_lap;
		}
		pend += c;
	};
	auto drain = [&](uint32_t n) {               // the n <= 64 oldest ring entries through the exact rule
		bool hit = false;
		uint32_t stream = 0, lap = 0, nerr = 0;
		uint64_t offset = 0;
		u32x4 rec = {0u, 0u, 0u, 0u};
		if (lane < n)
			rec = lds_ld4(ring_off + CAND_BYTES * ((q_head + lane) & (RING - 1)));
		const uint32_t code = rec.x;
		const uint64_t word = code_word(code, stream);
		if (lane < n) {
			const uint64_t w = ((uint64_t)alignbit(rec.w, rec.z, code) << 32) | alignbit(rec.z, rec.y, code);   // (shift = the low five bits)
			offset = word * 64 + (code & 63);
			hit = verify_lap_any<false>(a, w, lap, nerr);
		}
		if constexpr (ORD) {
			// A drain takes whole trips, so every hit of a segment (tile iteration code >> 12 of this wave) is in this batch: its
			// rank = the hits of the same tile with a smaller code (lane, offset) -- one scalar trip per hit of the batch --, its
			// place = slot `rank` of the segment.  The hit with the highest rank stores the segment's count.
			const uint64_t hm = __ballot(hit);
			if (hm) {
				uint32_t rank = 0, count = 0;
				// The ring is empty now (its records sit in registers) and lends its kilobyte: a hit counter per tile iteration of the
				// batch -- ring entries are in trip order, so the iterations run from the oldest entry's (even) one to the newest's -- and
				// room for four 12-bit codes per tile.  A hit's count = its tile's counter, its rank = the codes of its tile below its own.
				// (One scalar trip per hit of the batch over all lanes instead -- 300 instructions per drain -- cost the launch 8 %.)
				// A batch that spans 64 iterations or more (a sparse stream: few hits) or a tile with more than four hits: that loop.
				const uint32_t it_mine = code >> 12;
				const uint32_t it_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)it_mine) & ~1u;
				const uint32_t it_hi = (uint32_t)__builtin_amdgcn_readlane((int)it_mine, (int)(n - 1)) | 1u;
				bool fast = it_hi - it_lo < 64u;
				if (fast) {
					const uint32_t key = (it_mine - it_lo) & 63u;
					lds_st(ring_off + 4u * lane, 0u);
					uint32_t idx = 0;
					if (hit) {
						idx = __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t *>(ring_off + 4u * key), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
						if (idx < 4u)
							*reinterpret_cast<__attribute__((address_space(3))) uint16_t *>(ring_off + 256u + 8u * key + 2u * idx) = (uint16_t)(code & 0xfffu);
					}
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					if (hit)
						count = lds_ld(ring_off + 4u * key);
					if (__ballot(count > 4u)) {
						fast = false;
					} else if (hit) {
						const uint32_t lo2 = lds_ld(ring_off + 256u + 8u * key), hi2 = lds_ld(ring_off + 260u + 8u * key);
						const uint32_t mine = code & 0xfffu;
						rank = ((lo2 & 0xffffu) < mine ? 1u : 0u);                      // (entry 0 always exists; the own entry is not below itself)
						rank += count > 1u && (lo2 >> 16) < mine ? 1u : 0u;
						rank += count > 2u && (hi2 & 0xffffu) < mine ? 1u : 0u;
						rank += count > 3u && (hi2 >> 16) < mine ? 1u : 0u;
					}
				}
				if (!fast) {
					rank = count = 0;
					for (uint64_t r = hm; r; r &= r - 1) {
						const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)code, (int)__builtin_ctzll(r));
						const bool same = (code ^ cj) < 4096u;
						count += same ? 1u : 0u;
						rank += same && cj < code ? 1u : 0u;
					}
				}
				const uint32_t seg = stream * a.segs_per_stream + code_tile * (THREADS / 64) + wave;
				uint4 out;
				out.x = (uint32_t)offset;
				out.y = (uint32_t)(offset >> 32);
				out.z = lap;
				out.w = nerr | (stream << 16);
				const bool spill = hit && rank >= a.seg_slot_n;
				if (hit && !spill)       // (code & 0xfff = lane << 6 | offset in the word = the offset inside the wave's 63 words)
					a.seg_slots[(uint64_t)seg * a.seg_slot_n + rank] = (uint64_t)(code & 0xfffu) | ((uint64_t)lap << 12) | ((uint64_t)nerr << 36);
				if (hit && rank + 1 == count)
					a.seg_cnt[seg] = (uint16_t)count;            // (<= 4032 offsets per segment)
				const uint64_t om = __ballot(spill);
				if (om) {                                            // more hits in 4032 offsets than a segment has slots: rare
					uint32_t base = 0;
					if (lane == 0)
						base = atomicAdd(a.ovf_count, (uint32_t)__popcll(om));
					base = __builtin_amdgcn_readfirstlane(base);
					const uint32_t idx = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0));
					if (spill) {
						if (idx < a.ovf_cap) {
							reinterpret_cast<uint4 *>(a.ovf_recs)[idx] = out;
							a.ovf_meta[idx] = make_uint2(seg, rank);
						} else {
							*a.irregular = 1u;
						}
					}
				}
			}
		} else {
			push_hits(hit, stream, offset, lap, nerr);
		}
		q_head += n;
	};

	// The tile cursor carries its tile's address along (one 64-bit scalar add per tile; the products stream x pitch and tile x
	// words are formed again only when it crosses into the next stream), and a lane's word is that address + a byte offset
	// it computes once: `global_load ... v_off, s[base]` -- no 64-bit vector address arithmetic per load (round 5: the scalar
	// unit's instructions are not free, they take about two issue cycles each from the same wave).
	struct Cursor { uint32_t stream; uint32_t t; const uint64_t *tp; };
	const uint32_t tiles_per_stream = (uint32_t)a.tiles_per_stream;
	Cursor cur = {a.n_streams, 0, a.words};      // stream == n_streams: nothing (left) to do
	uint32_t handed = 0;
	auto tile_address = [&](const Cursor &c) { return a.words + (uint64_t)c.stream * a.pitch_words + (uint64_t)c.t * TILE_WORDS; };
	if (n_mine) {
		cur.stream = a.n_streams > 1 ? first_tile / tiles_per_stream : 0;
		cur.t = first_tile - cur.stream * tiles_per_stream;
		cur.tp = tile_address(cur);
	}
	// (the product is formed again at every tile -- two scalar multiplies: hoisted, it lived in a spilled SGPR pair and came back
	// through two v_readlane per tile, vector instructions on the path of every trip)
	auto step_words = [&]() {
		uint32_t ts = tile_step;
		asm volatile("" : "+s"(ts));
		return (uint64_t)ts * TILE_WORDS;
	};
	auto advance = [&](Cursor &c) {
		if (++handed >= n_mine) {
			c.stream = a.n_streams;
			return;
		}
		c.t += tile_step;
		c.tp += step_words();
		if (c.t >= tiles_per_stream) {
			while (c.t >= tiles_per_stream && c.stream < a.n_streams) {
				c.t -= tiles_per_stream;
				c.stream++;
			}
			c.tp = tile_address(c);
		}
	};
	auto tile_full = [&](uint32_t tt) { return tt < a.full_tiles; };
	uint32_t voff = wid * 8u;                                            // this lane's word in a tile, in bytes
	asm volatile("" : "+v"(voff));
	// A lane's two words (its own and the one behind it) come through a BUFFER descriptor over the tile: base = the cursor's tile
	// address, extent = the words of the stream that are left there, so the hardware's range check returns zero for a word
	// behind the stream's end (checked per dword) -- one 16-byte load from a 32-bit lane offset, no 64-bit vector address, no
	// zero-initialised destination, no exec mask for the ragged tile.  (Third session of round 6: the global loads cost seven
	// vector instructions per tile -- four v_mov, a v_mov_b64, a v_lshl_add_u64 -- on the path of every full tile.)
	auto load_pair = [&](const Cursor &c, uint64_t &lo, uint64_t &hi) {
		uint32_t bytes = 0;                                                  // wave-uniform
		if (c.stream < a.n_streams) {
			bytes = (TILE_WORDS + 2u) * 8u;                                  // (a full tile: its words and two behind it are in range)
			if (!tile_full(c.t)) {
				const uint64_t first = (uint64_t)c.t * TILE_WORDS;
				const uint64_t left = first < a.n_words ? a.n_words - first : 0;
				bytes = (uint32_t)(left < TILE_WORDS + 2u ? left : TILE_WORDS + 2u) * 8u;
			}
		}
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(c.tp), 0, (int)bytes, 0x00020000);
		const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
		lo = ((uint64_t)v.y << 32) | v.x;
		hi = ((uint64_t)v.w << 32) | v.z;
	};

	Cursor tc[TILES];
	uint64_t lo[TILES], hi[TILES];
#pragma unroll
	for (int u = 0; u < TILES; u++) {
		tc[u] = cur;
		load_pair(cur, lo[u], hi[u]);
		advance(cur);
	}

	for (uint32_t it = 0; tc[0].stream < a.n_streams; it += TILES) {
		__builtin_amdgcn_s_setprio(PRIO_FILTER);
		uint32_t d[TILES][4], m[TILES][2], c[TILES][3];
		uint32_t c2[TILES][CFG::LEVEL2 ? 3 : 1];                 // (two-level form) the second check stream, positions as c
#pragma unroll
		for (int u = 0; u < TILES; u++) {
			d[u][0] = (uint32_t)lo[u]; d[u][1] = (uint32_t)(lo[u] >> 32);
			d[u][2] = (uint32_t)hi[u]; d[u][3] = (uint32_t)(hi[u] >> 32);
			if constexpr (MSB) {
#pragma unroll
				for (int k = 0; k < 4; k++)
					d[u][k] = msb_dword(d[u][k]);
			}
			uint32_t cls_unused;
			barker32(d[u][1], d[u][2], live, m[u][0], cls_unused);      // offsets 0..31: window bits 57.. in d1:d2
			barker32(d[u][2], d[u][3], live, m[u][1], cls_unused);      // offsets 32..63
			// offsets beyond the search length (the last tile of a stream only): cut out of the masks BEHIND the filter -- as two
			// validity masks in front of it they were two register copies per tile on the path of every full tile
			if (tc[u].stream >= a.n_streams) {
				m[u][0] = m[u][1] = 0;
			} else if (!tile_full(tc[u].t)) {
				const uint64_t first_off = ((uint64_t)tc[u].t * TILE_WORDS + wid) * 64;
				const uint64_t valid = first_off >= a.search_bits ? 0ULL
					: (a.search_bits - first_off >= 64 ? FULL_MASK : ((1ULL << (a.search_bits - first_off)) - 1));
				m[u][0] &= (uint32_t)valid;
				m[u][1] &= (uint32_t)(valid >> 32);
			}
			c[u][0] = slide32<CFG::TAPS>(d[u][0], d[u][1], d[u][2]);
			c[u][1] = slide32<CFG::TAPS>(d[u][1], d[u][2], d[u][3]);
			if constexpr (CFG::INVERT) {
				c[u][0] = ~c[u][0];
				c[u][1] = ~c[u][1];
			}
			// positions 64..95 = the first check dword of the next lane's word (lane 63 has no offsets of its own, see above)
			c[u][2] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + 1) << 2), (int)c[u][0]);
			if constexpr (CFG::LEVEL2) {
				c2[u][0] = slide32<CFG::TAPS_B>(d[u][0], d[u][1], d[u][2]);
				c2[u][1] = slide32<CFG::TAPS_B>(d[u][1], d[u][2], d[u][3]);
				c2[u][2] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane + 1) << 2), (int)c2[u][0]);
			}
		}

		// A chain (32 offsets) as a pair of shift registers: its survivor mask and the 50 check bits its indices are cut from,
		// both moved down to the survivor in hand (one v_lshrrev_b64 instead of a funnel shift per survivor, no "m - 1").  Bit
		// 63 is a marker: its distance from the top is the offset the chain stands at, which only a candidate event asks for.
		uint64_t C[TILES][2];
#pragma unroll
		for (int u = 0; u < TILES; u++)
#pragma unroll
			for (int h = 0; h < 2; h++)
				C[u][h] = ((uint64_t)(ABS ? c[u][h + 1] : (c[u][h + 1] | 0x80000000u)) << 32) | c[u][h];
		struct Stage { uint32_t v[TILES][2], bw[TILES][2]; };
		auto any_left = [&]() {
			uint32_t any = 0;
#pragma unroll
			for (int u = 0; u < TILES; u++)
				any |= m[u][0] | m[u][1];
			return __ballot(any != 0) != 0;
		};
		uint32_t pos2[TILES][2];                     // (two-level form) where the chains stood when their pending look-ups were sent
		auto events = [&](const uint64_t (&cms)[TILES][2]) {   // append the wave's candidates of one pass to its ring
#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					const uint64_t cm = cms[u][h];
					if (!cm)
						continue;
					const bool cand = __builtin_amdgcn_inverse_ballot_w64(cm);
					// ring entries left for this chain; candidates ranked beyond them (a stream made of
					// sync words: tests/test_gpu_scan.py adversarial cases) go through the exact rule in place
					const uint32_t room = RING - (q_tail - q_head);
					uint32_t in_wave;                       // (asm: the compiler turns `popcount == 1` into a 64-bit VECTOR compare)
					asm("s_bcnt1_i32_b64 %0, %1" : "=s"(in_wave) : "s"(cm) : "scc");
					const uint32_t n = min(in_wave, room);
					if (cand) {
						uint32_t lane6 = lane << 6;
						asm volatile("" : "+v"(lane6));         // (otherwise four loop-invariant code bases sit in VGPRs through the pass loop)
						// the marker planted above the chain's check bits has moved down by exactly the offsets passed
						uint32_t pos;
						if constexpr (ABS)
							pos = pos2[u][h];                   // (a candidate's chain was not empty: 0 .. 31)
						else if constexpr (CFG::LEVEL2)
							pos = pos2[u][h];
						else
							asm("v_ffbh_u32 %0, %1" : "=v"(pos) : "v"((uint32_t)(C[u][h] >> 32)));
						// the record carries the three stream dwords the window lies in; the drain cuts it out (for sixty
						// candidates at once) instead of this branch (for one)
						const uint32_t code = pos | lane6 | (((it + u) << 12) | (h << 5));
						const u32x4 rec = {code, d[u][h], d[u][h + 1], d[u][h + 2]};
						if (in_wave == 1 && room) {
							// one candidate in the wave (nine events in ten): its slot is the ring tail, no ranking
							lds_st_rec(ring_off + CAND_BYTES * (q_tail & (RING - 1)), rec);
						} else {
							const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32),
									__builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0));
							if (rank < room) {
								lds_st_rec(ring_off + CAND_BYTES * ((q_tail + rank) & (RING - 1)), rec);
							} else {
								uint32_t stream, lap, nerr, cold = code;
								asm volatile("" : "+v"(cold));      // keeps the tile -> stream division of this cold path out of every trip
								const uint64_t word = code_word(cold, stream);
								const uint32_t wlo = alignbit(rec.z, rec.y, pos), whi = alignbit(rec.w, rec.z, pos);
								if (verify_lap_any<false>(a, ((uint64_t)whi << 32) | wlo, lap, nerr)) {
									if constexpr (ORD)
										*a.irregular = 1u;          // a hit outside the drains: its segment cannot be ranked here
									else
										emit_hit(a, stream, word * 64 + (cold & 63), lap, nerr);
								}
							}
						}
					}
					q_tail += n;
				}
		};
		// (An empty chain shifts itself out: its index becomes 0 or 1, which no table set contains -- context.cpp asserts it --,
		// so a lane without a survivor never looks like a candidate and the test needs no "this lane has one" term.)
		auto step = [&](int u, int h, Stage &g) {       // next survivor of a chain: index, set read in flight
			const uint32_t p = lowest_bit(m[u][h]);     // ~0 for an empty chain
			if constexpr (ABS) {
				g.v[u][h] = (uint32_t)(C[u][h] >> (p & 63));      // (an empty chain: bit 63 alone = index 0 or 1)
				g.bw[u][h] = lds_ld16((g.v[u][h] >> 3) & (SET_BYTES - 2));
				m[u][h] &= m[u][h] - 1u;
				pos2[u][h] = p;
			} else {
				m[u][h] >>= p & 31;
				C[u][h] >>= p & 63;
				g.v[u][h] = (uint32_t)C[u][h];
				g.bw[u][h] = lds_ld16((g.v[u][h] >> 3) & (SET_BYTES - 2));
				m[u][h] &= ~1u;
			}
		};
		auto member = [&](int u, int h, const Stage &g) {   // lanes whose index is in the set (the compare's own mask: no ballot)
			return sign16_after_shl(g.bw[u][h], g.v[u][h]);
		};
		// Two-level form: a third of the survivors are members of the LDS set; they alone (exec mask) look their SLIDE4B_BITS
		// positions of the second check stream up in the set in L2 -- one dword each, the four chains' loads in flight together.
		// The position comes from the chain's marker, as in a candidate event.  The look-ups of a pass are sent at its end and
		// looked at in the NEXT pass, behind that pass's own steps (level2_take): the L2's answer has a pass to arrive in.
		// (No "this chain has no member in any lane" shortcut: a branch per chain makes the compiler wait for the loads at
		// every merge -- 1.9 against 1.43 ms per GiB with tables for three errors, where a third of the chain-passes could skip.)
		uint32_t v2[TILES][2], w2[TILES][2] = {};            // (w2: a lane without a look-up in flight keeps a stale word; `sent` masks its answer)
		uint64_t sent[TILES][2], any_sent = 0;               // lanes with a look-up in flight, per chain
		auto level2_send = [&](const uint64_t (&cms)[TILES][2]) {
			any_sent = 0;
#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					sent[u][h] = cms[u][h];
					any_sent |= cms[u][h];
					asm("v_ffbh_u32 %0, %1" : "=v"(pos2[u][h]) : "v"((uint32_t)(C[u][h] >> 32)));
					v2[u][h] = alignbit(c2[u][h + 1], c2[u][h], pos2[u][h]);
					if (__builtin_amdgcn_inverse_ballot_w64(cms[u][h]))
						w2[u][h] = a.t.slide4b_bitmap[(v2[u][h] >> 5) & ((1u << (SLIDE4B_BITS - 5)) - 1)];
				}
		};
		auto level2_take = [&]() {
			if (!any_sent)
				return;
			uint64_t cms[TILES][2], any = 0;
#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					cms[u][h] = sent[u][h] & __ballot((int32_t)(w2[u][h] << (v2[u][h] & 31)) < 0);
					any |= cms[u][h];
				}
			any_sent = 0;
			if (any)
				events(cms);
		};
		auto pass = [&]() {
			Stage g;
#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++)
					step(u, h, g);
			// (the compiler knows nothing about the latency of the shift and compare written as asm in member(): without this it
			// slips each set read behind the previous chain's compare and waits for the reads one at a time)
			__builtin_amdgcn_sched_barrier(0);
			uint64_t cms[TILES][2], any = 0;
#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					cms[u][h] = member(u, h, g);
					any |= cms[u][h];
				}
			if constexpr (CFG::LEVEL2) {
				level2_take();                           // the previous pass's look-ups, then this pass's are sent
				// ("any member in the wave" formed HERE and on the scalar unit by name: carried across level2_take's branches the
				// compiler re-formed it from the six lane masks with twelve VECTOR instructions per pass)
				static_assert(!CFG::LEVEL2 || TILES == 3, "the scalar OR below is written for six chains");
				uint64_t any2;
				asm("s_or_b64 %0, %1, %2\n\ts_or_b64 %0, %0, %3\n\ts_or_b64 %0, %0, %4\n\ts_or_b64 %0, %0, %5\n\ts_or_b64 %0, %0, %6"
				    : "=&s"(any2) : "s"(cms[0][0]), "s"(cms[0][1]), "s"(cms[1][0]), "s"(cms[1][1]), "s"(cms[TILES - 1][0]), "s"(cms[TILES - 1][1]) : "scc");
				if (any2)
					level2_send(cms);
			} else if (any) {                            // some lane of the wave holds a candidate (half of the passes)
				events(cms);
			}
		};
		// Behind the fixed passes a handful of the wave's 2 * TILES * 64 chains still hold survivors (0.8 % have seven or more):
		// a pass then looks at the chains one by one and skips those that are empty wave-wide (the same ballots are the
		// loop's exit test), instead of paying the full pass for two or three lanes.
		auto sparse_tail = [&]() {
			uint64_t live[TILES][2], anyl = 0;

#pragma unroll
			for (int u = 0; u < TILES; u++)
#pragma unroll
				for (int h = 0; h < 2; h++) {
					live[u][h] = __ballot(m[u][h] != 0);
					anyl |= live[u][h];
				}
			while (anyl) {
				Stage g;
				uint64_t cms[TILES][2], anyc = 0;
				anyl = 0;
#pragma unroll
				for (int u = 0; u < TILES; u++)
#pragma unroll
					for (int h = 0; h < 2; h++) {
						cms[u][h] = 0;
						if (!live[u][h])
							continue;
						step(u, h, g);
						cms[u][h] = member(u, h, g);
						anyc |= cms[u][h];
						live[u][h] = __ballot(m[u][h] != 0);
						anyl |= live[u][h];
					}
				if (anyc)
					events(cms);
			}
		};
#ifdef SCAN_PROFILE
#pragma unroll
		for (int u = 0; u < TILES; u++) {
			PROF_PIN(m[u][0]); PROF_PIN(m[u][1]); PROF_PIN(c[u][0]); PROF_PIN(c[u][1]); PROF_PIN(c[u][2]);
		}
#endif
		PROF_MARK(0);
		uint32_t pass_no = 1;
		if constexpr (!CFG::DENSE) {
			__builtin_amdgcn_s_setprio(PRIO_LOOP);
#pragma unroll 1
			for (int k = 0; k < SLIDE_FIXED; k++) { // practically every trip needs these (TILES * 128 chains of ~4 survivors)
				pass();
				PROF_MARK(pass_no < 13 ? pass_no : 13);
				pass_no++;
			}
			sparse_tail();
			PROF_MARK(pass_no < 13 ? pass_no : 13);
			__builtin_amdgcn_s_setprio(PRIO_CAND);
			PROF_MARK(16);
			// (ORD: drained at 40, which costs nothing measurable -- profiles/r06_scan -- and leaves every trip room for 24 candidates
			// where it has 4.5: a hit verified in place, outside the drains, then only happens to streams made of sync words)
			if (q_tail - q_head >= (RING == 64 ? (ORD ? SLIDE_DRAIN_AT_ORD : SLIDE_DRAIN_AT) : 64u))
				drain(q_tail - q_head > 64 ? 64 : q_tail - q_head);
		} else {
			// The pass loop is left when the ring gets short of room (a.ring_margin entries: what a pass may add), drained at
			// the one site behind it and re-entered; candidates that still find no room are checked in place.  (With the
			// drain inside the pass loop its hit registers would be loop-carried through every pass.)
			for (;;) {
				__builtin_amdgcn_s_setprio(PRIO_LOOP);
				bool more = true;
				while (q_tail - q_head + a.ring_margin <= RING) {
					if (!any_left()) {
						more = false;
						break;
					}
					pass();
					PROF_MARK(pass_no < 13 ? pass_no : 13);
					pass_no++;
				}
				if constexpr (CFG::LEVEL2)
					level2_take();                       // (the look-ups of the last pass)
				__builtin_amdgcn_s_setprio(PRIO_CAND);
				PROF_MARK(16);
				if (more || q_tail - q_head >= 32u)
					drain(q_tail - q_head > 64 ? 64 : q_tail - q_head);
				if (!more)
					break;
			}
		}
		(void)pass_no;
		PROF_MARK(18);
#pragma unroll
		for (int u = 0; u < TILES; u++) {
			// no software prefetch: the other five waves of the SIMD cover the loads, and the eight registers it took are
			// worth more (round 5: 3.35 against 3.38 ms; round 6 again, the loads issued right behind the filter and checked in
			// the ISA to be waited for only at the next trip's head, 77 VGPRs: 3.03-3.06 against 2.93-2.95 -- profiles/r06_scan)
			tc[u] = cur;
			load_pair(cur, lo[u], hi[u]);
			advance(cur);
		}
		PROF_MARK(19);
	}
	while (q_tail != q_head)
		drain(q_tail - q_head > 64 ? 64 : q_tail - q_head);
	flush_hits();
#ifdef SCAN_PROFILE
	if (lane < 32)
		atomicAdd(&g_scan_prof[lane], (unsigned long long)lds_ld(prof_off + 4u * lane));
#endif
}


// ---- known LAP --------------------------------------------------------------------------

// Truth table of a three-input function whose inputs are (compile-time) inverted: index = a*4 + b*2 + c as v_bitop3 wants it.
// The sync word's top seven bits are the LAP's MSB and the barker code that follows from it (bluetooth_packet.c:81-113), so for
// a given class the mismatch planes of window bits 57..63 are the stream planes themselves or their complements -- the
// complement goes into the adders' truth tables instead of costing an XOR per plane (CLS = 0 / 1; -1 = every plane XORed with
// its run-time flip mask as before).
constexpr uint32_t tt3(uint32_t base, bool ia, bool ib, bool ic)
{
	uint32_t t = 0;
	for (uint32_t idx = 0; idx < 8; idx++) {
		const uint32_t a = ((idx >> 2) & 1) ^ (ia ? 1u : 0u), b = ((idx >> 1) & 1) ^ (ib ? 1u : 0u), c = (idx & 1) ^ (ic ? 1u : 0u);
		t |= ((base >> (a * 4 + b * 2 + c)) & 1) << idx;
	}
	return t;
}
// sync-word bit 57 + j of class CLS: 0x27 = 0100111b for LAP MSB 1, its complement for 0 (BARKER1 / BARKER0, common.h)
constexpr bool barker_bit(int cls, int j) { return (((cls ? BARKER1 : BARKER0) >> j) & 1) != 0; }
// (the truth table of v_bitop3 is an immediate: it has to reach the builtin as a template constant)
template <uint32_t TT>
__device__ __forceinline__ uint32_t bitop3_tt(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TT); }

// The planes of the known-LAP filters (round 6, late): the filter may count mismatches in ANY subset of the sync word's bits, so it
// takes them where the funnel shifts can be shared -- window bits 24 + j and 56 + j (j = 0 .. 7) of the offsets p of a 32-offset half
// are the stream bits p + 24 + j of two neighbouring dword pairs, and the UPPER planes of one half are the LOWER planes of the next:
// three sets of eight shifts per word instead of four (sixteen top bits per half: 32 v_alignbit per word -> 24).  Bits 57 .. 63 are
// still the class bits whose complement folds into the adders' truth tables.
// P[j] = stream bit p + 24 + j of hi:lo for the 32 offsets p of a half (j = FIRST .. 7)
template <int FIRST>
__device__ __forceinline__ void pair_planes(uint32_t lo, uint32_t hi, uint32_t *P)
{
#pragma unroll
	for (int j = FIRST; j < 8; j++)
		P[j] = alignbit(hi, lo, 24 + j);
}

// bit-sliced "mismatches in sync-word bits 28..31 and 56..63 <= limit" for 32 offsets: twelve planes
// (lowp[4 .. 7] = window bits 28 .. 31, highp[0 .. 7] = window bits 56 .. 63; flip[4 + k] = the sync word's bit of plane k),
// a carry-save adder tree to a 4-bit count per offset, and a bit-sliced compare with the run-time limit.  For limit 2 it keeps
// 79 / 4096 = 1.9 % of the offsets of a random stream.
template <int CLS>
__device__ __forceinline__ uint32_t top12_filter(const uint32_t *lowp, const uint32_t *highp, const uint32_t *flip, int limit)
{
	if (limit >= 12)
		return 0xffffffffu;
	uint32_t m[12];
#pragma unroll
	for (int k = 0; k < 12; k++) {                      // plane k: 0 .. 3 = window bits 28 .. 31, 4 = bit 56, 5 .. 11 = the class bits 57 .. 63
		m[k] = k < 4 ? lowp[4 + k] : highp[k - 4];
		if (CLS < 0 || k < 5)
			m[k] ^= flip[4 + k];
	}
	constexpr bool K = CLS >= 0;
#define INV(k) (K && barker_bit(CLS, (k) - 5))
#define FA_SUM(a, b, c) BITOP3((a), (b), (c), 0x96)
#define FA_CARRY(a, b, c) BITOP3((a), (b), (c), 0xe8)
	if (limit == 0) {                                   // no mismatch at all: the OR of the twelve planes (six instructions; third session of round 6)
		const uint32_t r0 = BITOP3(m[0], m[1], m[2], 0xfe);
		const uint32_t r1 = bitop3_tt<tt3(0xfe, false, false, INV(5))>(m[3], m[4], m[5]);
		const uint32_t r2 = bitop3_tt<tt3(0xfe, INV(6), INV(7), INV(8))>(m[6], m[7], m[8]);
		const uint32_t r3 = bitop3_tt<tt3(0xfe, INV(9), INV(10), INV(11))>(m[9], m[10], m[11]);
		return ~(BITOP3(r0, r1, r2, 0xfe) | r3);
	}
	const uint32_t s0 = FA_SUM(m[0], m[1], m[2]), c0 = FA_CARRY(m[0], m[1], m[2]);
	const uint32_t s1 = bitop3_tt<tt3(0x96, false, false, INV(5))>(m[3], m[4], m[5]), c1 = bitop3_tt<tt3(0xe8, false, false, INV(5))>(m[3], m[4], m[5]);
	const uint32_t s2 = bitop3_tt<tt3(0x96, INV(6), INV(7), INV(8))>(m[6], m[7], m[8]), c2 = bitop3_tt<tt3(0xe8, INV(6), INV(7), INV(8))>(m[6], m[7], m[8]);
	const uint32_t s3 = bitop3_tt<tt3(0x96, INV(9), INV(10), INV(11))>(m[9], m[10], m[11]), c3 = bitop3_tt<tt3(0xe8, INV(9), INV(10), INV(11))>(m[9], m[10], m[11]);
#undef INV
	const uint32_t o1 = FA_SUM(s0, s1, s2), k0 = FA_CARRY(s0, s1, s2);
	if (limit == 1) {                                   // count = o1 + s3 + 2 x (c0 .. c3, k0): <= 1 <=> none of those five and not both of o1, s3
		const uint32_t w = BITOP3(c0, c1, c2, 0xfe);
		const uint32_t x = BITOP3(c3, k0, w, 0xfe);
		return ~BITOP3(x, o1, s3, 0xf8);                // ~(x | (o1 & s3))
	}
	const uint32_t ones = o1 ^ s3, k1 = o1 & s3;
	const uint32_t t0 = FA_SUM(c0, c1, c2), f0 = FA_CARRY(c0, c1, c2);
	const uint32_t t1 = FA_SUM(c3, k0, k1), f1 = FA_CARRY(c3, k0, k1);
	const uint32_t twos = t0 ^ t1, f2 = t0 & t1;
	if (limit <= 3) {                                   // (see top16_filter)
		const uint32_t ge4 = BITOP3(f0, f1, f2, 0xfe);
		const uint32_t low = limit == 0 ? (twos | ones) : limit == 1 ? twos : limit == 2 ? (twos & ones) : 0u;
		return ~(ge4 | low);
	}
	const uint32_t fours = FA_SUM(f0, f1, f2), eights = FA_CARRY(f0, f1, f2);
#undef FA_SUM
#undef FA_CARRY
	// count = ones + 2 twos + 4 fours + 8 eights; keep offsets with count <= limit
	uint32_t gt = 0, eq = 0xffffffffu;
	const uint32_t planes[4] = { eights, fours, twos, ones };
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const uint32_t lim_bit = ((limit >> (3 - b)) & 1) ? 0xffffffffu : 0u;
		gt |= eq & planes[b] & ~lim_bit;
		eq &= ~(planes[b] ^ lim_bit);
	}
	return ~gt;
}

// The same over sixteen sync-word bits (24..31 and 56..63): five more adders, but for limit >= 2 it
// leaves a tenth of the survivors (0.2 % instead of 1.9 % at limit 2), which is worth more than it
// costs; for limit <= 1 the twelve-plane filter is already sparse enough and cheaper.
template <int CLS>
__device__ __forceinline__ uint32_t top16_filter(const uint32_t *lowp, const uint32_t *highp, const uint32_t *flip, int limit)
{
	if (limit >= 16)
		return 0xffffffffu;
	uint32_t m[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {                      // plane k: 0 .. 7 = window bits 24 .. 31, 8 = bit 56, 9 .. 15 = the class bits 57 .. 63
		m[k] = k < 8 ? lowp[k] : highp[k - 8];
		if (CLS < 0 || k < 9)
			m[k] ^= flip[k];
	}
	constexpr bool K = CLS >= 0;
#define INV(k) (K && barker_bit(CLS, (k) - 9))
#define FA_SUM(a, b, c) BITOP3((a), (b), (c), 0x96)
#define FA_CARRY(a, b, c) BITOP3((a), (b), (c), 0xe8)
	// weight 1
	const uint32_t s0 = FA_SUM(m[0], m[1], m[2]), c0 = FA_CARRY(m[0], m[1], m[2]);
	const uint32_t s1 = FA_SUM(m[3], m[4], m[5]), c1 = FA_CARRY(m[3], m[4], m[5]);
	const uint32_t s2 = FA_SUM(m[6], m[7], m[8]), c2 = FA_CARRY(m[6], m[7], m[8]);
	const uint32_t s3 = bitop3_tt<tt3(0x96, INV(9), INV(10), INV(11))>(m[9], m[10], m[11]), c3 = bitop3_tt<tt3(0xe8, INV(9), INV(10), INV(11))>(m[9], m[10], m[11]);
	const uint32_t s4 = bitop3_tt<tt3(0x96, INV(12), INV(13), INV(14))>(m[12], m[13], m[14]), c4 = bitop3_tt<tt3(0xe8, INV(12), INV(13), INV(14))>(m[12], m[13], m[14]);
	const uint32_t o1 = FA_SUM(s0, s1, s2), k0 = FA_CARRY(s0, s1, s2);
	const uint32_t o2 = bitop3_tt<tt3(0x96, false, false, INV(15))>(s3, s4, m[15]), k1 = bitop3_tt<tt3(0xe8, false, false, INV(15))>(s3, s4, m[15]);
#undef INV
	// limit 2 or 3 (third session of round 6): count = o1 + o2 + 2 x (bits set among W = c0 .. c4, k0, k1), so
	//   count <= 2  <=>  no bit of W, or exactly one and neither o1 nor o2     = at_most_one(W) & ~(any(W) & (o1 | o2))
	//   count <= 3  <=>  no bit of W, or exactly one and not both o1 and o2    = at_most_one(W) & ~(any(W) & o1 & o2)
	// at_most_one over the groups (c0 c1 c2) (c3 c4 k0) (k1): no group holds two, no two groups hold one -- nine instructions
	// where the twos / fours columns, their carries and the compare took thirteen (27 -> 23 per 32 offsets)
	if (limit == 2 || limit == 3) {
		const uint32_t a0 = BITOP3(c0, c1, c2, 0xfe), t0 = BITOP3(c0, c1, c2, 0xe8);     // any / at least two of a group
		const uint32_t a1 = BITOP3(c3, c4, k0, 0xfe), t1 = BITOP3(c3, c4, k0, 0xe8);
		const uint32_t two_groups = BITOP3(a0, a1, k1, 0xe8);
		const uint32_t any = BITOP3(a0, a1, k1, 0xfe);
		const uint32_t odd = limit == 2 ? BITOP3(any, o1, o2, 0xe0)                       // any & (o1 | o2)
						: BITOP3(any, o1, o2, 0x80);                      // any & o1 & o2
		const uint32_t bad = BITOP3(t0, t1, two_groups, 0xfe);
		return ~(bad | odd);
	}
	const uint32_t ones = o1 ^ o2, k2 = o1 & o2;
	// weight 2: c0..c4, k0, k1, k2
	const uint32_t t0 = FA_SUM(c0, c1, c2), f0 = FA_CARRY(c0, c1, c2);
	const uint32_t t1 = FA_SUM(c3, c4, k0), f1 = FA_CARRY(c3, c4, k0);
	const uint32_t t2 = FA_SUM(k1, k2, t0), f2 = FA_CARRY(k1, k2, t0);
	const uint32_t twos = t1 ^ t2, f3 = t1 & t2;
	// limit <= 3: "count >= 4" is all that matters of the upper weights, and it is the OR of the four carries out of the
	// twos column -- six adder instructions and the compare become three (the compiler cannot find this: it is not the
	// same function as the sum it replaces)
	if (limit <= 3) {
		const uint32_t ge4 = BITOP3(f0, f1, f2, 0xfe);
		const uint32_t low = limit == 0 ? (twos | ones) : limit == 1 ? twos : limit == 2 ? (twos & ones) : 0u;
		return ~BITOP3(ge4, f3, low, 0xfe);
	}
	// weight 4: f0..f3
	const uint32_t g0 = FA_SUM(f0, f1, f2), h0 = FA_CARRY(f0, f1, f2);
	const uint32_t fours = g0 ^ f3, h1 = g0 & f3;
	// weight 8, 16
	const uint32_t eights = h0 ^ h1, sixteens = h0 & h1;
#undef FA_SUM
#undef FA_CARRY
	// count = ones + 2 twos + 4 fours + 8 eights + 16 sixteens; keep offsets with count <= limit
	uint32_t gt = sixteens, eq = ~sixteens;
	const uint32_t planes[4] = { eights, fours, twos, ones };
#pragma unroll
	for (int b = 0; b < 4; b++) {
		const uint32_t lim_bit = ((limit >> (3 - b)) & 1) ? 0xffffffffu : 0u;
		gt |= eq & planes[b] & ~lim_bit;
		eq &= ~(planes[b] ^ lim_bit);
	}
	return ~gt;
}

// Known-LAP hits are staged in a per-wave LDS ring and flushed 64 at a time: one global
// counter atomic per 64 hits (a single counter word saturates near 88 M atomics/s on this
// chip, which a dense hit stream would otherwise run into).
#define KRING 128
#ifndef KL_WORDS
#define KL_WORDS 2                             // consecutive stream words per lane and tile (tile = KL_WORDS x 256 words; a power of two); the
#endif                                         // next tile's words are loaded while this one is worked on (0.466 against 0.4865 ms, round 3)
#define KL_SELECT_LIMIT 1                      // limits up to here: one survivor per lane and pass (scan_known_lap_kernel)
struct KnownHit { uint32_t off_lo, off_hi, stream_err; };      // 12 bytes per staged hit

// LIMIT = max_ac_errors when it is 0 .. 4 (the count <= limit compare of the filters then folds into a few
// and / andn of the count planes; with the limit in a register it is sixteen instructions with SGPR masks), -1 = any
// CLS = bit 23 of the LAP (the barker class of its sync word), -1 = not specialised
// ORD: the ordered scan's form -- hits leave through the segment slots (a template flag: the code that fills them costs the plain form
// eight registers, one wave per SIMD, if it is only branched around)
template <int LIMIT, int CLS, bool MSB, bool ORD = false>
// (round 6: the ORD form at 65 VGPRs = seven waves per SIMD; forced to 64 / eight by amdgpu_waves_per_eu: no difference, 0.540-0.542 against 0.538-0.544 ms per chain step)
__global__ __launch_bounds__(256) void scan_known_lap_kernel(ScanArgs a)
{
	__shared__ KnownHit ring_mem[4][KRING];
	__shared__ uint32_t slot_cnt[4][64];                   // (ordered scan, segment slots: hits per tile tag of a batch ...
	__shared__ uint16_t slot_code[4][64][4];               //  ... and up to four of their 12-bit offsets inside the segment)
	if (a.gate && *a.gate == 0)
		return;
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63;
	KnownHit *ring = ring_mem[tid >> 6];
	constexpr bool ord = ORD;
	uint32_t ac_lo = (uint32_t)a.syncword, ac_hi = (uint32_t)(a.syncword >> 32);
	asm volatile("" : "+v"(ac_lo), "+v"(ac_hi));          // (an SGPR operand halves the issue rate of the XORs in the survivor pass)
	// the planes of the filter are XORed with all-ones where the sync word has a 1: sixteen masks, kept in
	// VGPRs on purpose -- they are wave-uniform, and a VALU instruction with an SGPR source issues at half rate
	// (tools/valu_rate.hip: 4.2 against 2.5 cycles)
	uint32_t flip[16];
#pragma unroll
	for (int k = 0; k < 16; k++) {
		flip[k] = (((k < 8 ? ac_lo : ac_hi) >> (24 + (k & 7))) & 1) ? 0xffffffffu : 0u;   // plane k = sync-word bit 24 + k (k < 8), 48 + k (k >= 8)
		asm volatile("" : "+v"(flip[k]));
	}
	const int limit = LIMIT >= 0 ? LIMIT : (a.max_err < 0 ? -1 : a.max_err);
	if (limit < 0)
		return;
	const bool wide = limit >= 2;               // launch-uniform choice of the pre-filter
#ifdef SCAN_PROFILE
	// phases: 0 = wait for the tile's words, 1 = bit-sliced filter, 2 = survivor passes + hit staging, 3 = ring flush + tile cursor,
	// 4 = issuing the next tile's loads
	__shared__ uint32_t kl_prof[4][32];
	const uint32_t prof_off = (uint32_t)(uintptr_t)(lds_u32_t *)&kl_prof[tid >> 6][0];
	if (lane < 32)
		kl_prof[tid >> 6][lane] = 0;
	uint64_t prof_t;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_t) : : "memory");
#endif
	uint32_t q_head = 0, q_tail = 0;                // wave-uniform, free running

	auto flush = [&](uint32_t n) {                  // n <= 64 oldest entries -> global hit list
		uint32_t base = 0;
		if (lane == 0)
			base = atomicAdd(a.hit_count, n);
		base = __builtin_amdgcn_readfirstlane(base);
		if (lane < n) {
			const KnownHit k = ring[(q_head + lane) & (KRING - 1)];
			const uint32_t idx = base + lane;
			if (idx < a.hit_cap) {
				btbbx_hit h;
				h.offset = ((uint64_t)k.off_hi << 32) | k.off_lo;
				h.lap = a.lap;
				h.ac_errors = (uint8_t)(k.stream_err & 0xff);
				h.reserved = 0;
				h.stream = (uint16_t)(k.stream_err >> 8);
				a.hits[idx] = h;
				count_bucket(a, h.stream, h.offset);
			}
		}
		q_head += n;
	};
	// Ordered scan (round 6, as in scan_slide_kernel<..., ORD>): a SEGMENT = 4096 offsets = the 64 words of a tile one wave owns
	// (a tile is 2 x 256 words: two segments per wave).  Hits wait in the ring as before, but leave it at a tile end only -- every
	// hit of a segment is then in the batch --, ranked by offset inside their segment, into the segment's own slots.
	uint32_t iter = 0, ring_first_iter = 0;         // wave-uniform: tiles this wave has worked on; the tile of the oldest ring entry
	auto stage = [&](bool hit, uint32_t stream, uint64_t offset, uint32_t nerr) {
		const uint64_t mask = __ballot(hit);
		if (!mask)
			return;
		if (a.first) {                              // first-match mode: atomicMin, hits are sparse
			if (hit)
				emit_hit(a, stream, offset, a.lap, nerr);
			return;
		}
		if (q_tail - q_head + 64 > KRING) {
			if (ord) {                              // more than 64 hits in a wave's tile(s): a stream of sync words -- the general ordering redoes the call
				*a.irregular = 1u;
				return;
			}
			flush(64);
		}
		if (q_tail == q_head)
			ring_first_iter = iter;
		if (hit) {
			const uint32_t slot = q_tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
					__builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
			// (ordered scan: bits 24 .. 31 = the segment's tag inside the batch, KL_WORDS per tile: a wave's run of a tile starts at a
			// multiple of 64 KL_WORDS words, so the segment number's low bits tell which)
			const uint32_t tag = ((iter * KL_WORDS) | ((uint32_t)(offset >> 12) & (KL_WORDS - 1u))) & 0xffu;
			KnownHit k = { (uint32_t)offset, (uint32_t)(offset >> 32), (stream << 8) | nerr | (ord ? tag << 24 : 0u) };
			ring[slot & (KRING - 1)] = k;
		}
		q_tail += (uint32_t)__popcll(mask);
	};
	auto to_slots = [&](bool final) {               // at a tile end: the whole ring (<= 128 entries) into the segment slots
		const uint32_t n = q_tail - q_head;
		if (n == 0 || (n < 48 && !final))
			return;
		uint32_t *cnt = slot_cnt[tid >> 6];
		uint16_t (*codes)[4] = slot_code[tid >> 6];
		const bool tags_ok = iter - ring_first_iter < 64 / KL_WORDS;      // KL_WORDS tags per tile, 64 counters: no two segments of the batch share one
		// (one round of 64 entries at a time and nothing kept between the rounds: the kernel's 64 registers are its eight waves per SIMD)
		bool fast = tags_ok;
		if (tags_ok) {
			cnt[lane] = 0;
#pragma unroll 1
			for (uint32_t r = 0; r < n; r += 64)
				if (r + lane < n) {
					const KnownHit e = ring[(q_head + r + lane) & (KRING - 1)];
					const uint32_t key = (e.stream_err >> 24) & 63u;
					const uint32_t idx = atomicAdd(&cnt[key], 1u);
					if (idx < 4)
						codes[key][idx] = (uint16_t)(e.off_lo & 0xfffu);
				}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			if (__ballot(*(volatile __attribute__((address_space(3))) const uint32_t *)&cnt[lane] > 4u))
				fast = false;
		}
#pragma unroll 1
		for (uint32_t r = 0; r < n; r += 64) {
			const bool have = r + lane < n;
			const KnownHit e = ring[(q_head + r + lane) & (KRING - 1)];
			uint32_t count = 0, rank = 0;
			if (fast) {
				if (have) {
					const uint32_t key = (e.stream_err >> 24) & 63u, mine = e.off_lo & 0xfffu;
					count = *(volatile __attribute__((address_space(3))) const uint32_t *)&cnt[key];
					for (uint32_t j = 0; j < count; j++)
						rank += codes[key][j] < mine ? 1u : 0u;
				}
			} else {                                // a sparse stream (a batch over 32 tiles or more) or a crowded segment: every entry against every other
#pragma unroll 1
				for (uint32_t j = 0; j < n; j++) {
					const KnownHit o = ring[(q_head + j) & (KRING - 1)];          // (wave-uniform address: a broadcast)
					const bool same = ((o.stream_err ^ e.stream_err) & 0xffff00u) == 0 && o.off_hi == e.off_hi && (o.off_lo >> 12) == (e.off_lo >> 12);
					count += same ? 1u : 0u;
					rank += same && o.off_lo < e.off_lo ? 1u : 0u;
				}
			}
			const uint32_t stream = (e.stream_err >> 8) & 0xffffu;
			const uint32_t seg = stream * a.segs_per_stream + (uint32_t)((((uint64_t)e.off_hi << 32) | e.off_lo) >> 12);
			uint4 out;
			out.x = e.off_lo;
			out.y = e.off_hi;
			out.z = a.lap;
			out.w = (e.stream_err & 0xffu) | (stream << 16);
			const bool spill = have && rank >= a.seg_slot_n;
			if (have && !spill)
				a.seg_slots[(uint64_t)seg * a.seg_slot_n + rank] = (uint64_t)(e.off_lo & 0xfffu) | ((uint64_t)(a.lap & 0xffffffu) << 12) | ((uint64_t)(e.stream_err & 0xffu) << 36);
			if (have && rank + 1 == count)
				a.seg_cnt[seg] = (uint16_t)count;
			const uint64_t om = __ballot(spill);
			if (om) {
				uint32_t base = 0;
				if (lane == 0)
					base = atomicAdd(a.ovf_count, (uint32_t)__popcll(om));
				base = __builtin_amdgcn_readfirstlane(base);
				const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0));
				if (spill) {
					if (at < a.ovf_cap) {
						reinterpret_cast<uint4 *>(a.ovf_recs)[at] = out;
						a.ovf_meta[at] = make_uint2(seg, rank);
					} else {
						*a.irregular = 1u;
					}
				}
			}
		}
		q_head += n;
	};

	// division-free (stream, tile) cursor, as in the LAP_ANY kernel
	// (32-bit tile numbers: the launcher refuses more; 64-bit compares of wave-uniform values would run on the VALU)
	const uint32_t tiles_per_stream = (uint32_t)a.tiles_per_stream;
	uint32_t stream = 0;
	uint32_t t = blockIdx.x;
	while (t >= tiles_per_stream && stream < a.n_streams) {
		t -= tiles_per_stream;
		stream++;
	}
	// A tile is KL_WORDS x 256 words and a lane owns KL_WORDS CONSECUTIVE words of it (round 6, late; rounds 1-5: words 256 apart),
	// so the filter's planes are shared all along the lane's run of 2 * KL_WORDS halves: 8 x (2 * KL_WORDS + 1) funnel shifts per
	// tile instead of 8 x 3 x KL_WORDS, and one halo word per lane instead of one per word.  The next tile's words are loaded while
	// this one is worked on: the counters had 43 % of the wave-cycles in s_waitcnt with eight waves per SIMD taking turns at their
	// loads (profiles/r03_chain/pmc_known_before.json).
	constexpr int NCH = 2 * KL_WORDS;                   // chains (32-offset halves) per lane and tile
	const uint32_t lw = tid * KL_WORDS;                 // the lane's first word in a tile
	uint64_t nw[KL_WORDS + 1];                          // the lane's words of the next tile and the word behind them
	static_assert(KL_WORDS == 2, "fetch: one 16-byte and one 8-byte buffer load per lane");
	const uint32_t lw_bytes = lw * 8u;
	auto fetch = [&](uint32_t ft, uint32_t fstream) {
		// The lane's run of the next tile through a buffer descriptor over the tile (as scan_slide_kernel's load_pair: the hardware's
		// range check returns zero for the words behind the stream's end): no zero-initialised registers, no exec masks.
		uint32_t bytes = 0;                             // wave-uniform
		const uint64_t *tp = a.words;
		if (fstream < a.n_streams) {
			tp = a.words + (uint64_t)fstream * a.pitch_words + (uint64_t)ft * (KL_WORDS * 256);
			bytes = (KL_WORDS * 256u + 1u) * 8u;         // (a full tile: every word and the halo word are in range)
			if (ft >= a.full_tiles) {
				asm volatile("" ::: "memory");              // (a real branch: flattened, its 64-bit compares are vector instructions of every tile)
				const uint64_t first = (uint64_t)ft * (KL_WORDS * 256);
				const uint64_t left = first < a.n_words ? a.n_words - first : 0;
				bytes = (uint32_t)(left < KL_WORDS * 256u + 1u ? left : KL_WORDS * 256u + 1u) * 8u;
			}
		}
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(tp), 0, (int)bytes, 0x00020000);
		const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lw_bytes, 0, 0);
		const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lw_bytes, 16, 0);
		nw[0] = ((uint64_t)v.y << 32) | v.x;
		nw[1] = ((uint64_t)v.w << 32) | v.z;
		nw[2] = ((uint64_t)w.y << 32) | w.x;
	};
	fetch(t, stream);
	// The first tile's words are waited for HERE: with these loads still counted as pending at the loop head the compiler waits for
	// "everything in flight" (s_waitcnt vmcnt(0)) in front of the filter of EVERY tile -- right behind the next tile's loads, which
	// undid the prefetch (rounds 3-6: 15-25 % of a wave's time in that wait, profiles/r06_known).
#pragma unroll
	for (int u = 0; u <= KL_WORDS; u++)
		asm volatile("" : "+v"(nw[u]));
	while (stream < a.n_streams) {
		// word index and validity of this tile's offsets from the (wave-uniform) tile number: nothing per lane is carried
		// from the fetch but the words themselves.  Chain c = offsets 32 c .. 32 c + 31 of the lane's run; its windows lie in D[c .. c + 2].
		const uint64_t word0 = (uint64_t)t * (KL_WORDS * 256) + lw;
		uint32_t D[NCH + 2], m[NCH];
#pragma unroll
		for (int u = 0; u <= KL_WORDS; u++) {
			D[2 * u] = (uint32_t)nw[u];
			D[2 * u + 1] = (uint32_t)(nw[u] >> 32);
		}
		const bool ragged = t >= a.full_tiles;          // wave-uniform: offsets beyond the search length are cut out BEHIND the filter
		const uint32_t this_stream = stream;
		t += gridDim.x;
		while (t >= tiles_per_stream && stream < a.n_streams) {
			t -= tiles_per_stream;
			stream++;
		}
		fetch(t, stream);
		PROF_MARK(4);
#ifdef SCAN_PROFILE
#pragma unroll
		for (int k = 0; k < NCH + 2; k++)
			asm volatile("" : "+v"(D[k]));                  // this tile's words have arrived
		PROF_MARK(0);
#endif
		__builtin_amdgcn_s_setprio(0);                  // bit-sliced filter: lowest (see PRIO_FILTER above)
		if constexpr (MSB) {
#pragma unroll
			for (int k = 0; k < NCH + 2; k++)
				D[k] = msb_dword(D[k]);
		}
		{	// (pair_planes above: the planes of D[c + 1] : D[c + 2] are the upper planes of chain c and the lower ones of chain c + 1)
			uint32_t P[2][8];
			if (wide)
				pair_planes<0>(D[0], D[1], P[0]);
			else
				pair_planes<4>(D[0], D[1], P[0]);
#pragma unroll
			for (int c = 0; c < NCH; c++) {
				pair_planes<0>(D[c + 1], D[c + 2], P[(c + 1) & 1]);
				m[c] = (wide ? top16_filter<CLS>(P[c & 1], P[(c + 1) & 1], flip, limit)
					     : top12_filter<CLS>(P[c & 1], P[(c + 1) & 1], flip, limit));
			}
		}
		if (ragged) {                                   // (as four validity masks in front of the filter: a register copy and an AND per chain of every tile)
			asm volatile("" ::: "memory");              // (keeps the compiler from flattening the branch into selects)
#pragma unroll
			for (int c = 0; c < NCH; c++) {
				const uint64_t first_off = word0 * 64 + 32u * c;
				m[c] &= first_off >= a.search_bits ? 0u
					: (a.search_bits - first_off >= 32 ? 0xffffffffu : ((1u << (uint32_t)(a.search_bits - first_off)) - 1u));
			}
		}
#ifdef SCAN_PROFILE
#pragma unroll
		for (int c = 0; c < NCH; c++)
			PROF_PIN(m[c]);
		PROF_MARK(1);
#endif
		__builtin_amdgcn_s_setprio(3);                  // survivors, hit staging, flush and the next tile's loads: highest
		// Limits 0 and 1 (few survivors: the filter passes 2.6e-4 / 1.5e-5 of the offsets): ONE survivor per lane and pass -- the
		// next one of whichever chain holds one; a pass that looks at one offset of every chain costs NCH checks
		// for a small fraction of a survivor per lane.  4 GiB at limit 0: 1.85 -> 1.71 ms; at limit 2 nothing (2.52 / 2.50), at
		// limit 4 the lane's survivors queue up (3.14 -> 3.91): the every-chain pass stays for limits of 2 and more.
		if constexpr (LIMIT >= 0 && LIMIT <= KL_SELECT_LIMIT) {
		for (;;) {
			uint32_t mm = m[NCH - 1], da = D[NCH - 1], db = D[NCH], dc = D[NCH + 1], ci = NCH - 1;   // the lane's first chain that holds a survivor
#pragma unroll
			for (int c = NCH - 2; c >= 0; c--) {
				const bool s = m[c] != 0;
				mm = s ? m[c] : mm;
				da = s ? D[c] : da;
				db = s ? D[c + 1] : db;
				dc = s ? D[c + 2] : dc;
				ci = s ? (uint32_t)c : ci;
			}
			if (!__ballot(mm != 0))
				break;
			const uint32_t p1 = lowest_bit(mm);             // (-1 for no survivor: see check() below)
			const int e1 = __popc(alignbit(db, da, p1) ^ ac_lo) + __popc(alignbit(dc, db, p1) ^ ac_hi);          // :433
			const bool hit1 = mm != 0 && e1 <= limit;
			const uint32_t rest = mm & (mm - 1);
#pragma unroll
			for (int c = 0; c < NCH; c++)
				m[c] = ci == (uint32_t)c ? rest : m[c];
			if (__ballot(hit1))
				stage(hit1, this_stream, word0 * 64 + 32u * ci + p1, (uint32_t)e1);
		}
		} else {
		// wave-uniform survivor loop.  First pass: one offset of every chain (a wave's 64 lanes practically always hold a survivor in
		// each of the NCH chains).  Further passes: a chain has a second survivor in some lane in one tile of eight, so a chain that
		// is empty wave-wide is skipped (the ballots are the loop's exit test as well) instead of running its check for nobody.
		uint32_t p[NCH];
		int e[NCH];
		bool hit[NCH];
		auto check = [&](int c) {
			p[c] = lowest_bit(m[c]);                        // (-1 for an empty chain: the funnel shifts below take its low five bits, and `hit` is masked)
			e[c] = __popc(alignbit(D[c + 1], D[c], p[c]) ^ ac_lo)
				+ __popc(alignbit(D[c + 2], D[c + 1], p[c]) ^ ac_hi);          // :433
			hit[c] = m[c] != 0 && e[c] <= limit;
			m[c] &= m[c] - 1;
		};
		{
			uint32_t any = 0;
#pragma unroll
			for (int c = 0; c < NCH; c++)
				any |= m[c];
			if (__ballot(any != 0)) {
				bool anyhit = false;
#pragma unroll
				for (int c = 0; c < NCH; c++) {
					check(c);
					anyhit |= hit[c];
				}
				if (__ballot(anyhit)) {
#pragma unroll
					for (int c = 0; c < NCH; c++)
						stage(hit[c], this_stream, word0 * 64 + 32u * c + p[c], (uint32_t)e[c]);
				}
				for (;;) {
					uint64_t live[NCH], anyl = 0;
#pragma unroll
					for (int c = 0; c < NCH; c++) {
						live[c] = __ballot(m[c] != 0);
						anyl |= live[c];
					}
					if (!anyl)
						break;
#pragma unroll
					for (int c = 0; c < NCH; c++) {
						if (!live[c])
							continue;
						check(c);
						stage(hit[c], this_stream, word0 * 64 + 32u * c + p[c], (uint32_t)e[c]);
					}
				}
			}
		}
		}
		PROF_MARK(2);
		iter++;
		if (ord) {
			to_slots(false);
		} else {
			while (q_tail - q_head >= 64)
				flush(64);
		}
		PROF_MARK(3);
	}
	if (ord) {
		to_slots(true);
	} else if (q_tail != q_head) {
		flush(q_tail - q_head);
	}
#ifdef SCAN_PROFILE
	if (lane < 32)
		atomicAdd(&g_scan_prof[lane], (unsigned long long)kl_prof[tid >> 6][lane]);
#endif
}

// ---- symbol <-> packed conversion ---------------------------------------------------------

// 16 symbols (bit 0 of 16 bytes) -> 16 bits
__device__ __forceinline__ uint32_t gather16(uint4 v)
{
	auto nib = [](uint32_t x) {
		x &= 0x01010101u;
		return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
	};
	return nib(v.x) | (nib(v.y) << 4) | (nib(v.z) << 8) | (nib(v.w) << 12);
}

__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *sym, uint64_t n_sym, uint64_t *words, uint64_t n_words)
{
	// each lane converts 16 symbols; 4 adjacent lanes make one word
	uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	uint64_t n_chunks = n_words * 4;
	for (; chunk < ((n_chunks + 63) & ~63ULL); chunk += stride) {
		uint64_t s0 = chunk * 16;
		uint32_t bits = 0;
		if (s0 + 16 <= n_sym && (((uintptr_t)(sym + s0)) & 15) == 0) {
			bits = gather16(*reinterpret_cast<const uint4 *>(sym + s0));
		} else if (s0 < n_sym) {
			for (uint32_t i = 0; i < 16 && s0 + i < n_sym; i++)
				bits |= (uint32_t)(sym[s0 + i] & 1) << i;
		}
		uint32_t q = threadIdx.x & 3;
		uint64_t part = (uint64_t)bits << (16 * q);
		part |= __shfl_xor(part, 1);
		part |= __shfl_xor(part, 2);
		if (q == 0 && chunk < n_chunks)
			words[chunk >> 2] = part;
	}
}

__global__ __launch_bounds__(256) void unpack_kernel(const uint64_t *words, uint64_t n_sym, uint8_t *sym)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (; i * 8 < n_sym; i += stride) {              // 8 symbols per lane
		uint32_t byte = (uint32_t)(words[i >> 3] >> (8 * (i & 7))) & 0xff;
		uint64_t out = 0;
#pragma unroll
		for (int b = 0; b < 8; b++)
			out |= (uint64_t)((byte >> b) & 1) << (8 * b);
		if (i * 8 + 8 <= n_sym && (((uintptr_t)(sym + i * 8)) & 7) == 0) {
			*reinterpret_cast<uint64_t *>(sym + i * 8) = out;
		} else {
			for (uint32_t b = 0; b < 8 && i * 8 + b < n_sym; b++)
				sym[i * 8 + b] = (uint8_t)(out >> (8 * b));
		}
	}
}

// MSB-first packed bytes (8 symbols per byte, first received symbol in bit 7 -- the order a
// radio front end typically delivers) -> the library's LSB-first words: reverse the bits of
// every byte in place.  brev64 reverses everything, the byte swap puts the bytes back.
__global__ __launch_bounds__(256) void bitrev_bytes_kernel(uint64_t *words, uint64_t n_words)
{
	uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
	for (; i < n_words; i += step)
		words[i] = __builtin_bswap64(__brevll(words[i]));
}

// ---- launchers ----------------------------------------------------------------------------

static int check_scan_args(uint64_t n_words, uint64_t pitch_words, uint32_t n_streams, uint64_t search_bits)
{
	if (n_streams == 0 || n_streams > 65535) {
		set_error("btbbx_scan: n_streams must be 1..65535");
		return BTBBX_E_ARG;
	}
	if (n_streams > 1 && pitch_words < n_words) {
		set_error("btbbx_scan: pitch_words < n_words");
		return BTBBX_E_ARG;
	}
	if (search_bits + 63 > n_words * 64) {
		set_error("btbbx_scan: search_bits + 63 exceeds the stream (%llu > %llu bits)",
			  (unsigned long long)(search_bits + 63), (unsigned long long)(n_words * 64));
		return BTBBX_E_ARG;
	}
	return BTBBX_OK;
}

// geometry of the segment slots for a scan of these streams (sort.hip sizes its scratch from it); false: this scan has no slot form
// (known LAP, or tables for more than two errors)
bool scan_slot_geometry(uint64_t search_bits, uint32_t n_streams, uint32_t lap, uint32_t *segs_per_stream, uint64_t *n_segs)
{
	int table_errors = 0;
	ScanTables t;
	ctx_scan_snapshot(&t, &table_errors);
	const uint64_t search_words = (search_bits + 63) / 64;
	uint64_t per_stream;
	if (lap != BTBBX_LAP_ANY) {                        // known LAP: segments of 4096 offsets, eight per tile of 512 words
		per_stream = (search_words + 256ull * KL_WORDS - 1) / (256ull * KL_WORDS) * (4 * KL_WORDS);
	} else {
		if (table_errors > 2 || !t.slide_bitmap)
			return false;
		const uint64_t tiles = (search_words + SlideGeom<SlideStd>::TILE_WORDS - 1) / SlideGeom<SlideStd>::TILE_WORDS;
		per_stream = tiles * (SlideStd::THREADS / 64);
	}
	const uint64_t total = per_stream * n_streams;
	if (total >= (1ull << 31))
		return false;
	*segs_per_stream = (uint32_t)per_stream;
	*n_segs = total;
	return true;
}

int launch_scan(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
		uint32_t n_streams, uint64_t search_bits, uint32_t lap, int max_ac_errors,
		btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count,
		unsigned long long *d_first, hipStream_t stream, uint32_t *bucket_cnt = nullptr, uint64_t bucket_mul = 0,
		uint32_t bucket_shift = 0, bool msb = false, const ScanSlots *slots = nullptr, const uint32_t *gate = nullptr)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	rc = check_scan_args(n_words, pitch_words, n_streams, search_bits);
	if (rc)
		return rc;
	if (search_bits == 0)
		return BTBBX_OK;
	if ((uintptr_t)d_hits & 15) {
		set_error("btbbx_scan: the hit buffer must be 16-byte aligned (records are written as one 16-byte store)");
		return BTBBX_E_ARG;
	}
	Ctx &c = ctx();
	ScanArgs a;
	a.words = d_words;
	a.n_words = n_words;
	a.pitch_words = pitch_words;
	a.search_bits = search_bits;
	a.n_streams = n_streams;
	a.msb = msb ? 1u : 0u;
	a.lap = lap;
	a.syncword = 0;
	a.max_err = max_ac_errors;
	a.hits = d_hits;
	a.hit_cap = hit_cap;
	a.hit_count = d_hit_count;
	a.first = d_first;
	a.bucket_cnt = bucket_cnt;
	a.bucket_mul = bucket_mul;
	a.bucket_shift = bucket_shift;
	a.seg_slots = nullptr;
	a.seg_cnt = nullptr;
	a.seg_slot_n = 0;
	a.segs_per_stream = 0;
	a.ovf_recs = nullptr;
	a.ovf_meta = nullptr;
	a.ovf_cap = 0;
	a.ovf_count = nullptr;
	a.irregular = nullptr;
	a.gate = gate;
	if (slots) {
		a.seg_slots = slots->slots;
		a.seg_cnt = slots->cnt;
		a.seg_slot_n = slots->slot_n;
		a.segs_per_stream = slots->segs_per_stream;
		a.ovf_recs = slots->ovf_recs;
		a.ovf_meta = reinterpret_cast<uint2 *>(slots->ovf_meta);
		a.ovf_cap = slots->ovf_cap;
		a.ovf_count = slots->ovf_count;
		a.irregular = slots->irregular;
	}
	a.xcd_tiles = 0;
	a.ring_margin = 4;
	int table_errors = 0;
	ctx_scan_snapshot(&a.t, &table_errors);
	const uint64_t search_words = (search_bits + 63) / 64;
	if (lap == BTBBX_LAP_ANY) {
		// tables for <= 2 errors: the sliding-check kernel (1); for three and four: its two-level form (4: a 2^20-bit set in LDS, its members
		// looked up in a second set in L2, slide.h); for five every survivor probes a 2^26-bit bitmap in L2 (8: scan_lap_any_kernel)
		int run_variant = 1;
		if ((table_errors == 3 || table_errors == 4) && a.t.slide4_bitmap)
			run_variant = 4;
		else if (a.t.bitmap2 && table_errors >= 4)
			run_variant = 8;
		const uint32_t tile_words = run_variant == 1 ? SlideGeom<SlideStd>::TILE_WORDS : run_variant == 4 ? SlideGeom<Slide4>::TILE_WORDS : SCAN_THREADS;
		const uint32_t halo_words = run_variant == 8 ? 1 : 2;   // words behind a tile its last lane reads
		a.tiles_per_stream = (search_words + tile_words - 1) / tile_words;
		a.n_tiles = a.tiles_per_stream * n_streams;
		{	// tile t is full iff (t + 1) * tile_words + halo_words <= n_words and (t + 1) * tile_words * 64 <= search_bits
			const uint64_t by_words = n_words >= halo_words ? (n_words - halo_words) / tile_words : 0, by_bits = search_bits / (tile_words * 64ull);
			const uint64_t full = by_words < by_bits ? by_words : by_bits;
			a.full_tiles = full > 0xffffffffull ? 0xffffffffu : (uint32_t)full;
		}
		const uint64_t resident = (uint64_t)c.num_cus * (run_variant == 1 ? SLIDE_WGS : 1);
		uint64_t grid = a.n_tiles < resident ? a.n_tiles : resident;
		a.xcd_tiles = (grid % 8 == 0 && a.n_tiles >= grid) ? (uint32_t)((a.n_tiles + 7) / 8) : 0;
		if ((a.n_tiles + grid - 1) / grid >= (1u << 20) || (a.n_tiles >> 32)) {
			set_error("btbbx_scan: launch too large for the candidate encoding (split the stream)");
			return BTBBX_E_ARG;
		}
#ifdef SCAN_PROFILE
		static unsigned long long zero_prof[32];
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_scan_prof), zero_prof, sizeof(zero_prof)));
#endif
		if (slots && run_variant != 1) {
			set_error("btbbx_scan: internal: segment slots with a kernel that has none");
			return BTBBX_E_ARG;
		}
		switch (run_variant) {
		case 8:
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_lap_any_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS_BYTES));
			hipLaunchKernelGGL(scan_lap_any_kernel, dim3((uint32_t)grid), dim3(SCAN_THREADS), SCAN_LDS_BYTES, stream, a);
			break;
		case 4: {
			a.ring_margin = 24u;
			constexpr uint32_t lds_bytes = SlideGeom<Slide4>::LDS_BYTES;
#define LAUNCH_SLIDE4(MSB_) do { \
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_slide_kernel<Slide4, SLIDE4_TILES, MSB_>), \
						    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
			hipLaunchKernelGGL((scan_slide_kernel<Slide4, SLIDE4_TILES, MSB_>), dim3((uint32_t)grid), dim3(Slide4::THREADS), lds_bytes, stream, a); } while (0)
			if (msb) LAUNCH_SLIDE4(true); else LAUNCH_SLIDE4(false);
#undef LAUNCH_SLIDE4
			break;
		}
		case 1: {
			a.ring_margin = 24u;
			constexpr uint32_t lds_bytes = SlideGeom<SlideStd>::LDS_BYTES;
#define LAUNCH_SLIDE(MSB_, ORD_) do { \
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_slide_kernel<SlideStd, SLIDE_TILES, MSB_, ORD_>), \
						    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
			hipLaunchKernelGGL((scan_slide_kernel<SlideStd, SLIDE_TILES, MSB_, ORD_>), dim3((uint32_t)grid), dim3(SLIDE_THREADS), lds_bytes, stream, a); } while (0)
			if (table_errors >= 3) {
				set_error("btbbx_scan: internal: the tables for %d errors lack their second-level set", table_errors);
				return BTBBX_E_ARG;
			}
			if (slots) {
				if (a.segs_per_stream != a.tiles_per_stream * (SLIDE_THREADS / 64) || d_first) {
					set_error("btbbx_scan: internal: segment slots laid out for another geometry");
					return BTBBX_E_ARG;
				}
				if (msb) LAUNCH_SLIDE(true, true); else LAUNCH_SLIDE(false, true);
			} else {
				if (msb) LAUNCH_SLIDE(true, false); else LAUNCH_SLIDE(false, false);
			}
#undef LAUNCH_SLIDE
			break;
		}
		default: set_error("btbbx_scan: internal: no LAP_ANY kernel for this table set"); return BTBBX_E_ARG;
		}
#ifdef SCAN_PROFILE
		{
			unsigned long long prof[32], total = 0;
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_scan_prof), sizeof(prof)));
			for (int k = 0; k < 32; k++) total += prof[k];
			fprintf(stderr, "scan profile (%% of wave time):");
			for (int k = 0; k < 20; k++) fprintf(stderr, " %d:%.1f", k, 100.0 * (double)prof[k] / (double)(total ? total : 1));
			fprintf(stderr, "\n");
		}
#endif
	} else {
		a.syncword = host_gen_syncword(lap & 0xffffff);
		a.lap = lap;
		const uint64_t kl_tile = 256ull * KL_WORDS;
		a.tiles_per_stream = (search_words + kl_tile - 1) / kl_tile;
		a.n_tiles = a.tiles_per_stream * n_streams;
		{	// tile t (kl_tile words) is full iff (t + 1) * kl_tile + 1 <= n_words and (t + 1) * kl_tile * 64 <= search_bits
			const uint64_t by_words = n_words ? (n_words - 1) / kl_tile : 0, by_bits = search_bits / (kl_tile * 64ull);
			const uint64_t full = by_words < by_bits ? by_words : by_bits;
			a.full_tiles = full > 0xffffffffull ? 0xffffffffu : (uint32_t)full;
		}
		uint64_t cap = (uint64_t)c.num_cus * 8;
		uint64_t grid = a.n_tiles < cap ? a.n_tiles : cap;
		if (a.tiles_per_stream + grid >= (1ull << 32)) {
			set_error("btbbx_scan: stream too long for one launch (split it)");
			return BTBBX_E_ARG;
		}
		if (slots && (a.segs_per_stream != a.tiles_per_stream * (4 * KL_WORDS) || d_first)) {
			set_error("btbbx_scan: internal: segment slots laid out for another geometry");
			return BTBBX_E_ARG;
		}
		const bool cls1 = ((a.syncword >> 57) & 1) != 0;          // = bit 23 of the LAP
#define LAUNCH_KNOWN__(L, C_, M_, O_) hipLaunchKernelGGL((scan_known_lap_kernel<L, C_, M_, O_>), dim3((uint32_t)grid), dim3(256), 0, stream, a)
#define LAUNCH_KNOWN_(L, C_, M_) do { if (slots) LAUNCH_KNOWN__(L, C_, M_, true); else LAUNCH_KNOWN__(L, C_, M_, false); } while (0)
#define LAUNCH_KNOWN(L) do { if (cls1) { if (msb) LAUNCH_KNOWN_(L, 1, true); else LAUNCH_KNOWN_(L, 1, false); } \
		else { if (msb) LAUNCH_KNOWN_(L, 0, true); else LAUNCH_KNOWN_(L, 0, false); } } while (0)
		switch (max_ac_errors) {
		case 0: LAUNCH_KNOWN(0); break;
		case 1: LAUNCH_KNOWN(1); break;
		case 2: LAUNCH_KNOWN(2); break;
		case 3: LAUNCH_KNOWN(3); break;
		case 4: LAUNCH_KNOWN(4); break;
		default: LAUNCH_KNOWN(-1); break;
		}
#undef LAUNCH_KNOWN
#undef LAUNCH_KNOWN_
#undef LAUNCH_KNOWN__
#ifdef SCAN_PROFILE
		{
			unsigned long long prof[32], total = 0;
			HIP_TRY(hipDeviceSynchronize());
			HIP_TRY(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_scan_prof), sizeof(prof)));
			for (int k = 0; k < 32; k++) total += prof[k];
			fprintf(stderr, "known-LAP profile (%% of wave time; cumulative over launches):");
			for (int k = 0; k < 5; k++) fprintf(stderr, " %d:%.1f", k, 100.0 * (double)prof[k] / (double)(total ? total : 1));
			fprintf(stderr, "\n");
		}
#endif
	}
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_scan_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
				 uint32_t n_streams, uint64_t search_bits, uint32_t lap, int max_ac_errors,
				 btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count, void *hip_stream)
{
	if (!d_words || !d_hit_count || (!d_hits && hit_cap)) {
		set_error("btbbx_scan_device: null pointer");
		return BTBBX_E_ARG;
	}
	return launch_scan(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors,
			   d_hits, hit_cap, d_hit_count, nullptr, (hipStream_t)hip_stream);
}

extern "C" int btbbx_scan_device_fmt(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
				     uint32_t n_streams, uint64_t search_bits, uint32_t lap, int max_ac_errors, int format,
				     btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count, void *hip_stream)
{
	if (!d_words || !d_hit_count || (!d_hits && hit_cap) || (format != BTBBX_FMT_PACKED && format != BTBBX_FMT_PACKED_MSB)) {
		set_error("btbbx_scan_device_fmt: null pointer or a format that is not BTBBX_FMT_PACKED / BTBBX_FMT_PACKED_MSB");
		return BTBBX_E_ARG;
	}
	return launch_scan(d_words, n_words, pitch_words, n_streams, search_bits, lap, max_ac_errors,
			   d_hits, hit_cap, d_hit_count, nullptr, (hipStream_t)hip_stream, nullptr, 0, 0, format == BTBBX_FMT_PACKED_MSB);
}

extern "C" int btbbx_scan_first_device(const uint64_t *d_words, uint64_t n_words, uint64_t search_bits,
				       uint32_t lap, int max_ac_errors, uint64_t *d_first, void *hip_stream)
{
	if (!d_words || !d_first || search_bits >= (1ULL << 32)) {
		set_error("btbbx_scan_first_device: bad argument");
		return BTBBX_E_ARG;
	}
	return launch_scan(d_words, n_words, n_words, 1, search_bits, lap, max_ac_errors,
			   nullptr, 0, nullptr, reinterpret_cast<unsigned long long *>(d_first),
			   (hipStream_t)hip_stream);
}

extern "C" int btbbx_pack_device(const uint8_t *d_symbols, uint64_t n_symbols, uint64_t *d_words, void *hip_stream)
{
	if (n_symbols == 0)
		return BTBBX_OK;
	uint64_t n_words = (n_symbols + 63) / 64;
	uint64_t threads = n_words * 4;
	uint64_t blocks = (threads + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(pack_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream,
			   d_symbols, n_symbols, d_words, n_words);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_msb_to_lsb_device(uint64_t *d_words, uint64_t n_words, void *hip_stream)
{
	if (n_words == 0)
		return BTBBX_OK;
	uint64_t blocks = (n_words + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(bitrev_bytes_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream, d_words, n_words);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_unpack_device(const uint64_t *d_words, uint64_t n_symbols, uint8_t *d_symbols, void *hip_stream)
{
	if (n_symbols == 0)
		return BTBBX_OK;
	uint64_t threads = (n_symbols + 7) / 8;
	uint64_t blocks = (threads + 255) / 256;
	if (blocks > 65536) blocks = 65536;
	hipLaunchKernelGGL(unpack_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)hip_stream,
			   d_words, n_symbols, d_symbols);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

// ---- host convenience wrappers ---------------------------------------------------------------

#include <algorithm>
#include <vector>

// (stream, offset) order.  Large lists (a 1 GiB capture yields ~10^6 hits) go through an LSD radix
// sort with 16-bit digits on the key stream << 48 | offset, skipping digits that are equal in all
// keys -- three passes for a 4 GiB stream instead of std::sort's ~20 n comparisons, which used to
// be most of the PCIe-inclusive time of the streaming ingest.
extern "C" void btbbx_sort_hits(btbbx_hit *hits, size_t n)
{
	auto key = [](const btbbx_hit &h) { return ((uint64_t)h.stream << 48) | (h.offset & 0xffffffffffffULL); };
	bool small_offsets = true;
	uint64_t all_or = 0, all_and = ~0ULL;
	for (size_t i = 0; i < n; i++) {
		small_offsets &= (hits[i].offset >> 48) == 0;
		const uint64_t k = key(hits[i]);
		all_or |= k;
		all_and &= k;
	}
	if (n < 4096 || !small_offsets) {
		std::sort(hits, hits + n, [](const btbbx_hit &x, const btbbx_hit &y) {
			if (x.stream != y.stream) return x.stream < y.stream;
			return x.offset < y.offset;
		});
		return;
	}
	std::vector<btbbx_hit> tmp(n);
	std::vector<size_t> count(65536);
	btbbx_hit *src = hits, *dst = tmp.data();
	for (int shift = 0; shift < 64; shift += 16) {
		if ((((all_or ^ all_and) >> shift) & 0xffff) == 0)
			continue;                       // this digit is the same in every key
		std::fill(count.begin(), count.end(), 0);
		for (size_t i = 0; i < n; i++)
			count[(key(src[i]) >> shift) & 0xffff]++;
		size_t run = 0;
		for (size_t d = 0; d < 65536; d++) {
			const size_t c = count[d];
			count[d] = run;
			run += c;
		}
		for (size_t i = 0; i < n; i++)
			dst[count[(key(src[i]) >> shift) & 0xffff]++] = src[i];
		std::swap(src, dst);
	}
	if (src != hits)
		memcpy(hits, src, n * sizeof(btbbx_hit));
}

// Scan words already on the current device and bring the hits back in (stream, offset) order.
// `cap` limits what is WRITTEN, never what is found: when more offsets match than the device buffer
// of the first pass holds, the scan is repeated with a buffer of the size the counter reported, so
// that the records handed back are always the `cap` SMALLEST (stream, offset) ones -- a caller asking
// for one hit gets the first match, as btbb_find_ac would return it (bluetooth_packet.c:444-464).
static int64_t scan_resident(const uint64_t *d_words, uint64_t n_words, uint64_t search_bits, uint32_t lap,
			     int max_ac_errors, btbbx_hit *hits, uint64_t cap, uint64_t offset_base, hipStream_t q)
{
	// counter + records live in one grow-only block of the call's lease (context.cpp scope_hits): no allocation
	// in steady state.  Layout: 16 bytes for the counter, then the records (16-byte aligned).
	struct Dev {
		btbbx_hit *hits = nullptr;
		uint32_t *count = nullptr;
	} d;
	// first guess: room for what the caller can take, but no more than one hit per 256 offsets + slack
	uint64_t guess = search_bits / 256 + 4096;
	if (guess > cap)
		guess = cap;
	uint32_t dev_cap = guess > 0xffffffffULL ? 0xffffffffu : (uint32_t)guess;
	uint32_t count = 0;
	for (int pass = 0; pass < 2; pass++) {
		// counter, records and the ordering's scratch in ONE block of the call's own lease: the list comes back from the
		// scan in (stream, offset) order on the call's private stream -- nothing shared with other callers, no lock, no
		// allocation in steady state (round 3 ordered through btbbx_sort_hits_device's per-device scratch and its mutex)
		const size_t rec_bytes = ((size_t)dev_cap * sizeof(btbbx_hit) + 255) & ~(size_t)255;
		const size_t order_bytes = dev_cap >= 2 ? btbbx_scan_ordered_scratch_bytes(search_bits, 1, lap, dev_cap) : 0;   // (segment slots where the scan has them)
		char *block = (char *)scope_hits(256 + rec_bytes + order_bytes);
		if (!block)
			return BTBBX_E_NOMEM;
		d.count = (uint32_t *)block;
		d.hits = (btbbx_hit *)(block + 256);
		HIP_TRY(hipMemsetAsync(d.count, 0, sizeof(uint32_t), q));
		int rc = dev_cap >= 2 && search_bits
			? btbbx_scan_ordered_device(d_words, n_words, n_words, 1, search_bits, lap, max_ac_errors, d.hits, dev_cap, d.count,
						    block + 256 + rec_bytes, order_bytes, q)
			: btbbx_scan_device(d_words, n_words, n_words, 1, search_bits, lap, max_ac_errors, d.hits, dev_cap, d.count, q);
		if (rc)
			return rc;
		HIP_TRY(hipMemcpyAsync(&count, d.count, sizeof(count), hipMemcpyDeviceToHost, q));
		HIP_TRY(hipStreamSynchronize(q));
		if (count <= dev_cap || cap == 0)
			break;
		// more matches than records kept, and the kept ones are whichever lanes came first: repeat
		// with room for all of them, then keep the smallest
		dev_cap = count;
	}
	const uint32_t have = count < dev_cap ? count : dev_cap;
	if (have) {
		const uint64_t n = have < cap ? have : cap;
		HIP_TRY(hipMemcpyAsync(hits, d.hits, (size_t)n * sizeof(btbbx_hit), hipMemcpyDeviceToHost, q));
		HIP_TRY(hipStreamSynchronize(q));
		if (offset_base)
			for (uint64_t i = 0; i < n; i++)
				hits[i].offset += offset_base;
	}
	return (int64_t)count;
}

extern "C" int64_t btbbx_scan_host(const uint64_t *words, uint64_t n_words, uint64_t search_bits,
				   uint32_t lap, int max_ac_errors, btbbx_hit *hits, uint64_t cap)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	rc = check_scan_args(n_words, n_words, 1, search_bits);
	if (rc)
		return rc;
	CallScope scope;
	hipStream_t q = scope_stream();
	uint64_t *d_words = (uint64_t *)scope_device((n_words + 2) * 8);
	if (!d_words)
		return BTBBX_E_NOMEM;
	HIP_TRY(hipMemcpyAsync(d_words, words, n_words * 8, hipMemcpyHostToDevice, q));
	return scan_resident(d_words, n_words, search_bits, lap, max_ac_errors, hits, cap, 0, q);
}

extern "C" int64_t btbbx_scan_symbols(const char *symbols, uint64_t n_symbols, uint64_t search_length,
				      uint32_t lap, int max_ac_errors, btbbx_hit *hits, uint64_t cap)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (search_length + 63 > n_symbols) {
		set_error("btbbx_scan_symbols: search_length + 63 exceeds n_symbols");
		return BTBBX_E_ARG;
	}
	CallScope scope;
	hipStream_t q = scope_stream();
	uint64_t n_words = (n_symbols + 63) / 64;
	size_t sym_bytes = (n_symbols + 15) & ~15ULL;
	char *block = (char *)scope_device(sym_bytes + (n_words + 2) * 8);
	if (!block)
		return BTBBX_E_NOMEM;
	uint8_t *d_sym = (uint8_t *)block;
	uint64_t *d_words = (uint64_t *)(block + sym_bytes);
	HIP_TRY(hipMemcpyAsync(d_sym, symbols, n_symbols, hipMemcpyHostToDevice, q));
	rc = btbbx_pack_device(d_sym, n_symbols, d_words, q);
	if (rc)
		return rc;
	return scan_resident(d_words, n_words, search_length, lap, max_ac_errors, hits, cap, 0, q);
}

// First match of one symbol-per-byte buffer (what btbb_find_ac returns, bluetooth_packet.c:444-464):
// one pinned staging copy in, pack + scan (atomicMin over offset << 32 | lap << 8 | errors) queued
// behind it, 8 bytes back, one synchronisation.
extern "C" int btbbx_find_first_symbols(const char *symbols, uint64_t n_symbols, uint64_t search_length,
					uint32_t lap, int max_ac_errors, btbbx_hit *first_hit)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!symbols || !first_hit || search_length + 63 > n_symbols || search_length >= (1ULL << 32)) {
		set_error("btbbx_find_first_symbols: bad argument (search_length + 63 must not exceed n_symbols, search_length < 2^32)");
		return BTBBX_E_ARG;
	}
	if (search_length == 0)
		return 0;
	CallScope scope;                                  // private scratch + stream: callers may be concurrent
	hipStream_t q = scope_stream();
	const uint64_t n_sym = search_length + 63;            // last symbol the reference reads
	const uint64_t n_words = (n_sym + 63) / 64;
	const size_t sym_bytes = (n_sym + 15) & ~15ULL;
	// Device block: symbols | sentinel for the first-match word | packed words.  The sentinel sits
	// right behind the symbols so that ONE host-to-device copy from pinned staging brings both in.
	char *block = (char *)scope_device(sym_bytes + (n_words + 2) * 8 + 16);
	char *stage = (char *)scope_pinned(sym_bytes + 16);
	if (!block || !stage)
		return BTBBX_E_NOMEM;
	uint8_t *d_sym = (uint8_t *)block;
	uint64_t *d_first = (uint64_t *)(block + sym_bytes);
	uint64_t *d_words = d_first + 1;
	uint64_t first = ~0ULL;
	memcpy(stage, symbols, n_sym);
	memcpy(stage + sym_bytes, &first, 8);
	HIP_TRY(hipMemcpyAsync(d_sym, stage, sym_bytes + 8, hipMemcpyHostToDevice, q));
	rc = btbbx_pack_device(d_sym, n_sym, d_words, q);
	if (!rc)
		rc = btbbx_scan_first_device(d_words, n_words, search_length, lap, max_ac_errors, d_first, q);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(stage + sym_bytes + 8, d_first, 8, hipMemcpyDeviceToHost, q));
	HIP_TRY(hipStreamSynchronize(q));
	memcpy(&first, stage + sym_bytes + 8, 8);
	if (first == ~0ULL)
		return 0;
	memset(first_hit, 0, sizeof(*first_hit));
	first_hit->offset = first >> 32;
	first_hit->lap = lap == BTBBX_LAP_ANY ? (uint32_t)(first >> 8) & 0xffffff : lap;
	first_hit->ac_errors = (uint8_t)(first & 0xff);
	return 1;
}

// ---- time sharding over the GPUs of one node -------------------------------------
//
// The path shards with no exchange step: shard k owns a contiguous, word-aligned range of offsets and
// reads 63 symbols past its end (an access code that starts at the last owned offset ends there).
// The same plan serves one-process-per-GPU callers (bench.py, torch.distributed ranks: each rank asks
// for its own shard) and btbbx_scan_host_multi below (one host thread per listed device).

extern "C" int btbbx_shard_plan(uint64_t search_bits, uint32_t n_shards, uint32_t shard, btbbx_shard *out)
{
	if (!out || n_shards == 0 || shard >= n_shards) {
		set_error("btbbx_shard_plan: shard %u of %u", shard, n_shards);
		return BTBBX_E_ARG;
	}
	const uint64_t words_total = (search_bits + 63) / 64;
	const uint64_t per = (words_total + n_shards - 1) / n_shards;
	uint64_t w0 = (uint64_t)shard * per;
	if (w0 > words_total)
		w0 = words_total;
	uint64_t w1 = w0 + per;
	if (w1 > words_total)
		w1 = words_total;
	uint64_t end = w1 * 64;
	if (end > search_bits)
		end = search_bits;
	out->first_word = w0;
	out->first_offset = w0 * 64;
	out->search_bits = end > w0 * 64 ? end - w0 * 64 : 0;
	out->n_words = out->search_bits ? (out->search_bits + 63 + 63) / 64 : 0;
	return BTBBX_OK;
}

#include <thread>

extern "C" int64_t btbbx_scan_host_multi(const uint64_t *words, uint64_t n_words, uint64_t search_bits, uint32_t lap,
					 int max_ac_errors, btbbx_hit *hits, uint64_t cap, const int *devices,
					 int n_devices)
{
	if (!words || n_devices <= 0 || !devices || (!hits && cap)) {
		set_error("btbbx_scan_host_multi: bad argument");
		return BTBBX_E_ARG;
	}
	int rc = check_scan_args(n_words, n_words, 1, search_bits);
	if (rc)
		return rc;
	struct Part {
		btbbx_shard plan;
		std::vector<btbbx_hit> hits;
		int64_t found = 0;
		char err[256] = "";
	};
	std::vector<Part> parts((size_t)n_devices);
	std::vector<std::thread> workers;
	int home = 0;
	(void)hipGetDevice(&home);
	for (int k = 0; k < n_devices; k++) {
		Part &p = parts[(size_t)k];
		btbbx_shard_plan(search_bits, (uint32_t)n_devices, (uint32_t)k, &p.plan);
		if (!p.plan.search_bits)
			continue;
		const int dev = devices[k];
		workers.emplace_back([&p, dev, words, lap, max_ac_errors, cap]() {
			auto fail = [&p](int64_t code) {
				p.found = code;
				snprintf(p.err, sizeof(p.err), "%s", btbbx_last_error());
			};
			if (hipSetDevice(dev) != hipSuccess)
				return fail(hip_fail(hipGetLastError(), "hipSetDevice"));
			int rc = ctx_require();
			if (rc)
				return fail(rc);
			CallScope scope;
			hipStream_t q = scope_stream();
			uint64_t *d_words = (uint64_t *)scope_device((p.plan.n_words + 2) * 8);
			if (!d_words)
				return fail(BTBBX_E_NOMEM);
			if (hipMemcpyAsync(d_words, words + p.plan.first_word, p.plan.n_words * 8, hipMemcpyHostToDevice, q) !=
			    hipSuccess)
				return fail(hip_fail(hipGetLastError(), "shard upload"));
			// a shard can hold at most what the caller takes in total
			uint64_t want = p.plan.search_bits / 256 + 4096;
			if (want > cap)
				want = cap;
			for (;;) {
				p.hits.resize((size_t)want);
				const int64_t n = scan_resident(d_words, p.plan.n_words, p.plan.search_bits, lap, max_ac_errors,
								p.hits.data(), want, p.plan.first_offset, q);
				if (n < 0)
					return fail(n);
				p.found = n;
				if ((uint64_t)n <= want || want >= cap)
					break;
				want = (uint64_t)n < cap ? (uint64_t)n : cap;      // dense stream: once more with room
			}
			p.hits.resize((size_t)((uint64_t)p.found < want ? (uint64_t)p.found : want));
		});
	}
	for (std::thread &t : workers)
		t.join();
	(void)hipSetDevice(home);
	int64_t total = 0;
	uint64_t written = 0;
	for (const Part &p : parts) {
		if (p.found < 0) {
			set_error("btbbx_scan_host_multi: %s", p.err);
			return p.found;
		}
		total += p.found;
		// shards are disjoint and ascending: concatenation is the (stream, offset) order
		for (size_t i = 0; i < p.hits.size() && written < cap; i++)
			hits[written++] = p.hits[i];
	}
	return total;
}
