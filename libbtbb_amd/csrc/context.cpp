// context.cpp -- device context: table upload, scratch memory, error reporting.
//
// btbbx_init() plays the role of btbb_init() (lib/src/bluetooth_packet.c:279-292):
// it builds the syndrome -> error-pattern map (there a uthash map filled by
// gen_syndrome_map/cycle, :161-185; here an open-addressing table in HBM plus an
// LDS-resident candidate bitmap) for all patterns of weight <= max_ac_errors over
// sync-word bits 0..57.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <shared_mutex>
#include <vector>
#include "common.h"

// One context per HIP ordinal 0 .. BTBBX_MAX_DEVICES-1, plus one slot that is NEVER ready: an ordinal outside the
// range (or a failing hipGetDevice) maps there, so every entry point fails with BTBBX_E_NOTINIT / BTBBX_E_ARG and a
// message instead of silently running with device 0's tables and scratch.
#define NO_DEVICE_SLOT BTBBX_MAX_DEVICES
static Ctx g_ctx[BTBBX_MAX_DEVICES + 1];
static std::mutex g_init_lock;            // table builds / re-builds and shutdown
static std::shared_mutex g_tables_lock;   // the table description of a context while it is swapped (upload_tables) / copied by a launcher

// what a scan launch needs of the current device's tables, as ONE consistent set
void ctx_scan_snapshot(ScanTables *tables, int *table_errors)
{
	std::shared_lock<std::shared_mutex> g(g_tables_lock);
	const Ctx &c = ctx();
	*tables = c.scan;
	*table_errors = c.table_errors;
}
static int g_table_errors = 0;            // process-wide: the first non-zero btbb_init value
static thread_local char g_err[512] = "";

static int current_device()
{
	int d = 0;
	if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= BTBBX_MAX_DEVICES)
		return NO_DEVICE_SLOT;
	return d;
}

Ctx &ctx() { return g_ctx[current_device()]; }

void set_error(const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
	set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
	return BTBBX_E_NODEVICE;
}

extern "C" const char *btbbx_last_error(void) { return g_err; }

extern "C" int btbbx_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

extern "C" int btbbx_table_errors(void)
{
	const Ctx &c = ctx();
	return c.ready ? c.table_errors : -1;
}

int ctx_require()
{
	if (current_device() == NO_DEVICE_SLOT) {
		int d = -1;
		(void)hipGetDevice(&d);
		set_error("btbbx: the current HIP device (ordinal %d) is outside the %d contexts this library keeps "
			  "(BTBBX_MAX_DEVICES) or could not be queried", d, BTBBX_MAX_DEVICES);
		return BTBBX_E_ARG;
	}
	if (!ctx().ready) {
		set_error("btbbx: not initialised on this device (call btbb_init / btbbx_init / btbbx_init_devices first)");
		return BTBBX_E_NOTINIT;
	}
	return BTBBX_OK;
}

// ---- call scopes: leased scratch / staging / stream -----------------------------------------

struct CallBufs {
	int device = 0;
	void *d_scratch = nullptr;
	size_t scratch_bytes = 0;
	void *h_pinned = nullptr;
	size_t pinned_bytes = 0;
	void *pkt_dev = nullptr, *pkt_host = nullptr;
	size_t pkt_bytes = 0;
	void *d_hits = nullptr;                  // hit records + counter of the host-level scans (scope_hits)
	size_t hits_bytes = 0;
	hipStream_t stream = nullptr;
};

static std::mutex g_pool_lock;
static std::vector<CallBufs *> g_pool[BTBBX_MAX_DEVICES + 1];
static thread_local CallBufs *tl_bufs = nullptr;
static thread_local int tl_depth = 0;

static void bufs_free(CallBufs *b)
{
	if (b->d_scratch) (void)hipFree(b->d_scratch);
	if (b->h_pinned) (void)hipHostFree(b->h_pinned);
	if (b->pkt_dev) (void)hipFree(b->pkt_dev);
	if (b->pkt_host) (void)hipHostFree(b->pkt_host);
	if (b->d_hits) (void)hipFree(b->d_hits);
	if (b->stream) (void)hipStreamDestroy(b->stream);
	delete b;
}

CallScope::CallScope()
{
	if (tl_depth++ > 0)
		return;
	const int dev = current_device();
	{
		std::lock_guard<std::mutex> g(g_pool_lock);
		if (!g_pool[dev].empty()) {
			tl_bufs = g_pool[dev].back();
			g_pool[dev].pop_back();
		}
	}
	if (!tl_bufs) {
		tl_bufs = new CallBufs();
		tl_bufs->device = dev;
	}
}

CallScope::~CallScope()
{
	if (--tl_depth > 0)
		return;
	CallBufs *b = tl_bufs;
	tl_bufs = nullptr;
	if (!b)
		return;
	std::lock_guard<std::mutex> g(g_pool_lock);
	g_pool[b->device].push_back(b);
}

hipStream_t scope_stream()
{
	CallBufs *b = tl_bufs;
	if (!b)
		return nullptr;
	if (!b->stream && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess)
		b->stream = nullptr;                       // falls back to the NULL stream
	return b->stream;
}

void *scope_device(size_t bytes)
{
	CallBufs *b = tl_bufs;
	if (!b) {
		set_error("internal: scope_device outside a CallScope");
		return nullptr;
	}
	if (bytes <= b->scratch_bytes)
		return b->d_scratch;
	if (b->d_scratch) {
		if (b->stream) (void)hipStreamSynchronize(b->stream);
		(void)hipFree(b->d_scratch);
	}
	b->d_scratch = nullptr;
	b->scratch_bytes = 0;
	const size_t want = bytes + bytes / 4 + 4096;
	if (hipMalloc(&b->d_scratch, want) != hipSuccess) {
		set_error("btbbx: device scratch allocation of %zu bytes failed", want);
		return nullptr;
	}
	b->scratch_bytes = want;
	return b->d_scratch;
}

// Grow-only device block for the hit list (and its counter) of a host-level scan: in steady state a drop-in
// caller allocates nothing (hipFree synchronises the whole device, which would serialise concurrent callers).
void *scope_hits(size_t bytes)
{
	CallBufs *b = tl_bufs;
	if (!b) {
		set_error("internal: scope_hits outside a CallScope");
		return nullptr;
	}
	if (bytes <= b->hits_bytes)
		return b->d_hits;
	if (b->d_hits) {
		if (b->stream) (void)hipStreamSynchronize(b->stream);
		(void)hipFree(b->d_hits);
	}
	b->d_hits = nullptr;
	b->hits_bytes = 0;
	const size_t want = bytes + bytes / 4 + 4096;
	if (hipMalloc(&b->d_hits, want) != hipSuccess) {
		set_error("btbbx: hit buffer allocation of %zu bytes failed", want);
		return nullptr;
	}
	b->hits_bytes = want;
	return b->d_hits;
}

void *scope_pinned(size_t bytes)
{
	CallBufs *b = tl_bufs;
	if (!b) {
		set_error("internal: scope_pinned outside a CallScope");
		return nullptr;
	}
	if (bytes <= b->pinned_bytes)
		return b->h_pinned;
	if (b->h_pinned) {
		if (b->stream) (void)hipStreamSynchronize(b->stream);
		(void)hipHostFree(b->h_pinned);
	}
	b->h_pinned = nullptr;
	b->pinned_bytes = 0;
	const size_t want = bytes + bytes / 4 + 4096;
	if (hipHostMalloc(&b->h_pinned, want, hipHostMallocDefault) != hipSuccess) {
		set_error("btbbx: pinned host allocation of %zu bytes failed", want);
		return nullptr;
	}
	b->pinned_bytes = want;
	return b->h_pinned;
}

void *scope_packet_block(void **pinned_mirror, size_t bytes)
{
	CallBufs *b = tl_bufs;
	if (!b) {
		set_error("internal: scope_packet_block outside a CallScope");
		return nullptr;
	}
	if (b->pkt_bytes < bytes) {
		if (b->pkt_dev) (void)hipFree(b->pkt_dev);
		if (b->pkt_host) (void)hipHostFree(b->pkt_host);
		b->pkt_dev = b->pkt_host = nullptr;
		b->pkt_bytes = 0;
		hipError_t e = hipMalloc(&b->pkt_dev, bytes);
		if (e == hipSuccess)
			e = hipHostMalloc(&b->pkt_host, bytes, hipHostMallocDefault);
		if (e != hipSuccess) {
			hip_fail(e, "packet buffers");
			return nullptr;
		}
		b->pkt_bytes = bytes;
	}
	*pinned_mirror = b->pkt_host;
	return b->pkt_dev;
}

// ---- syndrome map ---------------------------------------------------------------------

static inline uint32_t hslot_hash(uint64_t key)
{
	return ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u;
}

struct MapBuilder {
	std::vector<uint64_t> slots;
	std::vector<uint32_t> bitmap2;   // empty = not used
	int shift2 = 0;
	uint64_t mask;
	int shift;
	uint64_t count = 0;

	void put(uint64_t syndrome, const int *pos, int n)
	{
		uint64_t packed = syndrome;
		for (int i = 0; i < 5; i++)
			packed |= (uint64_t)(i < n ? pos[i] : 63) << (34 + 6 * i);
		uint64_t h = hslot_hash(syndrome) >> shift;
		while (slots[h] != HSLOT_EMPTY)
			h = (h + 1) & mask;
		slots[h] = packed;
		if (!bitmap2.empty()) {
			const uint32_t i2 = ((uint32_t)syndrome * 0x9E3779B1u) >> shift2;
			bitmap2[i2 >> 5] |= 1u << (i2 & 31);
		}
		count++;
	}

	void enumerate(const HostTables &t, uint64_t syn, int *pos, int depth, int start, int want)
	{
		for (int i = start; i < 58; i++) {
			pos[depth] = i;
			uint64_t s = syn ^ t.col[i];
			if (depth + 1 == want)
				put(s, pos, want);
			else
				enumerate(t, s, pos, depth + 1, i + 1, want);
		}
	}
};

// The candidate set of the sliding checks (slide.h): index bit b = parity of the window ^ PN under the
// taps shifted to b; a window the reference accepts differs from a codeword in at most max_ac_errors of
// the bits 0..56 (the map's patterns reach bit 57, which the checks do not touch), so its index is K ^ the
// XOR of that many columns.  Host only.  Returns the number of members or a negative error.
#define SLIDE_WORDS (1u << (SLIDE_BITS - 5))
// the set of values `bits` positions of the check `taps` can take (XOR their value over PN) for a window the reference
// accepts with tables for max_ac_errors: sums of at most that many columns
static int build_slide_set(const HostTables &t, int max_ac_errors, std::vector<uint32_t> &slide_bitmap,
			   const int bits = SLIDE_BITS, const uint64_t taps = SLIDE_TAPS)
{
	slide_bitmap.assign(1u << (bits - 5), 0);
	for (int b = 0; b < bits; b++)           // every check must annihilate every codeword
		for (int r = 0; r < 30; r++) {
			const uint64_t row = (1ULL << (34 + r)) | t.col[34 + r];
			if (((taps << b) >> 57) || (__builtin_popcountll(row & (taps << b)) & 1)) {
				set_error("btbb_init: the sliding parity check does not hold for this generator");
				return BTBBX_E_ARG;
			}
		}
	uint32_t colv[57], k_pn = 0;
	for (int b = 0; b < bits; b++)
		k_pn |= (uint32_t)(__builtin_popcountll(SW_PN & (taps << b)) & 1) << b;
	for (int i = 0; i < 57; i++) {
		colv[i] = 0;
		for (int b = 0; b < bits; b++)
			if (i >= b && ((taps >> (i - b)) & 1))
				colv[i] |= 1u << b;
	}
	slide_bitmap[k_pn >> 5] |= 1u << (k_pn & 31);
	int idx[5];
	for (int k = 1; k <= max_ac_errors && k <= 5; k++) {
		for (int i = 0; i < k; i++)
			idx[i] = i;
		for (;;) {
			uint32_t v = k_pn;
			for (int i = 0; i < k; i++)
				v ^= colv[idx[i]];
			slide_bitmap[v >> 5] |= 1u << (v & 31);
			int i = k - 1;
			while (i >= 0 && idx[i] == 57 - k + i)
				i--;
			if (i < 0)
				break;
			idx[i]++;
			for (int j = i + 1; j < k; j++)
				idx[j] = idx[j - 1] + 1;
		}
	}
	int members = 0;
	for (uint32_t w : slide_bitmap)
		members += __builtin_popcount(w);
	return members;
}

// The two sets of scan_slide_kernel's two-level form (tables for three and four errors; slide.h), laid out as the kernel reads
// them.  first: 2^SLIDE4_BITS bits over SLIDE4_TAPS for the LDS.  An idle chain of the kernel indexes 0 or 1, and with four
// errors the all-zero value of these twenty checks IS a sum of four columns and PN's -- its complement is not: the kernel runs
// on the COMPLEMENTED check stream (Slide4::INVERT in scan.hip), so member i stands at index ~i here.  second: 2^SLIDE4B_BITS
// bits over SLIDE4B_TAPS, read from L2 one word per look-up: member bit of index i at bit 31 - (i & 31) of word i >> 5 (a left
// shift by i brings it to the sign).
static int build_two_level_sets(const HostTables &t, int max_ac_errors, std::vector<uint32_t> &first, std::vector<uint32_t> &second)
{
	int rc = build_slide_set(t, max_ac_errors, first, SLIDE4_BITS, SLIDE4_TAPS);
	if (rc >= 0)
		rc = build_slide_set(t, max_ac_errors, second, SLIDE4B_BITS, SLIDE4B_TAPS);
	if (rc < 0)
		return rc;
	std::vector<uint32_t> inv(first.size(), 0);
	const uint32_t full = (1u << SLIDE4_BITS) - 1;
	for (uint32_t i = 0; i <= full; i++)
		if ((first[i >> 5] >> (i & 31)) & 1)
			inv[(i ^ full) >> 5] |= 1u << ((i ^ full) & 31);
	first.swap(inv);
	if (first[0] & 3u) {
		set_error("btbbx_init: internal: index 0 / 1 of the sliding checks is a member of the candidate set");
		return BTBBX_E_ARG;
	}
	for (uint32_t &w : second) {
		uint32_t r = 0;
		for (int k = 0; k < 32; k++)
			r |= ((w >> k) & 1u) << (31 - k);
		w = r;
	}
	return BTBBX_OK;
}

// Tables for FIVE errors (round 6): 5.0 M patterns leave no set that fits the LDS and filters, so every barker survivor of
// scan_lap_any_kernel used to probe the 2^26-bit bitmap over the syndrome hash (8 MiB: half of it outside an XCD's L2) -- the
// kernel ran at that table's probe rate.  The 2^SLIDE4B_BITS-bit set over SLIDE4B_TAPS (2 MiB, L2-resident, the same layout as the
// second level of the two-level form) goes in front of it: 22.4 % of the survivors pass it (3 756 016 members), the rest never
// compute a syndrome or leave the L2.
static int build_five_error_front_set(const HostTables &t, std::vector<uint32_t> &second)
{
	const int rc = build_slide_set(t, 5, second, SLIDE4B_BITS, SLIDE4B_TAPS);
	if (rc < 0)
		return rc;
	for (uint32_t &w : second) {
		uint32_t r = 0;
		for (int k = 0; k < 32; k++)
			r |= ((w >> k) & 1u) << (31 - k);
		w = r;
	}
	return rc;
}

extern "C" int btbbx_slide_sets_two_level(int max_ac_errors, uint32_t *first_words, uint32_t *second_words, uint64_t *taps)
{
	if (max_ac_errors == 5 && second_words) {       // (tables for five errors: the front set of scan_lap_any_kernel only; first_words is not written)
		std::vector<uint32_t> second;
		const int rc5 = build_five_error_front_set(host_tables(), second);
		if (rc5 < 0)
			return rc5;
		memcpy(second_words, second.data(), second.size() * sizeof(uint32_t));
		if (taps) {
			taps[0] = 0;
			taps[1] = SLIDE4B_TAPS;
		}
		return BTBBX_OK;
	}
	if ((max_ac_errors != 3 && max_ac_errors != 4) || !first_words || !second_words) {
		set_error("btbbx_slide_sets_two_level: bad argument (tables for three, four or five errors have these sets)");
		return BTBBX_E_ARG;
	}
	std::vector<uint32_t> first, second;
	const int rc = build_two_level_sets(host_tables(), max_ac_errors, first, second);
	if (rc < 0)
		return rc;
	memcpy(first_words, first.data(), first.size() * sizeof(uint32_t));
	memcpy(second_words, second.data(), second.size() * sizeof(uint32_t));
	if (taps) {
		taps[0] = SLIDE4_TAPS;
		taps[1] = SLIDE4B_TAPS;
	}
	return BTBBX_OK;
}

extern "C" int btbbx_slide_set(int max_ac_errors, uint32_t *bitmap_words, uint64_t *taps)
{
	if (max_ac_errors < 0 || !bitmap_words) {
		set_error("btbbx_slide_set: bad argument");
		return BTBBX_E_ARG;
	}
	std::vector<uint32_t> set;
	const int members = build_slide_set(host_tables(), max_ac_errors, set);
	if (members < 0)
		return members;
	memcpy(bitmap_words, set.data(), set.size() * sizeof(uint32_t));
	if (taps)
		*taps = SLIDE_TAPS;
	return members;
}

// size of the second-level bitmap by the error count the tables are built for (2^bits bits; 26 = 8 MiB, rounds 1-3)
#define BITMAP2_BITS_3 26
#define BITMAP2_BITS_4 22          // 512 KiB.  (rounds 3-4, when every survivor probed it: 2^24 = 2 MiB, "2.78 against 3.32 ms per GiB with 8 MiB; 2^22: 3.51".)
                                   // Round 6: only the exact check of the two-level kernel's candidates looks at it now, and 2 MiB of it beside the 2 MiB
                                   // second-level set were the whole L2 of an XCD: 2.07 x the algorithmic bytes per launch -> 1.50 x, same time (profiles/r06_init4)
#define BITMAP2_BITS_5 26
static int upload_tables(int max_ac_errors)
{
	const HostTables &t = host_tables();
	Ctx &c = ctx();

	// entry count = sum_{k<=n} C(58,k)
	uint64_t entries = 0, binom = 1;
	for (int k = 1; k <= max_ac_errors; k++) {
		binom = binom * (uint64_t)(58 - k + 1) / (uint64_t)k;
		entries += binom;
	}
	// Slots: at least twice the patterns (open addressing, linear probing), and sixteen times while that stays within 2^20
	// slots (8 MiB).  The exact check of the LAP_ANY scan looks up sixty candidates at a time, two thirds of them false ones
	// that probe until they meet an empty slot: the batch waits for its LONGEST probe chain, every probe a dependent round trip
	// to the L2 -- at a load factor of 0.42 (two errors, 4096 slots) six to eight of them, at 0.03 one or two.
	int bits = 12;
	while ((1ULL << bits) < 2 * entries + 16)
		bits++;
	while (bits < 20 && (1ULL << bits) < 16 * entries)
		bits++;
	MapBuilder mb;
	mb.slots.assign(1ULL << bits, HSLOT_EMPTY);
	mb.mask = (1ULL << bits) - 1;
	mb.shift = 32 - bits;
	if (max_ac_errors >= 3) {
		// With 32 567 (3 errors) .. 5.0 M (5) patterns a bitmap over a hash of the low 32 syndrome bits (2^24 .. 2^26 bits,
		// L2 / Infinity Cache resident) prunes the candidates before the pattern table is probed (the exact check's first
		// look; for five errors also what every survivor of scan_lap_any_kernel probes).
		const int bits2 = max_ac_errors == 3 ? BITMAP2_BITS_3 : max_ac_errors == 4 ? BITMAP2_BITS_4 : BITMAP2_BITS_5;
		mb.shift2 = 32 - bits2;
		mb.bitmap2.assign(1u << (bits2 - 5), 0);
		mb.bitmap2[0] |= 1u;            // hash of syndrome 0
	}
	int pos[5];
	for (int k = 1; k <= max_ac_errors; k++)
		mb.enumerate(t, 0, pos, 0, 0, k);

	std::vector<uint32_t> tabA(LDS_TABA_WORDS), tabB(LDS_TABB_WORDS);
	for (uint32_t v = 0; v < LDS_TABA_WORDS; v++) {
		uint64_t s = 0;
		for (int i = 0; i < TABA_BITS; i++)
			if ((v >> i) & 1)
				s ^= t.col[TABA_FIRST + i];
		tabA[v] = (uint32_t)s;
	}
	uint64_t kclass[2];
	kclass[0] = host_syndrome(((uint64_t)BARKER0 << 57) ^ SW_PN);
	kclass[1] = host_syndrome(((uint64_t)BARKER1 << 57) ^ SW_PN);
	for (uint32_t v = 0; v < (1u << TABB_BITS); v++) {
		uint64_t s = kclass[0];
		for (int i = 0; i < TABB_BITS; i++)
			if ((v >> i) & 1)
				s ^= t.col[TABA_FIRST + TABA_BITS + i];
		tabB[v] = (uint32_t)s;
	}

	// The 2^SLIDE_BITS-bit set is read by the one-level form of scan_slide_kernel only (tables for <= 2 errors; three and four run the
	// two-level form on their own sets, five the probe kernel): for larger tables it stays empty -- launch_scan refuses the one-level
	// kernel with them -- and its C(57, n) sums are not enumerated for nothing.
	std::vector<uint32_t> slide_bitmap(SLIDE_WORDS, 0u);
	if (max_ac_errors <= 2) {
		const int rc_slide = build_slide_set(t, max_ac_errors, slide_bitmap);
		if (rc_slide < 0)
			return rc_slide;
		// scan_slide_kernel lets a chain that has run out of survivors shift itself out: its index is then 0 or 1.  Those two
		// must not be members, or a lane without a survivor would look like a candidate (true for the table sets of this code).
		if (slide_bitmap[0] & 3u) {
			set_error("btbbx_init: internal: index 0 / 1 of the sliding checks is a member of the candidate set");
			return BTBBX_E_ARG;
		}
	}

	// tables for three and four errors: the two sets of scan_slide_kernel's two-level form (slide.h)
	std::vector<uint32_t> slide4, slide4b;
	if (max_ac_errors == 3 || max_ac_errors == 4) {
		const int rc4 = build_two_level_sets(t, max_ac_errors, slide4, slide4b);
		if (rc4 < 0)
			return rc4;
	} else if (max_ac_errors == 5) {
		const int rc5 = build_five_error_front_set(t, slide4b);      // (slide4 stays empty: scan_lap_any_kernel, not the two-level form)
		if (rc5 < 0)
			return rc5;
	}

	// one block: tabA | tabB | slide set | the two sets of the two-level form
	size_t off_a = 0, off_b = off_a + 4 * LDS_TABA_WORDS, off_m = off_b + 4 * LDS_TABB_WORDS;
	size_t off_s = off_m;
	size_t off_s4 = off_s + 4 * SLIDE_WORDS, off_s4b = off_s4 + 4 * slide4.size();
	size_t total = off_s4b + 4 * slide4b.size();
	// Build the new set beside the old one and swap only when every copy has succeeded: a failure
	// leaves the context as it was; the replaced set outlives the swap by one re-build (see below).
	struct Fresh {
		void *tab = nullptr, *hslots = nullptr, *bitmap2 = nullptr;
		~Fresh() { if (tab) (void)hipFree(tab); if (hslots) (void)hipFree(hslots); if (bitmap2) (void)hipFree(bitmap2); }
	} fresh;
	if (!mb.bitmap2.empty()) {
		HIP_TRY(hipMalloc(&fresh.bitmap2, mb.bitmap2.size() * 4));
		HIP_TRY(hipMemcpy(fresh.bitmap2, mb.bitmap2.data(), mb.bitmap2.size() * 4, hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMalloc(&fresh.tab, total));
	HIP_TRY(hipMalloc(&fresh.hslots, mb.slots.size() * sizeof(uint64_t)));
	char *base = (char *)fresh.tab;
	HIP_TRY(hipMemcpy(base + off_a, tabA.data(), 4 * LDS_TABA_WORDS, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(base + off_b, tabB.data(), 4 * LDS_TABB_WORDS, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(base + off_s, slide_bitmap.data(), 4 * SLIDE_WORDS, hipMemcpyHostToDevice));
	if (!slide4.empty())
		HIP_TRY(hipMemcpy(base + off_s4, slide4.data(), 4 * slide4.size(), hipMemcpyHostToDevice));
	if (!slide4b.empty())
		HIP_TRY(hipMemcpy(base + off_s4b, slide4b.data(), 4 * slide4b.size(), hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(fresh.hslots, mb.slots.data(), mb.slots.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	HIP_TRY(hipDeviceSynchronize());                   // scans queued on any stream still read the old tables
	// The set just replaced is not freed here: a launcher on another thread may have copied `c.scan` (the old
	// pointers) a moment ago and not launched yet.  It is parked in the context and freed by the NEXT re-build,
	// i.e. after one more hipDeviceSynchronize() -- by then every launch that could have seen it has finished.
	// ... and it copies the whole table description under the shared side of g_tables_lock (ctx_scan_snapshot): either the
	// old set or the new one, never a mix of tables of one with the error count of the other.
	std::unique_lock<std::shared_mutex> swap_guard(g_tables_lock);
	if (c.d_retired_tab) (void)hipFree(c.d_retired_tab);
	if (c.d_retired_hslots) (void)hipFree(c.d_retired_hslots);
	if (c.d_retired_bitmap2) (void)hipFree(c.d_retired_bitmap2);
	c.d_retired_tab = c.d_tab_block;
	c.d_retired_hslots = c.d_hslots;
	c.d_retired_bitmap2 = c.d_bitmap2;
	c.d_tab_block = fresh.tab;
	c.d_hslots = fresh.hslots;
	c.d_bitmap2 = fresh.bitmap2;
	fresh.tab = fresh.hslots = fresh.bitmap2 = nullptr;
	base = (char *)c.d_tab_block;
	c.scan.tabA = (const uint32_t *)(base + off_a);
	c.scan.tabB = (const uint32_t *)(base + off_b);
	c.scan.slide_bitmap = (const uint32_t *)(base + off_s);
	c.scan.slide4_bitmap = slide4.empty() ? nullptr : (const uint32_t *)(base + off_s4);
	c.scan.slide4b_bitmap = slide4b.empty() ? nullptr : (const uint32_t *)(base + off_s4b);
	c.scan.hslots = (const uint64_t *)c.d_hslots;
	c.scan.hmask = mb.mask;
	c.scan.kclass[0] = kclass[0];
	c.scan.kclass[1] = kclass[1];
	c.scan.kdiff = (uint32_t)(kclass[0] ^ kclass[1]);
	c.scan.hi_mask[0] = c.scan.hi_mask[1] = 0;
	for (int j = 0; j < 57; j++) {
		if ((t.col[j] >> 32) & 1) c.scan.hi_mask[0] |= 1ULL << j;
		if ((t.col[j] >> 33) & 1) c.scan.hi_mask[1] |= 1ULL << j;
	}
	c.scan.bitmap2 = (const uint32_t *)c.d_bitmap2;
	c.scan.bitmap2_shift = (uint32_t)mb.shift2;
	c.table_errors = max_ac_errors;
	return BTBBX_OK;
}

// tables for the calling thread's current device; g_init_lock held
static int init_current_device(int max_ac_errors)
{
	const int dev = current_device();
	if (dev == NO_DEVICE_SLOT) {
		int d = -1;
		(void)hipGetDevice(&d);
		set_error("btbbx_init: HIP device ordinal %d is not below BTBBX_MAX_DEVICES = %d (or hipGetDevice failed)", d,
			  BTBBX_MAX_DEVICES);
		return BTBBX_E_ARG;
	}
	Ctx &c = g_ctx[dev];
	// first non-zero max_ac_errors builds the map, later calls keep it -- for the whole process, as in
	// the reference (bluetooth_packet.c:288-289: `if ((syndrome_map == NULL) && (max_ac_errors))`)
	if (g_table_errors == 0 && max_ac_errors > 0)
		g_table_errors = max_ac_errors;
	if (!c.ready) {
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) {
			set_error("btbbx_init: no HIP device available (%s) -- this library has no CPU path",
				  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
			return BTBBX_E_NODEVICE;
		}
		c.device = dev;
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, dev));
		c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		int rc = chain_upload(host_tables());
		if (rc)
			return rc;
		rc = upload_tables(g_table_errors);
		if (rc)
			return rc;
		c.ready = true;
		return BTBBX_OK;
	}
	if (c.table_errors != g_table_errors)
		return upload_tables(g_table_errors);
	return BTBBX_OK;
}

extern "C" int btbbx_init(int max_ac_errors)
{
	if (max_ac_errors < 0 || max_ac_errors > 5) {
		set_error("btbbx_init: max_ac_errors out of range");
		return BTBBX_E_ARG;
	}
	std::lock_guard<std::mutex> g(g_init_lock);
	return init_current_device(max_ac_errors);
}

extern "C" int btbbx_init_devices(const int *devices, int n_devices, int max_ac_errors)
{
	if (max_ac_errors < 0 || max_ac_errors > 5 || n_devices < 0 || (n_devices && !devices)) {
		set_error("btbbx_init_devices: bad argument");
		return BTBBX_E_ARG;
	}
	const int have = btbbx_device_count();
	int home = 0;
	if (have <= 0 || hipGetDevice(&home) != hipSuccess) {
		set_error("btbbx_init_devices: no HIP device available -- this library has no CPU path");
		return BTBBX_E_NODEVICE;
	}
	std::lock_guard<std::mutex> g(g_init_lock);
	int rc = BTBBX_OK;
	for (int i = 0; i < n_devices && !rc; i++) {
		if (devices[i] < 0 || devices[i] >= have || devices[i] >= BTBBX_MAX_DEVICES) {
			set_error("btbbx_init_devices: device %d does not exist (%d visible)", devices[i], have);
			rc = BTBBX_E_ARG;
			break;
		}
		if (hipSetDevice(devices[i]) != hipSuccess)
			rc = hip_fail(hipGetLastError(), "hipSetDevice");
		else
			rc = init_current_device(max_ac_errors);
	}
	(void)hipSetDevice(home);
	return rc;
}

void hop_pool_release();     // hop.hip
void sort_scratch_release(); // sort.hip

extern "C" void btbbx_shutdown(void)
{
	std::lock_guard<std::mutex> g(g_init_lock);
	int home = 0;
	(void)hipGetDevice(&home);
	hop_pool_release();
	sort_scratch_release();
	for (int d = 0; d < BTBBX_MAX_DEVICES; d++) {
		Ctx &c = g_ctx[d];
		std::vector<CallBufs *> mine;
		{
			std::lock_guard<std::mutex> p(g_pool_lock);
			mine.swap(g_pool[d]);
		}
		if (!c.ready && mine.empty())
			continue;
		(void)hipSetDevice(d);
		for (CallBufs *b : mine)
			bufs_free(b);
		if (c.d_tab_block) (void)hipFree(c.d_tab_block);
		if (c.d_hslots) (void)hipFree(c.d_hslots);
		if (c.d_bitmap2) (void)hipFree(c.d_bitmap2);
		if (c.d_retired_tab) (void)hipFree(c.d_retired_tab);
		if (c.d_retired_hslots) (void)hipFree(c.d_retired_hslots);
		if (c.d_retired_bitmap2) (void)hipFree(c.d_retired_bitmap2);
		c = Ctx();
	}
	g_table_errors = 0;
	(void)hipSetDevice(home);
}

// ---- memory helpers -----------------------------------------------------------------------

extern "C" void *btbbx_malloc(size_t bytes)
{
	void *p = nullptr;
	hipError_t e = hipMalloc(&p, bytes);
	if (e != hipSuccess) {
		hip_fail(e, "hipMalloc");
		return nullptr;
	}
	return p;
}

extern "C" void btbbx_free(void *p)
{
	if (p)
		(void)hipFree(p);
}

extern "C" int btbbx_memcpy_h2d(void *dst, const void *src, size_t bytes)
{
	HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
	return BTBBX_OK;
}

extern "C" int btbbx_memcpy_d2h(void *dst, const void *src, size_t bytes)
{
	HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
	return BTBBX_OK;
}

extern "C" int btbbx_memset(void *p, int value, size_t bytes)
{
	HIP_TRY(hipMemset(p, value, bytes));
	return BTBBX_OK;
}

extern "C" int btbbx_sync(void *hip_stream)
{
	HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
	return BTBBX_OK;
}
