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
#include <vector>
#include "common.h"

static Ctx g_ctx;
static thread_local char g_err[512] = "";

Ctx &ctx() { return g_ctx; }

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

extern "C" int btbbx_table_errors(void) { return g_ctx.ready ? g_ctx.table_errors : -1; }

int ctx_require()
{
	if (!g_ctx.ready) {
		set_error("btbbx: not initialised (call btbb_init/btbbx_init first)");
		return BTBBX_E_NOTINIT;
	}
	return BTBBX_OK;
}

void *ctx_scratch(size_t bytes)
{
	if (bytes <= g_ctx.scratch_bytes)
		return g_ctx.d_scratch;
	if (g_ctx.d_scratch)
		(void)hipFree(g_ctx.d_scratch);
	g_ctx.d_scratch = nullptr;
	g_ctx.scratch_bytes = 0;
	size_t want = bytes + bytes / 4 + 4096;
	if (hipMalloc(&g_ctx.d_scratch, want) != hipSuccess) {
		set_error("btbbx: device scratch allocation of %zu bytes failed", want);
		return nullptr;
	}
	g_ctx.scratch_bytes = want;
	return g_ctx.d_scratch;
}

void *ctx_pinned(size_t bytes)
{
	if (bytes <= g_ctx.pinned_bytes)
		return g_ctx.h_pinned;
	if (g_ctx.h_pinned)
		(void)hipHostFree(g_ctx.h_pinned);
	g_ctx.h_pinned = nullptr;
	g_ctx.pinned_bytes = 0;
	size_t want = bytes + bytes / 4 + 4096;
	if (hipHostMalloc(&g_ctx.h_pinned, want, hipHostMallocDefault) != hipSuccess) {
		set_error("btbbx: pinned host allocation of %zu bytes failed", want);
		return nullptr;
	}
	g_ctx.pinned_bytes = want;
	return g_ctx.h_pinned;
}

// ---- syndrome map ---------------------------------------------------------------------

static inline uint32_t hslot_hash(uint64_t key)
{
	return ((uint32_t)key ^ (uint32_t)(key >> 32)) * 0x9E3779B1u;
}

struct MapBuilder {
	std::vector<uint64_t> slots;
	std::vector<uint32_t> bitmap;
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
		uint32_t proj = (uint32_t)syndrome & ((1u << BITMAP_BITS) - 1);
		bitmap[proj >> 5] |= 1u << (proj & 31);
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

static int upload_tables(int max_ac_errors)
{
	const HostTables &t = host_tables();
	Ctx &c = g_ctx;

	// entry count = sum_{k<=n} C(58,k)
	uint64_t entries = 0, binom = 1;
	for (int k = 1; k <= max_ac_errors; k++) {
		binom = binom * (uint64_t)(58 - k + 1) / (uint64_t)k;
		entries += binom;
	}
	int bits = 12;
	while ((1ULL << bits) < 2 * entries + 16)
		bits++;
	MapBuilder mb;
	mb.slots.assign(1ULL << bits, HSLOT_EMPTY);
	mb.bitmap.assign(LDS_BITMAP_WORDS, 0);
	mb.mask = (1ULL << bits) - 1;
	mb.shift = 32 - bits;
	mb.bitmap[0] |= 1u;                     // the zero syndrome (error-free codeword)
	if (max_ac_errors >= 3) {
		// With 32 567 (3 errors) .. 5.0 M (5) patterns the 2^19-bit LDS bitmap passes 6 % .. 100 %
		// of the survivors; a 2^26-bit bitmap over a hash of the low 32 syndrome bits (8 MiB, L2 /
		// Infinity Cache resident) prunes them before the pattern table is probed.
		mb.shift2 = 32 - 26;
		mb.bitmap2.assign(1u << (26 - 5), 0);
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
				s ^= t.col[32 + i];
		tabA[v] = (uint32_t)s;
	}
	uint64_t kclass[2];
	kclass[0] = host_syndrome(((uint64_t)BARKER0 << 57) ^ SW_PN);
	kclass[1] = host_syndrome(((uint64_t)BARKER1 << 57) ^ SW_PN);
	for (uint32_t v = 0; v < (1u << TABB_BITS); v++) {
		uint64_t s = kclass[0];
		for (int i = 0; i < TABB_BITS; i++)
			if ((v >> i) & 1)
				s ^= t.col[32 + TABA_BITS + i];
		tabB[v] = (uint32_t)s;
	}

	// one block: tabA | tabB | bitmap
	size_t off_a = 0, off_b = off_a + 4 * LDS_TABA_WORDS, off_m = off_b + 4 * LDS_TABB_WORDS;
	size_t total = off_m + 4 * LDS_BITMAP_WORDS;
	if (c.d_tab_block) { (void)hipFree(c.d_tab_block); c.d_tab_block = nullptr; }
	if (c.d_hslots) { (void)hipFree(c.d_hslots); c.d_hslots = nullptr; }
	if (c.d_bitmap2) { (void)hipFree(c.d_bitmap2); c.d_bitmap2 = nullptr; }
	if (!mb.bitmap2.empty()) {
		HIP_TRY(hipMalloc(&c.d_bitmap2, mb.bitmap2.size() * 4));
		HIP_TRY(hipMemcpy(c.d_bitmap2, mb.bitmap2.data(), mb.bitmap2.size() * 4, hipMemcpyHostToDevice));
	}
	HIP_TRY(hipMalloc(&c.d_tab_block, total));
	HIP_TRY(hipMalloc(&c.d_hslots, mb.slots.size() * sizeof(uint64_t)));
	char *base = (char *)c.d_tab_block;
	HIP_TRY(hipMemcpy(base + off_a, tabA.data(), 4 * LDS_TABA_WORDS, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(base + off_b, tabB.data(), 4 * LDS_TABB_WORDS, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(base + off_m, mb.bitmap.data(), 4 * LDS_BITMAP_WORDS, hipMemcpyHostToDevice));
	HIP_TRY(hipMemcpy(c.d_hslots, mb.slots.data(), mb.slots.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
	c.scan.tabA = (const uint32_t *)(base + off_a);
	c.scan.tabB = (const uint32_t *)(base + off_b);
	c.scan.bitmap = (const uint32_t *)(base + off_m);
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

extern "C" int btbbx_init(int max_ac_errors)
{
	if (max_ac_errors < 0 || max_ac_errors > 5) {
		set_error("btbbx_init: max_ac_errors out of range");
		return BTBBX_E_ARG;
	}
	Ctx &c = g_ctx;
	if (!c.ready) {
		int n = 0;
		hipError_t e = hipGetDeviceCount(&n);
		if (e != hipSuccess || n <= 0) {
			set_error("btbbx_init: no HIP device available (%s) -- this library has no CPU path",
				  e == hipSuccess ? "device count 0" : hipGetErrorString(e));
			return BTBBX_E_NODEVICE;
		}
		HIP_TRY(hipGetDevice(&c.device));
		hipDeviceProp_t prop;
		HIP_TRY(hipGetDeviceProperties(&prop, c.device));
		c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
		int rc = chain_upload(host_tables());
		if (rc)
			return rc;
		rc = upload_tables(max_ac_errors);
		if (rc)
			return rc;
		c.ready = true;
		return BTBBX_OK;
	}
	// first non-zero max_ac_errors builds the map, later calls keep it
	// (bluetooth_packet.c:288-289: `if ((syndrome_map == NULL) && (max_ac_errors))`)
	if (c.table_errors == 0 && max_ac_errors > 0)
		return upload_tables(max_ac_errors);
	return BTBBX_OK;
}

void hop_pool_release();     // hop.hip
void sort_scratch_release(); // sort.hip

extern "C" void btbbx_shutdown(void)
{
	Ctx &c = g_ctx;
	hop_pool_release();
	sort_scratch_release();
	if (c.d_tab_block) (void)hipFree(c.d_tab_block);
	if (c.d_hslots) (void)hipFree(c.d_hslots);
	if (c.d_bitmap2) (void)hipFree(c.d_bitmap2);
	if (c.d_scratch) (void)hipFree(c.d_scratch);
	if (c.h_pinned) (void)hipHostFree(c.h_pinned);
	c = Ctx();
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
