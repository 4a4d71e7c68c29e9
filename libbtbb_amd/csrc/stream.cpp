// stream.cpp -- chunked ingest for live captures: what sits between a capture loop and the
// scan kernel.  The reference's callers hand btbb_find_ac a window of
// one-symbol-per-byte data and slide it themselves (lib/src/btbb.h:82-94); here the library
// owns the sliding: two pinned staging buffers, two HIP streams, a one-word carry between
// chunks.  While chunk k is scanned, chunk k+1 is copied and packed.
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "common.h"

extern "C" int64_t btbbx_stream_submit(btbbx_stream *s, uint64_t n_symbols, btbbx_hit *hits, uint64_t cap);

struct Slot {
	hipStream_t stream = nullptr;
	void *h_in = nullptr;          // pinned staging (symbols or words)
	uint8_t *d_sym = nullptr;      // device symbols (FMT_SYMBOLS)
	uint64_t *d_words = nullptr;   // [carry][chunk words][pad]
	btbbx_hit *d_hits = nullptr;
	uint32_t *d_count = nullptr;
	btbbx_hit *h_hits = nullptr;   // pinned
	uint32_t *h_count = nullptr;   // pinned
	uint64_t base = 0;             // global offset of buffer bit 0 (may be "negative": first chunk has no carry)
	bool first = false;
	bool busy = false;
	uint64_t n_words = 0;
};

struct btbbx_stream {
	uint32_t lap;
	int max_err;
	int format;
	uint64_t max_chunk;
	uint32_t hit_cap;
	Slot slot[2];
	int cur = 0;
	uint64_t fed = 0;              // symbols fed so far
	bool started = false;
	bool tail_fed = false;         // a chunk that was not a multiple of 64 ended the stream
};

static void free_slot(Slot &s)
{
	if (s.stream) (void)hipStreamDestroy(s.stream);
	if (s.h_in) (void)hipHostFree(s.h_in);
	if (s.d_sym) (void)hipFree(s.d_sym);
	if (s.d_words) (void)hipFree(s.d_words);
	if (s.d_hits) (void)hipFree(s.d_hits);
	if (s.d_count) (void)hipFree(s.d_count);
	if (s.h_hits) (void)hipHostFree(s.h_hits);
	if (s.h_count) (void)hipHostFree(s.h_count);
	s = Slot();
}

extern "C" void btbbx_stream_close(btbbx_stream *s)
{
	if (!s)
		return;
	for (auto &sl : s->slot) {
		if (sl.stream) (void)hipStreamSynchronize(sl.stream);
		free_slot(sl);
	}
	delete s;
}

extern "C" btbbx_stream *btbbx_stream_open(uint32_t lap, int max_ac_errors, uint64_t max_chunk_symbols, int format)
{
	if (ctx_require())
		return nullptr;
	if (max_chunk_symbols < 64 || format < BTBBX_FMT_PACKED || format > BTBBX_FMT_PACKED_MSB) {
		set_error("btbbx_stream_open: bad argument");
		return nullptr;
	}
	btbbx_stream *s = new btbbx_stream();
	s->lap = lap;
	s->max_err = max_ac_errors;
	s->format = format;
	s->max_chunk = (max_chunk_symbols + 63) & ~63ULL;
	// hits are sparse; size for one access code per 128 symbols (a packet is >= 68) + slack
	uint64_t cap = s->max_chunk / 128 + 4096;
	s->hit_cap = cap > 0x7fffffffULL ? 0x7fffffffu : (uint32_t)cap;
	const uint64_t words = s->max_chunk / 64 + 2;
	bool ok = true;
	for (auto &sl : s->slot) {
		ok = ok && hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking) == hipSuccess;
		size_t in_bytes = format == BTBBX_FMT_SYMBOLS ? s->max_chunk : s->max_chunk / 8;
		ok = ok && hipHostMalloc(&sl.h_in, in_bytes + 64, hipHostMallocDefault) == hipSuccess;
		if (format == BTBBX_FMT_SYMBOLS)
			ok = ok && hipMalloc(&sl.d_sym, s->max_chunk + 64) == hipSuccess;
		ok = ok && hipMalloc(&sl.d_words, words * 8) == hipSuccess;
		ok = ok && hipMalloc(&sl.d_hits, (size_t)s->hit_cap * sizeof(btbbx_hit)) == hipSuccess;
		ok = ok && hipMalloc(&sl.d_count, 4) == hipSuccess;
		ok = ok && hipHostMalloc(&sl.h_hits, (size_t)s->hit_cap * sizeof(btbbx_hit), hipHostMallocDefault) == hipSuccess;
		ok = ok && hipHostMalloc(&sl.h_count, 4, hipHostMallocDefault) == hipSuccess;
	}
	if (!ok) {
		set_error("btbbx_stream_open: allocation failed");
		btbbx_stream_close(s);
		return nullptr;
	}
	return s;
}

static void staging_copy(void *dst, const void *src, size_t bytes)
{
	const size_t piece = 4u << 20;
	unsigned hw = std::thread::hardware_concurrency();
	size_t parts = bytes / piece;
	if (parts > 8) parts = 8;
	if (hw && parts > hw) parts = hw;
	if (parts < 2) {
		memcpy(dst, src, bytes);
		return;
	}
	std::vector<std::thread> pool;
	const size_t each = ((bytes / parts) + 63) & ~(size_t)63;
	for (size_t k = 1; k < parts; k++) {
		const size_t lo = k * each, hi = k + 1 == parts ? bytes : (k + 1) * each;
		pool.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
	}
	memcpy(dst, src, each);
	for (auto &t : pool)
		t.join();
}

// wait for a slot's launch and hand its hits out with global offsets
static int64_t collect(btbbx_stream *s, Slot &sl, btbbx_hit *hits, uint64_t cap)
{
	if (!sl.busy)
		return 0;
	if (hipStreamSynchronize(sl.stream) != hipSuccess)
		return hip_fail(hipGetLastError(), "stream sync");
	sl.busy = false;
	uint32_t n = *sl.h_count;
	if (n > s->hit_cap) {
		set_error("btbbx_stream: more than %u hits in one chunk", s->hit_cap);
		return BTBBX_E_NOMEM;
	}
	// order them while they are still in HBM (the next chunk is scanning on the other slot's stream)
	if (n) {
		int rc = btbbx_sort_hits_device(sl.d_hits, n, sl.stream);
		if (rc)
			return rc;
		if (hipMemcpy(sl.h_hits, sl.d_hits, (size_t)n * sizeof(btbbx_hit), hipMemcpyDeviceToHost) != hipSuccess)
			return hip_fail(hipGetLastError(), "d2h hits");
	}
	uint64_t out = 0;
	for (uint32_t i = 0; i < n; i++) {
		btbbx_hit h = sl.h_hits[i];
		// buffer offset 0 of a carried buffer was the last offset of the previous launch
		if (sl.first ? h.offset < 64 : h.offset == 0)
			continue;
		h.offset = sl.base + h.offset;
		if (out < cap)
			hits[out] = h;
		out++;
	}
	return (int64_t)out;
}

extern "C" void *btbbx_stream_acquire(btbbx_stream *s)
{
	if (!s || s->slot[s->cur].busy)
		return nullptr;
	return s->slot[s->cur].h_in;
}

extern "C" int64_t btbbx_stream_feed(btbbx_stream *s, const void *data, uint64_t n_symbols, btbbx_hit *hits, uint64_t cap)
{
	if (!s || !data || n_symbols == 0 || n_symbols > s->max_chunk) {
		set_error("btbbx_stream_feed: bad argument (chunk too large or empty)");
		return BTBBX_E_ARG;
	}
	void *dst = btbbx_stream_acquire(s);
	if (!dst) {
		set_error("btbbx_stream_feed: internal: slot still busy");
		return BTBBX_E_ARG;
	}
	// the staging copy is the slowest stage of feed(): split large chunks over a few host threads
	// (a byte-packed MSB-first buffer holds ceil(n / 8) bytes: nothing beyond them is read; submit() zeroes the rest of the last word)
	staging_copy(dst, data, s->format == BTBBX_FMT_SYMBOLS ? n_symbols : s->format == BTBBX_FMT_PACKED_MSB ? (n_symbols + 7) / 8 : ((n_symbols + 63) / 64) * 8);
	return btbbx_stream_submit(s, n_symbols, hits, cap);
}

extern "C" int64_t btbbx_stream_submit(btbbx_stream *s, uint64_t n_symbols, btbbx_hit *hits, uint64_t cap)
{
	if (!s || n_symbols == 0 || n_symbols > s->max_chunk || s->tail_fed) {
		set_error("btbbx_stream_submit: bad argument (chunk too large, empty, or fed after a ragged chunk)");
		return BTBBX_E_ARG;
	}
	Slot &sl = s->slot[s->cur];
	Slot &prev = s->slot[s->cur ^ 1];
	// the slot we are about to reuse was collected two feeds ago; make sure
	if (sl.busy) {
		set_error("btbbx_stream_feed: internal: slot still busy");
		return BTBBX_E_ARG;
	}
	const uint64_t chunk_words = (n_symbols + 63) / 64;
	const bool first = !s->started;
	// stage + copy + pack on this slot's stream
	if (s->format == BTBBX_FMT_SYMBOLS) {
		if (hipMemcpyAsync(sl.d_sym, sl.h_in, n_symbols, hipMemcpyHostToDevice, sl.stream) != hipSuccess)
			return hip_fail(hipGetLastError(), "h2d symbols");
		int rc = btbbx_pack_device(sl.d_sym, n_symbols, sl.d_words + 1, sl.stream);
		if (rc)
			return rc;
	} else {
		if (n_symbols & 63) {                                       // clear the unused tail of the last word
			if (s->format == BTBBX_FMT_PACKED) {
				((uint64_t *)sl.h_in)[chunk_words - 1] &= (1ULL << (n_symbols & 63)) - 1;
			} else {                                            // MSB first: symbol i is bit 7 - i % 8 of byte i / 8
				uint8_t *bytes = (uint8_t *)sl.h_in;
				const uint64_t full = n_symbols / 8, part = n_symbols & 7;
				if (part)
					bytes[full] &= (uint8_t)(0xff00u >> part);
				memset(bytes + full + (part ? 1 : 0), 0, chunk_words * 8 - full - (part ? 1 : 0));
			}
		}
		if (hipMemcpyAsync(sl.d_words + 1, sl.h_in, chunk_words * 8, hipMemcpyHostToDevice, sl.stream) != hipSuccess)
			return hip_fail(hipGetLastError(), "h2d words");
		// (an MSB-first chunk stays as it is: the scan turns its dwords round as it loads them; carry and zero word are
		// format-neutral -- a raw copy of the previous chunk's last word, and zeros)
	}
	// carry = last word of the previous chunk (its packing must have finished)
	if (first) {
		if (hipMemsetAsync(sl.d_words, 0, 8, sl.stream) != hipSuccess)
			return hip_fail(hipGetLastError(), "carry clear");
	} else {
		if (hipStreamSynchronize(prev.stream) != hipSuccess)   // prev scan done => prev words final
			return hip_fail(hipGetLastError(), "prev sync");
		if (hipMemcpyAsync(sl.d_words, prev.d_words + prev.n_words - 1, 8, hipMemcpyDeviceToDevice, sl.stream) != hipSuccess)
			return hip_fail(hipGetLastError(), "carry copy");
	}
	if (hipMemsetAsync(sl.d_words + 1 + chunk_words, 0, 8, sl.stream) != hipSuccess ||
	    hipMemsetAsync(sl.d_count, 0, 4, sl.stream) != hipSuccess)
		return hip_fail(hipGetLastError(), "memset");
	sl.n_words = 1 + chunk_words;
	sl.first = first;
	sl.base = s->fed - 64;                      // global offset of buffer bit 0 (wraps for the first chunk)
	// Buffer = [carry word][n symbols][zero word].  A window at buffer offset o is complete iff
	// o <= n; offset 0 was the last offset of the previous launch and offsets below 64 of the
	// first chunk start in the (fake) carry: collect() drops both.
	const uint64_t search_bits = n_symbols + 1;
	int rc = btbbx_scan_device_fmt(sl.d_words, sl.n_words + 1, sl.n_words + 1, 1, search_bits, s->lap, s->max_err,
				       s->format == BTBBX_FMT_PACKED_MSB ? BTBBX_FMT_PACKED_MSB : BTBBX_FMT_PACKED,
				       sl.d_hits, s->hit_cap, sl.d_count, sl.stream);
	if (rc)
		return rc;
	if (hipMemcpyAsync(sl.h_count, sl.d_count, 4, hipMemcpyDeviceToHost, sl.stream) != hipSuccess)
		return hip_fail(hipGetLastError(), "d2h count");
	sl.busy = true;
	s->started = true;
	s->fed += n_symbols;
	if (n_symbols & 63)
		s->tail_fed = true;
	s->cur ^= 1;
	// hand out the previous chunk's hits (already synchronised above when there was one)
	return collect(s, prev, hits, cap);
}

extern "C" int64_t btbbx_stream_flush(btbbx_stream *s, btbbx_hit *hits, uint64_t cap)
{
	if (!s)
		return BTBBX_E_ARG;
	// the slot fed last is cur^1; the other one was collected by the last feed
	int64_t n = collect(s, s->slot[s->cur ^ 1], hits, cap);
	return n;
}
