// btbb_api.cpp -- the drop-in C ABI of include/btbb.h (packet half).
//
// Same names, arguments, return values and ownership as libbtbb's public packet API
// (lib/src/btbb.h:63-151, 198; implementation lib/src/bluetooth_packet.c:201-208,
// 268-366, 444-542, 1198-1338, 1371-1408).  Host code here only keeps the packet object
// and moves bytes; every function that looks at symbols launches the HIP kernels of
// scan.hip / packet.hip.  There is no CPU fallback: without a GPU these functions print a
// diagnostic and report failure.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "packet_obj.h"
#include "../../include/btbb.h"

int launch_trials_state(const uint8_t *d_sym, const btbbx_pkt_in *d_in, const btbbx_pkt_out *d_out, void *d_state,
			btbbx_trial *d_trials, hipStream_t stream);
int launch_trials_merge(const void *d_state, const btbbx_pkt_in *d_in, btbbx_pkt_out *d_out, uint8_t *d_pay,
			const TrialPlan *plan, hipStream_t stream);
size_t trials_state_bytes();
int launch_decode_bytes(const uint8_t *d_sym, uint8_t *d_pay, const btbbx_pkt_in *d_in, btbbx_pkt_out *d_out,
			uint32_t mode, bool with_payload, hipStream_t stream);
int launch_decode(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
		  btbbx_pkt_out *d_out, uint32_t mode, const TrialPlan *plan, hipStream_t stream);

#ifndef BTBB_RELEASE
#define BTBB_RELEASE "mi355x-r1"
#endif
#ifndef BTBB_VERSION
#define BTBB_VERSION "1.0"
#endif

static const char *const type_names[16] = {
	"NULL", "POLL", "FHS", "DM1", "DH1/2-DH1", "HV1", "HV2/2-EV3", "HV3/EV3/3-EV3",
	"DV/3-DH1", "AUX1", "DM3/2-DH3", "DH3/3-DH3", "EV4/2-EV5", "EV5/3-EV5", "DM5/2-DH5", "DH5/3-DH5"
};

static int gpu_ready(const char *who)
{
	if (ctx().ready)
		return 1;
	// like the reference without btbb_init(): an empty syndrome map
	int rc = btbbx_init(0);
	if (rc) {
		fprintf(stderr, "%s: GPU path unavailable: %s\n", who, btbbx_last_error());
		return 0;
	}
	return 1;
}

extern "C" {

const char *btbb_get_release(void) { return BTBB_RELEASE; }   /* bluetooth_packet.c:268 */
const char *btbb_get_version(void) { return BTBB_VERSION; }   /* :275 */

/* bluetooth_packet.c:279-292 */
int btbb_init(int max_ac_errors)
{
	if (max_ac_errors < 0 || max_ac_errors > 5) {
		fprintf(stderr, "%s: max_ac_errors out of range\n", __FUNCTION__);
		return -1;
	}
	int rc = btbbx_init(max_ac_errors);
	if (rc) {
		fprintf(stderr, "%s: %s\n", __FUNCTION__, btbbx_last_error());
		return rc;
	}
	return 0;
}

/* :294-317 */
btbb_packet *btbb_packet_new(void)
{
	btbb_packet *pkt = (btbb_packet *)calloc(1, sizeof(btbb_packet));
	if (pkt)
		pkt->refcount = 1;
	else
		fprintf(stderr, "Unable to allocate packet");
	return pkt;
}

void btbb_packet_ref(btbb_packet *pkt) { pkt->refcount++; }

void btbb_packet_unref(btbb_packet *pkt)
{
	pkt->refcount--;
	if (pkt->refcount == 0)
		free(pkt);
}

/* :482-494 */
void btbb_packet_set_flag(btbb_packet *pkt, int flag, int val)
{
	uint32_t mask = 1u << flag;
	pkt->flags &= ~mask;
	if (val)
		pkt->flags |= mask;
}

int btbb_packet_get_flag(const btbb_packet *pkt, int flag) { return (pkt->flags & (1u << flag)) != 0; }

/* :319-366 */
uint32_t btbb_packet_get_lap(const btbb_packet *pkt) { return pkt->LAP; }
void btbb_packet_set_uap(btbb_packet *pkt, uint8_t uap)
{
	pkt->UAP = uap;
	btbb_packet_set_flag(pkt, BTBB_UAP_VALID, 1);
}
uint8_t btbb_packet_get_uap(const btbb_packet *pkt) { return pkt->UAP; }
uint16_t btbb_packet_get_nap(const btbb_packet *pkt) { return pkt->NAP; }
uint32_t btbb_packet_get_clkn(const btbb_packet *pkt) { return pkt->clkn; }
uint8_t btbb_packet_get_channel(const btbb_packet *pkt) { return pkt->channel; }
void btbb_packet_set_modulation(btbb_packet *pkt, uint8_t m) { pkt->modulation = m; }
uint8_t btbb_packet_get_modulation(const btbb_packet *pkt) { return pkt->modulation; }
void btbb_packet_set_transport(btbb_packet *pkt, uint8_t t) { pkt->transport = t; }
uint8_t btbb_packet_get_transport(const btbb_packet *pkt) { return pkt->transport; }
uint8_t btbb_packet_get_ac_errors(const btbb_packet *pkt) { return pkt->ac_errors; }

/* :188-199 -- once per search, host side (24 XORs); the kernels get the result */
uint64_t btbb_gen_syncword(const int LAP) { return host_gen_syncword((uint32_t)LAP); }

/* A caller that walks a buffer with btbb_find_ac -- search, take the match, search again one symbol further: the loop of
 * gr-bluetooth's and of every all-matches caller -- pays a full host <-> device round trip per call for a
 * window the GPU has just looked at.  Per thread, the library remembers the LAST window it scanned for ALL its matches
 * together with a copy of its symbols; a later call whose window lies inside it, with the same search parameters and --
 * compared byte for byte -- the same symbols, is answered from that list.  Exactly what a fresh scan would return: the
 * list holds every match of the window in order, and a window whose bytes have changed is scanned again.  A window is
 * scanned for all its matches only when the previous call ended at the same address (the caller is walking a buffer);
 * single calls take the first-match kernel as before.  Buffers beyond 1 Mi symbols are not remembered. */
#define AC_CACHE_MAX_SYMBOLS (1u << 20)
#define AC_CACHE_HITS 4096
struct AcWindow {
	const char *base = nullptr, *last_end = nullptr;
	uint64_t n_sym = 0;                  // symbols of the remembered window (search length + 63)
	uint32_t lap = 0;
	int max_err = 0, table_errors = -1;
	uint64_t found = 0;                  // matches the window has (hits holds the smallest of them)
	uint64_t n_hits = 0;
	char *copy = nullptr;
	size_t copy_bytes = 0;
	btbbx_hit *hits = nullptr;
	~AcWindow() { free(copy); free(hits); }
};
static thread_local AcWindow tl_ac;

/* 1 / 0 = answered (found / no match in the window), -1 = the remembered window cannot answer */
static int ac_window_lookup(const char *stream, uint64_t search_length, uint32_t lap, int max_err, btbbx_hit *first)
{
	AcWindow &w = tl_ac;
	if (!w.base || lap != w.lap || max_err != w.max_err || btbbx_table_errors() != w.table_errors)
		return -1;
	if (stream < w.base || stream + search_length + 63 > w.base + w.n_sym)
		return -1;
	const uint64_t delta = (uint64_t)(stream - w.base);
	uint64_t lo = 0, hi = w.n_hits;              // first remembered match at or behind `delta`
	while (lo < hi) {
		const uint64_t mid = (lo + hi) / 2;
		if (w.hits[mid].offset < delta)
			lo = mid + 1;
		else
			hi = mid;
	}
	const bool inside = lo < w.n_hits && w.hits[lo].offset < delta + search_length;
	if (!inside && lo == w.n_hits && w.found > w.n_hits)
		return -1;                               // behind the last one remembered: only known when the list is complete
	/* The answer rests on the symbols it was worked out from, and on no others: "first match at offset h" on the windows
	 * that start in [delta, h], i.e. symbols [delta, h + 64); "none" on the whole window.  Comparing just those keeps a
	 * walk over a buffer with N matches at one pass over the buffer in total (round 4 compared the rest of the window on
	 * every call: N passes). */
	const uint64_t depends = inside ? w.hits[lo].offset - delta + 64 : search_length + 63;
	if (memcmp(stream, w.copy + delta, depends) != 0)
		return -1;
	if (!inside)
		return 0;
	*first = w.hits[lo];
	first->offset -= delta;
	return 1;
}

/* scan [stream, stream + search_length + 63) for all its matches and remember it; 1 / 0 / negative error */
static int ac_window_scan(const char *stream, uint64_t search_length, uint32_t lap, int max_err, btbbx_hit *first)
{
	AcWindow &w = tl_ac;
	const uint64_t n_sym = search_length + 63;
	if (n_sym > w.copy_bytes) {
		free(w.copy);
		w.copy = (char *)malloc(n_sym);
		w.copy_bytes = w.copy ? n_sym : 0;
	}
	if (!w.hits)
		w.hits = (btbbx_hit *)malloc(sizeof(btbbx_hit) * AC_CACHE_HITS);
	w.base = nullptr;
	if (!w.copy || !w.hits)
		return BTBBX_E_NOMEM;
	memcpy(w.copy, stream, n_sym);
	const int64_t n = btbbx_scan_symbols(w.copy, n_sym, search_length, lap, max_err, w.hits, AC_CACHE_HITS);
	if (n < 0)
		return (int)n;
	w.base = stream;
	w.n_sym = n_sym;
	w.lap = lap;
	w.max_err = max_err;
	w.table_errors = btbbx_table_errors();
	w.found = (uint64_t)n;
	w.n_hits = (uint64_t)n < AC_CACHE_HITS ? (uint64_t)n : AC_CACHE_HITS;
	if (!n)
		return 0;
	*first = w.hits[0];
	return 1;
}

/* :444-464 -- first match through the GPU scan */
int btbb_find_ac(char *stream, int search_length, uint32_t lap, int max_ac_errors, btbb_packet **pkt_ptr)
{
	if (search_length <= 0)
		return -1;
	if (!gpu_ready("btbb_find_ac"))
		return -1;
	btbbx_hit first;
	const uint32_t xlap = lap == LAP_ANY ? BTBBX_LAP_ANY : lap;
	const char *end = stream + (size_t)search_length + 63;
	static const bool remember = []() { const char *e = getenv("BTBB_FIND_AC_WINDOW"); return !(e && e[0] == '0'); }();   // (0: measurements)
	int rc = remember ? ac_window_lookup(stream, (uint64_t)search_length, xlap, max_ac_errors, &first) : -1;
	if (rc < 0) {
		if (remember && tl_ac.last_end == end && (uint64_t)search_length + 63 <= AC_CACHE_MAX_SYMBOLS)
			rc = ac_window_scan(stream, (uint64_t)search_length, xlap, max_ac_errors, &first);     // the caller walks this buffer
		else
			rc = btbbx_find_first_symbols(stream, (uint64_t)search_length + 63, (uint64_t)search_length, xlap, max_ac_errors, &first);
	}
	tl_ac.last_end = end;
	if (rc < 0) {
		fprintf(stderr, "btbb_find_ac: GPU scan failed: %s\n", btbbx_last_error());
		return -1;
	}
	if (rc == 0)
		return -1;
	if (*pkt_ptr == NULL)
		*pkt_ptr = btbb_packet_new();
	/* init_packet, :201-208.  Known-LAP searches store the caller's 32-bit value. */
	(*pkt_ptr)->LAP = lap == LAP_ANY ? first.lap : lap;
	(*pkt_ptr)->ac_errors = first.ac_errors;
	(*pkt_ptr)->flags = 0;
	btbb_packet_set_flag(*pkt_ptr, BTBB_WHITENED, 1);
	return (int)first.offset;
}

/* :467-480 */
void btbb_packet_set_data(btbb_packet *pkt, char *data, int length, uint8_t channel, uint32_t clkn)
{
	if (length > PKT_MAX_SYMBOLS)
		length = PKT_MAX_SYMBOLS;
	if (length > 0)
		memcpy(pkt->symbols, data, (size_t)length);
	pkt->length = (uint16_t)length;
	pkt->channel = channel;
	pkt->clkn = clkn >> 1;
}

const char *btbb_get_symbols(const btbb_packet *pkt) { return pkt->symbols; }
int btbb_packet_get_payload_length(const btbb_packet *pkt) { return pkt->payload_length; }
const char *btbb_get_payload(const btbb_packet *pkt) { return pkt->payload; }

static uint32_t bits_of(const char *air, int n)
{
	uint32_t v = 0;
	for (int i = 0; i < n; i++)
		v |= (uint32_t)(uint8_t)air[i] << i;
	return v;
}

/* :511-517 -- format conversion of already decoded bits (no symbol processing) */
int btbb_get_payload_packed(const btbb_packet *pkt, char *dst)
{
	for (int i = 0; i < pkt->payload_length; i++)
		dst[i] = (char)(uint8_t)bits_of(pkt->payload + 8 * i, 8);
	return pkt->payload_length;
}

uint8_t btbb_packet_get_type(const btbb_packet *pkt) { return pkt->packet_type; }
uint8_t btbb_packet_get_lt_addr(const btbb_packet *pkt) { return pkt->packet_lt_addr; }
uint8_t btbb_packet_get_header_flags(const btbb_packet *pkt) { return pkt->packet_flags; }
uint8_t btbb_packet_get_hec(const btbb_packet *pkt) { return pkt->packet_hec; }
uint32_t btbb_packet_get_header_packed(const btbb_packet *pkt) { return bits_of(pkt->packet_header, 18); }

/* :1371-1408 */
int btbb_header_present(const btbb_packet *pkt)
{
	int present = 0;
	if (packet_gpu_decode(const_cast<btbb_packet *>(pkt), 0, nullptr, &present, nullptr, nullptr))
		return 0;
	return present;
}

/* :1198-1221 */
int btbb_decode_header(btbb_packet *pkt)
{
	int rv = 0;
	if (packet_gpu_decode(pkt, DEC_HEADER, nullptr, nullptr, &rv, nullptr))
		return 0;
	return rv;
}

/* :1223-1297 */
int btbb_decode_payload(btbb_packet *pkt)
{
	int rv = 0;
	if (packet_gpu_decode(pkt, DEC_PAYLOAD, nullptr, nullptr, nullptr, &rv))
		return 0;
	return rv;
}

/* :1320-1338 */
void btbb_print_packet(const btbb_packet *pkt)
{
	if (btbb_packet_get_flag(pkt, BTBB_HAS_PAYLOAD)) {
		printf("  Type: %s\n", type_names[pkt->packet_type & 15]);
		if (pkt->payload_header_length > 0) {
			printf("  LT_ADDR: %d\n", pkt->packet_lt_addr);
			printf("  LLID: %d\n", pkt->payload_llid);
			printf("  flow: %d\n", pkt->payload_flow);
			printf("  payload length: %d\n", pkt->payload_length);
		}
		if (pkt->payload_length) {
			printf("  Data: ");
			for (int i = 0; i < pkt->payload_length; i++)
				printf(" %02x", bits_of(pkt->payload + 8 * i, 8));
			printf("\n");
		}
	}
}

/* :1300-1317 */
int btbb_decode(btbb_packet *pkt)
{
	int hdr = 0, rv = 0;
	btbb_packet_set_flag(pkt, BTBB_HAS_PAYLOAD, 0);
	if (packet_gpu_decode(pkt, DEC_HEADER | DEC_PAYLOAD, nullptr, nullptr, &hdr, &rv))
		return 0;
	if (!hdr)
		rv = 0;
	if (rv > 0) {
		printf("Packet decoded with clock 0x%02x (rv=%d)\n", pkt->clkn & 0x3f, rv);
		btbb_print_packet(pkt);
	}
	return rv;
}

/* ---- symbols the reference library also exports (lib/src/bluetooth_packet.h:114-144) ---- */

/* try_clock, bluetooth_packet.c:1178-1195: one candidate clock through the GPU chain */
uint8_t try_clock(int clock, btbb_packet *pkt)
{
	TrialPlan plan = { 1ULL, 0ULL, (uint32_t)clock & 63 };
	int uap = 0;
	if (packet_gpu_decode(pkt, DEC_TRIALS, &plan, nullptr, &uap, nullptr))
		return 0;
	return (uint8_t)uap;
}

/* crc_check, bluetooth_packet.c:708-769 */
int crc_check(int clock, btbb_packet *pkt)
{
	TrialPlan plan = { 0ULL, 1ULL, (uint32_t)clock & 63 };
	int rv = 1;
	if (packet_gpu_decode(pkt, DEC_TRIALS, &plan, nullptr, nullptr, &rv))
		return 1;
	return rv;
}

/* FHS field extractors, bluetooth_packet.c:1411-1441: read already decoded payload bits */
uint32_t lap_from_fhs(btbb_packet *pkt) { return bits_of(pkt->payload + 34, 24); }
uint8_t uap_from_fhs(btbb_packet *pkt) { return (uint8_t)bits_of(pkt->payload + 64, 8); }
uint16_t nap_from_fhs(btbb_packet *pkt) { return (uint16_t)bits_of(pkt->payload + 72, 16); }
uint32_t clock_from_fhs(btbb_packet *pkt) { return bits_of(pkt->payload + 115, 26); }

/* tun_format, bluetooth_packet.c:1340-1368: 6 bytes of meta data, 3 of header, payload bytes;
 * malloc'd, the caller frees */
char *tun_format(btbb_packet *pkt)
{
	int length = 9 + pkt->payload_length;
	char *out = (char *)malloc((size_t)length);
	if (!out)
		return NULL;
	out[0] = (char)(pkt->clkn & 0xff);
	out[1] = (char)((pkt->clkn >> 8) & 0xff);
	out[2] = (char)((pkt->clkn >> 16) & 0xff);
	out[3] = (char)((pkt->clkn >> 24) & 0xff);
	out[4] = (char)pkt->channel;
	out[5] = (char)(btbb_packet_get_flag(pkt, BTBB_CLK27_VALID) | (btbb_packet_get_flag(pkt, BTBB_NAP_VALID) << 1));
	out[6] = (char)bits_of(pkt->packet_header, 7);
	out[7] = (char)bits_of(pkt->packet_header + 7, 3);
	out[8] = (char)bits_of(pkt->packet_header + 10, 8);
	for (int i = 0; i < pkt->payload_length; i++)
		out[9 + i] = (char)bits_of(pkt->payload + 8 * i, 8);
	return out;
}

} // extern "C"

// ---- GPU round trips for one packet object ---------------------------------------------------

// One device block and a pinned host mirror with the same layout.  Everything a call sends sits
// at the front (symbols | pkt_in | pkt_out | payload bits), so one asynchronous copy takes it in,
// the kernels and the result copy are queued behind it, and each call synchronises exactly once.
#define PB_SYM     0u          // 3200 symbol bytes
#define PB_IN      3200u       // btbbx_pkt_in (64 reserved)
#define PB_OUT     3264u       // btbbx_pkt_out (384)
#define PB_PAY     3648u       // 2752 payload bit bytes
#define PB_PKT     6400u       // 50 packed words
#define PB_TRIALS  6848u       // 64 btbbx_trial
#define PB_STATE   8192u       // 64 per-trial write sets (packet.hip: TrialState)
#define PB_TOTAL   (8192u + 32768u)
static_assert(sizeof(btbbx_pkt_out) <= PB_PAY - PB_OUT, "pkt_out slot");

struct DevPacketBufs {
	char *dev;             // device block
	char *host;            // pinned mirror
	uint8_t *d_sym;
	uint8_t *d_pay;
	uint64_t *d_pkt;
	btbbx_pkt_in *d_in;
	btbbx_pkt_out *d_out;
	btbbx_trial *d_trials;
};

static int dev_bufs(DevPacketBufs &b)          // inside a CallScope
{
	void *mirror = nullptr;
	void *block = scope_packet_block(&mirror, PB_TOTAL);
	if (!block)
		return BTBBX_E_NOMEM;
	b.dev = (char *)block;
	b.host = (char *)mirror;
	b.d_sym = (uint8_t *)(b.dev + PB_SYM);
	b.d_in = (btbbx_pkt_in *)(b.dev + PB_IN);
	b.d_out = (btbbx_pkt_out *)(b.dev + PB_OUT);
	b.d_pay = (uint8_t *)(b.dev + PB_PAY);
	b.d_pkt = (uint64_t *)(b.dev + PB_PKT);
	b.d_trials = (btbbx_trial *)(b.dev + PB_TRIALS);
	return BTBBX_OK;
}

static void fill_in(const btbb_packet *pkt, btbbx_pkt_in &in)
{
	memset(&in, 0, sizeof(in));
	in.length = pkt->length;
	in.clkn = pkt->clkn;
	in.flags = pkt->flags;
	in.uap = pkt->UAP;
	in.type = pkt->packet_type;
	in.llid = pkt->payload_llid;
	in.flow = pkt->payload_flow;
}

// stage the whole symbol array: FEC 2/3 may read past pkt->length (stale tail)
static void stage_symbols(const btbb_packet *pkt, DevPacketBufs &b)
{
	memcpy(b.host + PB_SYM, pkt->symbols, PKT_MAX_SYMBOLS);
	memset(b.host + PB_SYM + PKT_MAX_SYMBOLS, 0, 3200 - PKT_MAX_SYMBOLS);
}

// entry state of the decoders as btbbx_pkt_out (what a call may leave untouched)
static void fill_out(const btbb_packet *pkt, btbbx_pkt_out &out)
{
	memset(&out, 0, sizeof(out));
	out.payload_length = pkt->payload_length;
	out.payload_header_length = pkt->payload_header_length;
	out.header_packed = bits_of(pkt->packet_header, 18);
	out.type = pkt->packet_type;
	out.lt_addr = pkt->packet_lt_addr;
	out.hdr_flags = pkt->packet_flags;
	out.hec = pkt->packet_hec;
	out.payload_header = bits_of(pkt->payload_header, 16);
}

static void apply_out(btbb_packet *pkt, const btbbx_pkt_out &out, bool payload_too, const DevPacketBufs &b)
{
	if (payload_too)
		memcpy(pkt->payload, b.host + PB_PAY, PKT_MAX_PAYLOAD_BITS);
	pkt->flags = out.flags;
	pkt->UAP = out.uap;
	pkt->packet_type = out.type;
	pkt->packet_lt_addr = out.lt_addr;
	pkt->packet_flags = out.hdr_flags;
	pkt->packet_hec = out.hec;
	for (int i = 0; i < 18; i++)
		pkt->packet_header[i] = (char)((out.header_packed >> i) & 1);
	pkt->payload_header_length = out.payload_header_length;
	for (int i = 0; i < 16; i++)
		pkt->payload_header[i] = (char)((out.payload_header >> i) & 1);
	pkt->payload_llid = out.llid;
	pkt->payload_flow = out.flow;
	pkt->payload_length = out.payload_length;
}

// Step 1 of btbb_uap_from_header: all 64 trials, each with its packet writes captured on the
// device, and the {try_clock, type, crc_check} table back on the host.
int packet_gpu_trials(const btbb_packet *pkt, btbbx_trial *trials64)
{
	if (!gpu_ready("btbb_uap_from_header"))
		return BTBBX_E_NODEVICE;
	CallScope scope;        // the caller (btbb_uap_from_header) holds the outer scope: same lease in _commit
	hipStream_t q = scope_stream();
	DevPacketBufs b;
	int rc = dev_bufs(b);
	if (rc) return rc;
	if (trials_state_bytes() > PB_TOTAL - PB_STATE) {
		set_error("internal: trial state area too small");
		return BTBBX_E_ARG;
	}
	stage_symbols(pkt, b);
	btbbx_pkt_in in;
	fill_in(pkt, in);
	btbbx_pkt_out out;
	fill_out(pkt, out);
	memcpy(b.host + PB_IN, &in, sizeof(in));
	memcpy(b.host + PB_OUT, &out, sizeof(out));
	memcpy(b.host + PB_PAY, pkt->payload, PKT_MAX_PAYLOAD_BITS);
	memset(b.host + PB_PAY + PKT_MAX_PAYLOAD_BITS, 0, 2752 - PKT_MAX_PAYLOAD_BITS);
	HIP_TRY(hipMemcpyAsync(b.dev, b.host, PB_PKT, hipMemcpyHostToDevice, q));
	rc = launch_trials_state(b.d_sym, b.d_in, b.d_out, b.dev + PB_STATE, b.d_trials, q);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(b.host + PB_TRIALS, b.d_trials, 64 * sizeof(btbbx_trial), hipMemcpyDeviceToHost, q));
	HIP_TRY(hipStreamSynchronize(q));
	memcpy(trials64, b.host + PB_TRIALS, 64 * sizeof(btbbx_trial));
	return BTBBX_OK;
}

// Step 2: leave the packet as the trials in `plan` leave it in the reference.
// Must follow packet_gpu_trials() of the same packet directly: everything it needs is still on
// the device.
int packet_gpu_trials_commit(btbb_packet *pkt, const TrialPlan *plan)
{
	CallScope scope;
	hipStream_t q = scope_stream();
	DevPacketBufs b;
	int rc = dev_bufs(b);
	if (rc) return rc;
	rc = launch_trials_merge(b.dev + PB_STATE, b.d_in, b.d_out, b.d_pay, plan, q);
	if (rc) return rc;
	HIP_TRY(hipMemcpyAsync(b.host + PB_OUT, b.dev + PB_OUT, PB_PKT - PB_OUT, hipMemcpyDeviceToHost, q));
	HIP_TRY(hipStreamSynchronize(q));
	btbbx_pkt_out out;
	memcpy(&out, b.host + PB_OUT, sizeof(out));
	apply_out(pkt, out, true, b);
	return BTBBX_OK;
}

int packet_gpu_decode(btbb_packet *pkt, uint32_t mode, const TrialPlan *plan, int *header_present,
		      int *header_rv, int *payload_rv)
{
	if (!gpu_ready("btbb_decode"))
		return BTBBX_E_NODEVICE;
	CallScope scope;
	hipStream_t q = scope_stream();
	DevPacketBufs b;
	int rc = dev_bufs(b);
	if (rc) return rc;
	stage_symbols(pkt, b);

	btbbx_pkt_in in;
	fill_in(pkt, in);
	btbbx_pkt_out out;
	fill_out(pkt, out);
	memcpy(b.host + PB_IN, &in, sizeof(in));
	memcpy(b.host + PB_OUT, &out, sizeof(out));
	const bool touches_payload = mode & (DEC_PAYLOAD | DEC_TRIALS);
	if (touches_payload) {
		// current payload bits travel too: a decoder only overwrites a prefix
		memcpy(b.host + PB_PAY, pkt->payload, PKT_MAX_PAYLOAD_BITS);
		memset(b.host + PB_PAY + PKT_MAX_PAYLOAD_BITS, 0, 2752 - PKT_MAX_PAYLOAD_BITS);
	}
	HIP_TRY(hipMemcpyAsync(b.dev, b.host, touches_payload ? PB_PKT : PB_PAY, hipMemcpyHostToDevice, q));
	if (mode & DEC_TRIALS) {                 // single try_clock / crc_check calls (rare): separate steps
		rc = btbbx_pack_device(b.d_sym, 3200, b.d_pkt, q);
		if (rc) return rc;
		uint64_t *d_out_payload = (uint64_t *)((char *)b.d_out + offsetof(btbbx_pkt_out, payload));
		rc = btbbx_pack_device(b.d_pay, 2752, d_out_payload, q);
		if (rc) return rc;
		rc = launch_decode(b.d_pkt, b.d_in, 1, b.d_out, mode, plan, q);
		if (rc) return rc;
		rc = btbbx_unpack_device(d_out_payload, 2752, b.d_pay, q);
		if (rc) return rc;
	} else {                                 // pack + decode + unpack in one launch
		rc = launch_decode_bytes(b.d_sym, b.d_pay, b.d_in, b.d_out, mode, touches_payload, q);
		if (rc) return rc;
	}
	// pkt_out and (when touched) the payload bits are adjacent: one copy back
	HIP_TRY(hipMemcpyAsync(b.host + PB_OUT, b.dev + PB_OUT, (touches_payload ? PB_PKT : PB_PAY) - PB_OUT,
			       hipMemcpyDeviceToHost, q));
	HIP_TRY(hipStreamSynchronize(q));
	memcpy(&out, b.host + PB_OUT, sizeof(out));
	if (header_present) *header_present = out.header_present;
	if (header_rv) *header_rv = out.header_rv;
	if (payload_rv) *payload_rv = out.payload_rv;
	if (mode == 0)
		return BTBBX_OK;           // btbb_header_present: const, nothing written back

	apply_out(pkt, out, touches_payload, b);
	return BTBBX_OK;
}
