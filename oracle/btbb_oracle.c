/*
 * oracle/btbb_oracle.c -- CPU restatement of libbtbb's baseband hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see btbb_oracle.h).  Parity status: PINNED against
 * the reference's golden vectors and against the compiled reference
 * (oracle/_ref/libbtbb_ref.so); see tests/test_oracle_*.py.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * behaviour it restates.  The code is written fresh: tables are derived from the
 * Bluetooth baseband polynomials at start-up instead of being transcribed, the
 * syndrome map is an open-addressing table instead of uthash, and the FEC 2/3
 * decoder uses a derived syndrome->position table instead of a switch.  Quirks
 * of the reference that influence results are reproduced and marked "QUIRK".
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "btbb_oracle.h"

/* ------------------------------------------------------------------------- */
/* spec constants                                                            */
/* ------------------------------------------------------------------------- */

/* (64,30) sync-word block code generator, octal 0260534236651, degree 34
 * (bluetooth_packet.c:67-72, python/utils/gen_check_tables.py:5) */
#define SW_POLY 0260534236651ULL
/* 64-bit PN overlay (bluetooth_packet.c:115) */
#define SW_PN 0x83848D96BBCC54FCULL
/* 7-bit window = LAP MSB + 6 barker bits, host order (python/utils/gen_barker_correct.py:3,
 * bit-reversed because host bit 57 is the LSB of the window, bluetooth_packet.c:390) */
#define BARKER_LAP1 0x27u   /* LAP bit 23 = 1: barker 010011 above it */
#define BARKER_LAP0 0x58u   /* LAP bit 23 = 0: barker 101100 above it */
/* limits (bluetooth_packet.c:34-40, bluetooth_packet.h:33) */
#define AC_ERROR_LIMIT 5
#define MAX_BARKER_ERRORS 1
#define ID_THRESHOLD 5

static int tables_ready;
static uint64_t sw_col[64];          /* x^j mod g(x): syndrome of a single set bit j */
static uint64_t syn_byte[8][256];    /* per-byte syndrome tables (sw_check_tables.h equivalent) */
static uint64_t sw_rows[24];         /* generator rows per LAP bit, MSB first (bluetooth_packet.c:73) */
static uint64_t sw_default;          /* sync word of LAP 0 (DEFAULT_CODEWORD, :43) */
static uint8_t barker_dist[128];     /* :55 */
static uint64_t barker_fix[128];     /* :81 */
static uint8_t whiten_seq[127];      /* :52 */
static uint8_t whiten_idx[64];       /* :49 */
static uint16_t fec23_rows[10];      /* :117 */
static int8_t fec23_pos[32];         /* 5-bit syndrome -> data bit to flip, -1 parity/none, -2 uncorrectable */

static unsigned popcnt64(uint64_t v) { return (unsigned)__builtin_popcountll(v); }

static uint64_t encode30(uint32_t data30)
{
	uint64_t cw = 0;
	int b;
	for (b = 0; b < 30; b++)
		if ((data30 >> b) & 1)
			cw ^= (1ULL << (34 + b)) | sw_col[34 + b];
	return cw;
}

static uint64_t syncword_from_spec(uint32_t lap)
{
	/* python/utils/encode_sw.py:47-65: info = barker|LAP, xor PN, encode, xor PN */
	uint32_t info = ((lap & 0x800000) ? 0x13u : 0x2cu) << 24 | (lap & 0xffffff);
	info ^= (uint32_t)(SW_PN >> 34);
	return encode30(info) ^ SW_PN;
}

void orc_tables_init(void)
{
	int i, j, b;
	uint64_t c;

	if (tables_ready)
		return;

	/* columns of the check matrix: remainder of x^j by the generator */
	c = 1;
	for (j = 0; j < 64; j++) {
		sw_col[j] = c;
		c <<= 1;
		if (c & (1ULL << 34))
			c ^= SW_POLY;
	}
	for (b = 0; b < 8; b++)
		for (i = 0; i < 256; i++) {
			uint64_t s = 0;
			for (j = 0; j < 8; j++)
				if ((i >> j) & 1)
					s ^= sw_col[8 * b + j];
			syn_byte[b][i] = s;
		}

	sw_default = syncword_from_spec(0);
	for (i = 0; i < 24; i++)
		sw_rows[i] = syncword_from_spec(0x800000u >> i) ^ sw_default;

	for (i = 0; i < 128; i++) {
		unsigned d1 = popcnt64((uint64_t)(i ^ BARKER_LAP1));
		unsigned d0 = popcnt64((uint64_t)(i ^ BARKER_LAP0));
		barker_dist[i] = (uint8_t)(d1 < d0 ? d1 : d0);
		barker_fix[i] = (uint64_t)(d1 < d0 ? BARKER_LAP1 : BARKER_LAP0) << 57;
	}

	/* whitening: x^7 + x^4 + 1, register = 1 || CLK6..1, output = MSB.
	 * whiten_seq is one period started from register 0x7f-equivalent phase chosen so
	 * that whiten_seq[] equals the m-sequence, whiten_idx[clk] = phase of clock clk. */
	{
		uint8_t state = 0x7f, seen_state[127];
		for (i = 0; i < 127; i++) {
			uint8_t out = (state >> 6) & 1;
			seen_state[i] = state;
			whiten_seq[i] = out;
			state = (uint8_t)((state << 1) & 0x7f);
			if (out)
				state ^= 0x11;
		}
		for (i = 0; i < 64; i++)
			for (j = 0; j < 127; j++)
				if (seen_state[j] == (0x40 | i))
					whiten_idx[i] = (uint8_t)j;
	}

	/* FEC 2/3: (15,10) shortened Hamming, g(D) = D^5 + D^4 + D^2 + 1.
	 * air bit k <-> D^(14-k); parity bit j (air 10+j) = coeff of D^(4-j). */
	for (i = 0; i < 32; i++)
		fec23_pos[i] = -2;
	fec23_pos[0] = -1;
	for (i = 0; i < 5; i++)
		fec23_pos[1 << i] = -1;
	for (i = 0; i < 10; i++) {
		unsigned p = 1u << (14 - i), par = 0;
		int k;
		for (k = 14; k >= 5; k--)
			if ((p >> k) & 1)
				p ^= 0x35u << (k - 5);
		for (k = 0; k < 5; k++)
			if ((p >> (4 - k)) & 1)
				par |= 1u << k;
		fec23_rows[i] = (uint16_t)((par << 10) | (1u << i));
		fec23_pos[par] = (int8_t)i;
	}
	tables_ready = 1;
}

int orc_table(const char *name, uint64_t *dst, int cap)
{
	int i, n = -1;
	orc_tables_init();
#define COPY(arr, cnt) do { n = (cnt); for (i = 0; i < n && i < cap; i++) dst[i] = (uint64_t)(arr)[i]; } while (0)
	if (!strcmp(name, "INDICES")) COPY(whiten_idx, 64);
	else if (!strcmp(name, "WHITENING_DATA")) COPY(whiten_seq, 127);
	else if (!strcmp(name, "BARKER_DISTANCE")) COPY(barker_dist, 128);
	else if (!strcmp(name, "barker_correct")) COPY(barker_fix, 128);
	else if (!strcmp(name, "sw_matrix")) COPY(sw_rows, 24);
	else if (!strcmp(name, "fec23_gen_matrix")) COPY(fec23_rows, 10);
	else if (!strcmp(name, "sw_check_table4")) COPY(syn_byte[4], 256);
	else if (!strcmp(name, "sw_check_table5")) COPY(syn_byte[5], 256);
	else if (!strcmp(name, "sw_check_table6")) COPY(syn_byte[6], 256);
	else if (!strcmp(name, "sw_check_table7")) COPY(syn_byte[7], 256);
	else if (!strcmp(name, "pn")) { dst[0] = SW_PN; n = 1; }
	else if (!strcmp(name, "DEFAULT_CODEWORD")) { dst[0] = sw_default; n = 1; }
	else if (!strcmp(name, "DEFAULT_AC")) { dst[0] = orc_gen_syncword(0x9e8b33) ^ SW_PN; n = 1; }
#undef COPY
	return n;
}

/* ------------------------------------------------------------------------- */
/* access code                                                               */
/* ------------------------------------------------------------------------- */

/* bluetooth_packet.c:188-199 -- XOR of generator rows for the set LAP bits */
uint64_t orc_gen_syncword(int lap)
{
	uint64_t w;
	int i;
	orc_tables_init();
	w = sw_default;
	for (i = 0; i < 24; i++)
		if (lap & (0x800000 >> i))
			w ^= sw_rows[i];
	return w;
}

/* bluetooth_packet.c:147-159 -- 34-bit syndrome; bytes 0..3 map to themselves
 * (low 32 bits) because the parity part of the check matrix is the identity */
uint64_t orc_gen_syndrome(uint64_t cw)
{
	uint64_t s = 0;
	int b;
	orc_tables_init();
	for (b = 0; b < 8; b++)
		s ^= syn_byte[b][(cw >> (8 * b)) & 0xff];
	return s;
}

/* syndrome -> error pattern map (bluetooth_packet.c:121-185): open addressing */
static struct { uint64_t *key, *val; uint64_t mask; unsigned count; } smap;

static void smap_put(uint64_t syndrome, uint64_t error)
{
	uint64_t h = (syndrome * 0x9E3779B97F4A7C15ULL) >> 20;
	for (;; h++) {
		h &= smap.mask;
		if (smap.val[h] == 0) {
			smap.key[h] = syndrome;
			smap.val[h] = error;
			smap.count++;
			return;
		}
		if (smap.key[h] == syndrome)
			return;   /* cannot happen: minimum distance 14 (see NOTEBOOK.md 4) */
	}
}

int orc_find_syndrome(uint64_t syndrome, uint64_t *error)
{
	uint64_t h;
	if (!smap.val)
		return 0;
	h = (syndrome * 0x9E3779B97F4A7C15ULL) >> 20;
	for (;; h++) {
		h &= smap.mask;
		if (smap.val[h] == 0)
			return 0;
		if (smap.key[h] == syndrome) {
			*error = smap.val[h];
			return 1;
		}
	}
}

/* :161-178 -- every error pattern of exactly `depth` more bits, positions start..57 */
static void enumerate(uint64_t error, int start, int depth)
{
	int i;
	for (i = start; i < 58; i++) {
		uint64_t e = error | (1ULL << i);
		if (depth > 1)
			enumerate(e, i + 1, depth - 1);
		else
			smap_put(orc_gen_syndrome(e), e);   /* gen_syndrome(DEFAULT_AC)==0, so key = H*e */
	}
}

unsigned orc_syndrome_count(void) { return smap.count; }

void orc_reset_syndrome_map(void)
{
	free(smap.key);
	free(smap.val);
	memset(&smap, 0, sizeof(smap));
}

/* bluetooth_packet.c:279-292.  QUIRK: the map is built by the FIRST
 * call with max_ac_errors > 0 and never rebuilt. */
int orc_init(int max_ac_errors)
{
	int k;
	orc_tables_init();
	if (max_ac_errors < 0 || max_ac_errors > AC_ERROR_LIMIT)
		return -1;
	if (!smap.val && max_ac_errors) {
		uint64_t slots = 64, entries = 0, binom = 1;
		for (k = 1; k <= max_ac_errors; k++) {
			binom = binom * (uint64_t)(58 - k + 1) / (uint64_t)k;   /* C(58,k) */
			entries += binom;
		}
		while (slots < 3 * entries)
			slots <<= 1;
		smap.key = calloc(slots, sizeof(uint64_t));
		smap.val = calloc(slots, sizeof(uint64_t));
		smap.mask = slots - 1;
		for (k = 1; k <= max_ac_errors; k++)
			enumerate(0, 0, k);
	}
	return 0;
}

/* bluetooth_packet.c:235-242; (T)air[i] << i OR-ed, no masking of the byte (QUIRK Q9) */
static uint64_t air_bits(const char *air, int bits)
{
	uint64_t v = 0;
	int i;
	for (i = 0; i < bits; i++)
		v |= (uint64_t)(uint8_t)air[i] << i;
	return v;
}

/* one offset of promiscuous_packet_search (bluetooth_packet.c:385-416).
 * Returns 1 on acceptance. */
static int lap_any_at(const char *sym, int max_ac_errors, uint32_t *lap, uint8_t *ac_errors)
{
	uint64_t w, cw, syn, err;
	unsigned barker = (unsigned)air_bits(sym + 57, 7);
	uint8_t nerr = 0;

	if (barker_dist[barker] > MAX_BARKER_ERRORS)
		return 0;
	w = air_bits(sym, 64);
	/* QUIRK Q1: bits 57..63 replaced, barker errors not counted */
	w = (w & 0x01ffffffffffffffULL) | barker_fix[w >> 57];
	cw = w ^ SW_PN;
	syn = orc_gen_syndrome(cw);
	if (syn) {
		if (orc_find_syndrome(syn, &err)) {
			w ^= err;                       /* QUIRK Q2: may flip bit 57 again */
			nerr = (uint8_t)popcnt64(err);
		} else {
			nerr = 0xff;
		}
	}
	*ac_errors = nerr;
	if ((int)nerr <= max_ac_errors) {
		*lap = (uint32_t)((w >> 34) & 0xffffff);
		return 1;
	}
	return 0;
}

/* bluetooth_packet.c:444-464 minus packet allocation */
int orc_find_ac(const char *stream, int search_length, uint32_t lap,
		int max_ac_errors, uint32_t *lap_out, uint8_t *ac_errors_out)
{
	int c;
	orc_tables_init();
	if (lap == ORC_LAP_ANY) {
		/* promiscuous_packet_search :368-420 */
		for (c = 0; c < search_length; c++) {
			uint32_t found;
			uint8_t nerr;
			if (lap_any_at(stream + c, max_ac_errors, &found, &nerr)) {
				*lap_out = found;
				*ac_errors_out = nerr;
				return c;
			}
		}
	} else {
		/* find_known_lap :423-441 -- all 64 bits count */
		uint64_t ac = orc_gen_syncword((int)lap);
		for (c = 0; c < search_length; c++) {
			unsigned nerr = popcnt64(air_bits(stream + c, 64) ^ ac);
			if ((int)nerr <= max_ac_errors) {
				*lap_out = lap;
				*ac_errors_out = (uint8_t)nerr;
				return c;
			}
		}
	}
	return -1;
}

size_t orc_find_all(const char *stream, uint64_t search_length, uint32_t lap,
		    int max_ac_errors, orc_hit *out, size_t cap)
{
	uint64_t c;
	size_t n = 0;
	uint64_t ac = 0;
	orc_tables_init();
	if (lap != ORC_LAP_ANY)
		ac = orc_gen_syncword((int)lap);
	for (c = 0; c < search_length; c++) {
		uint32_t found = lap;
		uint8_t nerr;
		int ok;
		if (lap == ORC_LAP_ANY) {
			ok = lap_any_at(stream + c, max_ac_errors, &found, &nerr);
		} else {
			nerr = (uint8_t)popcnt64(air_bits(stream + c, 64) ^ ac);
			ok = (int)nerr <= max_ac_errors;
		}
		if (ok) {
			if (n < cap) {
				memset(&out[n], 0, sizeof(out[n]));
				out[n].offset = c;
				out[n].lap = found;
				out[n].ac_errors = nerr;
			}
			n++;
		}
	}
	return n;
}

/* ------------------------------------------------------------------------- */
/* bit chain                                                                 */
/* ------------------------------------------------------------------------- */

/* bluetooth_packet.c:552-568 */
int orc_unfec13(const char *in, char *out, int length)
{
	int i, disagreements = 0;
	for (i = 0; i < length; i++) {
		char a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
		out[i] = (char)((a & b) | (b & c) | (c & a));
		disagreements += ((a ^ b) | (b ^ c) | (c ^ a));
	}
	return disagreements < (length / 4);
}

/* bluetooth_packet.c:571-582 */
uint16_t orc_fec23(uint16_t data)
{
	uint16_t cw = 0;
	int i;
	orc_tables_init();
	for (i = 0; i < 10; i++)
		if (data & (1 << i))
			cw ^= fec23_rows[i];
	return cw;
}

/* bluetooth_packet.c:585-649.  `length` = data bits before encoding; padded up to
 * a multiple of 10; one 15-symbol block per 10 data bits; a syndrome that is
 * neither zero, a parity-bit error nor one of the ten data-bit columns fails the
 * whole call.  out must hold the padded length. */
int orc_unfec23(const char *in, int length, char *out)
{
	int blocks, k, i;
	orc_tables_init();
	if (length % 10)
		length += 10 - (length % 10);
	blocks = length / 10;
	for (k = 0; k < blocks; k++) {
		const char *blk = in + 15 * k;
		char *o = out + 10 * k;
		uint16_t data = (uint16_t)air_bits(blk, 10);
		uint8_t check = (uint8_t)air_bits(blk + 10, 5);
		uint8_t diff = (uint8_t)(check ^ (orc_fec23(data) >> 10));
		for (i = 0; i < 10; i++)
			o[i] = blk[i];
		if (diff & (diff - 1)) {
			/* QUIRK: `check` is a uint8_t built from 5 bytes; with clean 0/1 symbols
			 * diff < 32.  Values >= 32 cannot occur for 0/1 input. */
			int pos = diff < 32 ? fec23_pos[diff] : -2;
			if (pos < 0)
				return 0;
			o[pos] ^= 1;
		}
	}
	return 1;
}

/* bluetooth_packet.c:653-668 */
void orc_unwhiten(const char *in, char *out, int clock, int length, int skip, int whitened)
{
	int i, idx;
	orc_tables_init();
	idx = (whiten_idx[clock & 0x3f] + skip) % 127;
	for (i = 0; i < length; i++) {
		out[i] = whitened ? (char)(in[i] ^ whiten_seq[idx]) : in[i];
		idx = (idx + 1) % 127;
	}
}

static uint8_t bitrev8(uint8_t b)
{
	b = (uint8_t)((b >> 4) | (b << 4));
	b = (uint8_t)(((b & 0xcc) >> 2) | ((b & 0x33) << 2));
	b = (uint8_t)(((b & 0xaa) >> 1) | ((b & 0x55) << 1));
	return b;
}

/* bluetooth_packet.c:671-690 -- CRC-CCITT run LSB-first, seeded with reversed UAP.
 * QUIRK: the loop counter is a uint16_t compared with an int length, so a negative
 * length runs zero iterations. */
uint16_t orc_crcgen(const char *bits, int length, int uap)
{
	uint16_t reg = (uint16_t)((bitrev8((uint8_t)uap) << 8) & 0xff00);
	uint16_t n;
	for (n = 0; (int)n < length; n++) {
		uint16_t fb = (uint16_t)((reg & 1) ^ (bits[n] & 1));
		reg = (uint16_t)((reg >> 1) | (fb << 15));
		reg ^= (uint16_t)((reg & 0x8000) >> 5);
		reg ^= (uint16_t)((reg & 0x8000) >> 12);
	}
	return reg;
}

/* bluetooth_packet.c:693-705 -- HEC LFSR run backwards over the 10 header bits */
uint8_t orc_uap_from_hec(uint16_t data, uint8_t hec)
{
	int i;
	for (i = 9; i >= 0; i--) {
		if (hec & 0x80)
			hec ^= 0x65;
		hec = (uint8_t)((hec << 1) | (((hec >> 7) ^ (data >> i)) & 1));
	}
	return bitrev8(hec);
}

/* forward direction of the above (no reference counterpart; used by synthetic TX) */
uint8_t orc_hec_from_uap(uint16_t data, uint8_t uap)
{
	uint8_t reg = bitrev8(uap);
	int i;
	for (i = 0; i < 10; i++) {
		uint8_t msb = (uint8_t)((reg & 1) ^ ((data >> i) & 1));
		reg = (uint8_t)((reg >> 1) | (msb << 7));
		if (msb)
			reg ^= 0x65;
	}
	return reg;
}

/* ------------------------------------------------------------------------- */
/* packet object                                                             */
/* ------------------------------------------------------------------------- */

orc_packet *orc_packet_new(void) { return calloc(1, sizeof(orc_packet)); }   /* :294 */
void orc_packet_free(orc_packet *p) { free(p); }

void orc_packet_set_flag(orc_packet *p, int flag, int val)      /* :482 */
{
	uint32_t m = 1u << flag;
	p->flags = val ? (p->flags | m) : (p->flags & ~m);
}
int orc_packet_get_flag(const orc_packet *p, int flag) { return (p->flags >> flag) & 1; }

void orc_packet_init_found(orc_packet *p, uint32_t lap, uint8_t ac_errors)   /* :201-208 */
{
	p->LAP = lap;
	p->ac_errors = ac_errors;
	p->flags = 0;
	orc_packet_set_flag(p, ORC_WHITENED, 1);
}

void orc_packet_set_data(orc_packet *p, const char *syms, int length, uint8_t channel, uint32_t clkn)  /* :467-480 */
{
	if (length > ORC_MAX_SYMBOLS)
		length = ORC_MAX_SYMBOLS;
	if (length > 0)
		memcpy(p->symbols, syms, (size_t)length);
	p->length = (uint16_t)length;
	p->channel = channel;
	p->clkn = clkn >> 1;
}

uint32_t orc_packet_header_packed(const orc_packet *p) { return (uint32_t)air_bits(p->packet_header, 18); }

int orc_payload_packed(const orc_packet *p, char *dst)   /* :511-517 */
{
	int i;
	for (i = 0; i < p->payload_length; i++)
		dst[i] = (char)(uint8_t)air_bits(p->payload + 8 * i, 8);
	return p->payload_length;
}

/* bluetooth_packet.c:1371-1408 */
int orc_header_present(const orc_packet *p)
{
	const char *s = p->symbols + 63;
	int errs = 0, k;
	char msb;
	if (p->length < 122)
		return 0;
	msb = s[0];
	errs += s[1] ^ !msb;
	errs += s[2] ^ msb;
	errs += s[3] ^ !msb;
	errs += s[4] ^ msb;
	s += 5;
	for (k = 0; k < 54; k += 3)
		errs += (s[k] ^ s[k + 1]) | (s[k + 1] ^ s[k + 2]) | (s[k + 2] ^ s[k]);
	return errs < ID_THRESHOLD;
}

static int is_whitened(const orc_packet *p) { return orc_packet_get_flag(p, ORC_WHITENED); }

/* bluetooth_packet.c:1178-1195.  QUIRK Q5: on FEC failure returns 0 and leaves
 * UAP / packet_type untouched. */
uint8_t orc_try_clock(int clock, orc_packet *p)
{
	char hdr[18], clear[18];
	if (!orc_unfec13(p->symbols + 68, hdr, 18))
		return 0;
	orc_unwhiten(hdr, clear, clock, 18, 0, is_whitened(p));
	p->UAP = orc_uap_from_hec((uint16_t)air_bits(clear, 10), (uint8_t)air_bits(clear + 10, 8));
	p->packet_type = (uint8_t)air_bits(clear + 3, 4);
	return p->UAP;
}

/* bluetooth_packet.c:1198-1221 */
int orc_decode_header(orc_packet *p)
{
	char hdr[18];
	if (orc_packet_get_flag(p, ORC_CLK6_VALID) && orc_unfec13(p->symbols + 68, hdr, 18)) {
		uint8_t hec, uap;
		orc_unwhiten(hdr, p->packet_header, (int)p->clkn, 18, 0, is_whitened(p));
		hec = (uint8_t)air_bits(p->packet_header + 10, 8);
		uap = orc_uap_from_hec((uint16_t)air_bits(p->packet_header, 10), hec);
		if (uap == p->UAP) {
			p->packet_lt_addr = (uint8_t)air_bits(p->packet_header, 3);
			p->packet_type = (uint8_t)air_bits(p->packet_header + 3, 4);
			p->packet_flags = (uint8_t)air_bits(p->packet_header + 7, 3);
			p->packet_hec = hec;
			return 1;
		}
	}
	return 0;
}

/* bluetooth_packet.c:772-781.  QUIRK Q7: for payload_length < 2 the bit count is
 * negative (CRC = seed) and the check word is read from in front of payload[]:
 * with the reference's struct layout (bluetooth_packet.h:79-99) the 8 chars before
 * payload[0] are payload_llid, payload_flow, two padding bytes (zero: objects come
 * from calloc) and the four little-endian bytes of payload_length. */
static int payload_crc_ok(const orc_packet *p)
{
	int nbits = (p->payload_length - 2) * 8;
	uint16_t crc = orc_crcgen(p->payload, nbits, p->UAP);
	uint16_t check;
	if (nbits >= 0) {
		check = (uint16_t)air_bits(p->payload + nbits, 16);
	} else {
		char shadow[16 + 16];
		uint32_t plen = (uint32_t)p->payload_length;
		int i, start;
		memset(shadow, 0, sizeof(shadow));
		memcpy(shadow, p->payload_header + 8, 8);
		shadow[8] = (char)p->payload_llid;
		shadow[9] = (char)p->payload_flow;
		for (i = 0; i < 4; i++)
			shadow[12 + i] = (char)((plen >> (8 * i)) & 0xff);
		memcpy(shadow + 16, p->payload, 16);
		start = 16 + nbits;                     /* nbits is -8 or -16 */
		check = 0;
		for (i = 0; i < 16; i++)
			check |= (uint16_t)((uint16_t)(uint8_t)shadow[start + i] << i);
	}
	return crc == check;
}

/* bluetooth_packet.c:783-818 */
int orc_fhs(int clock, orc_packet *p)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	char corrected[160];
	int c;

	p->payload_length = 20;
	if (size < p->payload_length * 12)
		return 1;
	if (!orc_unfec23(stream, 160, corrected))
		return 0;
	orc_unwhiten(corrected, p->payload, clock, 160, 18, is_whitened(p));
	if (payload_crc_ok(p))
		return 1000;
	for (c = 32; c < 64; c++) {
		orc_unwhiten(corrected, p->payload, c, 160, 18, is_whitened(p));
		if (payload_crc_ok(p))
			return 1000;
	}
	return 0;
}

/* bluetooth_packet.c:821-895 */
static int payload_header(const char *stream, int clock, int header_bytes, int size, int fec, orc_packet *p)
{
	int hbits = header_bytes == 2 ? 16 : 8;
	int cap;
	if (size < hbits)
		return 0;
	if (fec) {
		char corrected[20];
		if (size < (header_bytes == 2 ? 30 : 15))
			return 0;
		if (!orc_unfec23(stream, hbits, corrected))
			return 0;
		orc_unwhiten(corrected, p->payload_header, clock, hbits, 18, is_whitened(p));
	} else {
		orc_unwhiten(stream, p->payload_header, clock, hbits, 18, is_whitened(p));
	}
	if (header_bytes == 2)
		p->payload_length = (int)air_bits(p->payload_header + 3, 10) + 4;
	else
		p->payload_length = (int)air_bits(p->payload_header + 3, 5) + 3;

	/* QUIRK Q6: types outside this list clamp to 0 */
	switch (p->packet_type) {
	case 3:  cap = 20;  break;   /* DM1 */
	case 4:  cap = 30;  break;   /* DH1 */
	case 8:  cap = 12;  break;   /* DV  */
	case 10: cap = 125; break;   /* DM3 */
	case 11: cap = 187; break;   /* DH3 */
	case 14: cap = 228; break;   /* DM5 */
	case 15: cap = 343; break;   /* DH5 */
	default: cap = 0;   break;
	}
	if (p->payload_length > cap)
		p->payload_length = cap;
	p->payload_llid = (uint8_t)air_bits(p->payload_header, 2);
	p->payload_flow = (uint8_t)air_bits(p->payload_header + 2, 1);
	p->payload_header_length = header_bytes;
	return 1;
}

/* bluetooth_packet.c:898-958 */
int orc_DM(int clock, orc_packet *p)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	int header_bytes = 2, max_length, nbits;
	char corrected[ORC_MAX_PAYLOAD_BITS + 16];

	switch (p->packet_type) {
	case 8:  stream += 80; size -= 80; header_bytes = 1; max_length = 12; break;
	case 3:  header_bytes = 1; max_length = 20; break;
	case 10: max_length = 125; break;
	case 14: max_length = 228; break;
	default: return 0;
	}
	if (!payload_header(stream, clock, header_bytes, size, 1, p))
		return 0;
	if (p->payload_length > max_length)
		return 1;
	nbits = p->payload_length * 8;
	if (nbits > size)
		return 1;
	/* QUIRK: reads 15*ceil(nbits/10) symbols, possibly past p->length (stale tail) */
	if (!orc_unfec23(stream, nbits, corrected))
		return 0;
	orc_unwhiten(corrected, p->payload, clock, nbits, 18, is_whitened(p));
	return payload_crc_ok(p) ? 10 : 2;
}

/* bluetooth_packet.c:962-1011 */
int orc_DH(int clock, orc_packet *p)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	int header_bytes = 2, max_length, nbits;

	switch (p->packet_type) {
	case 9:
	case 4:  header_bytes = 1; max_length = 30; break;
	case 11: max_length = 187; break;
	case 15: max_length = 343; break;
	default: return 0;
	}
	if (!payload_header(stream, clock, header_bytes, size, 0, p))
		return 0;
	if (p->payload_length > max_length)
		return 1;
	nbits = p->payload_length * 8;
	if (nbits > size)
		return 1;
	orc_unwhiten(stream, p->payload, clock, nbits, 18, is_whitened(p));
	if (p->packet_type == 9)
		return 2;
	return payload_crc_ok(p) ? 10 : 2;
}

/* EV3 :1013-1042 and EV5 :1099-1128 differ only in the byte limit */
static int ev_bytewise(int clock, orc_packet *p, int maxlength)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	for (p->payload_length = 0; p->payload_length < maxlength; p->payload_length++) {
		int bits = p->payload_length * 8;
		if (bits + 8 > size)
			return 1;
		/* QUIRK: input pointer is NOT advanced -- every byte is the first 8
		 * payload symbols XOR a later part of the whitening sequence */
		orc_unwhiten(stream, p->payload + bits, clock, 8, 18 + bits, is_whitened(p));
		if (p->payload_length > 2 && payload_crc_ok(p))
			return 10;
	}
	return 2;
}
int orc_EV3(int clock, orc_packet *p) { return ev_bytewise(clock, p, 32); }
int orc_EV5(int clock, orc_packet *p) { return ev_bytewise(clock, p, 182); }

/* bluetooth_packet.c:1044-1097 */
int orc_EV4(int clock, orc_packet *p)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	int syms = 0, bits = 0;
	char corrected[10];

	p->payload_length = 1;
	while (syms < 1470) {
		if (syms + 15 > size)
			return 1;
		if (!orc_unfec23(stream + syms, 10, corrected))
			return syms < 45 ? 0 : 1;
		orc_unwhiten(corrected, p->payload + bits, clock, 10, 18 + bits, is_whitened(p));
		while (p->payload_length * 8 <= bits) {
			if (payload_crc_ok(p))
				return 10;
			p->payload_length++;
		}
		syms += 15;
		bits += 10;
	}
	return 2;
}

/* bluetooth_packet.c:1131-1174 */
int orc_HV(int clock, orc_packet *p)
{
	const char *stream = p->symbols + 122;
	int size = p->length - 122;
	char corrected[160];

	p->payload_header_length = 0;
	if (size < 240) {
		p->payload_length = 0;
		return 1;
	}
	switch (p->packet_type) {
	case 5:
		if (!orc_unfec13(stream, corrected, 80))
			return 0;
		p->payload_length = 10;
		orc_packet_set_flag(p, ORC_HAS_PAYLOAD, 1);
		orc_unwhiten(corrected, p->payload, clock, 80, 18, is_whitened(p));
		break;
	case 6:
		if (!orc_unfec23(stream, 160, corrected))
			return 0;
		p->payload_length = 20;
		orc_packet_set_flag(p, ORC_HAS_PAYLOAD, 1);
		orc_unwhiten(corrected, p->payload, clock, 160, 18, is_whitened(p));
		break;
	case 7:
		p->payload_length = 30;
		orc_packet_set_flag(p, ORC_HAS_PAYLOAD, 1);
		orc_unwhiten(stream, p->payload, clock, 240, 18, is_whitened(p));
		break;
	}
	return 2;
}

/* bluetooth_packet.c:708-769 */
int orc_crc_check(int clock, orc_packet *p)
{
	int rv = 1;
	switch (p->packet_type) {
	case 2:  rv = orc_fhs(clock, p); break;
	case 8: case 3: case 10: case 14: rv = orc_DM(clock, p); break;
	case 4: case 11: case 15: rv = orc_DH(clock, p); break;
	case 7:  rv = orc_EV3(clock, p); break;
	case 12: rv = orc_EV4(clock, p); break;
	case 13: rv = orc_EV5(clock, p); break;
	case 5:  rv = orc_HV(clock, p); break;
	default: break;
	}
	if (rv == 0 && p->packet_type != 2 && p->packet_type != 3 && p->packet_type != 5)
		return 1;
	if (rv > 1 && (p->packet_type == 7 || p->packet_type == 13))
		return 1;
	return rv;
}

/* bluetooth_packet.c:1223-1297 */
int orc_decode_payload(orc_packet *p)
{
	int rv = 0;
	int clock = (int)p->clkn;
	p->payload_header_length = 0;
	switch (p->packet_type) {
	case 0: case 1: p->payload_length = 0; rv = 1; break;
	case 2:  rv = orc_fhs(clock, p); break;
	case 3: case 8: case 10: case 14: rv = orc_DM(clock, p); break;
	case 4: case 9: case 11: case 15: rv = orc_DH(clock, p); break;
	case 5: case 6: rv = orc_HV(clock, p); break;
	case 7:
		rv = orc_EV3(clock, p);
		if (rv <= 1)
			rv = orc_HV(clock, p);
		break;
	case 12: rv = orc_EV4(clock, p); break;
	case 13: rv = orc_EV5(clock, p); break;
	}
	orc_packet_set_flag(p, ORC_HAS_PAYLOAD, 1);
	return rv;
}

/* bluetooth_packet.c:1300-1317 without the printf side effect */
int orc_decode(orc_packet *p)
{
	int rv = 0;
	orc_packet_set_flag(p, ORC_HAS_PAYLOAD, 0);
	if (orc_decode_header(p))
		rv = orc_decode_payload(p);
	return rv;
}

/* bluetooth_packet.c:1411-1441 */
uint32_t orc_lap_from_fhs(const orc_packet *p) { return (uint32_t)air_bits(p->payload + 34, 24); }
uint8_t orc_uap_from_fhs(const orc_packet *p) { return (uint8_t)air_bits(p->payload + 64, 8); }
uint16_t orc_nap_from_fhs(const orc_packet *p) { return (uint16_t)air_bits(p->payload + 72, 16); }
uint32_t orc_clock_from_fhs(const orc_packet *p) { return (uint32_t)air_bits(p->payload + 115, 26); }

/* ------------------------------------------------------------------------- */
/* piconet: the callers of the path                                          */
/* ------------------------------------------------------------------------- */

orc_piconet *orc_piconet_new(void) { return calloc(1, sizeof(orc_piconet)); }
void orc_piconet_free(orc_piconet *pn) { free(pn); }

void orc_piconet_set_flag(orc_piconet *pn, int flag, int val)
{
	uint32_t m = 1u << flag;
	pn->flags = val ? (pn->flags | m) : (pn->flags & ~m);
}
int orc_piconet_get_flag(const orc_piconet *pn, int flag) { return (pn->flags >> flag) & 1; }

void orc_init_piconet(orc_piconet *pn, uint32_t lap)   /* bluetooth_piconet.c:70-74 */
{
	pn->LAP = lap;
	orc_piconet_set_flag(pn, ORC_LAP_VALID, 1);
}

static void channel_seen(orc_piconet *pn, uint8_t ch)   /* :133-141 */
{
	if (!(pn->afh_map[ch / 8] & (1 << (ch % 8)))) {
		pn->afh_map[ch / 8] |= (uint8_t)(1 << (ch % 8));
		pn->used_channels++;
	}
}

void orc_piconet_reset(orc_piconet *pn)   /* :547-572 */
{
	if (orc_piconet_get_flag(pn, ORC_HOP_REVERSAL_INIT)) {
		free(pn->clock_candidates);
		pn->clock_candidates = NULL;   /* the reference leaves it dangling; never read again before re-init */
		pn->sequence = NULL;
	}
	orc_piconet_set_flag(pn, ORC_GOT_FIRST_PACKET, 0);
	orc_piconet_set_flag(pn, ORC_HOP_REVERSAL_INIT, 0);
	orc_piconet_set_flag(pn, ORC_UAP_VALID, 0);
	orc_piconet_set_flag(pn, ORC_CLK6_VALID, 0);
	orc_piconet_set_flag(pn, ORC_CLK27_VALID, 0);
	pn->packets_observed = 0;
	orc_piconet_set_flag(pn, ORC_IS_AFH, orc_piconet_get_flag(pn, ORC_LOOKS_LIKE_AFH));
}

/* bluetooth_piconet.c:648-750 (prints dropped) */
int orc_uap_from_header(orc_packet *p, orc_piconet *pn)
{
	int count, remaining = 0, first_clock = 0;
	uint32_t clkn = p->clkn;

	if (!orc_piconet_get_flag(pn, ORC_GOT_FIRST_PACKET))
		pn->first_pkt_time = clkn;
	channel_seen(pn, p->channel);
	if (pn->packets_observed < 1000) {
		pn->pattern_indices[pn->packets_observed] = (int)(clkn - pn->first_pkt_time);
		pn->pattern_channels[pn->packets_observed] = p->channel;
	} else {
		orc_piconet_reset(pn);
		return 0;
	}
	pn->packets_observed++;
	pn->total_packets_observed++;

	for (count = 0; count < 64; count++) {
		int first = !orc_piconet_get_flag(pn, ORC_GOT_FIRST_PACKET);
		if (pn->clock6_candidates[count] > -1 || first) {
			/* QUIRK: (int + uint32 - uint32) % 64 is evaluated in unsigned arithmetic */
			int clock = (int)(((uint32_t)count + clkn - pn->first_pkt_time) % 64);
			uint8_t uap = orc_try_clock(clock, p);
			int verdict = -1;
			if (first || uap == pn->clock6_candidates[count])
				verdict = orc_crc_check(clock, p);
			if (orc_piconet_get_flag(pn, ORC_UAP_VALID) && uap != pn->UAP)
				verdict = -1;
			switch (verdict) {
			case -1:
			case 0:
				pn->clock6_candidates[count] = -1;
				break;
			case 1:
			case 2:
				pn->clock6_candidates[count] = uap;
				first_clock = count;
				remaining++;
				break;
			default:
				pn->clk_offset = (count - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
				pn->UAP = uap;
				orc_piconet_set_flag(pn, ORC_CLK6_VALID, 1);
				orc_piconet_set_flag(pn, ORC_UAP_VALID, 1);
				pn->total_packets_observed = 0;
				return 1;
			}
		}
	}
	orc_piconet_set_flag(pn, ORC_GOT_FIRST_PACKET, 1);
	if (remaining == 1) {
		pn->clk_offset = (first_clock - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
		pn->UAP = (uint8_t)pn->clock6_candidates[first_clock];
		orc_piconet_set_flag(pn, ORC_CLK6_VALID, 1);
		orc_piconet_set_flag(pn, ORC_UAP_VALID, 1);
		pn->total_packets_observed = 0;
		return 1;
	}
	if (remaining == 0)
		orc_piconet_reset(pn);
	return 0;
}

/* try_hop, bluetooth_piconet.c:501-543 (prints dropped) */
static void try_hop(orc_packet *p, orc_piconet *pn)
{
	uint8_t filter_uap = pn->UAP;

	orc_decode(p);
	if (orc_piconet_get_flag(pn, ORC_HOP_REVERSAL_INIT)) {
		pn->pattern_indices[pn->packets_observed] = (int)(p->clkn - pn->first_pkt_time);
		pn->pattern_channels[pn->packets_observed] = p->channel;
		pn->packets_observed++;
		pn->total_packets_observed++;
		orc_winnow(pn);
	} else if (orc_piconet_get_flag(pn, ORC_CLK6_VALID)) {
		orc_uap_from_header(p, pn);
	} else if (orc_uap_from_header(p, pn)) {
		if (filter_uap == pn->UAP) {
			pn->hop_reversal_requests++;
			orc_init_hop_reversal(0, pn);
			orc_winnow(pn);
		}
	}
	if (!orc_piconet_get_flag(pn, ORC_UAP_VALID)) {
		orc_piconet_set_flag(pn, ORC_UAP_VALID, 1);
		pn->UAP = filter_uap;
	}
}

/* bluetooth_piconet.c:851-899, non-survey branches */
int orc_process_packet(orc_packet *p, orc_piconet *pn)
{
	if (pn)
		channel_seen(pn, p->channel);
	if (pn && orc_piconet_get_flag(pn, ORC_LAP_VALID) && orc_header_present(p)) {
		if (orc_piconet_get_flag(pn, ORC_FOLLOWING)) {
			p->UAP = pn->UAP;
			orc_packet_set_flag(p, ORC_UAP_VALID, 1);
			orc_packet_set_flag(p, ORC_CLK6_VALID, 1);
			orc_packet_set_flag(p, ORC_CLK27_VALID, 1);
			orc_decode(p);
		} else if (pn->UAP) {
			try_hop(p, pn);
			if (orc_piconet_get_flag(pn, ORC_CLK6_VALID) && orc_piconet_get_flag(pn, ORC_CLK27_VALID)) {
				orc_piconet_set_flag(pn, ORC_FOLLOWING, 1);
				return -1;
			}
		} else {
			orc_uap_from_header(p, pn);
		}
	}
	return 0;
}
