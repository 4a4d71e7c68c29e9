/*
 * oracle/ref_internals.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Build-time shim that pulls the *unmodified* reference translation unit
 * lib/src/bluetooth_packet.c in by #include (from where it lies under
 * $(REF), normally /root/reference) so that its `static` helpers can be
 * reached from the test-suite through exported wrappers.  No reference source
 * text lives in this repository: REF_PACKET_C is a path given on the compiler
 * command line by oracle/Makefile, and the output goes to oracle/_ref/ only.
 *
 * Wrapped statics (reference file:line):
 *   gen_syndrome   bluetooth_packet.c:147     unfec13   :552
 *   fec23          :571                       unfec23   :585
 *   unwhiten       :653                       crcgen    :671
 *   uap_from_hec   :693                       air_to_host* :211-242
 *   tables         :49-59, 73-119; sw_check_tables.h
 */
#include <string.h>
#include <stddef.h>
#include REF_PACKET_C

uint64_t refint_gen_syndrome(uint64_t cw) { return gen_syndrome(cw); }

int refint_unfec13(char *in, char *out, int length) { return unfec13(in, out, length); }

uint16_t refint_fec23(uint16_t data) { return fec23(data); }

/* returns 1 and fills out[padded length] on success, 0 when the reference returns NULL */
int refint_unfec23(char *in, int length, char *out)
{
	int padded = length;
	char *o = unfec23(in, length);
	if (!o)
		return 0;
	if (padded % 10)
		padded += 10 - (padded % 10);
	memcpy(out, o, padded);
	free(o);
	return 1;
}

void refint_unwhiten(char *in, char *out, int clock, int length, int skip, int whitened)
{
	btbb_packet p;
	memset(&p, 0, sizeof(p));
	btbb_packet_set_flag(&p, BTBB_WHITENED, whitened);
	unwhiten(in, out, clock, length, skip, &p);
}

uint16_t refint_crcgen(char *bits, int length, int uap) { return crcgen(bits, length, uap); }

uint8_t refint_uap_from_hec(uint16_t data, uint8_t hec) { return uap_from_hec(data, hec); }

uint8_t refint_reverse(uint8_t b) { return reverse((char)b); }

/* table access: returns element count, copies min(count, cap) 64-bit values */
int refint_table(const char *name, uint64_t *dst, int cap)
{
	int i, n = 0;
#define COPY(arr) do { n = (int)(sizeof(arr) / sizeof((arr)[0])); \
	for (i = 0; i < n && i < cap; i++) dst[i] = (uint64_t)(arr)[i]; } while (0)
	if (!strcmp(name, "INDICES")) COPY(INDICES);
	else if (!strcmp(name, "WHITENING_DATA")) COPY(WHITENING_DATA);
	else if (!strcmp(name, "BARKER_DISTANCE")) COPY(BARKER_DISTANCE);
	else if (!strcmp(name, "barker_correct")) COPY(barker_correct);
	else if (!strcmp(name, "sw_matrix")) COPY(sw_matrix);
	else if (!strcmp(name, "fec23_gen_matrix")) COPY(fec23_gen_matrix);
	else if (!strcmp(name, "sw_check_table4")) COPY(sw_check_table4);
	else if (!strcmp(name, "sw_check_table5")) COPY(sw_check_table5);
	else if (!strcmp(name, "sw_check_table6")) COPY(sw_check_table6);
	else if (!strcmp(name, "sw_check_table7")) COPY(sw_check_table7);
	else if (!strcmp(name, "pn")) { dst[0] = pn; n = 1; }
	else if (!strcmp(name, "DEFAULT_AC")) { dst[0] = DEFAULT_AC; n = 1; }
	else if (!strcmp(name, "DEFAULT_CODEWORD")) { dst[0] = DEFAULT_CODEWORD; n = 1; }
	else return -1;
#undef COPY
	return n;
}

/* number of entries currently in the reference's global syndrome map */
unsigned refint_syndrome_count(void) { return syndrome_map ? HASH_COUNT(syndrome_map) : 0; }

/* look an arbitrary syndrome up: 1 + *error on hit, 0 on miss */
int refint_find_syndrome(uint64_t syndrome, uint64_t *error)
{
	syndrome_struct *s = find_syndrome(syndrome);
	if (!s)
		return 0;
	*error = s->error;
	return 1;
}

/* The caller loop of an all-matches scan, natively: first-match btbb_find_ac resumed one
 * symbol past every hit.  hits[] receives offset / LAP / ac_errors triples.
 * Used by the test-suite and by bench.py's cpu_baseline leg ("kind": "reference"). */
size_t refint_find_all(char *stream, uint64_t search_length, uint32_t lap, int max_ac_errors,
		       uint64_t *hit_offset, uint32_t *hit_lap, uint8_t *hit_err, size_t cap)
{
	size_t n = 0;
	uint64_t off = 0;
	btbb_packet *pkt = NULL;
	while (off < search_length) {
		uint64_t left = search_length - off;
		int chunk = left > 0x40000000ULL ? 0x40000000 : (int)left;
		int r = btbb_find_ac(stream + off, chunk, lap, max_ac_errors, &pkt);
		if (r < 0) {
			off += (uint64_t)chunk;
			continue;
		}
		if (n < cap) {
			hit_offset[n] = off + (uint64_t)r;
			hit_lap[n] = btbb_packet_get_lap(pkt);
			hit_err[n] = btbb_packet_get_ac_errors(pkt);
		}
		n++;
		off += (uint64_t)r + 1;
	}
	if (pkt)
		btbb_packet_unref(pkt);
	return n;
}

/* packet object layout (bluetooth_packet.h:52-112) so tests can peek at fields */
size_t refint_packet_sizeof(void) { return sizeof(btbb_packet); }
size_t refint_packet_offsetof(const char *field)
{
#define OFF(f) if (!strcmp(field, #f)) return offsetof(btbb_packet, f)
	OFF(refcount); OFF(flags); OFF(channel); OFF(UAP); OFF(NAP); OFF(LAP);
	OFF(modulation); OFF(transport); OFF(packet_type); OFF(packet_lt_addr);
	OFF(packet_flags); OFF(packet_hec); OFF(packet_header);
	OFF(payload_header_length); OFF(payload_header); OFF(payload_llid);
	OFF(payload_flow); OFF(payload_length); OFF(payload); OFF(crc);
	OFF(clkn); OFF(ac_errors); OFF(length); OFF(symbols);
#undef OFF
	return (size_t)-1;
}
