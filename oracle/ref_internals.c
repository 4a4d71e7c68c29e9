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
#define _GNU_SOURCE
#include <string.h>
#include <stddef.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
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

/* ---- the all-matches scan on many host threads, timed natively (bench.py cpu_baseline) ----------------
 * n_threads workers over disjoint slices [bounds[i], bounds[i+1]) of one symbol-per-byte stream, each the
 * caller loop above.  They start together behind a barrier; every worker takes CLOCK_MONOTONIC right before
 * and right after its native loop -- nothing else is inside the timed region: no interpreter, no allocation
 * (hit arrays are preallocated by the caller, cap_per_thread records per worker), no unpacking.  cpus[i] >= 0
 * pins worker i to that logical CPU (one worker per physical core runs).  Returns 0, or -1 if a thread could
 * not be created.  wall_seconds = last end - first start. */

struct refint_mt_job {
	char *stream;
	uint64_t lo, hi;
	uint32_t lap;
	int max_ac_errors, cpu;
	uint64_t *hit_offset;
	uint32_t *hit_lap;
	uint8_t *hit_err;
	size_t cap, found;
	double t0, t1;
	pthread_barrier_t *start;
};

static double refint_now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *refint_mt_worker(void *arg)
{
	struct refint_mt_job *j = (struct refint_mt_job *)arg;
	size_t k, n;
	if (j->cpu >= 0) {
		cpu_set_t set;
		CPU_ZERO(&set);
		CPU_SET(j->cpu, &set);
		pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
	}
	pthread_barrier_wait(j->start);
	j->t0 = refint_now();
	n = refint_find_all(j->stream + j->lo, j->hi - j->lo, j->lap, j->max_ac_errors,
			    j->hit_offset, j->hit_lap, j->hit_err, j->cap);
	j->t1 = refint_now();
	j->found = n;
	for (k = 0; k < n && k < j->cap; k++)          /* slice-relative -> stream offsets (outside the timed region) */
		j->hit_offset[k] += j->lo;
	return NULL;
}

int refint_find_all_mt(char *stream, const uint64_t *bounds, int n_threads, uint32_t lap, int max_ac_errors,
		       const int *cpus, uint64_t *hit_offset, uint32_t *hit_lap, uint8_t *hit_err,
		       size_t cap_per_thread, uint64_t *found_per_thread, double *seconds_per_thread,
		       double *wall_seconds)
{
	struct refint_mt_job *jobs;
	pthread_t *tids;
	pthread_barrier_t start;
	int i, made = 0, rc = 0;
	double first = 0, last = 0;
	if (n_threads <= 0)
		return -1;
	jobs = (struct refint_mt_job *)calloc((size_t)n_threads, sizeof(*jobs));
	tids = (pthread_t *)calloc((size_t)n_threads, sizeof(*tids));
	if (!jobs || !tids)
		return -1;
	pthread_barrier_init(&start, NULL, (unsigned)n_threads);
	for (i = 0; i < n_threads; i++) {
		jobs[i].stream = stream;
		jobs[i].lo = bounds[i];
		jobs[i].hi = bounds[i + 1];
		jobs[i].lap = lap;
		jobs[i].max_ac_errors = max_ac_errors;
		jobs[i].cpu = cpus ? cpus[i] : -1;
		jobs[i].hit_offset = hit_offset + (size_t)i * cap_per_thread;
		jobs[i].hit_lap = hit_lap + (size_t)i * cap_per_thread;
		jobs[i].hit_err = hit_err + (size_t)i * cap_per_thread;
		jobs[i].cap = cap_per_thread;
		jobs[i].start = &start;
		if (pthread_create(&tids[i], NULL, refint_mt_worker, &jobs[i]) != 0)
			break;
		made++;
	}
	if (made < n_threads) {
		/* release the ones that wait: re-arming a barrier under waiters is not allowed, so give up loudly */
		fprintf(stderr, "refint_find_all_mt: only %d of %d threads could be created\n", made, n_threads);
		abort();
	}
	for (i = 0; i < n_threads; i++)
		pthread_join(tids[i], NULL);
	for (i = 0; i < n_threads; i++) {
		found_per_thread[i] = jobs[i].found;
		seconds_per_thread[i] = jobs[i].t1 - jobs[i].t0;
		if (i == 0 || jobs[i].t0 < first) first = jobs[i].t0;
		if (i == 0 || jobs[i].t1 > last) last = jobs[i].t1;
	}
	*wall_seconds = last - first;
	pthread_barrier_destroy(&start);
	free(jobs);
	free(tids);
	return rc;
}

/* packed LSB-first words -> one symbol per byte (the reference's input layout), on n_threads threads */
struct refint_unpack_job { const uint64_t *words; uint8_t *out; uint64_t w0, w1; };
static void *refint_unpack_worker(void *arg)
{
	struct refint_unpack_job *j = (struct refint_unpack_job *)arg;
	uint64_t w;
	int b;
	for (w = j->w0; w < j->w1; w++) {
		uint64_t v = j->words[w];
		uint8_t *o = j->out + w * 64;
		for (b = 0; b < 64; b++)
			o[b] = (uint8_t)((v >> b) & 1);
	}
	return NULL;
}
int refint_unpack_mt(const uint64_t *words, uint64_t n_words, uint8_t *out, int n_threads)
{
	struct refint_unpack_job *jobs;
	pthread_t *tids;
	int i;
	if (n_threads <= 0)
		n_threads = 1;
	jobs = (struct refint_unpack_job *)calloc((size_t)n_threads, sizeof(*jobs));
	tids = (pthread_t *)calloc((size_t)n_threads, sizeof(*tids));
	for (i = 0; i < n_threads; i++) {
		jobs[i].words = words;
		jobs[i].out = out;
		jobs[i].w0 = n_words * (uint64_t)i / (uint64_t)n_threads;
		jobs[i].w1 = n_words * (uint64_t)(i + 1) / (uint64_t)n_threads;
		if (pthread_create(&tids[i], NULL, refint_unpack_worker, &jobs[i]) != 0) {
			refint_unpack_worker(&jobs[i]);
			tids[i] = 0;
		}
	}
	for (i = 0; i < n_threads; i++)
		if (tids[i])
			pthread_join(tids[i], NULL);
	free(jobs);
	free(tids);
	return 0;
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

/* ---- bounded CPU legs of bench.py's secondary measurements ("kind": "reference") -------------
 * Natively looped so that Python is not in the timed path; still the unmodified reference code. */

/* BASELINE config 3 on one channel: all known-LAP matches (first-match btbb_find_ac resumed past
 * every hit) and, per match, what a caller with known UAP / CLK1-6 does with the packet:
 * btbb_packet_set_data, btbb_decode_header, btbb_decode_payload (bluetooth_packet.c:467-480,
 * 1198-1297; btbb_decode itself also prints every packet, :1311-1314).  CLK1-6 of a packet found
 * at offset o is (o / clk_div) & 63 -- the synthetic capture's rule.  Returns the number of
 * matches; *crc_ok counts payload results of 10 / 1000. */
size_t refint_known_lap_chain(char *stream, uint64_t n_symbols, uint32_t lap, int max_ac_errors, uint8_t uap,
			      uint32_t clk_div, uint64_t *crc_ok)
{
	size_t n = 0;
	uint64_t off = 0, good = 0;
	const uint64_t search_length = n_symbols - 63;
	btbb_packet *found = NULL;
	btbb_packet *pkt = btbb_packet_new();
	while (off < search_length) {
		uint64_t left = search_length - off, at, avail;
		int chunk = left > 0x40000000ULL ? 0x40000000 : (int)left;
		int r = btbb_find_ac(stream + off, chunk, lap, max_ac_errors, &found);
		if (r < 0) {
			off += (uint64_t)chunk;
			continue;
		}
		at = off + (uint64_t)r;
		avail = n_symbols - at;
		pkt->LAP = lap;
		pkt->flags = 0;
		btbb_packet_set_flag(pkt, BTBB_WHITENED, 1);
		btbb_packet_set_data(pkt, stream + at, avail > MAX_SYMBOLS ? MAX_SYMBOLS : (int)avail, 0,
				     (uint32_t)(((at / clk_div) & 63) << 1));
		btbb_packet_set_uap(pkt, uap);
		btbb_packet_set_flag(pkt, BTBB_CLK6_VALID, 1);
		if (btbb_header_present(pkt) && btbb_decode_header(pkt)) {
			int rv = btbb_decode_payload(pkt);
			good += rv == 10 || rv == 1000;
		}
		n++;
		off = at + 1;
	}
	if (found)
		btbb_packet_unref(found);
	btbb_packet_unref(pkt);
	*crc_ok = good;
	return n;
}

/* The same chain with what it decoded kept per match, for bench.py's in-run parity of the config-3 lines: one 32-byte
 * record per access code (at most cap are written; the return value counts all).  payload_hash covers the
 * payload_length * 8 payload bits the decoder left (LSB-first 64-bit words w_k, sum of w_k * (2 k + 1) mod 2^64) when
 * the payload decoder returned 2 / 10 / 1000, else 0. */
struct refint_chain_record {
	uint64_t offset;
	uint64_t payload_hash;
	int32_t payload_rv, payload_length;
	uint8_t ac_errors, header_rv, type, lt_addr, hdr_flags, hec, header_present, pad;
};
size_t refint_known_lap_chain_records(char *stream, uint64_t n_symbols, uint32_t lap, int max_ac_errors, uint8_t uap,
				      uint32_t clk_div, struct refint_chain_record *rec, size_t cap)
{
	size_t n = 0;
	uint64_t off = 0;
	const uint64_t search_length = n_symbols - 63;
	btbb_packet *found = NULL;
	btbb_packet *pkt = btbb_packet_new();
	while (off < search_length) {
		uint64_t left = search_length - off, at, avail;
		int chunk = left > 0x40000000ULL ? 0x40000000 : (int)left;
		int r = btbb_find_ac(stream + off, chunk, lap, max_ac_errors, &found);
		struct refint_chain_record q;
		if (r < 0) {
			off += (uint64_t)chunk;
			continue;
		}
		at = off + (uint64_t)r;
		avail = n_symbols - at;
		memset(&q, 0, sizeof(q));
		q.offset = at;
		q.ac_errors = btbb_packet_get_ac_errors(found);
		pkt->LAP = lap;
		pkt->flags = 0;
		btbb_packet_set_flag(pkt, BTBB_WHITENED, 1);
		btbb_packet_set_data(pkt, stream + at, avail > MAX_SYMBOLS ? MAX_SYMBOLS : (int)avail, 0,
				     (uint32_t)(((at / clk_div) & 63) << 1));
		btbb_packet_set_uap(pkt, uap);
		btbb_packet_set_flag(pkt, BTBB_CLK6_VALID, 1);
		q.header_present = (uint8_t)btbb_header_present(pkt);
		if (q.header_present && btbb_decode_header(pkt)) {
			int rv = btbb_decode_payload(pkt);
			q.header_rv = 1;
			q.payload_rv = rv;
			q.payload_length = pkt->payload_length;
			q.type = pkt->packet_type;
			q.lt_addr = pkt->packet_lt_addr;
			q.hdr_flags = pkt->packet_flags;
			q.hec = pkt->packet_hec;
			if (rv == 2 || rv == 10 || rv == 1000) {
				int nbits = pkt->payload_length * 8, k, b;
				uint64_t h = 0;
				for (k = 0; 64 * k < nbits; k++) {
					uint64_t w = 0;
					for (b = 0; b < 64 && 64 * k + b < nbits; b++)
						w |= (uint64_t)(pkt->payload[64 * k + b] & 1) << b;
					h += w * (uint64_t)(2 * k + 1);
				}
				q.payload_hash = h;
			}
		}
		if (n < cap)
			rec[n] = q;
		n++;
		off = at + 1;
	}
	if (found)
		btbb_packet_unref(found);
	btbb_packet_unref(pkt);
	return n;
}

/* BASELINE config 5: the 64-candidate loop of btbb_uap_from_header (bluetooth_piconet.c:675-690) on
 * n_packets packets of `stride` symbols each: try_clock + crc_check for every CLK1-6 value.  Returns a
 * checksum of the results so that nothing is optimised away; table[p * 64 + c] = uap | rv << 8. */
uint64_t refint_clk6_trials(char *symbols, uint32_t n_packets, uint32_t stride, uint32_t length, uint32_t lap,
			    uint32_t *table)
{
	uint64_t sum = 0;
	uint32_t p;
	int c;
	btbb_packet *pkt = btbb_packet_new();
	for (p = 0; p < n_packets; p++) {
		memset(pkt, 0, sizeof(*pkt));
		pkt->refcount = 1;
		pkt->LAP = lap;
		btbb_packet_set_flag(pkt, BTBB_WHITENED, 1);
		btbb_packet_set_data(pkt, symbols + (size_t)p * stride, (int)length, 0, 0);
		for (c = 0; c < 64; c++) {
			uint8_t u = try_clock(c, pkt);
			int rv = crc_check(c, pkt);
			if (table)
				table[(size_t)p * 64 + c] = (uint32_t)u | ((uint32_t)rv << 8);
			sum = sum * 31 + u + ((uint64_t)rv << 8);
		}
	}
	btbb_packet_unref(pkt);
	return sum;
}
