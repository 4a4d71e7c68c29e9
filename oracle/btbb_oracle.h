/*
 * oracle/btbb_oracle.h -- CPU restatement of libbtbb's baseband hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so, and only as the checker.  The product (libbtbb_amd/) never links,
 * imports or falls back to this code.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py)
 * against (1) every golden vector the reference's own tests hold for this path
 * (tests/test_syndromes.c:38-75, tests/test_fec23.c:38-85, tests/test_header.c:22-45),
 * (2) the unmodified reference compiled from /root/reference into
 * oracle/_ref/libbtbb_ref.so, on randomised inputs, and (3) fixtures under
 * tests/golden/ generated from that compiled reference (tests/golden/make_golden.py).
 *
 * Layout of the data this file works on is the reference's: ONE SYMBOL (0/1) PER
 * BYTE in air order (lib/src/btbb.h:82-94).
 */
#ifndef BTBB_ORACLE_H
#define BTBB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SYMBOLS 3125        /* bluetooth_packet.h:27 */
#define ORC_MAX_PAYLOAD_BITS 2744   /* bluetooth_packet.h:30 */
#define ORC_LAP_ANY 0xffffffffu     /* btbb.h:95 */

/* flag ids, btbb.h:27-42 */
enum {
	ORC_WHITENED = 0, ORC_NAP_VALID = 1, ORC_UAP_VALID = 2, ORC_LAP_VALID = 3,
	ORC_CLK6_VALID = 4, ORC_CLK27_VALID = 5, ORC_CRC_CORRECT = 6, ORC_HAS_PAYLOAD = 7,
	ORC_IS_EDR = 8, ORC_HOP_REVERSAL_INIT = 9, ORC_GOT_FIRST_PACKET = 10,
	ORC_IS_AFH = 11, ORC_LOOKS_LIKE_AFH = 12, ORC_IS_ALIASED = 13, ORC_FOLLOWING = 14
};

/* mirrors the fields of struct btbb_packet (bluetooth_packet.h:52-112) that the
 * hot path reads or writes; own layout, not ABI-compatible on purpose */
typedef struct orc_packet {
	uint32_t flags;
	uint8_t channel;
	uint8_t UAP;
	uint16_t NAP;
	uint32_t LAP;
	uint8_t packet_type;
	uint8_t packet_lt_addr;
	uint8_t packet_flags;
	uint8_t packet_hec;
	char packet_header[18];
	int payload_header_length;
	char payload_header[16];
	uint8_t payload_llid;
	uint8_t payload_flow;
	int payload_length;
	char payload[ORC_MAX_PAYLOAD_BITS];
	uint32_t clkn;
	uint8_t ac_errors;
	uint16_t length;
	char symbols[ORC_MAX_SYMBOLS];
} orc_packet;

/* subset of struct btbb_piconet (bluetooth_piconet.h:32-99) used by
 * btbb_uap_from_header / btbb_process_packet */
typedef struct orc_piconet {
	uint32_t flags;
	uint8_t afh_map[10];
	uint8_t used_channels;
	uint32_t LAP;
	uint8_t UAP;
	int packets_observed;
	int total_packets_observed;
	int clock6_candidates[64];
	int pattern_indices[1000];
	uint8_t pattern_channels[1000];
	int clk_offset;
	uint32_t first_pkt_time;
	int hop_reversal_requests;  /* count of times try_hop started the hop reversal */
	/* hop reversal, bluetooth_piconet.h:38-40, 56-85 */
	int aliased;                /* never set by the reference's public API (always 0) */
	int a1, b, c1, d1, e;
	int bank[79];
	char *sequence;             /* 2^27 channels, owned by the pattern cache */
	uint32_t *clock_candidates;
	int num_candidates;
	int winnowed;
} orc_piconet;

typedef struct orc_hit {
	uint64_t offset;    /* symbol index of the first sync-word bit */
	uint32_t lap;
	uint8_t ac_errors;
	uint8_t pad[3];
} orc_hit;

/* ---- tables (derived from the Bluetooth spec polynomials, not copied) ---- */
void orc_tables_init(void);
int orc_table(const char *name, uint64_t *dst, int cap);

/* ---- access code ---- */
uint64_t orc_gen_syncword(int lap);                 /* bluetooth_packet.c:188 */
uint64_t orc_gen_syndrome(uint64_t codeword);       /* :147 */
int orc_init(int max_ac_errors);                    /* :279 */
void orc_reset_syndrome_map(void);                  /* test helper: forget the map */
unsigned orc_syndrome_count(void);
int orc_find_syndrome(uint64_t syndrome, uint64_t *error);
/* first match, semantics of btbb_find_ac (:444) minus the packet allocation.
 * *lap_io: in = LAP or ORC_LAP_ANY, out = LAP found. Returns offset or -1. */
int orc_find_ac(const char *stream, int search_length, uint32_t lap,
		int max_ac_errors, uint32_t *lap_out, uint8_t *ac_errors_out);
/* every match: the loop `off=0; while((r=find_ac(s+off,n-off))>=0){emit;off+=r+1;}` */
size_t orc_find_all(const char *stream, uint64_t search_length, uint32_t lap,
		    int max_ac_errors, orc_hit *out, size_t cap);

/* ---- bit chain ---- */
int orc_unfec13(const char *in, char *out, int length);         /* :552 */
uint16_t orc_fec23(uint16_t data);                               /* :571 */
int orc_unfec23(const char *in, int length, char *out);         /* :585 (1 ok / 0 fail) */
void orc_unwhiten(const char *in, char *out, int clock, int length, int skip, int whitened); /* :653 */
uint16_t orc_crcgen(const char *bits, int length, int uap);     /* :671 */
uint8_t orc_uap_from_hec(uint16_t data, uint8_t hec);           /* :693 */
uint8_t orc_hec_from_uap(uint16_t data, uint8_t uap);           /* inverse, for synthetic TX */

/* ---- packet object ---- */
orc_packet *orc_packet_new(void);
void orc_packet_free(orc_packet *p);
void orc_packet_init_found(orc_packet *p, uint32_t lap, uint8_t ac_errors);      /* init_packet :201 */
void orc_packet_set_data(orc_packet *p, const char *syms, int length, uint8_t channel, uint32_t clkn); /* :467 */
void orc_packet_set_flag(orc_packet *p, int flag, int val);
int orc_packet_get_flag(const orc_packet *p, int flag);
uint32_t orc_packet_header_packed(const orc_packet *p);                           /* :539 */
int orc_payload_packed(const orc_packet *p, char *dst);                           /* :511 */

int orc_header_present(const orc_packet *p);      /* :1371 */
uint8_t orc_try_clock(int clock, orc_packet *p);  /* :1178 */
int orc_crc_check(int clock, orc_packet *p);      /* :708 */
int orc_decode_header(orc_packet *p);             /* :1198 */
int orc_decode_payload(orc_packet *p);            /* :1223 */
int orc_decode(orc_packet *p);                    /* :1300 (silent) */
int orc_fhs(int clock, orc_packet *p);
int orc_DM(int clock, orc_packet *p);
int orc_DH(int clock, orc_packet *p);
int orc_EV3(int clock, orc_packet *p);
int orc_EV4(int clock, orc_packet *p);
int orc_EV5(int clock, orc_packet *p);
int orc_HV(int clock, orc_packet *p);
uint32_t orc_lap_from_fhs(const orc_packet *p);   /* :1411 */
uint8_t orc_uap_from_fhs(const orc_packet *p);
uint16_t orc_nap_from_fhs(const orc_packet *p);
uint32_t orc_clock_from_fhs(const orc_packet *p);

/* ---- piconet (callers of the path) ---- */
orc_piconet *orc_piconet_new(void);
void orc_piconet_free(orc_piconet *pn);
void orc_init_piconet(orc_piconet *pn, uint32_t lap);
void orc_piconet_set_flag(orc_piconet *pn, int flag, int val);
int orc_piconet_get_flag(const orc_piconet *pn, int flag);
int orc_uap_from_header(orc_packet *p, orc_piconet *pn);     /* bluetooth_piconet.c:648 */
int orc_process_packet(orc_packet *p, orc_piconet *pn);      /* :851 (non-survey) */
void orc_piconet_reset(orc_piconet *pn);                     /* :547-572 */

/* ---- hop sequence and CLK1-27 reversal (btbb_oracle_hop.c) ---- */
#define ORC_SEQUENCE_LENGTH 134217728u                       /* bluetooth_piconet.h:102 */
int orc_perm5(int z, int p_high, int p_low);                 /* bluetooth_piconet.c:256 */
void orc_hop_precalc(orc_piconet *pn);                       /* :171 */
void orc_hop_address_precalc(int address, orc_piconet *pn);  /* :197 */
void orc_gen_hops(const orc_piconet *pn, char *sequence);    /* :311 (fills 2^27 bytes) */
void orc_get_hop_pattern(orc_piconet *pn);                   /* :391 (cache keyed by the low 32 key bits) */
void orc_hop_cache_clear(void);                              /* test helper: free every cached sequence */
char orc_single_hop(int clock, const orc_piconet *pn);       /* :415 */
void orc_piconet_set_afh_map(orc_piconet *pn, const uint8_t *afh_map);   /* :122 */
int orc_init_hop_reversal(int aliased, orc_piconet *pn);     /* :475 */
int orc_winnow(orc_piconet *pn);                             /* :614 */

#ifdef __cplusplus
}
#endif
#endif
