/*
 * btbb.h -- drop-in public header of the MI355X-native Bluetooth baseband library.
 *
 * Source compatible with libbtbb's lib/src/btbb.h for the baseband hot path: the same
 * names, argument meaning, return values and ownership rules, so Ubertooth /
 * gr-bluetooth style callers compile and link unchanged (SONAME libbtbb.so.1).
 * The computation behind every function that touches symbols runs in hand-written
 * gfx950 HIP kernels; there is NO CPU fallback.  Without a usable GPU btbb_init()
 * returns a negative value and the search/decode functions report "not found" /
 * failure after printing a diagnostic to stderr.
 *
 * Each declaration cites the reference declaration it replaces
 * (lib/src/btbb.h:<line> in /root/reference).
 *
 * Provided beyond the packet half: the piconet half including hop reversal
 * (btbb_init_hop_reversal / btbb_winnow, GPU candidate lists) and the BR/EDR pcap / pcapng
 * writers.  NOT provided: the Bluetooth LE half (lell_* and the LE capture writers) -- the
 * library exports those symbols only to abort with a diagnostic when one is called; a caller
 * that uses LE keeps the reference's objects for them.
 *
 * Threads: every call that goes to the GPU leases private staging memory and a private
 * stream, so different threads may work on DIFFERENT packets / piconets at the same time
 * (the reference's own rule: its functions only touch the caller's objects).  One object is
 * for one thread at a time; btbb_init() and the survey-mode globals are process wide.
 */
#ifndef INCLUDED_BTBB_H
#define INCLUDED_BTBB_H

#include <stdint.h>

/* packet / piconet flag numbers -- btbb.h:27-42 */
#define BTBB_WHITENED    0
#define BTBB_NAP_VALID   1
#define BTBB_UAP_VALID   2
#define BTBB_LAP_VALID   3
#define BTBB_CLK6_VALID  4
#define BTBB_CLK27_VALID 5
#define BTBB_CRC_CORRECT 6
#define BTBB_HAS_PAYLOAD 7
#define BTBB_IS_EDR      8

#define BTBB_HOP_REVERSAL_INIT 9
#define BTBB_GOT_FIRST_PACKET  10
#define BTBB_IS_AFH            11
#define BTBB_LOOKS_LIKE_AFH    12
#define BTBB_IS_ALIASED        13
#define BTBB_FOLLOWING         14

/* payload modulation -- btbb.h:44-47 */
#define BTBB_MOD_GFSK              0x00
#define BTBB_MOD_PI_OVER_2_DQPSK   0x01
#define BTBB_MOD_8DPSK             0x02

/* transport types -- btbb.h:49-54 */
#define BTBB_TRANSPORT_ANY     0x00
#define BTBB_TRANSPORT_SCO     0x01
#define BTBB_TRANSPORT_ESCO    0x02
#define BTBB_TRANSPORT_ACL     0x03
#define BTBB_TRANSPORT_CSB     0x04

#ifdef __cplusplus
extern "C"
{
#endif
/* the library is built with -fvisibility=hidden: exactly the names declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct btbb_packet btbb_packet;      /* btbb.h:63 */
typedef struct btbb_piconet btbb_piconet;    /* btbb.h:161 */

/* btbb.h:73 -- build the syndrome tables (on the GPU) for up to max_ac_errors (0..5)
 * corrected bit errors.  0 on success, negative on error.  As in the reference the
 * first non-zero value sticks. */
int btbb_init(int max_ac_errors);

const char* btbb_get_release(void);          /* btbb.h:75 */
const char* btbb_get_version(void);          /* btbb.h:76 */

btbb_packet *btbb_packet_new(void);          /* btbb.h:78 */
void btbb_packet_ref(btbb_packet *pkt);      /* btbb.h:79 */
void btbb_packet_unref(btbb_packet *pkt);    /* btbb.h:80 */

/* btbb.h:90-94 -- search `stream` (one 0/1 symbol per byte, air order, at least
 * search_length + 72 symbols long) for an access code with the given LAP or LAP_ANY,
 * tolerating max_ac_errors bit errors.  Returns the offset of the first match or a
 * negative number; on a match *pkt is allocated if NULL and LAP / ac_errors / flags
 * are (re)initialised. */
int btbb_find_ac(char *stream,
	       int search_length,
	       uint32_t lap,
	       int max_ac_errors,
	       btbb_packet **pkt);
#define LAP_ANY 0xffffffffUL                 /* btbb.h:95 */
#define UAP_ANY 0xff                         /* btbb.h:96 */

void btbb_packet_set_flag(btbb_packet *pkt, int flag, int val);      /* btbb.h:98 */
int btbb_packet_get_flag(const btbb_packet *pkt, int flag);          /* btbb.h:99 */

uint32_t btbb_packet_get_lap(const btbb_packet *pkt);                /* btbb.h:101 */
void btbb_packet_set_uap(btbb_packet *pkt, uint8_t uap);             /* btbb.h:102 */
uint8_t btbb_packet_get_uap(const btbb_packet *pkt);                 /* btbb.h:103 */
uint16_t btbb_packet_get_nap(const btbb_packet *pkt);                /* btbb.h:104 */

void btbb_packet_set_modulation(btbb_packet *pkt, uint8_t modulation);   /* btbb.h:106 */
void btbb_packet_set_transport(btbb_packet *pkt, uint8_t transport);     /* btbb.h:107 */
uint8_t btbb_packet_get_modulation(const btbb_packet *pkt);              /* btbb.h:108 */
uint8_t btbb_packet_get_transport(const btbb_packet *pkt);               /* btbb.h:109 */

uint8_t btbb_packet_get_channel(const btbb_packet *pkt);             /* btbb.h:111 */
uint8_t btbb_packet_get_ac_errors(const btbb_packet *pkt);           /* btbb.h:112 */
uint32_t btbb_packet_get_clkn(const btbb_packet *pkt);               /* btbb.h:113 */
uint32_t btbb_packet_get_header_packed(const btbb_packet* pkt);      /* btbb.h:114 */

/* btbb.h:116-120 -- copy up to 3125 symbols into the packet; clkn is CLK27-0 */
void btbb_packet_set_data(btbb_packet *pkt,
			  char *syms,
			  int length,
			  uint8_t channel,
			  uint32_t clkn);

const char *btbb_get_symbols(const btbb_packet* pkt);                /* btbb.h:123 */
int btbb_packet_get_payload_length(const btbb_packet* pkt);          /* btbb.h:125 */
const char *btbb_get_payload(const btbb_packet* pkt);                /* btbb.h:128 */
int btbb_get_payload_packed(const btbb_packet* pkt, char *dst);      /* btbb.h:131 */

uint8_t btbb_packet_get_type(const btbb_packet* pkt);                /* btbb.h:133 */
uint8_t btbb_packet_get_lt_addr(const btbb_packet* pkt);             /* btbb.h:134 */
uint8_t btbb_packet_get_header_flags(const btbb_packet* pkt);        /* btbb.h:135 */
uint8_t btbb_packet_get_hec(const btbb_packet *pkt);                 /* btbb.h:136 */

uint64_t btbb_gen_syncword(const int LAP);                           /* btbb.h:139 */

int btbb_decode_header(btbb_packet* pkt);                            /* btbb.h:142 */
int btbb_decode_payload(btbb_packet* pkt);                           /* btbb.h:145 */
void btbb_print_packet(const btbb_packet* pkt);                      /* btbb.h:148 */
int btbb_header_present(const btbb_packet* pkt);                     /* btbb.h:151 */

/* Also exported by the reference library (lib/src/bluetooth_packet.h:114-144), for callers
 * that link against them: one candidate clock / its CRC verdict, FHS fields, tun format. */
uint8_t try_clock(int clock, btbb_packet *pkt);                      /* bluetooth_packet.h:132 */
int crc_check(int clock, btbb_packet *pkt);                          /* bluetooth_packet.h:124 */
uint32_t lap_from_fhs(btbb_packet *pkt);                             /* bluetooth_packet.h:135 */
uint8_t uap_from_fhs(btbb_packet *pkt);                              /* bluetooth_packet.h:138 */
uint16_t nap_from_fhs(btbb_packet *pkt);                             /* bluetooth_packet.h:141 */
uint32_t clock_from_fhs(btbb_packet *pkt);                           /* bluetooth_packet.h:144 */
char *tun_format(btbb_packet *pkt);                                  /* bluetooth_packet.h:127 */

btbb_piconet *btbb_piconet_new(void);                                /* btbb.h:163 */
void btbb_piconet_ref(btbb_piconet *pn);                             /* btbb.h:164 */
void btbb_piconet_unref(btbb_piconet *pn);                           /* btbb.h:165 */
void btbb_init_piconet(btbb_piconet *pn, uint32_t lap);              /* btbb.h:168 */

void btbb_piconet_set_uap(btbb_piconet *pn, uint8_t uap);            /* btbb.h:170 */
uint8_t btbb_piconet_get_uap(const btbb_piconet *pn);                /* btbb.h:171 */
uint32_t btbb_piconet_get_lap(const btbb_piconet *pn);               /* btbb.h:172 */
uint16_t btbb_piconet_get_nap(const btbb_piconet *pn);               /* btbb.h:173 */
uint64_t btbb_piconet_get_bdaddr(const btbb_piconet *pn);            /* btbb.h:174 */
int btbb_piconet_get_clk_offset(const btbb_piconet *pn);             /* btbb.h:176 */
void btbb_piconet_set_clk_offset(btbb_piconet *pn, int clk_offset);  /* btbb.h:177 */
void btbb_piconet_set_flag(btbb_piconet *pn, int flag, int val);     /* btbb.h:179 */
int btbb_piconet_get_flag(const btbb_piconet *pn, int flag);         /* btbb.h:180 */
uint8_t btbb_piconet_set_channel_seen(btbb_piconet *pn, uint8_t channel);    /* btbb.h:182 */
uint8_t btbb_piconet_clear_channel_seen(btbb_piconet *pn, uint8_t channel);  /* btbb.h:183 */
uint8_t btbb_piconet_get_channel_seen(btbb_piconet *pn, uint8_t channel);    /* btbb.h:184 */
void btbb_piconet_set_afh_map(btbb_piconet *pn, uint8_t *afh_map);   /* btbb.h:185 */
uint8_t *btbb_piconet_get_afh_map(btbb_piconet *pn);                 /* btbb.h:186 */

/* btbb.h:189 -- extract LAP/UAP/CLK information from a received packet */
int btbb_process_packet(btbb_packet *pkt, btbb_piconet *pn);
/* btbb.h:192 -- use packet headers to determine UAP (64 CLK1-6 candidates on the GPU) */
int btbb_uap_from_header(btbb_packet *pkt, btbb_piconet *pn);
void btbb_print_afh_map(btbb_piconet *pn);                           /* btbb.h:195 */
/* btbb.h:198 -- decode a whole packet */
int btbb_decode(btbb_packet* pkt);

/* btbb.h:203 -- start the CLK1-27 reversal: the candidate clocks (all CLK1-27 values whose hop
 * is the channel of the first observed packet) are built on the GPU and stay in HBM; returns
 * their number.  `aliased` only sets BTBB_IS_ALIASED, exactly as in bluetooth_piconet.c:475-498 */
int btbb_init_hop_reversal(int aliased, btbb_piconet *pn);
/* btbb.h:206 -- narrow the candidates by all hops observed since the last call */
int btbb_winnow(btbb_piconet *pn);

int btbb_init_survey(void);                                          /* btbb.h:208 */
btbb_piconet *btbb_next_survey_result(void);                         /* btbb.h:210 */

/* ---- capture files (BR/EDR), btbb.h:212-229, 262-271 ---------------------------------------
 * Host file I/O over the fields the GPU decode left in the packet.  Return 0 or a negated
 * result code of lib/src/pcapng.h:163-172 / lib/src/pcap.c:32-37. */
typedef struct btbb_pcapng_handle btbb_pcapng_handle;
/* btbb.h:214 -- new PCAPNG file (fails if it exists), one LINKTYPE_BLUETOOTH_BREDR_BB interface */
int btbb_pcapng_create_file(const char *filename, const char *interface_desc, btbb_pcapng_handle **ph);
/* btbb.h:216 -- one enhanced packet block; ns = capture time in nanoseconds */
int btbb_pcapng_append_packet(btbb_pcapng_handle *h, const uint64_t ns,
			      const int8_t sigdbm, const int8_t noisedbm,
			      const uint32_t reflap, const uint8_t refuap,
			      const btbb_packet *pkt);
/* btbb.h:221 -- interface option 0xd340 */
int btbb_pcapng_record_bdaddr(btbb_pcapng_handle *h, const uint64_t bdaddr,
			      const uint8_t uapmask, const uint8_t napvalid);
/* btbb.h:224 -- interface option 0xd341 */
int btbb_pcapng_record_btclock(btbb_pcapng_handle *h, const uint64_t bdaddr,
			       const uint64_t ns, const uint32_t clk, const uint32_t clkmask);
int btbb_pcapng_close(btbb_pcapng_handle *h);                        /* btbb.h:226 */

typedef struct btbb_pcap_handle btbb_pcap_handle;
/* btbb.h:264 -- classic PCAP, nanosecond magic, LINKTYPE_BLUETOOTH_BREDR_BB */
int btbb_pcap_create_file(const char *filename, btbb_pcap_handle **ph);
int btbb_pcap_append_packet(btbb_pcap_handle *h, const uint64_t ns,      /* btbb.h:266 */
			    const int8_t sigdbm, const int8_t noisedbm,
			    const uint32_t reflap, const uint8_t refuap,
			    const btbb_packet *pkt);
int btbb_pcap_close(btbb_pcap_handle *h);                            /* btbb.h:270 */

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
} // __cplusplus defined.
#endif

#endif /* INCLUDED_BTBB_H */
