/*
 * btbbx.h -- batch (GPU-resident) entry points of the MI355X baseband scanner.
 *
 * Additive to the drop-in API in btbb.h.  Plain C ABI: pointers and sizes only.
 * Everything here runs on the GPU through hand-written gfx950 HIP kernels; there
 * is no CPU fallback -- every entry point fails with BTBBX_E_NODEVICE when no HIP
 * device is usable.
 *
 * Data layout ("packed stream"): LSB-first 64-bit words, stream bit i is bit
 * (i % 64) of word (i / 64), so that the 64-symbol window starting at bit c has
 * the value the reference obtains from air_to_host64(&stream[c], 64)
 * (lib/src/bluetooth_packet.c:235-242).
 *
 * Reference interfaces replaced (relative to /root/reference):
 *   btbbx_scan_*      <- the caller loop around btbb_find_ac, lib/src/btbb.h:82-94,
 *                        lib/src/bluetooth_packet.c:368-464 (all matches, not first)
 *   btbbx_pack_*      <- the one-symbol-per-byte convention of btbb.h:90,116
 *   btbbx_trials_*    <- try_clock + crc_check over the 64 CLK1-6 candidates,
 *                        lib/src/bluetooth_piconet.c:675-690,
 *                        lib/src/bluetooth_packet.c:708-769, 1178-1195
 *   btbbx_decode_*    <- btbb_header_present / btbb_decode_header / btbb_decode_payload,
 *                        lib/src/bluetooth_packet.c:1198-1297, 1371-1408
 *   btbbx_hop_*       <- gen_hops / hop / init_candidates / channel_winnow / btbb_winnow,
 *                        lib/src/bluetooth_piconet.c:311-362, 443-446, 455-498, 575-645
 */
#ifndef INCLUDED_BTBBX_H
#define INCLUDED_BTBBX_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the names declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define BTBBX_OK            0
#define BTBBX_E_NODEVICE   -2   /* no usable HIP device / runtime error (see btbbx_last_error) */
#define BTBBX_E_ARG        -3   /* bad argument */
#define BTBBX_E_NOTINIT    -4   /* btbbx_init / btbb_init not called */
#define BTBBX_E_NOMEM      -5

#define BTBBX_LAP_ANY 0xffffffffu
#define BTBBX_MAX_SYMBOLS 3125            /* bluetooth_packet.h:27 */
#define BTBBX_PKT_WORDS 50                /* 3200 bits >= 3125 symbols, one packed packet */

/* one detected access code */
typedef struct btbbx_hit {
	uint64_t offset;      /* symbol index of the first sync-word bit inside its stream */
	uint32_t lap;         /* LAP recovered (LAP_ANY) or searched for */
	uint8_t  ac_errors;   /* as btbb_packet_get_ac_errors() would report */
	uint8_t  reserved;
	uint16_t stream;      /* stream (channel) index of the launch */
} btbbx_hit;

/* result of one (packet, clock) trial: what try_clock + crc_check leave behind */
typedef struct btbbx_trial {
	uint8_t uap;          /* try_clock() return value / pkt->UAP */
	uint8_t type;         /* pkt->packet_type after try_clock */
	int16_t rv;           /* crc_check() return value: 0, 1, 2, 10 or 1000 */
} btbbx_trial;

/* one packet to decode: where it is and what is known about it */
typedef struct btbbx_pkt_in {
	uint32_t length;      /* symbols captured, <= 3125 (btbb_packet_set_data clamps) */
	uint32_t clkn;        /* CLK1-27 as stored by set_data (caller's clkn >> 1) */
	uint32_t flags;       /* packet flags (BTBB_WHITENED etc., btbb.h:27-35) */
	uint8_t  uap;         /* pkt->UAP on entry */
	uint8_t  type;        /* pkt->packet_type on entry (state left by earlier calls) */
	uint8_t  llid;        /* pkt->payload_llid on entry */
	uint8_t  flow;        /* pkt->payload_flow on entry */
} btbbx_pkt_in;

/* everything btbb_decode_header + btbb_decode_payload write into a packet */
typedef struct btbbx_pkt_out {
	int32_t  header_rv;          /* btbb_decode_header() */
	int32_t  payload_rv;         /* btbb_decode_payload() (0 if header failed) */
	int32_t  payload_length;
	int32_t  payload_header_length;
	uint32_t flags;              /* packet flags afterwards */
	uint32_t header_packed;      /* 18 unwhitened header bits */
	uint8_t  header_present;     /* btbb_header_present() */
	uint8_t  type, lt_addr, hdr_flags, hec;
	uint8_t  llid, flow, uap;
	uint64_t payload_header;     /* 16 payload-header bits, LSB first */
	uint64_t payload[43];        /* 2744 payload bits, LSB first */
} btbbx_pkt_out;

/* ---- context ---------------------------------------------------------------- */
/* Build the device tables for searches correcting up to max_ac_errors (0..5) bit
 * errors on the CURRENT HIP device.  Like btbb_init() the first non-zero value wins
 * (bluetooth_packet.c:288-289).  Returns 0 or a negative BTBBX_E_*. */
int btbbx_init(int max_ac_errors);
/* The same for every device of `devices[0..n_devices)` (HIP ordinals); the calling thread's current
 * device is restored.  Needed before btbbx_scan_host_multi; one-process-per-GPU callers just select
 * their device and call btbbx_init / btbb_init.  The tables are the same on every device: the first
 * non-zero max_ac_errors of the PROCESS wins, as with the reference's one global map. */
int btbbx_init_devices(const int *devices, int n_devices, int max_ac_errors);
void btbbx_shutdown(void);
const char *btbbx_last_error(void);
int btbbx_device_count(void);
int btbbx_table_errors(void);          /* the max_ac_errors the tables were built with */
/* Host-only diagnostic, no device needed: the candidate set the LAP_ANY scan probes for every offset that passes
 * the barker filter -- all values of nineteen sliding parity checks of the (64,30) code (gen_syndrome's generator,
 * bluetooth_packet.c:147-159) that a window within max_ac_errors of a sync word can take, as a 2^19-bit set in
 * 16384 words.  *taps (may be NULL) receives the check's tap pattern: check b of a window w is the parity of
 * w & (taps << b), b = 0..18.  Returns the number of members or a negative BTBBX_E_*. */
int btbbx_slide_set(int max_ac_errors, uint32_t *bitmap_words, uint64_t *taps);
/* The same for tables built for THREE or FOUR errors (max_ac_errors = 3 or 4), whose scan tests a survivor in two
 * levels, as the kernel reads them: first_words = a 2^20-bit set (32768 words) over twenty positions of the check
 * taps[0], indexed by the COMPLEMENT of the checks' value (index i, bit i & 31 of word i >> 5); second_words = a
 * 2^24-bit set (524288 words) over twenty-four positions of the check taps[1], index i at bit 31 - (i & 31) of word
 * i >> 5.  A window within max_ac_errors of a sync word is a member of both.  Returns 0 or a negative BTBBX_E_*. */
int btbbx_slide_sets_two_level(int max_ac_errors, uint32_t *first_words, uint32_t *second_words, uint64_t *taps);

/* ---- device memory helpers (so C callers need not link HIP themselves) -------- */
void *btbbx_malloc(size_t bytes);
void btbbx_free(void *dptr);
int btbbx_memcpy_h2d(void *dst, const void *src, size_t bytes);
int btbbx_memcpy_d2h(void *dst, const void *src, size_t bytes);
int btbbx_memset(void *dptr, int value, size_t bytes);
int btbbx_sync(void *hip_stream);

/* ---- access-code scan -------------------------------------------------------- */
/* All pointers are DEVICE pointers.  n_streams packed streams of n_words words each
 * lie pitch_words apart; offsets [0, search_bits) of every stream are tested, which
 * needs search_bits + 63 <= 64 * n_words.  Hits are appended (unordered) to d_hits (16-byte aligned),
 * *d_hit_count counts ALL hits even beyond hit_cap.  If the count exceeds hit_cap the hit_cap records
 * that were stored are an UNSPECIFIED subset of the matches (whichever wavefronts came first), not the
 * first ones: size the buffer from the count and scan again, use btbbx_scan_first_device for
 * first-match semantics, or use the host wrappers below, which do this themselves.  The caller zeroes
 * *d_hit_count.  Asynchronous on hip_stream (NULL = the null stream). */
int btbbx_scan_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
		      uint32_t n_streams, uint64_t search_bits,
		      uint32_t lap, int max_ac_errors,
		      btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count,
		      void *hip_stream);

/* The two packed layouts a capture can have in HBM.  BTBBX_FMT_PACKED: LSB-first words (above).  BTBBX_FMT_PACKED_MSB: 8 symbols
 * per byte with the FIRST symbol in bit 7 -- what a dongle's bit-packed dumps look like.  The scan kernels take
 * either: MSB-first dwords are turned round in registers as they are loaded, the capture is not rewritten.  (The packet
 * decoders read LSB-first words: convert once with btbbx_msb_to_lsb_device before handing hits of an MSB capture to them.) */
#define BTBBX_FMT_PACKED  0      /* LSB-first packed words (const uint64_t *) */
#define BTBBX_FMT_SYMBOLS 1      /* one 0/1 symbol per byte (const char *), packed on the GPU (streaming ingest only) */
#define BTBBX_FMT_PACKED_MSB 2   /* 8 symbols per byte, first symbol in bit 7 */
int btbbx_scan_device_fmt(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
			  uint32_t n_streams, uint64_t search_bits,
			  uint32_t lap, int max_ac_errors, int format,
			  btbbx_hit *d_hits, uint32_t hit_cap, uint32_t *d_hit_count,
			  void *hip_stream);

/* First match only (btbb_find_ac semantics) of ONE stream: *d_first receives
 * (offset << 32 | lap << 8 | ac_errors) of the smallest matching offset, or
 * UINT64_MAX.  search_bits < 2^32.  The caller presets *d_first to UINT64_MAX. */
int btbbx_scan_first_device(const uint64_t *d_words, uint64_t n_words, uint64_t search_bits,
			    uint32_t lap, int max_ac_errors, uint64_t *d_first,
			    void *hip_stream);

/* Host convenience wrappers (copy in, scan on the GPU, copy out, sort by
 * (stream, offset)).  Return the number of hits found or a negative BTBBX_E_*.  The count may exceed
 * cap; then the cap records written are the cap SMALLEST (stream, offset) ones -- cap = 1 is the
 * first match of btbb_find_ac (lib/src/bluetooth_packet.c:444-464).  Safe to call from several host
 * threads at once (each call leases its own scratch memory and stream). */
int64_t btbbx_scan_host(const uint64_t *words, uint64_t n_words, uint64_t search_bits,
			uint32_t lap, int max_ac_errors, btbbx_hit *hits, uint64_t cap);
int64_t btbbx_scan_symbols(const char *symbols, uint64_t n_symbols, uint64_t search_length,
			   uint32_t lap, int max_ac_errors, btbbx_hit *hits, uint64_t cap);
/* First match only, from host memory: *first_hit = the match with the smallest offset in
 * [0, search_length) -- what btbb_find_ac returns (lib/src/bluetooth_packet.c:444-464).  Returns 1
 * (found), 0 (none) or a negative BTBBX_E_*.  search_length + 63 <= n_symbols, search_length < 2^32. */
int btbbx_find_first_symbols(const char *symbols, uint64_t n_symbols, uint64_t search_length,
			     uint32_t lap, int max_ac_errors, btbbx_hit *first_hit);

/* Time sharding over several GPUs of one node (no data-path collective): shard k of n
 * owns offsets [first_offset, first_offset + search_bits) of the capture and must be given words
 * [first_word, first_word + n_words) -- its slice plus a 63-symbol halo.  Offsets a shard reports are
 * local; global = first_offset + local.  One-process-per-GPU callers (MPI-style ranks) use
 * the plan directly; btbbx_scan_host_multi applies it over the listed devices with one host thread
 * per device, sorts per device and concatenates.  Every listed device must have been initialised
 * (btbbx_init_devices); a device may be listed more than once. */
typedef struct btbbx_shard {
	uint64_t first_word;      /* first word of the capture this shard reads */
	uint64_t n_words;         /* words it reads (slice + halo), 0 for an empty shard */
	uint64_t search_bits;     /* offsets it tests */
	uint64_t first_offset;    /* = 64 * first_word */
} btbbx_shard;
int btbbx_shard_plan(uint64_t search_bits, uint32_t n_shards, uint32_t shard, btbbx_shard *out);
int64_t btbbx_scan_host_multi(const uint64_t *words, uint64_t n_words, uint64_t search_bits,
			      uint32_t lap, int max_ac_errors, btbbx_hit *hits, uint64_t cap,
			      const int *devices, int n_devices);
void btbbx_sort_hits(btbbx_hit *hits, size_t n);
/* the same order for a hit list still in device memory (the host wrappers and the streaming ingest
 * sort here before copying out); synchronises hip_stream */
int btbbx_sort_hits_device(btbbx_hit *d_hits, uint32_t n, void *hip_stream);
/* The same order with the list's length still in device memory (the counter btbbx_scan_device filled): orders the
 * first min(*d_count, cap) records of d_hits in place -- no host round trip, no synchronisation, all work on
 * hip_stream.  d_scratch: btbbx_order_hits_scratch_bytes(cap) bytes of device memory owned by the caller (16-byte
 * aligned), so callers on different streams share nothing.  Offsets must stay below 2^47. */
size_t btbbx_order_hits_scratch_bytes(uint32_t cap);
int btbbx_order_hits_device(btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap, void *d_scratch,
			    size_t scratch_bytes, void *hip_stream);
/* ... for the hit list of a btbbx_scan_device call over n_streams streams and search_bits offsets: the same without the
 * pass that looks for the list's largest stream number and offset */
int btbbx_order_scan_hits_device(btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap, uint32_t n_streams,
				 uint64_t search_bits, void *d_scratch, size_t scratch_bytes, void *hip_stream);
/* btbbx_scan_device with the list coming back in (stream, offset) order: the scan kernels count every record they write in
 * the bucket the ordering will put it in, so the list is not read again for a histogram.  Arguments as btbbx_scan_device plus
 * the ordering scratch (btbbx_order_hits_scratch_bytes(cap)); cap >= 2; nothing is synchronised.  The call zeroes
 * *d_count itself (on hip_stream): the list is built from this call's matches only, records an earlier scan appended
 * are not carried over -- chain scans with btbbx_scan_device and order the whole list once with btbbx_order_hits_device. */
/* Scratch for btbbx_scan_ordered_device over n_streams streams of search_bits offsets each (round 6).  Where the scan has its
 * segment-slot form -- LAP_ANY with tables for up to two errors -- every wave leaves its hits, ranked, in slots of the 4032 offsets
 * they lie in (16 bytes per 4032 offsets of scratch), and the ordered list is one compaction of those slots: no bucket counters, no
 * scatter, no ranking pass.  The size returned covers that (and the general ordering, which stays the fallback for a stream the
 * slots cannot rank: one made of sync words); for other scans it equals btbbx_order_hits_scratch_bytes(cap).  A call that is
 * handed btbbx_order_hits_scratch_bytes(cap) bytes only runs the general ordering.  Needs btbb_init / btbbx_init first (the
 * answer depends on the tables). */
size_t btbbx_scan_ordered_scratch_bytes(uint64_t search_bits, uint32_t n_streams, uint32_t lap, uint32_t cap);
int btbbx_scan_ordered_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, uint32_t n_streams,
			      uint64_t search_bits, uint32_t lap, int max_ac_errors, btbbx_hit *d_hits, uint32_t cap,
			      uint32_t *d_count, void *d_scratch, size_t scratch_bytes, void *hip_stream);
int btbbx_scan_ordered_device_fmt(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, uint32_t n_streams,
				  uint64_t search_bits, uint32_t lap, int max_ac_errors, int format, btbbx_hit *d_hits, uint32_t cap,
				  uint32_t *d_count, void *d_scratch, size_t scratch_bytes, void *hip_stream);

/* symbols (one 0/1 byte each, bit 0 is used) -> packed words; n_words_out =
 * ceil(n_symbols / 64), the tail of the last word is zero */
int btbbx_pack_device(const uint8_t *d_symbols, uint64_t n_symbols, uint64_t *d_words,
		      void *hip_stream);
int btbbx_unpack_device(const uint64_t *d_words, uint64_t n_symbols, uint8_t *d_symbols,
			void *hip_stream);
/* bytes holding 8 symbols MSB first (first received symbol in bit 7) -> LSB-first words, in place (for the packet decoders;
 * the scans take BTBBX_FMT_PACKED_MSB directly) */
int btbbx_msb_to_lsb_device(uint64_t *d_words, uint64_t n_words, void *hip_stream);

/* ---- streaming ingest (live captures) ------------------------------ */
/* Feeds a capture to the GPU chunk by chunk: pinned double buffers, asynchronous host->device
 * copies overlapped with the scan of the previous chunk, a one-word carry so that access codes
 * straddling chunk boundaries are found exactly once.  Offsets in the returned hits are global
 * (symbols since btbbx_stream_open).  Every chunk except the last must be a multiple of 64
 * symbols.  feed() returns the hits of the chunk fed BEFORE this one (the current one is still
 * in flight); flush() waits for and returns the rest. */
typedef struct btbbx_stream btbbx_stream;
/* format: BTBBX_FMT_PACKED, BTBBX_FMT_SYMBOLS (packed on the GPU) or BTBBX_FMT_PACKED_MSB (scanned as it is), see above */
btbbx_stream *btbbx_stream_open(uint32_t lap, int max_ac_errors, uint64_t max_chunk_symbols, int format);
int64_t btbbx_stream_feed(btbbx_stream *s, const void *data, uint64_t n_symbols, btbbx_hit *hits, uint64_t cap);
/* zero-copy variant: write the next chunk straight into the pinned staging buffer returned by
 * acquire() (max_chunk_symbols bytes for SYMBOLS, /8 for PACKED), then submit() = feed() minus
 * the memcpy */
void *btbbx_stream_acquire(btbbx_stream *s);
int64_t btbbx_stream_submit(btbbx_stream *s, uint64_t n_symbols, btbbx_hit *hits, uint64_t cap);
int64_t btbbx_stream_flush(btbbx_stream *s, btbbx_hit *hits, uint64_t cap);
void btbbx_stream_close(btbbx_stream *s);

/* ---- synthetic traffic (same generator as libbtbb_amd/synth.py) --------------- */
/* words [first_word, first_word + n_words) of the infinite stream `seed`:
 * iid noise plus one sync word per `stride` symbols (stride >= 512), LAP random or
 * fixed_lap (>= 0), k % err_cycle bit errors in sync-word bits 0..56. */
int btbbx_synth_device(uint64_t *d_words, uint64_t first_word, uint64_t n_words,
		       uint64_t seed, uint32_t stride, int64_t fixed_lap, uint32_t err_cycle,
		       void *hip_stream);

/* ---- packet chain ------------------------------------------------------------ */
/* Cut packets out of packed streams: packet i = bits [offset, offset + length) of
 * stream hits[i].stream, zero padded to BTBBX_PKT_WORDS words. */
int btbbx_gather_packets_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
				const btbbx_hit *d_hits, uint32_t n_packets, uint32_t max_length,
				uint64_t *d_packets, uint32_t *d_lengths, void *hip_stream);

/* 64 clock trials per packet: d_trials[i * 64 + c] = state after try_clock(c) and
 * crc_check(c) run in clock order c = 0..63 on packet i (bluetooth_piconet.c:675-690
 * with GOT_FIRST_PACKET clear).  d_in[i] gives each packet's entry state. */
int btbbx_trials_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
			btbbx_trial *d_trials, void *hip_stream);

/* The HEC-only half of the brute force: d_table[i * 64 + c] = try_clock(c)'s return value (the UAP
 * candidate for CLK1-6 = c, bluetooth_packet.c:1178-1195 / uap_from_hec :693-705) in the low byte
 * and the packet type that clock yields in the high byte; 0 when the header's FEC 1/3 fails.
 * Reads 8 bytes and writes 128 bytes per packet.  d_in may be NULL (all packets whitened);
 * d_table 16-byte aligned. */
int btbbx_uap_table_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
			   uint16_t *d_table, void *hip_stream);

/* header_present + decode_header + decode_payload with the clock / UAP in d_in */
int btbbx_decode_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
			btbbx_pkt_out *d_out, void *hip_stream);

/* The same for packets that still lie in the packed streams: packet i is what
 * btbbx_gather_packets_device would cut out for d_hits[i] (same max_length, same captured
 * length, zeros behind it), decoded without the intermediate 400-byte row.  d_in[i].length is
 * ignored; d_lengths (may be NULL) receives the captured lengths. */
int btbbx_decode_hits_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
			     const btbbx_hit *d_hits, const btbbx_pkt_in *d_in, uint32_t n_packets,
			     uint32_t max_length, btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream);
/* ... with the number of hits still in device memory: decodes the first min(*d_count, cap) hits (launched for cap);
 * scan -> btbbx_order_hits_device -> this call is the known-LAP chain without a host round trip */
int btbbx_decode_hits_counted_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
				     const btbbx_hit *d_hits, const btbbx_pkt_in *d_in, const uint32_t *d_count,
				     uint32_t cap, uint32_t max_length, btbbx_pkt_out *d_out, uint32_t *d_lengths,
				     void *hip_stream);
/* ... for a capture of ONE piconet, without a btbbx_pkt_in per packet: every packet enters with the state of *entry
 * (a HOST pointer; flags, UAP, type, llid, flow -- what btbb_packet_set_data / btbb_packet_set_uap / the flag setters leave,
 * lib/src/bluetooth_packet.c:467-480; its length is ignored) and the clock entry->clkn + offset / clk_div: CLK1-27 advances
 * once per 625 symbols at 1 Msym/s, so a receiver that knows its clock at the first symbol of the buffer knows it for
 * every access code found in it (the clkn argument of btbb_packet_set_data, lib/src/btbb.h:116).  d_count may be NULL
 * (then cap records are decoded). */
int btbbx_decode_hits_piconet_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
				     const btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap,
				     const btbbx_pkt_in *entry, uint32_t clk_div, uint32_t max_length,
				     btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream);
/* The call above takes symbol 0 of the buffer for the FIRST symbol of a slot.  A buffer that starts clk_phase symbols
 * into a slot (0 <= clk_phase < clk_div) is decoded with the clock entry->clkn + (offset + clk_phase) / clk_div: without the
 * phase, every access code behind the next slot boundary would get a clock one too low and fail its header check. */
int btbbx_decode_hits_piconet_phase_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
					   const btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap,
					   const btbbx_pkt_in *entry, uint32_t clk_div, uint32_t clk_phase, uint32_t max_length,
					   btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream);

/* ---- hop selection and CLK1-27 reversal ------------------------- */
#define BTBBX_SEQUENCE_LENGTH 134217728u   /* values of CLK1-27, bluetooth_piconet.h:102 */

/* the inputs of the hop selection kernel for one piconet (precalc + address_precalc,
 * bluetooth_piconet.c:171-217) */
typedef struct btbbx_hop_cfg {
	uint32_t address;        /* (UAP << 24 | LAP) & 0xfffffff */
	uint8_t  afh;            /* BTBB_IS_AFH: index the bank modulo used_channels */
	uint8_t  used_channels;
	uint8_t  reserved[2];
	uint8_t  bank[80];       /* frequency register bank; entries past the filled part are 0 */
} btbbx_hop_cfg;

/* afh_map == NULL: basic hopping over all 79 channels; otherwise the 10-byte AFH channel map */
void btbbx_hop_cfg_init(btbbx_hop_cfg *cfg, uint32_t address, const uint8_t *afh_map);

/* channels of CLK1-27 values [first, first + count) -- what gen_hops stores in
 * sequence[first .. first+count); first and count multiples of 64; d_sequence receives count
 * bytes.  The whole 2^27-entry pattern is 128 MiB. */
int btbbx_hop_sequence_device(const btbbx_hop_cfg *cfg, uint64_t first, uint64_t count,
			      uint8_t *d_sequence, void *hip_stream);
/* hop(clock) for arbitrary CLK1-27 values (taken modulo 2^27) */
int btbbx_hop_channels_device(const btbbx_hop_cfg *cfg, const uint32_t *d_clocks, uint32_t n,
			      uint8_t *d_channels, void *hip_stream);

/* CLK1-27 reversal: the candidate list lives in HBM.  open() = init_candidates
 * (bluetooth_piconet.c:455-472): all clocks congruent clk6 mod 64 whose hop is `channel`
 * (a value > 127 matches nothing, as the reference compares signed chars);
 * aliased != 0 compares ((ch + 24) % 25) + 26 instead (:449-452). */
typedef struct btbbx_hop_reversal btbbx_hop_reversal;
btbbx_hop_reversal *btbbx_hop_reversal_open(const btbbx_hop_cfg *cfg, uint32_t clk6, uint8_t channel,
					    int aliased, int *n_candidates);
/* channel_winnow over n_obs observed hops in order (index offset relative to the first packet,
 * channel), stopping after the first one that leaves <= 1 candidate (btbb_winnow, :614-645).
 * *stop = how many observations were applied before that one (n_obs if none did), *count =
 * candidates left, *cand0 = the first of them.  Returns 0 or a negative BTBBX_E_*. */
int btbbx_hop_reversal_winnow(btbbx_hop_reversal *h, const int32_t *index_offsets, const uint8_t *channels,
			      uint32_t n_obs, uint32_t *stop, uint32_t *count, uint32_t *cand0);
int64_t btbbx_hop_reversal_candidates(btbbx_hop_reversal *h, uint32_t *dst, uint64_t cap);  /* ascending */
void btbbx_hop_reversal_close(btbbx_hop_reversal *h);

/* piconet introspection for tests and tools: what the reference keeps in struct btbb_piconet
 * (bluetooth_piconet.h:59-85).  field: 0 num_candidates, 1 winnowed, 2 packets_observed,
 * 3 total_packets_observed, 4 first_pkt_time, 5 flags, 6 used_channels */
int64_t btbbx_piconet_state(const void *piconet, int field);
int64_t btbbx_piconet_candidates(const void *piconet, uint32_t *dst, uint64_t cap);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_BTBBX_H */
