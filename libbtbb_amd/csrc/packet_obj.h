// packet_obj.h -- host-side packet and piconet objects behind the opaque handles of
// include/btbb.h (counterparts of lib/src/bluetooth_packet.h:52-112 and
// lib/src/bluetooth_piconet.h:32-99; own definitions, never exposed to callers).
#pragma once
#include <stdint.h>

#define PKT_MAX_SYMBOLS 3125
#define PKT_MAX_PAYLOAD_BITS 2744
#define PN_MAX_PATTERN 1000

struct btbb_packet {
	uint32_t refcount;
	uint32_t flags;
	uint8_t channel;
	uint8_t UAP;
	uint16_t NAP;
	uint32_t LAP;
	uint8_t modulation;
	uint8_t transport;
	uint8_t packet_type;
	uint8_t packet_lt_addr;
	uint8_t packet_flags;
	uint8_t packet_hec;
	char packet_header[18];             // one bit per char
	int payload_header_length;
	char payload_header[16];
	uint8_t payload_llid;
	uint8_t payload_flow;
	int payload_length;
	char payload[PKT_MAX_PAYLOAD_BITS]; // one bit per char
	uint32_t clkn;                      // CLK1-27
	uint8_t ac_errors;
	uint16_t length;
	char symbols[PKT_MAX_SYMBOLS];      // one symbol per char
};

struct btbb_piconet {
	uint32_t refcount;
	uint32_t flags;
	uint8_t afh_map[10];
	uint8_t used_channels;
	uint32_t LAP;
	uint8_t UAP;
	uint16_t NAP;
	int packets_observed;
	int total_packets_observed;
	int clock6_candidates[64];
	int pattern_indices[PN_MAX_PATTERN];
	uint8_t pattern_channels[PN_MAX_PATTERN];
	int clk_offset;
	uint32_t first_pkt_time;
	// CLK1-27 reversal (bluetooth_piconet.h:38-40, 49-69)
	int aliased;                        // never written through the public API, as in the reference
	uint8_t bank[80];                   // frequency register bank of the last precalc
	const struct btbbx_hop_cfg *pattern;// the reference's `sequence`: cached per address, NULL = none
	struct btbbx_hop_reversal *reversal;// the reference's `clock_candidates`, kept in HBM
	int num_candidates;
	int winnowed;
};

// GPU round trips for one packet object (btbb_api.cpp)
#define DEC_HEADER   1
#define DEC_PAYLOAD  2
#define DEC_TRIALS   4
struct TrialPlan {            // which of the 64 candidate trials to replay, in count order
	uint64_t try_mask;        // counts whose try_clock runs
	uint64_t crc_mask;        // counts whose crc_check runs
	uint32_t clock_offset;    // clock(count) = (count + clock_offset) % 64
};
int packet_gpu_decode(btbb_packet *pkt, uint32_t mode, const TrialPlan *plan, int *header_present,
		      int *header_rv, int *payload_rv);
int packet_gpu_trials(const btbb_packet *pkt, struct btbbx_trial *trials64);
int packet_gpu_trials_commit(btbb_packet *pkt, const TrialPlan *plan);   // directly after packet_gpu_trials
