// piconet.cpp -- the callers of the hot path: piconet object, UAP / CLK1-6 discovery and the
// packet dispatcher of include/btbb.h.
//
// Restates lib/src/bluetooth_piconet.c:41-168 (object + accessors), :648-750
// (btbb_uap_from_header), :792-899 (AFH print, survey, btbb_process_packet).  The 64
// candidate trials (try_clock + crc_check per CLK1-6 value) run as ONE GPU launch
// (packet.hip: trials_kernel); the candidate elimination -- inherently sequential across the
// packets of a piconet -- is replayed here on the host from that table, and a second, small
// launch merges the writes of exactly the executed trials (captured per trial by the first
// launch) so that the packet object ends up as the reference leaves it.
// CLK1-27 reversal (:365-413 pattern cache, :475-498 btbb_init_hop_reversal, :501-543 try_hop,
// :575-645 winnowing) keeps the reference's host state machine; the hop selection itself and
// the candidate lists live on the GPU (hop.hip), with no 128 MiB pattern table: what the
// cache remembers per address is the 88-byte kernel configuration.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "packet_obj.h"
#include "../../include/btbb.h"

static int survey_mode = 0;
// pattern cache: like the reference's (whose uthash key length is 4 bytes, bluetooth_piconet.c:400,
// 407) it is keyed by UAP << 24 | LAP only -- the first pattern made for an address wins, whatever
// its AFH state was
static std::unordered_map<uint32_t, btbbx_hop_cfg *> pattern_cache;
static std::unordered_map<uint32_t, btbb_piconet *> survey_map;
static std::vector<uint32_t> survey_order;

extern "C" {

btbb_piconet *btbb_piconet_new(void)
{
	btbb_piconet *pn = (btbb_piconet *)calloc(1, sizeof(btbb_piconet));
	if (pn)
		pn->refcount = 1;
	return pn;
}

void btbb_piconet_ref(btbb_piconet *pn) { pn->refcount++; }

void btbb_piconet_unref(btbb_piconet *pn)
{
	pn->refcount--;
	if (pn->refcount == 0) {
		btbbx_hop_reversal_close(pn->reversal);
		free(pn);
	}
}

int btbb_init_survey(void)
{
	survey_mode = 1;
	return 0;
}

void btbb_piconet_set_flag(btbb_piconet *pn, int flag, int val)
{
	uint32_t mask = 1u << flag;
	pn->flags &= ~mask;
	if (val)
		pn->flags |= mask;
}

int btbb_piconet_get_flag(const btbb_piconet *pn, int flag) { return (pn->flags & (1u << flag)) != 0; }

void btbb_init_piconet(btbb_piconet *pn, uint32_t lap)
{
	pn->LAP = lap;
	btbb_piconet_set_flag(pn, BTBB_LAP_VALID, 1);
}

void btbb_piconet_set_uap(btbb_piconet *pn, uint8_t uap)
{
	pn->UAP = uap;
	btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 1);
}

uint8_t btbb_piconet_get_uap(const btbb_piconet *pn) { return pn->UAP; }
uint32_t btbb_piconet_get_lap(const btbb_piconet *pn) { return pn->LAP; }
uint16_t btbb_piconet_get_nap(const btbb_piconet *pn) { return pn->NAP; }

uint64_t btbb_piconet_get_bdaddr(const btbb_piconet *pn)
{
	return ((uint64_t)pn->NAP) << 32 | ((uint32_t)pn->UAP) << 24 | pn->LAP;
}

int btbb_piconet_get_clk_offset(const btbb_piconet *pn) { return pn->clk_offset; }
void btbb_piconet_set_clk_offset(btbb_piconet *pn, int clk_offset) { pn->clk_offset = clk_offset; }

uint8_t *btbb_piconet_get_afh_map(btbb_piconet *pn) { return pn->afh_map; }

uint8_t btbb_piconet_set_channel_seen(btbb_piconet *pn, uint8_t channel)
{
	if (!(pn->afh_map[channel / 8] & (1 << (channel % 8)))) {
		pn->afh_map[channel / 8] |= (uint8_t)(1 << (channel % 8));
		pn->used_channels++;
		return 1;
	}
	return 0;
}

uint8_t btbb_piconet_clear_channel_seen(btbb_piconet *pn, uint8_t channel)
{
	if (pn->afh_map[channel / 8] & (1 << (channel % 8))) {
		pn->afh_map[channel / 8] &= (uint8_t)~(1 << (channel % 8));
		pn->used_channels--;
		return 1;
	}
	return 0;
}

uint8_t btbb_piconet_get_channel_seen(btbb_piconet *pn, uint8_t channel)
{
	if (channel < 79)
		return (pn->afh_map[channel / 8] & (1 << (channel % 8))) != 0;
	return 1;
}

void btbb_print_afh_map(btbb_piconet *pn)
{
	const uint8_t *m = pn->afh_map;
	printf("AFH map: 0x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x\n",
	       m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9]);
}

} // extern "C"

/* bluetooth_piconet.c:547-572 */
static void piconet_reset(btbb_piconet *pn)
{
	if (btbb_piconet_get_flag(pn, BTBB_HOP_REVERSAL_INIT)) {
		btbbx_hop_reversal_close(pn->reversal);
		pn->reversal = NULL;
		pn->pattern = NULL;
	}
	btbb_piconet_set_flag(pn, BTBB_GOT_FIRST_PACKET, 0);
	btbb_piconet_set_flag(pn, BTBB_HOP_REVERSAL_INIT, 0);
	btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 0);
	btbb_piconet_set_flag(pn, BTBB_CLK6_VALID, 0);
	btbb_piconet_set_flag(pn, BTBB_CLK27_VALID, 0);
	pn->packets_observed = 0;
	btbb_piconet_set_flag(pn, BTBB_IS_AFH, btbb_piconet_get_flag(pn, BTBB_LOOKS_LIKE_AFH));
}

extern "C" {

/* bluetooth_piconet.c:648-750 */
int btbb_uap_from_header(btbb_packet *pkt, btbb_piconet *pn)
{
	const uint32_t clkn = pkt->clkn;
	int remaining = 0, first_clock = 0, result = -1;

	if (!btbb_piconet_get_flag(pn, BTBB_GOT_FIRST_PACKET))
		pn->first_pkt_time = clkn;
	btbb_piconet_set_channel_seen(pn, pkt->channel);
	if (pn->packets_observed < PN_MAX_PATTERN) {
		pn->pattern_indices[pn->packets_observed] = (int)(clkn - pn->first_pkt_time);
		pn->pattern_channels[pn->packets_observed] = pkt->channel;
	} else {
		printf("Oops. More hops than we can remember.\n");
		piconet_reset(pn);
		return 0;
	}
	pn->packets_observed++;
	pn->total_packets_observed++;

	// all 64 CLK1-6 candidates in one launch: trial[c] = {try_clock(c), type, crc_check(c)}
	btbbx_trial trial[64];
	if (packet_gpu_trials(pkt, trial)) {
		fprintf(stderr, "btbb_uap_from_header: GPU path failed: %s\n", btbbx_last_error());
		return 0;
	}

	TrialPlan plan = {0, 0, (uint32_t)((clkn - pn->first_pkt_time) & 63)};
	const int first = !btbb_piconet_get_flag(pn, BTBB_GOT_FIRST_PACKET);
	for (int count = 0; count < 64 && result < 0; count++) {
		if (pn->clock6_candidates[count] > -1 || first) {
			const uint32_t clock = ((uint32_t)count + clkn - pn->first_pkt_time) % 64;
			const btbbx_trial &t = trial[clock];
			const uint8_t UAP = t.uap;
			int crc_chk = -1;
			plan.try_mask |= 1ULL << count;
			if (first || UAP == pn->clock6_candidates[count]) {
				crc_chk = t.rv;
				plan.crc_mask |= 1ULL << count;
			}
			if (btbb_piconet_get_flag(pn, BTBB_UAP_VALID) && UAP != pn->UAP)
				crc_chk = -1;
			switch (crc_chk) {
			case -1:
			case 0:
				pn->clock6_candidates[count] = -1;
				break;
			case 1:
			case 2:
				pn->clock6_candidates[count] = UAP;
				first_clock = count;
				remaining++;
				break;
			default:
				pn->clk_offset = (count - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
				if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
					printf("Correct CRC! UAP = 0x%x found after %d total packets.\n",
					       UAP, pn->total_packets_observed);
				else
					printf("Correct CRC! CLK6 = 0x%x found after %d total packets.\n",
					       pn->clk_offset, pn->total_packets_observed);
				pn->UAP = UAP;
				btbb_piconet_set_flag(pn, BTBB_CLK6_VALID, 1);
				btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 1);
				pn->total_packets_observed = 0;
				result = 1;
				break;
			}
		}
	}

	// leave the packet object as the executed trials leave it in the reference
	if (packet_gpu_trials_commit(pkt, &plan))
		fprintf(stderr, "btbb_uap_from_header: state replay failed: %s\n", btbbx_last_error());
	if (result >= 0)
		return result;

	btbb_piconet_set_flag(pn, BTBB_GOT_FIRST_PACKET, 1);
	if (remaining == 1) {
		pn->clk_offset = (first_clock - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
		if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
			printf("UAP = 0x%x found after %d total packets.\n",
			       pn->clock6_candidates[first_clock], pn->total_packets_observed);
		else
			printf("CLK6 = 0x%x found after %d total packets.\n",
			       pn->clk_offset, pn->total_packets_observed);
		pn->UAP = (uint8_t)pn->clock6_candidates[first_clock];
		btbb_piconet_set_flag(pn, BTBB_CLK6_VALID, 1);
		btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 1);
		pn->total_packets_observed = 0;
		return 1;
	}
	if (remaining == 0)
		piconet_reset(pn);
	return 0;
}

/* bluetooth_piconet.c:817-849 */
static btbb_piconet *get_piconet(uint32_t lap)
{
	auto it = survey_map.find(lap);
	if (it != survey_map.end())
		return it->second;
	btbb_piconet *pn = btbb_piconet_new();
	btbb_init_piconet(pn, lap);
	survey_map[lap] = pn;
	survey_order.push_back(lap);
	return pn;
}

btbb_piconet *btbb_next_survey_result(void)
{
	while (!survey_order.empty()) {
		uint32_t lap = survey_order.front();
		survey_order.erase(survey_order.begin());
		auto it = survey_map.find(lap);
		if (it != survey_map.end()) {
			btbb_piconet *pn = it->second;
			survey_map.erase(it);
			return pn;
		}
	}
	return NULL;
}

/* get_hop_pattern + gen_hop_pattern, bluetooth_piconet.c:365-413 */
static void get_hop_pattern(btbb_piconet *pn)
{
	const uint32_t key = ((uint32_t)pn->UAP << 24) | pn->LAP;
	auto it = pattern_cache.find(key);
	if (it != pattern_cache.end()) {
		printf("\nFound hopping sequence in cache.\n");
		pn->pattern = it->second;
		return;
	}
	printf("\nCalculating complete hopping sequence.\n");
	const int afh = btbb_piconet_get_flag(pn, BTBB_IS_AFH);
	int j = 0;
	for (int i = 0; i < 79; i++) {                 /* precalc, :171-194: the bank persists in the piconet */
		const int chan = (i * 2) % 79;
		if (!afh)
			pn->bank[i] = (uint8_t)chan;
		else if (btbb_piconet_get_channel_seen(pn, (uint8_t)chan))
			pn->bank[j++] = (uint8_t)chan;
	}
	btbbx_hop_cfg *cfg = (btbbx_hop_cfg *)calloc(1, sizeof(*cfg));
	cfg->address = key & 0xfffffff;
	cfg->afh = (uint8_t)afh;
	cfg->used_channels = pn->used_channels;
	memcpy(cfg->bank, pn->bank, sizeof(cfg->bank));
	pattern_cache[key] = cfg;
	pn->pattern = cfg;
	printf("Hopping sequence calculated.\n");
}

void btbb_piconet_set_afh_map(btbb_piconet *pn, uint8_t *afh_map)
{
	pn->used_channels = 0;
	for (int i = 0; i < 10; i++) {
		pn->afh_map[i] = afh_map[i];
		pn->used_channels += (uint8_t)__builtin_popcount(afh_map[i]);
	}
	if (btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
		get_hop_pattern(pn);
}

/* bluetooth_piconet.c:475-498; init_candidates (:455-472) is one GPU pass over the 2^21 clocks
 * that agree with the known CLK1-6 */
int btbb_init_hop_reversal(int aliased, btbb_piconet *pn)
{
	get_hop_pattern(pn);
	btbbx_hop_reversal_close(pn->reversal);    /* the reference leaks the previous list */
	const uint32_t clock = ((uint32_t)pn->clk_offset + pn->first_pkt_time) & 0x3f;
	int n = 0;
	pn->reversal = btbbx_hop_reversal_open(pn->pattern, clock, pn->pattern_channels[0], pn->aliased, &n);
	if (!pn->reversal)
		fprintf(stderr, "btbb_init_hop_reversal: GPU path failed: %s\n", btbbx_last_error());
	pn->num_candidates = n;
	pn->winnowed = 0;
	btbb_piconet_set_flag(pn, BTBB_HOP_REVERSAL_INIT, 1);
	btbb_piconet_set_flag(pn, BTBB_CLK27_VALID, 0);
	btbb_piconet_set_flag(pn, BTBB_IS_ALIASED, aliased);
	printf("%d initial CLK1-27 candidates\n", pn->num_candidates);
	return pn->num_candidates;
}

/* btbb_winnow + channel_winnow, bluetooth_piconet.c:575-645.  All observations not yet used go
 * to the GPU in one call, which stops where the reference's loop would `break`; the host then
 * replays the loop's side effects (flags, clk_offset, reset, AFH heuristics) in order. */
int btbb_winnow(btbb_piconet *pn)
{
	int new_count = pn->num_candidates;
	if (pn->winnowed >= pn->packets_observed || !pn->reversal)
		return new_count;
	const uint32_t n_obs = (uint32_t)(pn->packets_observed - pn->winnowed);
	uint32_t stop = 0, count = 0, cand0 = 0;
	if (btbbx_hop_reversal_winnow(pn->reversal, pn->pattern_indices + pn->winnowed,
				      pn->pattern_channels + pn->winnowed, n_obs, &stop, &count, &cand0)) {
		fprintf(stderr, "btbb_winnow: GPU path failed: %s\n", btbbx_last_error());
		return new_count;
	}
	for (uint32_t k = 0; k < n_obs; k++, pn->winnowed++) {
		const int w = pn->winnowed;
		const int index = pn->pattern_indices[w];
		const uint8_t channel = pn->pattern_channels[w];
		if (k == stop) {                           /* this hop leaves <= 1 candidate */
			pn->num_candidates = new_count = (int)count;
			if (count == 1) {
				pn->clk_offset = (int)((cand0 << 1) - (pn->first_pkt_time << 1));
				printf("\nAcquired CLK1-27 = 0x%07x\n", cand0);
				btbb_piconet_set_flag(pn, BTBB_CLK27_VALID, 1);
			} else {
				piconet_reset(pn);
			}
			return new_count;                  /* pn->winnowed stays, as after the reference's break */
		}
		/* The reference also looks one entry below both arrays when w == 0 (:627-628), i.e. at
		 * clock6_candidates[63] and at the top byte of pattern_indices[999]; same values here. */
		const int last_index = w > 0 ? pn->pattern_indices[w - 1] : pn->clock6_candidates[63];
		const uint8_t last_channel = w > 0 ? pn->pattern_channels[w - 1]
						   : (uint8_t)((uint32_t)pn->pattern_indices[PN_MAX_PATTERN - 1] >> 24);
		if (!btbb_piconet_get_flag(pn, BTBB_LOOKS_LIKE_AFH) && index == last_index + 1 &&
		    channel == last_channel) {
			btbb_piconet_set_flag(pn, BTBB_LOOKS_LIKE_AFH, 1);
			printf("Hopping pattern appears to be AFH\n");
		}
	}
	pn->num_candidates = new_count = (int)count;
	return new_count;
}

/* try_hop, bluetooth_piconet.c:501-543 */
static void try_hop(btbb_packet *pkt, btbb_piconet *pn)
{
	uint8_t filter_uap = pn->UAP;
	btbb_decode(pkt);
	if (btbb_piconet_get_flag(pn, BTBB_HOP_REVERSAL_INIT)) {
		if (pn->packets_observed < PN_MAX_PATTERN) {   /* the reference writes past the arrays here */
			pn->pattern_indices[pn->packets_observed] = (int)(pkt->clkn - pn->first_pkt_time);
			pn->pattern_channels[pn->packets_observed] = pkt->channel;
			pn->packets_observed++;
			pn->total_packets_observed++;
		}
		btbb_winnow(pn);
		if (btbb_piconet_get_flag(pn, BTBB_CLK27_VALID)) {
			printf("got CLK1-27\n");
			printf("clock offset = %d.\n", pn->clk_offset);
		}
	} else if (btbb_piconet_get_flag(pn, BTBB_CLK6_VALID)) {
		btbb_uap_from_header(pkt, pn);
		if (btbb_piconet_get_flag(pn, BTBB_CLK27_VALID)) {
			printf("got CLK1-27\n");
			printf("clock offset = %d.\n", pn->clk_offset);
		}
	} else if (btbb_uap_from_header(pkt, pn)) {
		if (filter_uap == pn->UAP) {
			btbb_init_hop_reversal(0, pn);
			btbb_winnow(pn);
		} else {
			printf("failed to confirm UAP\n");
		}
	}
	if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID)) {
		btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 1);
		pn->UAP = filter_uap;
	}
}

int64_t btbbx_piconet_state(const void *piconet, int field)
{
	const btbb_piconet *pn = (const btbb_piconet *)piconet;
	switch (field) {
	case 0: return pn->num_candidates;
	case 1: return pn->winnowed;
	case 2: return pn->packets_observed;
	case 3: return pn->total_packets_observed;
	case 4: return pn->first_pkt_time;
	case 5: return pn->flags;
	case 6: return pn->used_channels;
	default: return BTBBX_E_ARG;
	}
}

int64_t btbbx_piconet_candidates(const void *piconet, uint32_t *dst, uint64_t cap)
{
	const btbb_piconet *pn = (const btbb_piconet *)piconet;
	if (!pn->reversal)
		return 0;
	return btbbx_hop_reversal_candidates(pn->reversal, dst, cap);
}

/* bluetooth_piconet.c:851-899 */
int btbb_process_packet(btbb_packet *pkt, btbb_piconet *pn)
{
	if (survey_mode) {
		pn = get_piconet(btbb_packet_get_lap(pkt));
		btbb_piconet_set_channel_seen(pn, pkt->channel);
		if (btbb_header_present(pkt) && !btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
			btbb_uap_from_header(pkt, pn);
		return 0;
	}
	if (pn)
		btbb_piconet_set_channel_seen(pn, pkt->channel);
	if (pn && btbb_piconet_get_flag(pn, BTBB_LAP_VALID) && btbb_header_present(pkt)) {
		if (btbb_piconet_get_flag(pn, BTBB_FOLLOWING)) {
			btbb_packet_set_uap(pkt, btbb_piconet_get_uap(pn));
			btbb_packet_set_flag(pkt, BTBB_CLK6_VALID, 1);
			btbb_packet_set_flag(pkt, BTBB_CLK27_VALID, 1);
			if (btbb_decode(pkt))
				btbb_print_packet(pkt);
			else
				printf("Failed to decode packet\n");
		} else if (btbb_piconet_get_uap(pn)) {
			try_hop(pkt, pn);
			if (btbb_piconet_get_flag(pn, BTBB_CLK6_VALID) &&
			    btbb_piconet_get_flag(pn, BTBB_CLK27_VALID)) {
				btbb_piconet_set_flag(pn, BTBB_FOLLOWING, 1);
				return -1;
			}
		} else {
			btbb_uap_from_header(pkt, pn);
		}
	}
	return 0;
}

} // extern "C"
