// piconet.cpp -- the callers of the hot path: piconet object, UAP / CLK1-6 discovery and the
// packet dispatcher of include/btbb.h.
//
// Restates lib/src/bluetooth_piconet.c:41-168 (object + accessors), :648-750
// (btbb_uap_from_header), :792-899 (AFH print, survey, btbb_process_packet).  The 64
// candidate trials (try_clock + crc_check per CLK1-6 value) run as ONE GPU launch
// (packet.hip: trials_kernel); the candidate elimination -- inherently sequential across the
// packets of a piconet -- is replayed here on the host from that table, and a second launch
// replays exactly the executed trials with all packet state written so that the packet
// object ends up as the reference leaves it.
// Hop reversal (CLK1-27, :170-645) is outside the hot path; where the reference would start
// it we print a note and leave BTBB_CLK27_VALID clear.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "packet_obj.h"
#include "../../include/btbb.h"

static int survey_mode = 0;
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
	if (pn->refcount == 0)
		free(pn);
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
	if (packet_gpu_decode(pkt, DEC_TRIALS, &plan, nullptr, nullptr, nullptr))
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

/* try_hop, bluetooth_piconet.c:501-543, up to the hop-reversal boundary */
static void try_hop(btbb_packet *pkt, btbb_piconet *pn)
{
	uint8_t filter_uap = pn->UAP;
	btbb_decode(pkt);
	if (btbb_piconet_get_flag(pn, BTBB_HOP_REVERSAL_INIT)) {
		fprintf(stderr, "btbb: CLK1-27 hop reversal is not part of this build\n");
	} else if (btbb_piconet_get_flag(pn, BTBB_CLK6_VALID)) {
		btbb_uap_from_header(pkt, pn);
	} else if (btbb_uap_from_header(pkt, pn)) {
		if (filter_uap == pn->UAP)
			fprintf(stderr, "btbb: UAP confirmed; CLK1-27 hop reversal is not part of this build\n");
		else
			printf("failed to confirm UAP\n");
	}
	if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID)) {
		btbb_piconet_set_flag(pn, BTBB_UAP_VALID, 1);
		pn->UAP = filter_uap;
	}
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
