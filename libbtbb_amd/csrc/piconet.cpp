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

// Back to "nothing known but the LAP" (bluetooth_piconet.c:547-572): everything the discovery
// stages established is dropped, the AFH guess of the last attempt becomes the next attempt's mode.
static void piconet_reset(btbb_piconet *pn)
{
	const uint32_t discovered = (1u << BTBB_GOT_FIRST_PACKET) | (1u << BTBB_HOP_REVERSAL_INIT) | (1u << BTBB_UAP_VALID) |
				    (1u << BTBB_CLK6_VALID) | (1u << BTBB_CLK27_VALID);
	if (pn->flags & (1u << BTBB_HOP_REVERSAL_INIT)) {
		btbbx_hop_reversal_close(pn->reversal);
		pn->reversal = NULL;
		pn->pattern = NULL;
	}
	pn->flags &= ~(discovered | (1u << BTBB_IS_AFH));
	if (pn->flags & (1u << BTBB_LOOKS_LIKE_AFH))
		pn->flags |= 1u << BTBB_IS_AFH;
	pn->packets_observed = 0;
}

// ---- UAP / CLK1-6 discovery (bluetooth_piconet.c:648-750) -------------------------------------
//
// The reference walks the 64 candidate values of "CLK1-6 of the first packet" one after the other,
// running try_clock / crc_check for each.  Here the GPU has already produced all 64 results
// (trial[clock] = {UAP, type, crc_check verdict}), and nothing a candidate does depends on an
// earlier candidate except the early exit at the first confirmed CRC.  So the elimination is
// written as set arithmetic over 64-bit candidate sets, one bit per candidate:
//
//   live    candidates still standing (all of them for the first packet of a piconet)
//   checked live candidates whose CRC the reference would have computed (UAP agrees with the one
//           remembered for that candidate; all of them for the first packet)
//   proven  checked candidates, consistent with a known piconet UAP, whose CRC verdict is a pass
//   kept    the same with an inconclusive verdict (1 or 2)
//
// The lowest member of `proven` ends the walk: candidates above it are left untouched.

struct CandidateSets {
	uint64_t live, checked, proven, kept;
	uint8_t uap[64];                              // UAP each candidate implies for this packet
};

static inline uint64_t below(int bit) { return bit >= 64 ? ~0ULL : (1ULL << bit) - 1; }

static CandidateSets classify_candidates(const btbb_piconet *pn, const btbbx_trial *trial, uint32_t rot)
{
	const bool opening = !btbb_piconet_get_flag(pn, BTBB_GOT_FIRST_PACKET);
	const bool uap_known = btbb_piconet_get_flag(pn, BTBB_UAP_VALID) != 0;
	CandidateSets s = {};
	for (int c = 0; c < 64; c++) {
		const btbbx_trial &t = trial[(c + rot) & 63];
		const uint64_t me = 1ULL << c;
		s.uap[c] = t.uap;
		if (!opening && pn->clock6_candidates[c] < 0)
			continue;
		s.live |= me;
		if (!opening && t.uap != pn->clock6_candidates[c])
			continue;
		s.checked |= me;
		if (uap_known && t.uap != pn->UAP)
			continue;                             // CRC was computed, verdict overruled
		if (t.rv == 1 || t.rv == 2)
			s.kept |= me;
		else if (t.rv != 0)
			s.proven |= me;
	}
	return s;
}

// record one observed hop; false = pattern memory exhausted
static bool remember_hop(btbb_piconet *pn, const btbb_packet *pkt)
{
	if (!btbb_piconet_get_flag(pn, BTBB_GOT_FIRST_PACKET))
		pn->first_pkt_time = pkt->clkn;
	btbb_piconet_set_channel_seen(pn, pkt->channel);
	const int slot = pn->packets_observed;
	if (slot >= PN_MAX_PATTERN)
		return false;
	pn->pattern_indices[slot] = (int)(pkt->clkn - pn->first_pkt_time);
	pn->pattern_channels[slot] = pkt->channel;
	pn->packets_observed = slot + 1;
	pn->total_packets_observed++;
	return true;
}

// candidate `count` is the CLK1-6 of the first packet: the piconet's UAP and clock offset follow
static void settle_clock6(btbb_piconet *pn, int count, uint8_t uap, const char *prefix)
{
	pn->clk_offset = (count - (int)(pn->first_pkt_time & 0x3f)) & 0x3f;
	if (btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
		printf("%sCLK6 = 0x%x found after %d total packets.\n", prefix, pn->clk_offset, pn->total_packets_observed);
	else
		printf("%sUAP = 0x%x found after %d total packets.\n", prefix, uap, pn->total_packets_observed);
	pn->UAP = uap;
	pn->flags |= (1u << BTBB_CLK6_VALID) | (1u << BTBB_UAP_VALID);
	pn->total_packets_observed = 0;
}

extern "C" {

int btbb_uap_from_header(btbb_packet *pkt, btbb_piconet *pn)
{
	if (!remember_hop(pn, pkt)) {
		printf("Oops. More hops than we can remember.\n");
		piconet_reset(pn);
		return 0;
	}

	CallScope scope;                              // trials and their commit share one buffer lease
	btbbx_trial trial[64];                        // one launch: trial[clock] for every CLK1-6 value
	if (packet_gpu_trials(pkt, trial)) {
		fprintf(stderr, "btbb_uap_from_header: GPU path failed: %s\n", btbbx_last_error());
		return 0;
	}
	const uint32_t rot = (pkt->clkn - pn->first_pkt_time) & 63;
	const CandidateSets s = classify_candidates(pn, trial, rot);

	// the walk covers everything below (and including) the first proven candidate
	const int winner = s.proven ? __builtin_ctzll(s.proven) : 64;
	const uint64_t walked = s.live & (below(winner) | (winner < 64 ? 1ULL << winner : 0));
	const uint64_t survivors = s.kept & below(winner);
	for (uint64_t w = walked & below(winner); w; w &= w - 1) {
		const int c = __builtin_ctzll(w);
		pn->clock6_candidates[c] = ((survivors >> c) & 1) ? (int)s.uap[c] : -1;
	}

	// the packet object ends up as the trials that really ran would have left it
	const TrialPlan plan = {walked, s.checked & walked, rot};
	if (packet_gpu_trials_commit(pkt, &plan))
		fprintf(stderr, "btbb_uap_from_header: state replay failed: %s\n", btbbx_last_error());

	if (winner < 64) {
		settle_clock6(pn, winner, s.uap[winner], "Correct CRC! ");
		return 1;
	}
	btbb_piconet_set_flag(pn, BTBB_GOT_FIRST_PACKET, 1);
	switch (__builtin_popcountll(survivors)) {
	case 1: {
		const int only = __builtin_ctzll(survivors);
		settle_clock6(pn, only, s.uap[only], "");
		return 1;
	}
	case 0:
		piconet_reset(pn);
		return 0;
	default:
		return 0;
	}
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

} // extern "C"

// ---- packet dispatch (bluetooth_piconet.c:501-543 try_hop, :851-899 btbb_process_packet) --------
//
// What happens to a packet depends only on how far the discovery of its piconet has come.  The
// stage is computed first, then one handler per stage runs; the handlers return what
// btbb_process_packet returns.
enum Stage {
	STAGE_IGNORE,        // no piconet, no LAP, or no header in the packet
	STAGE_FOLLOWING,     // UAP and full clock known: decode and print
	STAGE_WINNOWING,     // UAP and CLK1-6 known, CLK1-27 candidates are being eliminated by observed hops
	STAGE_CLK6_KNOWN,    // UAP and CLK1-6 known, hop reversal not opened
	STAGE_CONFIRM_UAP,   // a UAP is assumed, CLK1-6 not yet found with it
	STAGE_FIND_UAP,      // only the LAP is known
	STAGE_COUNT
};

static Stage stage_of(btbb_packet *pkt, const btbb_piconet *pn)
{
	if (!pn || !btbb_piconet_get_flag(pn, BTBB_LAP_VALID) || !btbb_header_present(pkt))
		return STAGE_IGNORE;
	if (btbb_piconet_get_flag(pn, BTBB_FOLLOWING))
		return STAGE_FOLLOWING;
	if (!btbb_piconet_get_uap(pn))
		return STAGE_FIND_UAP;
	if (btbb_piconet_get_flag(pn, BTBB_HOP_REVERSAL_INIT))
		return STAGE_WINNOWING;
	return btbb_piconet_get_flag(pn, BTBB_CLK6_VALID) ? STAGE_CLK6_KNOWN : STAGE_CONFIRM_UAP;
}

static int on_ignore(btbb_packet *, btbb_piconet *) { return 0; }

static int on_following(btbb_packet *pkt, btbb_piconet *pn)
{
	btbb_packet_set_uap(pkt, btbb_piconet_get_uap(pn));
	pkt->flags |= (1u << BTBB_CLK6_VALID) | (1u << BTBB_CLK27_VALID);
	if (btbb_decode(pkt))
		btbb_print_packet(pkt);
	else
		printf("Failed to decode packet\n");
	return 0;
}

static int on_find_uap(btbb_packet *pkt, btbb_piconet *pn)
{
	btbb_uap_from_header(pkt, pn);
	return 0;
}

// The three stages with an assumed UAP share a frame: decode first, let the stage use the packet,
// fall back to the assumed UAP if the stage's bookkeeping reset the piconet, and report -1 once
// both clocks are known (the caller then starts following).
static void announce_clk27(const btbb_piconet *pn)
{
	if (btbb_piconet_get_flag(pn, BTBB_CLK27_VALID))
		printf("got CLK1-27\nclock offset = %d.\n", pn->clk_offset);
}

static void stage_winnowing(btbb_packet *pkt, btbb_piconet *pn, uint8_t)
{
	const int slot = pn->packets_observed;
	if (slot < PN_MAX_PATTERN) {                  /* the reference writes past the arrays here */
		pn->pattern_indices[slot] = (int)(pkt->clkn - pn->first_pkt_time);
		pn->pattern_channels[slot] = pkt->channel;
		pn->packets_observed = slot + 1;
		pn->total_packets_observed++;
	}
	btbb_winnow(pn);
	announce_clk27(pn);
}

static void stage_clk6_known(btbb_packet *pkt, btbb_piconet *pn, uint8_t)
{
	btbb_uap_from_header(pkt, pn);
	announce_clk27(pn);
}

static void stage_confirm_uap(btbb_packet *pkt, btbb_piconet *pn, uint8_t assumed)
{
	if (!btbb_uap_from_header(pkt, pn))
		return;
	if (pn->UAP != assumed) {
		printf("failed to confirm UAP\n");
		return;
	}
	btbb_init_hop_reversal(0, pn);
	btbb_winnow(pn);
}

template <void (*STAGE)(btbb_packet *, btbb_piconet *, uint8_t)>
static int with_assumed_uap(btbb_packet *pkt, btbb_piconet *pn)
{
	const uint8_t assumed = pn->UAP;
	btbb_decode(pkt);
	STAGE(pkt, pn, assumed);
	if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID))
		btbb_piconet_set_uap(pn, assumed);
	const uint32_t both = (1u << BTBB_CLK6_VALID) | (1u << BTBB_CLK27_VALID);
	if ((pn->flags & both) != both)
		return 0;
	btbb_piconet_set_flag(pn, BTBB_FOLLOWING, 1);
	return -1;
}

typedef int (*StageHandler)(btbb_packet *, btbb_piconet *);
static const StageHandler stage_handler[STAGE_COUNT] = {
	/* STAGE_IGNORE      */ on_ignore,
	/* STAGE_FOLLOWING   */ on_following,
	/* STAGE_WINNOWING   */ with_assumed_uap<stage_winnowing>,
	/* STAGE_CLK6_KNOWN  */ with_assumed_uap<stage_clk6_known>,
	/* STAGE_CONFIRM_UAP */ with_assumed_uap<stage_confirm_uap>,
	/* STAGE_FIND_UAP    */ on_find_uap,
};

extern "C" {

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
	if (survey_mode)                              // the caller's piconet is ignored: one per LAP, kept here
		pn = get_piconet(btbb_packet_get_lap(pkt));
	if (pn)
		btbb_piconet_set_channel_seen(pn, pkt->channel);
	if (!survey_mode)
		return stage_handler[stage_of(pkt, pn)](pkt, pn);
	// a survey only ever looks for UAPs
	if (!btbb_piconet_get_flag(pn, BTBB_UAP_VALID) && btbb_header_present(pkt))
		btbb_uap_from_header(pkt, pn);
	return 0;
}

} // extern "C"
