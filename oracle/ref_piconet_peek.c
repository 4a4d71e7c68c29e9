/*
 * oracle/ref_piconet_peek.c -- TEST INFRASTRUCTURE ONLY.
 * Same idea as ref_internals.c for lib/src/bluetooth_piconet.c: include the
 * unmodified reference TU from $(REF) and export peek helpers for the piconet
 * object (bluetooth_piconet.h:32-99) so tests can compare the 64-candidate
 * state after btbb_uap_from_header (bluetooth_piconet.c:648-750).
 */
#include <string.h>
#include <stddef.h>
#include REF_PICONET_C

void refint_piconet_candidates(const btbb_piconet *pn, int *dst64)
{
	int i;
	for (i = 0; i < 64; i++)
		dst64[i] = pn->clock6_candidates[i];
}
int refint_piconet_packets_observed(const btbb_piconet *pn) { return pn->packets_observed; }
int refint_piconet_total_packets_observed(const btbb_piconet *pn) { return pn->total_packets_observed; }
uint32_t refint_piconet_first_pkt_time(const btbb_piconet *pn) { return pn->first_pkt_time; }
uint32_t refint_piconet_flags(const btbb_piconet *pn) { return pn->flags; }
void refint_survey_off(void) { survey_mode = 0; }

/* hop reversal state (bluetooth_piconet.c:311-645) */
const char *refint_piconet_sequence(const btbb_piconet *pn) { return pn->sequence; }
const uint32_t *refint_piconet_clock_candidates(const btbb_piconet *pn) { return pn->clock_candidates; }
int refint_piconet_num_candidates(const btbb_piconet *pn) { return pn->num_candidates; }
int refint_piconet_winnowed(const btbb_piconet *pn) { return pn->winnowed; }
void refint_piconet_set_aliased(btbb_piconet *pn, int aliased) { pn->aliased = aliased; }
void refint_piconet_hop_params(const btbb_piconet *pn, int *dst)   /* a1 b c1 d1 e bank[79] */
{
	int i;
	dst[0] = pn->a1; dst[1] = pn->b; dst[2] = pn->c1; dst[3] = pn->d1; dst[4] = pn->e;
	for (i = 0; i < BT_NUM_CHANNELS; i++)
		dst[5 + i] = pn->bank[i];
}
/* feed one observed hop the way btbb_uap_from_header / try_hop record it (:665-672, :510-514) */
void refint_piconet_observe(btbb_piconet *pn, int index, uint8_t channel)
{
	pn->pattern_indices[pn->packets_observed] = index;
	pn->pattern_channels[pn->packets_observed] = channel;
	pn->packets_observed++;
	pn->total_packets_observed++;
}
void refint_piconet_set_first_pkt_time(btbb_piconet *pn, uint32_t t) { pn->first_pkt_time = t; }
void refint_piconet_set_candidate6(btbb_piconet *pn, int i, int v) { pn->clock6_candidates[i] = v; }
void refint_piconet_set_pattern_index(btbb_piconet *pn, int i, int v) { pn->pattern_indices[i] = v; }
