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
