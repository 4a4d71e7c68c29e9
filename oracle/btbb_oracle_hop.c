/*
 * oracle/btbb_oracle_hop.c -- CPU restatement of libbtbb's hop-sequence generation and
 * CLK1-27 reversal (lib/src/bluetooth_piconet.c:170-645).
 *
 * TEST INFRASTRUCTURE ONLY (see btbb_oracle.h).  Checked against the compiled reference in
 * tests/test_oracle_hop_vs_reference.py: whole 2^27-entry sequences, single_hop, candidate
 * lists after every winnowing step.
 *
 * The oracle materialises the full 128 MiB sequence exactly like the reference does; the
 * product evaluates the selection kernel per index on the GPU instead, so the two sides of
 * the parity tests do not share a strategy.
 *
 * Reference quirks kept on purpose:
 *  H1  the pattern cache is looked up with a 4-byte key (bluetooth_piconet.c:400,407: the
 *      keylen argument is 4), i.e. by (UAP << 24 | LAP) only -- AFH flag and channel count
 *      do not take part, the first pattern generated for an address wins;
 *  H2  gen_hops uses f' = (16 t mod 79) mod used_channels (:355), single_hop uses
 *      16 t mod used_channels (:439) -- they disagree under AFH;
 *  H3  pn->aliased is never written by btbb_init_hop_reversal (:475-498): the `aliased`
 *      argument only sizes the candidate array and sets BTBB_IS_ALIASED;
 *  H4  btbb_winnow's AFH test reads pattern_indices[winnowed - 1] / pattern_channels
 *      [winnowed - 1] also for winnowed == 0 (:627-628), i.e. clock6_candidates[63] and the
 *      top byte of pattern_indices[999] (struct layout bluetooth_piconet.h:76-80);
 *  H5  on `break` (:624) pn->winnowed is not advanced, so the packet that reduced the list
 *      to <= 1 is applied again by the next call.
 * Not reproduced: gen_hops evaluates f % used_channels also without AFH (:355) and dies with
 * SIGFPE for a piconet that has not seen any channel; the candidate array overrun noted at
 * orc_init_hop_reversal.
 */
#include <stdlib.h>
#include <string.h>
#include "btbb_oracle.h"

#define NCHAN 79
#define NALIAS 25

/* Bluetooth core spec vol 2 part B 2.6.2.3 (figure: permutation operation): stage s swaps
 * wires (hi nibble, lo nibble) when control bit P_s is set; stages run 13 down to 0 */
static const uint8_t butterfly[14] = {
	0x01, 0x23, 0x12, 0x34, 0x04, 0x13, 0x02, 0x34, 0x14, 0x03, 0x24, 0x13, 0x03, 0x12
};

int orc_perm5(int z, int p_high, int p_low)
{
	unsigned ctl = ((unsigned)p_high << 9) | (unsigned)p_low;
	int s;

	for (s = 13; s >= 0; s--) {
		if ((ctl >> s) & 1) {
			int u = butterfly[s] >> 4, v = butterfly[s] & 15;
			int t = ((z >> u) ^ (z >> v)) & 1;
			z ^= (t << u) | (t << v);
		}
	}
	return z;
}

/* perm_tab[ctl * 32 + z], ctl = p_high << 9 | p_low (the reference's 512 KiB fast_perm table) */
static uint8_t *perm_tab;

static void perm_tab_init(void)
{
	unsigned ctl, z;

	if (perm_tab)
		return;
	perm_tab = malloc(16384 * 32);
	for (ctl = 0; ctl < 16384; ctl++)
		for (z = 0; z < 32; z++)
			perm_tab[ctl * 32 + z] = (uint8_t)orc_perm5((int)z, (int)(ctl >> 9), (int)(ctl & 511));
}

static int channel_seen(const orc_piconet *pn, int ch)   /* :155-161 */
{
	if (ch < NCHAN)
		return (pn->afh_map[ch / 8] >> (ch % 8)) & 1;
	return 1;
}

void orc_hop_precalc(orc_piconet *pn)
{
	int i, j = 0;

	for (i = 0; i < NCHAN; i++) {
		int chan = (2 * i) % NCHAN;
		if (orc_piconet_get_flag(pn, ORC_IS_AFH)) {
			if (channel_seen(pn, chan))
				pn->bank[j++] = chan;
		} else {
			pn->bank[i] = chan;
		}
	}
}

void orc_hop_address_precalc(int address, orc_piconet *pn)
{
	int i;

	pn->a1 = (address >> 23) & 0x1f;
	pn->b = (address >> 19) & 0x0f;
	pn->d1 = (address >> 10) & 0x1ff;
	/* c1 = address bits 0,2,4,6,8; e = address bits 1,3,5,...,13 */
	pn->c1 = 0;
	for (i = 0; i < 5; i++)
		pn->c1 |= ((address >> (2 * i)) & 1) << i;
	pn->e = 0;
	for (i = 0; i < 7; i++)
		pn->e |= ((address >> (2 * i + 1)) & 1) << i;
}

void orc_gen_hops(const orc_piconet *pn, char *seq)
{
	const int afh = orc_piconet_get_flag(pn, ORC_IS_AFH);
	const unsigned used = pn->used_channels;
	uint32_t t;   /* counts groups of 64 hops = values of CLK7-27 */

	perm_tab_init();
	for (t = 0; t < (1u << 21); t++) {
		const int k = t & 0x1ff, j = (t >> 9) & 0x1f, i = (t >> 14) & 0x1f;
		const int a = pn->a1 ^ i, c = pn->c1 ^ j, d = pn->d1 ^ k;
		const unsigned f = (16u * t) % NCHAN;
		const unsigned fsel = afh ? f % used : f;
		const unsigned mod = afh ? used : NCHAN;
		const uint8_t *p0 = perm_tab + (((unsigned)c << 9 | (unsigned)d) << 5);
		const uint8_t *p1 = perm_tab + (((unsigned)(c ^ 0x1f) << 9 | (unsigned)d) << 5);
		char *out = seq + ((size_t)t << 6);
		int x;

		for (x = 0; x < 32; x++) {
			const int in = ((x + a) & 31) ^ pn->b;
			out[2 * x] = (char)pn->bank[(p0[in] + (unsigned)pn->e + fsel) % mod];
			out[2 * x + 1] = (char)pn->bank[(p1[in] + (unsigned)pn->e + fsel + 32) % mod];
		}
	}
}

/* pattern cache; quirk H1: keyed by the low 32 bits of the reference's key */
static struct pattern {
	uint32_t key;
	char *sequence;
} *patterns;
static int n_patterns;

void orc_hop_cache_clear(void)
{
	int i;

	for (i = 0; i < n_patterns; i++)
		free(patterns[i].sequence);
	free(patterns);
	patterns = NULL;
	n_patterns = 0;
}

void orc_get_hop_pattern(orc_piconet *pn)
{
	const uint32_t key = ((uint32_t)pn->UAP << 24) | pn->LAP;
	int i;

	for (i = 0; i < n_patterns; i++) {
		if (patterns[i].key == key) {
			pn->sequence = patterns[i].sequence;
			return;
		}
	}
	/* gen_hop_pattern, :365-377 */
	pn->sequence = malloc(ORC_SEQUENCE_LENGTH);
	orc_hop_precalc(pn);
	orc_hop_address_precalc((int)((((uint32_t)pn->UAP << 24) | pn->LAP) & 0xfffffff), pn);
	orc_gen_hops(pn, pn->sequence);
	patterns = realloc(patterns, (size_t)(n_patterns + 1) * sizeof(*patterns));
	patterns[n_patterns].key = key;
	patterns[n_patterns].sequence = pn->sequence;
	n_patterns++;
}

char orc_single_hop(int clock, const orc_piconet *pn)
{
	const int x = (clock >> 2) & 0x1f, y1 = (clock >> 1) & 1;
	const int a = (pn->a1 ^ (clock >> 21)) & 0x1f;
	const int c = (pn->c1 ^ (clock >> 16)) & 0x1f;
	const int d = (pn->d1 ^ (clock >> 7)) & 0x1ff;
	const uint32_t base_f = (uint32_t)(clock >> 3) & 0x1fffff0;
	const int perm = orc_perm5(((x + a) % 32) ^ pn->b, (y1 * 0x1f) ^ c, d);

	if (orc_piconet_get_flag(pn, ORC_IS_AFH))
		return (char)pn->bank[((unsigned)perm + (unsigned)pn->e + base_f % pn->used_channels + 32u * (unsigned)y1)
				      % pn->used_channels];
	return (char)pn->bank[((unsigned)perm + (unsigned)pn->e + base_f % NCHAN + 32u * (unsigned)y1) % NCHAN];
}

void orc_piconet_set_afh_map(orc_piconet *pn, const uint8_t *afh_map)
{
	int i;

	pn->used_channels = 0;
	for (i = 0; i < 10; i++) {
		pn->afh_map[i] = afh_map[i];
		pn->used_channels += (uint8_t)__builtin_popcount(afh_map[i]);
	}
	if (orc_piconet_get_flag(pn, ORC_UAP_VALID))
		orc_get_hop_pattern(pn);
}

static int observable(const orc_piconet *pn, uint32_t index)
{
	int ch = pn->sequence[index % ORC_SEQUENCE_LENGTH];

	if (pn->aliased)
		ch = ((ch + 24) % NALIAS) + 26;   /* :449-452 */
	return ch;
}

int orc_init_hop_reversal(int aliased, orc_piconet *pn)
{
	const int chan = (char)pn->pattern_channels[0];
	uint32_t i, clock;
	int count = 0;

	orc_get_hop_pattern(pn);
	/* the reference sizes this 2^27/79/32 (or /25/32) and overruns it when more candidates
	 * appear (small AFH maps); the oracle allocates the bound instead of reproducing that */
	pn->clock_candidates = malloc(sizeof(uint32_t) * (ORC_SEQUENCE_LENGTH / 64));
	clock = ((uint32_t)pn->clk_offset + pn->first_pkt_time) & 0x3f;
	for (i = clock; i < ORC_SEQUENCE_LENGTH; i += 0x40)   /* init_candidates, :455-472 */
		if (observable(pn, i) == chan)
			pn->clock_candidates[count++] = i;
	pn->num_candidates = count;
	pn->winnowed = 0;
	orc_piconet_set_flag(pn, ORC_HOP_REVERSAL_INIT, 1);
	orc_piconet_set_flag(pn, ORC_CLK27_VALID, 0);
	orc_piconet_set_flag(pn, ORC_IS_ALIASED, aliased);
	return count;
}

static int channel_winnow(int offset, int chan, orc_piconet *pn)   /* :575-611 */
{
	int i, kept = 0;

	for (i = 0; i < pn->num_candidates; i++)
		if (observable(pn, pn->clock_candidates[i] + (uint32_t)offset) == chan)
			pn->clock_candidates[kept++] = pn->clock_candidates[i];
	pn->num_candidates = kept;
	if (kept == 1) {
		pn->clk_offset = (int)((pn->clock_candidates[0] << 1) - (pn->first_pkt_time << 1));
		orc_piconet_set_flag(pn, ORC_CLK27_VALID, 1);
	} else if (kept == 0) {
		orc_piconet_reset(pn);
	}
	return kept;
}

int orc_winnow(orc_piconet *pn)
{
	int count = pn->num_candidates;

	for (; pn->winnowed < pn->packets_observed; pn->winnowed++) {
		const int w = pn->winnowed;
		const int index = pn->pattern_indices[w];
		const uint8_t channel = pn->pattern_channels[w];
		int last_index;
		uint8_t last_channel;

		count = channel_winnow(index, (char)channel, pn);
		if (count <= 1)
			break;                                            /* H5 */
		/* H4: spelled out instead of indexing below the arrays */
		last_index = w > 0 ? pn->pattern_indices[w - 1] : pn->clock6_candidates[63];
		last_channel = w > 0 ? pn->pattern_channels[w - 1]
				     : (uint8_t)((uint32_t)pn->pattern_indices[999] >> 24);
		if (!orc_piconet_get_flag(pn, ORC_LOOKS_LIKE_AFH) && index == last_index + 1 && channel == last_channel)
			orc_piconet_set_flag(pn, ORC_LOOKS_LIKE_AFH, 1);
	}
	return count;
}
