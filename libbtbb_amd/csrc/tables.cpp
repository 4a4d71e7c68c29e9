// tables.cpp -- host-side derivation of every constant table from the Bluetooth
// baseband polynomials.  Replaces the transcribed tables of the reference
// (lib/src/bluetooth_packet.c:43-119, lib/src/sw_check_tables.h) with their
// definitions:
//   * (64,30) sync-word code: systematic polynomial code, generator
//     g(x) = 0260534236651 (octal, degree 34); codeword bit i <-> x^i, so the
//     syndrome of a word is its remainder mod g(x) and the syndrome of bit j is
//     x^j mod g(x)  (python/utils/gen_check_tables.py:5-59 builds the same matrix)
//   * sync word = encode((barker | LAP) ^ PN[63:34]) ^ PN  (python/utils/encode_sw.py:47-65)
//   * whitening: LFSR x^7 + x^4 + 1, register = 1 || CLK6..1, output = MSB
//   * FEC 2/3: (15,10) shortened Hamming, g(D) = D^5 + D^4 + D^2 + 1, air bit k <-> D^(14-k)
#include "common.h"

static HostTables g_tables;
static bool g_tables_ready = false;

static uint64_t encode30(const HostTables &t, uint32_t info)
{
	uint64_t cw = 0;
	for (int b = 0; b < 30; b++)
		if ((info >> b) & 1)
			cw ^= (1ULL << (34 + b)) | t.col[34 + b];
	return cw;
}

static uint64_t syncword_spec(const HostTables &t, uint32_t lap)
{
	uint32_t info = ((lap & 0x800000) ? 0x13u : 0x2cu) << 24 | (lap & 0xffffffu);
	info ^= (uint32_t)(SW_PN >> 34);
	return encode30(t, info) ^ SW_PN;
}

const HostTables &host_tables()
{
	if (g_tables_ready)
		return g_tables;
	HostTables &t = g_tables;

	uint64_t c = 1;
	for (int j = 0; j < 64; j++) {
		t.col[j] = c;
		c <<= 1;
		if (c & (1ULL << 34))
			c ^= SW_POLY;
	}
	for (int b = 0; b < 8; b++)
		for (int v = 0; v < 256; v++) {
			uint64_t s = 0;
			for (int j = 0; j < 8; j++)
				if ((v >> j) & 1)
					s ^= t.col[8 * b + j];
			t.bytetab[b][v] = s;
		}
	t.sw_default = syncword_spec(t, 0);
	for (int i = 0; i < 24; i++)
		t.gen_rows[i] = syncword_spec(t, 0x800000u >> i) ^ t.sw_default;

	// whitening m-sequence and the phase at which each CLK1-6 value starts
	{
		uint8_t state = 0x7f, phase_of[128] = {0};
		for (int i = 0; i < 127; i++) {
			uint8_t out = (state >> 6) & 1;
			phase_of[state] = (uint8_t)i;
			t.whiten[i] = out;
			state = (uint8_t)((state << 1) & 0x7f);
			if (out)
				state ^= 0x11;
		}
		for (int clk = 0; clk < 64; clk++)
			t.whiten_idx[clk] = phase_of[0x40 | clk];
	}

	for (int i = 0; i < 32; i++)
		t.fec23_fix[i] = -2;
	t.fec23_fix[0] = -1;
	for (int i = 0; i < 5; i++)
		t.fec23_fix[1 << i] = -1;
	for (int i = 0; i < 10; i++) {
		unsigned p = 1u << (14 - i), par = 0;
		for (int k = 14; k >= 5; k--)
			if ((p >> k) & 1)
				p ^= 0x35u << (k - 5);
		for (int k = 0; k < 5; k++)
			if ((p >> (4 - k)) & 1)
				par |= 1u << k;
		t.fec23_par[i] = (uint8_t)par;
		t.fec23_fix[par] = (int8_t)i;
	}
	g_tables_ready = true;
	return g_tables;
}

// btbb_gen_syncword (bluetooth_packet.c:188-199)
uint64_t host_gen_syncword(uint32_t lap)
{
	const HostTables &t = host_tables();
	uint64_t w = t.sw_default;
	for (int i = 0; i < 24; i++)
		if (lap & (0x800000u >> i))
			w ^= t.gen_rows[i];
	return w;
}

// gen_syndrome (bluetooth_packet.c:147-159)
uint64_t host_syndrome(uint64_t cw)
{
	const HostTables &t = host_tables();
	uint64_t s = 0;
	for (int b = 0; b < 8; b++)
		s ^= t.bytetab[b][(cw >> (8 * b)) & 0xff];
	return s;
}
