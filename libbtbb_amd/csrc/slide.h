// slide.h -- the sliding parity check of the LAP_ANY survivor loop, derived at compile time from the
// (64,30) code's generator polynomial (common.h SW_POLY; nothing is transcribed from the reference).
//
// Every sync word is (a multiple of g(x), degree < 64) ^ PN, bit i of the window = coefficient of x^i
// (gen_syndrome, bluetooth_packet.c:147-159, divides by the same g).  g divides x^63 + 1, so with
// h = (x^63 + 1) / g and h~ = h reversed, every coefficient 30..62 of codeword(x) * h(x) vanishes: the
// 11..17 bits of the codeword under h~ shifted to positions j .. j + 29 have even parity for j = 1..33,
// and the same holds for every multiple q = h~ * a(x) as far as it stays inside the codeword.  Bits
// 57..63 of a window are replaced by the barker correction before the syndrome is taken
// (bluetooth_packet.c:387-399), so only the checks that lie inside bits 1..56 say something about a
// window as it is in the stream: with deg q = 37 those are SLIDE_BITS = 19 shifts of ONE tap pattern.
//
// That turns "syndrome of the window at offset o" into "19 consecutive bits of a check stream":
//   c(x)   = XOR of stream[x + k] over the taps k of q, k shifted by one  (bit-sliced: 11 funnel shifts
//            and 5 three-input XORs per 32 positions, once per stream dword)
//   idx(o) = c(o) .. c(o + 18)                                           (one funnel shift per survivor)
// and idx(o) ^ K (K = the same checks over PN) is the XOR of at most max_ac_errors columns for a window
// the reference accepts -- a 2^19-bit set, the candidate bitmap of the survivor loop.  It replaces the
// two syndrome-table reads and the window extraction per survivor; the exact rule still runs on the
// candidates (scan.hip verify_lap_any).
#pragma once
#include <stdint.h>

#define SLIDE_BITS 19                         // (tables for three and four errors use SLIDE4_BITS = 20 below: one workgroup per CU)
#define SLIDE_SPAN (56 - SLIDE_BITS)          // highest tap of q: checks 1 .. SLIDE_BITS stay inside bits 1 .. 56

namespace slide {

constexpr int degree(uint64_t p) { return p ? 63 - __builtin_clzll(p) : -1; }
constexpr int weight(uint64_t p) { return __builtin_popcountll(p); }

// (x^63 + 1) / g over GF(2); the remainder must be 0 (checked by the static_assert below)
constexpr uint64_t cofactor(uint64_t g, uint64_t *rem)
{
	uint64_t a = (1ULL << 63) | 1ULL, q = 0;
	const int dg = degree(g);
	while (degree(a) >= dg) {
		const int s = degree(a) - dg;
		q |= 1ULL << s;
		a ^= g << s;
	}
	*rem = a;
	return q;
}

constexpr uint64_t reversed(uint64_t p)
{
	const int d = degree(p);
	uint64_t r = 0;
	for (int i = 0; i <= d; i++)
		if ((p >> i) & 1)
			r |= 1ULL << (d - i);
	return r;
}

constexpr uint64_t clmul(uint64_t a, uint64_t b)
{
	uint64_t r = 0;
	for (; b; b &= b - 1)
		r ^= a << __builtin_ctzll(b);
	return r;
}

// the lightest multiple of h~ with degree <= span (first one found among equals)
constexpr uint64_t lightest_check(uint64_t g, int span)
{
	uint64_t rem = 0;
	const uint64_t hr = reversed(cofactor(g, &rem));
	uint64_t best = hr;
	for (uint64_t a = 1; a < (2ULL << (span - degree(hr))); a += 2) {
		const uint64_t q = clmul(hr, a);
		if (degree(q) <= span && weight(q) < weight(best))
			best = q;
	}
	return best;
}

constexpr uint64_t remainder_of(uint64_t g)
{
	uint64_t rem = 1;
	(void)cofactor(g, &rem);
	return rem;
}

}  // namespace slide

// taps of the check stream: bit k set = stream[x + k] takes part in c(x)  (q shifted by one: check 0 of a
// codeword of length 64 does not hold, checks 1 .. 33 do)
constexpr uint64_t SLIDE_TAPS = slide::lightest_check(0260534236651ULL, SLIDE_SPAN) << 1;

// Tables for FOUR errors (scan_slide4 in scan.hip).  263 247 of the 2^19 values of the nineteen checks above are sums of at
// most four columns -- half of all survivors would pass -- and no set that fits the LDS can do much better (397 k patterns against
// 2^20 .. 2^21 bits).  So the kernel for four errors runs one workgroup per CU with a 2^20-bit set over TWENTY checks (the
// lightest multiple of degree <= 36: 31.8 % pass) and sends those through a second level in L2: twenty-four positions of a
// second, independent check stream (the lightest multiple of degree <= 32), a 2^24-bit set (2 MiB) that 2.9 % of them pass.
// The two streams together have the full rank (27) of the checks that lie inside bits 1 .. 56; the exact rule still decides.
#define SLIDE4_BITS 20
#define SLIDE4B_BITS 24
constexpr uint64_t SLIDE4_TAPS = slide::lightest_check(0260534236651ULL, 56 - SLIDE4_BITS) << 1;
constexpr uint64_t SLIDE4B_TAPS = slide::lightest_check(0260534236651ULL, 56 - SLIDE4B_BITS) << 1;
static_assert(slide::degree(SLIDE4_TAPS) + SLIDE4_BITS - 1 <= 56 && slide::degree(SLIDE4B_TAPS) + SLIDE4B_BITS - 1 <= 56,
	      "the checks must stay below the barker bits");
static_assert(SLIDE4_TAPS != SLIDE4B_TAPS, "the second level must be a different check");

static_assert(slide::remainder_of(0260534236651ULL) == 0, "the generator must divide x^63 + 1");
static_assert(slide::degree(SLIDE_TAPS) + SLIDE_BITS - 1 <= 56, "the checks must stay below the barker bits");
static_assert(slide::degree(SLIDE_TAPS) <= 64, "32 positions of the check stream must be computable from three stream dwords");
