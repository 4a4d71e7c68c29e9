// packet.hip -- the bit chain behind an access code, on packed symbols, for gfx950.
//
// Restates, one lane per (packet, clock) trial or per packet, the reference's
//   unfec13 (lib/src/bluetooth_packet.c:552)    unfec23 / fec23 (:571-649)
//   unwhiten (:653)   crcgen / payload_crc (:671, :772)   uap_from_hec (:693)
//   try_clock (:1178) crc_check (:708) fhs/DM/DH/EV3/EV4/EV5/HV (:783-1174)
//   btbb_header_present (:1371) btbb_decode_header (:1198) btbb_decode_payload (:1223)
// Symbols are bits of 64-bit words (bit i of the packet = symbol i), so FEC 1/3 is three
// masked ANDs, whitening is an XOR with a slice of the 127-bit sequence, and the CRC runs
// a byte at a time.  Identities used (all exact):
//   * crcgen over n bytes followed by its own 16 check bits leaves the register at 0, so
//     payload_crc() == (CRC over all payload_length bytes == 0) for payload_length >= 2;
//   * with payload_length == 1 payload_crc() can never succeed (the check word read from
//     in front of payload[] has bit 4 set by payload_length itself, the seed's low byte
//     is 0), which removes the llid/flow dependency of EV4;
//   * crc_check() maps every EV3/EV5 result to 1 (:760-766).
// Reference quirks kept: FEC-2/3 reads past pkt->length (DM), EV3/EV5 re-use the first
// payload byte (:1036, :1122), DV whitening restarts at 18 (:913-937), a failed FEC 1/3
// leaves UAP/type from the previous trial (:1186-1187).
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>
#include "common.h"
#include "packet_obj.h"

__constant__ ChainTables g_chain;

// Per-workgroup LDS copies of the small tables the decoders index with per-lane values inside
// their inner loops (whitening slice, CRC byte / word step, FEC 2/3 parity and correction): a DS read
// instead of a divergent constant-memory load or a 10-step loop.  4.6 KiB, copied by
// chain_lds_init() at kernel start from the image chain_upload() built on the host.
struct __attribute__((aligned(16))) ChainLds {
	uint32_t wh32[128];        // 32 whitening bits from phase idx (idx <= 126)
	uint16_t crc[256];         // crc_byte(0, x): one byte through the reflected CRC-CCITT register
	uint16_t crc_z[3][256];    // the same followed by 1, 2, 3 zero bytes (slicing: four bytes per step)
	uint8_t  par23[1024];      // FEC 2/3 parity of 10 data bits
	int8_t   fix23[32];
	uint16_t fixm23[32];       // the same as a mask: the data bit to flip, 0x8000 = undecodable (long_payloads)
	uint8_t  whiten_idx[64];
	uint16_t adv32[2][256];    // the CRC register 32 zero bytes later, by its low / high byte (linear: XOR the two)
};
// the image every workgroup copies (built once on the host in chain_upload)
__device__ __attribute__((aligned(16))) ChainLds g_chain_lds_image;

// Table of trials_linear_kernel.  The CRC register is GF(2)-linear in seed, data and whitening, the seed has
// eight free bits (the UAP, bits 8..15) and the whitening sequence from any phase is the GF(2) combination of
// seven basis sequences selected by its own first seven bits (a 7-stage LFSR).  Row L (payload length in
// bytes) holds what each of those fifteen bits contributes to the register after L bytes:
//   [0..7]  register after L zero bytes from the seed with only bit 8 + b set
//   [8..14] register after the first L bytes of the whitening sequence whose first seven bits are unit vector j
//   [15]    0
#define LIN_MAXLEN 344                      // payload lengths 0 .. 343 (DH5)
__device__ __attribute__((aligned(16))) uint16_t g_lin[LIN_MAXLEN * 16];
// g_advw[i - 1][h][x]: the CRC register 4 i zero bytes after holding x in its low (h = 0) / high (h = 1) byte,
// i = 1 .. 7 (linear: XOR the two halves) -- what carries a chunk's start register to a word inside the chunk
__device__ __attribute__((aligned(16))) uint16_t g_advw[7 * 2 * 256];
// g_adv64inv[j][k]: the CRC register that holds 1 << k after 64 j zero bits -- the matrix A^(-64 j) by columns (A = one
// zero bit through the register: invertible), what lane j of the long-payload phase of decode_hits_kernel applies to the
// register of payload word j alone
__device__ __attribute__((aligned(16))) uint16_t g_adv64inv[64 * 16];
// g_adv64fwd[j][k]: the register 64 j zero bits AFTER holding 1 << k -- A^(+64 j): carries the XOR the lanes of an EV4 / EV5
// payload have accumulated in front of word j back into the true register in front of that word
__device__ __attribute__((aligned(16))) uint16_t g_adv64fwd[64 * 16];

static uint32_t host_crc_byte(uint32_t crc, uint32_t byte)
{
	uint32_t x = (crc ^ byte) & 0xff;
	x ^= (x << 4) & 0xff;
	return ((crc >> 8) ^ (x << 8) ^ (x << 3) ^ (x >> 4)) & 0xffff;
}

int chain_upload(const HostTables &t)
{
	// trials_linear_kernel puts the trials of a batch into type order by arithmetic (its step 1b), which rests on one property of
	// the whitening sequence: the four bits that whiten the header's type field (header bits 3 .. 6) take every value for
	// exactly four of the 64 CLK1-6 candidates.  Checked here, once, on the table the kernels are built from.
	{
		int seen[16] = {0};
		for (int clk = 0; clk < 64; clk++) {
			uint32_t v = 0;
			for (int j = 0; j < 4; j++)
				v |= (uint32_t)t.whiten[(t.whiten_idx[clk] + 3 + j) % 127] << j;
			seen[v]++;
		}
		for (int v = 0; v < 16; v++)
			if (seen[v] != 4) {
				set_error("btbbx_init: internal: the type-field whitening is not four clocks per value");
				return BTBBX_E_ARG;
			}
	}
	// the LDS image of the decoders
	{
		static ChainLds img;
		memset(&img, 0, sizeof(img));
		for (int i = 0; i < 128; i++)
			for (int j = 0; j < 32; j++)
				img.wh32[i] |= (uint32_t)t.whiten[((i < 127 ? i : 0) + j) % 127] << j;
		for (int i = 0; i < 256; i++) {
			uint32_t c = host_crc_byte(0, (uint32_t)i);
			img.crc[i] = (uint16_t)c;
			for (int k = 0; k < 3; k++) {
				c = host_crc_byte(c, 0);
				img.crc_z[k][i] = (uint16_t)c;
			}
		}
		for (int i = 0; i < 1024; i++) {
			uint32_t par = 0;
			for (int bit = 0; bit < 10; bit++)
				if ((i >> bit) & 1)
					par ^= t.fec23_par[bit];
			img.par23[i] = (uint8_t)par;
		}
		memcpy(img.fix23, t.fec23_fix, 32);
		for (int i = 0; i < 32; i++)
			img.fixm23[i] = t.fec23_fix[i] == -2 ? 0x8000u : t.fec23_fix[i] >= 0 ? (uint16_t)(1u << t.fec23_fix[i]) : 0u;
		memcpy(img.whiten_idx, t.whiten_idx, 64);
		for (int h = 0; h < 2; h++)
			for (int i = 0; i < 256; i++) {
				uint32_t c = (uint32_t)i << (8 * h);
				for (int k = 0; k < 32; k++)
					c = host_crc_byte(c, 0);
				img.adv32[h][i] = (uint16_t)c;
			}
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_lds_image), &img, sizeof(img)));
	}
	// the rows of trials_linear_kernel
	{
		static uint16_t lin[LIN_MAXLEN * 16];
		memset(lin, 0, sizeof(lin));
		for (int bit = 0; bit < 8; bit++) {
			uint32_t crc = 1u << (8 + bit);
			for (int L = 0; L < LIN_MAXLEN; L++) {
				lin[L * 16 + bit] = (uint16_t)crc;
				crc = host_crc_byte(crc, 0);
			}
		}
		// the recurrence of the whitening sequence: w[k + 7] = XOR of the w[k + j] picked by `taps` (found, not assumed)
		uint32_t taps = 0;
		for (uint32_t m = 1; m < 128 && !taps; m++) {
			bool ok = true;
			for (int k = 0; k < 127 && ok; k++) {
				uint32_t x = 0;
				for (int j = 0; j < 7; j++)
					if ((m >> j) & 1)
						x ^= t.whiten[(k + j) % 127];
				ok = x == t.whiten[(k + 7) % 127];
			}
			if (ok)
				taps = m;
		}
		if (!taps) {
			set_error("chain_upload: the whitening sequence is not a 7-stage LFSR sequence");
			return BTBBX_E_ARG;
		}
		for (int e = 0; e < 7; e++) {
			static uint8_t seq[8 * LIN_MAXLEN + 8];
			for (int k = 0; k < 7; k++)
				seq[k] = k == e;
			for (int k = 7; k < 8 * LIN_MAXLEN; k++) {
				uint32_t x = 0;
				for (int j = 0; j < 7; j++)
					if ((taps >> j) & 1)
						x ^= seq[k - 7 + j];
				seq[k] = (uint8_t)x;
			}
			uint32_t crc = 0;
			for (int L = 1; L < LIN_MAXLEN; L++) {
				uint32_t byte = 0;
				for (int j = 0; j < 8; j++)
					byte |= (uint32_t)seq[8 * (L - 1) + j] << j;
				crc = host_crc_byte(crc, byte);
				lin[L * 16 + 8 + e] = (uint16_t)crc;
			}
		}
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_lin), lin, sizeof(lin)));
		static uint16_t advw[7 * 2 * 256];
		for (int i = 1; i <= 7; i++)
			for (int h = 0; h < 2; h++)
				for (int x = 0; x < 256; x++) {
					uint32_t c = (uint32_t)x << (8 * h);
					for (int k = 0; k < 4 * i; k++)
						c = host_crc_byte(c, 0);
					advw[((i - 1) * 2 + h) * 256 + x] = (uint16_t)c;
				}
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_advw), advw, sizeof(advw)));
		// A^(-64): the register u that holds 1 << k 64 zero bits later, found by running every u forward (nothing assumed
		// about the polynomial beyond "the step is invertible", which the search checks)
		static uint16_t adv64inv[64 * 16];
		{
			static uint16_t back[65536];
			static uint8_t hit[65536];
			memset(hit, 0, sizeof(hit));
			for (uint32_t u = 0; u < 65536; u++) {
				uint32_t c = u;
				for (int b = 0; b < 8; b++)
					c = host_crc_byte(c, 0);
				back[c] = (uint16_t)u;
				hit[c] = 1;
			}
			for (uint32_t v = 0; v < 65536; v++)
				if (!hit[v]) {
					set_error("chain_upload: the CRC register's zero-byte step is not invertible");
					return BTBBX_E_ARG;
				}
			for (int k = 0; k < 16; k++) {
				uint32_t c = 1u << k;
				for (int j = 0; j < 64; j++) {
					adv64inv[j * 16 + k] = (uint16_t)c;
					c = back[c];
				}
			}
		}
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_adv64inv), adv64inv, sizeof(adv64inv)));
		static uint16_t adv64fwd[64 * 16];
		for (int k = 0; k < 16; k++) {
			uint32_t c = 1u << k;
			for (int j = 0; j < 64; j++) {
				adv64fwd[j * 16 + k] = (uint16_t)c;
				for (int b = 0; b < 8; b++)
					c = host_crc_byte(c, 0);
			}
		}
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_adv64fwd), adv64fwd, sizeof(adv64fwd)));
	}
	ChainTables c;
	memset(&c, 0, sizeof(c));
	for (int j = 0; j < 256; j++)
		if (t.whiten[j % 127])
			c.whiten2[j >> 6] |= 1ULL << (j & 63);
	memcpy(c.whiten_idx, t.whiten_idx, 64);
	memcpy(c.fec23_par, t.fec23_par, 10);
	memcpy(c.fec23_fix, t.fec23_fix, 32);
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_chain), &c, sizeof(c)));
	return BTBBX_OK;
}

#define F_WHITENED    (1u << 0)
#define F_CLK6_VALID  (1u << 4)
#define F_HAS_PAYLOAD (1u << 7)
#define DHL_MIN_BITS  256              // payloads longer than this leave decode_hits_kernel's lanes for its wave phase (= 64 DH_OUT_WORDS)

// ---- bit helpers ----------------------------------------------------------------------------

// n (1..64) bits of the packet starting at symbol pos; pos + n <= 3200
__device__ __forceinline__ uint64_t pk_bits(const uint64_t *w, uint32_t pos, uint32_t n)
{
	uint32_t i = pos >> 6, s = pos & 63;
	uint64_t v = w[i] >> s;
	if (s + n > 64)
		v |= w[i + 1] << (64 - s);
	return n == 64 ? v : v & ((1ULL << n) - 1);
}

// the same for n <= 32 through two dword reads and one funnel shift (the 64-bit form costs two 8-byte reads, two
// 64-bit shifts and a 64-bit mask); pos + n <= 3200, so dword (pos >> 5) + 1 is still inside the 50-word row
__device__ __forceinline__ uint32_t pk_bits32(const uint64_t *w, uint32_t pos, uint32_t n)
{
	const uint32_t *d = reinterpret_cast<const uint32_t *>(w);
	const uint32_t i = pos >> 5;
	const uint32_t v = __builtin_amdgcn_alignbit(d[i + 1], d[i], pos & 31);
	return n == 32 ? v : v & ((1u << n) - 1);
}

// n (1..64) whitening bits starting at phase idx (0..126), straight from constant memory (used
// where a kernel needs a handful of them; the decoders below use the LDS copies)
__device__ __forceinline__ uint64_t wh_bits_const(uint32_t idx, uint32_t n)
{
	uint32_t i = idx >> 6, s = idx & 63;
	uint64_t v = g_chain.whiten2[i] >> s;
	if (s + n > 64)
		v |= g_chain.whiten2[i + 1] << (64 - s);
	return n == 64 ? v : v & ((1ULL << n) - 1);
}

__device__ __forceinline__ uint32_t wh_start_const(uint32_t clock, uint32_t skip)
{
	return (g_chain.whiten_idx[clock & 63] + skip) % 127u;
}

__shared__ ChainLds g_lds;

__device__ __forceinline__ uint64_t wh_bits(uint32_t idx, uint32_t n)
{
	uint64_t v = g_lds.wh32[idx];
	if (n > 32) {
		const uint32_t j = idx + 32;
		v |= (uint64_t)g_lds.wh32[j >= 127 ? j - 127 : j] << 32;
	}
	return n == 64 ? v : v & ((1ULL << n) - 1);
}

__device__ __forceinline__ uint32_t wh_start(uint32_t clock, uint32_t skip)
{
	return (g_lds.whiten_idx[clock & 63] + skip) % 127u;
}

__device__ __forceinline__ uint32_t rev8(uint32_t b) { return __brev(b) >> 24; }

// one byte through the reflected CRC-CCITT register of crcgen (:681-687)
__device__ __forceinline__ uint32_t crc_byte_calc(uint32_t crc, uint32_t byte)
{
	uint32_t x = (crc ^ byte) & 0xff;
	x ^= (x << 4) & 0xff;
	return ((crc >> 8) ^ (x << 8) ^ (x << 3) ^ (x >> 4)) & 0xffff;
}
// the same through the LDS table (the register update is linear: crc' = crc >> 8 ^ T[(crc ^ byte) & 0xff])
__device__ __forceinline__ uint32_t crc_byte(uint32_t crc, uint32_t byte)
{
	return (crc >> 8) ^ g_lds.crc[(crc ^ byte) & 0xff];
}

// Four bytes per step.  The register update is linear over GF(2) and the register is 16 bits wide, so
// after the bytes b0..b3 (b0 first) it holds
//     Z3[(crc ^ b0) & 0xff] ^ Z2[(crc >> 8) ^ b1] ^ Z1[b2] ^ Z0[b3],   Zk[x] = byte x followed by k zero bytes:
// four INDEPENDENT table reads instead of a chain of four dependent ones (the CRC over the 187 / 343 bytes
// of a DH3 / DH5 trial is what the brute force spends its time in, and it was bound by that latency).
__device__ __forceinline__ uint32_t crc_word(uint32_t crc, uint32_t w)
{
	const uint32_t x = crc ^ w;
	return g_lds.crc_z[2][x & 0xff] ^ g_lds.crc_z[1][(x >> 8) & 0xff] ^ g_lds.crc_z[0][(w >> 16) & 0xff] ^ g_lds.crc[w >> 24];
}

__device__ __forceinline__ uint32_t crc_seed(uint32_t uap) { return rev8(uap & 0xff) << 8; }

// uap_from_hec (:693-705)
__device__ __forceinline__ uint32_t uap_from_hec(uint32_t data, uint32_t hec)
{
#pragma unroll
	for (int i = 9; i >= 0; i--) {
		if (hec & 0x80)
			hec ^= 0x65;
		hec = ((hec << 1) | (((hec >> 7) ^ (data >> i)) & 1)) & 0xff;
	}
	return rev8(hec);
}

// FEC 1/3 of n <= 21 triples held in the low 3n bits of v: majority bits (compacted) and
// the number of triples that disagree (:552-568)
__device__ __forceinline__ uint32_t fec13(uint64_t v, uint32_t n, uint32_t &disagree)
{
	const uint64_t M = 0x9249249249249249ULL;          // every third bit
	uint64_t a = v & M, b = (v >> 1) & M, c = (v >> 2) & M;
	uint64_t maj = (a & b) | (b & c) | (c & a);
	uint64_t dis = (a ^ b) | (b ^ c) | (c ^ a);
	if (n < 21) {
		uint64_t keep = (1ULL << (3 * n)) - 1;
		maj &= keep;
		dis &= keep;
	}
	disagree = __popcll(dis);
	// bit 3 i -> bit i: pairs, nibbles, bytes, ... close ranks (five shift / or / and steps instead of n)
	uint64_t x = maj;
	x = (x | x >> 2) & 0x30c30c30c30c30c3ULL;
	x = (x | x >> 4) & 0xf00f00f00f00f00fULL;
	x = (x | x >> 8) & 0x00ff0000ff0000ffULL;
	x = (x | x >> 16) & 0xffff00000000ffffULL;
	x = (x | x >> 32) & 0xffffffffULL;
	return (uint32_t)x;
}

// one (15,10) block: 15 symbols in -> 10 corrected data bits, false if uncorrectable (:602-646)
__device__ __forceinline__ bool fec23_block(uint32_t blk, uint32_t &data)
{
	data = blk & 0x3ff;
	uint32_t diff = (blk >> 10) ^ g_lds.par23[data];
	int fix = g_lds.fix23[diff & 31];
	if (fix == -2)
		return false;
	if (fix >= 0)
		data ^= 1u << fix;
	return true;
}

// all threads of the workgroup; ends with a barrier
__device__ void chain_lds_init()
{
	static_assert(sizeof(ChainLds) % 16 == 0, "the image is copied 16 bytes at a time");
	const uint4 *src = reinterpret_cast<const uint4 *>(&g_chain_lds_image);
	uint4 *dst = reinterpret_cast<uint4 *>(&g_lds);
	for (uint32_t i = threadIdx.x; i < sizeof(ChainLds) / 16; i += blockDim.x)
		dst[i] = src[i];
	__syncthreads();
}

// ---- packet state -----------------------------------------------------------------------------

// Where a decoder's payload words go: HBM, or (decode_hits_kernel, small packets) the wave's LDS.  Two address spaces
// behind one generic pointer would make every access a FLAT instruction, which counts on both vmcnt and lgkmcnt: each
// table lookup behind a payload store then waits for the store to come back from the memory pipeline.
struct OutRef {
	uint64_t *g = nullptr;
	uint32_t l = 0;          // LDS byte address + 1, or 0
	__device__ OutRef() {}
	__device__ OutRef(uint64_t *p) : g(p) {}
	__device__ __forceinline__ static OutRef lds(uint32_t byte_address) { OutRef r; r.l = byte_address + 1; return r; }
	__device__ __forceinline__ uint64_t ld(uint32_t k) const
	{
		if (l)
			return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t *>(l - 1 + 8u * k);
		return g[k];
	}
	__device__ __forceinline__ void st(uint32_t k, uint64_t v) const
	{
		if (l)
			*reinterpret_cast<__attribute__((address_space(3))) uint64_t *>(l - 1 + 8u * k) = v;
		else
			g[k] = v;
	}
};
struct PState {
	const uint64_t *w;       // 50 packed words (a gathered packet), or -- `direct` -- the stream word the packet starts in
	// direct mode (decode_hits_kernel): the packet is bits [sh, sh + length) of w[0 .. wlimit)
	uint32_t sh = 0;         // bit of w[0] the packet starts at
	uint32_t wlimit = 0;     // stream words that exist from w on
	uint32_t staged = 0;     // direct mode: w[0 .. staged) have a copy in LDS at LDS byte address `stage_off`
	uint32_t stage_off = 0;
	bool direct = false;
	// the FEC 1/3 decoded header and its disagreeing triples, when the caller has them already (decode_hits_kernel)
	bool has_pre = false;
	uint32_t pre_hdr = 0, pre_dis = 0;
	int length;              // pkt->length
	uint32_t flags;
	uint32_t uap, type;
	uint32_t lt_addr, hdr_flags, hec, header18;
	int plen;                // payload_length
	int phl;                 // payload_header_length
	uint32_t ph16;           // payload_header bits
	uint32_t ph_written;     // how many payload_header chars were written
	uint32_t llid, flow;
	// payload writer
	OutRef out;              // 43 words or nothing
	bool spoiled = false;    // out is the caller's scratch copy (LDS) and holds a payload the reference would not have written
	// decode_hits_kernel: a DM / DH payload of more than 256 bits is not walked by its lane; do_DM / do_DH return after
	// their checks with the bit count here and the wave works it off afterwards, a group of lanes per packet (long_payloads)
	// (long_wave, at the end of decode_hits_kernel); the sixteen bytes that phase needs to know go straight into def_slot from here
	uint4 *def_slot = nullptr;   // where (null: every payload is walked by its lane)
	uint32_t def_pkt8 = 0;       // the record's index in its workgroup's 256
	uint32_t def_nbits = 0;      // != 0: deferred
	uint32_t written;        // payload bits written (prefix)
	// which fields a trial assigned (replay_kernel merges 64 trials by "last writer wins")
	uint32_t dirty;          // D_* bits
	uint32_t ph_mask;        // payload-header bits assigned
};
#define D_UT    1u           // uap, type        (try_clock)
#define D_PLEN  2u           // payload_length
#define D_PHL   4u           // payload_header_length
#define D_LF    8u           // llid, flow

// n (1..64) symbols of the packet from symbol pos.  A gathered packet is 50 words with zeros behind the captured
// length; in direct mode the same view is taken of the stream itself: symbols at and behind `length` (which the
// reference's decoders do read, :898-958) and words behind the end of the stream read as 0.
__device__ __forceinline__ uint64_t s_bits(const PState &s, uint32_t pos, uint32_t n)
{
	if (!s.direct)
		return pk_bits(s.w, pos, n);
	if ((int)pos >= s.length)
		return 0;
	const uint32_t q = pos + s.sh, i = q >> 6, sft = q & 63;
	// words the wave staged through LDS (decode_hits_kernel) come from there, anything behind them from the stream
	auto word = [&](uint32_t k) -> uint64_t {
		if (k < s.staged)
			return *reinterpret_cast<const __attribute__((address_space(3))) uint64_t *>(s.stage_off + 8u * k);
		return k < s.wlimit ? ((const __attribute__((address_space(1))) uint64_t *)(uintptr_t)s.w)[k] : 0ULL;    // (direct mode: the stream in HBM)
	};
	uint64_t v = word(i) >> sft;
	if (sft + n > 64)
		v |= word(i + 1) << (64 - sft);
	const uint32_t have = (uint32_t)s.length - pos;         // symbols left in front of `length`
	const uint32_t keep = n < have ? n : have;
	return keep == 64 ? v : v & ((1ULL << keep) - 1);
}

// streams payload bits into the CRC (whole bytes) and, when WRITE, into the output words
template <bool WRITE>
struct Sink {
	uint64_t acc = 0;
	uint32_t nacc = 0;
	uint32_t crc;
	uint64_t oacc = 0;
	uint32_t onacc = 0, oword = 0;
	OutRef out;
	__device__ Sink(uint32_t seed, OutRef o) : crc(seed), out(o) {}
	__device__ __forceinline__ void push(uint64_t bits, uint32_t n)   // n <= 32
	{
		if (n == 32 && nacc == 0) {                                   // whole word on a byte boundary
			crc = crc_word(crc, (uint32_t)bits);
		} else {
			acc |= bits << nacc;
			nacc += n;
			while (nacc >= 8) {
				crc = crc_byte(crc, (uint32_t)acc & 0xff);
				acc >>= 8;
				nacc -= 8;
			}
		}
		if (WRITE) {
			oacc |= bits << onacc;
			onacc += n;
			if (onacc >= 64) {
				out.st(oword++, oacc);
				onacc -= 64;
				oacc = onacc ? bits >> (n - onacc) : 0;
			}
		}
	}
	// merge the unfinished word with what the output already holds
	__device__ __forceinline__ void flush()
	{
		if (WRITE && onacc) {
			uint64_t keep = ~0ULL << onacc;
			out.st(oword, (out.ld(oword) & keep) | oacc);
		}
	}
};

__device__ __forceinline__ bool whitened(const PState &s) { return s.flags & F_WHITENED; }

__device__ __forceinline__ uint64_t wh(const PState &s, uint32_t idx, uint32_t n)
{
	return whitened(s) ? wh_bits(idx, n) : 0ULL;
}

// Four consecutive (15,10) blocks from symbol `pos` on, their symbols and table reads issued together (a lane that decodes
// block after block waits for four dependent LDS round trips per block; decode_hits_kernel is bound by exactly those waits).
// ok bit j = block j decodes; blocks behind `count` are not looked at.
__device__ __forceinline__ uint32_t fec23_blocks4(const PState &s, uint32_t pos, uint32_t count, uint32_t (&data)[4])
{
	uint32_t blk[4], diff[4];
	const uint64_t sym60 = s_bits(s, pos, 60);             // (symbols behind `length` read as 0 either way)
#pragma unroll
	for (int j = 0; j < 4; j++)
		blk[j] = (uint32_t)j < count ? (uint32_t)(sym60 >> (15 * j)) & 0x7fff : 0;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		data[j] = blk[j] & 0x3ff;
		diff[j] = (blk[j] >> 10) ^ g_lds.par23[data[j]];
	}
	uint32_t ok = 0;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int fix = g_lds.fix23[diff[j] & 31];
		if (fix >= 0)
			data[j] ^= 1u << fix;
		if (fix != -2)
			ok |= 1u << j;
	}
	return ok;
}

// all FEC-2/3 blocks of `nblocks` decodable?
__device__ __forceinline__ bool fec23_ok(const PState &s, uint32_t pos, uint32_t nblocks)
{
	for (uint32_t k = 0; k < nblocks; k += 4) {
		uint32_t d[4];
		const uint32_t cnt = nblocks - k < 4 ? nblocks - k : 4;
		if ((fec23_blocks4(s, pos + 15 * k, cnt, d) & ((1u << cnt) - 1)) != (1u << cnt) - 1)
			return false;
	}
	return true;
}

// The payload of s is left to the lane-group phase (long_payloads): what that phase needs, sixteen bytes.
//   a: address of the stream word the packet starts in | stream words to load << 48 | bit the packet starts at << 55
//   b: record index | captured length << 8 | bits << 20 | kind << 32 | whitened << 34 | whitening phase of the payload's
//      first bit << 35 | UAP << 42
// kind / bits: DHL_DH, DHL_DM: payload_length * 8; DHL_EV4: ten per block its loop may look at (min(98, size / 15));
// DHL_EV5: eight per byte its loop may write (min(182, size / 8))
#define DHL_DH  0u
#define DHL_DM  1u
#define DHL_EV4 2u
#define DHL_EV5 3u
__device__ __forceinline__ void defer_payload(PState &s, uint32_t clock, uint32_t nbits, uint32_t kind)
{
	// stream words the decoder looks at: 122 symbols of access code and header, then the payload -- FEC 2/3 blocks may
	// lie behind the captured length (they read as zeros), never behind word 45; EV5 reads one byte; nothing
	// behind the stream's end is loaded
	const uint32_t ext = kind == DHL_DH ? nbits : kind == DHL_EV5 ? 8u : 15u * ((nbits + 9u) / 10u);
	const uint32_t nw = (s.sh + 122u + ext + 63u) >> 6;
	const uint64_t a = (uint64_t)(uintptr_t)s.w | (uint64_t)(nw < s.wlimit ? nw : s.wlimit) << 48 | (uint64_t)s.sh << 55;
	const uint64_t b = (uint64_t)s.def_pkt8 | (uint64_t)(uint32_t)s.length << 8 | (uint64_t)nbits << 20 | (uint64_t)kind << 32
		| (uint64_t)((s.flags & 1u) ? 1u : 0u) << 34 | (uint64_t)wh_start(clock, 18) << 35 | (uint64_t)(s.uap & 0xffu) << 42;
	*s.def_slot = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
	s.def_nbits = nbits;
}

// fhs (:783-818)
// The same for a caller that only looks at the register at the end (DM): bytes go through the CRC four at a time -- one
// step of four independent table reads instead of four dependent ones -- and finish() takes the last one to three.
template <bool WRITE>
struct SinkW : Sink<WRITE> {
	__device__ SinkW(uint32_t seed, OutRef o) : Sink<WRITE>(seed, o) {}
	__device__ __forceinline__ void push(uint64_t bits, uint32_t n)   // n <= 32
	{
		this->acc |= bits << this->nacc;
		this->nacc += n;
		if (this->nacc >= 32) {
			this->crc = crc_word(this->crc, (uint32_t)this->acc);
			this->acc >>= 32;
			this->nacc -= 32;
		}
		if (WRITE) {
			this->oacc |= bits << this->onacc;
			this->onacc += n;
			if (this->onacc >= 64) {
				this->out.st(this->oword++, this->oacc);
				this->onacc -= 64;
				this->oacc = this->onacc ? bits >> (n - this->onacc) : 0;
			}
		}
	}
	__device__ __forceinline__ void finish()
	{
		while (this->nacc >= 8) {
			this->crc = crc_byte(this->crc, (uint32_t)this->acc & 0xff);
			this->acc >>= 8;
			this->nacc -= 8;
		}
	}
};
template <bool WRITE>
__device__ __forceinline__ int do_fhs(PState &s, uint32_t clock)
{
	int size = s.length - 122;
	s.plen = 20;
	s.dirty |= D_PLEN;
	if (size < 240)
		return 1;
	uint64_t corr[3] = {0, 0, 0};
#pragma unroll
	for (uint32_t k = 0; k < 16; k += 4) {
		uint32_t d4[4];
		const uint32_t ok = fec23_blocks4(s, 122 + 15 * k, 4, d4);
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			if (!((ok >> j) & 1))
				return 0;
			const uint32_t d = d4[j], bit = 10 * (k + j);
			corr[bit >> 6] |= (uint64_t)d << (bit & 63);
			if ((bit & 63) > 54)
				corr[(bit >> 6) + 1] |= (uint64_t)d >> (64 - (bit & 63));
		}
	}
	int rv = 0;
	uint32_t c = clock;
	for (int attempt = 0; attempt < 33; attempt++) {
		if (attempt)
			c = 31 + attempt;
		uint32_t idx = wh_start(c, 18);
		uint32_t crc = crc_seed(s.uap);
		uint64_t pl[3];
		for (int i = 0; i < 3; i++) {
			uint32_t n = i < 2 ? 64 : 32;
			pl[i] = corr[i] ^ wh(s, idx, n);
			idx = (idx + n) % 127u;
			for (uint32_t b = 0; b < n; b += 32)                  // 20 bytes = five words
				crc = crc_word(crc, (uint32_t)(pl[i] >> b));
		}
		if (WRITE) {
			s.out.st(0, pl[0]);
			s.out.st(1, pl[1]);
			s.out.st(2, (s.out.ld(2) & ~0xffffffffULL) | pl[2]);
			if (s.written < 160) s.written = 160;
		}
		if (crc == 0) {
			rv = 1000;
			break;
		}
	}
	return rv;
}

// decode_payload_header (:821-895)
template <bool WRITE>
__device__ __forceinline__ bool do_payload_header(PState &s, uint32_t pos, uint32_t clock, int header_bytes, int size, bool fec)
{
	uint32_t hbits = header_bytes == 2 ? 16 : 8;
	if (size < (int)hbits)
		return false;
	uint32_t raw;
	if (fec) {
		if (size < (header_bytes == 2 ? 30 : 15))
			return false;
		uint32_t d0, d1 = 0;
		if (!fec23_block((uint32_t)s_bits(s, pos, 15), d0))
			return false;
		if (header_bytes == 2 && !fec23_block((uint32_t)s_bits(s, pos + 15, 15), d1))
			return false;
		raw = (d0 | (d1 << 10)) & ((1u << hbits) - 1);
	} else {
		raw = (uint32_t)s_bits(s, pos, hbits);
	}
	uint32_t ph = raw ^ (uint32_t)wh(s, wh_start(clock, 18), hbits);
	s.ph16 = (s.ph16 & ~((1u << hbits) - 1)) | ph;
	s.ph_mask |= (1u << hbits) - 1;
	s.dirty |= D_PLEN | D_LF | D_PHL;
	if (s.ph_written < hbits) s.ph_written = hbits;
	int plen = header_bytes == 2 ? (int)((s.ph16 >> 3) & 0x3ff) + 4 : (int)((s.ph16 >> 3) & 0x1f) + 3;
	int cap;
	switch (s.type) {
	case 3:  cap = 20;  break;
	case 4:  cap = 30;  break;
	case 8:  cap = 12;  break;
	case 10: cap = 125; break;
	case 11: cap = 187; break;
	case 14: cap = 228; break;
	case 15: cap = 343; break;
	default: cap = 0;   break;
	}
	s.plen = plen < cap ? plen : cap;
	s.llid = s.ph16 & 3;
	s.flow = (s.ph16 >> 2) & 1;
	s.phl = header_bytes;
	return true;
}

// DM (:898-958)
template <bool WRITE>
__device__ __forceinline__ int do_DM(PState &s, uint32_t clock)
{
	uint32_t pos = 122;
	int size = s.length - 122;
	int header_bytes = 2, max_length;
	switch (s.type) {
	case 8:  pos += 80; size -= 80; header_bytes = 1; max_length = 12; break;
	case 3:  header_bytes = 1; max_length = 20; break;
	case 10: max_length = 125; break;
	case 14: max_length = 228; break;
	default: return 0;
	}
	if (!do_payload_header<WRITE>(s, pos, clock, header_bytes, size, true))
		return 0;
	if (s.plen > max_length)
		return 1;
	int nbits = s.plen * 8;
	if (nbits > size)
		return 1;
	uint32_t nblocks = (nbits + 9) / 10;
	if (WRITE && s.def_slot && !s.out.l && nbits > DHL_MIN_BITS) {
		defer_payload(s, clock, (uint32_t)nbits, DHL_DM);
		return 2;                                           // (replaced by the lane-group phase's verdict)
	}
	// The reference writes nothing when a block fails.  Into HBM that takes a pass over all blocks first; a scratch copy
	// is written as the blocks decode and marked as not to be kept when one fails.
	if (WRITE && !s.out.l && !fec23_ok(s, pos, nblocks))
		return 0;
	SinkW<WRITE> sink(crc_seed(s.uap), s.out);
	uint32_t idx = wh_start(clock, 18);
	int left = nbits;
	for (uint32_t k = 0; k < nblocks; k += 4) {             // four blocks = 40 data bits per step
		uint32_t d[4];
		const uint32_t cnt = nblocks - k < 4 ? nblocks - k : 4;
		const uint32_t ok = fec23_blocks4(s, pos + 15 * k, cnt, d);
		const uint64_t w40 = wh(s, idx, 40);
		idx = idx + 40 >= 127 ? idx + 40 - 127 : idx + 40;
#pragma unroll
		for (uint32_t j = 0; j < 4; j++) {
			if (j >= cnt)
				break;
			if (!((ok >> j) & 1)) {
				if (WRITE)
					s.spoiled = true;
				return 0;
			}
			const uint32_t n = left < 10 ? left : 10;
			sink.push((d[j] ^ (uint32_t)(w40 >> (10 * j))) & ((1u << n) - 1), n);
			left -= n;
		}
	}
	sink.finish();
	sink.flush();
	if (WRITE && s.written < (uint32_t)nbits) s.written = nbits;
	return sink.crc == 0 ? 10 : 2;
}

// DH (:962-1011)
template <bool WRITE>
__device__ __forceinline__ int do_DH(PState &s, uint32_t clock)
{
	const uint32_t pos = 122;
	int size = s.length - 122;
	int header_bytes = 2, max_length;
	switch (s.type) {
	case 9:
	case 4:  header_bytes = 1; max_length = 30; break;
	case 11: max_length = 187; break;
	case 15: max_length = 343; break;
	default: return 0;
	}
	if (!do_payload_header<WRITE>(s, pos, clock, header_bytes, size, false))
		return 0;
	if (s.plen > max_length)
		return 1;
	int nbits = s.plen * 8;
	if (nbits > size)
		return 1;
	if (WRITE && s.def_slot && !s.out.l && nbits > DHL_MIN_BITS) {
		defer_payload(s, clock, (uint32_t)nbits, DHL_DH);
		return 2;                                           // (replaced by the lane-group phase's verdict)
	}
	Sink<WRITE> sink(crc_seed(s.uap), s.out);
	uint32_t idx = wh_start(clock, 18);
	for (int done = 0; done < nbits; done += 32) {
		uint32_t n = nbits - done < 32 ? nbits - done : 32;
		sink.push(s_bits(s, pos + done, n) ^ wh(s, idx, n), n);
		idx = (idx + n) % 127u;
	}
	sink.flush();
	if (WRITE && s.written < (uint32_t)nbits) s.written = nbits;
	if (s.type == 9)
		return 2;
	return sink.crc == 0 ? 10 : 2;
}

// EV3 (:1013-1042) / EV5 (:1099-1128)
template <bool WRITE>
__device__ __forceinline__ int do_EV35(PState &s, uint32_t clock, int maxlength)
{
	int size = s.length - 122;
	uint32_t first8 = (uint32_t)s_bits(s, 122, 8);
	Sink<WRITE> sink(crc_seed(s.uap), s.out);
	uint32_t idx = wh_start(clock, 18);
	int rv = 2;
	int L;
	for (L = 0; L < maxlength; L++) {
		if (8 * L + 8 > size) {
			rv = 1;
			break;
		}
		// the reference writes byte L, then tests the CRC over bytes 0..L-1
		uint32_t byte = first8 ^ (uint32_t)wh(s, idx, 8);
		idx = (idx + 8) % 127u;
		bool match = L > 2 && sink.crc == 0;     // CRC over bytes 0..L-1 == 0
		sink.push(byte, 8);
		if (WRITE && s.written < (uint32_t)(8 * L + 8)) s.written = 8 * L + 8;
		if (match) {
			rv = 10;
			break;
		}
	}
	sink.flush();
	s.plen = L;
	s.dirty |= D_PLEN;
	return rv;
}

// EV4 (:1044-1097)
template <bool WRITE>
__device__ __forceinline__ int do_EV4(PState &s, uint32_t clock)
{
	int size = s.length - 122;
	uint32_t crc = crc_seed(s.uap);
	uint64_t acc = 0;           // payload bits produced but not yet consumed by the CRC
	uint32_t nacc = 0;
	uint64_t oacc = 0;
	uint32_t onacc = 0, oword = 0;
	int L = 1;
	int rv = 2;
	for (int b = 0; b < 98; b++) {
		int syms = 15 * b, bits = 10 * b;
		if (syms + 15 > size) { rv = 1; break; }
		uint32_t d;
		if (!fec23_block((uint32_t)s_bits(s, 122 + syms, 15), d)) { rv = syms < 45 ? 0 : 1; break; }
		uint64_t ten = d ^ (uint32_t)wh(s, wh_start(clock, 18 + bits), 10);
		acc |= ten << nacc;
		nacc += 10;
		if (WRITE) {
			oacc |= ten << onacc;
			onacc += 10;
			if (onacc >= 64) {
				s.out.st(oword++, oacc);
				onacc -= 64;
				oacc = onacc ? ten >> (10 - onacc) : 0;
			}
			if (s.written < (uint32_t)(bits + 10)) s.written = bits + 10;
		}
		bool hit = false;
		while (L * 8 <= bits) {
			crc = crc_byte(crc, (uint32_t)acc & 0xff);      // byte L-1
			acc >>= 8;
			nacc -= 8;
			if (L >= 2 && crc == 0) { hit = true; break; }
			L++;
		}
		if (hit) { rv = 10; break; }
	}
	if (WRITE && onacc) {
		uint64_t keep = ~0ULL << onacc;
		s.out.st(oword, (s.out.ld(oword) & keep) | oacc);
	}
	s.plen = L;
	s.dirty |= D_PLEN;
	return rv;
}

// HV (:1131-1174)
template <bool WRITE>
__device__ __forceinline__ int do_HV(PState &s, uint32_t clock)
{
	int size = s.length - 122;
	s.phl = 0;
	s.dirty |= D_PHL;
	if (size < 240) {
		s.plen = 0;
		s.dirty |= D_PLEN;
		return 1;
	}
	uint32_t idx = wh_start(clock, 18);
	if (s.type == 5) {
		uint32_t data[4], total = 0;
		for (int i = 0; i < 4; i++) {       // 80 triples = 4 x 20
			uint32_t dis;
			data[i] = fec13(s_bits(s, 122 + 60 * i, 60), 20, dis);
			total += dis;
		}
		if (!(total < 20))
			return 0;
		s.plen = 10;
		s.dirty |= D_PLEN;
		s.flags |= F_HAS_PAYLOAD;
		if (WRITE) {
			Sink<true> sink(0, s.out);
			for (int i = 0; i < 4; i++) {
				sink.push(data[i] ^ (uint32_t)wh(s, idx, 20), 20);
				idx = (idx + 20) % 127u;
			}
			sink.flush();
			if (s.written < 80) s.written = 80;
		}
	} else if (s.type == 6) {
		if (!fec23_ok(s, 122, 16))
			return 0;
		s.plen = 20;
		s.dirty |= D_PLEN;
		s.flags |= F_HAS_PAYLOAD;
		if (WRITE) {
			Sink<true> sink(0, s.out);
			for (uint32_t k = 0; k < 16; k++) {
				uint32_t d;
				fec23_block((uint32_t)s_bits(s, 122 + 15 * k, 15), d);
				sink.push(d ^ (uint32_t)wh(s, idx, 10), 10);
				idx = (idx + 10) % 127u;
			}
			sink.flush();
			if (s.written < 160) s.written = 160;
		}
	} else if (s.type == 7) {
		s.plen = 30;
		s.dirty |= D_PLEN;
		s.flags |= F_HAS_PAYLOAD;
		if (WRITE) {
			Sink<true> sink(0, s.out);
			for (int done = 0; done < 240; done += 30) {
				sink.push(s_bits(s, 122 + done, 30) ^ wh(s, idx, 30), 30);
				idx = (idx + 30) % 127u;
			}
			sink.flush();
			if (s.written < 240) s.written = 240;
		}
	}
	return 2;
}

// crc_check (:708-769)
template <bool WRITE>
__device__ int do_crc_check(PState &s, uint32_t clock)
{
	int rv = 1;
	switch (s.type) {
	case 2:  rv = do_fhs<WRITE>(s, clock); break;
	case 8: case 3: case 10: case 14: rv = do_DM<WRITE>(s, clock); break;
	case 4: case 11: case 15: rv = do_DH<WRITE>(s, clock); break;
	case 7:  rv = WRITE ? do_EV35<WRITE>(s, clock, 32) : 1; break;     // always mapped to 1 below
	case 12: rv = do_EV4<WRITE>(s, clock); break;
	case 13: rv = WRITE ? do_EV35<WRITE>(s, clock, 182) : 1; break;
	case 5:  rv = do_HV<WRITE>(s, clock); break;
	default: break;
	}
	if (rv == 0 && s.type != 2 && s.type != 3 && s.type != 5)
		return 1;
	if (rv > 1 && (s.type == 7 || s.type == 13))
		return 1;
	return rv;
}

// FEC-1/3 decoded header bits and the number of disagreeing triples
__device__ __forceinline__ uint32_t header_fec13(const uint64_t *w, uint32_t &disagree)
{
	return fec13(pk_bits(w, 68, 54), 18, disagree);
}
__device__ __forceinline__ uint32_t header_fec13(const PState &s, uint32_t &disagree)
{
	return fec13(s_bits(s, 68, 54), 18, disagree);
}

// try_clock (:1178-1195); returns the reference's return value
__device__ __forceinline__ uint32_t do_try_clock(PState &s, uint32_t clock, uint32_t hdr, uint32_t disagree)
{
	if (!(disagree < 4))
		return 0;
	uint32_t clear = hdr ^ (uint32_t)wh(s, wh_start(clock, 0), 18);
	s.uap = uap_from_hec(clear & 0x3ff, clear >> 10);
	s.type = (clear >> 3) & 0xf;
	s.dirty |= D_UT;
	return s.uap;
}

// btbb_header_present (:1371-1408)
// (dis: the disagreeing triples of the header, header_fec13)
__device__ __forceinline__ int do_header_present(const PState &s, uint32_t dis)
{
	if (s.length < 122)
		return 0;
	const uint32_t five = (uint32_t)s_bits(s, 63, 5);
	uint32_t msb = five & 1;
	uint32_t tr = five >> 1;
	uint32_t want = msb ? 0xAu : 0x5u;          // !m, m, !m, m  (LSB first)
	uint32_t errs = __popc(tr ^ want);
	return (errs + dis) < 5;
}
__device__ __forceinline__ int do_header_present(const PState &s)
{
	uint32_t dis;
	(void)fec13(s_bits(s, 68, 54), 18, dis);
	return do_header_present(s, dis);
}

// ---- kernels --------------------------------------------------------------------------------

// The throughput shape for large batches (BASELINE config 5: 10^6 detected packets).  What a trial's crc_check
// spends its time on when it is run as written (do_crc_check above) -- FEC 2/3 over up to 183 blocks, whitening, a CRC over up to 343 bytes
// -- does not depend on the clock candidate except through two XORs:
//   * FEC 2/3 is undone BEFORE whitening (:898-958), so the decoded bits, and which block fails first, are
//     properties of the packet;
//   * the CRC register is GF(2)-linear:  reg(seed, data ^ whitening, L bytes)
//         = A^L(seed)  ^  reg(0, data, L)  ^  reg(0, whitening, L),
//     the first from g_il (eight 16-bit terms selected by the UAP), the last from g_pw.
// So a workgroup first works out, once per packet, the decoded bytes of the two FEC 2/3 layouts (payload at
// 122, DV data at 202), their first failing block, the HV1 verdict, and reg(0, data, 4i) for every fourth
// byte count of the three data layouts (raw, FEC at 122, FEC at 202) -- and a DM / DH / FHS trial is then a
// payload header, a length, a handful of table reads and a compare.  EV4 (which scans for the first byte
// count whose CRC is zero) still walks bytes, but bytes that are already decoded.
// a word of LDS as it is now (another lane of the wave may just have changed it): a volatile read through a generic
// pointer is a FLAT load, which waits on both memory counters -- and with it on every prefetched word still in flight
__device__ __forceinline__ uint32_t lds_now(const uint32_t *p)
{
	return *(volatile __attribute__((address_space(3))) const uint32_t *)p;
}

#define TL_THREADS 512                  // two workgroups per CU, 32 packets per batch, 74 KiB of LDS each (one 1024-thread workgroup,
                                        // 64 packets, 126 KiB: 0.930 against 0.921-0.926 ms per 2^20 packets, profiles/r05_trials)
#define TL_PACKETS (TL_THREADS / 16)
#define TL_WGS_PER_CU (1024 / TL_THREADS)
#define TL_TRIALS  (TL_PACKETS * 64)
#define TL_A_BLOCKS 183                     // DM5: 228 bytes = 1824 bits
#define TL_A_BYTES  232                     // >= 229, multiple of 4
#define TL_B_BLOCKS 10                      // DV: 12 bytes
#define TL_B_BYTES  16
#ifdef TL_PROFILE
__device__ unsigned long long g_tl_prof[16];
// per-workgroup counters in LDS (global atomics here would sit in the same in-order queue as the loads the
// kernel waits for, and the profile would show their latency instead of the kernel's)
#define TL_PROF_START __shared__ uint32_t tl_acc[16]; if (threadIdx.x < 16) tl_acc[threadIdx.x] = 0; uint64_t tl_t = __builtin_readcyclecounter()
#define TL_PROF(k) do { const uint64_t n_ = __builtin_readcyclecounter(); if (tid == 0) tl_acc[k] += (uint32_t)(n_ - tl_t); tl_t = n_; } while (0)
#define TL_PROF_END do { __syncthreads(); if (tid < 16) atomicAdd(&g_tl_prof[tid], (unsigned long long)tl_acc[tid]); } while (0)
#else
#define TL_PROF_START do { } while (0)
#define TL_PROF(k) do { } while (0)
#define TL_PROF_END do { } while (0)
#endif
#define TL_AFAIL(p) lds_now(&a_fail[p])
__global__ __launch_bounds__(TL_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void trials_linear_kernel(const uint64_t *packets, const btbbx_pkt_in *in,
							     uint32_t n_packets, btbbx_trial *trials)
{
	__shared__ uint64_t pk[TL_PACKETS][BTBBX_PKT_WORDS + 1];
	__shared__ __attribute__((aligned(8))) btbbx_pkt_in pin[TL_PACKETS];
	__shared__ uint32_t hdr_ut[TL_PACKETS];
	__shared__ uint16_t clk_ut[64];
	__shared__ __attribute__((aligned(16))) uint16_t pw20[64];   // register after the 20 whitening bytes of an FHS attempt
	__shared__ __attribute__((aligned(16))) uint16_t lin[LIN_MAXLEN * 16];
	__shared__ __attribute__((aligned(16))) uint16_t advw[7 * 2 * 256];
	__shared__ uint16_t order[TL_TRIALS];
	__shared__ uint32_t t_info[TL_TRIALS];            // per trial: try_clock's return value | type << 8 | UAP << 16
	__shared__ int16_t t_rv[TL_TRIALS];
	__shared__ uint32_t pk_sort[TL_PACKETS];          // per packet: type key | varies with the clock << 4 | rank among its like << 8
	__shared__ uint32_t type_base[18];                // first trial slot of every type; [17] = packets whose type varies with the clock
	// per packet
	// (a10 is dead behind step 2a, the barrier that follows it separates it from step 2c: its rows are then the packet's p4a and
	// p4c rows.  93 dwords per row: odd, so the rows of consecutive packets start in different banks.)
	__shared__ uint16_t a10[TL_PACKETS][TL_A_BLOCKS + 3];       // decoded 10-bit groups, payload at 122
	static_assert(TL_A_BYTES / 4 + LIN_MAXLEN / 4 <= TL_A_BLOCKS + 3, "p4a and p4c of a packet lie in its a10 row");
	__shared__ uint16_t b10[TL_PACKETS][TL_B_BLOCKS + 2];       // ... DV data at 202
	__shared__ uint32_t a_bytes[TL_PACKETS][TL_A_BYTES / 4], b_bytes[TL_PACKETS][TL_B_BYTES / 4];
	__shared__ uint32_t a_fail[TL_PACKETS], b_fail[TL_PACKETS]; // first undecodable block
	__shared__ uint16_t p4b[TL_PACKETS][TL_B_BYTES / 4];
	auto p4a = [&](uint32_t p) { return &a10[p][0]; };
	auto p4c = [&](uint32_t p) { return &a10[p][TL_A_BYTES / 4]; };
	__shared__ int8_t hv_rv[TL_PACKETS];
	__shared__ uint16_t chunk_reg[TL_PACKETS][20];
	const uint32_t tid = threadIdx.x, lane = tid & 63;
	TL_PROF_START;
	// tables once per workgroup (the workgroups are persistent: each takes every gridDim.x-th batch)
	for (uint32_t i = tid; i < LIN_MAXLEN * 2; i += TL_THREADS)
		reinterpret_cast<uint4 *>(lin)[i] = reinterpret_cast<const uint4 *>(g_lin)[i];
	for (uint32_t i = tid; i < 7 * 2 * 256 / 8; i += TL_THREADS)
		reinterpret_cast<uint4 *>(advw)[i] = reinterpret_cast<const uint4 *>(g_advw)[i];
	chain_lds_init();                                       // ends with a barrier
	if (tid >= 64 && tid < 128) {
		const uint32_t wb = (uint32_t)wh_bits(wh_start(tid - 64, 0), 18);
		// bits 12, 13: the clock's rank among the FOUR clocks that whiten the type field alike (the 64 clocks map onto the
		// sixteen 4-bit values four times each -- an affine map of full rank; tables.cpp checks it on the host): with that,
		// where a trial stands in type order is arithmetic (step 1b below)
		const uint32_t wt = (wb >> 3) & 0xf;
		uint32_t crank = 0;
		for (uint32_t k = 0; k < 16; k++) {
			const uint64_t mk = __ballot(wt == k);
			if (wt == k)
				crank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0));
		}
		clk_ut[tid - 64] = (uint16_t)(uap_from_hec(wb & 0x3ff, wb >> 10) | (wt << 8) | (crank << 12));
		const uint32_t v = (uint32_t)wh_bits(wh_start(tid - 64, 18), 7);
		uint32_t x = 0;
		for (int j = 0; j < 7; j++)
			if ((v >> j) & 1)
				x ^= lin[20 * 16 + 8 + j];
		pw20[tid - 64] = (uint16_t)x;
	}
	const uint32_t n_batches = (n_packets + TL_PACKETS - 1) / TL_PACKETS;
	// a batch is 16 x 51 packet words + 16 x 2 words of btbbx_pkt_in, four per thread; the next batch's words fly
	// while this one is worked on (all of them the same kind of guarded load: anything the compiler has to merge
	// with an old value, or may re-issue at its use, ends up waited for right behind the prefetch)
	constexpr uint32_t PK_ELEMS = TL_PACKETS * (BTBBX_PKT_WORDS + 1), IN_ELEMS = TL_PACKETS * sizeof(btbbx_pkt_in) / 8;
	constexpr uint32_t PER_THREAD = (PK_ELEMS + IN_ELEMS + TL_THREADS - 1) / TL_THREADS;
	static_assert(sizeof(btbbx_pkt_in) == 16, "two words per btbbx_pkt_in");
	uint64_t pre[PER_THREAD];
	// every load unconditional, from a clamped address (validity is applied when the words go to LDS)
	// (round 4: the element -> (packet, word) arithmetic of the two lambdas is done where it is used -- `tv` is opaque to the
	// compiler --: hoisted out of the batch loop it is a dozen registers that live in scratch, and a reload from scratch counts
	// on the same in-order counter as the prefetched words)
	auto fetch = [&](uint32_t b) {
		const uint32_t f = b * TL_PACKETS, have = b < n_batches ? (n_packets - f < TL_PACKETS ? n_packets - f : TL_PACKETS) : 0;
		uint32_t tv = tid;
		asm volatile("" : "+v"(tv));
#pragma unroll
		for (uint32_t k = 0; k < PER_THREAD; k++) {
			const uint32_t i = tv + TL_THREADS * k;
			const uint32_t p = i / (BTBBX_PKT_WORDS + 1), w = i % (BTBBX_PKT_WORDS + 1), e = i - PK_ELEMS;
			const uint64_t *src = packets;
			if (i < PK_ELEMS) {
				if (p < have && w < BTBBX_PKT_WORDS)
					src = packets + (uint64_t)(f + p) * BTBBX_PKT_WORDS + w;
			} else if (e < 2 * have) {
				src = reinterpret_cast<const uint64_t *>(in) + (uint64_t)f * 2 + e;
			}
			pre[k] = *src;
		}
	};
	// the prefetched words of batch b go to LDS (and the per-batch counters are reset)
	auto stage_in = [&](uint32_t b) {
		const uint32_t f = b * TL_PACKETS, have = b < n_batches ? (n_packets - f < TL_PACKETS ? n_packets - f : TL_PACKETS) : 0;
		uint32_t tv = tid;
		asm volatile("" : "+v"(tv));
#pragma unroll
		for (uint32_t k = 0; k < PER_THREAD; k++) {
			const uint32_t i = tv + TL_THREADS * k;
			const uint32_t p = i / (BTBBX_PKT_WORDS + 1), w = i % (BTBBX_PKT_WORDS + 1);
			if (i < PK_ELEMS)
				pk[p][w] = (p < have && w < BTBBX_PKT_WORDS) ? pre[k] : 0;
			else if (i < PK_ELEMS + IN_ELEMS)
				reinterpret_cast<uint64_t *>(pin)[i - PK_ELEMS] = pre[k];
		}
		if (tid < TL_PACKETS) {
			a_fail[tid] = TL_A_BLOCKS;
			b_fail[tid] = TL_B_BLOCKS;
		}
	};
	fetch(blockIdx.x);
	stage_in(blockIdx.x);
	fetch(blockIdx.x + gridDim.x);
	for (uint32_t batch = blockIdx.x; batch < n_batches; batch += gridDim.x) {
	const uint32_t first = batch * TL_PACKETS;
	const uint32_t mine = n_packets - first < TL_PACKETS ? n_packets - first : TL_PACKETS;

	__syncthreads();                                        // this batch is in LDS, the previous batch's results are read
	TL_PROF(0);

	// 1. try_clock (:1178-1195).  uap_from_hec (:693-705) and the type field are GF(2)-linear in the 18 header
	// bits and unwhitening XORs a clock-dependent constant onto them, so try_clock(c) = U(header) ^ U(whitening
	// bits of c): one U per packet here, one per clock in clk_ut (as in uap_table_kernel), one XOR per trial below.
	// Wrong candidate clocks turn the 4 type bits into noise, so the 64 trials of a packet spread over all
	// sixteen decoders: the trial numbers are counting-sorted by type (LDS atomics) and step 3 walks them in
	// that order, a wave's 64 consecutive entries being trials of ONE type except at the few type boundaries.
	// (Trials are independent of each other -- the one cross-trial dependency of the reference, EV4 reading
	// the llid / flow a previous trial left, cannot change a result, see do_EV4 -- so their order is free.)
	// 1b. (round 5) The trial numbers in type order WITHOUT a sort over the 4096 trials.  The type of trial (packet, clock)
	// is the packet's four raw type bits XOR four whitening bits that depend on the clock alone, and every 4-bit value is the
	// whitening of exactly four clocks: a packet whose type varies with the clock (whitened, header FEC 1/3 decodable) puts
	// exactly FOUR trials into EVERY type.  So type t starts at slot 4 nvar t + 64 x (packets of fixed type < t), the trial of
	// varying packet number r and clock c is slot base[type] + 4 r + (rank of c among its four), and the 64 trials of a
	// fixed-type packet (not whitened, or FEC 1/3 failed: SURVEY Q5) lie together behind them.  One wave ranks the packets of a batch;
	// rounds 2-4 counted every trial into its type with an LDS atomic (sixteen counters, 4096 atomics per batch) and scattered
	// the trial numbers in a second pass behind a scan of the counters.
	static_assert(TL_PACKETS <= 64, "one wave ranks the packets of a batch");
	if (tid < 64) {
		const bool live = tid < mine;
		uint32_t h = 0;
		if (live) {
			uint32_t dis;
			const uint32_t hdr = header_fec13(pk[tid], dis);
			h = uap_from_hec(hdr & 0x3ff, hdr >> 10) | (((hdr >> 3) & 0xf) << 8) | ((dis < 4 ? 1u : 0u) << 16);
			hdr_ut[tid] = h;
		}
		const bool fec_ok = (h & 0x10000u) != 0;
		const bool var = live && fec_ok && (pin[live ? tid : 0].flags & F_WHITENED);
		const uint32_t key = !live ? 0u : fec_ok ? (h >> 8) & 0xfu : (uint32_t)pin[tid].type & 0xfu;
		const uint64_t vm = __ballot(var);
		const uint32_t nvar = (uint32_t)__popcll(vm);
		uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(vm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vm, 0));
		const bool fixed = live && !var;
		uint32_t cf = 0;                                    // lane k: packets of fixed type k
		if (__ballot(fixed)) {
			for (uint32_t k = 0; k < 16; k++) {
				const uint64_t fm = __ballot(fixed && key == k);
				if (fixed && key == k)
					rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0));
				if (lane == k)
					cf = (uint32_t)__popcll(fm);
			}
		}
		uint32_t incl = cf;
#pragma unroll
		for (int dd = 1; dd < 16; dd <<= 1) {
			const uint32_t up = __shfl_up(incl, dd);
			if (lane >= (uint32_t)dd)
				incl += up;
		}
		if (lane < 16)
			type_base[lane] = 4u * nvar * lane + 64u * (incl - cf);
		if (lane == 17)
			type_base[17] = nvar;
		if (tid < TL_PACKETS)
			pk_sort[tid] = key | ((var ? 1u : 0u) << 4) | (rank << 8);
	}
	// 2a. FEC 2/3 of both layouts: sixteen threads per packet (one quarter of a wave), sixteen blocks of the
	// payload layout per round, and no further round once a block of the packet has failed -- nothing behind
	// the first undecodable block can matter to any trial, and in the noise behind a short packet half of
	// all blocks fail.  HV1 verdict (:1131-1150).
	static_assert(TL_THREADS == 16 * TL_PACKETS, "sixteen threads per packet");
	// 32-bit words of the payload layout that lie in front of the packet's first undecodable block (+ the one it starts in)
	auto a_words = [&](uint32_t p) {
		const uint32_t blocks = TL_AFAIL(p) < TL_A_BLOCKS ? TL_AFAIL(p) : TL_A_BLOCKS, n = (blocks * 10 + 31) / 32 + 1;
		return n < TL_A_BYTES / 4 ? n : (uint32_t)(TL_A_BYTES / 4);
	};
	{
		const uint32_t p = tid >> 4, sub = tid & 15;
		if (p < mine) {
			uint32_t d;
			if (sub < TL_B_BLOCKS) {
				if (!fec23_block(pk_bits32(pk[p], 202 + 15 * sub, 15), d))
					atomicMin(&b_fail[p], sub);
				b10[p][sub] = (uint16_t)d;
			}
#pragma unroll 1
			for (uint32_t k0 = 0; k0 < TL_A_BLOCKS; k0 += 16) {
				const uint32_t k = k0 + sub;
				if (k < TL_A_BLOCKS) {
					if (!fec23_block(pk_bits32(pk[p], 122 + 15 * k, 15), d))
						atomicMin(&a_fail[p], k);
					a10[p][k] = (uint16_t)d;
				}
				// the sixteen lanes are in one wave and a wave's LDS operations complete in order: its
				// atomics above are done when this read is served
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				if (TL_AFAIL(p) < k0 + 16)
					break;
			}
			// the decoded bits as bytes, four per thread and step, as far as they decode: by the same sixteen
			// lanes, which wrote every 10-bit group these words are made of (same wave: in order)
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			auto word_from = [&](const uint16_t *src, uint32_t nblk, uint32_t word) {
				const uint32_t bit = 32 * word, k0 = bit / 10, sh = bit % 10;    // bits 32 word .. 32 word + 31 of the groups
				uint64_t acc = 0;
				for (uint32_t j = 0; j < 5; j++)
					acc |= (uint64_t)(k0 + j < nblk ? src[k0 + j] : 0) << (10 * j);
				return (uint32_t)(acc >> sh);
			};
			if (sub < TL_B_BYTES / 4)
				b_bytes[p][sub] = word_from(b10[p], TL_B_BLOCKS, sub);
			const uint32_t need = a_words(p);
			for (uint32_t word = sub; word < need; word += 16)
				a_bytes[p][word] = word_from(a10[p], TL_A_BLOCKS, word);
		}
	}
	if (tid >= 128 && tid < 128 + mine) {
		const uint32_t p = tid - 128;
		int rv = 1;
		if ((int)pin[p].length - 122 >= 240) {
			uint32_t total = 0;
			for (int i = 0; i < 4; i++) {
				uint32_t dis;
				(void)fec13(pk_bits(pk[p], 122 + 60 * i, 60), 20, dis);
				total += dis;
			}
			rv = total < 20 ? 2 : 0;
		}
		hv_rv[p] = (int8_t)rv;
	}
	__syncthreads();
	TL_PROF(1);
	const uint32_t total = mine * 64;
	for (uint32_t i = tid; i < total; i += TL_THREADS) {
		const uint32_t p = i >> 6;
		const uint32_t h = hdr_ut[p];
		uint32_t uap = pin[p].uap, type = pin[p].type, ret = 0;     // FEC 1/3 failure: nothing changes
		const uint32_t ps = pk_sort[p], cu = clk_ut[lane];
		if (h & 0x10000u) {
			const uint32_t v = (h ^ ((pin[p].flags & F_WHITENED) ? (cu & 0xfffu) : 0u)) & 0xffff;
			uap = ret = v & 0xff;
			type = v >> 8;
		}
		t_info[i] = ret | (type << 8) | (uap << 16);
		// where this trial stands in type order (step 1b)
		const uint32_t slot = (ps & 0x10u) ? type_base[type & 15] + 4u * (ps >> 8) + (cu >> 12)
						   : type_base[type & 15] + 4u * type_base[17] + 64u * (ps >> 8) + lane;
		order[slot] = (uint16_t)i;
	}
	// 2c. reg(0, data, 4 i) for the three layouts (raw: 86 words, FEC at 122: 58, FEC at 202: 4), in chunks of
	// eight words = twenty chunks per packet: (i) every chunk from 0, all in parallel, storing the register in front
	// of each of its words; (ii) per layout the chunk starts, start' = adv32(start) ^ chunk (a dozen dependent
	// steps instead of 86).  reg(0, data, 4 q) is then (register stored for word q) ^ (the chunk's start carried
	// 4 (q mod 8) bytes forward, two reads of advw) -- put together by the trial that asks for it.
	// task t -> (packet, chunk slot r, layout, chunk j, words): raw and DV chunks first (12 per packet), then the
	// payload-layout chunks, of which a packet needs only those in front of its first failing block
	auto chunk_of = [&](uint32_t t, uint32_t &p, uint32_t &r, uint32_t &layout, uint32_t &j, uint32_t &nwords) {
		if (t < mine * 12) {
			p = t / 12;
			j = t % 12;
			if (j < 11) { layout = 0; r = j; nwords = j < 10 ? 8 : LIN_MAXLEN / 4 - 80; }
			else { layout = 2; r = 19; j = 0; nwords = TL_B_BYTES / 4; }
		} else {
			// chunk-major: chunk 0 of every packet, then chunk 1 of every packet, ... -- a packet needs only the chunks in front of its
			// first failing block, so the tasks that have work lie at the front of the list and the second round of TL_THREADS
			// tasks (640 tasks on 512 threads) is empty for everything but full-length DM3 / DM5 payloads (round 6; packet-major
			// left a quarter of the threads a full chunk each in that round: 0.924-0.931 against 0.873 ms per 2^20 packets, all types
			// 0.862 against 0.823 -- profiles/r06_trials; the EV4 scan, taken out as a probe, is 2 % of the batch)
			const uint32_t u = t - mine * 12;
			j = mine == TL_PACKETS ? u / TL_PACKETS : u / mine;
			p = u - j * mine;
			layout = 1;
			r = 11 + j;
			const uint32_t need = a_words(p);
			nwords = need > 8 * j ? (need - 8 * j < 8 ? need - 8 * j : 8) : 0;
		}
	};
	auto data_word = [&](uint32_t p, uint32_t layout, uint32_t i) {
		return layout == 0 ? pk_bits32(pk[p], 122 + 32 * i, 32) : layout == 1 ? a_bytes[p][i] : b_bytes[p][i];
	};
	// the (up to) eight words of a chunk, all loaded before any is used: one LDS round trip, not eight.  The raw
	// payload starts at symbol 122 = dword 3, bit 26 of the packet row, so its words are funnel shifts by 26 of
	// nine consecutive dwords
	auto chunk_words = [&](uint32_t p, uint32_t layout, uint32_t j, uint32_t nwords, uint32_t (&w8)[8]) {
		if (layout == 0) {
			const uint32_t *d = reinterpret_cast<const uint32_t *>(pk[p]) + 3 + 8 * j;
			uint32_t raw[9];
#pragma unroll
			for (uint32_t i = 0; i < 9; i++)
				raw[i] = i <= nwords ? d[i] : 0;
#pragma unroll
			for (uint32_t i = 0; i < 8; i++)
				w8[i] = __builtin_amdgcn_alignbit(raw[i + 1], raw[i], 26);
		} else {
#pragma unroll
			for (uint32_t i = 0; i < 8; i++)
				w8[i] = i < nwords ? data_word(p, layout, 8 * j + i) : 0;
		}
	};
	for (uint32_t t = tid; t < mine * 20; t += TL_THREADS) {
		uint32_t p, r, layout, j, nwords, crc = 0;
		chunk_of(t, p, r, layout, j, nwords);
		if (!nwords)
			continue;
		uint32_t w8[8];
		chunk_words(p, layout, j, nwords, w8);
		uint16_t *dst = layout == 0 ? p4c(p) : layout == 1 ? p4a(p) : p4b[p];
#pragma unroll
		for (uint32_t i = 0; i < 8; i++)
			if (i < nwords) {
				dst[8 * j + i] = (uint16_t)crc;                // the register in front of word i, from 0 at the chunk start
				crc = crc_word(crc, w8[i]);
			}
		chunk_reg[p][r] = (uint16_t)crc;
	}
	__syncthreads();
	TL_PROF(8);
	if (tid >= 64 && tid < 64 + 3 * mine) {
		const uint32_t p = (tid - 64) / 3, layout = (tid - 64) % 3;
		const uint32_t r0 = layout == 0 ? 0 : layout == 1 ? 11 : 19;
		const uint32_t n = layout == 0 ? 11 : layout == 1 ? (a_words(p) + 7) / 8 : 1;
		uint32_t start = 0;
		for (uint32_t j = 0; j < n; j++) {
			const uint32_t c = chunk_reg[p][r0 + j];
			chunk_reg[p][r0 + j] = (uint16_t)start;
			start = g_lds.adv32[0][start & 0xff] ^ g_lds.adv32[1][start >> 8] ^ c;
		}
	}
	__syncthreads();
	TL_PROF(4);

	// 3. crc_check (:708-769) in type order
	for (uint32_t kk = tid; kk < total; kk += TL_THREADS) {
		const uint32_t i = order[kk], p = i >> 6, clock = i & 63;
		const uint32_t info = t_info[i], type = (info >> 8) & 0xff, uap = (info >> 16) & 0xff;
		const bool wht = pin[p].flags & F_WHITENED;
		const int size = (int)pin[p].length - 122;
		const uint32_t seed = crc_seed(uap);
		// what seed and whitening contribute to the register after L bytes: the terms of row L that the
		// seed's eight bits and the seven first whitening bits of this clock select
		const uint32_t sel = (seed >> 8) | (wht ? (uint32_t)wh_bits(wh_start(clock, 18), 7) << 8 : 0u);
		auto seed_row20 = [&](uint32_t sd) {                   // the seed's terms of row 20 alone (FHS tries other clocks)
			const uint4 r0 = reinterpret_cast<const uint4 *>(lin)[40];
			const uint32_t r[4] = {r0.x, r0.y, r0.z, r0.w};
			uint32_t x = 0;
#pragma unroll
			for (int bit = 0; bit < 8; bit++)
				x ^= (0u - ((sd >> (8 + bit)) & 1)) & (r[bit >> 1] >> (16 * (bit & 1)));
			return x & 0xffff;
		};
		auto lin_terms = [&](uint32_t L) {
			uint32_t x = 0;
#pragma unroll
			for (int half = 0; half < 2; half++) {             // (both 16-byte halves of the row asked for together: 115 VGPRs; one at a time -- rounds 4-6a --
			                                                   //  was 1-2 % slower: 0.861 / 0.817 against 0.853 / 0.795 ms, profiles/r06_trials)
				const uint4 q = reinterpret_cast<const uint4 *>(lin)[2 * L + half];
				const uint32_t r[4] = {q.x, q.y, q.z, q.w};
				const uint32_t sl = sel >> (8 * half);
#pragma unroll
				for (int bit = 0; bit < 8; bit++)
					x ^= (0u - ((sl >> bit) & 1)) & (r[bit >> 1] >> (16 * (bit & 1)));
			}
			return x & 0xffff;
		};
		// reg(0, data, L) from the every-fourth-byte table and up to three more bytes
		auto data_reg = [&](int layout, uint32_t L) {
			const uint32_t q = L >> 2, r = L & 3;
			uint32_t crc, w;
			if (layout == 0) { crc = p4c(p)[q]; w = r ? pk_bits32(pk[p], 122 + 32 * q, 32) : 0; }
			else if (layout == 1) { crc = p4a(p)[q]; w = r ? a_bytes[p][q] : 0; }
			else { crc = p4b[p][q]; w = r ? b_bytes[p][q] : 0; }
			// + the chunk's start register carried to word q
			const uint32_t start = chunk_reg[p][(layout == 0 ? 0u : layout == 1 ? 11u : 19u) + (q >> 3)], iw = q & 7;
			crc ^= iw ? (uint32_t)(advw[((iw - 1) * 2) * 256 + (start & 0xff)] ^ advw[((iw - 1) * 2 + 1) * 256 + (start >> 8)]) : start;
			for (uint32_t j = 0; j < r; j++)
				crc = crc_byte(crc, (w >> (8 * j)) & 0xff);
			return crc;
		};
		auto crc_is_zero = [&](int layout, uint32_t L) {
			return (data_reg(layout, L) ^ lin_terms(L)) == 0;
		};
		int rv = 1;
		switch (type) {
		case 2: {                                               // fhs (:783-818)
			if (size < 240) { rv = 1; break; }
			if (a_fail[p] < 16) { rv = 0; break; }
			const uint32_t x = data_reg(1, 20) ^ seed_row20(seed);   // zero register <=> x == reg(0, whitening of the attempt, 20)
			rv = 0;
			if (!wht) {
				if (x == 0) rv = 1000;
			} else {
				// attempt 0 is the trial's own clock, attempts 1..32 are clocks 32..63: their registers in four
				// 16-byte reads (not unrolled: the kernel sits at its 128-VGPR ceiling)
				uint32_t hit = x == pw20[clock];
				const uint32_t xx = x | (x << 16);
#pragma unroll 1
				for (int v = 0; v < 4; v++) {
					const uint4 q = reinterpret_cast<const uint4 *>(&pw20[32])[v];
					const uint32_t d0 = q.x ^ xx, d1 = q.y ^ xx, d2 = q.z ^ xx, d3 = q.w ^ xx;
					hit |= ((d0 & 0xffff) == 0) | ((d0 >> 16) == 0) | ((d1 & 0xffff) == 0) | ((d1 >> 16) == 0)
					     | ((d2 & 0xffff) == 0) | ((d2 >> 16) == 0) | ((d3 & 0xffff) == 0) | ((d3 >> 16) == 0);
				}
				if (hit) rv = 1000;
			}
			break;
		}
		case 3: case 8: case 10: case 14:                       // DM (:898-958)
		case 4: case 11: case 15: {                             // DH (:962-1011)
			const bool fec = type == 3 || type == 8 || type == 10 || type == 14;
			const int layout = !fec ? 0 : (type == 8 ? 2 : 1);
			const int psize = type == 8 ? size - 80 : size;
			const int hb = (type == 3 || type == 8 || type == 4) ? 1 : 2;
			const int hbits = 8 * hb;
			const uint32_t fail = layout == 2 ? b_fail[p] : a_fail[p];
			rv = 0;
			if (psize < hbits) break;                           // decode_payload_header (:821-895) gives up
			uint32_t raw;
			if (fec) {
				if (psize < (hb == 2 ? 30 : 15)) break;
				if (fail < (uint32_t)hb) break;
				raw = (layout == 2 ? b_bytes[p][0] : a_bytes[p][0]) & ((1u << hbits) - 1);
			} else {
				raw = pk_bits32(pk[p], 122, hbits);
			}
			const uint32_t ph = raw ^ (wht ? (uint32_t)wh_bits(wh_start(clock, 18), hbits) : 0u);
			int plen = hb == 2 ? (int)((ph >> 3) & 0x3ff) + 4 : (int)((ph >> 3) & 0x1f) + 3;
			int cap;
			switch (type) {
			case 3:  cap = 20;  break;
			case 4:  cap = 30;  break;
			case 8:  cap = 12;  break;
			case 10: cap = 125; break;
			case 11: cap = 187; break;
			case 14: cap = 228; break;
			default: cap = 343; break;
			}
			if (plen > cap) plen = cap;
			const int nbits = plen * 8;
			if (nbits > psize) { rv = 1; break; }
			if (fec && fail < (uint32_t)(nbits + 9) / 10) break; // a block of the payload does not decode
			rv = crc_is_zero(layout, (uint32_t)plen) ? 10 : 2;
			break;
		}
		case 12: {                                              // EV4 (:1044-1097)
			// iterations b = 0 .. B-1 of the reference's block loop get past its two checks
			uint32_t B = size >= 15 ? (uint32_t)size / 15 : 0;
			if (B > 98) B = 98;
			if (B > a_fail[p]) B = a_fail[p];
			const uint32_t lmax = B ? 5 * (B - 1) / 4 : 0;        // bytes L-1 with ceil(4 L / 5) <= B - 1 are reached
			uint32_t crc = seed, idx = wh_start(clock, 18);
			rv = B == 98 ? 2 : 1;
			// Four bytes per step, and the register after EACH of them from ten independent table reads (the
			// same slicing as crc_word: the 16-bit register is used up by the first two bytes) -- the scan for
			// the first zero register is a chain of up to 121 dependent steps otherwise, and with the trials
			// sorted by type the EV4 waves are what the other fifteen wait for at the barrier.
			for (uint32_t L0 = 0; L0 < lmax; L0 += 4) {
				const uint32_t w = a_bytes[p][L0 >> 2] ^ (wht ? (uint32_t)wh_bits(idx, 32) : 0u);
				idx = idx + 32 >= 127 ? idx + 32 - 127 : idx + 32;
				const uint32_t x0 = (crc ^ w) & 0xff, x1 = ((crc ^ w) >> 8) & 0xff, b2 = (w >> 16) & 0xff, b3 = w >> 24;
				const uint32_t c1 = (crc >> 8) ^ g_lds.crc[x0];
				const uint32_t c2 = g_lds.crc_z[0][x0] ^ g_lds.crc[x1];
				const uint32_t c3 = g_lds.crc_z[1][x0] ^ g_lds.crc_z[0][x1] ^ g_lds.crc[b2];
				const uint32_t c4 = g_lds.crc_z[2][x0] ^ g_lds.crc_z[1][x1] ^ g_lds.crc_z[0][b2] ^ g_lds.crc[b3];
				// byte counts L0 + 1 .. L0 + 4; a zero register counts from 2 bytes on and up to lmax
				const bool z1 = c1 == 0 && L0 + 1 >= 2 && L0 + 1 <= lmax, z2 = c2 == 0 && L0 + 2 <= lmax;
				const bool z3 = c3 == 0 && L0 + 3 <= lmax, z4 = c4 == 0 && L0 + 4 <= lmax;
				if (z1 || z2 || z3 || z4) { rv = 10; break; }
				crc = c4;
			}
			break;
		}
		case 5: rv = hv_rv[p]; break;                           // HV1
		default: rv = 1; break;                                 // EV3 / EV5 always map to 1, the rest is not checked
		}
		if (rv == 0 && type != 2 && type != 3 && type != 5)
			rv = 1;
		t_rv[i] = (int16_t)rv;
	}
	__syncthreads();
	TL_PROF(5);
	// The next batch moves in and the one after that is requested BEFORE this batch's results are stored: gfx9
	// counts loads and stores in one in-order counter, so a wait for prefetched words that comes after the
	// stores also waits for the stores (47 % of the kernel when it was written the other way round).
	stage_in(batch + gridDim.x);
	fetch(batch + 2 * gridDim.x);
	// 4. out, in (packet, clock) order (the t_* arrays are not touched before the barrier at the loop top)
	static_assert(sizeof(btbbx_trial) == 4, "one dword per trial");
	for (uint32_t i = tid; i < total; i += TL_THREADS)
		reinterpret_cast<uint32_t *>(trials)[(uint64_t)first * 64 + i] =
			(t_info[i] & 0xffff) | ((uint32_t)(uint16_t)t_rv[i] << 16);
	TL_PROF(6);
	}
	TL_PROF_END;
}

// Two other shapes of this kernel were built and measured in round 4 and are NOT in the source (kept as text in
// profiles/r04_trials/, both bit-exact on every GPU test):
//   * trials_wave_kernel: a WAVE owns four packets from first word to last result, no workgroup barrier at all.  It does
//     what it was built for -- SQ_WAIT_ANY 72 % -> 43 % of the wave-cycles, VALU-active 9.6 % -> 22 % -- and is slower,
//     1.33 against 0.94 ms per 2^20 packets: 780 VALU wave-instructions per packet against 421 (pmc_*.json there).  A
//     sort over the 256 trials of four packets leaves four types in every pass of 64 (the workgroup-wide sort over 4096
//     leaves one), so the DM/DH, FHS and EV4 code runs in every pass with a quarter of the lanes; the one-lane-per-
//     packet steps are issued by every wave instead of one in sixteen; 80 chunk tasks on 64 lanes are two passes.
//   * trials_hybrid_kernel: the packet-local phases wave-local as above, the type sort workgroup-wide as here, three
//     barriers per batch instead of six: 1.10 ms (the redundant one-lane steps and the second chunk pass cost more than
//     the three barriers saved).
// What stayed: the a_fail reads below no longer go through a generic pointer (lds_now: a volatile generic read is a
// FLAT load, which waits for every prefetched word in flight): 0.957 -> 0.944 ms; the index arithmetic of fetch / stage_in is
// kept out of the batch loop's preheader and the FEC loop rolled (13 spilled registers -> none); every wave works the type
// bases out for itself, which takes thread 0's sixteen dependent LDS steps and their barrier off the path.  The last two
// are within the noise (0.927-0.938 ms; 670 -> 689 M packets/s on random packets of every type, profiles/r04_trials/
// trials_ab2.txt): the kernel waits on the dependent LDS steps of its phases, not on these.
// Small batches (a handful of packets from a live receiver): one workgroup per (packet, clock),
// lane 0 runs the trial.  64 x n waves spread over the CUs, none of them serialising different packet
// types, so the call takes as long as the longest single trial -- the lane-per-clock kernel above is
// the throughput shape, this one the latency shape.
__global__ __launch_bounds__(64) void trials_wide_kernel(const uint64_t *packets, const btbbx_pkt_in *in,
							  uint32_t n_packets, btbbx_trial *trials)
{
	chain_lds_init();
	const uint32_t pkt = blockIdx.x >> 6, clock = blockIdx.x & 63;
	if (threadIdx.x || pkt >= n_packets)
		return;
	const btbbx_pkt_in pi = in[pkt];
	PState s;
	s.w = packets + (uint64_t)pkt * BTBBX_PKT_WORDS;
	s.length = (int)pi.length;
	s.flags = pi.flags;
	s.uap = pi.uap;
	s.type = pi.type;
	s.llid = pi.llid;
	s.flow = pi.flow;
	s.plen = 0; s.phl = 0; s.ph16 = 0; s.ph_written = 0; s.dirty = 0; s.ph_mask = 0;
	s.lt_addr = s.hdr_flags = s.hec = s.header18 = 0;
	s.out = OutRef();
	s.written = 0;
	uint32_t dis;
	const uint32_t hdr = header_fec13(s.w, dis);
	const uint32_t uap = do_try_clock(s, clock, hdr, dis);
	const int rv = do_crc_check<false>(s, clock);
	btbbx_trial t;
	t.uap = (uint8_t)uap;
	t.type = (uint8_t)s.type;
	t.rv = (int16_t)rv;
	trials[(uint64_t)pkt * 64 + clock] = t;
}

// The HEC-only half of the brute force (config 5 of BASELINE.json: "64 whitening seeds x HEC
// check"): table[p * 64 + c] = try_clock(c)'s return value | packet_type(c) << 8, 0 when the FEC 1/3
// of the header fails.  uap_from_hec (:693-705) and the type field are GF(2)-linear in the 18
// header bits, and unwhitening XORs a clock-dependent constant onto them, so
//     UAP(c) = U(header) ^ U(whitening bits of c),
// one LFSR run per packet and a 64-entry constant table instead of 64 runs.  The kernel is then
// pure data movement: 8 useful bytes in (the header symbols 68..121 sit in word 1 of a packed
// packet), 128 bytes out per packet.  A wave takes 64 packets; lane L first decodes packet L,
// then the wave writes 8 x 1 KiB: in store j lane L emits the 8 clocks 8 (L & 7).. of packet
// 8 j + (L >> 3), fetching that packet's value with one lane-to-lane read.
__global__ __launch_bounds__(256) void uap_table_kernel(const uint64_t *packets, const btbbx_pkt_in *in, uint32_t n,
							 uint4 *table)
{
	const uint32_t lane = threadIdx.x & 63;
	const uint32_t pkt0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
	if (pkt0 >= n)
		return;
	uint32_t ut = 0;                               // U | type << 8 | fec ok << 16 | whitened << 17
	if (pkt0 + lane < n) {
		const uint32_t p = pkt0 + lane;
		uint32_t dis;
		const uint32_t hdr = fec13((packets[(uint64_t)p * BTBBX_PKT_WORDS + 1] >> 4) & ((1ULL << 54) - 1), 18, dis);
		const uint32_t wht = in ? (in[p].flags & F_WHITENED) : 1u;
		ut = uap_from_hec(hdr & 0x3ff, hdr >> 10) | (((hdr >> 3) & 0xf) << 8) | ((dis < 4 ? 1u : 0u) << 16) | (wht << 17);
	}
	uint32_t wc[4] = {0, 0, 0, 0};                 // this lane's 8 clocks, two 16-bit entries per word
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const uint32_t wb = (uint32_t)wh_bits_const(wh_start_const(8 * (lane & 7) + k, 0), 18);
		const uint32_t e = uap_from_hec(wb & 0x3ff, wb >> 10) | (((wb >> 3) & 0xf) << 8);
		wc[k >> 1] |= e << (16 * (k & 1));
	}
#pragma unroll
	for (int j = 0; j < 8; j++) {
		const uint32_t src = 8 * j + (lane >> 3);
		const uint32_t v = (uint32_t)__shfl((int)ut, (int)src);
		const uint32_t both = (v & 0xffff) * 0x10001u;
		const uint32_t okm = 0u - ((v >> 16) & 1u), whm = 0u - ((v >> 17) & 1u);
		uint4 o;
		o.x = (both ^ (wc[0] & whm)) & okm;
		o.y = (both ^ (wc[1] & whm)) & okm;
		o.z = (both ^ (wc[2] & whm)) & okm;
		o.w = (both ^ (wc[3] & whm)) & okm;
		if (pkt0 + src < n)
			table[(uint64_t)(pkt0 + src) * 8 + (lane & 7)] = o;
	}
}

// mode bits of decode_kernel (packet_obj.h):
//   DEC_HEADER   btbb_decode_header
//   DEC_PAYLOAD  btbb_decode_payload (after a successful header when DEC_HEADER is set)
// (DEC_TRIALS -- leave the packet as a set of try_clock / crc_check calls leaves it -- is
//  replay_kernel / trials_state_kernel + trials_merge_kernel below)

#ifdef DH_PROFILE
__device__ unsigned long long g_dh_prof[8];
#define DH_MARK(k) do { uint64_t now_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) : : "memory"); \
	__builtin_amdgcn_sched_barrier(0); dh_acc[k] += (uint32_t)(now_ - dh_t); dh_t = now_; } while (0)
#define DH_PARAMS , uint32_t *dh_acc, uint64_t &dh_t
#define DH_PASS , dh_acc, dh_t
#else
#define DH_MARK(k) do { } while (0)
#define DH_PARAMS
#define DH_PASS
#endif
// header_present + decode_header / decode_payload of one packet (w = its 50 packed words)
// `s` arrives with its view of the packet set (w, length and, for a packet read straight from the stream, sh /
// wlimit / direct); everything else of the entry state comes from pi and *o
__device__ __forceinline__ void decode_view(PState &s, const btbbx_pkt_in &pi, btbbx_pkt_out *o, uint32_t mode,
					    OutRef pay_out, uint64_t *head_out, const uint64_t *head_in DH_PARAMS)
{

	s.flags = pi.flags;
	s.uap = pi.uap;
	s.type = pi.type;
	s.llid = pi.llid;
	s.flow = pi.flow;
	// The fixed part of btbbx_pkt_out (40 bytes in front of the payload words) is read and written as FIVE 8-byte
	// words: a lane per packet means every vector memory instruction touches 64 different sectors, and the address
	// unit works those off one by one -- twenty field-sized accesses per packet were most of decode_hits_kernel's time.
	static_assert(offsetof(btbbx_pkt_out, payload) == 40 && offsetof(btbbx_pkt_out, payload_header) == 32, "head of btbbx_pkt_out");
	union Head {
		uint64_t q[5];
		struct {
			int32_t header_rv, payload_rv, payload_length, payload_header_length;
			uint32_t flags, header_packed;
			uint8_t header_present, type, lt_addr, hdr_flags, hec, llid, flow, uap;
			uint64_t payload_header;
		} f;
	} hd;
	{
		const uint64_t *src = head_in ? head_in : reinterpret_cast<const uint64_t *>(o);
#pragma unroll
		for (int k = 0; k < 5; k++)
			hd.q[k] = src[k];
	}
	s.plen = hd.f.payload_length;
	s.phl = hd.f.payload_header_length;
	s.ph16 = (uint32_t)hd.f.payload_header;
	s.ph_written = 0;
	s.dirty = 0;
	s.ph_mask = 0;
	s.lt_addr = hd.f.lt_addr; s.hdr_flags = hd.f.hdr_flags; s.hec = hd.f.hec; s.header18 = hd.f.header_packed;
	s.out = (pay_out.l || pay_out.g) ? pay_out : OutRef(o->payload);
	s.written = 0;

	int header_rv = 0, payload_rv = 0;
	DH_MARK(3);
	uint32_t hraw, hdis;
	if (s.has_pre) {
		hraw = s.pre_hdr;
		hdis = s.pre_dis;
	} else {
		hraw = header_fec13(s, hdis);
	}
	hd.f.header_present = (uint8_t)do_header_present(s, hdis);
	DH_MARK(4);

	{
		bool go = true;
		if (mode & DEC_HEADER) {
			// btbb_decode_header (:1198-1221)
			const uint32_t dis = hdis, hdr = hraw;
			go = false;
			if ((s.flags & F_CLK6_VALID) && dis < 4) {
				uint32_t clear = hdr ^ (uint32_t)wh(s, wh_start(pi.clkn, 0), 18);
				s.header18 = clear;
				uint32_t hec = clear >> 10;
				if (uap_from_hec(clear & 0x3ff, hec) == s.uap) {
					s.lt_addr = clear & 7;
					s.type = (clear >> 3) & 0xf;
					s.hdr_flags = (clear >> 7) & 7;
					s.hec = hec;
					header_rv = 1;
					go = true;
				}
			}
		}
		DH_MARK(5);
		if ((mode & DEC_PAYLOAD) && go) {
			// btbb_decode_payload (:1223-1297)
			uint32_t clock = pi.clkn;
			s.phl = 0;
			switch (s.type) {
			case 0: case 1: s.plen = 0; payload_rv = 1; break;
			case 2:  payload_rv = do_fhs<true>(s, clock); break;
			case 3: case 8: case 10: case 14: payload_rv = do_DM<true>(s, clock); break;
			case 4: case 9: case 11: case 15: payload_rv = do_DH<true>(s, clock); break;
			case 5: case 6: payload_rv = do_HV<true>(s, clock); break;
			case 7:
				payload_rv = do_EV35<true>(s, clock, 32);
				if (payload_rv <= 1)
					payload_rv = do_HV<true>(s, clock);
				break;
			case 12: case 13: {
				// EV4 / EV5 into HBM: the lane-group phase (payload_length and the verdict come from ev_payloads)
				const uint32_t size = s.length - 122u, unit = s.type == 12 ? 15u : 8u;
				if (s.def_slot && !s.out.l && s.length >= 122u + unit) {
					const uint32_t most = s.type == 12 ? 98u : 182u, units = size / unit < most ? size / unit : most;
					defer_payload(s, clock, (s.type == 12 ? 10u : 8u) * units, s.type == 12 ? DHL_EV4 : DHL_EV5);
					payload_rv = 2;
				} else {
					payload_rv = s.type == 12 ? do_EV4<true>(s, clock) : do_EV35<true>(s, clock, 182);
				}
				break;
			}
			}
			s.flags |= F_HAS_PAYLOAD;
		}
	}
	DH_MARK(6);
	hd.f.header_rv = header_rv;
	hd.f.payload_rv = payload_rv;
	hd.f.payload_length = s.plen;
	hd.f.payload_header_length = s.phl;
	hd.f.flags = s.flags;
	hd.f.header_packed = s.header18;
	hd.f.type = (uint8_t)s.type;
	hd.f.lt_addr = (uint8_t)s.lt_addr;
	hd.f.hdr_flags = (uint8_t)s.hdr_flags;
	hd.f.hec = (uint8_t)s.hec;
	hd.f.llid = (uint8_t)s.llid;
	hd.f.flow = (uint8_t)s.flow;
	hd.f.uap = (uint8_t)s.uap;
	hd.f.payload_header = s.ph16;
	{
		uint64_t *dst = head_out ? head_out : reinterpret_cast<uint64_t *>(o);
#pragma unroll
		for (int k = 0; k < 5; k++)
			dst[k] = hd.q[k];
	}
}

__device__ void decode_one(const uint64_t *w, const btbbx_pkt_in &pi, btbbx_pkt_out *o, uint32_t mode)
{
	PState s;
	s.w = w;
	s.length = (int)pi.length;
#ifdef DH_PROFILE
	uint32_t dh_acc[8];
	uint64_t dh_t = 0;
#endif
	decode_view(s, pi, o, mode, OutRef(), nullptr, nullptr DH_PASS);
}

__global__ __launch_bounds__(64) void decode_kernel(const uint64_t *packets, const btbbx_pkt_in *in,
						     uint32_t n_packets, btbbx_pkt_out *outs, uint32_t mode)
{
	chain_lds_init();
	uint32_t pkt = blockIdx.x * blockDim.x + threadIdx.x;
	if (pkt >= n_packets)
		return;
	decode_one(packets + (uint64_t)pkt * BTBBX_PKT_WORDS, in[pkt], outs + pkt, mode);
}

// Decode straight from the packed streams: what gather_kernel + decode_kernel do, without the 400-byte row that
// the first writes and the second reads back (profiles/r02_v4/pmc_secondary.json: the two moved 2.0 GB per
// 1.29 M packets, of which the packets themselves are 0.5 GB).  One lane per hit; the captured length is the
// gather's: min(max_length, 3125, symbols left in the stream), and d_in[i].length is ignored.
//
// A lane walking its packet word by word from HBM fetched 753 B per packet for ~300 needed (every 8-byte read
// drags a 64-byte sector through the L2, profiles/traffic_secondary.json, round 2) and sat out a latency per step.
// The kernel's phases now (NOTEBOOK.md 3.4 has the numbers behind each):
//   A  every lane loads its hit and, in one batch, words 1 .. 4 of its packet; from those it decodes the header and
//      the payload header under its clock: the packet's type and EXACTLY how many symbols its decoder will read
//   B  the workgroup's 256 packets change hands (counting sort on decoder and length): one decoder per wave
//   C  the wave copies its 64 packets into LDS (global_load_lds, a dozen instructions in flight together)
//   D  one lane per packet decodes from LDS; payloads of up to 256 bits go to a per-lane LDS copy of the record
//   E  the wave stores head + payload of packet after packet as consecutive words (one 64-byte sector for most)
// s_bits() takes words the staging did not cover (DH_STAGE_WORDS per wave) from the stream as before, so the extents
// only decide where a word comes from, never what it is.
#define DH_STAGE_WORDS 384u                  // LDS words per wave for staged packets (3 KiB; 4 waves per workgroup)
__device__ __forceinline__ uint32_t symbols_of_type(uint32_t type)
{
	// 122 symbols of access code + trailer + header, then the longest payload of the type (FEC 2/3: 15 symbols per
	// 10 bits): bluetooth_packet.c:771-1196.  Single-slot types 366, three-slot 1626, five-slot the whole capture.
	if (type == 10 || type == 11 || type == 12 || type == 13)
		return 1626;
	if (type == 14 || type == 15)
		return BTBBX_MAX_SYMBOLS;
	return 366;
}

// which payload decoder a type runs (decode_view's switch)
__device__ __forceinline__ uint32_t decoder_of_type(uint32_t type)
{
	// 0 none, 1 FHS, 2 DM, 3 DH, 4 HV, 5 EV3 (+ HV), 6 EV4, 7 EV5: a nibble per type
	return (uint32_t)(0x3276323254432100ULL >> (4 * type)) & 0xf;
}
#define DH_OUT_WORDS 4u                      // payload words per lane that leave through LDS
#define DH_OUT_SECTOR 3u                     // ... of which these share the 64-byte sector of the record's head
// How many symbols of the packet the payload decoder of `type` will look at under this clock, and whether what it
// writes fits DH_OUT_WORDS words (small; wide: it needs the last of them, which lies in the record's second sector).
// DM / DH / AUX1 / DV carry their length in the payload header (do_payload_header, the
// decoders' own first step, on a scratch copy of the state): a DM3 with twelve bytes in it is 6 words of stream, not
// the 26 its type could have -- with the type's bound alone a wave with sixteen DM3 in it ran out of its LDS stage
// and half its lanes read their packets from HBM word by word.  An estimate that is too small only sends s_bits() to
// the stream for the rest; it never changes what is read.
__device__ __forceinline__ uint32_t payload_extent(const PState &s0, uint32_t type, uint32_t clock, bool &small, bool &wide)
{
	bool fec = false;
	int header_bytes = 2;
	uint32_t pos = 122;
	switch (type) {
	case 3:  fec = true; header_bytes = 1; break;
	case 8:  fec = true; header_bytes = 1; pos = 202; break;
	case 10: case 14: fec = true; break;
	case 4: case 9: header_bytes = 1; break;
	case 11: case 15: break;
	default: {
		// payload bits the other single-slot decoders write at most: nothing for NULL / POLL, FHS 160, HV1 80,
		// HV2 160, HV3 240 (type 7 tries EV3 first: 256); EV4 / EV5 run over several slots
		const uint32_t bits = (0x85300500u >> (4 * (type & 7)) & 0xf) * 32u;   // (rounded up to 32; types >= 8 never get here as small)
		small = type < 8 && bits <= 64 * DH_OUT_WORDS;
		wide = small && bits > 64 * DH_OUT_SECTOR;
		return symbols_of_type(type);
	}
	}
	PState s = s0;
	s.type = type;
	s.ph16 = 0; s.ph_mask = 0; s.dirty = 0; s.ph_written = 0;
	small = true;
	wide = false;
	if (!do_payload_header<false>(s, pos, clock, header_bytes, s.length - (int)pos, fec))
		return pos + 30;
	const uint32_t nbits = (uint32_t)s.plen * 8;
	small = nbits <= 64 * DH_OUT_WORDS;
	wide = small && nbits > 64 * DH_OUT_SECTOR;
	return pos + (fec ? 15 * ((nbits + 9) / 10) : nbits);
}

// ---- long payloads: a group of lanes per packet ---------------------------------------------------------------------
// A lane that walks a DM3 / DH3 / DM5 / DH5 payload alone reads one stream word and writes one record word per step,
// each a sector of its own, one latency after the other: 1.4 - 3.5 ms per 1.29 M full-length packets against 0.13 - 0.15
// for the single-slot types (profiles/r03_chain/decode_by_type.txt).  do_DM / do_DH therefore stop after their checks
// when the payload has more than DHL_MIN_BITS bits, EV4 / EV5 before their loops (PState::def_nbits, defer_payload), and
// the wave works those packets off together, a group of G lanes per packet, 64 / G packets per round:
//   long_payloads   DM and DH, TWO payload words per lane, G = 8 / 16 / 32 (the workgroup sort keeps packets of one G
//                   together).  Per round, lane `sub` of a group
//     1. holds four stream words of its packet, requested one round ahead (a round's stores and loads all have a round's
//        worth of work to complete in: gfx9 counts both in one in-order counter);
//     2. DH (:962-1011): payload words 2 sub, 2 sub + 1 are funnel shifts of three of them.  DM (:898-958): the words go
//        to LDS (zeroed at and behind the captured length when a block reaches there: the reference reads zeros), FOUR
//        consecutive blocks of the (15,10) code per lane and step are decoded from LDS and their 40 bits ORed into the
//        packed payload in LDS (two ds_or: they start on a byte); one failing block anywhere in the packet and nothing
//        is written (rv 0), as in the reference;
//     3. unwhitens its words with the whitening bits from (start + 64 j) mod 127 and cuts at payload_length;
//     4. CRC (:671-690, :772-781): the register is GF(2)-linear; a seed is the same as its bits XORed onto the first
//        sixteen message bits; zero bits appended to a message advance the register by an invertible map, so "register
//        == 0" can be tested on the payload padded to whole words; and with A = "advance by one bit" the register after
//        n words is an invertible map applied to the XOR over the words of A^(-64 j) (register of word j alone).  So
//        every lane runs its own two words from a zero register (four four-byte steps), applies the FIXED matrix
//        A^(-128 sub) -- sixteen 16-bit columns per lane from g_adv64inv, loaded once per wave -- and the group XORs:
//        zero <=> the reference's compare of the computed with the received CRC succeeds.  No lane needs another lane's
//        word, whatever the payload length;
//     5. stores its words (344 contiguous bytes for a DH5; the last word keeps the record's bits behind the payload).
//   dh_payloads     a wave with DH payloads only: the same without step 2's LDS, THREE words per lane, G = 8 / 16.
//   ev_payloads     EV4 (:1044-1097) / EV5 (:1099-1128), one word per lane: the payload ends at the first byte count whose
//                   CRC register is zero -- a prefix of registers over the lanes.
// tests/_wave_model.py is the numpy model of the CRC steps (pinned against the oracle on the CPU).
// two LDS areas per wave (in decode_hits_kernel: its input stage and its result stage, both free by then):
#define DHL_STG_WORDS 288u                   // `stg`: the round's DM packets as they lie in the stream: 4 G (+ G / 4 + 1: LDS banks) words per group
#define DHL_LIST   0u                        // `lst`: 64 x 2 words: what the owner lanes know about their deferred packets
#define DHL_PB     128u                      //        128 words: decoded FEC 2/3 bits, packed, 2 G words per group
#define DHL_LST_WORDS 256u
struct __attribute__((packed, aligned(8))) dhl_pair_t { uint64_t a, b; };    // two payload words of a record: one 16-byte store
typedef __attribute__((address_space(3))) uint64_t dhl_u64_t;
typedef __attribute__((address_space(3))) uint32_t dhl_u32_t;
typedef const __attribute__((address_space(1))) uint64_t dhl_g64_t;

// XOR over the 2^logg lanes of a group (3 <= logg <= 6), every lane gets the result
__device__ __forceinline__ uint32_t group_xor(uint32_t x, uint32_t logg)
{
	x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]
	x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);      // quad_perm [2,3,0,1]
	x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xf, 0xf, true);     // row_half_mirror: the other quad of eight
	if (logg > 3)
		x ^= (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xf, 0xf, true); // row_mirror: the other eight of sixteen
	if (logg > 4)
		x ^= (uint32_t)__shfl_xor((int)x, 16);
	if (logg > 5)
		x ^= (uint32_t)__shfl_xor((int)x, 32);
	return x;
}

// the register after the matrix whose columns are the sixteen 16-bit halves of c[0..7]: per pair of register bits
// two sign-extending bit-field extracts, one byte permute that joins their low / high halves, one and-xor
__device__ __forceinline__ uint32_t apply_columns(const uint32_t (&c)[8], uint32_t reg)
{
	uint32_t x = 0;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)reg, 2 * k, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)reg, 2 * k + 1, 1);
		x ^= c[k] & __builtin_amdgcn_perm(m1, m0, 0x07060100u);
	}
	return (x ^ (x >> 16)) & 0xffffu;
}

// All 64 lanes of a wave; the wave's n_def DH / DM list entries are in LDS (lst[DHL_LIST ..]).  `stg` = DHL_STG_WORDS words
// of LDS of this wave, `lst` = DHL_LST_WORDS more; `outs` = the records of the workgroup of decode_hits_kernel that
// deferred the packets.
__device__ __forceinline__ void long_payloads(dhl_u64_t *stg, dhl_u64_t *lst, uint32_t n_def, uint32_t logg, btbbx_pkt_out *outs, uint32_t lane)
{
	dhl_u32_t *const stg32 = (dhl_u32_t *)stg, *const lst32 = (dhl_u32_t *)lst;
	const uint32_t G = 1u << logg, R = 64u >> logg;
	const uint32_t sub = lane & (G - 1), grp = lane >> logg, gbase = grp << logg;
	const uint64_t gmask = (1ULL << G) - 1;                     // (G <= 32: 43 words at two per lane)
	lst[DHL_PB + 2 * lane] = 0;
	lst[DHL_PB + 2 * lane + 1] = 0;
	// this lane's matrix: sixteen columns of A^(-128 sub)
	uint32_t col[8];
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(g_adv64inv) + 4 * sub;
		const uint4 a = src[0], b = src[1];
		col[0] = a.x; col[1] = a.y; col[2] = a.z; col[3] = a.w; col[4] = b.x; col[5] = b.y; col[6] = b.z; col[7] = b.w;
	}
	// a group's staged words: 4 G + G / 4 + 1 words apart, so that the groups' 15-bit reads fall into different LDS banks
	// (4 G words = a multiple of 256 bytes: every group on the same banks)
	const uint32_t stg_base = grp * (4u * G + (G >> 2) + 1u);
	for (uint32_t i = lane; i < DHL_STG_WORDS; i += 64)
		stg[i] = 0;
	const uint32_t wh_lane = (128u * sub) % 127u;              // whitening phase of word 2 sub relative to the payload's first bit
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint32_t rounds = (n_def + R - 1) >> (6 - logg);
	// the words of round r on their way: four stream words of the group's packet -- DH: the three that hold payload words
	// 2 sub and 2 sub + 1; DM: words sub, sub + G, sub + 2 G, sub + 3 G of the packet
	auto request = [&](uint32_t r, uint64_t (&w)[4]) {
		const uint32_t e = r * R + grp;
#pragma unroll
		for (int k = 0; k < 4; k++)
			w[k] = 0;
		if (e < n_def) {
			const uint64_t a = lst[DHL_LIST + 2 * e], b = lst[DHL_LIST + 2 * e + 1];
			dhl_g64_t *const src = (dhl_g64_t *)(uintptr_t)(a & 0xffffffffffffULL);
			const uint32_t p_nw = (uint32_t)(a >> 48) & 127u, p_sh = (uint32_t)(a >> 55) & 63u, kind = (uint32_t)(b >> 32) & 3u;
			const bool p_fec = kind == DHL_DM;
			const uint32_t i0 = p_fec ? sub : 2u * sub + ((p_sh + 122u) >> 6), step = p_fec ? G : 1u;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t i = i0 + (uint32_t)k * step;
				if (i < p_nw && (p_fec || k < 3))
					w[k] = src[i];
			}
		}
	};
	uint64_t nw[4];
	request(0, nw);
	// a round's words are stored at the start of the next round
	uint64_t st_val0 = 0, st_val1 = 0;
	uint32_t st_pkt = 0, st_n = 0;
	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t e = r * R + grp;
		const bool has = e < n_def;
		uint64_t pa = 0, pb = 0;
		if (has) {
			pa = lst[DHL_LIST + 2 * e];
			pb = lst[DHL_LIST + 2 * e + 1];
		}
		const uint32_t p_sh = (uint32_t)(pa >> 55) & 63u;
		const uint32_t p_pkt = (uint32_t)pb & 0xffu, p_len = (uint32_t)(pb >> 8) & 0xfffu, nbits = (uint32_t)(pb >> 20) & 0xfffu;
		const uint32_t kind = (uint32_t)(pb >> 32) & 3u, p_widx = (uint32_t)(pb >> 35) & 127u, p_uap = (uint32_t)(pb >> 42) & 0xffu;
		const bool p_fec = has && kind == DHL_DM, p_wht = (pb >> 34) & 1u;
		const uint32_t nblocks = (nbits + 9u) / 10u;
		const uint32_t T = nbits >> 6, nwp = (nbits + 63u) >> 6;
		// 2a. DH: payload words 2 sub, 2 sub + 1 are funnel shifts of the lane's three stream words
		// (computed by every lane, wanted or not: the one wait for the words asked for a round ago then sits here, on every
		// path, and the compiler needs no second one in front of the next request)
		const uint32_t sft = (p_sh + 122u) & 63u;
		uint64_t word0 = sft ? (nw[0] >> sft) | (nw[1] << (64u - sft)) : nw[0];
		uint64_t word1 = sft ? (nw[1] >> sft) | (nw[2] << (64u - sft)) : nw[1];
		if (!(has && kind == DHL_DH)) {
			word0 = 0;
			word1 = 0;
		}
		const uint64_t any_fec = __ballot(p_fec);
		bool fail = false;
		if (any_fec) {
			// the DM packets of the round into LDS, cut at the captured length when a block reaches behind it
			if (__ballot(p_fec && 122u + 15u * nblocks > p_len)) {
				const uint32_t valid = p_sh + p_len;                        // stream bits of the packet's words that are symbols of the capture
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t first = 64u * (sub + (uint32_t)k * G), h = valid > first ? valid - first : 0u;
					if (h < 64)
						nw[k] &= (1ULL << h) - 1;
				}
			}
#pragma unroll
			for (int k = 0; k < 4; k++)
				stg[stg_base + (uint32_t)k * G + sub] = nw[k];
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// 2b. DM: the (15,10) blocks of the packet
			// FOUR consecutive blocks per lane and step: 60 stream bits from three staged dwords (every read unconditional:
			// a lane without blocks reads the group's first words), four parity and four correction look-ups in flight
			// together, and the 40 payload bits start on a byte of the packed payload -- two ds_or, no branch.  Blocks behind
			// the packet's last are zeroed before they are decoded (zeros decode to zeros); bits of the last block behind
			// payload_length end in the partial last word, which step 3 cuts, or in a word no lane keeps.  (Two blocks per
			// lane and step, one ds_or pair each: 37 instructions per block against 14.)
			for (uint32_t n0 = 0; ; n0 += G) {
				const uint32_t n = n0 + sub;
				const bool on = p_fec && 4u * n < nblocks;
				if (!__ballot(on))
					break;
				const uint32_t left = nblocks - 4u * n, have = on ? (left < 4u ? left : 4u) : 0u;
				const uint32_t q = p_sh + 122u + 60u * n, i = 2u * stg_base + (on ? q >> 5 : 0u);
				const uint32_t w0 = stg32[i], w1 = stg32[i + 1], w2 = stg32[i + 2];
				const uint64_t vm = (1ULL << (15u * have)) - 1;
				const uint32_t x0 = __builtin_amdgcn_alignbit(w1, w0, q & 31u) & (uint32_t)vm;
				const uint32_t x1 = __builtin_amdgcn_alignbit(w2, w1, q & 31u) & (uint32_t)(vm >> 32);
				const uint32_t b2 = __builtin_amdgcn_alignbit(x1, x0, 30);
				uint32_t d0 = x0 & 0x3ffu, d1 = (x0 >> 15) & 0x3ffu, d2 = b2 & 0x3ffu, d3 = (x1 >> 13) & 0x3ffu;
				const uint32_t m0 = g_lds.fixm23[((x0 >> 10) & 31u) ^ g_lds.par23[d0]], m1 = g_lds.fixm23[((x0 >> 25) & 31u) ^ g_lds.par23[d1]];
				const uint32_t m2 = g_lds.fixm23[((b2 >> 10) & 31u) ^ g_lds.par23[d2]], m3 = g_lds.fixm23[((x1 >> 23) & 31u) ^ g_lds.par23[d3]];
				if ((m0 | m1 | m2 | m3) >> 15)
					fail = true;
				d0 ^= m0 & 0x3ffu;
				d1 ^= m1 & 0x3ffu;
				d2 ^= m2 & 0x3ffu;
				d3 ^= m3 & 0x3ffu;
				if (on) {
					const uint32_t byte = 5u * n, d = 2u * DHL_PB + 4u * gbase + (byte >> 2);
					// Bits behind payload_length never leave the group's packed area: at 128 bytes (1024 bits = exactly the sixteen
					// words of a group of eight lanes) the six spare bits of block 102 would otherwise be ORed into word 0 of the next
					// group's packet -- non-zero whenever that block is mis-corrected or the packet is noise.
					const uint32_t room = nbits - 40u * n;                  // > 0: 4 n < nblocks = ceil(nbits / 10)
					uint64_t dv = (uint64_t)(d3 >> 2) << 32 | (d0 | d1 << 10 | d2 << 20 | d3 << 30);
					if (room < 40u)
						dv &= (1ULL << room) - 1;
					const uint64_t v = dv << (8u * (byte & 3u));
					__hip_atomic_fetch_or(lst32 + d, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					__hip_atomic_fetch_or(lst32 + d + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (p_fec) {
				word0 = lst[DHL_PB + 2 * lane];
				word1 = lst[DHL_PB + 2 * lane + 1];
			}
			lst[DHL_PB + 2 * lane] = 0;                                 // the packed bits are consumed: ready for the next round
			lst[DHL_PB + 2 * lane + 1] = 0;
		}
		// the stream words are used up: the previous round's words go out, the next round's words are asked for, and the
		// lane that writes a partial last word asks for what the record holds there (used at the end of the round) --
		// all of it behind the last wait of this round for memory, in front of ~200 instructions that need none
		if (st_n == 2) {
			*reinterpret_cast<dhl_pair_t *>(outs[st_pkt].payload + 2 * sub) = dhl_pair_t{st_val0, st_val1};
		} else if (st_n == 1) {
			outs[st_pkt].payload[2 * sub] = st_val0;
		}
		if (r + 1 < rounds)
			request(r + 1, nw);
		const uint32_t j0 = 2u * sub;
		const bool act0 = has && j0 < nwp, act1 = has && j0 + 1u < nwp;
		const bool part = has && (T >> 1) == sub && (nbits & 63u);  // this lane holds the partial last word
		uint64_t oldw = 0;
		if (part)
			oldw = outs[p_pkt].payload[T];
		const uint64_t fail_mask = __ballot(fail);
		const bool group_fail = ((fail_mask >> gbase) & gmask) != 0;
		// 3. unwhitened, cut at the payload length
		uint64_t out0 = 0, out1 = 0;
		const uint64_t keep = (1ULL << (nbits & 63u)) - 1;          // (of the partial last word)
		if (act0) {
			uint32_t idx = p_widx + wh_lane;
			idx = idx >= 127u ? idx - 127u : idx;
			out0 = word0 ^ (p_wht ? wh_bits(idx, 64) : 0ULL);
			if (part && !(T & 1u))
				out0 &= keep;
			if (act1) {
				idx += 64u;
				idx = idx >= 127u ? idx - 127u : idx;
				out1 = word1 ^ (p_wht ? wh_bits(idx, 64) : 0ULL);
				if (part && (T & 1u))
					out1 &= keep;
			}
		}
		// 4. CRC: the lane's two words from a zero register (the seed's bits on the first sixteen of the payload), carried back
		// over the words in front of them
		const uint64_t cw = out0 ^ (sub == 0 ? (uint64_t)crc_seed(p_uap) : 0ULL);
		uint32_t reg = crc_word(crc_word(0, (uint32_t)cw), (uint32_t)(cw >> 32));
		reg = crc_word(crc_word(reg, (uint32_t)out1), (uint32_t)(out1 >> 32));
		const uint32_t total = group_xor(apply_columns(col, reg), logg);
		int rv = total == 0 ? 10 : 2;
		if (p_fec && group_fail)
			rv = 0;
		// 5. out (DM: nothing when a block failed)
		st_n = rv == 0 ? 0u : act1 ? 2u : act0 ? 1u : 0u;
		st_val0 = part && !(T & 1u) ? out0 | (oldw & ~keep) : out0;
		st_val1 = part && (T & 1u) ? out1 | (oldw & ~keep) : out1;
		st_pkt = p_pkt;
		if (has && sub == 0)
			outs[p_pkt].payload_rv = rv;                            // (decode_hits_kernel left a placeholder)
	}
	if (st_n == 2) {
		*reinterpret_cast<dhl_pair_t *>(outs[st_pkt].payload + 2 * sub) = dhl_pair_t{st_val0, st_val1};
	} else if (st_n == 1) {
		outs[st_pkt].payload[2 * sub] = st_val0;
	}
}

// A wave whose deferred payloads are all DH (no FEC 2/3: nothing goes through LDS but the list): THREE payload words per lane,
// G = 8 / 16 lanes per packet -- a DH5 takes 15 lanes of 16 and four share a round (two words per lane: 22 of 32, two per round),
// a DH3 eight of eight.  The steps are long_payloads' 1, 2a, 3, 4, 5 with the matrix A^(-192 sub); the lane's four stream words are
// what a DM lane stages, so the prefetch costs the same registers.
__device__ __forceinline__ void dh_payloads(dhl_u64_t *lst, uint32_t n_def, uint32_t logg, btbbx_pkt_out *outs, uint32_t lane)
{
	const uint32_t G = 1u << logg, R = 64u >> logg;
	const uint32_t sub = lane & (G - 1), grp = lane >> logg;
	uint32_t col[8];
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(g_adv64inv) + 6 * sub;
		const uint4 a = src[0], b = src[1];
		col[0] = a.x; col[1] = a.y; col[2] = a.z; col[3] = a.w; col[4] = b.x; col[5] = b.y; col[6] = b.z; col[7] = b.w;
	}
	const uint32_t wh_lane = (192u * sub) % 127u;              // whitening phase of word 3 sub relative to the payload's first bit
	const uint32_t rounds = (n_def + R - 1) >> (6 - logg);
	auto request = [&](uint32_t r, uint64_t (&w)[4]) {
		const uint32_t e = r * R + grp;
#pragma unroll
		for (int k = 0; k < 4; k++)
			w[k] = 0;
		if (e < n_def) {
			const uint64_t a = lst[DHL_LIST + 2 * e];
			dhl_g64_t *const src = (dhl_g64_t *)(uintptr_t)(a & 0xffffffffffffULL);
			const uint32_t p_nw = (uint32_t)(a >> 48) & 127u, p_sh = (uint32_t)(a >> 55) & 63u;
			const uint32_t i0 = 3u * sub + ((p_sh + 122u) >> 6);
#pragma unroll
			for (int k = 0; k < 4; k++)
				if (i0 + (uint32_t)k < p_nw)
					w[k] = src[i0 + (uint32_t)k];
		}
	};
	uint64_t nw[4];
	request(0, nw);
	uint64_t st_val[3] = {0, 0, 0};
	uint32_t st_pkt = 0, st_n = 0;
	auto store = [&]() {
		uint64_t *const dst = outs[st_pkt].payload + 3 * sub;
		if (st_n >= 2)
			*reinterpret_cast<dhl_pair_t *>(dst) = dhl_pair_t{st_val[0], st_val[1]};
		else if (st_n == 1)
			dst[0] = st_val[0];
		if (st_n == 3)
			dst[2] = st_val[2];
	};
	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t e = r * R + grp;
		const bool has = e < n_def;
		uint64_t pa = 0, pb = 0;
		if (has) {
			pa = lst[DHL_LIST + 2 * e];
			pb = lst[DHL_LIST + 2 * e + 1];
		}
		const uint32_t p_sh = (uint32_t)(pa >> 55) & 63u;
		const uint32_t p_pkt = (uint32_t)pb & 0xffu, nbits = (uint32_t)(pb >> 20) & 0xfffu;
		const uint32_t p_widx = (uint32_t)(pb >> 35) & 127u, p_uap = (uint32_t)(pb >> 42) & 0xffu;
		const bool p_wht = (pb >> 34) & 1u;
		const uint32_t T = nbits >> 6, nwp = (nbits + 63u) >> 6;
		// the lane's three payload words: funnel shifts of its four stream words (by every lane, wanted or not: the round's one
		// wait for memory sits here)
		const uint32_t sft = (p_sh + 122u) & 63u;
		uint64_t word[3];
#pragma unroll
		for (int k = 0; k < 3; k++) {
			word[k] = sft ? (nw[k] >> sft) | (nw[k + 1] << (64u - sft)) : nw[k];
			if (!has)
				word[k] = 0;
		}
		store();
		if (r + 1 < rounds)
			request(r + 1, nw);
		const uint32_t j0 = 3u * sub, Tq = T / 3u, Tr = T - 3u * Tq;
		const bool part = has && Tq == sub && (nbits & 63u);        // this lane holds the partial last word
		uint64_t oldw = 0;
		if (part)
			oldw = outs[p_pkt].payload[T];
		const uint64_t keep = (1ULL << (nbits & 63u)) - 1;          // (of the partial last word)
		uint64_t out[3] = {0, 0, 0};
		uint32_t idx = p_widx + wh_lane;
		idx = idx >= 127u ? idx - 127u : idx;
		uint32_t n_act = 0;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			if (has && j0 + (uint32_t)k < nwp) {
				out[k] = word[k] ^ (p_wht ? wh_bits(idx, 64) : 0ULL);
				if (part && Tr == (uint32_t)k)
					out[k] &= keep;
				n_act = (uint32_t)k + 1u;
			}
			idx += 64u;
			idx = idx >= 127u ? idx - 127u : idx;
		}
		const uint64_t cw = out[0] ^ (sub == 0 ? (uint64_t)crc_seed(p_uap) : 0ULL);
		uint32_t reg = crc_word(crc_word(0, (uint32_t)cw), (uint32_t)(cw >> 32));
		reg = crc_word(crc_word(reg, (uint32_t)out[1]), (uint32_t)(out[1] >> 32));
		reg = crc_word(crc_word(reg, (uint32_t)out[2]), (uint32_t)(out[2] >> 32));
		const uint32_t total = group_xor(apply_columns(col, reg), logg);
		st_n = n_act;
#pragma unroll
		for (int k = 0; k < 3; k++)
			st_val[k] = part && Tr == (uint32_t)k ? out[k] | (oldw & ~keep) : out[k];
		st_pkt = p_pkt;
		if (has && sub == 0)
			outs[p_pkt].payload_rv = total == 0 ? 10 : 2;           // (decode_hits_kernel left a placeholder)
	}
	store();
}

// EV4 (:1044-1097) and EV5 (:1099-1128) payloads of a wave, in a loop of their own (rare types; and what they keep in
// registers -- two more matrices, a prefix over the lanes, eight registers per word -- stays out of long_payloads, whose
// allocation decides the occupancy of decode_hits_kernel).  n_ev list entries from DHL_LIST + 2 first on; G = 8 .. 32 lanes
// per packet, one per payload word.  The first byte count L whose CRC register is zero ends the payload:
//   register in front of word `sub` = A^(64 sub) applied to the XOR over the words j in front of it of A^(-64 (j + 1))
//   (register of word j alone)  -- a per-lane matrix, a plain XOR prefix over the lanes, a per-lane matrix --,
// then the lane's eight bytes one by one, a zero register noted per byte; the lowest lane with a noted byte decides.
// A round's stream words are asked for one round ahead; the record's old last word is read where the length is known.
__device__ __forceinline__ void ev_payloads(dhl_u64_t *stg, dhl_u64_t *lst, uint32_t first, uint32_t n_ev, uint32_t logg, btbbx_pkt_out *outs, uint32_t lane)
{
	dhl_u32_t *const stg32 = (dhl_u32_t *)stg, *const lst32 = (dhl_u32_t *)lst;
	const uint32_t G = 1u << logg, R = 64u >> logg;
	const uint32_t sub = lane & (G - 1), grp = lane >> logg, gbase = grp << logg;
	const uint64_t gmask = (1ULL << G) - 1;                     // (G <= 32: an EV payload has at most 23 words)
	lst[DHL_PB + lane] = 0;
	if (lane < 2)
		stg[128 + lane] = 0;
	const uint32_t wh_lane = (64u * sub) % 127u;
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	const uint32_t rounds = (n_ev + R - 1) >> (6 - logg);
	// this lane's two matrices: A^(-64 (sub + 1)) and A^(64 sub), sixteen 16-bit columns each
	uint32_t rinv[8], rfwd[8];
	{
		const uint4 *si = reinterpret_cast<const uint4 *>(g_adv64inv) + 2 * (sub + 1), *sf = reinterpret_cast<const uint4 *>(g_adv64fwd) + 2 * sub;
		const uint4 a = si[0], b = si[1], c = sf[0], d = sf[1];
		rinv[0] = a.x; rinv[1] = a.y; rinv[2] = a.z; rinv[3] = a.w; rinv[4] = b.x; rinv[5] = b.y; rinv[6] = b.z; rinv[7] = b.w;
		rfwd[0] = c.x; rfwd[1] = c.y; rfwd[2] = c.z; rfwd[3] = c.w; rfwd[4] = d.x; rfwd[5] = d.y; rfwd[6] = d.z; rfwd[7] = d.w;
	}
	// the stream words of round r: EV4 words sub and sub + G of the packet (its blocks all lie inside the capture:
	// min(98, size / 15)); EV5 reads ONE byte, every payload byte is the first one under the whitening of its place
	auto request = [&](uint32_t r, uint64_t &w0, uint64_t &w1) {
		const uint32_t e = r * R + grp;
		w0 = 0;
		w1 = 0;
		if (e < n_ev) {
			const uint64_t a = lst[DHL_LIST + 2 * (first + e)], b = lst[DHL_LIST + 2 * (first + e) + 1];
			dhl_g64_t *const src = (dhl_g64_t *)(uintptr_t)(a & 0xffffffffffffULL);
			const uint32_t p_nw = (uint32_t)(a >> 48) & 127u, p_sh = (uint32_t)(a >> 55) & 63u;
			const bool is4 = ((uint32_t)(b >> 32) & 3u) == DHL_EV4;
			const uint32_t i0 = is4 ? sub : (p_sh + 122u) >> 6, i1 = is4 ? sub + G : i0 + 1u;
			if (is4 || sub == 0) {
				if (i0 < p_nw)
					w0 = src[i0];
				if (i1 < p_nw)
					w1 = src[i1];
			}
		}
	};
	uint64_t nw0, nw1;
	request(0, nw0, nw1);
#pragma unroll 1
	for (uint32_t r = 0; r < rounds; r++) {
		const uint32_t e = r * R + grp;
		const bool has = e < n_ev;
		uint64_t pa = 0, pb = 0;
		if (has) {
			pa = lst[DHL_LIST + 2 * (first + e)];
			pb = lst[DHL_LIST + 2 * (first + e) + 1];
		}
		const uint32_t p_sh = (uint32_t)(pa >> 55) & 63u;
		const uint32_t p_pkt = (uint32_t)pb & 0xffu, nbits = (uint32_t)(pb >> 20) & 0xfffu;
		const uint32_t kind = (uint32_t)(pb >> 32) & 3u, p_widx = (uint32_t)(pb >> 35) & 127u, p_uap = (uint32_t)(pb >> 42) & 0xffu;
		const bool p_wht = (pb >> 34) & 1u, is4 = has && kind == DHL_EV4;
		const uint32_t nblocks = nbits / 10u;                   // (EV4)
		// 1. the stream words asked for a round ago
		const uint64_t w0 = nw0, w1 = nw1;
		const uint32_t sft = (p_sh + 122u) & 63u;
		const uint32_t low8 = (uint32_t)(sft ? (w0 >> sft) | (w1 << (64u - sft)) : w0) & 0xffu;
		const uint32_t first8 = (uint32_t)__shfl((int)low8, (int)gbase);
		uint64_t word = (uint64_t)(first8 * 0x01010101u) | (uint64_t)(first8 * 0x01010101u) << 32;
		// 2. EV4: the (15,10) blocks, as in long_payloads -- and which block is the first that does not decode
		uint32_t first_fail = nblocks;
		if (__ballot(is4)) {
			stg[2 * gbase + sub] = w0;
			stg[2 * gbase + G + sub] = w1;
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			// (four consecutive blocks per lane and step, as in long_payloads)
			for (uint32_t n0 = 0; ; n0 += G) {
				const uint32_t n = n0 + sub;
				const bool on = is4 && 4u * n < nblocks;
				if (!__ballot(on))
					break;
				const uint32_t left = nblocks - 4u * n, have = on ? (left < 4u ? left : 4u) : 0u;
				const uint32_t q = p_sh + 122u + 60u * n, i = 4u * gbase + (on ? q >> 5 : 0u);
				const uint32_t v0 = stg32[i], v1 = stg32[i + 1], v2 = stg32[i + 2];
				const uint64_t vm = (1ULL << (15u * have)) - 1;
				const uint32_t x0 = __builtin_amdgcn_alignbit(v1, v0, q & 31u) & (uint32_t)vm;
				const uint32_t x1 = __builtin_amdgcn_alignbit(v2, v1, q & 31u) & (uint32_t)(vm >> 32);
				const uint32_t b2 = __builtin_amdgcn_alignbit(x1, x0, 30);
				uint32_t d0 = x0 & 0x3ffu, d1 = (x0 >> 15) & 0x3ffu, d2 = b2 & 0x3ffu, d3 = (x1 >> 13) & 0x3ffu;
				const uint32_t m0 = g_lds.fixm23[((x0 >> 10) & 31u) ^ g_lds.par23[d0]], m1 = g_lds.fixm23[((x0 >> 25) & 31u) ^ g_lds.par23[d1]];
				const uint32_t m2 = g_lds.fixm23[((b2 >> 10) & 31u) ^ g_lds.par23[d2]], m3 = g_lds.fixm23[((x1 >> 23) & 31u) ^ g_lds.par23[d3]];
				const uint32_t bad = (m0 >> 15) | (m1 >> 15) << 1 | (m2 >> 15) << 2 | (m3 >> 15) << 3;  // which of the four do not decode
				d0 ^= m0 & 0x3ffu;
				d1 ^= m1 & 0x3ffu;
				d2 ^= m2 & 0x3ffu;
				d3 ^= m3 & 0x3ffu;
				if (on) {
					const uint32_t byte = 5u * n, d = 2u * DHL_PB + 2u * gbase + (byte >> 2);
					const uint64_t v = ((uint64_t)(d3 >> 2) << 32 | (d0 | d1 << 10 | d2 << 20 | d3 << 30)) << (8u * (byte & 3u));
					__hip_atomic_fetch_or(lst32 + d, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					__hip_atomic_fetch_or(lst32 + d + 1, (uint32_t)(v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
				}
				const uint64_t gm = (__ballot(bad != 0) >> gbase) & gmask;
				const uint32_t gl = gm ? (uint32_t)__builtin_ctzll(gm) : 0u;
				const uint32_t gb = (uint32_t)__shfl((int)bad, (int)(gbase + gl));
				if (gm && first_fail == nblocks)
					first_fail = 4u * (n0 + gl) + (uint32_t)__builtin_ctz(gb | 16u);
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
			if (is4)
				word = lst[DHL_PB + lane];
			lst[DHL_PB + lane] = 0;
		}
		if (r + 1 < rounds)
			request(r + 1, nw0, nw1);
		// 3. unwhitened, cut at the bits the decoder may look at
		const uint32_t T = nbits >> 6;
		const bool active = has && 64u * sub < nbits;
		uint64_t out = 0;
		if (active) {
			uint32_t idx = p_widx + wh_lane;
			idx = idx >= 127u ? idx - 127u : idx;
			out = word ^ (p_wht ? wh_bits(idx, 64) : 0ULL);
			if (sub == T)
				out &= (1ULL << (nbits & 63u)) - 1;
		}
		// 4. the register in front of this lane's word
		uint32_t reg;
		{
			const uint64_t cw = out ^ (sub == 0 ? (uint64_t)crc_seed(p_uap) : 0ULL);
			const uint32_t reg0 = crc_word(crc_word(0, (uint32_t)cw), (uint32_t)(cw >> 32));
			const uint32_t q = apply_columns(rinv, reg0);
			uint32_t x = q, t;                                      // inclusive XOR prefix over the lanes of the group
			t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x111, 0xf, 0xf, true); x ^= sub >= 1 ? t : 0u;      // row_shr:1
			t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x112, 0xf, 0xf, true); x ^= sub >= 2 ? t : 0u;
			t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x114, 0xf, 0xf, true); x ^= sub >= 4 ? t : 0u;
			if (logg > 3) {
				t = (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x118, 0xf, 0xf, true); x ^= sub >= 8 ? t : 0u;
			}
			if (logg > 4) {                                         // the second sixteen of a group of 32: + the first sixteen's total
				t = (uint32_t)__shfl((int)x, (int)(gbase + 15));
				x ^= sub >= 16 ? t : 0u;
			}
			reg = apply_columns(rfwd, x ^ q);
		}
		// 5. byte by byte; which byte counts may end the payload: EV4 2 .. 5 (blocks that decoded - 1) / 4 (byte L - 1 is looked
		// at by the loop's step b when 8 L <= 10 b), EV5 3 .. bytes - 1
		const uint32_t ok_blocks = first_fail < nblocks ? first_fail : nblocks;
		const uint32_t hi = is4 ? (ok_blocks ? 5u * (ok_blocks - 1u) / 4u : 0u) : (nbits >> 3) - 1u, lo = is4 ? 2u : 3u;
		uint32_t zero = 0;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			uint32_t byte = (uint32_t)(out >> (8 * i)) & 0xffu;
			if (i == 1 && sub == 0)
				byte ^= crc_seed(p_uap) >> 8;                       // (the seed sits on bits 8 .. 15 of the first word)
			reg = crc_byte(reg, byte);
			const uint32_t L = 8u * sub + (uint32_t)i + 1u;
			if (reg == 0 && L >= lo && L <= hi)
				zero |= 1u << i;
		}
		const uint64_t hm = (__ballot(has && zero != 0) >> gbase) & gmask;
		const uint32_t hl = hm ? (uint32_t)__builtin_ctzll(hm) : 0u;
		const uint32_t hz = (uint32_t)__shfl((int)zero, (int)(gbase + hl));
		const uint32_t L_hit = hm ? 8u * hl + (uint32_t)__builtin_ctz(hz | 0x100u) + 1u : 0u;
		// 6. length, verdict, and the bits the decoder wrote before it stopped
		if (has) {
			uint32_t plen, wbits;
			int rv;
			if (is4) {
				if (L_hit) {
					rv = 10; plen = L_hit; wbits = 10u * ((8u * L_hit + 9u) / 10u + 1u);
				} else {
					plen = hi + 1u; wbits = 10u * ok_blocks;
					rv = ok_blocks == 98u ? 2 : first_fail < nblocks && first_fail < 3u ? 0 : 1;    // all 98 | stopped by an undecodable block in the first 45 symbols | later, or by the capture's end
				}
			} else {
				const uint32_t bytes = nbits >> 3;
				if (L_hit) {
					rv = 10; plen = L_hit; wbits = 8u * (L_hit + 1u);
				} else {
					plen = bytes; wbits = nbits; rv = bytes == 182u ? 2 : 1;
				}
			}
			const uint32_t wT = wbits >> 6, wrem = wbits & 63u;
			if (64u * sub < wbits) {
				uint64_t v = out;
				if (sub == wT) {                                        // (a partial last word keeps what the record held behind it)
					const uint64_t wm = (1ULL << wrem) - 1;
					v = (out & wm) | (outs[p_pkt].payload[sub] & ~wm);
				}
				outs[p_pkt].payload[sub] = v;
			}
			if (sub == 0) {
				outs[p_pkt].payload_length = (int32_t)plen;
				outs[p_pkt].payload_rv = rv;
			}
		}
	}
}

// The deferred payloads of one wave of decode_hits_kernel: dmask = which of its 64 list slots `slots` are filled
// (defer_payload).  DH / DM entries to the front of the LDS list, EV4 / EV5 behind them; lanes per packet = one per
// payload word of the longest payload of either kind (the sort of decode_hits_kernel keeps like with like).  `outs` = the
// records of that workgroup; all 64 lanes.
__device__ __forceinline__ void long_wave(dhl_u64_t *stg, dhl_u64_t *lst, const uint4 *slots, uint64_t dmask, btbbx_pkt_out *outs, uint32_t lane)
{
	const bool mine = (dmask >> lane) & 1;
	uint4 e = make_uint4(0, 0, 0, 0);
	if (mine)
		e = slots[lane];
	const bool ev = mine && (e.w & 3u) >= DHL_EV4;
	const uint64_t ev_mask = __ballot(ev), dh_mask = dmask & ~ev_mask;
	const uint32_t n_dh = (uint32_t)__popcll(dh_mask), n_ev = (uint32_t)__popcll(ev_mask);
	const uint32_t own_words = mine ? (((e.z >> 20) & 0xfffu) + 63u) >> 6 : 0u;
	if (mine) {
		const uint64_t among = ev ? ev_mask : dh_mask;
		const uint32_t rank = (ev ? n_dh : 0u) + __builtin_amdgcn_mbcnt_hi((uint32_t)(among >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)among, 0u));
		lst[DHL_LIST + 2 * rank] = (uint64_t)e.x | (uint64_t)e.y << 32;
		lst[DHL_LIST + 2 * rank + 1] = (uint64_t)e.z | (uint64_t)e.w << 32;
	}
	if (n_dh) {
		const uint32_t w = ev ? 0u : own_words;
		if (__ballot(mine && !ev && (e.w & 3u) == DHL_DM)) {    // (two payload words per lane)
			const uint32_t logg = __ballot(w > 32) ? 5u : __ballot(w > 16) ? 4u : 3u;
			long_payloads(stg, lst, n_dh, logg, outs, lane);
		} else {                                                // DH only: three
			dh_payloads(lst, n_dh, __ballot(w > 24) ? 4u : 3u, outs, lane);
		}
	}
	if (n_ev) {
		const uint32_t w = ev ? own_words : 0u;
		const uint32_t logg = __ballot(w > 16) ? 5u : __ballot(w > 8) ? 4u : 3u;
		ev_payloads(stg, lst, n_dh, n_ev, logg, outs, lane);
	}
}

#define DH_WAVES_PER_EU 6
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DH_WAVES_PER_EU, DH_WAVES_PER_EU)))
void decode_hits_kernel(const uint64_t *words, uint64_t n_words, uint64_t pitch_words,
							  const btbbx_hit *hits, const btbbx_pkt_in *in, uint32_t n_packets,
							  const uint32_t *d_count, uint32_t max_length, btbbx_pkt_out *outs,
							  uint32_t *lengths, uint32_t mode, btbbx_pkt_in one_in, uint32_t clk_div,
							  uint4 *long_list)
{
	__shared__ uint64_t stage[4][DH_STAGE_WORDS];
	__shared__ uint64_t ostage[4][64 * DH_OUT_WORDS];
	chain_lds_init();
	uint32_t pkt = blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (d_count)                                        // the list's length lives in HBM (no host round trip): n_packets is its capacity
		n_packets = min(n_packets, *d_count);
	bool live = pkt < n_packets;
	if (blockIdx.x * blockDim.x >= n_packets)
		return;
#ifdef DH_PROFILE
	uint32_t dh_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	uint64_t dh_t;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dh_t) : : "memory");
#endif
	btbbx_hit h;
	h.offset = 0;
	h.stream = 0;
	if (live)
		h = hits[pkt];
	const uint64_t total_bits = n_words * 64;
	const uint64_t avail = h.offset < total_bits ? total_bits - h.offset : 0;
	uint32_t len = avail < max_length ? (uint32_t)avail : max_length;
	if (len > BTBBX_MAX_SYMBOLS)
		len = BTBBX_MAX_SYMBOLS;
	const uint64_t first_word = h.offset >> 6;
	PState s;
	s.w = words + (uint64_t)h.stream * pitch_words + first_word;
	s.sh = (uint32_t)(h.offset & 63);
	s.wlimit = first_word < n_words ? (uint32_t)(n_words - first_word < 64 ? n_words - first_word : 64) : 0;
	s.direct = true;
	s.length = live ? (int)len : 0;
	btbbx_pkt_in pi;
	pi.length = 0; pi.clkn = 0; pi.flags = 0; pi.uap = 0; pi.type = 0; pi.llid = 0; pi.flow = 0;
	if (live) {
		if (in) {
			pi = in[pkt];
		} else {
			// a capture of one piconet: every packet enters with the same state, its clock follows from where it was found
			// (CLK1-27 advances once per clk_div symbols: 625 at 1 Msym/s)
			// (one_in.length = the symbols of the current slot that had already passed at the buffer's first symbol)
			pi = one_in;
			const uint64_t since = h.offset + one_in.length;
			pi.clkn = one_in.clkn + (since >> 32 ? (uint32_t)(since / clk_div) : (uint32_t)since / clk_div);
		}
	}
	pi.length = len;

	asm volatile("" : "+v"(pi.clkn), "+v"(len));
	DH_MARK(0);                                         // hit + btbbx_pkt_in loaded
	// how much of the packet the decoders can want: the type the header yields under this packet's clock
	uint32_t want = 0, dtype = 0;
	bool small = false, wide = false;                   // its payload fits DH_OUT_WORDS words / needs more than DH_OUT_SECTOR
	uint32_t hdr = 0, dis = 0;
	typedef __attribute__((address_space(3))) uint64_t lds_u64_t;
	{
		// the header and the payload header (symbols 68 .. 232 of the packet) are in its words 1 .. 4: four loads in
		// flight together, parked in the input stage, instead of one s_bits() after the other going to the stream
		uint64_t hw[4];
#pragma unroll
		for (uint32_t k = 0; k < 4; k++)
			hw[k] = live && k + 1 < s.wlimit ? s.w[k + 1] : 0ULL;
#pragma unroll
		for (uint32_t k = 0; k < 4; k++)
			stage[wave][lane * 5 + k + 1] = hw[k];
		s.staged = s.wlimit < 5 ? s.wlimit : 5;
		s.stage_off = (uint32_t)(uintptr_t)(lds_u64_t *)(&stage[wave][lane * 5]);
	}
	if (live) {
		want = len < 126 ? len : 126;
		s.flags = pi.flags;
		hdr = header_fec13(s, dis);
		if ((mode & DEC_PAYLOAD) && len > 126) {
			uint32_t type = pi.type;
			if (mode & DEC_HEADER)
				type = ((hdr ^ (uint32_t)wh(s, wh_start(pi.clkn, 0), 18)) >> 3) & 0xf;
			const uint32_t bound = payload_extent(s, type, pi.clkn, small, wide);
			want = len < bound ? len : bound;
			dtype = type;
		}
	}
	asm volatile("" : "+v"(want));
	s.staged = 0;
	if (live && lengths)
		lengths[pkt] = len;
	// The workgroup's 256 packets change hands so that a wave decodes packets of one kind and about one length: a wave
	// with DM, DH and FHS packets in it runs the three decoders one after the other with a third of its lanes each,
	// and a loop over FEC blocks runs as long as its longest packet.  (With the stores, the staging and the exact
	// extents fixed the kernel issues vector instructions 68 % of the time, profiles/r03_chain/pmc_decode_mid.json;
	// while it sat in s_waitcnt the same sort gained nothing.)  Counting sort on (decoder, symbols wanted); what a
	// thread knows about its packet goes to the thread that takes it over through the input stage, which is still empty.
	{
		// (the counters and the permutation live in ostage, which nothing uses before the sort is over)
		uint32_t *const sort_cnt = reinterpret_cast<uint32_t *>(&ostage[0][0]);
		uint8_t *const perm = reinterpret_cast<uint8_t *>(&ostage[0][32]);
		uint64_t *const xch = &stage[0][0];
		const uint32_t tid = threadIdx.x;
		if (tid < 64)
			sort_cnt[tid] = 0;
		__syncthreads();
		uint32_t key = 63;
		if (live) {
			const uint32_t cls = want <= 126 ? 0 : decoder_of_type(dtype);
			const uint32_t lb = want <= 126 ? 0 : (want - 122) >> 5;
			key = cls * 8 + (lb < 7 ? lb : 7);
			// payloads that go to the wave phase (long_payloads): together, by the lanes a packet takes there (keys that
			// are all but unused otherwise: class 0 has one length, HV packets that are not cut short another)
			if (!small && want > 126 && (cls == 2 || cls == 3)) {
				const uint32_t pbits = cls == 2 ? (want - 122) / 15 * 10 : want - 122, words = (pbits + 63) >> 6;   // (about: the grouping only)
				key = cls == 2 ? 32u + (words > 32 ? 2u : words > 16 ? 1u : 0u)          // (DM, two words per lane: groups of 32 / 16 / 8)
					       : 1u + (words > 24 ? 1u : 0u);                                // (DH, three: 16 / 8)
			}
		}
		const uint32_t r = atomicAdd(&sort_cnt[key], 1u);
		xch[tid] = h.offset;
		xch[256 + tid] = (uint64_t)h.stream | (uint64_t)want << 16 | (uint64_t)wide << 28 | (uint64_t)small << 30 | (uint64_t)live << 31 | (uint64_t)pi.clkn << 32;
		xch[512 + tid] = (uint64_t)pi.flags | (uint64_t)pi.uap << 32 | (uint64_t)pi.type << 40 | (uint64_t)pi.llid << 48 | (uint64_t)pi.flow << 56;
		xch[768 + tid] = (uint64_t)hdr | (uint64_t)dtype << 24 | (uint64_t)dis << 32;
		__syncthreads();
		uint32_t c = sort_cnt[lane], incl = c;
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t u = __shfl_up(incl, d);
			if (lane >= (uint32_t)d)
				incl += u;
		}
		perm[__shfl(incl - c, key) + r] = (uint8_t)tid;
		__syncthreads();
		const uint32_t q = perm[tid];
		const uint64_t x0 = xch[q], x1 = xch[256 + q], x2 = xch[512 + q], x3 = xch[768 + q];
		__syncthreads();                                // the stage is free again
		pkt = blockIdx.x * blockDim.x + q;
		h.offset = x0;
		h.stream = (uint16_t)x1;
		want = (uint32_t)(x1 >> 16) & 0xfff;            // <= 3125
		wide = (x1 >> 28) & 1;
		small = (x1 >> 30) & 1;
		live = (x1 >> 31) & 1;
		pi.clkn = (uint32_t)(x1 >> 32);
		pi.flags = (uint32_t)x2;
		pi.uap = (uint8_t)(x2 >> 32); pi.type = (uint8_t)(x2 >> 40); pi.llid = (uint8_t)(x2 >> 48); pi.flow = (uint8_t)(x2 >> 56);
		const uint64_t avail2 = h.offset < total_bits ? total_bits - h.offset : 0;
		len = avail2 < max_length ? (uint32_t)avail2 : max_length;
		if (len > BTBBX_MAX_SYMBOLS)
			len = BTBBX_MAX_SYMBOLS;
		const uint64_t fw = h.offset >> 6;
		s.w = words + (uint64_t)h.stream * pitch_words + fw;
		s.sh = (uint32_t)(h.offset & 63);
		s.wlimit = fw < n_words ? (uint32_t)(n_words - fw < 64 ? n_words - fw : 64) : 0;
		s.length = live ? (int)len : 0;
		pi.length = len;
		hdr = (uint32_t)x3 & 0x3ffffu;
		dtype = (uint32_t)(x3 >> 24) & 0xfu;
		dis = (uint32_t)(x3 >> 32);
	}
	s.has_pre = true;
	s.pre_hdr = hdr;
	s.pre_dis = dis;
	DH_MARK(1);                                         // header read from the stream, type known
	// Results leave through LDS.  A lane storing its own packet's words touches 64 different sectors per instruction
	// (the phase after the decoders was 29 % of the wave time, 6 % now).  The payload words of a packet that writes
	// <= 256 bits (FHS 160, DM1 / DH1 / AUX1 / DV <= 240, HV 240, EV3 256, short multi-slot packets) are collected in
	// ostage, the head in the input stage once every lane is done reading it, and the wave stores head + payload of
	// packet after packet as consecutive words.  ostage starts from what the record holds, so bits the decoders leave
	// alone stay.  Head + three payload words = the record's first 64-byte sector; the fourth word (`wide` packets
	// only) is in the second.
	const uint64_t small_mask = __ballot(small), wide_mask = __ballot(wide), live_mask = __ballot(live);
	// the record's head (entry state of the decoders): on its way while the packets are staged
	uint64_t head_in[5] = {0, 0, 0, 0, 0};
	if (live) {
#pragma unroll
		for (int k = 0; k < 5; k++)
			head_in[k] = reinterpret_cast<const uint64_t *>(outs + pkt)[k];
	}
	uint32_t nw = live ? (s.sh + want + 63) / 64 : 0;              // words of the stream that hold those symbols
	if (nw > s.wlimit)
		nw = s.wlimit;
	// a payload that will be left to the wave phase: its lane reads the header and the payload header, four words
	if (live && !small && want > 126 && nw > 4 && (decoder_of_type(dtype) == 2 || decoder_of_type(dtype) == 3))
		nw = 4;
	// LDS slots in lane order; a packet that does not fit the wave's budget any more stays in the stream
	uint32_t before = nw;
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t t = __shfl_up(before, d);
		if (lane >= (uint32_t)d)
			before += t;
	}
	before -= nw;
	if (before + nw > DH_STAGE_WORDS)
		nw = 0;
	const uint32_t stage_base = (uint32_t)(uintptr_t)(lds_u64_t *)(&stage[wave][0]);
	// The words go from HBM to LDS without passing through registers (global_load_lds_dword: the wave's LDS base is
	// uniform, lane i fills dword i): packet j of the wave is one instruction -- lanes below twice its word count --
	// and all 64 packets' loads are in flight together, one HBM latency per wave.  (Round 3 first staged one packet at
	// a time through registers -- the wave sat out 64 latencies in a row, 80 % of its life in s_waitcnt,
	// profiles/r03_chain/pmc_decode_before.json -- then sixteen at a time, which cost 48 registers.)
	typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
	typedef __attribute__((address_space(1))) const uint32_t glb_u32_t;
	{
		// what the records hold in the payload words the small packets will leave through ostage
		lds_u32_t *const obase = (lds_u32_t *)(lds_u64_t *)(&ostage[wave][0]);
#pragma unroll
		for (uint32_t t = 0; t < 2 * DH_OUT_WORDS; t++) {
			const uint32_t f = t * 64 + lane, p = f / (2 * DH_OUT_WORDS), k = f % (2 * DH_OUT_WORDS);
			const uint32_t pkt_p = __shfl(pkt, p);
			if (((small_mask >> p) & 1) && (k < 2 * DH_OUT_SECTOR || ((wide_mask >> p) & 1)))
				__builtin_amdgcn_global_load_lds((glb_u32_t *)(uintptr_t)(reinterpret_cast<const uint32_t *>(outs + pkt_p) + 10 + k),
								 obase + t * 64, 4, 0, 0);
		}
	}
	{
		// The staged packets lie back to back in the wave's stage, so the stage is one run of dwords and instruction
		// i fills dwords 64 i .. 64 i + 63 of it, whichever packets they belong to: every packet first writes its lane
		// number into the slots it will get, the lane that loads dword D reads the owner from there and takes the
		// owner's stream address.  (One instruction per packet was 64 rounds of readlanes and compares: 820 of the
		// kernel's 2 700 vector instructions per wave.)
		lds_u32_t *const sbase = (lds_u32_t *)(lds_u64_t *)(&stage[wave][0]);
		const uint64_t staged_mask = __ballot(nw > 0);
		const uint32_t last = staged_mask ? 63u - (uint32_t)__builtin_clzll(staged_mask) : 0u;
		const uint32_t total2 = staged_mask ? 2u * (uint32_t)__builtin_amdgcn_readlane(before + nw, last) : 0u;
		for (uint32_t k = 0; __ballot(k < nw); k++)
			if (k < nw)
				sbase[2 * (before + k)] = lane;
		const uint64_t adj = (uint64_t)(uintptr_t)s.w - 8ull * before;        // dword D of the stage is at adj + 4 D
		for (uint32_t d0 = 0; d0 < total2; d0 += 64) {
			const uint32_t d = d0 + lane;
			const uint32_t owner = d < total2 ? sbase[d & ~1u] : 0u;
			const uint64_t a = __shfl(adj, owner) + 4ull * d;            // (every lane takes part in the shuffle)
			if (d < total2)
				__builtin_amdgcn_global_load_lds((glb_u32_t *)(uintptr_t)a, sbase + d0, 4, 0, 0);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
	__builtin_amdgcn_wave_barrier();
	s.staged = nw;
	s.stage_off = stage_base + 8u * before;
	DH_MARK(2);                                         // packets staged

	uint64_t head[5] = {0, 0, 0, 0, 0};
	if (long_list) {
		s.def_slot = long_list + ((size_t)(blockIdx.x * 4 + wave) * 64 + lane);
		s.def_pkt8 = pkt - blockIdx.x * blockDim.x;
	}
	if (live)
		decode_view(s, pi, outs + pkt, mode,
			    small ? OutRef::lds((uint32_t)(uintptr_t)(lds_u64_t *)(&ostage[wave][lane * DH_OUT_WORDS])) : OutRef(), head, head_in DH_PASS);
	__builtin_amdgcn_wave_barrier();                    // every lane is done with the staged packets
	// which of the wave's 64 list slots hold a payload that was left for later (do_DM / do_DH, defer_payload)
	const uint64_t long_mask = long_list ? __ballot(live && s.def_nbits != 0) : 0ULL;
	const uint64_t keep_mask = __ballot(small && !s.spoiled);
#pragma unroll
	for (int k = 0; k < 5; k++)
		stage[wave][lane * 5 + k] = head[k];
	__builtin_amdgcn_wave_barrier();
#pragma unroll
	for (uint32_t t = 0; t < 5 + DH_OUT_WORDS; t++) {
		const uint32_t f = t * 64 + lane, p = f / (5 + DH_OUT_WORDS), k = f % (5 + DH_OUT_WORDS);
		const uint32_t pkt_p = __shfl(pkt, p);
		if ((live_mask >> p) & 1) {
			uint64_t *dst = reinterpret_cast<uint64_t *>(outs + pkt_p);
			if (k < 5)
				dst[k] = stage[wave][p * 5 + k];
			else if (((keep_mask >> p) & 1) && (k - 5 < DH_OUT_SECTOR || ((wide_mask >> p) & 1)))
				dst[k] = ostage[wave][p * DH_OUT_WORDS + k - 5];
		}
	}
	DH_MARK(7);                                         // decoded, results stored
	if (__builtin_expect(long_mask != 0, 0)) {
		// The payloads the lanes left alone, a group of lanes per packet (long_payloads), in the wave's input stage: behind
		// the store phase, when nothing of the lanes' decoders is alive any more.  Fused into this kernel rather than run
		// as a kernel of its own behind it (round 4 measured both): the phase is bound by instruction issue, the lanes' phases by latency -- waves
		// in the one fill the gaps of waves in the other (DH5 at full length: 497 against 562 us per 1.29 M packets).  What
		// is known about a packet comes back from the list its lane wrote (defer_payload): 16 bytes, still in the L2.
		asm volatile("s_waitcnt vmcnt(0)" : : : "memory");            // the list entries are written
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// (nothing of the lanes' phase is handed over in vector registers: lane number and wave number are made afresh, so no
		// value computed for the long phase is kept alive through the decoders)
		uint32_t lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		asm volatile("" : "+v"(lane2));
		uint32_t wave2 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
		asm volatile("" : "+s"(wave2));
		long_wave((dhl_u64_t *)(lds_u64_t *)(&stage[wave2][0]), (dhl_u64_t *)(lds_u64_t *)(&ostage[wave2][0]), long_list + (size_t)(blockIdx.x * 4 + wave2) * 64, long_mask,
			  outs + (size_t)blockIdx.x * blockDim.x, lane2);
	}
#ifdef DH_PROFILE
	if (lane == 0)
		for (int k = 0; k < 8; k++)
			atomicAdd(&g_dh_prof[k], (unsigned long long)dh_acc[k]);
#endif
}

// 64 symbols, one per byte (bit 0 counts), -> one packed word
__device__ __forceinline__ uint64_t pack64(const uint8_t *sym)
{
	uint64_t v = 0;
	const uint4 *p = reinterpret_cast<const uint4 *>(sym);
	for (int q = 0; q < 4; q++) {
		const uint4 x = p[q];
		const uint32_t d[4] = {x.x, x.y, x.z, x.w};
		for (int k = 0; k < 4; k++) {
			const uint32_t b = d[k] & 0x01010101u;
			v |= (uint64_t)((b | (b >> 7) | (b >> 14) | (b >> 21)) & 0xfu) << (16 * q + 4 * k);
		}
	}
	return v;
}

// The drop-in's single-packet decode in ONE launch: the symbol bytes and the current payload bit
// bytes are packed by the workgroup, lane 0 decodes, and the payload bits are unpacked again --
// instead of pack + pack + decode + unpack launches around a one-lane kernel.
__global__ __launch_bounds__(64) void decode_bytes_kernel(const uint8_t *sym, uint8_t *pay, const btbbx_pkt_in *in,
							   btbbx_pkt_out *o, uint32_t mode, int with_payload)
{
	__shared__ uint64_t pkt[BTBBX_PKT_WORDS + 2];
	const uint32_t lane = threadIdx.x;
	if (lane < BTBBX_PKT_WORDS + 2)
		pkt[lane] = lane < BTBBX_PKT_WORDS ? pack64(sym + 64 * lane) : 0;      // 3200 staged bytes
	if (with_payload && lane < 43)
		o->payload[lane] = pack64(pay + 64 * lane);                            // 2752 staged bytes
	chain_lds_init();
	if (lane == 0)
		decode_one(pkt, in[0], o, mode);
	__syncthreads();
	if (with_payload && lane < 43) {
		const uint64_t v = o->payload[lane];
		for (int k = 0; k < 64; k += 4) {
			const uint32_t n = (uint32_t)(v >> k) & 0xf;
			*reinterpret_cast<uint32_t *>(pay + 64 * lane + k) = (n * 0x00204081u) & 0x01010101u;
		}
	}
}

// DEC_TRIALS for one packet, 64 trials at once: lane = candidate count, every lane starts from the
// entry state and writes into a private payload buffer; what the reference's sequential loop
// (bluetooth_piconet.c:675-690) leaves in the packet is then "last writer wins" per field and per
// payload bit, taken in lane order.  Valid because a trial never reads what an earlier trial
// wrote: try_clock + crc_check(c) depend on the entry UAP / type only when FEC 1/3 fails (then for
// every clock alike), the payload header merge is bitwise, and EV4's llid / flow read never
// decides anything (see the identities at the top of this file).
__global__ __launch_bounds__(64) void replay_kernel(const uint64_t *packet, const btbbx_pkt_in *in, btbbx_pkt_out *o,
						     TrialPlan plan)
{
	__shared__ uint64_t pay[64][44];
	__shared__ uint32_t wrote[64];
	chain_lds_init();
	const uint32_t lane = threadIdx.x;
	const btbbx_pkt_in pi = in[0];
	PState s;
	s.w = packet;
	s.length = (int)pi.length;
	s.flags = pi.flags;
	s.uap = pi.uap;
	s.type = pi.type;
	s.llid = pi.llid;
	s.flow = pi.flow;
	s.plen = o->payload_length;
	s.phl = o->payload_header_length;
	s.ph16 = (uint32_t)o->payload_header;
	s.ph_written = 0;
	s.dirty = 0;
	s.ph_mask = 0;
	s.lt_addr = o->lt_addr; s.hdr_flags = o->hdr_flags; s.hec = o->hec; s.header18 = o->header_packed;
	s.out = pay[lane];
	s.written = 0;
	for (int j = 0; j < 44; j++)
		pay[lane][j] = 0;
	const uint32_t entry_flags = s.flags;

	const bool do_try = (plan.try_mask >> lane) & 1, do_crc = (plan.crc_mask >> lane) & 1;
	const uint32_t clock = (lane + plan.clock_offset) & 63;
	uint32_t dis;
	const uint32_t hdr = header_fec13(s.w, dis);
	int header_rv = 0, payload_rv = 0;
	if (do_try)
		header_rv = (int)do_try_clock(s, clock, hdr, dis);
	if (do_crc)
		payload_rv = do_crc_check<true>(s, clock);
	wrote[lane] = s.written;
	__syncthreads();

	// scalar fields: the highest lane that assigned them
	auto last = [&](bool mine) { const uint64_t m = __ballot(mine); return m ? 63 - (int)__builtin_clzll(m) : -1; };
	const int l_ut = last(s.dirty & D_UT), l_plen = last(s.dirty & D_PLEN), l_phl = last(s.dirty & D_PHL);
	const int l_lf = last(s.dirty & D_LF), l_ph8 = last(s.ph_mask & 0xff), l_ph16 = last(s.ph_mask & 0xff00);
	const int l_try = last(do_try), l_crc = last(do_crc);
	const uint32_t f_uap = l_ut >= 0 ? (uint32_t)__shfl((int)s.uap, l_ut) : pi.uap;
	const uint32_t f_type = l_ut >= 0 ? (uint32_t)__shfl((int)s.type, l_ut) : pi.type;
	const int f_plen = l_plen >= 0 ? __shfl(s.plen, l_plen) : o->payload_length;
	const int f_phl = l_phl >= 0 ? __shfl(s.phl, l_phl) : o->payload_header_length;
	const uint32_t f_llid = l_lf >= 0 ? (uint32_t)__shfl((int)s.llid, l_lf) : pi.llid;
	const uint32_t f_flow = l_lf >= 0 ? (uint32_t)__shfl((int)s.flow, l_lf) : pi.flow;
	uint32_t f_ph = (uint32_t)o->payload_header;
	if (l_ph8 >= 0)
		f_ph = (f_ph & ~0xffu) | ((uint32_t)__shfl((int)s.ph16, l_ph8) & 0xffu);
	if (l_ph16 >= 0)
		f_ph = (f_ph & ~0xff00u) | ((uint32_t)__shfl((int)s.ph16, l_ph16) & 0xff00u);
	uint32_t f_flags = s.flags & ~entry_flags;                  // bits this trial added (HAS_PAYLOAD)
	for (int d = 32; d; d >>= 1)
		f_flags |= (uint32_t)__shfl_xor((int)f_flags, d);
	f_flags |= entry_flags;
	const int f_hrv = l_try >= 0 ? __shfl(header_rv, l_try) : 0;
	const int f_prv = l_crc >= 0 ? __shfl(payload_rv, l_crc) : 0;

	// payload: word j takes, from the highest lane down, the bits that lane's prefix covers
	if (lane < 43) {
		uint64_t word = o->payload[lane], undecided = ~0ULL;
		for (int k = 63; k >= 0 && undecided; k--) {
			const uint32_t w = wrote[k];
			if (w <= 64u * lane)
				continue;
			const uint32_t nb = w - 64u * lane;
			const uint64_t covers = (nb >= 64 ? ~0ULL : ((1ULL << nb) - 1)) & undecided;
			word = (word & ~covers) | (pay[k][lane] & covers);
			undecided &= ~covers;
		}
		o->payload[lane] = word;
	}
	if (lane == 0) {
		o->header_present = (uint8_t)do_header_present(s);
		o->header_rv = f_hrv;
		o->payload_rv = f_prv;
		o->payload_length = f_plen;
		o->payload_header_length = f_phl;
		o->flags = f_flags;
		o->type = (uint8_t)f_type;
		o->llid = (uint8_t)f_llid;
		o->flow = (uint8_t)f_flow;
		o->uap = (uint8_t)f_uap;
		o->payload_header = f_ph;
	}
}

// The same for btbb_uap_from_header in two steps, so that the 64 trials run once: step 1 runs every
// trial with its writes captured per lane (TrialState in global memory) and returns the
// {try_clock, type, crc_check} table; the host then eliminates candidates exactly like the
// reference and hands back which trials the reference would have executed; step 2 merges those.
struct TrialState {
	uint32_t dirty, ph16, ph_mask, flags_added, written;
	int32_t plen, phl;
	uint8_t uap, type, llid, flow;
	uint64_t payload[44];
};

__global__ __launch_bounds__(64) void trials_state_kernel(const uint8_t *sym, const btbbx_pkt_in *in,
							   const btbbx_pkt_out *o, TrialState *st, btbbx_trial *trials)
{
	// every workgroup packs the 3200 staged symbol bytes for itself (LDS): no separate pack launch
	__shared__ uint64_t packet[BTBBX_PKT_WORDS + 2];
	if (threadIdx.x < BTBBX_PKT_WORDS + 2)
		packet[threadIdx.x] = threadIdx.x < BTBBX_PKT_WORDS ? pack64(sym + 64 * threadIdx.x) : 0;
	// one workgroup per candidate clock: 64 waves on 64 CUs each run ONE trial (no divergence between
	// packet types inside a wave), so the latency of the call is that of the longest single trial
	// instead of the sum over all types a 64-lane wave would have to serialise
	chain_lds_init();
	if (threadIdx.x)
		return;
	const uint32_t lane = blockIdx.x;
	const btbbx_pkt_in pi = in[0];
	TrialState *me = st + lane;
	PState s;
	s.w = packet;
	s.length = (int)pi.length;
	s.flags = pi.flags;
	s.uap = pi.uap;
	s.type = pi.type;
	s.llid = pi.llid;
	s.flow = pi.flow;
	s.plen = o->payload_length;
	s.phl = o->payload_header_length;
	s.ph16 = (uint32_t)o->payload_header;
	s.ph_written = 0;
	s.dirty = 0;
	s.ph_mask = 0;
	s.lt_addr = o->lt_addr; s.hdr_flags = o->hdr_flags; s.hec = o->hec; s.header18 = o->header_packed;
	s.out = me->payload;
	s.written = 0;
	for (int j = 0; j < 44; j++)
		me->payload[j] = 0;
	uint32_t dis;
	const uint32_t hdr = header_fec13(s.w, dis);
	const uint32_t uap = do_try_clock(s, lane, hdr, dis);
	const int rv = do_crc_check<true>(s, lane);
	btbbx_trial t;
	t.uap = (uint8_t)uap;
	t.type = (uint8_t)s.type;
	t.rv = (int16_t)rv;
	trials[lane] = t;
	me->dirty = s.dirty;
	me->ph16 = s.ph16;
	me->ph_mask = s.ph_mask;
	me->flags_added = s.flags & ~pi.flags;
	me->written = s.written;
	me->plen = s.plen;
	me->phl = s.phl;
	me->uap = (uint8_t)s.uap;
	me->type = (uint8_t)s.type;
	me->llid = (uint8_t)s.llid;
	me->flow = (uint8_t)s.flow;
}

// lane = candidate count; the trial it stands for ran with clock (count + clock_offset) & 63
__global__ __launch_bounds__(64) void trials_merge_kernel(const TrialState *st, const btbbx_pkt_in *in, btbbx_pkt_out *o,
							   uint8_t *pay, TrialPlan plan)
{
	__shared__ uint32_t wrote[64];
	__shared__ uint32_t src_of[64];
	const uint32_t lane = threadIdx.x;
	const btbbx_pkt_in pi = in[0];
	const TrialState *me = st + ((lane + plan.clock_offset) & 63);
	const bool did_try = (plan.try_mask >> lane) & 1, did_crc = (plan.crc_mask >> lane) & 1;
	const uint32_t dirty = (did_try ? me->dirty & D_UT : 0u) | (did_crc ? me->dirty & ~D_UT : 0u);
	const uint32_t ph_mask = did_crc ? me->ph_mask : 0u;
	wrote[lane] = did_crc ? me->written : 0u;
	src_of[lane] = (lane + plan.clock_offset) & 63;
	__syncthreads();
	auto last = [&](bool mine) { const uint64_t m = __ballot(mine); return m ? 63 - (int)__builtin_clzll(m) : -1; };
	const int l_ut = last(dirty & D_UT), l_plen = last(dirty & D_PLEN), l_phl = last(dirty & D_PHL);
	const int l_lf = last(dirty & D_LF), l_ph8 = last(ph_mask & 0xff), l_ph16 = last(ph_mask & 0xff00);
	uint32_t f_flags = did_crc ? me->flags_added : 0u;
	for (int d = 32; d; d >>= 1)
		f_flags |= (uint32_t)__shfl_xor((int)f_flags, d);
	if (lane < 43) {
		// entry payload bits come in, and the merged ones go out, one per byte (`pay`, 2752 bytes)
		uint64_t word = pack64(pay + 64 * lane), undecided = ~0ULL;
		for (int k = 63; k >= 0 && undecided; k--) {
			const uint32_t w = wrote[k];
			if (w <= 64u * lane)
				continue;
			const uint32_t nb = w - 64u * lane;
			const uint64_t covers = (nb >= 64 ? ~0ULL : ((1ULL << nb) - 1)) & undecided;
			word = (word & ~covers) | (st[src_of[k]].payload[lane] & covers);
			undecided &= ~covers;
		}
		o->payload[lane] = word;
		for (int k = 0; k < 64; k += 4)
			*reinterpret_cast<uint32_t *>(pay + 64 * lane + k) = (((uint32_t)(word >> k) & 0xf) * 0x00204081u) & 0x01010101u;
	}
	if (lane == 0) {
		auto at = [&](int l) { return st + ((l + plan.clock_offset) & 63); };
		if (l_ut >= 0) { o->uap = at(l_ut)->uap; o->type = at(l_ut)->type; } else { o->uap = pi.uap; o->type = pi.type; }
		if (l_plen >= 0) o->payload_length = at(l_plen)->plen;
		if (l_phl >= 0) o->payload_header_length = at(l_phl)->phl;
		if (l_lf >= 0) { o->llid = at(l_lf)->llid; o->flow = at(l_lf)->flow; } else { o->llid = pi.llid; o->flow = pi.flow; }
		uint32_t ph = (uint32_t)o->payload_header;
		if (l_ph8 >= 0) ph = (ph & ~0xffu) | (at(l_ph8)->ph16 & 0xffu);
		if (l_ph16 >= 0) ph = (ph & ~0xff00u) | (at(l_ph16)->ph16 & 0xff00u);
		o->payload_header = ph;
		o->flags = pi.flags | f_flags;
		o->header_rv = 0;
		o->payload_rv = 0;
	}
}

int launch_trials_state(const uint8_t *d_sym, const btbbx_pkt_in *d_in, const btbbx_pkt_out *d_out, void *d_state,
			btbbx_trial *d_trials, hipStream_t stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	hipLaunchKernelGGL(trials_state_kernel, dim3(64), dim3(64), 0, stream, d_sym, d_in, d_out, (TrialState *)d_state, d_trials);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

int launch_trials_merge(const void *d_state, const btbbx_pkt_in *d_in, btbbx_pkt_out *d_out, uint8_t *d_pay,
			const TrialPlan *plan, hipStream_t stream)
{
	hipLaunchKernelGGL(trials_merge_kernel, dim3(1), dim3(64), 0, stream, (const TrialState *)d_state, d_in, d_out, d_pay,
			   *plan);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

size_t trials_state_bytes() { return 64 * sizeof(TrialState); }

// cut packets out of the packed streams
// five packets per 256-thread workgroup: thread -> (packet, output word); the output rows of a workgroup are
// contiguous (5 x 400 bytes), so its stores coalesce across packets
#define GATHER_PACKETS (256 / BTBBX_PKT_WORDS)
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *words, uint64_t n_words, uint64_t pitch_words,
						      const btbbx_hit *hits, uint32_t n_packets, uint32_t max_length,
						      uint64_t *packets, uint32_t *lengths)
{
	uint32_t pkt = blockIdx.x * GATHER_PACKETS + threadIdx.x / BTBBX_PKT_WORDS;
	uint32_t i = threadIdx.x % BTBBX_PKT_WORDS;          // output word
	if (pkt >= n_packets || threadIdx.x >= GATHER_PACKETS * BTBBX_PKT_WORDS)
		return;
	const btbbx_hit h = hits[pkt];
	const uint64_t *base = words + (uint64_t)h.stream * pitch_words;
	uint64_t total_bits = n_words * 64;
	uint64_t avail = h.offset < total_bits ? total_bits - h.offset : 0;
	uint32_t len = avail < max_length ? (uint32_t)avail : max_length;
	if (len > BTBBX_MAX_SYMBOLS)
		len = BTBBX_MAX_SYMBOLS;
	uint64_t bit = h.offset + 64ULL * i;
	uint64_t j = bit >> 6;
	uint32_t sh = (uint32_t)(bit & 63);
	uint64_t lo = j < n_words ? base[j] : 0, hi = (j + 1) < n_words ? base[j + 1] : 0;
	uint64_t v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
	// zero everything at and beyond `len`
	uint32_t first = 64 * i;
	if (first >= len)
		v = 0;
	else if (len - first < 64)
		v &= (1ULL << (len - first)) - 1;
	packets[(uint64_t)pkt * BTBBX_PKT_WORDS + i] = v;
	if (i == 0)
		lengths[pkt] = len;
}

// ---- launchers --------------------------------------------------------------------------------

extern "C" int btbbx_gather_packets_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
					   const btbbx_hit *d_hits, uint32_t n_packets, uint32_t max_length,
					   uint64_t *d_packets, uint32_t *d_lengths, void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!n_packets)
		return BTBBX_OK;
	hipLaunchKernelGGL(gather_kernel, dim3((n_packets + GATHER_PACKETS - 1) / GATHER_PACKETS), dim3(256), 0,
			   (hipStream_t)hip_stream, d_words, n_words, pitch_words, d_hits, n_packets, max_length, d_packets, d_lengths);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_trials_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
				   btbbx_trial *d_trials, void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!n_packets)
		return BTBBX_OK;
	if (n_packets <= 256) {      // latency shape while 64 n waves are only a few rounds over the chip
		hipLaunchKernelGGL(trials_wide_kernel, dim3(n_packets * 64), dim3(64), 0, (hipStream_t)hip_stream,
				   d_packets, d_in, n_packets, d_trials);
	} else {                     // per-packet FEC / CRC prefix work once, O(1) per DM / DH / FHS trial
		const uint32_t batches = (n_packets + TL_PACKETS - 1) / TL_PACKETS;
		const uint32_t resident = (uint32_t)ctx().num_cus * TL_WGS_PER_CU;
		hipLaunchKernelGGL(trials_linear_kernel, dim3(batches < resident ? batches : resident), dim3(TL_THREADS), 0,
				   (hipStream_t)hip_stream, d_packets, d_in, n_packets, d_trials);
	}
#ifdef TL_PROFILE
	if (n_packets > 256) {
		unsigned long long prof[16], tot = 0;
		(void)hipDeviceSynchronize();
		(void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_tl_prof), sizeof(prof));
		for (int k = 0; k < 16; k++) tot += prof[k];
		fprintf(stderr, "trials profile:");
		for (int k = 0; k < 10; k++) fprintf(stderr, " %d:%.1f%%", k, 100.0 * (double)prof[k] / (double)(tot ? tot : 1));
		fprintf(stderr, "\n");
		static unsigned long long z[16];
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_prof), z, sizeof(z));
	}
#endif
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_uap_table_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
				      uint16_t *d_table, void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!n_packets)
		return BTBBX_OK;
	if ((uintptr_t)d_table & 15) {
		set_error("btbbx_uap_table_device: table must be 16-byte aligned");
		return BTBBX_E_ARG;
	}
	hipLaunchKernelGGL(uap_table_kernel, dim3((n_packets + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream,
			   d_packets, d_in, n_packets, (uint4 *)d_table);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

int launch_decode_bytes(const uint8_t *d_sym, uint8_t *d_pay, const btbbx_pkt_in *d_in, btbbx_pkt_out *d_out,
			uint32_t mode, bool with_payload, hipStream_t stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	hipLaunchKernelGGL(decode_bytes_kernel, dim3(1), dim3(64), 0, stream, d_sym, d_pay, d_in, d_out, mode,
			   with_payload ? 1 : 0);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

int launch_decode(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
		  btbbx_pkt_out *d_out, uint32_t mode, const TrialPlan *plan, hipStream_t stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!n_packets)
		return BTBBX_OK;
	TrialPlan p = {0, 0, 0};
	if (plan)
		p = *plan;
	if (mode & DEC_TRIALS) {                 // single try_clock / crc_check calls of the drop-in API
		if (mode != DEC_TRIALS || n_packets != 1) {
			set_error("decode: trial replay is a one-packet mode");
			return BTBBX_E_ARG;
		}
		hipLaunchKernelGGL(replay_kernel, dim3(1), dim3(64), 0, stream, d_packets, d_in, d_out, p);
		HIP_TRY(hipGetLastError());
		return BTBBX_OK;
	}
	hipLaunchKernelGGL(decode_kernel, dim3((n_packets + 63) / 64), dim3(64), 0, stream,
			   d_packets, d_in, n_packets, d_out, mode);
	HIP_TRY(hipGetLastError());
	return BTBBX_OK;
}

extern "C" int btbbx_decode_device(const uint64_t *d_packets, const btbbx_pkt_in *d_in, uint32_t n_packets,
				   btbbx_pkt_out *d_out, void *hip_stream)
{
	return launch_decode(d_packets, d_in, n_packets, d_out, DEC_HEADER | DEC_PAYLOAD, nullptr, (hipStream_t)hip_stream);
}

// decode_hits_kernel with the list its lanes leave for its own lane-group phase (DM / DH / EV payloads beyond 256 bits, sixteen
// bytes per packet).  The list lives in a block from the device's stream-ordered pool -- asked for and given back on the
// caller's stream, so concurrent callers on other streams share nothing and nothing is synchronised; the pool keeps what it has
// (release threshold raised once).  A runtime or device without such a pool still decodes: the kernel takes a null list, and
// every lane then walks its payload itself as in round 3 (slower for multi-slot packets, same results).
static int launch_decode_hits(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words, const btbbx_hit *d_hits,
			      const btbbx_pkt_in *d_in, uint32_t n_packets, const uint32_t *d_count, uint32_t max_length,
			      btbbx_pkt_out *d_out, uint32_t *d_lengths, const btbbx_pkt_in &one_in, uint32_t clk_div, hipStream_t stream)
{
	const uint32_t groups = (uint32_t)(((uint64_t)n_packets + 255) / 256);
	void *block = nullptr;
	{
		static std::atomic<uint64_t> pool_ready{0}, pool_absent{0};
		int dev = 0;
		HIP_TRY(hipGetDevice(&dev));
		const bool tracked = dev >= 0 && dev < 64;
		bool usable = !tracked || !((pool_absent.load() >> dev) & 1);
		if (usable && tracked && !((pool_ready.load() >> dev) & 1)) {
			hipMemPool_t pool;
			uint64_t keep = UINT64_MAX;
			if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess &&
			    hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) == hipSuccess) {
				pool_ready.fetch_or(1ULL << dev);
			} else {
				(void)hipGetLastError();
				pool_absent.fetch_or(1ULL << dev);
				usable = false;
			}
		}
		if (usable && hipMallocAsync(&block, (size_t)groups * 4 * 64 * sizeof(uint4), stream) != hipSuccess) {
			(void)hipGetLastError();
			block = nullptr;
		}
	}
	hipLaunchKernelGGL(decode_hits_kernel, dim3(groups), dim3(256), 0, stream, d_words, n_words, pitch_words, d_hits, d_in, n_packets,
			   d_count, max_length, d_out, d_lengths, DEC_HEADER | DEC_PAYLOAD, one_in, clk_div, (uint4 *)block);
	hipError_t e = hipGetLastError();
	if (block) {
		const hipError_t f = hipFreeAsync(block, stream);
		if (e == hipSuccess)
			e = f;
	}
	HIP_TRY(e);
	return BTBBX_OK;
}

extern "C" int btbbx_decode_hits_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
					const btbbx_hit *d_hits, const btbbx_pkt_in *d_in, uint32_t n_packets,
					uint32_t max_length, btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!n_packets)
		return BTBBX_OK;
	if (!d_words || !d_hits || !d_in || !d_out) {
		set_error("btbbx_decode_hits_device: null pointer");
		return BTBBX_E_ARG;
	}
	return launch_decode_hits(d_words, n_words, pitch_words, d_hits, d_in, n_packets, nullptr, max_length, d_out, d_lengths, btbbx_pkt_in{}, 1u,
				  (hipStream_t)hip_stream);
}

// The same with the number of hits still on the device (the counter btbbx_scan_device filled): decodes
// min(*d_count, cap) packets, launches for `cap`.  With btbbx_order_hits_device in front of it the chain
// scan -> order -> decode runs without a host round trip.
extern "C" int btbbx_decode_hits_counted_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
						const btbbx_hit *d_hits, const btbbx_pkt_in *d_in, const uint32_t *d_count,
						uint32_t cap, uint32_t max_length, btbbx_pkt_out *d_out, uint32_t *d_lengths,
						void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!cap)
		return BTBBX_OK;
	if (!d_words || !d_hits || !d_in || !d_out || !d_count) {
		set_error("btbbx_decode_hits_counted_device: null pointer");
		return BTBBX_E_ARG;
	}
	rc = launch_decode_hits(d_words, n_words, pitch_words, d_hits, d_in, cap, d_count, max_length, d_out, d_lengths, btbbx_pkt_in{}, 1u,
				(hipStream_t)hip_stream);
	if (rc)
		return rc;
#ifdef DH_PROFILE
	{
		unsigned long long prof[8], total = 0;
		HIP_TRY(hipDeviceSynchronize());
		HIP_TRY(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_dh_prof), sizeof(prof)));
		for (int k = 0; k < 8; k++) total += prof[k];
		fprintf(stderr, "decode_hits profile (%% of wave time): loads %.1f type %.1f staging %.1f head-load %.1f present %.1f header %.1f payload %.1f store %.1f; s_memtime ticks per wave %.0f\n",
			100.0 * prof[0] / total, 100.0 * prof[1] / total, 100.0 * prof[2] / total, 100.0 * prof[3] / total,
			100.0 * prof[4] / total, 100.0 * prof[5] / total, 100.0 * prof[6] / total, 100.0 * prof[7] / total,
			(double)total / ((cap + 63) / 64));
		unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dh_prof), zero, sizeof(zero)));
	}
#endif
	return BTBBX_OK;
}

// A capture of ONE piconet, decoded without a btbbx_pkt_in per packet: every packet enters with the state *entry
// describes (flags, UAP, ...; its length field is ignored) and the clock entry->clkn + offset / clk_div -- CLK1-27
// advances once per 625 symbols at 1 Msym/s, so a receiver that knows the clock at the first symbol of its buffer knows
// it for every access code the scan found in it.  The list's length is read from HBM as above (d_count may be NULL:
// then `cap` records are decoded).
extern "C" int btbbx_decode_hits_piconet_phase_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
						      const btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap,
						      const btbbx_pkt_in *entry, uint32_t clk_div, uint32_t clk_phase, uint32_t max_length,
						      btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream)
{
	int rc = ctx_require();
	if (rc)
		return rc;
	if (!cap)
		return BTBBX_OK;
	if (!d_words || !d_hits || !entry || !d_out || !clk_div || clk_phase >= clk_div) {
		set_error("btbbx_decode_hits_piconet_device: null pointer, clk_div = 0 or clk_phase >= clk_div");
		return BTBBX_E_ARG;
	}
	btbbx_pkt_in one = *entry;
	one.length = clk_phase;                     // (the kernel takes the captured length from the stream, the field carries the phase)
	return launch_decode_hits(d_words, n_words, pitch_words, d_hits, nullptr, cap, d_count, max_length, d_out, d_lengths, one, clk_div,
				  (hipStream_t)hip_stream);
}

// ... for a buffer whose first symbol is the first symbol of a slot (clk_phase 0)
extern "C" int btbbx_decode_hits_piconet_device(const uint64_t *d_words, uint64_t n_words, uint64_t pitch_words,
						const btbbx_hit *d_hits, const uint32_t *d_count, uint32_t cap,
						const btbbx_pkt_in *entry, uint32_t clk_div, uint32_t max_length,
						btbbx_pkt_out *d_out, uint32_t *d_lengths, void *hip_stream)
{
	return btbbx_decode_hits_piconet_phase_device(d_words, n_words, pitch_words, d_hits, d_count, cap, entry, clk_div, 0, max_length,
						      d_out, d_lengths, hip_stream);
}
