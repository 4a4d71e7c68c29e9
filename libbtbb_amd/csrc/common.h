// common.h -- shared declarations of the MI355X baseband scanner (host + device).
//
// Everything is derived from the Bluetooth baseband spec polynomials at start-up
// (tables.cpp); nothing is transcribed from the reference's tables.  Reference
// citations are relative to /root/reference.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/btbbx.h"
#include "slide.h"

// ---- spec constants ---------------------------------------------------------------
#define SW_POLY   0260534236651ULL         // (64,30) block code generator, degree 34
#define SW_PN     0x83848D96BBCC54FCULL    // PN overlay (bluetooth_packet.c:115)
#define BARKER1   0x27u                    // 7-bit window when LAP bit 23 = 1 (host order)
#define BARKER0   0x58u                    // 7-bit window when LAP bit 23 = 0
#define LOW57     0x01ffffffffffffffULL

// ---- scan kernel geometry ---------------------------------------------------------
#define SCAN_THREADS   1024                // one workgroup per CU, 16 wave64
#define SCAN_WAVES     (SCAN_THREADS / 64)
#define TABA_FIRST     34                  // the (64,30) code is systematic: window bits 0..33 are their own syndrome,
#define TABA_BITS      11                  //   only bits 34..56 need tables: 34..44 (tabA) and
#define TABB_BITS      12                  //   45..56 (tabB, with the class-0 barker / PN constant folded in)
#define QRING          128                 // per-wave candidate ring (entries; it never holds more than 127)
#define SCAN_UNROLL    2                   // tiles a wave works on per loop trip (independent LDS chains)
#define PARK_SLOTS     2                   // private candidate slots per lane
#define CAND_BYTES     16                  // a parked candidate: position code + its 64-bit window + pad (one ds_*_b128)

// LDS layout of scan_lap_any_kernel (bytes).  Both table bases fit the 16-bit DS offset immediate, so a
// probe needs no address adds.
#define LDS_TABB_WORDS   (1u << TABB_BITS)            // 4096 u32  = 16 KiB
#define LDS_TABA_WORDS   (1u << TABA_BITS)            // 2048 u32  =  8 KiB
#define LDS_OFF_TABB     0u
#define LDS_OFF_TABA     (LDS_OFF_TABB + 4u * LDS_TABB_WORDS)
#define LDS_OFF_QUEUE    (LDS_OFF_TABA + 4u * LDS_TABA_WORDS)                 // 16 x QRING candidates
#define LDS_OFF_PARK     (LDS_OFF_QUEUE + CAND_BYTES * SCAN_WAVES * QRING)    // 16 x 64 x PARK_SLOTS candidates
#define LDS_OFF_PROF     (LDS_OFF_PARK + CAND_BYTES * SCAN_WAVES * 64u * PARK_SLOTS)   // -DSCAN_PROFILE: 32 counters per wave
#ifdef SCAN_PROFILE
#define SCAN_LDS_BYTES   (LDS_OFF_PROF + 128u * SCAN_WAVES)
#else
#define SCAN_LDS_BYTES   LDS_OFF_PROF                                                   // = 88 KiB
#endif

// ---- device-side table bundle -----------------------------------------------------
struct ScanTables {
	const uint32_t *tabA;      // [2048]   low-32 syndrome of window bits 34..44
	const uint32_t *tabB;      // [4096]   low-32 syndrome of bits 45..56 ^ class-0 constant
	const uint32_t *slide_bitmap;  // 2^SLIDE_BITS-bit set over the sliding checks (slide.h), PN constant folded in
	const uint64_t *hslots;    // open-addressing table of packed (syndrome, positions)
	uint64_t hmask;            // slots - 1
	uint64_t kclass[2];        // full syndrome of (corrected barker | pn), class 0 / 1
	uint32_t kdiff;            // low 32 bits of kclass[0] ^ kclass[1]
	uint64_t hi_mask[2];       // window bits (0..56) whose syndrome has bit 32 / bit 33 set
	const uint32_t *slide4_bitmap;  // tables for four errors: 2^SLIDE4_BITS-bit set over the checks SLIDE4_TAPS (LDS, one workgroup per CU) ...
	const uint32_t *slide4b_bitmap; // ... and the 2^SLIDE4B_BITS-bit set over SLIDE4B_TAPS its members are looked up in (L2), words bit-reversed; else null
	const uint32_t *bitmap2;   // second-level filter in global memory (tables for >= 3 errors), or null
	uint32_t bitmap2_shift;    // index = (low32 * golden) >> shift
};

// Segment slots of the ordered LAP_ANY scan (scan.hip scan_slide_kernel<..., ORD> fills them, sort.hip lays them out and compacts them)
struct ScanSlots {
	uint64_t *slots;           // [segments][slot_n], 8 bytes each (ScanArgs::seg_slots)
	uint32_t seg_offsets;      // offsets a segment covers: 4032 (LAP_ANY: 63 words) or 4096 (known LAP)
	uint16_t *cnt;             // [segments] hits per segment, zeroed before the scan
	uint32_t slot_n;
	uint32_t segs_per_stream;
	btbbx_hit *ovf_recs;       // hits ranked beyond a segment's slots ...
	void *ovf_meta;            // ... with their (segment, rank) as uint2
	uint32_t ovf_cap;
	uint32_t *ovf_count;
	uint32_t *irregular;
};

// packed hash slot: bits 0..33 syndrome, then five 6-bit error positions (63 = unused),
// ascending.  Empty slot = all ones.
#define HSLOT_EMPTY 0xffffffffffffffffULL

struct HostTables {
	uint64_t col[64];          // x^j mod g: syndrome of single bit j
	uint64_t bytetab[8][256];
	uint64_t gen_rows[24];     // generator rows per LAP bit, MSB first
	uint64_t sw_default;       // sync word of LAP 0
	uint8_t  whiten[127];
	uint8_t  whiten_idx[64];
	uint8_t  fec23_par[10];    // parity column of data bit i
	int8_t   fec23_fix[32];    // 5-bit syndrome -> data bit, -1 = none/parity, -2 = fail
};

const HostTables &host_tables();
uint64_t host_gen_syncword(uint32_t lap);
uint64_t host_syndrome(uint64_t cw);

// ---- context ------------------------------------------------------------------------
// One context per HIP device (tables are replicated, nothing is shared between devices).  Every
// entry point works on the context of the calling thread's CURRENT device, so one-process-per-GPU
// callers (hipSetDevice once) and one-process-many-GPUs callers (btbbx_scan_host_multi: one host
// thread per device) go through the same code.
#define BTBBX_MAX_DEVICES 64          // CPX partitioning exposes up to 64 ordinals per node

struct Ctx {
	bool ready = false;
	int device = -1;
	int table_errors = 0;       // max_ac_errors the tables were built for
	int num_cus = 256;
	ScanTables scan{};          // device pointers
	void *d_tab_block = nullptr;
	void *d_hslots = nullptr;
	void *d_bitmap2 = nullptr;
	void *d_retired_tab = nullptr, *d_retired_hslots = nullptr, *d_retired_bitmap2 = nullptr;   // previous table set (context.cpp upload_tables)
};

Ctx &ctx();                              // context of the current device (never null; maybe !ready)
int ctx_require();                       // BTBBX_OK or error (sets last error)
void ctx_scan_snapshot(ScanTables *tables, int *table_errors);   // the current device's scan tables as one consistent set (a re-init may run beside the caller)
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

// Scratch memory, pinned staging and a stream for ONE host call.  A CallScope at the top of an entry
// point leases a buffer set of the current device for the calling thread (nested scopes share the
// outermost lease), so concurrent callers never see each other's staged symbols or results.
struct CallScope {
	CallScope();
	~CallScope();
	CallScope(const CallScope &) = delete;
	CallScope &operator=(const CallScope &) = delete;
};
void *scope_device(size_t bytes);        // grow-only device scratch of the innermost live scope
void *scope_pinned(size_t bytes);        // grow-only pinned host staging
void *scope_hits(size_t bytes);          // grow-only device block for hit records + counter of host-level scans
void *scope_packet_block(void **pinned_mirror, size_t bytes);   // fixed-size device block + pinned mirror
hipStream_t scope_stream();              // private non-blocking stream of the lease

#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_fail(_e, #expr); } while (0)

// device constants for the packet chain (whitening etc.), defined in packet.hip
struct ChainTables {
	uint64_t whiten2[4];       // two periods of the 127-bit whitening sequence, packed (254 bits)
	uint8_t  whiten_idx[64];
	uint8_t  fec23_par[10];
	int8_t   fec23_fix[32];
};
int chain_upload(const HostTables &t);
