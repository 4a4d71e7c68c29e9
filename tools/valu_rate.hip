// tools/valu_rate.hip -- micro-benchmark: issue rate of the integer VALU ops the scan kernel
// is made of, on gfx950, at the scan kernel's occupancy (1024-thread workgroups, 1 per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define ITERS 4096
#define UNROLL 32

#define KERNEL(name, body)                                                            \
__global__ __launch_bounds__(1024) void name(uint32_t *out, uint32_t seed)             \
{                                                                                      \
	uint32_t a = threadIdx.x ^ seed, b = a * 3 + 1, c = a + 7, d = b ^ 0x55;            \
	uint32_t e = a + 11, f = b + 13, g = c + 17, h = d + 19;                           \
	for (int i = 0; i < ITERS; i++) {                                                  \
		_Pragma("unroll") for (int u = 0; u < UNROLL / 8; u++) { body }                 \
	}                                                                                  \
	out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;        \
}

// 8 independent chains per unroll step
#define OP8(ins) asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
	ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed));
#define OP8_3(ins) asm volatile(ins " %0, %0, %1, %8\n" ins " %1, %1, %2, %8\n" ins " %2, %2, %3, %8\n" ins " %3, %3, %4, %8\n" \
	ins " %4, %4, %5, %8\n" ins " %5, %5, %6, %8\n" ins " %6, %6, %7, %8\n" ins " %7, %7, %0, %8\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed));
#define OP8_1(ins) asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" \
	ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));

KERNEL(k_and, OP8("v_and_b32"))
KERNEL(k_xor, OP8("v_xor_b32"))
KERNEL(k_add, OP8("v_add_u32"))
KERNEL(k_lshr, OP8("v_lshrrev_b32"))
KERNEL(k_alignbit, OP8_3("v_alignbit_b32"))
KERNEL(k_bfe, OP8_3("v_bfe_u32"))
#define OP8_C(ins, lit) asm volatile(ins " %0, %0, %1, " lit "\n" ins " %1, %1, %2, " lit "\n" ins " %2, %2, %3, " lit "\n" ins " %3, %3, %4, " lit "\n" \
	ins " %4, %4, %5, " lit "\n" ins " %5, %5, %6, " lit "\n" ins " %6, %6, %7, " lit "\n" ins " %7, %7, %0, " lit "\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
KERNEL(k_alignbit_const, OP8_C("v_alignbit_b32", "7"))
KERNEL(k_ffbh, OP8_1("v_ffbh_u32"))
KERNEL(k_mov, OP8_1("v_mov_b32"))
KERNEL(k_lshl_add, OP8_3("v_lshl_add_u32"))
KERNEL(k_and_or, OP8_3("v_and_or_b32"))
KERNEL(k_ffbl, OP8_1("v_ffbl_b32"))
KERNEL(k_bcnt, OP8("v_bcnt_u32_b32"))
KERNEL(k_mul_lo, OP8_3("v_mad_u32_u24"))
KERNEL(k_fma, OP8_3("v_fma_f32"))
KERNEL(k_perm, OP8_3("v_perm_b32"))
KERNEL(k_alignbyte, OP8_3("v_alignbyte_b32"))
KERNEL(k_cndmask, OP8("v_cndmask_b32"))
KERNEL(k_or3, OP8_3("v_or3_b32"))
KERNEL(k_xad, OP8_3("v_xad_u32"))
KERNEL(k_lshl_or, OP8_3("v_lshl_or_b32"))
KERNEL(k_bfi, OP8_3("v_bfi_b32"))
KERNEL(k_mbcnt, OP8("v_mbcnt_lo_u32_b32"))
KERNEL(k_bfm, OP8("v_bfm_b32"))
KERNEL(k_movdpp, asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)
__global__ __launch_bounds__(1024) void k_bitop3(uint32_t *out, uint32_t seed)
{
	uint32_t a = threadIdx.x ^ seed, b = a * 3 + 1, c = a + 7, d = b ^ 0x55;
	uint32_t e = a + 11, f = b + 13, g = c + 17, h = d + 19;
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int u = 0; u < UNROLL / 8; u++) {
			asm volatile("v_bitop3_b32 %0, %0, %1, %8 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %8 bitop3:0x96\n"
				     "v_bitop3_b32 %2, %2, %3, %8 bitop3:0x96\n v_bitop3_b32 %3, %3, %4, %8 bitop3:0x96\n"
				     "v_bitop3_b32 %4, %4, %5, %8 bitop3:0x96\n v_bitop3_b32 %5, %5, %6, %8 bitop3:0x96\n"
				     "v_bitop3_b32 %6, %6, %7, %8 bitop3:0x96\n v_bitop3_b32 %7, %7, %0, %8 bitop3:0x96\n"
				     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed));
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}

// 64-bit shift: four independent chains, twice per body (8 instructions like the others)
__global__ __launch_bounds__(1024) void k_lshr64(uint32_t *out, uint32_t seed)
{
	uint64_t A = threadIdx.x ^ seed, B = A * 3 + 1, C = A + 7, D = B ^ 0x55;
	for (int i = 0; i < ITERS; i++) {
#pragma unroll
		for (int u = 0; u < UNROLL / 8; u++)
			asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n"
				     "v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3\n"
				     : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(seed));
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(A ^ B ^ C ^ D) ^ (uint32_t)((A ^ B ^ C ^ D) >> 32);
}
// two-operand op with a 32-bit literal (8-byte encoding) and with the same constant in an SGPR
#define OP8_LIT(ins, lit) asm volatile(ins " %0, " lit ", %0\n" ins " %1, " lit ", %1\n" ins " %2, " lit ", %2\n" ins " %3, " lit ", %3\n" \
	ins " %4, " lit ", %4\n" ins " %5, " lit ", %5\n" ins " %6, " lit ", %6\n" ins " %7, " lit ", %7\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
#define OP8_SGPR(ins) asm volatile(ins " %0, %8, %0\n" ins " %1, %8, %1\n" ins " %2, %8, %2\n" ins " %3, %8, %3\n" \
	ins " %4, %8, %4\n" ins " %5, %8, %5\n" ins " %6, %8, %6\n" ins " %7, %8, %7\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(seed));
KERNEL(k_or_lit, OP8_LIT("v_or_b32", "0x12345"))
KERNEL(k_or_sgpr, OP8_SGPR("v_or_b32"))
KERNEL(k_or_inline, OP8_LIT("v_or_b32", "1"))
#define OP8_SDWA(ins) asm volatile(ins " %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
	ins " %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" ins " %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
	ins " %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" ins " %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
	ins " %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" ins " %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
	ins " %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed));
KERNEL(k_lshr_sdwa, OP8_SDWA("v_lshrrev_b32_sdwa"))
KERNEL(k_and_sdwa, OP8_SDWA("v_and_b32_sdwa"))


// compares: to vcc (VOP2-sized encoding) and to an SGPR pair (VOP3 encoding) -- the pass of scan_slide_kernel ends in one per chain
#define CMP8(dst) asm volatile("v_cmp_gt_i32 " dst ", 0, %0\n v_cmp_gt_i32 " dst ", 0, %1\n v_cmp_gt_i32 " dst ", 0, %2\n v_cmp_gt_i32 " dst ", 0, %3\n" \
	"v_cmp_gt_i32 " dst ", 0, %4\n v_cmp_gt_i32 " dst ", 0, %5\n v_cmp_gt_i32 " dst ", 0, %6\n v_cmp_gt_i32 " dst ", 0, %7\n" \
	: : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h) : "vcc", "s20", "s21");
KERNEL(k_cmp_vcc, CMP8("vcc"))
KERNEL(k_cmp_sgpr, CMP8("s[20:21]"))
// (the same with the lane mask consumed by the scalar unit, as the pass does)
KERNEL(k_cmp_vcc_sor, asm volatile("v_cmp_gt_i32 vcc, 0, %0\n s_or_b64 s[22:23], s[22:23], vcc\n v_cmp_gt_i32 vcc, 0, %1\n s_or_b64 s[22:23], s[22:23], vcc\n"
	"v_cmp_gt_i32 vcc, 0, %2\n s_or_b64 s[22:23], s[22:23], vcc\n v_cmp_gt_i32 vcc, 0, %3\n s_or_b64 s[22:23], s[22:23], vcc\n"
	"v_cmp_gt_i32 vcc, 0, %4\n s_or_b64 s[22:23], s[22:23], vcc\n v_cmp_gt_i32 vcc, 0, %5\n s_or_b64 s[22:23], s[22:23], vcc\n"
	"v_cmp_gt_i32 vcc, 0, %6\n s_or_b64 s[22:23], s[22:23], vcc\n v_cmp_gt_i32 vcc, 0, %7\n s_or_b64 s[22:23], s[22:23], vcc\n"
	: : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h) : "vcc", "scc", "s22", "s23");)

// ---- source-operand VGPR banks (round 5): does it matter whether two or three sources of one instruction sit in the
// same register bank (register number mod 4)?  Eight independent instructions per body on fixed registers: sources
// v8..v19, destinations v20..v27, no dependence between them, so the figure is pure issue rate.
#define BANK_KERNEL(name, i0, i1, i2, i3, i4, i5, i6, i7)                                  \
__global__ __launch_bounds__(1024) void name(uint32_t *out, uint32_t seed)                  \
{                                                                                           \
	uint32_t acc = threadIdx.x ^ seed;                                                      \
	asm volatile("v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n" \
		     "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"   \
		     "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n"   \
		     : : "v"(acc) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19"); \
	for (int i = 0; i < ITERS; i++) {                                                       \
		_Pragma("unroll") for (int u = 0; u < UNROLL / 8; u++)                              \
			asm volatile(i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n" : : :    \
				     "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
				     "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "vcc", "s20", "s21", "s22", "s23");  \
	}                                                                                       \
	asm volatile("v_xor_b32 %0, %0, v20\n v_xor_b32 %0, %0, v27" : "+v"(acc) : : "v20", "v27"); \
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                       \
}
#define B3(op, d, a, b, c, tail) op " v" #d ", v" #a ", v" #b ", v" #c tail
#define B2(op, d, a, b) op " v" #d ", v" #a ", v" #b
// three sources in three banks / two in one bank / all three in one bank
BANK_KERNEL(k_bitop3_b012, B3("v_bitop3_b32", 20, 8, 9, 10, " bitop3:0x96"), B3("v_bitop3_b32", 21, 9, 10, 11, " bitop3:0x96"),
	    B3("v_bitop3_b32", 22, 12, 13, 14, " bitop3:0x96"), B3("v_bitop3_b32", 23, 13, 14, 15, " bitop3:0x96"),
	    B3("v_bitop3_b32", 24, 16, 17, 18, " bitop3:0x96"), B3("v_bitop3_b32", 25, 17, 18, 19, " bitop3:0x96"),
	    B3("v_bitop3_b32", 26, 10, 11, 12, " bitop3:0x96"), B3("v_bitop3_b32", 27, 14, 15, 16, " bitop3:0x96"))
BANK_KERNEL(k_bitop3_b001, B3("v_bitop3_b32", 20, 8, 12, 9, " bitop3:0x96"), B3("v_bitop3_b32", 21, 9, 13, 10, " bitop3:0x96"),
	    B3("v_bitop3_b32", 22, 10, 14, 11, " bitop3:0x96"), B3("v_bitop3_b32", 23, 11, 15, 12, " bitop3:0x96"),
	    B3("v_bitop3_b32", 24, 12, 16, 13, " bitop3:0x96"), B3("v_bitop3_b32", 25, 13, 17, 14, " bitop3:0x96"),
	    B3("v_bitop3_b32", 26, 14, 18, 15, " bitop3:0x96"), B3("v_bitop3_b32", 27, 15, 19, 16, " bitop3:0x96"))
BANK_KERNEL(k_bitop3_b000, B3("v_bitop3_b32", 20, 8, 12, 16, " bitop3:0x96"), B3("v_bitop3_b32", 21, 9, 13, 17, " bitop3:0x96"),
	    B3("v_bitop3_b32", 22, 10, 14, 18, " bitop3:0x96"), B3("v_bitop3_b32", 23, 11, 15, 19, " bitop3:0x96"),
	    B3("v_bitop3_b32", 24, 8, 12, 16, " bitop3:0x96"), B3("v_bitop3_b32", 25, 9, 13, 17, " bitop3:0x96"),
	    B3("v_bitop3_b32", 26, 10, 14, 18, " bitop3:0x96"), B3("v_bitop3_b32", 27, 11, 15, 19, " bitop3:0x96"))
BANK_KERNEL(k_xor_b01, B2("v_xor_b32", 20, 8, 9), B2("v_xor_b32", 21, 9, 10), B2("v_xor_b32", 22, 10, 11), B2("v_xor_b32", 23, 11, 12),
	    B2("v_xor_b32", 24, 12, 13), B2("v_xor_b32", 25, 13, 14), B2("v_xor_b32", 26, 14, 15), B2("v_xor_b32", 27, 15, 16))
BANK_KERNEL(k_xor_b00, B2("v_xor_b32", 20, 8, 12), B2("v_xor_b32", 21, 9, 13), B2("v_xor_b32", 22, 10, 14), B2("v_xor_b32", 23, 11, 15),
	    B2("v_xor_b32", 24, 12, 16), B2("v_xor_b32", 25, 13, 17), B2("v_xor_b32", 26, 14, 18), B2("v_xor_b32", 27, 15, 19))
BANK_KERNEL(k_alignbit_b012, B3("v_alignbit_b32", 20, 8, 9, 10, ""), B3("v_alignbit_b32", 21, 9, 10, 11, ""),
	    B3("v_alignbit_b32", 22, 12, 13, 14, ""), B3("v_alignbit_b32", 23, 13, 14, 15, ""),
	    B3("v_alignbit_b32", 24, 16, 17, 18, ""), B3("v_alignbit_b32", 25, 17, 18, 19, ""),
	    B3("v_alignbit_b32", 26, 10, 11, 12, ""), B3("v_alignbit_b32", 27, 14, 15, 16, ""))
BANK_KERNEL(k_alignbit_b000, B3("v_alignbit_b32", 20, 8, 12, 16, ""), B3("v_alignbit_b32", 21, 9, 13, 17, ""),
	    B3("v_alignbit_b32", 22, 10, 14, 18, ""), B3("v_alignbit_b32", 23, 11, 15, 19, ""),
	    B3("v_alignbit_b32", 24, 8, 12, 16, ""), B3("v_alignbit_b32", 25, 9, 13, 17, ""),
	    B3("v_alignbit_b32", 26, 10, 14, 18, ""), B3("v_alignbit_b32", 27, 11, 15, 19, ""))
// a funnel shift by a constant reads two registers: same bank / different banks
BANK_KERNEL(k_alignbitc_b01, "v_alignbit_b32 v20, v8, v9, 7", "v_alignbit_b32 v21, v9, v10, 7", "v_alignbit_b32 v22, v10, v11, 7",
	    "v_alignbit_b32 v23, v11, v12, 7", "v_alignbit_b32 v24, v12, v13, 7", "v_alignbit_b32 v25, v13, v14, 7",
	    "v_alignbit_b32 v26, v14, v15, 7", "v_alignbit_b32 v27, v15, v16, 7")
BANK_KERNEL(k_alignbitc_b00, "v_alignbit_b32 v20, v8, v12, 7", "v_alignbit_b32 v21, v9, v13, 7", "v_alignbit_b32 v22, v10, v14, 7",
	    "v_alignbit_b32 v23, v11, v15, 7", "v_alignbit_b32 v24, v12, v16, 7", "v_alignbit_b32 v25, v13, v17, 7",
	    "v_alignbit_b32 v26, v14, v18, 7", "v_alignbit_b32 v27, v15, v19, 7")
// the running-shift pass of round 5: a 64-bit shift by a per-lane amount
BANK_KERNEL(k_lshr64_reg, "v_lshrrev_b64 v[20:21], v8, v[12:13]", "v_lshrrev_b64 v[22:23], v9, v[14:15]", "v_lshrrev_b64 v[24:25], v10, v[16:17]",
	    "v_lshrrev_b64 v[26:27], v11, v[18:19]", "v_lshrrev_b64 v[20:21], v8, v[12:13]", "v_lshrrev_b64 v[22:23], v9, v[14:15]",
	    "v_lshrrev_b64 v[24:25], v10, v[16:17]", "v_lshrrev_b64 v[26:27], v11, v[18:19]")


// ---- round 6, third session: the 16-bit and packed forms a 16-bit-entry set would need, and a few more plain ones
KERNEL(k_lshl, OP8("v_lshlrev_b32"))
KERNEL(k_sub, OP8("v_sub_u32"))
KERNEL(k_min, OP8("v_min_u32"))
KERNEL(k_ashr, OP8("v_ashrrev_i32"))
KERNEL(k_mul24, OP8("v_mul_u32_u24"))
KERNEL(k_not, OP8_1("v_not_b32"))
KERNEL(k_bfrev, OP8_1("v_bfrev_b32"))
KERNEL(k_lshl16, OP8("v_lshlrev_b16"))
KERNEL(k_add16, OP8("v_add_u16"))
#define OP8_PK(ins, mods) asm volatile(ins " %0, %8, %0 " mods "\n" ins " %1, %8, %1 " mods "\n" ins " %2, %8, %2 " mods "\n" ins " %3, %8, %3 " mods "\n" \
	ins " %4, %8, %4 " mods "\n" ins " %5, %8, %5 " mods "\n" ins " %6, %8, %6 " mods "\n" ins " %7, %8, %7 " mods "\n" \
	: "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed));
KERNEL(k_pk_lshl16, OP8_PK("v_pk_lshlrev_b16", ""))
KERNEL(k_pk_lshl16_sel, OP8_PK("v_pk_lshlrev_b16", "op_sel:[1,0] op_sel_hi:[1,0]"))
KERNEL(k_pk_add16, OP8_PK("v_pk_add_u16", ""))
#define CMP8X(ins, dst) asm volatile(ins " " dst ", 0, %0\n " ins " " dst ", 0, %1\n " ins " " dst ", 0, %2\n " ins " " dst ", 0, %3\n" \
	ins " " dst ", 0, %4\n " ins " " dst ", 0, %5\n " ins " " dst ", 0, %6\n " ins " " dst ", 0, %7\n" \
	: : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h) : "vcc", "s20", "s21");
KERNEL(k_cmp16_vcc, CMP8X("v_cmp_gt_i16", "vcc"))
KERNEL(k_cmp16_sgpr, CMP8X("v_cmp_gt_i16", "s[20:21]"))
KERNEL(k_cmpne_vcc, CMP8X("v_cmp_ne_u32", "vcc"))


// v_cndmask_b32: the chained form above reads 23 cycles; these say whether that is the instruction or the probe
BANK_KERNEL(k_cndmask_indep, "v_cndmask_b32 v20, v8, v9, vcc", "v_cndmask_b32 v21, v9, v10, vcc", "v_cndmask_b32 v22, v10, v11, vcc",
	    "v_cndmask_b32 v23, v11, v12, vcc", "v_cndmask_b32 v24, v12, v13, vcc", "v_cndmask_b32 v25, v13, v14, vcc",
	    "v_cndmask_b32 v26, v14, v15, vcc", "v_cndmask_b32 v27, v15, v16, vcc")
BANK_KERNEL(k_cndmask_e64, "v_cndmask_b32_e64 v20, v8, v9, s[20:21]", "v_cndmask_b32_e64 v21, v9, v10, s[20:21]", "v_cndmask_b32_e64 v22, v10, v11, s[20:21]",
	    "v_cndmask_b32_e64 v23, v11, v12, s[20:21]", "v_cndmask_b32_e64 v24, v12, v13, s[20:21]", "v_cndmask_b32_e64 v25, v13, v14, s[20:21]",
	    "v_cndmask_b32_e64 v26, v14, v15, s[20:21]", "v_cndmask_b32_e64 v27, v15, v16, s[20:21]")
BANK_KERNEL(k_cndmask_const, "v_cndmask_b32_e64 v20, 0, 1, s[20:21]", "v_cndmask_b32_e64 v21, 0, 1, s[20:21]", "v_cndmask_b32_e64 v22, 0, 1, s[20:21]",
	    "v_cndmask_b32_e64 v23, 0, 1, s[20:21]", "v_cndmask_b32_e64 v24, 0, 1, s[20:21]", "v_cndmask_b32_e64 v25, 0, 1, s[20:21]",
	    "v_cndmask_b32_e64 v26, 0, 1, s[20:21]", "v_cndmask_b32_e64 v27, 0, 1, s[20:21]")
BANK_KERNEL(k_bitop3_select, "v_bitop3_b32 v20, v8, v9, v10 bitop3:0xca", "v_bitop3_b32 v21, v9, v10, v11 bitop3:0xca", "v_bitop3_b32 v22, v10, v11, v12 bitop3:0xca",
	    "v_bitop3_b32 v23, v11, v12, v13 bitop3:0xca", "v_bitop3_b32 v24, v12, v13, v14 bitop3:0xca", "v_bitop3_b32 v25, v13, v14, v15 bitop3:0xca",
	    "v_bitop3_b32 v26, v14, v15, v16 bitop3:0xca", "v_bitop3_b32 v27, v15, v16, v17 bitop3:0xca")
BANK_KERNEL(k_lshl_const, "v_lshlrev_b32 v20, 2, v8", "v_lshlrev_b32 v21, 2, v9", "v_lshlrev_b32 v22, 2, v10", "v_lshlrev_b32 v23, 2, v11",
	    "v_lshlrev_b32 v24, 2, v12", "v_lshlrev_b32 v25, 2, v13", "v_lshlrev_b32 v26, 2, v14", "v_lshlrev_b32 v27, 2, v15")
BANK_KERNEL(k_lshr_const, "v_lshrrev_b32 v20, 2, v8", "v_lshrrev_b32 v21, 2, v9", "v_lshrrev_b32 v22, 2, v10", "v_lshrrev_b32 v23, 2, v11",
	    "v_lshrrev_b32 v24, 2, v12", "v_lshrrev_b32 v25, 2, v13", "v_lshrrev_b32 v26, 2, v14", "v_lshrrev_b32 v27, 2, v15")
BANK_KERNEL(k_lshl_add_const, "v_lshl_add_u32 v20, v8, 2, v9", "v_lshl_add_u32 v21, v9, 2, v10", "v_lshl_add_u32 v22, v10, 2, v11", "v_lshl_add_u32 v23, v11, 2, v12",
	    "v_lshl_add_u32 v24, v12, 2, v13", "v_lshl_add_u32 v25, v13, 2, v14", "v_lshl_add_u32 v26, v14, 2, v15", "v_lshl_add_u32 v27, v15, 2, v16")
BANK_KERNEL(k_mul_u24_const, "v_mul_u32_u24 v20, 4, v8", "v_mul_u32_u24 v21, 4, v9", "v_mul_u32_u24 v22, 4, v10", "v_mul_u32_u24 v23, 4, v11",
	    "v_mul_u32_u24 v24, 4, v12", "v_mul_u32_u24 v25, 4, v13", "v_mul_u32_u24 v26, 4, v14", "v_mul_u32_u24 v27, 4, v15")


// compare + select pairs as the compiler writes them: through vcc (VOP2 select) and through an SGPR pair (VOP3 select)
BANK_KERNEL(k_cmp_sel_vcc, "v_cmp_gt_u32 vcc, v8, v9", "v_cndmask_b32 v20, v10, v11, vcc", "v_cmp_gt_u32 vcc, v12, v13", "v_cndmask_b32 v21, v14, v15, vcc",
	    "v_cmp_gt_u32 vcc, v9, v10", "v_cndmask_b32 v22, v11, v12, vcc", "v_cmp_gt_u32 vcc, v13, v14", "v_cndmask_b32 v23, v15, v16, vcc")
BANK_KERNEL(k_cmp_sel_sgpr, "v_cmp_gt_u32 s[20:21], v8, v9", "v_cndmask_b32_e64 v20, v10, v11, s[20:21]", "v_cmp_gt_u32 s[22:23], v12, v13", "v_cndmask_b32_e64 v21, v14, v15, s[22:23]",
	    "v_cmp_gt_u32 s[20:21], v9, v10", "v_cndmask_b32_e64 v22, v11, v12, s[20:21]", "v_cmp_gt_u32 s[22:23], v13, v14", "v_cndmask_b32_e64 v23, v15, v16, s[22:23]")
BANK_KERNEL(k_addc_vcc, "v_add_co_u32 v20, vcc, v8, v9", "v_addc_co_u32 v21, vcc, v10, v11, vcc", "v_add_co_u32 v22, vcc, v12, v13", "v_addc_co_u32 v23, vcc, v14, v15, vcc",
	    "v_add_co_u32 v24, vcc, v9, v10", "v_addc_co_u32 v25, vcc, v11, v12, vcc", "v_add_co_u32 v26, vcc, v13, v14", "v_addc_co_u32 v27, vcc, v15, v16, vcc")

template <typename K>
static void run(const char *name, K kernel, uint32_t *d_out, int waves_per_simd)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	// up to 4 waves per SIMD in one 1024-thread workgroup per CU; 8 = two such workgroups per CU
	int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd), blocks = 256 * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 1u);
	hipEventRecord(e0);
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 2u);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	double wave_instrs_per_simd = (double)ITERS * UNROLL * waves_per_simd;      // each SIMD hosts waves_per_simd waves
	double cycles = ms * 1e-3 * 2.4e9;
	printf("%-26s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n",
	       name, waves_per_simd, ms, cycles / wave_instrs_per_simd);
}

int main()
{
	uint32_t *d_out;
	hipMalloc(&d_out, 512 * 1024 * 4);
	for (int w = 4; w <= 8; w *= 2) {
		run("v_and_b32", k_and, d_out, w);
		run("v_xor_b32", k_xor, d_out, w);
		run("v_add_u32", k_add, d_out, w);
		run("v_lshrrev_b32", k_lshr, d_out, w);
		run("v_alignbit_b32", k_alignbit, d_out, w);
		run("v_bfe_u32", k_bfe, d_out, w);
		run("v_alignbit const", k_alignbit_const, d_out, w);
		run("v_ffbh_u32", k_ffbh, d_out, w);
		run("v_mov_b32", k_mov, d_out, w);
		run("v_lshl_add_u32", k_lshl_add, d_out, w);
		run("v_and_or_b32", k_and_or, d_out, w);
		run("v_bitop3_b32", k_bitop3, d_out, w);
		run("v_ffbl_b32", k_ffbl, d_out, w);
		run("v_bcnt_u32_b32", k_bcnt, d_out, w);
		run("v_mad_u32_u24", k_mul_lo, d_out, w);
		run("v_fma_f32", k_fma, d_out, w);
		run("v_perm_b32", k_perm, d_out, w);
		run("v_lshrrev_b64", k_lshr64, d_out, w);
		run("v_or_b32 literal", k_or_lit, d_out, w);
		run("v_or_b32 sgpr", k_or_sgpr, d_out, w);
		run("v_or_b32 inline", k_or_inline, d_out, w);
		run("v_lshrrev_b32_sdwa", k_lshr_sdwa, d_out, w);
		run("v_and_b32_sdwa", k_and_sdwa, d_out, w);
		run("v_alignbyte_b32", k_alignbyte, d_out, w);
		run("v_cndmask_b32", k_cndmask, d_out, w);
		run("v_or3_b32", k_or3, d_out, w);
		run("v_xad_u32", k_xad, d_out, w);
		run("v_lshl_or_b32", k_lshl_or, d_out, w);
		run("v_bfi_b32", k_bfi, d_out, w);
		run("v_mbcnt_lo", k_mbcnt, d_out, w);
		run("v_bfm_b32", k_bfm, d_out, w);
		run("v_mov_dpp", k_movdpp, d_out, w);
		run("v_cmp -> vcc", k_cmp_vcc, d_out, w);
		run("v_cmp -> sgpr pair", k_cmp_sgpr, d_out, w);
		run("v_cmp -> vcc + s_or", k_cmp_vcc_sor, d_out, w);
		run("bitop3 banks 0,1,2", k_bitop3_b012, d_out, w);
		run("bitop3 banks 0,0,1", k_bitop3_b001, d_out, w);
		run("bitop3 banks 0,0,0", k_bitop3_b000, d_out, w);
		run("xor banks 0,1", k_xor_b01, d_out, w);
		run("xor banks 0,0", k_xor_b00, d_out, w);
		run("alignbit banks 0,1,2", k_alignbit_b012, d_out, w);
		run("alignbit banks 0,0,0", k_alignbit_b000, d_out, w);
		run("alignbit const banks 0,1", k_alignbitc_b01, d_out, w);
		run("alignbit const banks 0,0", k_alignbitc_b00, d_out, w);
		run("lshrrev_b64 by vgpr", k_lshr64_reg, d_out, w);
		run("v_lshlrev_b32", k_lshl, d_out, w);
		run("v_sub_u32", k_sub, d_out, w);
		run("v_min_u32", k_min, d_out, w);
		run("v_ashrrev_i32", k_ashr, d_out, w);
		run("v_mul_u32_u24", k_mul24, d_out, w);
		run("v_not_b32", k_not, d_out, w);
		run("v_bfrev_b32", k_bfrev, d_out, w);
		run("v_lshlrev_b16", k_lshl16, d_out, w);
		run("v_add_u16", k_add16, d_out, w);
		run("v_pk_lshlrev_b16", k_pk_lshl16, d_out, w);
		run("v_pk_lshlrev_b16 op_sel", k_pk_lshl16_sel, d_out, w);
		run("v_pk_add_u16", k_pk_add16, d_out, w);
		run("v_cmp_gt_i16 -> vcc", k_cmp16_vcc, d_out, w);
		run("v_cmp_gt_i16 -> sgpr pair", k_cmp16_sgpr, d_out, w);
		run("v_cmp_ne_u32 -> vcc", k_cmpne_vcc, d_out, w);
		run("v_cndmask e32 independent", k_cndmask_indep, d_out, w);
		run("v_cndmask e64 sgpr pair", k_cndmask_e64, d_out, w);
		run("v_cndmask e64 0,1", k_cndmask_const, d_out, w);
		run("v_bitop3 select (0xca)", k_bitop3_select, d_out, w);
		run("v_lshlrev_b32 const", k_lshl_const, d_out, w);
		run("v_lshrrev_b32 const", k_lshr_const, d_out, w);
		run("v_lshl_add_u32 const", k_lshl_add_const, d_out, w);
		run("v_mul_u32_u24 const", k_mul_u24_const, d_out, w);
		run("cmp+cndmask via vcc", k_cmp_sel_vcc, d_out, w);
		run("cmp+cndmask via sgpr pair", k_cmp_sel_sgpr, d_out, w);
		run("add_co+addc via vcc", k_addc_vcc, d_out, w);
	}
	return 0;
}
