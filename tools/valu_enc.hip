// valu_enc.hip -- microbenchmark: does the ENCODING of a vector instruction (4-byte VOP1/VOP2/VOPC against 8-byte VOP3 / SDWA / DPP /
// literal forms) set its issue cost on gfx950?  Follow-up of tools/valu_occ.hip, whose VOP2 stream (v_xor_b32) issued three times
// faster than its VOP3 stream (v_alignbit_b32) at map_kernel's occupancy.  Same harness: cycles per wave64 instruction per SIMD for
// waves-per-SIMD-inside-a-workgroup x workgroups-per-CU layouts.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_enc tools/valu_enc.hip
//   run  : tools/bin/valu_enc > profiles/r04_valu_enc.txt
// Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int ITER = 400;

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define BODY64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)
#define OPS_DECL uint32_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7
#define OPS_IO "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define OPS_SUM (r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7)

#define STAMP(T0)                                                                                          \
    if (OPS_SUM == 0x12345u) out[0] = 1;                                                                   \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - T0;

#define DEF_KERNEL(NAME, BODY)                                                                        \
    __global__ void NAME(uint32_t seed, uint32_t other, unsigned long long *out) {                    \
        extern __shared__ uint32_t lds[];                                                             \
        OPS_DECL;                                                                                     \
        lds[threadIdx.x] = seed;                                                                      \
        __syncthreads();                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < ITER; ++it) asm volatile(BODY : OPS_IO : "v"(other), "s"(seed) : "memory", "vcc", "scc", "s20", "s21", "s22", "s23"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        STAMP(t0)                                                                                     \
    }

#define I_XOR_E32(r) "v_xor_b32_e32 " r ", " r ", %8\n"
#define I_XOR_E64(r) "v_xor_b32_e64 " r ", " r ", %8\n"
#define I_XOR_LIT(r) "v_xor_b32_e32 " r ", 0x12345679, " r "\n"
#define I_XOR_SGPR(r) "v_xor_b32_e32 " r ", %9, " r "\n"
#define I_XOR_INL(r) "v_xor_b32_e32 " r ", 1, " r "\n"
#define I_AND(r) "v_and_b32_e32 " r ", " r ", %8\n"
#define I_OR(r) "v_or_b32_e32 " r ", " r ", %8\n"
#define I_ADD(r) "v_add_u32_e32 " r ", " r ", %8\n"
#define I_SUB(r) "v_sub_u32_e32 " r ", " r ", %8\n"
#define I_ADDCO(r) "v_add_co_u32_e32 " r ", vcc, " r ", %8\n"
#define I_LSHLV(r) "v_lshlrev_b32_e32 " r ", %8, " r "\n"
#define I_MAX(r) "v_max_u32_e32 " r ", " r ", %8\n"
#define I_XOR_OTHER(r) "v_xor_b32_e32 " r ", %8, %8\n"
#define I_BFE(r) "v_bfe_u32 " r ", " r ", 3, 7\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %8, %8\n"
#define I_CNDMASK_S(r) "v_cndmask_b32_e64 " r ", " r ", %8, s[20:21]\n"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", %8, " r "\n"
#define I_MUL24(r) "v_mul_u32_u24_e32 " r ", " r ", %8\n"
#define I_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 7\n"
#define I_LSHL(r) "v_lshlrev_b32_e32 " r ", 1, " r "\n"
#define I_MOV(r) "v_mov_b32_e32 " r ", %8\n"
#define I_AND_SDWA(r) "v_and_b32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_ADD_DPP(r) "v_add_u32_dpp " r ", " r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADD3(r) "v_add3_u32 " r ", " r ", %8, 1\n"
#define I_LSHL_OR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define I_MIN(r) "v_min_u32_e32 " r ", " r ", %8\n"
#define I_CMP_E32(r) "v_cmp_lt_u32_e32 vcc, " r ", %8\n"
#define I_CMP_E64(r) "v_cmp_lt_u32_e64 s[20:21], " r ", %8\n"
#define I_ADDC(r) "v_addc_co_u32_e32 " r ", vcc, " r ", " r ", vcc\n"
#define I_SALU(r) "s_add_u32 s20, s20, 1\n"
#define I_DEP(r) "v_xor_b32_e32 %0, %0, %8\n"
#define I_DEP3(r) "v_alignbit_b32 %0, %0, %8, 7\n"
// three VOP2 then one VOP3
#define I_3TO1(r) "v_xor_b32_e32 " r ", " r ", %8\n"
#define MIX_3TO1 "v_xor_b32_e32 %0, %0, %8\nv_xor_b32_e32 %1, %1, %8\nv_xor_b32_e32 %2, %2, %8\nv_alignbit_b32 %3, %3, %8, 7\n" \
                 "v_xor_b32_e32 %4, %4, %8\nv_xor_b32_e32 %5, %5, %8\nv_xor_b32_e32 %6, %6, %8\nv_alignbit_b32 %7, %7, %8, 7\n"
#define MIX8(M) M M M M M M M M
#define MIX_1TO1 "v_xor_b32_e32 %0, %0, %8\nv_alignbit_b32 %1, %1, %8, 7\nv_xor_b32_e32 %2, %2, %8\nv_alignbit_b32 %3, %3, %8, 7\n" \
                 "v_xor_b32_e32 %4, %4, %8\nv_alignbit_b32 %5, %5, %8, 7\nv_xor_b32_e32 %6, %6, %8\nv_alignbit_b32 %7, %7, %8, 7\n"
// one VOP3 among seven VOP2
#define MIX_7TO1 "v_xor_b32_e32 %0, %0, %8\nv_xor_b32_e32 %1, %1, %8\nv_xor_b32_e32 %2, %2, %8\nv_xor_b32_e32 %3, %3, %8\n" \
                 "v_xor_b32_e32 %4, %4, %8\nv_xor_b32_e32 %5, %5, %8\nv_xor_b32_e32 %6, %6, %8\nv_alignbit_b32 %7, %7, %8, 7\n"
// VOP2 with one SALU per four
#define MIX_SALU "v_xor_b32_e32 %0, %0, %8\nv_xor_b32_e32 %1, %1, %8\nv_xor_b32_e32 %2, %2, %8\ns_add_u32 s20, s20, 1\nv_xor_b32_e32 %3, %3, %8\n" \
                 "v_xor_b32_e32 %4, %4, %8\nv_xor_b32_e32 %5, %5, %8\nv_xor_b32_e32 %6, %6, %8\ns_add_u32 s21, s21, 1\nv_xor_b32_e32 %7, %7, %8\n"
// the stage-B step as built (11 VALU), and with its VOP3 / VCC parts replaced one at a time
#define STEP_B                                     \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32 vcc, %8, %4\n"                   \
    "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"         \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
#define STEP_B_NOVCC                               \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_xor_b32 %6, %6, %4\n"                       \
    "v_xor_b32 %6, %6, %5\n"                       \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
#define STEP_B_NOVOP3                              \
    "v_lshlrev_b32 %4, 13, %1\n"                   \
    "v_lshrrev_b32 %5, 19, %2\n"                   \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32 vcc, %8, %4\n"                   \
    "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"         \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
#define STEP_B_ALLVOP2                             \
    "v_lshlrev_b32 %4, 13, %1\n"                   \
    "v_lshrrev_b32 %5, 19, %2\n"                   \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_xor_b32 %6, %6, %4\n"                       \
    "v_xor_b32 %6, %6, %5\n"                       \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"

// candidates for the step's test: (a) as built; (b) the two high words by v_lshlrev + v_lshrrev + v_or each (VOP2 only) -- the same
// arithmetic as v_alignbit; (c) cmp into an SGPR pair + addc from it (VOP3 forms); (d) bound test by subtraction: (bhi - m) sign bit
#define STEP_B_SHIFTS                              \
    "v_lshlrev_b32 %4, 13, %1\n"                   \
    "v_lshrrev_b32 %5, 19, %0\n"                   \
    "v_or_b32 %4, %4, %5\n"                        \
    "v_lshrrev_b32 %5, 19, %2\n"                   \
    "v_lshlrev_b32 %6, 13, %3\n"                   \
    "v_or_b32 %5, %5, %6\n"                        \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32 vcc, %8, %4\n"                   \
    "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"         \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
#define STEP_B_SUBSIGN                             \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_sub_u32 %4, %8, %4\n"                       \
    "v_lshrrev_b32 %4, 31, %4\n"                   \
    "v_add_u32 %6, %6, %6\n"                       \
    "v_or_b32 %6, %6, %4\n"                        \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
// the four xors alone / the test alone: what each half of the step costs by itself
#define STEP_B_XORS "v_xor_b32 %0, %0, %5\nv_xor_b32 %1, %1, %7\nv_xor_b32 %2, %2, %5\nv_xor_b32 %3, %3, %7\n"
#define STEP_B_TEST "v_alignbit_b32 %4, %1, %0, 13\nv_alignbit_b32 %5, %2, %3, 19\nv_min_u32 %4, %4, %5\nv_cmp_ge_u32 vcc, %8, %4\nv_addc_co_u32 %6, vcc, %6, %6, vcc\n"
// (e) cmp into an SGPR pair, the four xors, then addc from the pair: the flag pair no longer back to back
#define STEP_B_SPLIT                               \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32_e64 s[20:21], %8, %4\n"          \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"                       \
    "v_addc_co_u32_e64 %6, s[22:23], %6, %6, s[20:21]\n"
// (f) the same with vcc (e32 forms), the pair apart
#define STEP_B_SPLITVCC                            \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32 vcc, %8, %4\n"                   \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"                       \
    "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"
// (g) min3 with a clamp constant, sub, then ONE alignbit shifts the sign bit into the flag word (12 VALU... 11: no cmp/addc)
#define STEP_B_SUBALIGN                            \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min3_u32 %4, %4, %5, %8\n"                  \
    "v_sub_u32 %4, %8, %4\n"                       \
    "v_alignbit_b32 %6, %6, %4, 31\n"              \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
// (h) min3 clamp, sub, lshr, add, or (13 VALU)
#define STEP_B_SUBSIGN3                            \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min3_u32 %4, %4, %5, %8\n"                  \
    "v_sub_u32 %4, %8, %4\n"                       \
    "v_lshrrev_b32 %4, 31, %4\n"                   \
    "v_add_u32 %6, %6, %6\n"                       \
    "v_or_b32 %6, %6, %4\n"                        \
    "v_lshrrev_b32 %5, 8, %7\n"                    \
    "v_and_b32 %5, 0xf0, %5\n"                     \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
// (i) as built, with the real step's SDWA offset instead of lshr + and (10 VALU)
#define STEP_B_REAL                                \
    "v_alignbit_b32 %4, %1, %0, 13\n"              \
    "v_alignbit_b32 %5, %2, %3, 19\n"              \
    "v_min_u32 %4, %4, %5\n"                       \
    "v_cmp_ge_u32 vcc, %8, %4\n"                   \
    "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"         \
    "v_and_b32_sdwa %5, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" \
    "v_xor_b32 %0, %0, %5\n"                       \
    "v_xor_b32 %1, %1, %7\n"                       \
    "v_xor_b32 %2, %2, %5\n"                       \
    "v_xor_b32 %3, %3, %7\n"
DEF_KERNEL(k_stepb_split, MIX8(STEP_B_SPLIT))
DEF_KERNEL(k_stepb_splitvcc, MIX8(STEP_B_SPLITVCC))
DEF_KERNEL(k_stepb_subalign, MIX8(STEP_B_SUBALIGN))
DEF_KERNEL(k_stepb_subsign3, MIX8(STEP_B_SUBSIGN3))
DEF_KERNEL(k_stepb_real, MIX8(STEP_B_REAL))
DEF_KERNEL(k_stepb_shifts, MIX8(STEP_B_SHIFTS))
DEF_KERNEL(k_stepb_subsign, MIX8(STEP_B_SUBSIGN))
DEF_KERNEL(k_stepb_xors, MIX8(STEP_B_XORS))
DEF_KERNEL(k_stepb_test, MIX8(STEP_B_TEST))
DEF_KERNEL(k_xor_e32, BODY64(I_XOR_E32))
DEF_KERNEL(k_xor_e64, BODY64(I_XOR_E64))
DEF_KERNEL(k_xor_lit, BODY64(I_XOR_LIT))
DEF_KERNEL(k_xor_sgpr, BODY64(I_XOR_SGPR))
DEF_KERNEL(k_alignbit, BODY64(I_ALIGNBIT))
DEF_KERNEL(k_xor_inl, BODY64(I_XOR_INL))
DEF_KERNEL(k_and, BODY64(I_AND))
DEF_KERNEL(k_or, BODY64(I_OR))
DEF_KERNEL(k_add, BODY64(I_ADD))
DEF_KERNEL(k_sub, BODY64(I_SUB))
DEF_KERNEL(k_addco, BODY64(I_ADDCO))
DEF_KERNEL(k_lshlv, BODY64(I_LSHLV))
DEF_KERNEL(k_max, BODY64(I_MAX))
DEF_KERNEL(k_bfe, BODY64(I_BFE))
DEF_KERNEL(k_perm, BODY64(I_PERM))
DEF_KERNEL(k_mul24, BODY64(I_MUL24))
DEF_KERNEL(k_lshl, BODY64(I_LSHL))
DEF_KERNEL(k_mov, BODY64(I_MOV))
DEF_KERNEL(k_sdwa, BODY64(I_AND_SDWA))
DEF_KERNEL(k_dpp, BODY64(I_ADD_DPP))
DEF_KERNEL(k_add3, BODY64(I_ADD3))
DEF_KERNEL(k_lshl_or, BODY64(I_LSHL_OR))
DEF_KERNEL(k_min, BODY64(I_MIN))
DEF_KERNEL(k_cmp32, BODY64(I_CMP_E32))
DEF_KERNEL(k_cmp64, BODY64(I_CMP_E64))
DEF_KERNEL(k_addc, BODY64(I_ADDC))
DEF_KERNEL(k_salu, BODY64(I_SALU))
DEF_KERNEL(k_dep2, BODY64(I_DEP))
DEF_KERNEL(k_dep3, BODY64(I_DEP3))
DEF_KERNEL(k_3to1, MIX8(MIX_3TO1))
DEF_KERNEL(k_1to1, MIX8(MIX_1TO1))
DEF_KERNEL(k_7to1, MIX8(MIX_7TO1))
DEF_KERNEL(k_mixsalu, MIX8(MIX_SALU))
DEF_KERNEL(k_stepb, MIX8(STEP_B))
DEF_KERNEL(k_stepb_novcc, MIX8(STEP_B_NOVCC))
DEF_KERNEL(k_stepb_novop3, MIX8(STEP_B_NOVOP3))
DEF_KERNEL(k_stepb_allvop2, MIX8(STEP_B_ALLVOP2))

struct Test {
    const char *name;
    void (*fn)(uint32_t, uint32_t, unsigned long long *);
    int per_iter;
};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# valu_enc: %s, %d CUs, clock %d MHz; ITER=%d\n", prop.name, n_cu, prop.clockRate / 1000, ITER);
    printf("# cell: cycles per instruction per SIMD (median wave's time / (w * instructions)); layout wpb x bpc = waves per SIMD inside a workgroup x workgroups per CU\n");
    const std::vector<Test> tests = {
        {"v_xor_b32_e32 (VOP2, 4 bytes)", k_xor_e32, 64},
        {"v_xor_b32_e64 (the same operation, VOP3 encoding, 8 bytes)", k_xor_e64, 64},
        {"v_xor_b32_e32 with a 32-bit literal (8 bytes)", k_xor_lit, 64},
        {"v_xor_b32_e32 with an SGPR operand (4 bytes)", k_xor_sgpr, 64},
        {"v_xor_b32_e32 with an inline constant", k_xor_inl, 64},
        {"v_and_b32_e32", k_and, 64},
        {"v_or_b32_e32", k_or, 64},
        {"v_add_u32_e32", k_add, 64},
        {"v_sub_u32_e32", k_sub, 64},
        {"v_add_co_u32_e32 (writes vcc)", k_addco, 64},
        {"v_lshlrev_b32_e32 by a VGPR amount", k_lshlv, 64},
        {"v_max_u32_e32", k_max, 64},
        {"v_bfe_u32 (VOP3)", k_bfe, 64},
        {"v_perm_b32 (VOP3)", k_perm, 64},
        {"v_mul_u32_u24_e32", k_mul24, 64},
        {"v_alignbit_b32 (VOP3, three sources)", k_alignbit, 64},
        {"v_lshlrev_b32_e32 (VOP2, inline constant)", k_lshl, 64},
        {"v_mov_b32_e32 (VOP1)", k_mov, 64},
        {"v_and_b32_sdwa (8 bytes)", k_sdwa, 64},
        {"v_add_u32_dpp row_shr:1 (8 bytes)", k_dpp, 64},
        {"v_add3_u32 (VOP3)", k_add3, 64},
        {"v_lshl_or_b32 (VOP3)", k_lshl_or, 64},
        {"v_min_u32_e32 (VOP2)", k_min, 64},
        {"v_cmp_lt_u32_e32 -> vcc (VOPC, 4 bytes)", k_cmp32, 64},
        {"v_cmp_lt_u32_e64 -> SGPR pair (8 bytes)", k_cmp64, 64},
        {"v_addc_co_u32_e32 (reads and writes vcc)", k_addc, 64},
        {"s_add_u32 (SALU)", k_salu, 64},
        {"v_xor_b32_e32, ONE dependent chain", k_dep2, 64},
        {"v_alignbit_b32, ONE dependent chain", k_dep3, 64},
        {"3 x v_xor_e32 : 1 x v_alignbit", k_3to1, 64},
        {"1 x v_xor_e32 : 1 x v_alignbit", k_1to1, 64},
        {"7 x v_xor_e32 : 1 x v_alignbit", k_7to1, 64},
        {"8 x v_xor_e32 : 2 x s_add_u32 (per vector instruction)", k_mixsalu, 64},
        {"stage-B step as built (11 VALU: 2 VOP3 + cmp/addc on vcc)", k_stepb, 88},
        {"stage-B step, cmp/addc -> two v_xor", k_stepb_novcc, 88},
        {"stage-B step, v_alignbit -> one VOP2 shift (not the same arithmetic)", k_stepb_novop3, 88},
        {"stage-B step, all VOP2, no vcc", k_stepb_allvop2, 88},
        {"stage-B step, high words by shl + shr + or (VOP2 only, same arithmetic), 15 VALU: cycles per STEP / 11", k_stepb_shifts, 88},
        {"stage-B step, flag by sub + shr + add + or instead of cmp/addc, 13 VALU: cycles per STEP / 11", k_stepb_subsign, 88},
        {"stage-B step (e): cmp_e64 -> SGPR pair, xors, addc_e64 from the pair (11 VALU)", k_stepb_split, 88},
        {"stage-B step (f): cmp -> vcc, xors, addc from vcc: the pair apart (11 VALU)", k_stepb_splitvcc, 88},
        {"stage-B step (g): min3 clamp + sub + alignbit into the flag word (11 VALU)", k_stepb_subalign, 88},
        {"stage-B step (h): min3 clamp + sub + shr + add + or (13 VALU): cycles per STEP / 11", k_stepb_subsign3, 88},
        {"stage-B step (i): as built with the SDWA table offset (10 VALU): cycles per STEP / 11", k_stepb_real, 88},
        {"stage-B step: the four xors alone (per instruction)", k_stepb_xors, 32},
        {"stage-B step: the test alone, 2 alignbit + min + cmp + addc (per instruction)", k_stepb_test, 40},
    };
    struct Cfg { int wpb, bpc; };
    const std::vector<Cfg> cfgs = {{1, 1}, {2, 2}, {3, 2}, {4, 2}};
    unsigned long long *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (1 + 8 * 4 * 1024) * sizeof(unsigned long long)));
    for (const Test &t : tests) {
        if (argc > 1) {  // run only the tests whose name contains one of the arguments
            bool hit = false;
            for (int a = 1; a < argc; ++a) hit |= strstr(t.name, argv[a]) != nullptr;
            if (!hit) continue;
        }
        printf("%s\n", t.name);
        for (const Cfg &c : cfgs) {
            const int w = c.wpb * c.bpc, threads = 256 * c.wpb;
            size_t lds = (size_t)(160 * 1024 / c.bpc) & ~(size_t)1023;
            if (lds > 64 * 1024 && c.bpc > 1) lds = 64 * 1024;
            if (c.bpc == 1) lds = 96 * 1024;
            CHECK(hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = n_cu * c.bpc;
            std::vector<double> cyc;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(threads), lds, 0, 12345u + rep, 77u, d_out);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                if (rep == 0) continue;
                std::vector<unsigned long long> h((size_t)grid * c.wpb * 4);
                CHECK(hipMemcpy(h.data(), d_out + 1, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                cyc.push_back((double)h[h.size() / 2] / ((double)ITER * t.per_iter));
            }
            const double cy = *std::min_element(cyc.begin(), cyc.end());
            printf("   w=%d (%dx%d) %6.2f", w, c.wpb, c.bpc, cy / w);
            if (&c == &cfgs.back() || (&c - &cfgs[0]) % 4 == 3) printf("\n");
        }
        fflush(stdout);
    }
    CHECK(hipFree(d_out));
    return 0;
}
