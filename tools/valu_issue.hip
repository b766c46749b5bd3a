// valu_issue.hip -- microbenchmark: cycles per wave64 instruction per SIMD on gfx950, at 1/2/4/8 waves per SIMD.
// Settles the "2 or 4 cycles per VALU instruction" question behind DESIGN.md's issue-ceiling estimate for map_kernel.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_issue tools/valu_issue.hip
//   run  : tools/bin/valu_issue > profiles/r02_valu_issue.txt
// Method: every CU gets exactly `w` waves per SIMD (workgroups of 256*min(w,4) threads, dynamic LDS sized so that only
// 1 or 2 workgroups fit a CU); each wave runs ITER iterations of an unrolled body of N instructions between two
// s_memtime stamps.  Reported: cycles per instruction as ONE wave sees it (T / instrs) and per SIMD (T / (w * instrs)).
// Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

constexpr int ITER = 2000;

// eight independent registers r0..r7, one instruction each: dependency distance 8
#define R8(OP)                                                                                              \
    OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define BODY64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define OPS_DECL uint32_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7
#define OPS_IO "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define OPS_SUM (r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7)

#define I_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 7\n"
#define I_XOR(r) "v_xor_b32 " r ", " r ", %8\n"
#define I_MIN(r) "v_min_u32 " r ", " r ", %8\n"
#define I_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %8, %8\n"
#define I_BFE(r) "v_bfe_u32 " r ", " r ", 3, 9\n"
#define I_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define I_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %8\n"
#define I_MUL(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define I_MAD(r) "v_mad_u32_u24 " r ", " r ", %8, %8\n"
#define I_DPP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPWAVE(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", %8, " r "\n"
#define I_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8\n"
#define I_ALIGN_SALU(r) "v_alignbit_b32 " r ", " r ", %8, 7\ns_add_u32 s20, s20, 1\n"
#define I_ALIGN_2SALU(r) "v_alignbit_b32 " r ", " r ", %8, 7\ns_add_u32 s20, s20, 1\ns_xor_b32 s21, s21, s20\n"
#define I_SALU(r) "s_add_u32 s20, s20, 1\n"
#define I_READLANE(r) "v_readlane_b32 s20, " r ", 5\n"
#define I_CMP(r) "v_cmp_le_u32 vcc, " r ", %8\n"
#define I_CMP_BR(r) "v_cmp_le_u32 vcc, " r ", %8\ns_cbranch_vccnz 1f\n1:\n"
#define I_BPERMUTE(r) "ds_bpermute_b32 " r ", %8, " r "\n"

#define DEF_KERNEL(NAME, OP, EXTRA_CLOBBER)                                                              \
    __global__ void NAME(uint32_t seed, uint32_t other, unsigned long long *out) {                       \
        extern __shared__ uint32_t lds[];                                                                \
        OPS_DECL;                                                                                        \
        lds[threadIdx.x] = seed;                                                                         \
        __syncthreads();                                                                                 \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                      \
        for (int it = 0; it < ITER; ++it) asm volatile(BODY64(OP) : OPS_IO : "v"(other) : EXTRA_CLOBBER); \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                      \
        if (OPS_SUM == 0x12345u) out[0] = 1;                                                             \
        if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; \
    }

DEF_KERNEL(k_alignbit, I_ALIGNBIT, "memory")
DEF_KERNEL(k_xor, I_XOR, "memory")
DEF_KERNEL(k_min, I_MIN, "memory")
DEF_KERNEL(k_add, I_ADD, "memory")
DEF_KERNEL(k_perm, I_PERM, "memory")
DEF_KERNEL(k_bfe, I_BFE, "memory")
DEF_KERNEL(k_lshlor, I_LSHLOR, "memory")
DEF_KERNEL(k_andor, I_ANDOR, "memory")
DEF_KERNEL(k_mul, I_MUL, "memory")
DEF_KERNEL(k_mad24, I_MAD, "memory")
DEF_KERNEL(k_dpp, I_DPP, "memory")
DEF_KERNEL(k_dppwave, I_DPPWAVE, "memory")
DEF_KERNEL(k_cndmask, I_CNDMASK, "memory")
DEF_KERNEL(k_mbcnt, I_MBCNT, "memory")
DEF_KERNEL(k_bcnt, I_BCNT, "memory")
#define CL_S "memory", "s20", "s21", "scc"
DEF_KERNEL(k_align_salu, I_ALIGN_SALU, CL_S)
DEF_KERNEL(k_align_2salu, I_ALIGN_2SALU, CL_S)
DEF_KERNEL(k_salu, I_SALU, CL_S)
DEF_KERNEL(k_readlane, I_READLANE, CL_S)
#define CL_V "memory", "vcc", "scc"
DEF_KERNEL(k_cmp, I_CMP, CL_V)
DEF_KERNEL(k_cmp_br, I_CMP_BR, CL_V)
DEF_KERNEL(k_bpermute, I_BPERMUTE, "memory")

// 64-bit add as the compiler emits it (SipHash): v_add_co_u32 + v_addc_co_u32 on four independent pairs
__global__ void k_add64(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    OPS_DECL;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define P64(a, b) "v_add_co_u32 " a ", vcc, " a ", %8\nv_addc_co_u32 " b ", vcc, " b ", %8, vcc\n"
#define B8 P64("%0", "%1") P64("%2", "%3") P64("%4", "%5") P64("%6", "%7")
    for (int it = 0; it < ITER; ++it) asm volatile(B8 B8 B8 B8 B8 B8 B8 B8 : OPS_IO : "v"(other) : "memory", "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (OPS_SUM == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// 64-bit shift (v_lshlrev_b64) on four independent pairs: 32 instructions per body
__global__ void k_shl64(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    unsigned long long a = seed, b = seed + 1, c = seed + 2, d = seed + 3;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define S64 "v_lshlrev_b64 %0, 1, %0\nv_lshlrev_b64 %1, 1, %1\nv_lshlrev_b64 %2, 1, %2\nv_lshlrev_b64 %3, 1, %3\n"
    for (int it = 0; it < ITER; ++it)
        asm volatile(S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 S64 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(other) : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a ^ b ^ c ^ d) == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// the stage-B step of map_kernel in miniature: 13 VALU (4 alignbit, 4 xor, min, cmp + 3 address ops) + one
// ds_read_b128 of a 16-entry table + 1 scalar branch, four look-ups in flight.  64 steps per body.
__global__ void k_rollstep(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    uint4 *tab = reinterpret_cast<uint4 *>(lds);
    if (threadIdx.x < 16) tab[threadIdx.x] = make_uint4(seed * threadIdx.x, seed + threadIdx.x, seed ^ threadIdx.x, seed - threadIdx.x);
    __syncthreads();
    uint32_t flo = seed + threadIdx.x, fhi = seed * 3 + threadIdx.x, rlo = seed * 5, rhi = seed * 7 + threadIdx.x, x = other + threadIdx.x;
    uint32_t cnt = 0;
    uint4 tv[4];
    for (int s = 0; s < 4; ++s) tv[s] = tab[(x >> (4 * s)) & 15u];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const bool cand = (fhi < rhi ? fhi : rhi) <= other;
            if (__ballot(cand)) cnt++;
            const uint32_t nfhi = __builtin_amdgcn_alignbit(fhi, flo, 31), nflo = __builtin_amdgcn_alignbit(flo, fhi, 31);
            const uint32_t nrlo = __builtin_amdgcn_alignbit(rhi, rlo, 1), nrhi = __builtin_amdgcn_alignbit(rlo, rhi, 1);
            const uint4 v = tv[t & 3];
            flo = nflo ^ v.x;
            fhi = nfhi ^ v.y;
            rlo = nrlo ^ v.z;
            rhi = nrhi ^ v.w;
            x = x * 5u + 1u;
            tv[t & 3] = tab[(x >> 7) & 15u];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((flo ^ fhi ^ rlo ^ rhi ^ cnt) == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// alignbit with one ds_read_b128 (conflict-free broadcast-ish table read) after every 13 VALU
__global__ void k_align_lds(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    OPS_DECL;
    lds[threadIdx.x] = seed;
    __syncthreads();
    uint32_t addr = (threadIdx.x & 15u) * 16u;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    v4u sink;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define I_AB9(r) "v_alignbit_b32 " r ", " r ", %9, 7\n"
#define A13(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r) I_AB9(r)
#define AL(r) A13(r) "ds_read_b128 %8, %10\n"
    for (int it = 0; it < ITER; ++it)
        asm volatile(AL("%0") AL("%1") AL("%2") AL("%3") AL("%4") AL("%5") AL("%6") AL("%7") "s_waitcnt lgkmcnt(0)\n"
                     : OPS_IO, "=v"(sink)
                     : "v"(other), "v"(addr)
                     : "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((OPS_SUM ^ sink.x) == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Test {
    const char *name;
    void (*fn)(uint32_t, uint32_t, unsigned long long *);
    int instrs_per_iter;  // instructions of the class being priced
    const char *note;
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# valu_issue: %s, %d CUs, clock %d MHz; ITER=%d, body = 64 instructions unless noted\n", prop.name, n_cu, prop.clockRate / 1000, ITER);
    printf("# columns: waves/SIMD | cycles per instruction seen by one wave (median over waves) | cycles per instruction per SIMD\n");
    const std::vector<Test> tests = {
        {"v_alignbit_b32", k_alignbit, 64, "8 independent registers"},
        {"v_xor_b32", k_xor, 64, ""},
        {"v_min_u32", k_min, 64, ""},
        {"v_add_u32", k_add, 64, ""},
        {"v_perm_b32", k_perm, 64, ""},
        {"v_bfe_u32", k_bfe, 64, ""},
        {"v_lshl_or_b32", k_lshlor, 64, ""},
        {"v_and_or_b32", k_andor, 64, ""},
        {"v_mul_lo_u32", k_mul, 64, ""},
        {"v_mad_u32_u24", k_mad24, 64, ""},
        {"v_cndmask_b32", k_cndmask, 64, ""},
        {"v_mbcnt_lo_u32_b32", k_mbcnt, 64, ""},
        {"v_bcnt_u32_b32", k_bcnt, 64, ""},
        {"v_cmp_le_u32 (vcc)", k_cmp, 64, ""},
        {"v_cmp + s_cbranch_vccnz", k_cmp_br, 64, "pairs; priced per pair"},
        {"v_mov_b32 dpp row_shr:1", k_dpp, 64, ""},
        {"v_mov_b32 dpp wave_shr:1", k_dppwave, 64, ""},
        {"v_readlane_b32", k_readlane, 64, ""},
        {"ds_bpermute_b32", k_bpermute, 64, ""},
        {"v_add_co_u32 + v_addc_co_u32", k_add64, 64, "64-bit add; priced per 32-bit instruction"},
        {"v_lshlrev_b64", k_shl64, 64, ""},
        {"s_add_u32 only", k_salu, 64, "SALU alone"},
        {"v_alignbit + 1 s_add_u32 each", k_align_salu, 64, "priced per VALU; SALU rides along"},
        {"v_alignbit + 2 SALU each", k_align_2salu, 64, "priced per VALU"},
        {"13 v_alignbit + 1 ds_read_b128", k_align_lds, 104, "priced per VALU; 8 LDS reads per body"},
        {"stage-B step (13 VALU + ds_read_b128 + branch)", k_rollstep, 64, "priced per STEP (64 steps per body)"},
    };
    unsigned long long *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (1 + 8 * 4 * 1024) * sizeof(unsigned long long)));
    for (const Test &t : tests) {
        printf("%-46s", t.name);
        for (int w : {1, 2, 4, 8}) {
            const int wpb = std::min(w, 4);           // waves per SIMD inside one workgroup
            const int bpc = w / wpb;                  // workgroups per CU
            const int threads = 256 * wpb;
            const size_t lds = (bpc == 1 ? 96u : 64u) * 1024u;  // > half (or > third) of 160 KiB: exactly bpc workgroups per CU
            CHECK(hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = n_cu * bpc;
            std::vector<double> cyc;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(threads), lds, 0, 12345u + rep, 77u, d_out);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                if (rep == 0) continue;  // warm-up (clock ramp, code fetch)
                std::vector<unsigned long long> h((size_t)grid * wpb * 4);
                CHECK(hipMemcpy(h.data(), d_out + 1, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                cyc.push_back((double)h[h.size() / 2] / ((double)ITER * t.instrs_per_iter));
            }
            const double c = *std::min_element(cyc.begin(), cyc.end());
            printf(" | w=%d %6.2f %6.2f", w, c, c / w);
        }
        printf("   %s\n", t.note);
        fflush(stdout);
    }
    CHECK(hipFree(d_out));
    return 0;
}
