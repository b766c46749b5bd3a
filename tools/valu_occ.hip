// valu_occ.hip -- microbenchmark: cycles per wave64 instruction per SIMD on gfx950 as a function of HOW the waves of a SIMD are
// arranged: waves per SIMD inside one workgroup (wpb) x workgroups per CU (bpc), for every total w = wpb * bpc in 1..8.
// Follow-up of tools/valu_issue.hip (which only ran w = 1, 2, 4 as ONE workgroup and w = 8 as two): decides which occupancy
// (5, 6, 8 waves per SIMD) and which workgroup shape map_kernel should be rebuilt for.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_occ tools/valu_occ.hip
//   run  : tools/bin/valu_occ > profiles/r03_valu_occ.txt
// Bodies: one VOP3 stream (v_alignbit), one VOP2 stream (v_xor), the instruction mix of a stage-B step without its LDS
// read (4 xor, 2 alignbit, min, cmp + addc, shift, and), and the same step with its ds_read_b128 ring.
// Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int ITER = 1500;

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
#define BODY64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)
#define OPS_DECL uint32_t r0 = seed, r1 = seed + 1, r2 = seed + 2, r3 = seed + 3, r4 = seed + 4, r5 = seed + 5, r6 = seed + 6, r7 = seed + 7
#define OPS_IO "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)
#define OPS_SUM (r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7)
#define I_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 7\n"
#define I_XOR(r) "v_xor_b32 " r ", " r ", %8\n"

#define STAMP(T0)                                                                                          \
    if (OPS_SUM == 0x12345u) out[0] = 1;                                                                   \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - T0;

#define DEF_KERNEL(NAME, OP)                                                                          \
    __global__ void NAME(uint32_t seed, uint32_t other, unsigned long long *out) {                    \
        extern __shared__ uint32_t lds[];                                                             \
        OPS_DECL;                                                                                     \
        lds[threadIdx.x] = seed;                                                                      \
        __syncthreads();                                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                   \
        for (int it = 0; it < ITER; ++it) asm volatile(BODY64(OP) : OPS_IO : "v"(other) : "memory"); \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                   \
        STAMP(t0)                                                                                     \
    }
DEF_KERNEL(k_alignbit, I_ALIGNBIT)
DEF_KERNEL(k_xor, I_XOR)
// selects: v_cndmask with vcc (as the compiler emits it for ?:), with an SGPR-pair mask, and the arithmetic select v_bfi
#define I_CND_VCC(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define I_CND_SGPR(r) "v_cndmask_b32_e64 " r ", " r ", %8, s[20:21]\n"
#define I_BFI(r) "v_bfi_b32 " r ", %8, " r ", %8\n"
__global__ void k_cnd_vcc(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    OPS_DECL;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) asm volatile("v_cmp_lt_u32 vcc, %0, %8\n" BODY64(I_CND_VCC) : OPS_IO : "v"(other) : "memory", "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    STAMP(t0)
}
__global__ void k_cnd_sgpr(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    OPS_DECL;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n" BODY64(I_CND_SGPR) : OPS_IO : "v"(other) : "memory", "s20", "s21");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    STAMP(t0)
}
DEF_KERNEL(k_bfi, I_BFI)

// a stage-B step without its table read: 4 xor, 2 alignbit, min, cmp + addc, shift, and = 11 VALU on a 4-register state
#define STEP_MIX                                   \
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
#define MIX8 STEP_MIX STEP_MIX STEP_MIX STEP_MIX STEP_MIX STEP_MIX STEP_MIX STEP_MIX
__global__ void k_mix(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    OPS_DECL;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) asm volatile(MIX8 : OPS_IO : "v"(other) : "memory", "vcc");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    STAMP(t0)
}

// the same step WITH its ds_read_b128 (16-entry table, conflict-free), four look-ups in flight (C++: the compiler schedules it)
__global__ void k_step_lds(uint32_t seed, uint32_t other, unsigned long long *out) {
    extern __shared__ uint32_t lds[];
    uint4 *tab = reinterpret_cast<uint4 *>(lds);
    if (threadIdx.x < 64) tab[threadIdx.x] = make_uint4(seed * threadIdx.x, seed + threadIdx.x, seed ^ threadIdx.x, seed - threadIdx.x);
    __syncthreads();
    uint32_t glo = seed + threadIdx.x, ghi = seed * 3 + threadIdx.x, hlo = seed * 5, hhi = seed * 7 + threadIdx.x;
    uint32_t x = other * 2654435761u + threadIdx.x, fb = 0;
    uint4 tv[4];
    for (int s = 0; s < 4; ++s) tv[s] = tab[(x >> (4 * s)) & 15u];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint32_t fh = __builtin_amdgcn_alignbit(ghi, glo, 32 - (t + 1)), rh = __builtin_amdgcn_alignbit(hlo, hhi, t + 1);
            const uint32_t m = fh < rh ? fh : rh;
            asm("v_cmp_ge_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(fb) : "v"(m), "s"(other) : "vcc");
            const uint4 e = tv[t & 3];
            glo ^= e.x;
            ghi ^= e.y;
            hlo ^= e.z;
            hhi ^= e.w;
            tv[t & 3] = tab[((x >> (2 * t)) & 15u) + 16u * (t & 3)];
        }
        x = x * 5u + fb;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((glo ^ ghi ^ hlo ^ hhi ^ fb) == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Test {
    const char *name;
    void (*fn)(uint32_t, uint32_t, unsigned long long *);
    int per_iter;
};

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# valu_occ: %s, %d CUs, clock %d MHz; ITER=%d\n", prop.name, n_cu, prop.clockRate / 1000, ITER);
    printf("# cell: cycles per instruction per SIMD (median wave's time / (w * instructions)); layout wpb x bpc = waves per SIMD inside a workgroup x workgroups per CU\n");
    const std::vector<Test> tests = {
        {"v_alignbit_b32 (VOP3)", k_alignbit, 64},
        {"v_xor_b32 (VOP2)", k_xor, 64},
        {"v_cndmask_b32 with vcc (one v_cmp per 64)", k_cnd_vcc, 64},
        {"v_cndmask_b32_e64 with an SGPR-pair mask", k_cnd_sgpr, 64},
        {"v_bfi_b32 (arithmetic select)", k_bfi, 64},
        {"stage-B mix, 11 VALU per step, no LDS", k_mix, 88},
        {"stage-B step with ds_read_b128 ring (per VALU, 11 per step)", k_step_lds, 16 * 11},
    };
    struct Cfg { int wpb, bpc; };
    const std::vector<Cfg> cfgs = {{1, 1}, {1, 2}, {2, 1}, {1, 3}, {3, 1}, {1, 4}, {2, 2}, {4, 1}, {1, 5}, {1, 6}, {2, 3}, {3, 2}, {1, 7}, {1, 8}, {2, 4}, {4, 2}};
    unsigned long long *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (1 + 8 * 4 * 1024) * sizeof(unsigned long long)));
    for (const Test &t : tests) {
        printf("%s\n", t.name);
        for (const Cfg &c : cfgs) {
            const int w = c.wpb * c.bpc, threads = 256 * c.wpb;
            // dynamic LDS so that exactly bpc workgroups fit a CU: more than 160 KiB / (bpc + 1), at most 160 KiB / bpc (8 is the wave-slot limit anyway)
            size_t lds = (size_t)(160 * 1024 / c.bpc) & ~(size_t)1023;
            if (lds > 64 * 1024 && c.bpc > 1) lds = 64 * 1024 + 0;
            if (c.bpc == 1) lds = 96 * 1024;
            if (c.bpc == 2) lds = 64 * 1024;
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
    // workgroups whose wave count is no multiple of 4 (10 waves x 2 per CU = 5 per SIMD if the hardware spreads them evenly): the
    // median and the slowest wave against the even layouts above
    for (const Test &t : tests) {
        if (t.per_iter != 88 && t.per_iter != 16 * 11) continue;
        for (int waves : {10, 6}) {
            const int bpc = 2, threads = 64 * waves;
            const size_t lds = 64 * 1024;
            CHECK(hipFuncSetAttribute((const void *)t.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const int grid = n_cu * bpc;
            double med = 1e30, mx = 1e30, mn = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(threads), lds, 0, 12345u + rep, 77u, d_out);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                if (rep == 0) continue;
                std::vector<unsigned long long> h((size_t)grid * waves);
                CHECK(hipMemcpy(h.data(), d_out + 1, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                const double den = (double)ITER * t.per_iter * (waves * bpc / 4.0);
                med = std::min(med, (double)h[h.size() / 2] / den);
                mx = std::min(mx, (double)h.back() / den);
                mn = std::min(mn, (double)h[0] / den);
            }
            printf("%s: %d-wave workgroups x 2 (%.1f waves per SIMD): fastest / median / slowest wave %.2f / %.2f / %.2f cycles per instruction per SIMD\n", t.name, waves,
                   waves * bpc / 4.0, mn, med, mx);
        }
    }
    CHECK(hipFree(d_out));
    return 0;
}
