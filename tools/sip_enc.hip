// sip_enc.hip -- microbenchmark: what does one SipHash round (mq_device.hpp MQ_SIPROUND: four 64-bit adds, four rotations by
// 13/16/21/17, two by 32, four 64-bit xors) cost per wave at map_kernel's occupancy, written several ways?  Follow-up of
// tools/valu_enc.hip (issue cost by encoding).  Same units and layouts as that tool.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/sip_enc tools/sip_enc.hip
//   run  : tools/bin/sip_enc > profiles/r04_sip_enc.txt
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

constexpr int ITER = 200;
constexpr int ROUNDS = 8;  // rounds per loop iteration (straight-line)

// (a) as the product writes it: 64-bit values, the compiler's choice of instructions
struct SipU64 {
    uint64_t v0, v1, v2, v3;
    __device__ __forceinline__ void init(uint32_t s) {
        v0 = 0x736f6d6570736575ULL ^ s;
        v1 = 0x646f72616e646f6dULL + s;
        v2 = 0x6c7967656e657261ULL ^ ((uint64_t)s << 32);
        v3 = 0x7465646279746573ULL + ((uint64_t)s << 17);
    }
    static __device__ __forceinline__ uint64_t rotl(uint64_t x, uint32_t r) { return __builtin_rotateleft64(x, (uint64_t)r); }
    __device__ __forceinline__ void round() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    }
    __device__ __forceinline__ void word(uint64_t m) { v3 ^= m; round(); v0 ^= m; }
    __device__ __forceinline__ uint64_t sum() const { return v0 ^ v1 ^ v2 ^ v3; }
};

// 32-bit halves; ADD = 0: v_add_co + v_addc (inline asm, so that the compiler does not fuse them back into a 64-bit add),
// 1: the compiler's 64-bit add on a packed pair.  ROT = 0: v_alignbit x 2, 1: shl/shr/or (VOP2 only)
struct H2 { uint32_t lo, hi; };
template <int ADD, int ROT>
struct SipHalves {
    H2 v0, v1, v2, v3;
    __device__ __forceinline__ void init(uint32_t s) {
        SipU64 t; t.init(s);
        v0 = {(uint32_t)t.v0, (uint32_t)(t.v0 >> 32)}; v1 = {(uint32_t)t.v1, (uint32_t)(t.v1 >> 32)};
        v2 = {(uint32_t)t.v2, (uint32_t)(t.v2 >> 32)}; v3 = {(uint32_t)t.v3, (uint32_t)(t.v3 >> 32)};
    }
    static __device__ __forceinline__ H2 add(H2 a, H2 b) {
        H2 r;
        if (ADD == 0) {
            asm("v_add_co_u32_e32 %0, vcc, %2, %3\n\tv_addc_co_u32_e32 %1, vcc, %4, %5, vcc" : "=&v"(r.lo), "=v"(r.hi) : "v"(a.lo), "v"(b.lo), "v"(a.hi), "v"(b.hi) : "vcc");
        } else {
            const uint64_t s = (((uint64_t)a.hi << 32) | a.lo) + (((uint64_t)b.hi << 32) | b.lo);
            r.lo = (uint32_t)s; r.hi = (uint32_t)(s >> 32);
        }
        return r;
    }
    template <uint32_t R>
    static __device__ __forceinline__ H2 rotl(H2 a) {  // 0 < R < 32
        H2 r;
        if (ROT == 0) {
            r.hi = __builtin_amdgcn_alignbit(a.hi, a.lo, 32u - R);
            r.lo = __builtin_amdgcn_alignbit(a.lo, a.hi, 32u - R);
        } else {
            uint32_t t0, t1;
            asm("v_lshlrev_b32_e32 %0, %4, %5\n\tv_lshrrev_b32_e32 %1, %6, %7\n\tv_lshlrev_b32_e32 %2, %4, %7\n\tv_lshrrev_b32_e32 %3, %6, %5"
                : "=&v"(r.hi), "=&v"(t0), "=&v"(r.lo), "=&v"(t1) : "n"(R), "v"(a.hi), "n"(32u - R), "v"(a.lo));
            r.hi |= t0; r.lo |= t1;
        }
        return r;
    }
    static __device__ __forceinline__ H2 x(H2 a, H2 b) { return {a.lo ^ b.lo, a.hi ^ b.hi}; }
    static __device__ __forceinline__ H2 swap(H2 a) { return {a.hi, a.lo}; }
    __device__ __forceinline__ void round() {
        v0 = add(v0, v1); v1 = rotl<13>(v1); v1 = x(v1, v0); v0 = swap(v0);
        v2 = add(v2, v3); v3 = rotl<16>(v3); v3 = x(v3, v2);
        v0 = add(v0, v3); v3 = rotl<21>(v3); v3 = x(v3, v0);
        v2 = add(v2, v1); v1 = rotl<17>(v1); v1 = x(v1, v2); v2 = swap(v2);
    }
    __device__ __forceinline__ void word(uint64_t m) {
        const H2 mm = {(uint32_t)m, (uint32_t)(m >> 32)};
        v3 = x(v3, mm); round(); v0 = x(v0, mm);
    }
    __device__ __forceinline__ uint64_t sum() const {
        return (((uint64_t)(v0.hi ^ v1.hi ^ v2.hi ^ v3.hi)) << 32) | (v0.lo ^ v1.lo ^ v2.lo ^ v3.lo);
    }
};

template <class S>
__global__ void k_sip(uint32_t seed, uint32_t other, unsigned long long *out, unsigned long long *sums) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = seed;
    __syncthreads();
    S h;
    h.init(seed + threadIdx.x * other);
    const uint64_t m = ((uint64_t)other << 32) | (threadIdx.x * 2654435761u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) h.word(m + (uint32_t)r);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const uint64_t s = h.sum();
    if (blockIdx.x == 0) sums[threadIdx.x] = s;
    if (s == 0x12345u) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

struct Test {
    const char *name;
    void (*fn)(uint32_t, uint32_t, unsigned long long *, unsigned long long *);
};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# sip_enc: %s, %d CUs; ITER=%d x %d rounds (each: xor in a word, one SipRound, xor)\n", prop.name, n_cu, ITER, ROUNDS);
    printf("# cell: tools/valu_enc.hip's unit (cycles per SIMD, median wave's time / waves per SIMD) per ROUND; layout wpb x bpc\n");
    const std::vector<Test> tests = {
        {"(a) uint64_t arithmetic, the compiler's instructions (the product's MQ_SIPROUND)", k_sip<SipU64>},
        {"(b) 32-bit halves: v_add_co + v_addc, rotations by 2 x v_alignbit, rot 32 = renaming", k_sip<SipHalves<0, 0>>},
        {"(c) 32-bit halves: the compiler's 64-bit add on packed pairs, 2 x v_alignbit", k_sip<SipHalves<1, 0>>},
        {"(d) 32-bit halves: v_add_co + v_addc, rotations by shl/shr/or (VOP2 only)", k_sip<SipHalves<0, 1>>},
        {"(e) 32-bit halves: the compiler's 64-bit add, rotations by shl/shr/or", k_sip<SipHalves<1, 1>>},
    };
    struct Cfg { int wpb, bpc; };
    const std::vector<Cfg> cfgs = {{1, 1}, {2, 2}, {3, 2}, {4, 2}};
    unsigned long long *d_out = nullptr, *d_sums = nullptr;
    CHECK(hipMalloc((void **)&d_out, (1 + 8 * 4 * 1024) * sizeof(unsigned long long)));
    CHECK(hipMalloc((void **)&d_sums, 1024 * sizeof(unsigned long long)));
    std::vector<unsigned long long> ref;
    for (const Test &t : tests) {
        if (argc > 1) {
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
                hipLaunchKernelGGL(t.fn, dim3(grid), dim3(threads), lds, 0, 12345u, 77u, d_out, d_sums);
                CHECK(hipGetLastError());
                CHECK(hipDeviceSynchronize());
                if (rep == 0) continue;
                std::vector<unsigned long long> h((size_t)grid * c.wpb * 4);
                CHECK(hipMemcpy(h.data(), d_out + 1, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                cyc.push_back((double)h[h.size() / 2] / ((double)ITER * ROUNDS));
            }
            if (c.wpb == 1) {  // every variant must compute the same thing
                std::vector<unsigned long long> s(256);
                CHECK(hipMemcpy(s.data(), d_sums, s.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                if (ref.empty()) ref = s;
                else if (ref != s) printf("   RESULT DIFFERS from variant (a)\n");
            }
            const double cy = *std::min_element(cyc.begin(), cyc.end());
            printf("   w=%d (%dx%d) %7.2f", w, c.wpb, c.bpc, cy / w);
        }
        printf("\n");
        fflush(stdout);
    }
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_sums));
    return 0;
}
