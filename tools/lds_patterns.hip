// lds_patterns.hip -- microbenchmark: LDS cycles per wave64 instruction for the access patterns of the seed stages' tables, so
// that SQ_LDS_BANK_CONFLICT (40 % of the LDS-active cycles of map_kernel) can be pinned on tables:
//   rot[]  ds_read_b128, 16 entries of 16 B in one 256-byte row, lanes pick at random  (stage B, one per step)
//   quad[] ds_read_b128, 256 entries of 16 B, random                                    (window hash: stage B's first window, stage R)
//   lut[]  ds_read_u16, 1024 entries of 2 B, random                                     (stage A's homopolymer look-up, 16 per super-row)
//   same entry for all lanes (broadcast) and lane-linear addresses as the conflict-free references; ds_write_b16 at a stride of
//   32 B per lane (stage B's flag words) and ds_or_b32 at neighbouring dwords (stage A's code stream).
// Four waves of one workgroup alone on a CU (one per SIMD, every wave up to eight accesses in flight: the LDS pipe is the limit,
// and only lanes of one instruction can conflict); reported: CU cycles per instruction = time of the workgroup / (4 x 512).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_patterns tools/lds_patterns.hip ; run: tools/bin/lds_patterns
// Diagnostic tool only; not part of the product path.
#include <hip/hip_runtime.h>

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

constexpr int ITER = 64, UNROLL = 8, WAVES = 4;

template <int MODE>
__global__ void k(const uint32_t *addr, unsigned long long *out, uint32_t *sink) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[8192];  // 32 KB
    for (int i = threadIdx.x; i < 8192; i += 64 * WAVES) lds[i] = i * 2654435761u;
    uint32_t a[UNROLL];
    for (int u = 0; u < UNROLL; ++u) a[u] = addr[u * 64 + (threadIdx.x & 63)] + (MODE >= 2 ? (threadIdx.x >> 6) * 8192u : 0u);  // writers: a region per wave
    __syncthreads();
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (MODE == 0) {  // ds_read_b128
                uint4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a[u]) : "memory");
                acc ^= v.x;
            } else if (MODE == 1) {  // ds_read_u16
                uint32_t v;
                asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a[u]) : "memory");
                acc ^= v;
            } else if (MODE == 2) {  // ds_write_b16
                asm volatile("ds_write_b16 %0, %1" ::"v"(a[u]), "v"(acc) : "memory");
            } else {  // ds_or_b32
                asm volatile("ds_or_b32 %0, %1" ::"v"(a[u]), "v"(acc | 1u) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    sink[threadIdx.x] = acc ^ lds[threadIdx.x];
}

int main() {
    uint32_t *d_addr, *d_sink;
    unsigned long long *d_out;
    CHECK(hipMalloc(&d_addr, UNROLL * 64 * 4));
    CHECK(hipMalloc(&d_sink, 64 * WAVES * 4));
    CHECK(hipMalloc(&d_out, 8));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# lds_patterns: %d waves of one workgroup alone on a CU, %d instructions per wave and measurement (clock %d MHz)\n", WAVES, ITER * UNROLL, prop.clockRate / 1000);
    srand(7);
    auto run = [&](const char *name, int mode, auto gen) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            std::vector<uint32_t> h(UNROLL * 64);
            for (int u = 0; u < UNROLL; ++u)
                for (int l = 0; l < 64; ++l) h[u * 64 + l] = gen(u, l);
            CHECK(hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            for (int w = 0; w < 3; ++w) {
                if (mode == 0) k<0><<<1, 64 * WAVES>>>(d_addr, d_out, d_sink);
                else if (mode == 1) k<1><<<1, 64 * WAVES>>>(d_addr, d_out, d_sink);
                else if (mode == 2) k<2><<<1, 64 * WAVES>>>(d_addr, d_out, d_sink);
                else k<3><<<1, 64 * WAVES>>>(d_addr, d_out, d_sink);
                CHECK(hipDeviceSynchronize());
                unsigned long long t;
                CHECK(hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost));
                const double cyc = (double)t / (ITER * UNROLL * WAVES);  // s_memtime counts shader clocks here
                if (cyc < best) best = cyc;
            }
        }
        printf("%-78s %6.1f cycles per instruction\n", name, best);
    };
    run("ds_read_b128, every lane the same entry (broadcast)", 0, [](int, int) { return 0u; });
    run("ds_read_b128, lane-linear (lane * 16 B)", 0, [](int, int l) { return (uint32_t)l * 16u; });
    run("ds_read_b128, rot[]: random among 16 entries of one 256-byte row", 0, [](int u, int) { return (uint32_t)(u * 256 + (rand() & 15) * 16); });
    run("ds_read_b128, quad[]: random among 256 entries (4 KB)", 0, [](int, int) { return (uint32_t)((rand() & 255) * 16); });
    run("ds_read_u16, lane-linear (lane * 2 B)", 1, [](int, int l) { return (uint32_t)l * 2u; });
    run("ds_read_u16, lut[]: random among 1024 entries (2 KB)", 1, [](int, int) { return (uint32_t)((rand() & 1023) * 2); });
    run("ds_write_b16, lane-linear (lane * 2 B)", 2, [](int, int l) { return (uint32_t)l * 2u; });
    run("ds_write_b16, stage B's flag words: lane * 32 B", 2, [](int u, int l) { return (uint32_t)(l * 32 + u * 2); });
    run("ds_or_b32, lane-linear dwords", 3, [](int, int l) { return (uint32_t)l * 4u; });
    run("ds_or_b32, stage A's code stream: lane l at dword l * 3 / 4 (neighbours share a dword)", 3, [](int u, int l) { return (uint32_t)((l * 3 / 4 + u * 64) * 4); });
    return 0;
}
