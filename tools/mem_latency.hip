// mem_latency.hip -- how long does ONE dependent random 16-byte load take on this device, by footprint and by how many waves are
// doing the same?  (map_kernel's waves wait for such loads: the home bucket of a k-min-mer in the 17-GB index table, the first
// super-row of a read.)  Every wave runs a chain of N dependent loads -- the next address comes out of the loaded value -- and
// times it with s_memtime; all 64 lanes load (addresses 64 B apart inside one 4-KB page, or each lane its own random line).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/mem_latency tools/mem_latency.hip
//   run  : tools/bin/mem_latency > profiles/r04_mem_latency.txt
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

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

// SCATTER = false: the wave's 64 lanes read 64 consecutive 64-byte lines (one 4-KB page per step: a read's super-row);
// SCATTER = true: every lane its own random line (a lane-batch of index probes)
template <bool SCATTER>
__global__ void chase(const uint4 *__restrict__ buf, uint64_t n_lines, int steps, uint64_t seed, unsigned long long *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint64_t x = mix(seed + wave * 77u);
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        uint64_t line;
        if (SCATTER) line = mix(x + lane) % n_lines;
        else line = ((x % (n_lines >> 6)) << 6) + lane;
        const uint4 v = buf[line * 4u];  // 64-byte lines, the first 16 bytes of each
        acc += v.x;
        // the next step's address depends on the loaded value (zero in the buffer: x changes through acc all the same)
        x = mix(x + (uint64_t)__builtin_amdgcn_readfirstlane((int)(v.y + acc)) + (uint64_t)s);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345u) out[0] = 1;
    if (lane == 0) out[1 + wave] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("# mem_latency: %d CUs; chains of dependent 16-byte loads, median wave's s_memtime ticks per load and ns (ticks calibrated against HIP events per run)\n", n_cu);
    unsigned long long *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_out, (1 + 256 * 64) * sizeof(unsigned long long)));
    const int steps = 400;
    for (uint64_t gib4 : {1ull, 16ull, 68ull}) {  // quarter GiBs: 0.25, 4, 17 GiB
        const uint64_t bytes = gib4 << 28;
        uint4 *buf = nullptr;
        CHECK(hipMalloc((void **)&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        const uint64_t n_lines = bytes / 64;
        for (int scatter = 0; scatter < 2; ++scatter)
            for (int wpc : {1, 4, 16}) {  // waves per CU
                const int grid = n_cu, threads = 64 * wpc;
                hipEvent_t e0, e1;
                CHECK(hipEventCreate(&e0));
                CHECK(hipEventCreate(&e1));
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
                    if (scatter) hipLaunchKernelGGL(chase<true>, dim3(grid), dim3(threads), 0, 0, buf, n_lines, steps, 1234567ull + rep, d_out);
                    else hipLaunchKernelGGL(chase<false>, dim3(grid), dim3(threads), 0, 0, buf, n_lines, steps, 1234567ull + rep, d_out);
                    CHECK(hipGetLastError());
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipDeviceSynchronize());
                }
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<unsigned long long> h((size_t)grid * wpc);
                CHECK(hipMemcpy(h.data(), d_out + 1, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(h.begin(), h.end());
                const double ticks_per_ns = (double)h.back() / (ms * 1e6);  // the slowest wave spans (nearly) the whole launch
                printf("%6.2f GiB  %-34s %2d waves/CU: %8.1f ticks = %7.0f ns per load  (%.2f ticks/ns)\n", bytes / 1073741824.0,
                       scatter ? "each lane its own random line" : "64 consecutive lines (one page)", wpc, (double)h[h.size() / 2] / steps,
                       (double)h[h.size() / 2] / steps / ticks_per_ns, ticks_per_ns);
                fflush(stdout);
            }
        CHECK(hipFree(buf));
    }
    CHECK(hipFree(d_out));
    return 0;
}
